#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes (each in its own run, as the guide
# prescribes) of `bench.py --workload survivability`, then tools/summarize_surv.py.  Output: gpurun_out/<tag>_survivability/.
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT="$ROOT/gpurun_out/${TAG}_survivability"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--workload survivability --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ktrace" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/ktrace.log" 2>&1 || { echo "ktrace failed"; tail -5 "$OUT/ktrace.log"; exit 1; }
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d "$OUT/pmc_$ctr" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_$ctr.log" 2>&1 || { echo "pmc $ctr failed"; tail -5 "$OUT/pmc_$ctr.log"; exit 1; }
done
python3 "$ROOT/tools/summarize_surv.py" "$OUT" | tee "$OUT/summary.txt"
