#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + PMC passes of the three legs of bench.py, each leg in its own
# invocation (--leg closed | step | raycast) so that every k_stages row of a trace belongs to one leg.
# Counters are collected in their own runs (never together with a trace).  Outputs land in gpurun_out/<tag>/ ; copy
# summary.txt, kernel_stats_*.csv and pmc_latest.json into profiles/ afterwards.
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; rm -rf "$OUT"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
LEGS=${LEGS:-closed step raycast}
for leg in $LEGS; do
  ARGS="--steps 600 --warmup 300 --no-cpu-baseline --workers 0 --leg $leg ${BENCH_ARGS:-}"
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$leg/ktrace" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/$leg.ktrace.log" 2>&1 || { echo "ktrace $leg failed"; tail -5 "$OUT/$leg.ktrace.log"; exit 1; }
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 600 rocprofv3 --pmc $ctr --output-format csv -d "$OUT/$leg/pmc_$ctr" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/$leg.pmc_$ctr.log" 2>&1 || { echo "pmc $ctr $leg failed"; exit 1; }
  done
  timeout -k 10 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$OUT/$leg/pmc_sq" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/$leg.pmc_sq.log" 2>&1 || { echo "pmc sq $leg failed"; exit 1; }
  echo "leg $leg done"
done
python3 "$ROOT/tools/summarize_profile.py" "$OUT" "${BENCH_ARGS:-}" > "$OUT/summary.txt" 2>&1
tail -60 "$OUT/summary.txt"
