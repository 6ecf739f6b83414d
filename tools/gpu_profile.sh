#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace profile + PMC passes of the bench workload.
# Outputs land in gpurun_out/<tag>/ ; copy the summaries into profiles/ afterwards.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; rm -rf "$OUT"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 600 --warmup 300 --no-cpu-baseline --workers 0 ${BENCH_ARGS:-}"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ktrace" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/ktrace.log" 2>&1
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $ctr --output-format csv -d "$OUT/pmc_$ctr" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_$ctr.log" 2>&1
done
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$OUT/pmc_sq" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_sq.log" 2>&1
python3 "$ROOT/tools/summarize_profile.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
# instruction cache / issue counters of the persistent kernel (own pass; tolerated to fail on a pool without them)
timeout 600 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d "$OUT/pmc_sq2" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_sq2.log" 2>&1
python3 "$ROOT/tools/summarize_profile.py" "$OUT" > "$OUT/summary.txt" 2>&1
tail -40 "$OUT/summary.txt"
