#!/bin/bash
# Runs on the GPU box: tools/gpu_profile.sh for every launch shape the bench line reports: the default workload,
# its 65 536-env shapes, BASELINE configs 3, 4, 5 (closed loop) and config 5's fused step.  Outputs under gpurun_out/<tag>_<shape>/.
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
run() {  # name, bench args
  echo "=== $1: $2"
  BENCH_ARGS="$2" bash "$ROOT/tools/gpu_profile.sh" "${TAG}_$1" > "$ROOT/gpurun_out/${TAG}_$1.log" 2>&1 || { echo "FAILED $1"; tail -5 "$ROOT/gpurun_out/${TAG}_$1.log"; }
  tail -3 "$ROOT/gpurun_out/${TAG}_$1.log"
}
for shape in ${SHAPES:-config2 config2_65536 config3 config4 config5 config5step}; do
  case $shape in
    config2) run config2 "--large 0" ;;
    config2_65536) run config2_65536 "--envs 65536 --distinct-worlds 4096 --large 0" ;;
    config3) run config3 "--workload config3 --distinct-worlds 512 --large 0" ;;
    config4) run config4 "--workload config4 --distinct-worlds 512 --large 0" ;;
    config5) run config5 "--workload config5 --distinct-worlds 256" ;;     # (launches of 300 steps like the others: the chunk)
    config5step) run config5step "--workload config5-step --distinct-worlds 256 --steps 200 --warmup 100 --prologue 100" ;;
  esac
done
