#!/bin/bash
# Runs on the GPU box: same-call A/B of several builds of the library (csrc/<lib>) on the closed loops the round tracks --
# config 2 at 600 / 300 and in the driver's 20 / 5 window, configs 3, 4, 5 and config 2 at 65 536 envs.  tools/ab_closed.sh libA.so libB.so ...
# SHAPES="c2 d c3 c4 c5 c2L" picks a subset.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
L="$@"
for s in ${SHAPES:-c2 d c3 c4 c5 c2L}; do
  case $s in
    c2) echo "== config 2, 600/300"; ROUNDS=2 LEG=closed BENCH_ARGS="--large 0" bash tools/gpu_ab2.sh $L ;;
    d) echo "== config 2, driver window 20/5"
       for r in 1 2; do for lib in $L; do echo -n "$lib "; D2D_LIB=$ROOT/gym-drone2d-activeperception_amd/csrc/$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --leg closed --large 0 2>/dev/null | tail -1 | python -c "import sys,json; print('%.3e' % json.loads(sys.stdin.read())['value'])"; done; done | sort | awk '{a[$1]=a[$1]" "$2} END{for(k in a) print k, a[k]}' | sort ;;
    c3) echo "== config 3"; ROUNDS=1 LEG=closed BENCH_ARGS="--workload config3 --distinct-worlds 512 --large 0" bash tools/gpu_ab2.sh $L ;;
    c4) echo "== config 4"; ROUNDS=1 LEG=closed BENCH_ARGS="--workload config4 --distinct-worlds 512 --large 0" bash tools/gpu_ab2.sh $L ;;
    c5) echo "== config 5"; ROUNDS=1 LEG=closed BENCH_ARGS="--workload config5 --distinct-worlds 256" bash tools/gpu_ab2.sh $L ;;
    c2L) echo "== config 2 at 65 536 envs"; ROUNDS=1 LEG=closed BENCH_ARGS="--envs 65536 --distinct-worlds 4096 --large 0" bash tools/gpu_ab2.sh $L ;;
  esac
done
