#!/bin/bash
# interleaved A/B of several builds of the HIP library (D2D_LIB): each is benchmarked ROUNDS times, round-robin, in
# separate processes; prints the step kernel's launch time (us) and the closed loop's env-steps/s per round
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
ROUNDS=${ROUNDS:-3}
for r in $(seq $ROUNDS); do
  for lib in "$@"; do
    v=$(D2D_LIB=$ROOT/gym-drone2d-activeperception_amd/csrc/$lib timeout 300 python bench.py --steps 600 --warmup 300 --no-cpu-baseline --workers 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f/%.3e' % (d['step_kernel']['launch_us'], d['value']))")
    echo "$lib $v"
  done
done | sort | awk '{a[$1]=a[$1]" "$2} END{for(k in a) print k, a[k]}' | sort
