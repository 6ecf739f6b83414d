#!/bin/bash
# interleaved A/B: each library is benchmarked ROUNDS times, round-robin, in separate processes
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
ROUNDS=${ROUNDS:-3}
for r in $(seq $ROUNDS); do
  for lib in "$@"; do
    v=$(D2D_LIB=$ROOT/gym-drone2d-activeperception_amd/csrc/$lib timeout 300 python bench.py --steps 400 --warmup 40 --no-cpu-baseline --workers 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f' % d['roofline']['launch_us'])")
    echo "$lib $v"
  done
done | sort | awk '{a[$1]=a[$1]" "$2} END{for(k in a) print k, a[k]}' | sort
