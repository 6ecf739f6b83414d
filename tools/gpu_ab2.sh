#!/bin/bash
# interleaved A/B of several builds of the HIP library (D2D_LIB): each is benchmarked ROUNDS times, round-robin, in
# separate processes; prints per round: step kernel us / raycast stage us / closed loop env-steps/s (LEGS=all), or the
# legs chosen with LEG=step|raycast|closed
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
ROUNDS=${ROUNDS:-3}
LEG=${LEG:-all}
for r in $(seq $ROUNDS); do
  for lib in "$@"; do
    v=$(D2D_LIB=$ROOT/gym-drone2d-activeperception_amd/csrc/$lib timeout 300 python bench.py --steps 600 --warmup 300 --no-cpu-baseline --workers 8 --leg $LEG ${BENCH_ARGS:-} 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read())
s=d.get('step_kernel',{}).get('roofline',{}).get('launch_us'); r=d.get('raycast_stage',{}).get('roofline',{}).get('launch_us'); v=d.get('value')
print('/'.join(['%.2f' % s if s else '-', '%.2f' % r if r else '-', '%.3e' % v if v else '-']))")
    echo "$lib $v"
  done
done | sort | awk '{a[$1]=a[$1]" "$2} END{for(k in a) print k, a[k]}' | sort
