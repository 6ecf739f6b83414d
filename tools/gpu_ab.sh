#!/bin/bash
# A/B the bench over several builds of the HIP library (D2D_LIB) in one GPU session.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out
for lib in "$@"; do
  echo "== $lib"
  D2D_LIB=$ROOT/gym-drone2d-activeperception_amd/csrc/$lib timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value %.3e  ms/step %.4f  launch_us %.2f  frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['launch_us'], d['roofline']['frac']))"
  D2D_LIB=$ROOT/gym-drone2d-activeperception_amd/csrc/$lib timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --mode graph 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('graph: value %.3e  ms/step %.4f  launch_us %.2f' % (d['value'], d['ms_per_step'], d['roofline']['launch_us']))"
done
