set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_plugins.py -m gpu -x -q 2>&1 | tail -2
run() { python bench.py --no-cpu-baseline --leg closed --large 0 --workers 8 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; print('%.4e' % json.loads(sys.stdin.read())['value'])"; }
for ch in 150 300 600 1200; do
  echo "chunk $ch: c4 $(run --steps 1200 --warmup 300 --chunk $ch --workload config4 --distinct-worlds 512)  c2L $(run --steps 1200 --warmup 300 --chunk $ch --envs 65536 --distinct-worlds 4096) c2 $(run --steps 1200 --warmup 300 --chunk $ch)"
done
