set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4_gputest3.log 2>&1 || { tail -40 gpurun_out/r4_gputest3.log; exit 1; }
tail -4 gpurun_out/r4_gputest3.log
D2D_RANDOM_SEEDS=300 D2D_RANDOM_BASE=2000000 timeout -k 10 600 python -m pytest tests/test_gpu_plugins_random.py -m gpu -x -q 2>&1 | tail -3
