set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_guard_pages.py -m gpu -q -rs --durations=5 2>&1 | tail -15
