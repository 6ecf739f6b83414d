set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_vs_oracle.py tests/test_gpu_step_random.py tests/test_gpu_guard_pages.py -m gpu -x -q 2>&1 | tail -2
bash tools/ab_stage_times.sh ab_base.so libd2d_hip.so config4 32768 ALL RAYCAST 2>&1 | tail -10
ROUNDS=1 LEG=all BENCH_ARGS="--workload config4 --distinct-worlds 512 --large 0" bash tools/gpu_ab2.sh ab_base.so libd2d_hip.so
