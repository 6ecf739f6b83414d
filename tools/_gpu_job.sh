cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_plugins.py -m gpu -x -q -k "test_closed_loop_matches_oracle" 2>&1 | tail -2
D2D_RANDOM_SEEDS=3500 D2D_RANDOM_BASE=8001000 timeout -k 10 520 python -m pytest tests/test_gpu_plugins_random.py -m gpu -q -n 5 -p no:cacheprovider > gpurun_out/r4_soak_plugins3.log 2>&1
tail -2 gpurun_out/r4_soak_plugins3.log
D2D_RANDOM_SEEDS=2600 D2D_RANDOM_BASE=9000000 timeout -k 10 480 python -m pytest tests/test_gpu_step_random.py -m gpu -q -n 5 -p no:cacheprovider > gpurun_out/r4_soak_step3.log 2>&1
tail -2 gpurun_out/r4_soak_step3.log
