cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
D2D_RANDOM_SEEDS=3500 D2D_RANDOM_BASE=4000000 timeout -k 10 520 python -m pytest tests/test_gpu_plugins_random.py -m gpu -q -n 5 -p no:cacheprovider > gpurun_out/r4_soak_plugins.log 2>&1
tail -3 gpurun_out/r4_soak_plugins.log
D2D_RANDOM_SEEDS=2600 D2D_RANDOM_BASE=5000000 timeout -k 10 520 python -m pytest tests/test_gpu_step_random.py -m gpu -q -n 5 -p no:cacheprovider > gpurun_out/r4_soak_step.log 2>&1
tail -3 gpurun_out/r4_soak_step.log
