cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
D2D_RANDOM_SEEDS=3500 D2D_RANDOM_BASE=6000000 timeout -k 10 540 python -m pytest tests/test_gpu_plugins_random.py -m gpu -q -n 5 -p no:cacheprovider > gpurun_out/r4_soak_plugins2.log 2>&1
tail -2 gpurun_out/r4_soak_plugins2.log
D2D_RANDOM_SEEDS=2600 D2D_RANDOM_BASE=7000000 timeout -k 10 540 python -m pytest tests/test_gpu_step_random.py -m gpu -q -n 5 -p no:cacheprovider > gpurun_out/r4_soak_step2.log 2>&1
tail -2 gpurun_out/r4_soak_step2.log
