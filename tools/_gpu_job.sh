set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_plugins.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
D2D_RANDOM_SEEDS=400 D2D_RANDOM_BASE=3000000 timeout -k 10 600 python -m pytest tests/test_gpu_plugins_random.py -m gpu -x -q 2>&1 | tail -2
SHAPES="c2 d c2L c4 c3 c5" bash tools/ab_closed.sh ab_base.so libd2d_hip.so 2>&1 | tee gpurun_out/r4_ab_quick.txt
B=1 D2D_LIB=$PWD/gym-drone2d-activeperception_amd/csrc/libd2d_hip.so python tools/lone_wave.py 2>&1 | grep -v amdgpu | head -1
B=1 D2D_LIB=$PWD/gym-drone2d-activeperception_amd/csrc/ab_base.so python tools/lone_wave.py 2>&1 | grep -v amdgpu | head -1
