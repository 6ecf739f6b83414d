set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SHAPES="c2 d c2L c4" bash tools/ab_closed.sh ab_base.so ab_s.so libd2d_hip.so 2>&1 | tee gpurun_out/r4_ab_fence2.txt
B=1 D2D_LIB=$PWD/gym-drone2d-activeperception_amd/csrc/libd2d_hip.so python tools/lone_wave.py 2>&1 | head -2
B=1 D2D_LIB=$PWD/gym-drone2d-activeperception_amd/csrc/ab_s.so python tools/lone_wave.py 2>&1 | head -2
