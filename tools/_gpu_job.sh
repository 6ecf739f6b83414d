set -e
cd $GRAFT_REPO_ROOT
for lib in ab_base.so ab_soA.so ab_soB.so ab_soC.so ab_soD.so; do echo -n "$lib "; D2D_LIB=$PWD/gym-drone2d-activeperception_amd/csrc/$lib python tools/search_bench.py --envs 1 2>&1 | grep deadlock | cut -c1-150; echo -n "   4096: "; D2D_LIB=$PWD/gym-drone2d-activeperception_amd/csrc/$lib python tools/search_bench.py --envs 4096 2>&1 | grep deadlock | cut -c20-150;  done
