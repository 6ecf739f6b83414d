set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r4_gputest1.log 2>&1 || { tail -30 gpurun_out/r4_gputest1.log; exit 1; }
tail -3 gpurun_out/r4_gputest1.log
ROUNDS=2 LEG=closed BENCH_ARGS="--workload config4 --distinct-worlds 512 --large 0" bash tools/gpu_ab2.sh ab_base.so libd2d_hip.so > gpurun_out/r4_ab_c4.txt 2>&1; cat gpurun_out/r4_ab_c4.txt
ROUNDS=2 LEG=step BENCH_ARGS="--workload config4 --distinct-worlds 512 --large 0" bash tools/gpu_ab2.sh ab_base.so libd2d_hip.so > gpurun_out/r4_ab_c4s.txt 2>&1; cat gpurun_out/r4_ab_c4s.txt
ROUNDS=1 LEG=closed BENCH_ARGS="--workload config3 --distinct-worlds 512 --large 0" bash tools/gpu_ab2.sh ab_base.so libd2d_hip.so > gpurun_out/r4_ab_c3.txt 2>&1; cat gpurun_out/r4_ab_c3.txt
ROUNDS=1 LEG=step BENCH_ARGS="--workload config3 --distinct-worlds 512 --large 0" bash tools/gpu_ab2.sh ab_base.so libd2d_hip.so > gpurun_out/r4_ab_c3s.txt 2>&1; cat gpurun_out/r4_ab_c3s.txt
