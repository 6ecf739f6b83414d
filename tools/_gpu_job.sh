set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_plugins.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
ROUNDS=2 LEG=closed BENCH_ARGS="--large 0" bash tools/gpu_ab2.sh libd2d_hip.so ab_inl.so > gpurun_out/r4_inl_c2.txt 2>&1; cat gpurun_out/r4_inl_c2.txt
for r in 1 2; do for lib in libd2d_hip.so ab_inl.so; do echo -n "$lib 20/5 "; D2D_LIB=$PWD/gym-drone2d-activeperception_amd/csrc/$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --leg closed --large 0 2>/dev/null | tail -1 | python -c "import sys,json; print('%.3e' % json.loads(sys.stdin.read())['value'])"; done; done
ROUNDS=1 LEG=closed BENCH_ARGS="--workload config3 --distinct-worlds 512 --large 0" bash tools/gpu_ab2.sh libd2d_hip.so ab_inl.so > gpurun_out/r4_inl_c3.txt 2>&1; cat gpurun_out/r4_inl_c3.txt
ROUNDS=1 LEG=closed BENCH_ARGS="--workload config4 --distinct-worlds 512 --large 0" bash tools/gpu_ab2.sh libd2d_hip.so ab_inl.so > gpurun_out/r4_inl_c4.txt 2>&1; cat gpurun_out/r4_inl_c4.txt
ROUNDS=1 LEG=closed BENCH_ARGS="--workload config5 --distinct-worlds 256" bash tools/gpu_ab2.sh libd2d_hip.so ab_inl.so > gpurun_out/r4_inl_c5.txt 2>&1; cat gpurun_out/r4_inl_c5.txt
ROUNDS=1 LEG=closed BENCH_ARGS="--envs 65536 --distinct-worlds 4096 --large 0" bash tools/gpu_ab2.sh libd2d_hip.so ab_inl.so > gpurun_out/r4_inl_c2L.txt 2>&1; cat gpurun_out/r4_inl_c2L.txt
