set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r4_gputest4.log 2>&1 || { tail -40 gpurun_out/r4_gputest4.log; exit 1; }
tail -14 gpurun_out/r4_gputest4.log
