set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/gpu_profile_all.sh r04 > gpurun_out/r04_all.log 2>&1 || true
tail -3 gpurun_out/r04_all.log
ls gpurun_out | grep r04_ | head -20
