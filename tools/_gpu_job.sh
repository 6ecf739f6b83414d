set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/gpu_profile_all.sh r04 > gpurun_out/r04_all.log 2>&1 || true
tail -2 gpurun_out/r04_all.log
