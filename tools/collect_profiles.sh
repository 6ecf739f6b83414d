#!/bin/bash
# collect_profiles.sh <tag> : copies what tools/gpu_profile_all.sh (and tools/gpu_profile_surv.sh) left under gpurun_out/<tag>_<shape>/ into profiles/<tag>/ (tracked):
# summary_<shape>.txt, kernel_stats_<shape>_<leg>.csv (the rocprofv3 --kernel-trace --stats summaries), pmc_<shape>.json, and merges the PMC
# records of all shapes into profiles/pmc_latest.json (what bench.py scales to its own launch).
TAG=${1:-r03}
cd "$(dirname "$0")/.."
mkdir -p profiles/$TAG
files=()
for shape in config2 config2_65536 config3 config4 config5 config5step survivability; do
  d=gpurun_out/${TAG}_$shape
  [ -f $d/pmc_latest.json ] || { echo "missing $d"; continue; }
  cp $d/summary.txt profiles/$TAG/summary_$shape.txt
  cp $d/pmc_latest.json profiles/$TAG/pmc_$shape.json
  for leg in closed step raycast; do [ -f $d/kernel_stats_$leg.csv ] && cp $d/kernel_stats_$leg.csv profiles/$TAG/kernel_stats_${shape}_$leg.csv; done
  [ -f $d/kernel_stats_survivability.csv ] && cp $d/kernel_stats_survivability.csv profiles/$TAG/kernel_stats_survivability.csv   # (tools/gpu_profile_surv.sh)
  files+=(profiles/$TAG/pmc_$shape.json)
done
python3 tools/merge_pmc.py "${files[@]}" > profiles/pmc_latest.json
echo "merged ${#files[@]} shapes into profiles/pmc_latest.json"
