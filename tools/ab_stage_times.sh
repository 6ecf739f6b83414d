#!/bin/bash
# Same-box A/B of two builds with tools/stage_times.py:  tools/ab_stage_times.sh <libA.so> <libB.so> <workload> <B> [case ...]
# (run under gpurun; writes gpurun_out/ab_stage_<workload>.txt)
A=$1; Bl=$2; WL=$3; NB=$4; shift 4
CS=gym-drone2d-activeperception_amd/csrc
out=gpurun_out/ab_stage_$WL.txt; : > $out
for rep in 1 2; do
  for lib in $A $Bl; do
    for c in "$@"; do
      echo -n "$lib rep$rep " >> $out
      D2D_LIB=$PWD/$CS/$lib WORKLOAD=$WL B=$NB WORLDS=64 WORKERS=8 ONLY=$c timeout -k 10 200 python3 tools/stage_times.py 2>/dev/null | grep us/launch >> $out || exit 1
    done
  done
done
cat $out
