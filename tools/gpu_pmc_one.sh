#!/bin/bash
# gpu_pmc_one.sh <stage> [<stage> ...]: instruction counters per wave of stage subsets launched alone (tools/stage_times.py ONLY=<stage>)
# on the library D2D_LIB names (default: the shipped one)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_one; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for st in "$@"; do
  ONLY=$st timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d $OUT/$st -- python3 $ROOT/tools/stage_times.py > $OUT/$st.log 2>&1
  python3 - <<PY
import csv,glob,collections
d=collections.defaultdict(list)
for f in glob.glob('$OUT/$st/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'k_stages' in row['Kernel_Name']:
            d[row['Counter_Name']].append(float(row['Counter_Value']))
print('== $st', [l for l in open('$OUT/$st.log').read().strip().splitlines() if 'us/launch' in l][-1:])
print('  ', {k: round(sum(v[-150:])/len(v[-150:])/4096,1) for k,v in sorted(d.items())})
PY
  rm -rf $OUT/$st
done
