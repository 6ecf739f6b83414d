#!/bin/bash
# PMC counters of one stage subset launched alone (tools/stage_times.py ONLY=<name>)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_stage; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for st in "$@"; do
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC"; do
    tag=$(echo $grp | md5sum | cut -c1-6)
    ONLY=$st timeout 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/${st}_$tag -- python3 $ROOT/tools/stage_times.py > $OUT/${st}_$tag.log 2>&1
  done
  python3 - <<PY
import csv,glob,collections
d=collections.defaultdict(list)
for f in glob.glob('$OUT/${st}_*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'k_stages' in row['Kernel_Name']:
            d[row['Counter_Name']].append(float(row['Counter_Value']))
print('== $st (avg over the LAST 150 dispatches = the timed single-stage launches)')
for k,v in sorted(d.items()):
    v=v[-150:]
    print(f'  {k:28s} {sum(v)/len(v):14.1f}')
PY
done
