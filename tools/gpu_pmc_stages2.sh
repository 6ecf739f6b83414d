ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_stage2; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for st in ALL FSM+CONTROL AGENTS RAYCAST DYNGRID TRACKER COLLIDE OBS; do
  ONLY=$st timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d $OUT/$st -- python3 $ROOT/tools/stage_times.py > $OUT/$st.log 2>&1
  python3 - <<PY
import csv,glob,collections
d=collections.defaultdict(list)
for f in glob.glob('$OUT/$st/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'k_stages' in row['Kernel_Name']:
            d[row['Counter_Name']].append(float(row['Counter_Value']))
print('== $st', open('$OUT/$st.log').read().strip().splitlines()[-1])
print('  ', {k: round(sum(v[-150:])/len(v[-150:])/4096,1) for k,v in sorted(d.items())})
PY
done
