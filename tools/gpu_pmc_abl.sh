ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_abl; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in ab_nomarch ab_notan; do
  for st in RAYCAST; do
  export D2D_LIB=$ROOT/gym-drone2d-activeperception_amd/csrc/$lib.so
  ONLY=$st timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d $OUT/${lib}_$st -- python3 $ROOT/tools/stage_times.py > $OUT/${lib}_$st.log 2>&1
  python3 - <<PY
import csv,glob,collections
d=collections.defaultdict(list)
for f in glob.glob('$OUT/${lib}_$st/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'k_stages' in row['Kernel_Name']:
            d[row['Counter_Name']].append(float(row['Counter_Value']))
print('== $lib $st', open('$OUT/${lib}_$st.log').read().strip().splitlines()[-1])
print('  ', {k: round(sum(v[-150:])/len(v[-150:])/4096,1) for k,v in sorted(d.items())})
PY
  done
done
