ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in libd2d_r01.so libd2d_hip.so; do
OUT=$ROOT/gpurun_out/pmc_ph_$lib; rm -rf $OUT; mkdir -p $OUT
D2D_LIB=$ROOT/gym-drone2d-activeperception_amd/csrc/$lib timeout -k 10 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/sq -- python3 $ROOT/bench.py --workload config4 --envs 4096 --distinct-worlds 512 --steps 100 --warmup 50 --prologue 300 --no-persistent --leg closed --no-cpu-baseline --workers 0 > $OUT/run.log 2>&1
D2D_LIB=$ROOT/gym-drone2d-activeperception_amd/csrc/$lib timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 $ROOT/bench.py --workload config4 --envs 4096 --distinct-worlds 512 --steps 100 --warmup 50 --prologue 300 --no-persistent --leg closed --no-cpu-baseline --workers 0 > $OUT/run2.log 2>&1
python3 - <<PY
import csv,glob,collections
d=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/sq/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k=row['Kernel_Name']
        for s in ('k_gaze','k_plan','k_stages'):
            if s+'(' in k or s+'<' in k:
                d[s][row['Counter_Name']].append(float(row['Counter_Value']))
print('#### $lib')
for k,cs in d.items():
    print('==',k,{c: round(sum(v)/len(v)/4096,1) for c,v in sorted(cs.items())})
t=collections.defaultdict(list)
for f in glob.glob('$OUT/kt/**/*kernel_trace.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        t[row['Kernel_Name'][:50]].append((int(row['End_Timestamp'])-int(row['Start_Timestamp']))/1e3)
for k,v in sorted(t.items(), key=lambda kv:-sum(kv[1]))[:3]:
    print(f'{k:50s} n={len(v)} avg={sum(v)/len(v):.2f} us')
PY
done
