#!/bin/bash
# Per-phase instruction counters of the closed loop: bench.py --no-persistent launches gaze / perceive / plan / act as
# separate kernels, so rocprofv3 --pmc attributes wave-instructions and cycles to each phase.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_phases; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d $OUT/sq -- python3 $ROOT/bench.py --steps 200 --warmup 50 --prologue 300 --no-persistent --leg closed --no-cpu-baseline --workers 0 > $OUT/run.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 $ROOT/bench.py --steps 200 --warmup 50 --prologue 300 --no-persistent --leg closed --no-cpu-baseline --workers 0 > $OUT/run2.log 2>&1
python3 - <<PY
import csv,glob,collections
d=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/sq/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k=row['Kernel_Name']
        for s in ('k_gaze','k_plan','k_stages','k_closed'):
            if s+'(' in k or s+'<' in k:
                d[s+('' if s!='k_stages' else '')][row['Counter_Name']].append(float(row['Counter_Value']))
for k,cs in d.items():
    print('==',k,{c: round(sum(v)/len(v)/4096,1) for c,v in sorted(cs.items())}, 'launches', len(next(iter(cs.values()))))
t=collections.defaultdict(list)
for f in glob.glob('$OUT/kt/**/*kernel_trace.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        t[row['Kernel_Name'][:60]].append((int(row['End_Timestamp'])-int(row['Start_Timestamp']))/1e3)
for k,v in sorted(t.items(), key=lambda kv:-sum(kv[1]))[:6]:
    print(f'{k:60s} n={len(v)} avg={sum(v)/len(v):.2f} us total={sum(v)/1e3:.2f} ms')
PY
