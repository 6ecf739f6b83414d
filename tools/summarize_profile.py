#!/usr/bin/env python3
"""Condenses a tools/gpu_profile.sh output directory (one sub-directory per bench leg: closed / step / raycast) into
summary.txt (stdout) and pmc_latest.json: per kernel the rocprofv3 kernel-trace duration statistics and, PER ENV-STEP,
the HBM bytes of the FETCH_SIZE / WRITE_SIZE passes and the SQ instruction / cycle counters, together with the launch
shape they were measured on -- bench.py scales them to its own launch and refuses a shape that does not match.

gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads, so the
corrected sum is 2 x FETCH_SIZE + WRITE_SIZE; the guide calibrates that for wide reads only, the raw sum is kept
beside it (these kernels read 1- to 16-byte pieces): the two bracket the truth.  FETCH_SIZE / WRITE_SIZE are in KiB.
SQ_INSTS_* count wave-instructions; SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES / SQ_WAIT_ANY count quad-cycles (x4 = cycles)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
bench_args = sys.argv[2] if len(sys.argv) > 2 else ''
ENVS, WORKLOAD, CHUNK = 4096, 'config2', 300
toks = bench_args.split()
for i, t in enumerate(toks):
    if t == '--envs':
        ENVS = int(toks[i + 1])
    if t == '--workload':
        WORKLOAD = toks[i + 1]
    if t == '--chunk':
        CHUNK = int(toks[i + 1])
if '--envs' not in toks:
    ENVS = {'config2': 4096, 'config3': 65536, 'config4': 32768, 'config5': 32768, 'config5-step': 32768}[WORKLOAD]
# leg -> (kernel-name prefix of interest, key in pmc_latest.json, env-steps per launch)
LEGS = {'closed': ('k_closed<', 'k_closed', ENVS * CHUNK), 'step': ('k_stages<', 'k_stages', ENVS),
        'raycast': ('k_stages<', 'raycast_stage', ENVS)}
if WORKLOAD == 'config5-step':      # a non-persistent workload: its timed leg is k_stages launches
    LEGS['closed'] = ('k_stages<', 'k_stages_timed', ENVS)


def find(leg, pattern):
    return sorted(glob.glob(os.path.join(out, leg, pattern), recursive=True))


def own_rows(leg, rows):
    """Counter rows of the leg's kernel in dispatch order WITHOUT the leading eighth: the step / raycast legs launch their kernel
    in 4 batches (warm-up + 3 timed), and before them `bench.py --leg raycast` positions the worlds of a non-persistent workload
    with up to 100 FULL-step launches of the same kernel name -- the same rows the duration statistics drop."""
    rows = [v for _, v in sorted(rows, key=lambda kv: kv[0])]
    return rows[len(rows) // 8:] if leg in ('raycast', 'step') else rows


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from src_hash import source_hash
SRC = source_hash()      # the kernels these figures were measured on (tests/test_profiles.py checks it against the tree)

res = {}
for leg, (prefix, key, per_launch) in LEGS.items():
    if not os.path.isdir(os.path.join(out, leg)):
        continue
    r = {'shape': {'workload': WORKLOAD, 'envs': ENVS}, 'env_steps_per_launch': per_launch, 'src_hash': SRC,
         'source': f'tools/gpu_profile.sh, bench.py --leg {leg} {bench_args}'.strip()}
    if leg == 'closed' and key == 'k_closed':
        r['shape']['persistent'] = True
    for f in find(leg, 'ktrace/**/*kernel_trace.csv'):
        d = defaultdict(list)
        for row in csv.DictReader(open(f)):
            d[row['Kernel_Name']].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
        print(f'== [{leg}] kernel trace', os.path.relpath(f, out))
        for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
            v2 = sorted(v)
            print(f'{k[:80]:80s} n={len(v):6d} total={sum(v)/1e3:9.3f} ms avg={sum(v)/len(v):10.3f} us '
                  f'min={v2[0]:9.3f} med={v2[len(v2)//2]:9.3f} max={v2[-1]:10.3f}')
            if prefix in k:
                if leg == 'raycast' or leg == 'step':
                    v = v[len(v) // 8:]          # the leg's own warm-up launches (first of 4 batches) run on a cold clock
                r.update(avg_us_rocprof=sum(v) / len(v), med_us_rocprof=sorted(v)[len(v) // 2], max_us_rocprof=max(v), launches=len(v))
    for f in find(leg, 'ktrace/**/*kernel_stats.csv'):
        print(f'== [{leg}] rocprofv3 --stats', os.path.relpath(f, out))
        for line in open(f).read().splitlines()[:6]:
            print(line[:220])
        dst = os.path.join(out, f'kernel_stats_{leg}.csv')
        open(dst, 'w').write(open(f).read())
    for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
        for f in find(leg, f'pmc_{ctr}/**/*counter_collection.csv'):
            vals = own_rows(leg, [(int(row['Dispatch_Id']), float(row['Counter_Value'])) for row in csv.DictReader(open(f))
                                  if row['Counter_Name'] == ctr and prefix in row['Kernel_Name']])
            if vals:
                r[ctr + '_KiB_per_launch'] = sum(vals) / len(vals)
                print(f'== [{leg}] pmc {ctr}: n={len(vals)} avg={r[ctr + "_KiB_per_launch"]:.3f} KiB/launch')
    for f in find(leg, 'pmc_sq/**/*counter_collection.csv'):
        d = defaultdict(list)
        for row in csv.DictReader(open(f)):
            if prefix in row['Kernel_Name']:
                d[row['Counter_Name']].append((int(row['Dispatch_Id']), float(row['Counter_Value'])))
        for cn, v in d.items():
            v = own_rows(leg, v)
            r['sq_' + cn + '_per_launch'] = sum(v) / len(v)
            print(f'== [{leg}] pmc {cn:24s} avg={sum(v)/len(v):18.1f} per launch')
    if 'FETCH_SIZE_KiB_per_launch' in r and 'WRITE_SIZE_KiB_per_launch' in r:
        fz, wz = r['FETCH_SIZE_KiB_per_launch'] * 1024, r['WRITE_SIZE_KiB_per_launch'] * 1024
        r['hbm_bytes_per_env_step_raw'] = (fz + wz) / per_launch
        r['hbm_bytes_per_env_step'] = (2 * fz + wz) / per_launch
        r['fetch_bytes_per_env_step_raw'] = fz / per_launch
        r['write_bytes_per_env_step'] = wz / per_launch
    g = lambda n: r.get(f'sq_{n}_per_launch')
    if g('SQ_INSTS_VALU'):
        r['valu_insts_per_env_step'] = g('SQ_INSTS_VALU') / per_launch
        r['salu_insts_per_env_step'] = (g('SQ_INSTS_SALU') or 0) / per_launch
        r['valu_busy_cycles_per_env_step'] = 4 * g('SQ_ACTIVE_INST_VALU') / per_launch
        r['wave_cycles_per_env_step'] = 4 * g('SQ_WAVE_CYCLES') / per_launch
        r['wait_any_frac_of_wave_cycles'] = g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES')
        r['waves_per_launch'] = g('SQ_WAVES')
    res[key] = r
    print(f'== [{leg}] per env-step:', json.dumps({k: v for k, v in r.items() if 'per_env_step' in k or k.endswith('rocprof')}, indent=1))
res['note'] = ('hbm_bytes_per_env_step = (2 x FETCH_SIZE + WRITE_SIZE) / env-steps of a launch (gfx950: FETCH_SIZE counts 64 B per 128-B '
               'request on wide coalesced reads, MI355X_MICROARCH.md HBM section; uncalibrated for the 1- to 16-byte pieces these kernels '
               'read, so ..._raw = FETCH_SIZE + WRITE_SIZE is kept beside it and the two bracket the truth).  SQ_INSTS_* = wave-instructions; '
               'SQ_ACTIVE_INST_VALU and SQ_WAVE_CYCLES count quad-cycles, x4 = shader cycles summed over all waves.')
res['src_hash'] = SRC
json.dump(res, open(os.path.join(out, 'pmc_latest.json'), 'w'), indent=1)
print('== wrote', os.path.join(out, 'pmc_latest.json'))
