#!/usr/bin/env python3
"""Condenses a tools/gpu_profile.sh output directory: per-kernel time statistics from the kernel trace and
per-launch HBM traffic from the FETCH_SIZE / WRITE_SIZE passes (gfx950 correction: FETCH_SIZE counts 64 B per
128-B request on wide coalesced reads, MI355X_MICROARCH.md section HBM -- both raw and corrected values are
kept; units of FETCH_SIZE / WRITE_SIZE are KiB).  Kernels of interest: k_closed (the persistent closed loop),
k_stages (the fused step), k_gaze / k_plan (the plugin stages when launched separately)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
KERNELS = ('k_closed_args', 'k_closed', 'k_stages', 'k_gaze', 'k_plan_reset', 'k_plan', 'k_reset')


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


def short(name):
    for k in KERNELS:
        if k + '<' in name or k + '(' in name or name.endswith(k):
            return k
    return None


res = defaultdict(dict)
for f in find('ktrace/**/*kernel_trace.csv'):
    d = defaultdict(list)
    for row in csv.DictReader(open(f)):
        d[row['Kernel_Name']].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
    print('== kernel trace', os.path.relpath(f, out))
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        v2 = sorted(v)
        print(f'{k[:70]:70s} n={len(v):6d} total={sum(v)/1e3:9.3f} ms avg={sum(v)/len(v):9.3f} us '
              f'min={v2[0]:8.3f} med={v2[len(v2)//2]:8.3f} max={v2[-1]:9.3f}')
        s = short(k)
        if s:
            res[s].update(avg_us=sum(v) / len(v), med_us=v2[len(v2) // 2], max_us=v2[-1], calls=len(v))
for f in find('ktrace/**/*kernel_stats.csv'):
    print('== rocprofv3 --stats', os.path.relpath(f, out))
    for line in open(f).read().splitlines()[:8]:
        print(line[:200])
for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in find(f'pmc_{ctr}/**/*counter_collection.csv'):
        d = defaultdict(list)
        for row in csv.DictReader(open(f)):
            if row['Counter_Name'] == ctr:
                d[row['Kernel_Name']].append(float(row['Counter_Value']))
        print('== pmc', ctr, os.path.relpath(f, out))
        for k, v in d.items():
            s = short(k)
            if s:
                print(f'{k[:70]:70s} n={len(v):6d} avg={sum(v)/len(v):12.3f} KiB/launch')
                res[s][ctr + '_KiB_per_launch'] = sum(v) / len(v)
for f in find('pmc_sq*/**/*counter_collection.csv'):
    d = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f)):
        d[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
    print('== pmc SQ', os.path.relpath(f, out))
    for k, cs in d.items():
        s = short(k)
        if s in ('k_closed', 'k_stages', 'k_gaze', 'k_plan'):
            for cn, v in cs.items():
                print(f'  {s:12s} {cn:24s} avg={sum(v)/len(v):16.1f}')
                res[s]['sq_' + cn] = sum(v) / len(v)
for s, r in res.items():
    if 'FETCH_SIZE_KiB_per_launch' in r and 'WRITE_SIZE_KiB_per_launch' in r:
        r['hbm_bytes_per_launch_raw'] = (r['FETCH_SIZE_KiB_per_launch'] + r['WRITE_SIZE_KiB_per_launch']) * 1024
        r['hbm_bytes_per_launch'] = (2 * r['FETCH_SIZE_KiB_per_launch'] + r['WRITE_SIZE_KiB_per_launch']) * 1024
flat = dict(res)
if 'k_closed' in res and 'hbm_bytes_per_launch' in res['k_closed']:
    flat['hbm_bytes_per_launch'] = res['k_closed']['hbm_bytes_per_launch']
if 'k_stages' in res and 'hbm_bytes_per_launch' in res['k_stages']:
    flat['step_kernel_hbm_bytes_per_launch'] = res['k_stages']['hbm_bytes_per_launch']
print('== json')
print(json.dumps(flat))
json.dump(flat, open(os.path.join(out, 'pmc_summary.json'), 'w'), indent=1)
