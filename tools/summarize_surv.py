#!/usr/bin/env python3
"""Condenses a tools/gpu_profile_surv.sh output directory (bench.py --workload survivability under rocprofv3) into summary.txt
(stdout) and pmc_latest.json (one record, merged into profiles/pmc_latest.json by tools/collect_profiles.sh): the k_stages launches grouped by launch size (envs = work-items / 64: one wave per env) -- the
table's half-batches on two streams and the per-agent-count whole batches on one -- with their kernel-trace durations and, PER
ENV-STEP over all of the run's k_stages launches, the HBM bytes of the FETCH_SIZE / WRITE_SIZE passes (same corrections as
tools/summarize_profile.py: corrected = 2 x FETCH_SIZE + WRITE_SIZE, raw = the plain sum; KiB units)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from src_hash import source_hash
PREFIX = 'k_stages<'


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


res = {'shape': {'workload': 'survivability'}, 'src_hash': source_hash(),
       'source': 'tools/gpu_profile_surv.sh, bench.py --workload survivability --no-cpu-baseline'}
for f in find('ktrace/**/*kernel_trace.csv'):
    d = defaultdict(list)
    for row in csv.DictReader(open(f)):
        if PREFIX in row['Kernel_Name']:
            d[(re.search(r'k_stages<\d+>', row['Kernel_Name']).group(0), int(row['Grid_Size_X']) // 64)].append(
                (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
    groups = []
    for (name, envs), v in sorted(d.items()):
        v2 = sorted(v)
        groups.append({'kernel': name, 'envs_per_launch': envs, 'launches': len(v), 'avg_us': sum(v) / len(v), 'med_us': v2[len(v2) // 2],
                       'min_us': v2[0], 'max_us': v2[-1]})
        print(f'{name:16s} envs/launch {envs:6d}  n={len(v):5d}  avg={sum(v)/len(v):8.2f} us  med={v2[len(v2)//2]:8.2f}  min={v2[0]:8.2f}  max={v2[-1]:8.2f}')
    res['launch_groups'] = groups
for f in find('ktrace/**/*kernel_stats.csv'):
    print('== rocprofv3 --stats', os.path.relpath(f, out))
    for line in open(f).read().splitlines()[:4]:
        print(line[:220])
    open(os.path.join(out, 'kernel_stats_survivability.csv'), 'w').write(open(f).read())
tot = {}
for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in find(f'pmc_{ctr}/**/*counter_collection.csv'):
        kib, steps, n = 0.0, 0, 0
        for row in csv.DictReader(open(f)):
            if row['Counter_Name'] == ctr and PREFIX in row['Kernel_Name']:
                kib += float(row['Counter_Value'])
                steps += int(row['Grid_Size']) // 64
                n += 1
        if n:
            tot[ctr] = (kib * 1024, steps, n)
            print(f'== pmc {ctr}: {n} k_stages launches, {steps} env-steps, {kib * 1024 / steps:.1f} B per env-step')
if len(tot) == 2:
    (fz, s1, _), (wz, s2, _) = tot['FETCH_SIZE'], tot['WRITE_SIZE']
    res.update(env_steps_counted=s1, fetch_bytes_per_env_step_raw=fz / s1, write_bytes_per_env_step=wz / s2,
               hbm_bytes_per_env_step_raw=fz / s1 + wz / s2, hbm_bytes_per_env_step=2 * fz / s1 + wz / s2)
# the layout tools/merge_pmc.py and bench.py read: the record under its kernel's key, found by its `shape`
json.dump({'k_stages': res, 'src_hash': res['src_hash']}, open(os.path.join(out, 'pmc_latest.json'), 'w'), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != 'launch_groups'}, indent=1))
