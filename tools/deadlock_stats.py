#!/usr/bin/env python3
"""Diagnostic (GPU): how many Kalman trackers are active when a Primitive search runs into its iteration cap -- the searches the
longest chains of a persistent launch consist of.  Round 2: 1 402 of 51 983 searches in 400 steps of the bench
workload, every one of them with at least one active tracker (824 / 345 / 165 / 60 / 8 with 1 / 2 / 3 / 4 / 5), so a "nothing
changed since the last failing search" memo would never hit: the trackers move every step."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import drone2d_amd as pkg
from drone2d_amd import vec_env, _abi as A
p = pkg.Params(planner='Primitive', gaze_method='Oxford', agent_number=10, agent_radius=15, agent_max_speed=20, drone_max_speed=40, map_id=1)
B = 4096
worlds = vec_env.build_worlds(p, B, workers=8)
env = vec_env.VecDrone2DEnv(p, B, planner='Primitive', worlds=worlds, device_plugins=True, gaze='Oxford')
prev = env.plugins.t['plan_stat'][:, 0].clone()
fails = 0; fails_noact = 0; fails_samemap = 0; fails_both = 0; total_search = 0
last_fail_map = {}
dm_prev = None
hist = np.zeros(12, int)
for t in range(400):
    dm_before = env.state.dmap.clone()
    env.closed_loop(1, auto_reset=True)
    st = env.plugins.t['plan_stat']
    searched = (st[:, 0] != prev)
    prev = st[:, 0].clone()
    failed = searched & (env.state.plan_ok == 0) & (st[:, 1] >= 98)
    nact = env.state.active.sum(1)
    total_search += int(searched.sum()); fails += int(failed.sum())
    fails_noact += int((failed & (nact == 0)).sum())
    for k in nact[failed].cpu().numpy(): hist[min(int(k), 11)] += 1
print('steps 400 searches', total_search, 'long failing', fails, 'with no active tracker', fails_noact)
print('active trackers at long failing searches:', hist)
