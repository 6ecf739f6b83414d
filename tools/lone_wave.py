#!/usr/bin/env python3
"""Exploration: time per step of ONE env (a lone wave) in the persistent closed loop, per plugin combination -- the
latency floor the slowest env of a launch runs at.  python tools/lone_wave.py [envs]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import drone2d_amd as pkg  # noqa: E402
from drone2d_amd import vec_env  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for planner, gaze in (('Primitive', 'Oxford'), ('Primitive', 'Rotating'), ('Primitive', 'NoControl')):
    for map_id in (1, 7):
        p = pkg.Params(planner=planner, gaze_method=gaze, agent_number=10, agent_radius=15, agent_max_speed=20, drone_max_speed=40,
                       map_id=map_id)
        env = vec_env.VecDrone2DEnv(p, B, planner=planner, device_plugins=True, gaze=gaze)
        env.closed_loop(300, auto_reset=True)
        torch.cuda.synchronize()
        s0 = env.plugins.t['plan_stat'][:, 0].sum().item()
        t0 = time.perf_counter()
        for _ in range(4):
            env.closed_loop(300, auto_reset=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 1200
        s1 = env.plugins.t['plan_stat'][:, 0].sum().item()
        print(f'{planner:9s} + {gaze:9s} map_id {map_id} B={B}: {dt * 1e6:6.1f} us per step, {(s1 - s0) / B / 1200:.3f} searches per step')
