#!/usr/bin/env python3
"""Diagnostic (needs the -DD2D_CHAIN_PROF build: csrc/chainprof.so): per-env chain time of one persistent closed-loop launch
-- the launch lasts as long as the longest chain -- and the share of it spent in searches."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('D2D_LIB', os.path.join(ROOT, 'gym-drone2d-activeperception_amd', 'csrc', 'chainprof.so'))
import numpy as np
import torch
import drone2d_amd as pkg
from drone2d_amd import vec_env

B = int(os.environ.get('B', 4096))
STEPS = int(os.environ.get('STEPS', 300))
# WORKLOAD=config3|config4|config5: bench.py's table (WORLDS distinct seeded worlds tiled over the batch)
import bench
WL = os.environ.get('WORKLOAD', 'config2')
NW = min(B, int(os.environ.get('WORLDS', 512 if WL != 'config2' else B)))
p = pkg.Params(planner='Primitive', gaze_method='Oxford', drone_max_speed=40, map_id=1, **bench.WORKLOADS[WL][1])
worlds = vec_env.build_worlds(p, NW, workers=0)
env = vec_env.VecDrone2DEnv(p, B, planner='Primitive', device_plugins=True, gaze='Oxford', worlds=[worlds[i % NW] for i in range(B)])
env.closed_loop(300, auto_reset=True)
torch.cuda.synchronize()
import ctypes as C
phase = torch.zeros(B, 16, dtype=torch.int64, device='cuda')
env.backend.lib.d2d_debug_set_phase_buf.argtypes = [C.c_void_p]
assert env.backend.lib.d2d_debug_set_phase_buf(phase.data_ptr()) == 0
for rep in range(2):
    phase.zero_()
    s0 = env.plugins.t['plan_stat'][:, 0].clone()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    env.closed_loop(STEPS, auto_reset=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    st = env.plugins.t['plan_stat'].cpu().numpy().astype(np.int64)
    chain, srch = st[:, 1] * 16, st[:, 2] * 16
    ns = st[:, 0] - s0.cpu().numpy()
    clk = chain.max() / (ms * 1e3)      # clocks per microsecond, from the longest chain ~ the launch
    if B > 4096:                        # several rounds of envs on the chip's 4096 wave slots: the longest chain is NOT the launch
        clk = float(os.environ.get('CLOCK_MHZ', 2100))
    print(f'launch {ms:.2f} ms for {STEPS} steps x {B} envs = {B * STEPS / ms / 1e3:.3e} env-steps/s; clock ~{clk:.0f} MHz')
    pct = lambda a, q: np.percentile(a, q)
    print('chain / longest chain : mean %.3f  median %.3f  p90 %.3f  p99 %.3f' % (chain.mean() / chain.max(), pct(chain, 50) / chain.max(), pct(chain, 90) / chain.max(), pct(chain, 99) / chain.max()))
    print('search share of chain : mean %.3f ; of the 16 longest chains %.3f' % ((srch / chain).mean(), (srch / chain)[np.argsort(chain)[-16:]].mean()))
    top = np.argsort(chain)[-8:][::-1]
    for e in top:
        print(f'  env {e:5d}: chain {chain[e] / clk / 1e3:7.2f} ms, searches {ns[e]:4d}, search time {srch[e] / clk / 1e3:7.2f} ms, per step w/o search {(chain[e] - srch[e]) / clk / STEPS:6.1f} us')
    ph = phase.cpu().numpy().astype(np.float64) / clk / STEPS          # us per step and phase
    names = ['gaze', 'perceive', 'planner every-step part + act (one call unless a search follows)', 'search', 'act behind a search']
    print('per-step time by phase (us), mean over envs / the 16 longest chains: ' + ', '.join(
        f'{n} {ph[:, i].mean():.1f} / {ph[np.argsort(chain)[-16:], i].mean():.1f}' for i, n in enumerate(names)))
    gz = ['tables + directions', 'seen pass', 'swept map + fence', 'live cells', 'rewards + bits', 'hot blocks', 'block sums', 'tree + argmax']
    print('gaze sections (us per step, stamped: shares): ' + ', '.join(f'{n} {ph[:, 5 + i].mean():.2f}' for i, n in enumerate(gz)))
    nos = (chain - srch) / clk / STEPS
    print('per-step time without searches (us): mean %.1f  p10 %.1f  p90 %.1f' % (nos.mean(), pct(nos, 10), pct(nos, 90)))
    print('per-search time (us): mean %.1f' % (srch.sum() / max(ns.sum(), 1) / clk))
