#!/usr/bin/env python3
"""Debug aid (GPU box): replays one seed of tests/test_gpu_plugins_random.py step by step, device vs oracle, and prints
the first step at which any field differs, with the planner statistics of both sides.
python tools/random_seed_debug.py 1892 [2563 ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np
import torch


def main():
    import drone2d_amd as pkg
    from drone2d_amd import _lib
    from oracle_lib import OracleBackend
    import test_gpu_plugins as G
    import test_gpu_plugins_random as R
    hip, oracle = _lib.HipBackend('cuda:0'), OracleBackend()
    for seed in [int(a) for a in sys.argv[1:]]:
        rng = np.random.RandomState(1000 + seed)
        kw = R._random_cfg(rng)
        B, T, chunk = int(rng.choice([3, 5, 8])), 160, int(rng.choice([5, 9, 16]))
        mode = dict(auto_reset=True) if seed % 3 else dict(freeze_done=True)
        print(f'== seed {seed} B={B} chunk={chunk} {mode} {kw}', flush=True)
        # the test's own launch pattern first (chunked), then step by step
        for pattern in ('chunk', 'single'):
            dev, ref = G._pair(pkg, hip, oracle, B, **kw)
            step = chunk if pattern == 'chunk' else 1
            prev = None
            for t in range(0, T, step):
                dev.closed_loop(step, **mode)
                ref.closed_loop(step, **mode)
                dev.sync()
                bad = []
                for name in G.FIELDS + ('action', 'plan_ok', 'wp_valid', 'wp'):
                    a, b = dev.state.t[name].cpu(), ref.state.t[name]
                    if not torch.equal(a, b):
                        bad.append((name, (a != b).nonzero()[:4].tolist()))
                for name in G.PLUGIN_FIELDS:
                    a, b = dev.plugins.t[name].cpu(), ref.plugins.t[name]
                    if not torch.equal(a, b):
                        bad.append((name, (a != b).nonzero()[:4].tolist()))
                ps_d, ps_r = dev.plugins.t['plan_stat'].cpu().numpy(), ref.plugins.t['plan_stat'].numpy()
                if bad:
                    print(f'  [{pattern}] first difference after step {t + step}: {bad}')
                    e = bad[0][1][0][0]
                    print(f'    env {e}: counters dev {dev.state.counters[e].cpu().tolist()} ref {ref.state.counters[e].tolist()}')
                    print(f'    flags dev {dev.state.flags[e].cpu().tolist()} ref {ref.state.flags[e].tolist()}')
                    print(f'    drone dev {dev.state.drone[e].cpu().tolist()}\n          ref {ref.state.drone[e].tolist()}')
                    print(f'    plan_stat dev {ps_d[e].tolist()} ref {ps_r[e].tolist()}  (before this step: dev {prev[0][e].tolist()} ref {prev[1][e].tolist()})' if prev else '')
                    print(f'    traj_hdr dev {dev.plugins.t["traj_hdr"][e].cpu().tolist()} ref {ref.plugins.t["traj_hdr"][e].tolist()}')
                    print(f'    plan_ok dev {dev.state.plan_ok[e].item()} ref {ref.state.plan_ok[e].item()}')
                    break
                prev = (ps_d.copy(), ps_r.copy())
            else:
                print(f'  [{pattern}] all {T} steps match')


if __name__ == '__main__':
    main()
