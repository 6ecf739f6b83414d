#!/usr/bin/env python3
"""Debug aid (GPU box): replays seeds of tests/test_gpu_step_random.py and prints, at the first step where the device
and the oracle differ, the tracker-related state of the first differing env.
python tools/step_seed_debug.py 106 244"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np
import torch

np.set_printoptions(precision=17, linewidth=200)


def main():
    import drone2d_amd as pkg
    from drone2d_amd import _lib
    from oracle_lib import OracleBackend
    import test_gpu_vs_oracle as G
    import test_gpu_step_random as R
    hip, oracle = _lib.HipBackend('cuda:0'), OracleBackend()
    A = pkg._abi
    for seed in [int(a) for a in sys.argv[1:]]:
        rng = np.random.RandomState(7000 + seed)
        kw = R._cfg(rng)
        B, T = int(rng.choice([2, 5, 9])), 24
        external = bool(rng.rand() < 0.5)
        print(f'== seed {seed} B={B} external={external} {kw}', flush=True)
        dev, ref = G._pair(pkg, hip, oracle, B, planner='Primitive' if external else 'NoMove', **kw)
        W, H = dev.cfg.W_px, dev.cfg.H_px
        for t in range(T):
            a = rng.uniform(-1, 1, B)
            tele = False
            if rng.rand() < 0.2:
                tele = True
                lo = 1 if rng.rand() < 0.3 else 15
                xy = np.stack([rng.randint(lo, W - lo, B), rng.randint(lo, H - lo, B)], 1).astype(np.float64)
                if rng.rand() < 0.3:
                    xy += rng.choice([0.0, 0.25, 0.5], (B, 2))
                for env in (dev, ref):
                    env.state.drone[:, :2] = torch.from_numpy(xy).to(env.device)
            if external:
                ok = rng.rand(B) < 0.7
                valid = ok & (rng.rand(B) < 0.8)
                wp = np.concatenate([np.stack([rng.uniform(12, W - 12, B), rng.uniform(12, H - 12, B)], 1).round()
                                     + rng.choice([0.0, 0.5], (B, 2)), rng.uniform(-40, 40, (B, 2)), np.zeros((B, 2))], axis=1)
                for env in (dev, ref):
                    env.set_plan(ok, valid, wp)
            pre = {k: ref.state.t[k].clone() for k in ('drone', 'agents', 'active', 'kf', 'kf_len', 'counters')}
            dev.step(a)
            ref.step(a)
            dev.sync()
            bad = []
            for name in G.FIELDS:
                x, y = dev.state.t[name].cpu(), ref.state.t[name]
                if not torch.equal(x, y):
                    bad.append((name, (x != y).nonzero()[:6].tolist()))
            if bad:
                print(f'  first difference at step {t + 1} (teleport this step: {tele}): {bad}')
                e = bad[0][1][0][0]
                print(f'  env {e}: drone before {pre["drone"][e].numpy()} action {a[e]}')
                print(f'    counters dev {dev.state.counters[e].cpu().tolist()} ref {ref.state.counters[e].tolist()} before {pre["counters"][e].tolist()}')
                print(f'    hit dev {dev.state.hit[e].cpu().tolist()} ref {ref.state.hit[e].tolist()}')
                print(f'    active dev {dev.state.active[e].cpu().tolist()} ref {ref.state.active[e].tolist()} before {pre["active"][e].tolist()}')
                print(f'    kf_len dev {dev.state.kf_len[e].cpu().tolist()} ref {ref.state.kf_len[e].tolist()} before {pre["kf_len"][e].tolist()}')
                print(f'    newly dev {dev.state.newly[e].item()} ref {ref.state.newly[e].item()}')
                kd, kr = dev.state.kf[e].cpu().numpy(), ref.state.kf[e].numpy()
                for k in range(kd.shape[0]):
                    if not np.array_equal(kd[k], kr[k]):
                        print(f'    kf[{k}] dev {kd[k]}\n          ref {kr[k]}\n       before {pre["kf"][e][k].numpy()}')
                        print(f'       agent {k}: {ref.state.agents[e][:, k].numpy()} radius {ref.state.agent_unit[e][:, k].numpy() if hasattr(ref.state, "agent_unit") else None}')
                break
        else:
            print('  all steps match')


if __name__ == '__main__':
    main()
