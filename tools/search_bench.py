#!/usr/bin/env python3
"""Microbenchmark of the device A* (k_plan): B copies of the dead-lock world (target inside the border wall, every
step runs a full 99-expansion failing search) and of the README world (typical successful searches).  Prints the
HIP-event time of each plan stage launch.  python tools/search_bench.py [--envs 4096]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=4096)
    args = ap.parse_args()
    import torch
    import drone2d_amd as pkg
    from drone2d_amd import _lib, device_plugins as DP
    import replay
    be = _lib.HipBackend('cuda:0')
    for name, steps in (('deadlock_primitive', 8), ('readme_oxford_primitive', 60)):
        R = replay.Replay(pkg, be, name, kf=True, copies=args.envs)
        ps = DP.PluginState(R.p, R.cfg, be.device, [R.world['tracker_radius']] * args.envs, planner='Primitive',
                            gaze='Oxford' if 'oxford' in name else 'external')
        plan = ps.struct()
        out = []
        for t in range(steps):
            s = R.st.struct()
            if 'oxford' in name:
                be.gaze_stage(R.cfg, s, plan)
            else:
                R.st.action.fill_(float(R.fx['t_action'][t]))
            be.perceive(R.cfg, s)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            be.plan_stage(R.cfg, s, plan)
            e1.record()
            be.act(R.cfg, s)
            torch.cuda.synchronize()
            st = ps.t['plan_stat'][0].cpu().numpy()
            out.append((e0.elapsed_time(e1) * 1e3, int(st[0]), int(st[1]), int(st[2])))
        searched = [o for i, o in enumerate(out) if i == 0 or o[1] != out[i - 1][1]]
        print(name, f'B={args.envs}:', ' '.join(f'{us:.0f}us/{ex}exp/{nn}n' for us, _, ex, nn in searched[:10]),
              '| no-search launches:', ' '.join(f'{o[0]:.0f}' for i, o in enumerate(out) if i > 0 and o[1] == out[i - 1][1])[:80])


if __name__ == '__main__':
    main()
