#!/usr/bin/env python3
"""Exploration: env-steps/s of the closed loop (Oxford + Primitive on the device, auto reset) for a batch cut into
S sub-batches on their own HIP streams.  python tools/closed_loop_bench.py --envs 4096 --streams 1,2,4,8 --steps 300"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=4096)
    ap.add_argument('--streams', default='1,4')
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--warmup', type=int, default=60)
    ap.add_argument('--chunk', type=int, default=5)
    ap.add_argument('--worlds', type=int, default=256, help='distinct seeded worlds (tiled over the batch)')
    args = ap.parse_args()
    import torch
    import drone2d_amd as pkg
    from drone2d_amd import vec_env
    p = pkg.Params(planner='Primitive', gaze_method='Oxford', agent_number=10, agent_radius=15, agent_max_speed=20,
                   drone_max_speed=40, map_id=1)
    worlds = vec_env.build_worlds(p, args.worlds, workers=8)
    B = args.envs
    for S in [int(x) for x in args.streams.split(',')]:
        envs, streams = [], []
        for i in range(S):
            lo, hi = (B * i) // S, (B * (i + 1)) // S
            envs.append(vec_env.VecDrone2DEnv(p, hi - lo, planner='Primitive', device_plugins=True, gaze='Oxford',
                                              worlds=[worlds[(lo + k) % len(worlds)] for k in range(hi - lo)]))
            streams.append(torch.cuda.Stream())

        def run(n):
            for c0 in range(0, n, args.chunk):
                m = min(args.chunk, n - c0)
                for env, st in zip(envs, streams):
                    with torch.cuda.stream(st):
                        env.closed_loop(m, auto_reset=True)
        run(args.warmup)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(args.steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        stat = torch.cat([e.plugins.t['plan_stat'] for e in envs]).cpu()
        print(f'B={B} streams={S}: {B * args.steps / dt:.3e} env-steps/s, {dt / args.steps * 1e6:.1f} us/step, '
              f'searches/env {stat[:, 0].double().mean():.2f} over {args.steps + args.warmup} steps, overflow {int(stat[:, 3].sum())}', flush=True)
        del envs


if __name__ == '__main__':
    main()
