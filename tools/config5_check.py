#!/usr/bin/env python3
"""BASELINE config 5 geometry (map 6400 x 6400 px -> 640 x 640 cells, 640 rays, N = 100): parity of a few envs
against the oracle, then timing of a larger batch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import drone2d_amd as pkg
from drone2d_amd import vec_env, _lib
from oracle_lib import OracleBackend
p = pkg.Params(planner='NoMove', agent_number=100, agent_radius=10, agent_max_speed=40, map_id=0, map_size=[6400, 6400],
               init_pos=[3200, 3200], target_list=[[6000, 6000]])
t0 = time.time(); worlds = vec_env.build_worlds(p, 4); print('world build %.2f s each' % ((time.time() - t0) / 4))
hip = _lib.HipBackend()
ref = vec_env.VecDrone2DEnv(p, 4, backend=OracleBackend(), worlds=worlds)
dev = vec_env.VecDrone2DEnv(p, 4, backend=hip, worlds=worlds)
rng = np.random.RandomState(0)
for t in range(4):
    a = rng.uniform(-1, 1, 4)
    if t == 2:
        xy = torch.tensor([[3000., 3100.], [200., 300.], [6300., 6200.], [3333., 1000.]], dtype=torch.float64)
        dev.state.drone[:, :2] = xy.cuda(); ref.state.drone[:, :2] = xy
    dev.step(a); ref.step(a); dev.sync()
    for name in ('agents', 'gt', 'dmap', 'drone', 'flags', 'hit', 'obs_local', 'counters', 'kf', 'dyn_prev'):
        assert torch.equal(dev.state.t[name].cpu(), ref.state.t[name]), (t, name)
print('parity ok: 4 envs x 4 steps, R =', dev.cfg.R, 'grid', dev.cfg.W, 'x', dev.cfg.H)
B = int(os.environ.get('B', 2048))
big = vec_env.VecDrone2DEnv(p, B, backend=hip, worlds=[worlds[i % 4] for i in range(B)])
acts = (torch.rand(40, 4, dtype=torch.float64, device='cuda') * 2 - 1).repeat(1, B // 4 + 1)[:, :B].contiguous()   # env e acts like env e % 4
for t in range(5): big.step(acts[t])
torch.cuda.synchronize(); t0 = time.perf_counter()
for t in range(5, 35): big.step(acts[t])
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
print(f'B={B}: {dt * 1e6:.1f} us per step, {B / dt:.3e} env-steps/s; grids {2 * B * 640 * 640 / 2**30:.2f} GiB')
# size-independent property at the full size: the replicas of a world (same actions) stay identical, bit for bit
if B % 4 == 0 and B > 4:
    for name in ('gt', 'dmap', 'drone', 'agents', 'counters', 'kf', 'obs_local', 'hit'):
        x = big.state.t[name]
        x = x.view(B // 4, 4, *x.shape[1:])
        assert bool((x == x[:1]).all()), name
    ref4 = vec_env.VecDrone2DEnv(p, 4, backend=hip, worlds=worlds)
    for t in range(35): ref4.step(acts[t, :4])
    torch.cuda.synchronize()
    for name in ('gt', 'dmap', 'drone', 'agents', 'counters', 'kf', 'obs_local', 'hit'):
        assert torch.equal(big.state.t[name][:4], ref4.state.t[name]), name
    print(f'replica property ok at B={B}: every env equals the 4-env run of its world; device memory in use '
          f'{torch.cuda.memory_allocated() / 2**30:.1f} GiB')
