#!/usr/bin/env python3
"""Per-stage timing of the fused step on the bench workload: each stage subset is launched alone through
d2d_run_stages (HIP events over many launches).  Diagnostic only (subsets change how the state evolves)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import drone2d_amd as pkg
from drone2d_amd import vec_env, _abi as A
from bench import synth_plan, WORKLOADS

B = int(os.environ.get('B', 4096))
wl = os.environ.get('WORKLOAD')          # e.g. WORKLOAD=config5 B=8192: the stages at that configuration's geometry
if wl:
    params = pkg.Params(planner='Primitive', drone_max_speed=40, map_id=1, **WORKLOADS[wl][1])
else:
    extra = {}
    if os.environ.get('MAP'):            # e.g. MAP=600,600: a non-default geometry (the generic kernel)
        w, h = (int(v) for v in os.environ['MAP'].split(','))
        extra = dict(map_size=[w, h], init_pos=[w // 2, h // 2], target_list=[[w - 60, h - 60]])
    params = pkg.Params(planner='Primitive', agent_number=int(os.environ.get('N', 10)), agent_radius=int(os.environ.get('R', 15)),
                        agent_max_speed=20, map_id=1, **extra)
worlds = vec_env.build_worlds(params, min(B, int(os.environ.get('WORLDS', 512))), workers=int(os.environ.get('WORKERS', 0)))
env = vec_env.VecDrone2DEnv(params, B, planner='external', worlds=[worlds[i % len(worlds)] for i in range(B)],
                            grid_layout=os.environ.get('LAYOUT') or None)     # LAYOUT=rowmajor | tiled (default: the library's choice)
print('grid layout:', 'tiled 16x16' if env.cfg.grid_tile else 'row-major', flush=True)
T = 260
g = torch.Generator().manual_seed(1)
actions = (torch.rand(T, B, generator=g, dtype=torch.float64) * 2 - 1).cuda()
wp = synth_plan(torch, T, B, params.map_size[0], params.map_size[1], 9, 'cuda')
env.state.plan_ok.fill_(1); env.state.wp_valid.fill_(1)
st = env.state.struct(); cfg = env.cfg; fn = env.backend.fn['run_stages']
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(mask, t):
    st.action = actions.data_ptr() + t * B * 8
    st.wp = wp.data_ptr() + t * B * 48
    rc = fn(C.byref(cfg), C.byref(st), mask, sp)
    assert rc == 0, env.backend.fn['last_error']()


for t in range(60):          # bring the envs into a typical state
    run(A.ST_ALL, t)
torch.cuda.synchronize()
cases = [('ALL', A.ST_ALL), ('FSM+CONTROL', A.ST_FSM | A.ST_CONTROL), ('AGENTS', A.ST_AGENTS), ('RAYCAST', A.ST_RAYCAST),
         ('DYNGRID', A.ST_DYNGRID), ('TRACKER', A.ST_TRACKER), ('COLLIDE', A.ST_COLLIDE), ('OBS', A.ST_OBS),
         ('PERCEIVE', A.ST_PERCEIVE), ('ACT', A.ST_ACT), ('ALL-TRACKER', A.ST_ALL & ~A.ST_TRACKER),
         ('ALL-OBS', A.ST_ALL & ~A.ST_OBS), ('ALL-RAYCAST', A.ST_ALL & ~A.ST_RAYCAST), ('ALL-DYNGRID', A.ST_ALL & ~A.ST_DYNGRID), ('ALL', A.ST_ALL)]
only = os.environ.get('ONLY')
for name, mask in cases:
    if only and name != only:
        continue
    for t in range(60, 70):
        run(mask, t)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 180
    for t in range(70, 70 + n):
        run(mask, t)
    e1.record()
    torch.cuda.synchronize()
    print(f'{name:14s} {e0.elapsed_time(e1) * 1e3 / n:8.2f} us/launch')
