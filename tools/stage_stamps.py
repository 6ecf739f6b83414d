#!/usr/bin/env python3
"""In-kernel stage stamps (diagnostic build -DD2D_STAMPS, csrc/stamps.so): median shader-clock cycles spent
between stage boundaries of the fused step, per wave.  The stamps drain vmcnt/lgkmcnt, so read shares, not
absolute kernel time."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['D2D_LIB'] = os.path.join(ROOT, 'gym-drone2d-activeperception_amd', 'csrc', 'stamps.so')
import torch
import drone2d_amd as pkg
from drone2d_amd import vec_env, _abi as A
from bench import synth_plan

B = int(os.environ.get('B', 4096))
params = pkg.Params(planner='Primitive', agent_number=10, agent_radius=15, agent_max_speed=20, map_id=1)
worlds = vec_env.build_worlds(params, 512, workers=0)
env = vec_env.VecDrone2DEnv(params, B, planner='external', worlds=[worlds[i % 512] for i in range(B)])
T = 120
g = torch.Generator().manual_seed(1)
actions = (torch.rand(T, B, generator=g, dtype=torch.float64) * 2 - 1).cuda()
wp = synth_plan(torch, T, B, 500, 500, 9, 'cuda')
env.state.plan_ok.fill_(1); env.state.wp_valid.fill_(1)
st = env.state.struct(); cfg = env.cfg; fn = env.backend.fn['run_stages']
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
stamps = torch.zeros(B, 16, dtype=torch.int64, device='cuda')
lib = env.backend.lib
lib.d2d_debug_set_stamps.argtypes = [C.c_void_p]
assert lib.d2d_debug_set_stamps(stamps.data_ptr()) == 0
names = {1: 'load_regs', 2: 'fsm+control', 3: 'agents->LDS', 4: 'window+cull', 5: 'angle+tan+prefilter', 6: 'march',
         7: 'hit/newly', 8: 'dyngrid', 9: 'tracker', 10: 'collide', 11: 'fence', 12: 'obs', 13: 'store_regs'}
acc = {}
for t in range(T):
    st.action = actions.data_ptr() + t * B * 8
    st.wp = wp.data_ptr() + t * B * 48
    assert fn(C.byref(cfg), C.byref(st), A.ST_ALL, sp) == 0
    if t >= 60:
        torch.cuda.synchronize()
        s = stamps.cpu().numpy()
        for k in range(1, 14):
            d = (s[:, k] - s[:, k - 1])
            acc.setdefault(k, []).append(float(__import__('numpy').median(d)))
        acc.setdefault('total', []).append(float(__import__('numpy').median(s[:, 13] - s[:, 0])))
        acc.setdefault('span', []).append(float(s[:, 13].max() - s[:, 0].min()))
import numpy as np
tot = np.mean(acc['total'])
for k in range(1, 14):
    v = np.mean(acc[k])
    print(f'{names[k]:22s} {v:9.0f} cycles  {100 * v / tot:5.1f} %')
print(f'wave lifetime (median) {tot:9.0f} cycles ; kernel span {np.mean(acc["span"]):9.0f} cycles')
