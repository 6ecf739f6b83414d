#!/usr/bin/env python3
"""Experiment: the 4096-env batch split over S streams (S independent sub-batches stepped concurrently)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import drone2d_amd as pkg
from drone2d_amd import vec_env
from bench import synth_plan
B = 4096; K = 400; Wm = 40
params = pkg.Params(planner='Primitive', agent_number=10, agent_radius=15, agent_max_speed=20, map_id=1)
worlds = vec_env.build_worlds(params, 512, workers=0)
import itertools
for S, JOIN in ((1, False), (2, False), (4, False), (2, True), (4, True)):
    b = B // S
    envs, acts, wps, sts, streams = [], [], [], [], []
    for i in range(S):
        e = vec_env.VecDrone2DEnv(params, b, planner='external', worlds=[worlds[(i * b + j) % 512] for j in range(b)])
        e.state.plan_ok.fill_(1); e.state.wp_valid.fill_(1)
        g = torch.Generator().manual_seed(i)
        acts.append((torch.rand(K + Wm, b, generator=g, dtype=torch.float64) * 2 - 1).cuda())
        wps.append(synth_plan(torch, K + Wm, b, 500, 500, 9 + i, 'cuda'))
        envs.append(e); sts.append(e.state.struct()); streams.append(torch.cuda.Stream())
    fn = envs[0].backend.fn['step']
    cur = torch.cuda.current_stream()
    evs = [torch.cuda.Event() for _ in range(S)]
    ev0 = torch.cuda.Event()
    def launch(t):
        if JOIN:
            ev0.record(cur)
        for i in range(S):
            if JOIN:
                streams[i].wait_event(ev0)
            sts[i].action = acts[i].data_ptr() + t * b * 8
            sts[i].wp = wps[i].data_ptr() + t * b * 48
            assert fn(C.byref(envs[i].cfg), C.byref(sts[i]), C.c_void_p(streams[i].cuda_stream)) == 0
            if JOIN:
                evs[i].record(streams[i])
                cur.wait_event(evs[i])
    for t in range(Wm): launch(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(Wm, Wm + K): launch(t)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'streams={S} join={JOIN}: {dt / K * 1e6:.2f} us per 4096-env step, {B * K / dt:.3e} env-steps/s')
