#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched Drone2D step on MI355X (BASELINE.json metric).

  python bench.py --gpus 1 --steps 200 --warmup 20
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Workload (N=1): BASELINE.json configs[1] — 4096 batched envs x 10 agents, agent_radius=15, 50x50 grid,
50 rays, env i = the reference world for map_id 1+i.  A "step" = one fused Drone2DEnv2.step over the
whole batch (d2d_step launches: by default the batch is cut into 2 independent halves stepped on two
free-running HIP streams, envs being independent; --streams 1 = one launch per step).  Gaze actions are fixed-seed U(-1,1) and the planner result is a
synthetic resident waypoint stream (the device follows it exactly as it follows a Primitive trajectory
head; Oxford/Primitive themselves are host plugins in the reference and not part of this hot path —
SURVEY.md 8(d) C2).  Inputs are resident in HBM before the timed region.  N>1: every rank steps its own
shard of 4096 envs (weak scaling, no per-step collective; one RCCL all_gather of episode statistics).

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_ENV_STEP = 3844      # N=10, c=9, R=50, S=10, L=33
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8 TB/s


def synth_plan(torch, T, B, W_px, H_px, seed, device):
    """Resident synthetic planner heads [T, B, 6]: each drone follows its own smooth closed curve through the
    map interior at <= 40 px/s, positions rounded like Primitive's np.around heads (traj_planner.py:121)."""
    g = torch.Generator().manual_seed(seed)
    ph = torch.rand(B, 4, generator=g, dtype=torch.float64) * 6.283185307179586
    fr = 0.5 + torch.rand(B, 2, generator=g, dtype=torch.float64)
    t = torch.arange(T, dtype=torch.float64).view(T, 1) * 0.1
    cx, cy = W_px / 2.0, H_px / 2.0
    ax, ay = W_px / 2.0 - 45.0, H_px / 2.0 - 45.0
    w = 0.08
    x = cx + ax * torch.sin(w * fr[:, 0] * t + ph[:, 0])
    y = cy + ay * torch.sin(w * fr[:, 1] * t + ph[:, 1])
    vx = ax * w * fr[:, 0] * torch.cos(w * fr[:, 0] * t + ph[:, 0])
    vy = ay * w * fr[:, 1] * torch.cos(w * fr[:, 1] * t + ph[:, 1])
    wp = torch.stack([x.round(), y.round(), vx, vy, torch.zeros_like(x), torch.zeros_like(x)], dim=2)
    return wp.contiguous().to(device)


def host_cores():
    """Cores this process may really use: affinity mask, capped by the cgroup CPU quota if there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = max(1, min(n, int(float(q) / float(per))))
    except Exception:
        pass
    return n


def cpu_baseline(pkg, params, budget_s=14.0):
    """The CPU oracle (oracle/, a scalar C port of the reference step) timed on this box's host cores on a
    bounded sample of the same workload: 2048 envs of the same family, NoMove, same kind of action stream.
    Timed on 1 thread and on min(cores, 32) OpenMP threads over envs; the faster is reported with the thread
    count actually used."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import numpy as np
    from oracle_lib import OracleBackend
    from drone2d_amd import vec_env
    ob = OracleBackend()
    avail = host_cores()
    many = max(1, min(avail, 32))
    B = 2048
    worlds = vec_env.build_worlds(params, 64, workers=0)
    env = vec_env.VecDrone2DEnv(params, B, backend=ob, planner='NoMove', worlds=[worlds[i % 64] for i in range(B)])
    rng = np.random.RandomState(0)
    out = {}
    for threads in sorted({1, many}):
        ob.lib.d2d_oracle_set_threads(threads)
        env.reset()
        n = 0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < budget_s / 2:
            env.step(rng.uniform(-1, 1, B))
            n += 1
        out[threads] = B * n / (time.perf_counter() - t0)
    ob.lib.d2d_oracle_set_threads(1)
    best = max(out, key=out.get)
    return {'value': out[best], 'unit': 'env-steps/s', 'cores': best, 'kind': 'port',
            'single_core_value': out[1], 'host_cores_available': avail,
            'sample': f'oracle/d2d_oracle.c, {B} envs x 10 agents (64 distinct seeded worlds of the GPU workload\'s '
                      f'family, NoMove), ~{budget_s / 2:.0f} s per thread count {sorted(out)}; '
                      'reference Python itself: 268 env-steps/s on 1 core (BASELINE.md, build container)'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=500)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--envs', type=int, default=4096, help='envs per GPU')
    ap.add_argument('--agents', type=int, default=10)
    ap.add_argument('--static-map', default='maps/empty_map.npy', help='exploration only: other BASELINE configs')
    ap.add_argument('--agent-speed', type=int, default=20)
    ap.add_argument('--agent-radius', type=int, default=15)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--workers', type=int, default=min(8, os.cpu_count() or 1),
                    help='host processes building the worlds (forked BEFORE the GPU is touched; 0 = in-process, '
                         'use 0 under rocprofv3)')
    ap.add_argument('--dist-backend', default='nccl', choices=['nccl', 'gloo'],
                    help='gloo + --single-device: dry run of the multi-rank path on a 1-GPU box')
    ap.add_argument('--single-device', action='store_true', help='every rank uses cuda:0 (dry run only)')
    ap.add_argument('--streams', type=int, default=2,
                    help='>1: the batch is cut into that many sub-batches stepped on free-running HIP streams (envs are '
                         'independent); the roofline object then describes one sub-batch launch')
    ap.add_argument('--chunk', type=int, default=10, help='chain mode: steps queued per d2d_rollout call')
    ap.add_argument('--mode', default='chain', choices=['chain', 'launch', 'graph'],
                    help='chain: the K steps of a stream are queued by ONE d2d_rollout call (default); launch: one '
                         'd2d_step call per step from Python; graph: the K launches captured in one hipGraph')
    args = ap.parse_args()

    import torch
    import drone2d_amd as pkg
    from drone2d_amd import vec_env, _abi as A

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    B, K, Wm = args.envs, args.steps, args.warmup
    params = pkg.Params(planner='Primitive', gaze_method='Oxford', agent_number=args.agents, agent_radius=args.agent_radius,
                        agent_max_speed=args.agent_speed, drone_max_speed=40, map_id=1, static_map=args.static_map)
    # host world construction (the reference's __init__, seeded per global env id) before any GPU call
    worlds = vec_env.build_worlds(params, B, env_offset=rank * B, workers=args.workers)
    if args.single_device:
        local = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local)
        if args.dist_backend == 'nccl':     # 'nccl' IS RCCL on ROCm: collectives over xGMI
            dist.init_process_group('nccl', device_id=torch.device(f'cuda:{local}'))
        else:
            dist.init_process_group('gloo')
    device = f'cuda:{local}'
    torch.cuda.set_device(local)
    coll_dev = device if args.dist_backend == 'nccl' else 'cpu'

    env = vec_env.VecDrone2DEnv(params, B, device=device, planner='external', env_offset=rank * B, worlds=worlds)
    T = K + Wm
    g = torch.Generator().manual_seed(1234 + rank)
    actions = (torch.rand(T, B, generator=g, dtype=torch.float64) * 2 - 1).to(device)
    wp = synth_plan(torch, T, B, params.map_size[0], params.map_size[1], 99 + rank, device)
    env.state.plan_ok.fill_(1)
    env.state.wp_valid.fill_(1)
    be = env.backend
    fn_step = be.fn['step']
    a_ptr, w_ptr = actions.data_ptr(), wp.data_ptr()
    S = max(1, min(args.streams, B))
    import copy
    main_stream = torch.cuda.current_stream(device)
    base = env.state.struct()
    subs = []
    for i in range(S):
        lo, hi = (B * i) // S, (B * (i + 1)) // S
        cfg_i = copy.copy(env.cfg)
        cfg_i.B = hi - lo
        st_i = A.State()
        for name in A.STATE_FIELDS:
            t = env.state.t.get(name)
            ptr = getattr(base, name)
            setattr(st_i, name, None if (t is None or not ptr) else ptr + lo * t.stride(0) * t.element_size())
        stream_i = main_stream if S == 1 else torch.cuda.Stream(device)
        subs.append((lo, cfg_i, st_i, stream_i, C.c_void_p(stream_i.cuda_stream)))
    # chain mode: each sub-batch owns contiguous [T][b] actions and [T][b][6] planner heads
    chain_in = [(actions[:, lo:lo + c_.B].contiguous(), wp[:, lo:lo + c_.B].contiguous()) for lo, c_, _, _, _ in subs]
    fn_roll = be.fn['rollout']

    def chain(t0, n, chunk=10):
        # the chains are fed round-robin in chunks of `chunk` steps so that the streams stay abreast of each other
        for c0 in range(t0, t0 + n, chunk):
            m = min(chunk, t0 + n - c0)
            for (lo, cfg_i, st_i, _, sp_i), (a_i, w_i) in zip(subs, chain_in):
                b = cfg_i.B
                rc = fn_roll(C.byref(cfg_i), C.byref(st_i), m, a_i.data_ptr() + c0 * b * 8,
                             w_i.data_ptr() + c0 * b * 48, None, None, sp_i)
                if rc:
                    raise RuntimeError(be.fn['last_error']().decode())

    def launch(t):
        for lo, cfg_i, st_i, _, sp_i in subs:
            st_i.action = a_ptr + (t * B + lo) * 8
            st_i.wp = w_ptr + (t * B + lo) * 6 * 8
            rc = fn_step(C.byref(cfg_i), C.byref(st_i), sp_i)
            if rc:
                raise RuntimeError(be.fn['last_error']().decode())

    if args.mode == 'chain':
        chain(0, Wm)
    else:
        for t in range(Wm):
            launch(t)
    torch.cuda.synchronize()

    graph = None
    if args.mode == 'graph':
        assert S == 1, '--mode graph needs --streams 1'
        graph = torch.cuda.CUDAGraph()
        cs = torch.cuda.Stream(device)
        with torch.cuda.graph(graph, stream=cs):
            subs[0] = subs[0][:4] + (C.c_void_p(torch.cuda.current_stream(device).cuda_stream),)
            for t in range(Wm, T):
                launch(t)
        torch.cuda.synchronize()

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in subs]
    t0 = time.perf_counter()
    for (e0, _), sub in zip(ev, subs):
        e0.record(sub[3])
    if graph is not None:
        graph.replay()
    elif args.mode == 'chain':
        chain(Wm, K, args.chunk)
    else:
        for t in range(Wm, T):
            launch(t)
    for (_, e1), sub in zip(ev, subs):
        e1.record(sub[3])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    gpu_ms = sum(e0.elapsed_time(e1) for e0, e1 in ev) / len(ev)      # HIP-event time of the K launches of a stream

    # episode statistics: the only exchange of the path (RCCL all_gather over xGMI), once per run
    stats = env.episode_stats()
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        stats = stats.to(coll_dev)
        allstats = [torch.empty_like(stats) for _ in range(world)]
        dist.all_gather(allstats, stats)
        stats = torch.cat(allstats)

    if rank == 0:
        value = world * B * K / elapsed
        launch_us = gpu_ms * 1e3 / K                       # HIP-event time per launch on the launch stream
        achieved = ALGO_BYTES_PER_ENV_STEP * (B / S) / (launch_us * 1e-6) / 1e9   # bytes of ONE launch / its duration
        traffic = None
        pj = os.path.join(ROOT, 'profiles', 'pmc_latest.json')
        if os.path.isfile(pj):
            try:
                traffic = json.load(open(pj)).get('hbm_bytes_per_launch')
            except Exception:
                traffic = None
        line = {
            'metric': 'env-steps/sec (batched) at 10 agents, map_id=1', 'value': value, 'unit': 'env-steps/s',
            'n_gpus': world, 'steps': K, 'warmup': Wm, 'ms_per_step': elapsed * 1e3 / K,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': f'configs[1]: {B} batched envs per GPU x {args.agents} agents, agent_radius=15, '
                                   '50x50 uint8 grid, 50 rays, map_id=1+env',
                       'envs_per_gpu': B, 'agents': env.N, 'launch_mode': args.mode, 'streams': S,
                       'gaze': 'fixed-seed U(-1,1) actions resident in HBM (Oxford is a host plugin)',
                       'planner': 'synthetic resident waypoint heads, followed as Primitive heads are '
                                  '(Primitive is a host plugin)',
                       'kalman_trackers': 'on device', 'auto_reset': False},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'kernel': 'k_stages (fused step)',
                         'launch_us': launch_us, 'envs_per_launch': B // S, 'concurrent_launches': S,
                         'algo_bytes_per_env_step': ALGO_BYTES_PER_ENV_STEP,
                         'aggregate_GBs': ALGO_BYTES_PER_ENV_STEP * B * K / elapsed / 1e9},
            'episode_stats': {'envs': int(stats.shape[0]), 'dynamic_collisions': int(stats[:, 3].sum()),
                              'mean_cells_discovered': float(stats[:, 6].double().mean())},
        }
        if not args.no_cpu_baseline and world == 1:
            line['cpu_baseline'] = cpu_baseline(pkg, params)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
