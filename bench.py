#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched Drone2D environment on MI355X (BASELINE.json metric).

  python bench.py --gpus 1 --steps 600 --warmup 300
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Workload (N=1): BASELINE.json configs[1] — 4096 batched envs x 10 agents, agent_radius=15, 50x50 grid, 50 rays,
Oxford gaze, Primitive planner; env i = the reference world for map_id 1+i.  A "step" = one reference-style step
of every env: a = Oxford.plan(info); Drone2DEnv2.step(a) with Primitive.replan_check / plan in the middle
(experiment.py:68-70), ALL of it on the device (d2d_closed_loop: one persistent launch, every wave loops over the
steps of its own env), finished episodes restarting from their seeded world (main.py:26-57).  Nothing is replayed
or skipped inside the timed region.  State is resident in HBM before the timed region.

Besides the headline the line carries `step_kernel`: the fused Drone2DEnv2.step kernel alone (k_stages; gaze
actions and planner heads resident in HBM, SURVEY.md 8(d) C2's replay mode), timed with HIP events after the
timed region -- the HBM-roofline figure of the raycast / step kernel the north star asks for.

N>1: every rank runs its own shard of 4096 envs (weak scaling, no per-step collective; one RCCL all_gather of
episode statistics).  Prints ONE JSON line on rank 0.
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
N_SIMD = 256 * 4                    # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9                    # nominal shader clock (the chip runs at or below it under load)


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


def cpu_baseline(pkg, params, budget_s=16.0):
    """The CPU oracle (oracle/, a scalar C port of the reference: step + Oxford + Primitive) timed on this box's
    host cores on a bounded sample of the same workload: the same closed loop (gaze -> perceive -> plan -> act,
    auto reset) over 512 envs of the same family.  Timed on 1 thread and on min(cores, 32) OpenMP threads over
    envs; the faster is reported with the thread count actually used."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from oracle_lib import OracleBackend
    from drone2d_amd import vec_env
    ob = OracleBackend()
    avail = host_cores()
    many = max(1, min(avail, 32))
    B = 512
    worlds = vec_env.build_worlds(params, 64, workers=0)
    out = {}
    for threads in sorted({1, many}):
        env = vec_env.VecDrone2DEnv(params, B, backend=ob, planner='Primitive', device_plugins=True, gaze='Oxford',
                                    worlds=[worlds[i % 64] for i in range(B)])
        ob.lib.d2d_oracle_set_threads(threads)
        n = 0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < budget_s / 2:
            env.closed_loop(2, auto_reset=True)
            n += 2
        out[threads] = B * n / (time.perf_counter() - t0)
    ob.lib.d2d_oracle_set_threads(1)
    best = max(out, key=out.get)
    return {'value': out[best], 'unit': 'env-steps/s', 'cores': best, 'kind': 'port',
            'single_core_value': out[1], 'host_cores_available': avail,
            'sample': f'oracle/d2d_oracle.c closed loop (Oxford + Primitive + step, auto reset), {B} envs x 10 agents (64 '
                      f'distinct seeded worlds of the GPU workload\'s family), ~{budget_s / 2:.0f} s per thread count '
                      f'{sorted(out)}; reference Python itself: 16.3 env-steps/s for this loop on 1 core (BASELINE.md, '
                      'build container)'}


def step_kernel_leg(torch, env, params, rank, device, B, K=500, Wm=200):
    """The fused Drone2DEnv2.step kernel alone: K d2d_step launches over the batch (one launch = B envs), gaze
    actions and planner heads resident in HBM, timed with HIP events on the launch stream."""
    T = K + Wm
    g = torch.Generator().manual_seed(1234 + rank)
    actions = (torch.rand(T, B, generator=g, dtype=torch.float64) * 2 - 1).to(device)
    wp = synth_plan(torch, T, B, params.map_size[0], params.map_size[1], 99 + rank, device)
    env.state.plan_ok.fill_(1)
    env.state.wp_valid.fill_(1)
    be = env.backend
    st = env.state.struct()
    stream = torch.cuda.current_stream(device)
    sp = C.c_void_p(stream.cuda_stream)

    def roll(t0, n):
        rc = be.fn['rollout'](C.byref(env.cfg), C.byref(st), n, actions.data_ptr() + t0 * B * 8, wp.data_ptr() + t0 * B * 48,
                              None, None, sp)
        if rc:
            raise RuntimeError(be.fn['last_error']().decode())
    roll(0, Wm)
    torch.cuda.synchronize()
    reps = []
    for _ in range(3):       # best of three repetitions of the same K launches (clock transients after the long kernel)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        roll(Wm, K)
        e1.record(stream)
        torch.cuda.synchronize()
        reps.append(e0.elapsed_time(e1) * 1e3 / K)
    launch_us = min(reps)
    achieved = ALGO_BYTES_PER_ENV_STEP * B / (launch_us * 1e-6) / 1e9
    return {'kernel': 'k_stages (fused Drone2DEnv2.step: agents, raycast, dynamic grid, trackers, control, collision, obs)',
            'inputs': 'fixed-seed U(-1,1) gaze actions and synthetic waypoint heads resident in HBM (replay mode)',
            'launches': K, 'repetitions_us': reps, 'envs_per_launch': B, 'launch_us': launch_us, 'env_steps_per_s': B / (launch_us * 1e-6),
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': _pmc('step_kernel_hbm_bytes_per_launch'),
                         'algo_bytes_per_env_step': ALGO_BYTES_PER_ENV_STEP},
            'valu': _valu('k_stages', B / (launch_us * 1e-6))}


def _valu(kernel, env_steps_per_s_per_gpu):
    """VALU-issue view of a kernel ("report both the HBM fraction and VALU utilisation"): the vector-pipe cycles
    one env-step holds a SIMD (rocprofv3 --pmc SQ_ACTIVE_INST_VALU, committed in profiles/pmc_latest.json) against
    1024 SIMDs x 2.4 GHz, with the rate measured live in this run."""
    k = _pmc(kernel) or {}
    busy = k.get('valu_busy_cycles_per_env_step')
    if not busy:
        return None
    ceiling = N_SIMD * CLOCK_HZ / busy
    return {'insts_per_env_step': k.get('valu_insts_per_env_step'), 'salu_insts_per_env_step': k.get('salu_insts_per_env_step'),
            'busy_cycles_per_env_step': busy, 'ceiling_env_steps_per_s': ceiling, 'frac': env_steps_per_s_per_gpu / ceiling,
            'note': 'issue ceiling of this instruction stream = 1024 SIMDs x 2.4 GHz / vector-pipe cycles per env-step (PMC: per-wave '
                    'issue cycles, ~4 per instruction; fp64 needs all 4, two waves\' int / f32 instructions can overlap on the SIMD-32, '
                    'so the true ceiling lies between this figure and twice it)'}


def _pmc(key):
    pj = os.path.join(ROOT, 'profiles', 'pmc_latest.json')
    try:
        return json.load(open(pj)).get(key)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=600)
    ap.add_argument('--warmup', type=int, default=300)
    ap.add_argument('--envs', type=int, default=4096, help='envs per GPU')
    ap.add_argument('--agents', type=int, default=10)
    ap.add_argument('--static-map', default='maps/empty_map.npy', help='exploration only: other BASELINE configs')
    ap.add_argument('--agent-speed', type=int, default=20)
    ap.add_argument('--agent-radius', type=int, default=15)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-step-kernel', action='store_true', help='skip the step-kernel-only leg after the timed region')
    ap.add_argument('--workers', type=int, default=min(8, os.cpu_count() or 1),
                    help='host processes building the worlds (forked BEFORE the GPU is touched; 0 = in-process, '
                         'use 0 under rocprofv3)')
    ap.add_argument('--dist-backend', default='nccl', choices=['nccl', 'gloo'],
                    help='gloo + --single-device: dry run of the multi-rank path on a 1-GPU box')
    ap.add_argument('--single-device', action='store_true', help='every rank uses cuda:0 (dry run only)')
    ap.add_argument('--chunk', type=int, default=300, help='steps per persistent d2d_closed_loop launch')
    ap.add_argument('--no-persistent', action='store_true',
                    help='exploration: one launch per stage per step (gaze, perceive, plan, act) instead of the persistent kernel')
    args = ap.parse_args()

    import torch
    import drone2d_amd as pkg
    from drone2d_amd import vec_env

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

    env = vec_env.VecDrone2DEnv(params, B, device=device, planner='Primitive', env_offset=rank * B, worlds=worlds,
                                device_plugins=True, gaze='Oxford')
    if args.no_persistent:
        env._plan.launch_args = None
    stream = torch.cuda.current_stream(device)

    def run(n):
        for c0 in range(0, n, args.chunk):
            env.closed_loop(min(args.chunk, n - c0), auto_reset=True)

    run(Wm)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    run(K)
    e1.record(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    gpu_ms = e0.elapsed_time(e1)                  # HIP-event time of the K steps on the launch stream
    nlaunch = (K + args.chunk - 1) // args.chunk

    # episode statistics: the only exchange of the path (RCCL all_gather over xGMI), once per run
    stats = env.episode_stats()
    pstat = env.plugins.t['plan_stat'].long()
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
        launch_us = gpu_ms * 1e3 / nlaunch              # one persistent launch = `chunk` steps of every env
        steps_per_launch = K / nlaunch
        achieved = ALGO_BYTES_PER_ENV_STEP * B * steps_per_launch / (launch_us * 1e-6) / 1e9
        line = {
            'metric': 'env-steps/sec (batched) at 10 agents, map_id=1', 'value': value, 'unit': 'env-steps/s',
            'n_gpus': world, 'steps': K, 'warmup': Wm, 'ms_per_step': elapsed * 1e3 / K,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': f'configs[1]: {B} batched envs per GPU x {args.agents} agents, agent_radius=15, '
                                   '50x50 uint8 grid, 50 rays, map_id=1+env, Oxford gaze + Primitive planner',
                       'envs_per_gpu': B, 'agents': env.N,
                       'gaze': 'Oxford on the device (yaw_planner.py:41-127), every step',
                       'planner': 'Primitive on the device (traj_planner.py:78-233): replan_check every step, A* search '
                                  'whenever the trajectory is empty',
                       'kalman_trackers': 'on device', 'auto_reset': True,
                       'launch_mode': ('one launch per stage per step' if args.no_persistent else
                                       f'persistent: one launch per {args.chunk} steps, each wave loops over its own env'),
                       'searches_per_env': float(pstat[:, 0].double().mean()), 'search_overflows': int(pstat[:, 3].sum())},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': _pmc('hbm_bytes_per_launch'),
                         'kernel': 'k_closed (persistent closed loop: Oxford + Drone2DEnv2.step + Primitive)',
                         'launch_us': launch_us, 'envs_per_launch': B, 'steps_per_launch': steps_per_launch,
                         'algo_bytes_per_env_step': ALGO_BYTES_PER_ENV_STEP,
                         'note': 'algorithmic bytes of the Drone2DEnv2.step stages (SURVEY 8(d)) x envs x steps of one launch; '
                                 'the plugin phases are latency / issue bound, `step_kernel` is the step kernel alone'},
            'valu': _valu('k_closed', B * steps_per_launch / (launch_us * 1e-6)),
            'episode_stats': {'envs': int(stats.shape[0]), 'running_dynamic_collisions': int(stats[:, 3].sum()),
                              'mean_cells_discovered': float(stats[:, 6].double().mean())},
        }
        if not args.no_step_kernel:       # after the timed region, on its own state
            env_hot = vec_env.VecDrone2DEnv(params, B, device=device, planner='external', env_offset=rank * B, worlds=worlds)
            line['step_kernel'] = step_kernel_leg(torch, env_hot, params, rank, device, B)
        if not args.no_cpu_baseline and world == 1:
            line['cpu_baseline'] = cpu_baseline(pkg, params)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
