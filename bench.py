#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched Drone2D environment on MI355X (BASELINE.json metric).

  python bench.py --gpus 1 --steps 600 --warmup 300
  python bench.py --gpus N --steps K --warmup W          (no launcher: bench.py starts its own N ranks, one per GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Workload (default, N=1): BASELINE.json configs[1] — 4096 batched envs x 10 agents, agent_radius=15, 50x50 grid, 50
rays, Oxford gaze, Primitive planner; env i = the reference world for map_id 1+i.  A "step" = one reference-style
step of every env: a = Oxford.plan(info); Drone2DEnv2.step(a) with Primitive.replan_check / plan in the middle
(experiment.py:68-70), ALL of it on the device (d2d_closed_loop: one persistent launch, every wave loops over the
steps of its own env), finished episodes restarting from their seeded world (main.py:26-57).  Nothing is replayed or
skipped inside the timed region.  State is resident in HBM before the timed region.

Steady state whatever --steps / --warmup say: an untimed PROLOGUE (--prologue, 300 steps) runs before the warm-up, so
that the timed window holds episode ends, restarts and the steady search rate even when it is 20 steps long; the
line reports `searches_per_env_per_step` and `episode_ends` OF THE TIMED WINDOW.

Other BASELINE configurations: --workload config3 | config4 | config5 (envs per GPU: 65536 / 32768 / 32768, override
with --envs); config5 (640 x 640 cells, 640 rays, 100 agents) runs the closed loop like the others since round 3, config5-step
its fused Drone2DEnv2.step alone (NoMove, fixed-seed gaze actions: the figure of rounds 1 and 2).  The BASELINE jobs of 262144
envs are `--workload config4 --gpus 8` and `--workload config5 --gpus 8` (32768 envs per GPU).

Besides the headline the line carries, measured after the timed region with HIP events on the launch stream:
  step_kernel    the fused Drone2DEnv2.step kernel alone (k_stages; actions and planner heads resident in HBM)
  raycast_stage  d2d_run_stages(AGENTS | RAYCAST): the raycast kernel the north star's HBM target is quoted on
(--leg picks one of them for a profiler run: every k_stages row of that trace is then that leg), and `large_batch`:
the closed loop, the step kernel and the raycast stage once more on 65 536 envs (--large; the workload's worlds tiled),
where launch ramp and the tail of slow envs are amortised -- the throughput regime of the same kernels.
PMC-derived fields (`traffic`, `valu`) come from profiles/pmc_latest.json, which stores them PER ENV-STEP together
with the launch shape they were measured on; they are scaled to this run's launch and emitted only when workload,
envs and launch mode match the profile (else null), with `traffic_source` naming the file.

N>1: every rank runs its own shard (weak scaling, no per-step collective; one RCCL all_gather of episode statistics,
timed separately as `gather_ms`).  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8 TB/s
N_SIMD = 256 * 4                    # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9                    # nominal shader clock (the chip runs at or below it under load)
PMC_FILE = os.path.join('profiles', 'pmc_latest.json')

WORKLOADS = {
    # name: (envs per GPU, Params overrides, closed loop?, description)
    'config2': (4096, dict(agent_number=10, agent_radius=15, agent_max_speed=20), True,
                'configs[1]: {B} batched envs per GPU x 10 agents, agent_radius=15, 50x50 uint8 grid, 50 rays, map_id=1+env, '
                'Oxford gaze + Primitive planner'),
    'config3': (65536, dict(agent_number=50, agent_radius=10, agent_max_speed=40, static_map='maps/random_map_0.npy'), True,
                'configs[2]: {B} envs per GPU x 50 agents at max_speed=40 + 122 random_map_0 agents, Oxford + Primitive'),
    'config4': (32768, dict(agent_number=10, agent_radius=15, agent_max_speed=20, static_map='maps/obstacle_map.npy'), True,
                'configs[3]: {B} envs per GPU (262144 over 8) x 10 agents + 14 obstacle_map agents, Oxford + Primitive'),
    'config5': (32768, dict(agent_number=100, agent_radius=15, agent_max_speed=40, map_size=[6400, 6400], init_pos=[3200, 3200],
                            target_list=[[6000, 6000]]), True,
                'configs[4]: {B} envs per GPU (262144 over 8) x 100 agents, 640x640 uint8 grid (16x16-cell tiles), 640 rays, '
                'Oxford + Primitive (round 3: the device plugins take maps of this size)'),
    'config5-step': (32768, dict(agent_number=100, agent_radius=15, agent_max_speed=40, map_size=[6400, 6400], init_pos=[3200, 3200],
                                 target_list=[[6000, 6000]]), False,
                     'configs[4] geometry, the fused Drone2DEnv2.step alone: {B} envs per GPU x 100 agents, 640x640 uint8 grid, 640 rays; '
                     'NoMove, fixed-seed gaze actions (what rounds 1 and 2 reported as config 5)'),
}


def algo_bytes(cfg, agent_unit, stages='step'):
    """SURVEY.md 8(d), from the ACTUAL configuration: 36 N (agents) + 3 c N (dynamic grid, c = (2 u + 1)^2 cells per agent)
    + N (hit mask) + R S + R (S - 1) (ground-truth reads + drone-map writes, S samples per ray) + 2 L^2 (observation crop)
    + 76 (drone state, action, plan in; state, flags out).  `stages` = 'raycast': the AGENTS | RAYCAST launch alone
    (agents, hit mask, ray cells, 44 B of drone state in)."""
    N, R, L = cfg.N, cfg.R, cfg.L
    S = int(math.ceil(cfg.depth / (cfg.scale - 1.0))) + 1
    rays = R * S + R * (S - 1)
    if stages == 'raycast':
        return 36 * N + N + rays + 44
    cells = float(((2 * agent_unit.double() + 1) ** 2).sum(1).mean()) if N else 0.0
    return 36 * N + 3 * cells + N + rays + 2 * L * L + 76


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


def cpu_baseline(pkg, params, closed, budget_s=16.0):
    """The CPU oracle (oracle/, a scalar C port of the reference: step + Oxford + Primitive) timed on this box's
    host cores on a bounded sample of the same workload: the same loop over 512 envs (64 for the 640 x 640 maps) of the
    same family.  Timed on 1 thread and on min(cores, 32) OpenMP threads over envs; the faster is reported with the
    thread count actually used."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from oracle_lib import OracleBackend
    from drone2d_amd import vec_env
    ob = OracleBackend()
    avail = host_cores()
    many = max(1, min(avail, 32))
    big_map = params.map_size[0] * params.map_size[1] > 1000 * 1000
    B, nw = (64, 16) if big_map else (512, 64)
    worlds = vec_env.build_worlds(params, nw, workers=0)
    out = {}
    for threads in sorted({1, many}):
        if closed:
            env = vec_env.VecDrone2DEnv(params, B, backend=ob, planner='Primitive', device_plugins=True, gaze='Oxford',
                                        worlds=[worlds[i % nw] for i in range(B)])
        else:
            env = vec_env.VecDrone2DEnv(params, B, backend=ob, planner='NoMove', worlds=[worlds[i % nw] for i in range(B)])
        ob.lib.d2d_oracle_set_threads(threads)
        n = 0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < budget_s / 2:
            if closed:
                env.closed_loop(2, auto_reset=True)
            else:
                env.step(0.25)
                env.step(-0.5)
            n += 2
        out[threads] = B * n / (time.perf_counter() - t0)
    ob.lib.d2d_oracle_set_threads(1)
    best = max(out, key=out.get)
    loop = 'closed loop (Oxford + Primitive + step, auto reset)' if closed else 'fused step (NoMove)'
    return {'value': out[best], 'unit': 'env-steps/s', 'cores': best, 'kind': 'port',
            'single_core_value': out[1], 'host_cores_available': avail,
            'sample': f'oracle/d2d_oracle.c {loop}, {B} envs x {env.N} agents ({nw} distinct seeded worlds of the GPU '
                      f'workload\'s family), ~{budget_s / 2:.0f} s per thread count {sorted(out)}; reference Python itself: '
                      '16.3 env-steps/s for the config-2 closed loop on 1 core (BASELINE.md, build container)'}


class Clock:
    """HIP events on the launch stream (CPU dry runs of the multi-rank path: wall clock)."""

    def __init__(self, torch, device):
        self.torch, self.device = torch, torch.device(device)
        self.gpu = self.device.type == 'cuda'
        self.stream = torch.cuda.current_stream(self.device) if self.gpu else None

    def sync(self):
        if self.gpu:
            self.torch.cuda.synchronize(self.device)

    def timed_us(self, fn):
        import gc
        on = gc.isenabled()
        gc.disable()       # (as timeit does: a full collection of the interpreter's heap between two launches is ~10 ms of an empty queue)
        try:
            return self._timed_us(fn)
        finally:
            if on:
                gc.enable()

    def _timed_us(self, fn):
        if self.gpu:
            e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
            e0.record(self.stream)
            fn()
            e1.record(self.stream)
            # the host learns of the end by polling the event, then synchronises (a no-op by then): a blocking synchronize alone
            # wakes up 0.1-0.2 ms late, a tenth of a 20-step window's 2 ms
            while not e1.query():
                pass
            self.sync()
            return e0.elapsed_time(e1) * 1e3
        t0 = time.perf_counter()
        fn()
        return (time.perf_counter() - t0) * 1e6

    def stream_ptr(self):
        return C.c_void_p(self.stream.cuda_stream) if self.gpu else None


def _pmc_entry(kernel, shape):
    """The per-env-step PMC record of `kernel` if it was measured on this launch shape, else None."""
    try:
        d = json.load(open(os.path.join(ROOT, PMC_FILE)))
    except Exception:
        return None
    # the record of the default workload, then those of other launch shapes (`more`: BASELINE configs 3 and 5, same script)
    for k in [d.get(kernel)] + [m.get(kernel) for m in d.get('more', [])]:
        if k and all(k.get('shape', {}).get(n, False if n == 'persistent' else None) == v for n, v in shape.items()):
            return k
    return None


def roofline(kernel, shape, bytes_per_env_step, envs, steps_per_launch, launch_us, extra=None):
    achieved = bytes_per_env_step * envs * steps_per_launch / (launch_us * 1e-6) / 1e9
    k = _pmc_entry(kernel, shape)
    traffic = k['hbm_bytes_per_env_step'] * envs * steps_per_launch if k and k.get('hbm_bytes_per_env_step') else None
    traffic_raw = k['hbm_bytes_per_env_step_raw'] * envs * steps_per_launch if k and k.get('hbm_bytes_per_env_step_raw') else None
    r = {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
         'traffic': traffic, 'traffic_raw': traffic_raw,    # corrected (2 x FETCH_SIZE + WRITE_SIZE, the guide's gfx950 rule) and as counted
         'traffic_source': (f'{PMC_FILE} [{kernel}]: {k["hbm_bytes_per_env_step"]:.0f} B per env-step (rocprofv3 --pmc FETCH_SIZE / '
                            f'WRITE_SIZE on this launch shape, {k.get("source", "")}) x {envs} envs x {steps_per_launch:g} steps')
                           if traffic else f'{PMC_FILE} holds no PMC pass for this launch shape {shape}',
         'launch_us': launch_us, 'envs_per_launch': envs, 'steps_per_launch': steps_per_launch,
         'algo_bytes_per_env_step': bytes_per_env_step}
    if extra:
        r.update(extra)
    return r


def valu(kernel, shape, env_steps_per_s_per_gpu):
    """VALU-issue view ("report both the HBM fraction and VALU utilisation"): the vector-pipe cycles one
    env-step holds a SIMD (rocprofv3 --pmc SQ_ACTIVE_INST_VALU, per env-step in profiles/pmc_latest.json, same launch shape)
    against 1024 SIMDs x 2.4 GHz, with the rate measured live in this run."""
    k = _pmc_entry(kernel, shape)
    busy = k.get('valu_busy_cycles_per_env_step') if k else None
    if not busy:
        return None
    ceiling = N_SIMD * CLOCK_HZ / busy
    return {'insts_per_env_step': k.get('valu_insts_per_env_step'), 'salu_insts_per_env_step': k.get('salu_insts_per_env_step'),
            'busy_cycles_per_env_step': busy, 'ceiling_env_steps_per_s': ceiling, 'frac': env_steps_per_s_per_gpu / ceiling,
            'source': f'{PMC_FILE} [{kernel}]',
            'note': 'issue ceiling of this instruction stream = 1024 SIMDs x 2.4 GHz / vector-pipe cycles per env-step (PMC: per-wave '
                    'issue cycles; fp64 needs 4 per instruction, two waves\' int / f32 instructions can overlap on the SIMD-32, so the '
                    'true ceiling lies between this figure and twice it)'}


def stage_leg(torch, clock, env, params, rank, B, stages, K=500, Wm=100):
    """K launches of one k_stages stage set over the batch, actions and planner heads resident in HBM."""
    A = env.backend.fn
    T = K + Wm
    device = env.device
    g = torch.Generator().manual_seed(1234 + rank)
    actions = (torch.rand(T, B, generator=g, dtype=torch.float64) * 2 - 1).to(device)
    sp = clock.stream_ptr()
    if stages == 'step':
        wp = synth_plan(torch, T, B, params.map_size[0], params.map_size[1], 99 + rank, device)
        env.state.plan_ok.fill_(1)
        env.state.wp_valid.fill_(1)
        st = env.state.struct()

        def roll(t0, n):
            rc = A['rollout'](C.byref(env.cfg), C.byref(st), n, actions.data_ptr() + t0 * B * 8,
                              wp.data_ptr() + t0 * B * 48 if env.cfg.planner_mode == 0 else None, None, None, sp)
            if rc:
                raise RuntimeError(A['last_error']().decode())
    else:
        st = env.state.struct()
        bits = 2 | 4                                                  # D2D_ST_AGENTS | D2D_ST_RAYCAST (include/d2d.h)

        def roll(t0, n):
            for _ in range(n):
                rc = A['run_stages'](C.byref(env.cfg), C.byref(st), bits, sp)
                if rc:
                    raise RuntimeError(A['last_error']().decode())
    roll(0, Wm)
    clock.sync()
    reps = [clock.timed_us(lambda: roll(Wm, K)) / K for _ in range(3)]   # best of three (clock transients after a long kernel)
    return min(reps), reps


def large_batch_legs(torch, clock, pkg, vec_env, params, worlds, rank, BL, chunk, workload):
    """The same three kernels on a batch far above the chip's wave slots (the workload's worlds tiled over it): the closed
    loop, the fused step and the raycast stage -- measured after the headline, never part of `value`."""
    nw = len(worlds)
    tiled = [worlds[i % nw] for i in range(BL)]
    shape = {'workload': workload, 'envs': BL}
    out = {'envs': BL, 'distinct_worlds': nw}
    env = vec_env.VecDrone2DEnv(params, BL, device=clock.device, planner='Primitive', worlds=tiled, device_plugins=True, gaze='Oxford')
    env.closed_loop(300, auto_reset=True)
    clock.sync()
    K = 600
    us = clock.timed_us(lambda: [env.closed_loop(min(chunk, K - c0), auto_reset=True) for c0 in range(0, K, chunk)])
    algo = algo_bytes(env.cfg, env.state.agent_unit)
    nl = (K + chunk - 1) // chunk
    out['closed_loop'] = {'env_steps_per_s': BL * K / (us * 1e-6), 'steps': K, 'prologue_steps': 300,
                          'roofline': roofline('k_closed', dict(shape, persistent=True), algo, BL, K / nl, us / nl)}
    us, reps = stage_leg(torch, clock, env, params, rank, BL, 'raycast', K=200, Wm=40)
    out['raycast_stage'] = {'env_steps_per_s': BL / (us * 1e-6), 'repetitions_us': reps,
                            'roofline': roofline('raycast_stage', shape, algo_bytes(env.cfg, env.state.agent_unit, 'raycast'), BL, 1, us),
                            'valu': valu('raycast_stage', shape, BL / (us * 1e-6))}
    del env
    env = vec_env.VecDrone2DEnv(params, BL, device=clock.device, planner='external', worlds=tiled)
    us, reps = stage_leg(torch, clock, env, params, rank, BL, 'step', K=200, Wm=40)
    out['step_kernel'] = {'env_steps_per_s': BL / (us * 1e-6), 'repetitions_us': reps, 'roofline': roofline('k_stages', shape, algo, BL, 1, us)}
    return out


def survivability_bench(args):
    """SURVEY 8(f) f4, the reference's own batched consumer of the pure hot path, end to end on one MI355X: the whole table of
    script/difficulty_calculator/glob_survivability_calculator.py:44-57 -- 20 maps x 27 settings x 8 x 8 start cells x 240 steps =
    8.3 M env-steps -- through sweeps.survivability_table (every (setting, start cell) one env, the 240 steps of a batch one
    d2d_rollout call = 240 k_stages launches per stream).  `value` = env-steps / device time with the worlds resident in HBM; the
    host's share (building 540 seeded worlds, unpacking the flags into the reference's array) is reported beside it.  The table
    of map 0 is checked against the CPU oracle's, which is also the cpu_baseline."""
    import numpy as np
    import torch
    import drone2d_amd as pkg
    from drone2d_amd import sweeps, _lib
    maps = list(range(args.maps))
    # the host's share first: the 27 x maps seeded worlds (the reference's __init__ per setting, pure Python), over --workers forked
    # processes BEFORE this process touches the GPU (0 = one after the other, in-process)
    b0 = time.perf_counter()
    worlds = sweeps.survivability_worlds(map_ids=maps, workers=args.workers)
    worlds_s = time.perf_counter() - b0
    hip = _lib.HipBackend('cuda:0')
    device = str(hip.device)                 # (a CPU backend injected by the tests' dry-run harness reports 'cpu')
    if device != 'cpu':
        sweeps.survivability_table(map_ids=maps, device=device, backend=hip, worlds=worlds)   # warm-up: the same table once (modules
        #                                                                      loaded, the side streams and the allocator's blocks exist)
    t = {}
    import gc
    gc.collect()
    gc.disable()           # (as timeit does: a full collection of the interpreter's heap is ~10 ms, a third of the table's device time)
    w0 = time.perf_counter()
    table = sweeps.survivability_table(map_ids=maps, device=device, backend=hip, timings=t, worlds=worlds)
    wall = worlds_s + time.perf_counter() - w0
    order_n = [n for _ in maps for n in (10, 20, 30) for _ in range(9)]          # agent count of every setting, in the table's order
    # the kernel's own average launch time, per agent count, on ONE stream with HIP events (the rollout above runs two half-batches
    # on two free-running streams, whose launches overlap: its time / launches is not a kernel duration)
    per_n = []
    for rec in t['batches']:
        t1 = {}
        sel = dict(map_ids=maps, agent_numbers=(rec['N'],), device=device, backend=hip, timings=t1, streams=1,
                   worlds=[w for w, n in zip(worlds, order_n) if n == rec['N']])
        sweeps.survivability_table(**sel)
        if device != 'cpu':
            first = t1['device_s']
            t1.clear()                           # twice, the faster one kept: the first single-stream chain after the two-stream table
            sweeps.survivability_table(**sel)    # has shown one-off stalls of several ms (stream teardown of the runs before)
            if first < t1['device_s']:
                t1['device_s'] = first
        env = t1['last_env']
        algo = algo_bytes(env.cfg, env.state.agent_unit)
        launch_us = t1['device_s'] * 1e6 / t1['launches']
        per_n.append({'agents': rec['N'], 'envs_per_launch': rec['envs'], 'launches': t1['launches'], 'launch_us_one_stream': launch_us,
                      'algo_bytes_per_env_step': algo, 'achieved_GBs': algo * rec['envs'] / (launch_us * 1e-6) / 1e9,
                      'device_s_two_streams': rec['device_s'], 'device_s_one_stream': t1['device_s']})
        del t1['last_env']
    gc.enable()
    algo_total = sum(r['algo_bytes_per_env_step'] * r['envs_per_launch'] * 240 for r in per_n)
    achieved = algo_total / t['device_s'] / 1e9
    dom = max(per_n, key=lambda r: r['device_s_two_streams'])
    k = _pmc_entry('k_stages', {'workload': 'survivability'})       # tools/gpu_profile_surv.sh: bytes per env-step over the run's launches
    traffic = k['hbm_bytes_per_env_step'] * t['env_steps'] if k and k.get('hbm_bytes_per_env_step') else None
    line = {'metric': 'env-steps/sec (survivability table: glob_survivability_calculator.py, NoMove / NoControl, drone pinned per start cell)',
            'value': t['env_steps'] / t['device_s'], 'unit': 'env-steps/s', 'n_gpus': 1, 'steps': 240, 'warmup': 0,
            'ms_per_step': t['device_s'] * 1e3 / (240 * len(t['batches'])), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': f'f4 survivability table: {len(maps)} maps x 27 settings (agents 10/20/30 x size 5/10/15 x speed 20/40/60) x '
                                   f'8 x 8 start cells x 240 steps = {t["env_steps"]} env-steps; one batch per agent count '
                                   f'({t["batches"][0]["envs"]} envs), two half-batches on two streams',
                       'name': 'survivability', 'table_shape': list(table.shape), 'collisions_recorded': int(table.sum())},
            'end_to_end': {'wall_s': wall, 'host_worlds_s': worlds_s, 'host_worlds_workers': args.workers, 'host_build_s': t['build_s'],
                           'device_s': t['device_s'], 'host_post_s': t['post_s'], 'env_steps_per_s_wall': t['env_steps'] / wall,
                           'note': 'wall = the seeded worlds built on the host (host_worlds_s, over --workers processes) + the table call: '
                                   'batches assembled and uploaded (host_build_s), the 240-step rollouts (device_s), flags unpacked (host_post_s)'},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': traffic, 'traffic_raw': k['hbm_bytes_per_env_step_raw'] * t['env_steps'] if traffic else None,
                         'traffic_source': (f'{PMC_FILE} [k_stages, survivability]: {k["hbm_bytes_per_env_step"]:.0f} B per env-step corrected '
                                            f'({k["hbm_bytes_per_env_step_raw"]:.0f} as counted) x {t["env_steps"]} env-steps of the table '
                                            f'(like `achieved`: the whole table, not one launch)') if traffic
                                           else f'{PMC_FILE} holds no PMC pass of this workload',
                         'kernel': 'k_stages (fused Drone2DEnv2.step; d2d_rollout = one launch per step per stream)',
                         'note': 'achieved = algorithmic bytes (SURVEY 8(d), per agent count) of all launches / device time of the table; '
                                 'per agent count: the kernel\'s average launch time on one stream (synchronised wall clock around the 240 queued launches)',
                         'dominant_batch': dom, 'per_agent_count': per_n},
            }
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        from oracle_lib import OracleBackend
        ob = OracleBackend()
        threads = max(1, min(host_cores(), 32))
        ob.lib.d2d_oracle_set_threads(threads)
        c0 = time.perf_counter()
        ref = sweeps.survivability_table(map_ids=[0], device='cpu', backend=ob)
        cs = time.perf_counter() - c0
        ob.lib.d2d_oracle_set_threads(1)
        same = bool(np.array_equal(ref, table[:27]))
        line['cpu_baseline'] = {'value': 27 * 64 * 240 / cs, 'unit': 'env-steps/s', 'cores': threads, 'kind': 'port',
                                'sample': f'oracle/d2d_oracle.c, the 27 settings of map 0 (414720 env-steps) incl. host world construction, '
                                          f'{cs:.1f} s on {threads} threads; the reference\'s Python: 12.3 ms per env.step (SURVEY 8a)',
                                'table_of_map_0_equals_oracle': same}
        if not same:
            print(json.dumps(line), flush=True)
            sys.exit('bench.py: the survivability table of map 0 differs from the oracle\'s')
    print(json.dumps(line), flush=True)
    return 0


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh child processes of this script, one rank per GPU
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, exactly what torch.distributed.run would set), relay
    rank 0's JSON line, and fail if any rank fails or the line does not report N ranks.  The parent never initialises the
    GPU and never replaces a running program: the ranks are ordinary children (subprocess)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    entry = os.environ.get('D2D_BENCH_ENTRY', os.path.abspath(__file__))   # tests substitute their CPU dry-run harness
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, entry] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # rank 0's line is read by a thread while ALL children are polled: a rank that dies early (out of memory, a bad GPU, an
    # import error) would otherwise leave the others in their first collective until its timeout, and this parent with them
    import threading
    buf = []
    reader = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    reader.start()

    def stop_all(*_):
        for q in procs:
            if q.poll() is None:
                q.terminate()
        for q in procs:
            try:
                q.wait(timeout=10)
            except subprocess.TimeoutExpired:
                q.kill()

    import signal
    old_term = signal.signal(signal.SIGTERM, lambda *a: (stop_all(), sys.exit(1)))
    try:
        while any(q.poll() is None for q in procs):
            if any(q.poll() not in (None, 0) for q in procs):
                stop_all()
                break
            time.sleep(0.2)
    except KeyboardInterrupt:
        stop_all()
        raise
    finally:
        signal.signal(signal.SIGTERM, old_term)
    reader.join(timeout=10)
    out0 = buf[0] if buf else ''
    codes = [p.wait() for p in procs]
    if any(codes):
        sys.stdout.write(out0)
        print(f'bench.py: rank exit codes {codes}', file=sys.stderr)
        return 1
    lines = [ln for ln in out0.splitlines() if ln.startswith('{')]
    if len(lines) != 1:
        sys.stdout.write(out0)
        print('bench.py: rank 0 printed no JSON line', file=sys.stderr)
        return 1
    j = json.loads(lines[0])
    sys.stdout.write(out0)
    sys.stdout.flush()
    if j.get('n_gpus') != n or j.get('n_ranks_seen', n) != n:   # (a --leg step / raycast line carries no collective)
        print(f'bench.py: asked for {n} ranks, the line reports n_gpus={j.get("n_gpus")} n_ranks_seen={j.get("n_ranks_seen")}', file=sys.stderr)
        return 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=600)
    ap.add_argument('--warmup', type=int, default=300)
    ap.add_argument('--prologue', type=int, default=300,
                    help='untimed steps before the warm-up: the timed window then sees steady-state episodes whatever --warmup is')
    ap.add_argument('--workload', default='config2', choices=sorted(WORKLOADS) + ['survivability'])
    ap.add_argument('--maps', type=int, default=20, help='--workload survivability: map ids 0..maps-1 (the reference\'s table: 20)')
    ap.add_argument('--envs', type=int, default=0, help='envs per GPU (default: the workload\'s)')
    ap.add_argument('--leg', default='all', choices=['all', 'closed', 'step', 'raycast'],
                    help='closed: only the timed region; step / raycast: only that k_stages leg (profiler runs)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--workers', type=int, default=min(8, os.cpu_count() or 1),
                    help='host processes building the worlds (forked BEFORE the GPU is touched; 0 = in-process, '
                         'use 0 under rocprofv3)')
    ap.add_argument('--dist-backend', default='nccl', choices=['nccl', 'gloo'],
                    help='gloo + --single-device: dry run of the multi-rank path on a 1-GPU box')
    ap.add_argument('--single-device', action='store_true', help='every rank uses cuda:0 (dry run only)')
    ap.add_argument('--chunk', type=int, default=300, help='steps per persistent d2d_closed_loop launch')
    ap.add_argument('--no-persistent', action='store_true',
                    help='exploration: one launch per stage per step (gaze, perceive, plan, act) instead of the persistent kernel')
    ap.add_argument('--large', type=int, default=65536,
                    help='envs of the large-batch legs reported beside the headline (the same kernels where launch ramp and the tail '
                         'of slow envs are amortised: the throughput regime); 0 = skip')
    ap.add_argument('--grid-layout', default=None, choices=['rowmajor', 'tiled'],
                    help='device layout of the two grids (default: the library\'s choice -- 16 x 16-cell tiles above 256 x 256 cells)')
    ap.add_argument('--distinct-worlds', type=int, default=0,
                    help='build only this many seeded worlds per rank and tile them over the batch (0 = one world per env)')
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error('--gpus must be >= 1')
    if args.workload == 'survivability':
        if args.gpus != 1:
            ap.error('--workload survivability is a one-GPU job')
        sys.exit(survivability_bench(args))
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process only starts the N ranks (nothing here has touched the GPU)
        sys.exit(launch_ranks(args.gpus))
    if int(os.environ.get('WORLD_SIZE', '1')) != args.gpus:
        sys.exit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ.get("WORLD_SIZE")}: the line would misreport n_gpus; '
                 'launch with --nproc-per-node equal to --gpus (or without a launcher: bench.py starts its own ranks)')

    import torch
    import drone2d_amd as pkg
    from drone2d_amd import vec_env

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    B0, pkw, closed, descr = WORKLOADS[args.workload]
    if args.workload.startswith('config5'):
        args.large = 0          # 32768 envs of 0.8 MB are the large batch already (65536 of them would not fit beside the legs)
    B, K, Wm = (args.envs or B0), args.steps, args.warmup
    params = pkg.Params(planner='Primitive' if closed else 'NoMove', gaze_method='Oxford' if closed else 'NoControl',
                        drone_max_speed=40, map_id=1, **pkw)
    # host world construction (the reference's __init__, seeded per global env id) before any GPU call
    nw = min(B, args.distinct_worlds) if args.distinct_worlds else B
    worlds = vec_env.build_worlds(params, nw, env_offset=rank * B, workers=args.workers)
    if nw < B:
        worlds = [worlds[i % nw] for i in range(B)]
    if args.single_device:
        local = 0
    use_cuda = torch.cuda.is_available()
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if use_cuda:
            torch.cuda.set_device(local)
        if args.dist_backend == 'nccl':     # 'nccl' IS RCCL on ROCm: collectives over xGMI
            dist.init_process_group('nccl', device_id=torch.device(f'cuda:{local}'))
        else:
            dist.init_process_group('gloo')
    device = f'cuda:{local}'
    if use_cuda:
        torch.cuda.set_device(local)

    if closed:
        env = vec_env.VecDrone2DEnv(params, B, device=device, planner='Primitive', env_offset=rank * B, worlds=worlds,
                                    device_plugins=True, gaze='Oxford', grid_layout=args.grid_layout)
        if args.no_persistent:
            env._plan.launch_args = None
    else:
        env = vec_env.VecDrone2DEnv(params, B, device=device, planner='NoMove', env_offset=rank * B, worlds=worlds,
                                    grid_layout=args.grid_layout)
    device = env.device                     # a CPU backend injected by a dry-run harness reports 'cpu'
    coll_dev = device if args.dist_backend == 'nccl' else 'cpu'
    clock = Clock(torch, device)
    algo = algo_bytes(env.cfg, env.state.agent_unit)
    shape = {'workload': args.workload, 'envs': B, 'persistent': bool(closed and not args.no_persistent)}
    nlaunch_of = lambda n: (n + args.chunk - 1) // args.chunk if closed else n

    if closed:
        def run(n):
            for c0 in range(0, n, args.chunk):
                env.closed_loop(min(args.chunk, n - c0), auto_reset=True)
    else:
        g = torch.Generator().manual_seed(4321 + rank)
        T = args.prologue + Wm + K
        acts = (torch.rand(min(T, 512), B, generator=g, dtype=torch.float64) * 2 - 1).to(device)
        cursor = [0]

        def run(n):                         # d2d_rollout: n fused steps queued by one call; finished envs restart per chunk
            done = 0
            while done < n:
                m = min(n - done, acts.shape[0] - cursor[0] % acts.shape[0], args.chunk)
                env.backend.rollout(env.cfg, env._st, m, acts[cursor[0] % acts.shape[0]:], None, None)
                env.reset(env.state.flags[:, pkg._abi.F_DONE])
                cursor[0] += m
                done += m

    # BASELINE.json's metric names config 2 (10 agents, map_id = 1 + env); the other workloads say what they are
    metric = ('env-steps/sec (batched) at 10 agents, map_id=1' if args.workload == 'config2' else
              f'env-steps/sec (batched) at {env.N} agents, {args.workload}, map_id=1+env')
    line = None
    if args.leg in ('all', 'closed'):
        run(args.prologue)
        run(Wm)
        clock.sync()
        pstat0 = env.plugins.t['plan_stat'][:, 0].clone() if closed else None
        stats0 = env.episode_stats()
        if world > 1:
            dist.barrier()
        clock.sync()
        t0 = time.perf_counter()
        gpu_us = clock.timed_us(lambda: run(K))   # HIP events around the K steps on the launch stream; syncs
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        nlaunch = nlaunch_of(K)

        # episode statistics: the only exchange of the path (RCCL all_gather over xGMI), once per run
        stats = env.episode_stats()
        gather_ms, ranks_seen = None, 1
        if world > 1:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
            stats_c = stats.to(coll_dev)
            allstats = [torch.empty_like(stats_c) for _ in range(world)]
            clock.sync()
            tg = time.perf_counter()
            dist.all_gather(allstats, stats_c)
            clock.sync()
            gather_ms = (time.perf_counter() - tg) * 1e3
            stats = torch.cat(allstats)
            ranks_seen = dist.get_world_size()
        if rank == 0:
            value = world * B * K / elapsed
            launch_us = gpu_us / nlaunch                # closed loop: one persistent launch = `chunk` steps of every env
            steps_per_launch = K / nlaunch
            window = {'episode_ends': int((env.episode_stats()[:, 0] < stats0[:, 0] + K).sum())}
            if closed:
                window['searches_per_env_per_step'] = float((env.plugins.t['plan_stat'][:, 0] - pstat0).double().mean()) / K
            cfgd = {'workload': descr.format(B=B), 'name': args.workload, 'envs_per_gpu': B, 'agents': env.N,
                    'grid': [env.cfg.W, env.cfg.H], 'grid_layout': 'tiled 16x16' if env.cfg.grid_tile else 'row-major [W][H]',
                    'rays': env.cfg.R, 'prologue_steps': args.prologue,
                    'distinct_worlds_per_gpu': nw, 'auto_reset': True, 'kalman_trackers': 'on device', 'timed_window': window}
            if closed:
                cfgd.update({'gaze': 'Oxford on the device (yaw_planner.py:41-127), every step',
                             'planner': 'Primitive on the device (traj_planner.py:78-233): replan_check every step, A* search '
                                        'whenever the trajectory is empty',
                             'launch_mode': ('one launch per stage per step' if args.no_persistent else
                                             f'persistent: one launch per {args.chunk} steps, each wave loops over its own env'),
                             'search_overflows': int(env.plugins.t['plan_stat'][:, 3].sum())})
                kern = 'k_closed'
                kname = 'k_closed (persistent closed loop: Oxford + Drone2DEnv2.step + Primitive)'
                note = ('algorithmic bytes of the Drone2DEnv2.step stages (SURVEY 8(d), from this run\'s N / R / L / cells per agent) '
                        'x envs x steps of one launch; the plugin phases are latency / issue bound, `step_kernel` is the step alone')
            else:
                cfgd.update({'gaze': 'fixed-seed U(-1,1) actions resident in HBM', 'planner': 'NoMove on the device',
                             'launch_mode': 'd2d_rollout: one k_stages launch per step'})
                kern, kname, note = 'k_stages', 'k_stages (fused Drone2DEnv2.step)', 'one launch = one step of every env'
            line = {
                'metric': metric, 'value': value, 'unit': 'env-steps/s',
                'n_gpus': world, 'steps': K, 'warmup': Wm, 'ms_per_step': elapsed * 1e3 / K,
                'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
                'config': cfgd,
                'roofline': roofline(kern, shape, algo, B, steps_per_launch, launch_us, {'kernel': kname, 'note': note}),
                'valu': valu(kern, shape, B * steps_per_launch / (launch_us * 1e-6)),
                'n_ranks_seen': ranks_seen, 'gather_ms': gather_ms,
                'episode_stats': {'envs': int(stats.shape[0]), 'running_dynamic_collisions': int(stats[:, 3].sum()),
                                  'mean_cells_discovered': float(stats[:, 6].double().mean())},
            }
    elif rank == 0:
        line = {'metric': metric, 'value': None, 'unit': 'env-steps/s',
                'n_gpus': world, 'leg': args.leg, 'config': {'workload': descr.format(B=B), 'name': args.workload}}

    if rank == 0:
        sshape = {'workload': args.workload, 'envs': B}
        if args.leg in ('all', 'step'):       # after the timed region, on its own state
            env_hot = vec_env.VecDrone2DEnv(params, B, device=device, planner='external' if closed else 'NoMove',
                                            env_offset=rank * B, worlds=worlds, grid_layout=args.grid_layout)
            us, reps = stage_leg(torch, clock, env_hot, params, rank, B, 'step')
            line['step_kernel'] = {
                'kernel': 'k_stages (fused Drone2DEnv2.step: agents, raycast, dynamic grid, trackers, control, collision, obs)',
                'inputs': 'fixed-seed U(-1,1) gaze actions and synthetic waypoint heads resident in HBM (replay mode)',
                'launches': 500, 'repetitions_us': reps, 'env_steps_per_s': B / (us * 1e-6),
                'roofline': roofline('k_stages', sshape, algo, B, 1, us), 'valu': valu('k_stages', sshape, B / (us * 1e-6))}
            del env_hot
        if args.leg in ('all', 'raycast'):
            # the raycast stage on mid-episode worlds: after the timed region on the closed-loop state, else after a short flight
            if args.leg == 'raycast':
                run(min(args.prologue, 100))
            us, reps = stage_leg(torch, clock, env, params, rank, B, 'raycast')
            rb = algo_bytes(env.cfg, env.state.agent_unit, 'raycast')
            line['raycast_stage'] = {
                'kernel': 'k_stages with D2D_ST_AGENTS | D2D_ST_RAYCAST (d2d_run_stages): Agent.step + Raycast.castRays, utils.py:472-493,593-713',
                'state': 'the worlds as the flight above left them (drones spread over their maps, explored maps part filled)',
                'launches': 500, 'repetitions_us': reps, 'env_steps_per_s': B / (us * 1e-6),
                'roofline': roofline('raycast_stage', sshape, rb, B, 1, us,
                                     {'note': 'bytes: 36 N agents + N hit mask + R S ground-truth reads + R (S - 1) drone-map writes + 44 B pose'}),
                'valu': valu('raycast_stage', sshape, B / (us * 1e-6))}
        if args.large and world == 1 and args.leg == 'all' and closed and use_cuda:
            line['large_batch'] = large_batch_legs(torch, clock, pkg, vec_env, params, worlds, rank, args.large, args.chunk, args.workload)
        if not args.no_cpu_baseline and world == 1 and args.leg == 'all':
            line['cpu_baseline'] = cpu_baseline(pkg, params, closed)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
