"""Device planner / gaze plugins on the GPU (run with -m gpu): the HIP kernels against the episodes captured from
the imported reference, against libm (sin / cos), and against the CPU oracle on seeded batches -- every field of
the env state AND of the plugin state bit for bit."""
import ctypes as C

import numpy as np
import pytest
import torch

import replay
from test_gpu_vs_oracle import FIELDS, _worlds
from test_plugins_cpu import _closed_loop
from test_sincos import sincos_arguments

pytestmark = pytest.mark.gpu

PLUGIN_FIELDS = ('traj_hdr', 'trk_radius', 'trk_prev', 'seen_step')


@pytest.mark.parametrize('name', replay.TRACES_CLOSED + replay.TRACES_PLANNED)
def test_hip_plugins_reproduce_the_reference_episode(pkg, hip, name):
    _closed_loop(pkg, hip, name)


def test_device_sincos_is_bit_identical_to_libm(hip):
    rng = np.random.RandomState(6)
    x = sincos_arguments(rng)
    xd = torch.from_numpy(x).to(hip.device)
    sd, cd = torch.empty_like(xd), torch.empty_like(xd)
    hip.sincos_array(xd, sd, cd)
    hip.sync()
    libm = C.CDLL('libm.so.6')
    idx = np.concatenate([rng.randint(0, x.size, 300000), np.arange(x.size - 12000, x.size)])
    for name, got in (('sin', sd.cpu().numpy()), ('cos', cd.cpu().numpy())):
        fn = getattr(libm, name)
        fn.restype = C.c_double
        fn.argtypes = [C.c_double]
        ref = np.array([fn(float(v)) for v in x[idx]])
        assert np.array_equal(got[idx].view(np.int64), ref.view(np.int64)), name


def _pair(pkg, hip, oracle, B, **pk):
    from drone2d_amd import vec_env
    p = pkg.Params(planner='Primitive', gaze_method='Oxford', **pk)
    ref = vec_env.VecDrone2DEnv(p, B, backend=oracle, planner='Primitive', device_plugins=True, gaze='Oxford')
    dev = vec_env.VecDrone2DEnv(p, B, backend=hip, planner='Primitive', device_plugins=True, gaze='Oxford', worlds=_worlds(ref))
    return dev, ref


def _assert_same(dev, ref, tag):
    dev.sync()
    for name in FIELDS + ('action', 'plan_ok', 'wp_valid', 'wp'):
        a, b = dev.state.logical(name).cpu(), ref.state.t[name]     # grids in [W][H] whatever the device layout
        if not torch.equal(a, b):
            bad = (a != b).nonzero()
            raise AssertionError(f'{tag}: field {name} differs at {bad[:5].tolist()} ({len(bad)} elements)')
    for name in PLUGIN_FIELDS:
        a, b = dev.plugins.t[name].cpu(), ref.plugins.t[name]
        if not torch.equal(a, b):
            bad = (a != b).nonzero()
            raise AssertionError(f'{tag}: plugin field {name} differs at {bad[:5].tolist()} ({len(bad)} elements)')
    # the stored part of every trajectory
    hd = ref.plugins.t['traj_hdr']
    ta, tb = dev.plugins.t['traj'].cpu(), ref.plugins.t['traj']
    for e in range(ref.num_envs):
        h, n = int(hd[e, 0]), int(hd[e, 1])
        assert torch.equal(ta[e, h:n], tb[e, h:n]), f'{tag}: trajectory of env {e}'
    assert int(dev.plugins.t['plan_stat'][:, 3].sum()) == 0


CASES = [
    dict(B=48, T=160, chunk=8, agent_number=10, agent_radius=15, agent_max_speed=20, drone_max_speed=40, map_id=1),
    dict(B=24, T=90, chunk=5, agent_number=30, agent_radius=10, agent_max_speed=40, map_id=40),
    dict(B=16, T=120, chunk=6, agent_number=12, agent_radius=10, agent_max_speed=25, map_id=70, pillar_number=7),
    dict(B=6, T=60, chunk=4, agent_number=10, agent_radius=10, agent_max_speed=20, map_id=4, drone_max_speed=60),
    dict(B=8, T=70, chunk=7, agent_number=8, agent_radius=12, agent_max_speed=20, map_id=90, drone_max_speed=30,
         drone_view_range=120, drone_view_depth=100, target_list=[[250, 250], [450, 60]]),
    dict(B=6, T=40, chunk=5, agent_number=10, agent_radius=10, agent_max_speed=40, map_id=3,
         static_map='maps/obstacle_map.npy'),
    dict(B=5, T=60, chunk=10, agent_number=0, agent_radius=10, agent_max_speed=20, map_id=12, pillar_number=6),
    # 172 agents (50 + the 122 of random_map_0): the any-N specialisation, two envs per workgroup
    dict(B=4, T=30, chunk=6, agent_number=50, agent_radius=10, agent_max_speed=40, map_id=0, static_map='maps/random_map_0.npy',
         init_pos=[250, 30]),
    # a 1000 x 800 px map (100 x 80 cells), deeper and wider view: the generic instantiation of every phase
    dict(B=4, T=60, chunk=12, agent_number=25, agent_radius=12, agent_max_speed=40, map_id=6, map_size=[1000, 800],
         drone_view_depth=120, drone_view_range=120, init_pos=[300, 400], target_list=[[900, 700]]),
    # 52 x 40 = 2080 cells: numpy's pairwise recursion is NOT a perfect tree there (one half of a split is a block, the other splits
    # again) -- the level-by-level additions of the gaze stage, not its butterfly
    dict(B=6, T=80, chunk=8, agent_number=8, agent_radius=10, agent_max_speed=20, map_id=21, map_size=[520, 400],
         init_pos=[60, 60], target_list=[[460, 340], [60, 340]]),
    # 100 x 100 cells with a view 15 cells deep: a view box of 35 cells a side is more than the sparse pairwise path holds (two blocks
    # per box row, 64 in all) -- such a configuration takes the DENSE plan of the whole map (128 blocks) although the map has more
    # than 4096 cells (round 3 refused it with -4)
    dict(B=4, T=40, chunk=8, agent_number=12, agent_radius=12, agent_max_speed=30, map_id=33, map_size=[1000, 1000],
         drone_view_depth=150, drone_view_range=100, init_pos=[300, 300], target_list=[[800, 800]]),
    # 2000 x 1600 px with pillars and fast agents: trajectories of many hundred waypoints, so the gaze stage and replan_check walk them
    # through the chunk boxes (d2d_plan.traj_box) -- walls that come into view on the way (replans), trackers crossing far chunks
    dict(B=6, T=240, chunk=16, agent_number=30, agent_radius=12, agent_max_speed=60, map_id=51, map_size=[2000, 1600],
         pillar_number=9, init_pos=[120, 120], target_list=[[1880, 1480], [120, 1480]]),
    # 40 x 30 = 1200 cells: a perfect tree of 16 blocks (half the butterfly's lanes hold +0.0)
    dict(B=6, T=80, chunk=8, agent_number=6, agent_radius=10, agent_max_speed=20, map_id=22, map_size=[400, 300],
         init_pos=[50, 50], target_list=[[340, 240], [60, 240]]),
]


@pytest.mark.parametrize('case', CASES, ids=lambda c: f"N{c['agent_number']}_B{c['B']}_v{c.get('drone_max_speed', 40)}" + ('_%dx%d' % tuple(c['map_size']) if 'map_size' in c else ''))
def test_closed_loop_matches_oracle(pkg, hip, oracle, case):
    """gaze -> perceive -> plan -> act with auto reset, `chunk` steps per call, compared after every call."""
    case = dict(case)
    B, T, chunk = case.pop('B'), case.pop('T'), case.pop('chunk')
    dev, ref = _pair(pkg, hip, oracle, B, **case)
    oracle_threads = getattr(oracle.lib, 'd2d_oracle_set_threads')
    oracle_threads(8)
    try:
        done_seen = 0
        for t in range(0, T, chunk):
            dev.closed_loop(chunk, auto_reset=True)
            ref.closed_loop(chunk, auto_reset=True)
            _assert_same(dev, ref, f'after step {t + chunk}')
            done_seen += int((ref.state.counters[:, pkg._abi.C_STEPS] < t + chunk).sum())
        assert done_seen > 0 or T < 100      # the longer cases do see episodes end and restart
    finally:
        oracle_threads(1)


@pytest.mark.parametrize('kw', [dict(agent_number=14, agent_radius=12, agent_max_speed=40, map_id=30),
                                dict(agent_number=14, agent_radius=12, agent_max_speed=40, map_id=30, drone_view_range=100)],
                         ids=['persistent', 'per_stage'])
def test_freeze_mode_matches_oracle(pkg, hip, oracle, kw):
    """D2D_DONE_FREEZE: one episode per env, finished envs stay exactly as they ended -- on the persistent kernel
    (default geometry) and on the launch-per-stage path (any other geometry)."""
    dev, ref = _pair(pkg, hip, oracle, 12, **kw)
    oracle.lib.d2d_oracle_set_threads(8)
    try:
        for _ in range(6):
            dev.closed_loop(40, freeze_done=True)
            ref.closed_loop(40, freeze_done=True)
            _assert_same(dev, ref, 'freeze')
    finally:
        oracle.lib.d2d_oracle_set_threads(1)
    done = ref.state.flags[:, 3].bool()
    assert done.any()
    steps = ref.state.counters[:, pkg._abi.C_STEPS]
    assert int(steps[done].max()) < 240          # they stopped where their episode ended


def test_continue_mode_past_the_longest_episode(pkg, hip, oracle):
    """D2D_DONE_CONTINUE keeps stepping finished envs, also past max_flight_time (where Oxford's time table ends and
    the policy holds the yaw): device == oracle throughout."""
    dev, ref = _pair(pkg, hip, oracle, 6, agent_number=8, agent_radius=10, agent_max_speed=20, map_id=55, max_flight_time=3)
    for _ in range(5):
        dev.closed_loop(12)
        ref.closed_loop(12)
        _assert_same(dev, ref, 'continue')
    assert int(ref.state.counters[:, pkg._abi.C_STEPS].min()) == 60


def test_full_size_closed_loop_properties(pkg, hip):
    """BASELINE config 2 with the plugins on the device: 4096 envs, Oxford + Primitive.  Size-independent properties:
    copies of one world stay identical, actions are yaw-rate candidates, trajectories are consistent."""
    from drone2d_amd import vec_env, host_init
    p = pkg.Params(planner='Primitive', gaze_method='Oxford', agent_number=10, agent_radius=15, agent_max_speed=20,
                   drone_max_speed=40, map_id=1)
    worlds = [host_init.init_world(_with_map(p, 1 + (i % 64))) for i in range(64)]
    B = 4096
    env = vec_env.VecDrone2DEnv(p, B, backend=hip, planner='Primitive', device_plugins=True, gaze='Oxford',
                                worlds=[worlds[i % 64] for i in range(B)])
    env.closed_loop(120, auto_reset=True)
    env.sync()
    for name in ('drone', 'counters', 'dmap', 'gt', 'kf'):
        t = env.state.t[name]
        assert torch.equal(t[:64], t[64:128]) and torch.equal(t[:64], t[B - 64:]), name
    a = env.state.action.cpu().numpy()
    cand = np.concatenate([np.arange(-80, 80, 80 / 3) / 80, [0.0]])
    assert np.isin(a, cand).all()
    hd = env.plugins.t['traj_hdr'].cpu().numpy()
    assert (hd[:, 0] <= hd[:, 1]).all() and (hd[:, 1] % 20 == 0).all()
    assert int(env.plugins.t['plan_stat'][:, 3].sum()) == 0 and int(env.plugins.t['plan_stat'][:, 0].min()) >= 1
    # README episode (map_id 1) ends at step 210 with the goal reached: env 0 is that world
    fx = replay.load('readme_oxford_primitive')
    assert np.array_equal(env.state.drone[0, :3].cpu().numpy(), fx['t_drone'][119])


def test_replicas_stay_identical_at_65536_envs(pkg, hip):
    """65 536 envs (20 GB of plugin scratch: every per-env offset beyond 32 bits): 256 worlds tiled over the batch, 150
    closed-loop steps with auto reset -- every env must equal the first replica of its world in the env state AND the plugin
    state, which an index overflow anywhere in the high envs would break."""
    from drone2d_amd import vec_env, host_init
    p = pkg.Params(planner='Primitive', gaze_method='Oxford', agent_number=10, agent_radius=15, agent_max_speed=20,
                   drone_max_speed=40, map_id=1)
    K, B = 256, 65536
    worlds = [host_init.init_world(_with_map(p, 1 + i)) for i in range(K)]
    env = vec_env.VecDrone2DEnv(p, B, backend=hip, planner='Primitive', device_plugins=True, gaze='Oxford',
                                worlds=[worlds[i % K] for i in range(B)])
    for _ in range(3):
        env.closed_loop(50, auto_reset=True)
    env.sync()
    for name in ('drone', 'counters', 'dmap', 'gt', 'kf', 'agents', 'active', 'flags', 'action', 'wp'):
        t = env.state.t[name]
        t = t.view(B // K, K, *t.shape[1:])
        assert bool((t == t[:1]).all()), name
    for name in ('traj_hdr', 'seen_step', 'trk_radius', 'trk_prev', 'plan_stat'):
        t = env.plugins.t[name]
        t = t.view(B // K, K, *t.shape[1:])
        assert bool((t == t[:1]).all()), name
    assert int(env.plugins.t['plan_stat'][:, 3].sum()) == 0
    # and the first replicas are what a 256-env batch computes
    small = vec_env.VecDrone2DEnv(p, K, backend=hip, planner='Primitive', device_plugins=True, gaze='Oxford', worlds=worlds)
    for _ in range(3):
        small.closed_loop(50, auto_reset=True)
    small.sync()
    for name in ('drone', 'counters', 'dmap', 'kf', 'flags'):
        assert torch.equal(env.state.t[name][:K], small.state.t[name]), name
    assert torch.equal(env.plugins.t['traj_hdr'][:K], small.plugins.t['traj_hdr'])


@pytest.mark.parametrize('label,B,K,kw', [
    ('config3', 65536, 64, dict(agent_number=50, agent_radius=10, agent_max_speed=40, static_map='maps/random_map_0.npy')),
    ('config4_shard', 32768, 128, dict(agent_number=10, agent_radius=15, agent_max_speed=20, static_map='maps/obstacle_map.npy')),
])
def test_baseline_configs_closed_loop_at_full_size(pkg, hip, label, B, K, kw):
    """BASELINE configs 3 (65536 envs x 172 agents) and 4 (one GPU's shard: 32768 envs x 24 agents) with Oxford + Primitive on
    the device, at full size: K seeded worlds tiled over the batch, 60 closed-loop steps with auto reset; every env equals
    the first replica of its world, and the first replicas equal a K-env run (which the oracle checks at small sizes)."""
    from drone2d_amd import vec_env, host_init
    p = pkg.Params(planner='Primitive', gaze_method='Oxford', drone_max_speed=40, map_id=1, **kw)
    worlds = [host_init.init_world(pkg.with_defaults(_with_map(p, 1 + i))) for i in range(K)]
    env = vec_env.VecDrone2DEnv(p, B, backend=hip, planner='Primitive', device_plugins=True, gaze='Oxford',
                                worlds=[worlds[i % K] for i in range(B)])
    small = vec_env.VecDrone2DEnv(p, K, backend=hip, planner='Primitive', device_plugins=True, gaze='Oxford', worlds=worlds)
    for _ in range(2):
        env.closed_loop(30, auto_reset=True)
        small.closed_loop(30, auto_reset=True)
    env.sync()
    for name in ('drone', 'counters', 'dmap', 'gt', 'kf', 'agents', 'active', 'flags', 'action', 'wp', 'hit'):
        t = env.state.t[name]
        assert bool((t.view(B // K, K, *t.shape[1:]) == small.state.t[name].unsqueeze(0)).all()), f'{label}: {name}'
    for name in ('traj_hdr', 'seen_step', 'trk_radius', 'trk_prev'):
        t = env.plugins.t[name]
        assert bool((t.view(B // K, K, *t.shape[1:]) == small.plugins.t[name].unsqueeze(0)).all()), f'{label}: {name}'
    assert int(env.plugins.t['plan_stat'][:, 3].sum()) == 0 and env.N == worlds[0]['N']


@pytest.mark.parametrize('label,kw', [
    ('config3', dict(agent_number=50, agent_radius=10, agent_max_speed=40, static_map='maps/random_map_0.npy')),
    ('config4', dict(agent_number=10, agent_radius=15, agent_max_speed=20, static_map='maps/obstacle_map.npy')),
    # the upper end of the whole-grid kernel (SPEC 2: up to 40 agents) and the first agent count past it, on the default geometry
    ('n40', dict(agent_number=40, agent_radius=10, agent_max_speed=40)),
    ('n26_obstacles', dict(agent_number=26, agent_radius=8, agent_max_speed=30, static_map='maps/obstacle_map.npy')),
    ('n41', dict(agent_number=41, agent_radius=8, agent_max_speed=40)),
])
def test_baseline_configs_closed_loop_vs_oracle(pkg, hip, oracle, label, kw):
    """The same two configurations (and the agent counts around the whole-grid kernel's limit) against the oracle: 6 seeded worlds,
    80 closed-loop steps with auto reset, every field."""
    dev, ref = _pair(pkg, hip, oracle, 6, drone_max_speed=40, map_id=1, **kw)
    for _ in range(2):
        dev.closed_loop(40, auto_reset=True)
        ref.closed_loop(40, auto_reset=True)
        _assert_same(dev, ref, label)


def _with_map(p, map_id):
    import copy
    q = copy.copy(p)
    q.map_id = map_id
    return q


@pytest.mark.parametrize('layout', ['rowmajor', 'tiled'])
def test_config5_closed_loop_vs_oracle(pkg, hip, oracle, layout):
    """BASELINE config 5's geometry (640 x 640 cells, 640 rays, 100 agents) with Oxford and Primitive ON THE DEVICE -- refused until
    round 3 (-4: the W x H pairwise-summation plan did not fit the wave's LDS).  The gaze stage now finds the blocks of numpy's
    pairwise sum that hold a non-zero term by walking the recursion (sparse path, d2d_plugins.h), the planner probes the explored
    map in HBM.  4 worlds x 40 steps with auto reset, drones started next to agents of their worlds (trackers, replans), both grid
    layouts: every field of the env and plugin state equals the oracle's, bit for bit."""
    from drone2d_amd import vec_env
    from test_gpu_vs_oracle import CFG5, _cfg5_poses
    p = pkg.Params(planner='Primitive', gaze_method='Oxford', drone_max_speed=40, map_id=5, **CFG5)
    ref = vec_env.VecDrone2DEnv(p, 4, backend=oracle, planner='Primitive', device_plugins=True, gaze='Oxford')
    worlds = _worlds(ref)
    dev = vec_env.VecDrone2DEnv(p, 4, backend=hip, planner='Primitive', device_plugins=True, gaze='Oxford', worlds=worlds,
                                grid_layout=layout)
    assert dev.cfg.W == 640 and dev.cfg.R == 640 and dev.cfg.grid_tile == (16 if layout == 'tiled' else 0)
    # three drones start 45 px below an agent of their world, looking at it (yaw 270 looks along +y), one in the open field
    from drone2d_amd import _abi as A
    ag = ref.state.agents
    xy = torch.stack([ag[:, A.A_PX, 7].floor() + 6.0, ag[:, A.A_PY, 7].floor() - 45.0], dim=1).clamp(40.0, 6360.0)
    xy[0] = torch.tensor([3200., 3200.])
    for env in (dev, ref):
        env.state.drone[:, :2] = xy.to(env.device)
    tracked = 0
    oracle.lib.d2d_oracle_set_threads(8)
    try:
        for t in range(0, 40, 4):
            dev.closed_loop(4, auto_reset=True)
            ref.closed_loop(4, auto_reset=True)
            _assert_same(dev, ref, f'config 5 closed loop ({layout}) after step {t + 4}')
            tracked = max(tracked, int(ref.state.active.sum()))
    finally:
        oracle.lib.d2d_oracle_set_threads(1)
    assert int(ref.plugins.t['plan_stat'][:, 0].sum()) >= 4 and tracked > 0   # searches ran, trackers were active on the way


def test_config5_closed_loop_at_shard_scale(pkg, hip):
    """BASELINE config 5 with Oxford + Primitive on the device at ONE GPU's shard of the 262144-env job: 32768 envs x 100 agents on
    640 x 640 cells.  The plugin state is large there -- `seen_step` [B][640][640] int32 = 53.7 GB, 238 KB of search nodes per env,
    26.8 GB of grids (twice with the reset snapshot) -- so a 32-bit offset anywhere in the plugin stages would show exactly here.
    4 worlds tiled, drones started next to agents of their worlds (trackers, replans, searches), 8 closed-loop steps with auto
    reset: every env equals the 4-env run in env state AND plugin state."""
    from drone2d_amd import vec_env
    from drone2d_amd import _abi as A
    from test_gpu_vs_oracle import CFG5
    B = 32768
    free = torch.cuda.mem_get_info()[0]
    if free < 150 << 30:
        pytest.skip(f'needs 150 GB of free device memory, {free >> 30} GB are free')
    p = pkg.Params(planner='Primitive', gaze_method='Oxford', drone_max_speed=40, map_id=5, **CFG5)
    from drone2d_amd import host_init
    worlds = [host_init.init_world(pkg.with_defaults(_with_map(p, 5 + i))) for i in range(4)]
    small = vec_env.VecDrone2DEnv(p, 4, backend=hip, planner='Primitive', device_plugins=True, gaze='Oxford', worlds=worlds)
    big = vec_env.VecDrone2DEnv(p, B, backend=hip, planner='Primitive', device_plugins=True, gaze='Oxford',
                                worlds=[worlds[i % 4] for i in range(B)])
    assert big.plugins.t['seen_step'].numel() * 4 > 2 ** 35 and big.state.gt.numel() > 2 ** 33
    ag = small.state.agents
    xy = torch.stack([ag[:, A.A_PX, 7].floor() + 6.0, ag[:, A.A_PY, 7].floor() - 45.0], dim=1).clamp(40.0, 6360.0)
    xy[0] = torch.tensor([3200., 3200.], device=xy.device)
    small.state.drone[:, :2] = xy
    big.state.drone[:, :2] = xy.repeat(B // 4, 1)
    for _ in range(2):
        big.closed_loop(4, auto_reset=True)
        small.closed_loop(4, auto_reset=True)
    big.sync()
    small.sync()

    def same(x, y, name):
        for c0 in range(0, B, 2048):                      # in slices: comparing a 53.7 GB field whole would allocate 13 GB of bools
            xs = x[c0:c0 + 2048]
            assert bool((xs.view(xs.shape[0] // 4, 4, *x.shape[1:]) == y.unsqueeze(0)).all()), f'{name}: envs {c0}..'
    for name in FIELDS + ('action', 'plan_ok', 'wp_valid', 'wp'):
        same(big.state.t[name], small.state.t[name], name)
    for name in PLUGIN_FIELDS + ('traj', 'traj_box', 'plan_stat'):
        same(big.plugins.t[name], small.plugins.t[name], name)
    assert int(small.plugins.t['plan_stat'][:, 0].sum()) >= 4 and int(small.state.active.sum()) > 0   # searches ran, trackers are active
    assert int(big.plugins.t['plan_stat'][:, 3].sum()) == 0
    del big
    torch.cuda.empty_cache()


def test_exact_only_gaze_build(hip):
    """The gaze stage settles four steps in five with sums formed in any order and falls back to numpy's pairwise sums only when two
    candidates are close (d2d_plugins.h, "the quick decision").  csrc/libd2d_hip_exact.so is the same library with that block
    compiled out: the closed-loop parity cases and the reference episodes run through it in a child process (D2D_LIB), so the exact
    path is exercised on EVERY step, not on 3 % of them."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, 'gym-drone2d-activeperception_amd', 'csrc', 'libd2d_hip_exact.so')
    assert os.path.isfile(lib), 'build it with __graft_entry__.build()'
    env = dict(os.environ, D2D_LIB=lib)
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-m', 'gpu', '-q', '-x', '-k',
                        'test_closed_loop_matches_oracle or test_hip_plugins_reproduce_the_reference_episode or test_config5_closed_loop_vs_oracle'],
                       env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert ' passed' in r.stdout and 'skipped' not in r.stdout.splitlines()[-1], r.stdout[-500:]
