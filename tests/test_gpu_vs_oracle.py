"""HIP path vs the CPU oracle on seeded random batches, step for step (run with -m gpu).
Every field of the state is compared bit for bit (integers and fp64 alike)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FIELDS = ('agents', 'dyn_prev', 'gt', 'dmap', 'drone', 'target', 'counters', 'active', 'kf', 'kf_len', 'hit',
          'newly', 'flags', 'obs_local', 'obs_yaw')


def _pair(pkg, hip, oracle, B, **pk):
    from drone2d_amd import vec_env
    planner = pk.pop('planner', 'NoMove')
    layout = pk.pop('grid_layout', None)       # device layout of the grids (None: the library's choice); the oracle is row-major
    p = pkg.Params(planner=planner, **pk)
    ref = vec_env.VecDrone2DEnv(p, B, backend=oracle)
    dev = vec_env.VecDrone2DEnv(p, B, backend=hip, worlds=_worlds(ref), grid_layout=layout)
    return dev, ref


def _worlds(env):
    """Re-use the host-built worlds of `env` (saves building them twice)."""
    s = env.init_state
    out = []
    for e in range(env.num_envs):
        out.append(dict(agents=s.agents[e].numpy(), agent_unit=s.agent_unit[e].numpy(), dyn_prev=s.dyn_prev[e].numpy(),
                        gt=s.gt[e].numpy(), dmap=s.dmap[e].numpy(), drone=s.drone[e].numpy(),
                        target=s.target[e].numpy(), targets=s.targets[e].numpy(), counters=s.counters[e].numpy(),
                        tracker_radius=env.tracker_radius[e].numpy(), N=env.cfg.N, T=env.cfg.T))
    return out


def _assert_same(dev, ref, tag):
    dev.sync()
    for name in FIELDS:
        a, b = dev.state.logical(name).cpu(), ref.state.t[name]      # grids in the reference's [W][H] whatever the device layout
        if not torch.equal(a, b):
            bad = (a != b).nonzero()
            raise AssertionError(f'{tag}: field {name} differs at {bad[:5].tolist()} ({len(bad)} elements)')


CASES = [
    dict(B=37, T=40, agent_number=10, agent_radius=15, agent_max_speed=20, map_id=100),
    dict(B=9, T=25, agent_number=30, agent_radius=-1, agent_max_speed=60, map_id=7, pillar_number=3),
    dict(B=5, T=12, agent_number=50, agent_radius=10, agent_max_speed=40, map_id=0, static_map='maps/random_map_0.npy'),
    dict(B=6, T=20, agent_number=70, agent_radius=6, agent_max_speed=40, map_id=3, map_size=[1000, 800],
         drone_view_depth=120, drone_view_range=200, init_pos=[420, 400], target_list=[[900, 700], [100, 100]]),
    dict(B=8, T=30, agent_number=6, agent_radius=12, agent_max_speed=4, map_id=11),
    dict(B=4, T=15, agent_number=0, agent_radius=10, agent_max_speed=20, map_id=2),
    # 260 x 260 cells: no LDS coverage bitmap (loop fallback), 260 rays = 5 lane passes, wide view
    dict(B=2, T=6, agent_number=24, agent_radius=22, agent_max_speed=60, map_id=4, map_size=[2600, 2600],
         drone_view_depth=150, drone_view_range=170, init_pos=[1300, 1200], target_list=[[2500, 2500]]),
    # big agents: dynamic blocks of 5 x 5 / 7 x 7 cells (generic dyn path), scale 20
    dict(B=5, T=20, agent_number=9, agent_radius=45, agent_max_speed=50, map_id=8, map_scale=20, map_size=[1000, 1000],
         init_pos=[500, 480], target_list=[[900, 900]]),
]


@pytest.mark.parametrize('case', CASES, ids=lambda c: f"N{c['agent_number']}_B{c['B']}")
def test_fused_step_matches_oracle(pkg, hip, oracle, case):
    case = dict(case)
    B, T = case.pop('B'), case.pop('T')
    dev, ref = _pair(pkg, hip, oracle, B, **case)
    rng = np.random.RandomState(B * 1000 + T)
    for t in range(T):
        a = rng.uniform(-1, 1, B)
        if t % 7 == 3:   # teleport some drones (external mutation API, validation_speed.py:135-138)
            xy = torch.from_numpy(np.stack([rng.randint(15, dev.cfg.W_px - 15, B), rng.randint(15, dev.cfg.H_px - 15, B)], 1).astype(np.float64))
            dev.state.drone[:, :2] = xy.to(dev.device)
            ref.state.drone[:, :2] = xy
        dev.step(a)
        ref.step(a)
        _assert_same(dev, ref, f'step {t + 1}')


def test_external_planner_inputs_and_split_halves(pkg, hip, oracle):
    """Random plan_ok / waypoint inputs (brake + follow branches), perceive()+act() on the device vs
    the fused oracle step."""
    B, T = 16, 40
    dev, ref = _pair(pkg, hip, oracle, B, planner='Primitive', agent_number=12, agent_radius=10, agent_max_speed=30, map_id=50)
    rng = np.random.RandomState(3)
    for t in range(T):
        a = rng.uniform(-1, 1, B)
        ok = rng.rand(B) < 0.7
        valid = ok & (rng.rand(B) < 0.8)
        wp = np.concatenate([rng.uniform(20, 480, (B, 2)).round() + rng.choice([0.0, 0.5], (B, 2)),
                             rng.uniform(-40, 40, (B, 2)), np.zeros((B, 2))], axis=1)
        for env in (dev, ref):
            env.set_plan(ok, valid, wp)
        dev.perceive()
        dev.act(a)
        ref.step(a)
        _assert_same(dev, ref, f'step {t + 1}')


def test_rollout_and_reset_on_device(pkg, hip, oracle):
    B, T = 21, 30
    dev, ref = _pair(pkg, hip, oracle, B, agent_number=20, agent_radius=10, agent_max_speed=40, map_id=5)
    rng = np.random.RandomState(9)
    acts = rng.uniform(-1, 1, (T, B))
    pin = np.stack([rng.randint(30, 470, B), rng.randint(30, 470, B)], 1).astype(np.float64)
    cd = dev.rollout(acts, pin=pin, collisions=True)
    cr = ref.rollout(acts, pin=pin, collisions=True)
    dev.sync()
    assert torch.equal(cd.cpu(), cr)
    _assert_same(dev, ref, 'after rollout')
    # the same chain split over 3 independent streams (ragged sub-batches of 7 envs)
    dev2, _ = _pair(pkg, hip, oracle, B, agent_number=20, agent_radius=10, agent_max_speed=40, map_id=5)
    cd2 = dev2.rollout(acts, pin=pin, collisions=True, streams=3)
    dev2.sync()
    assert torch.equal(cd2.cpu(), cr)
    _assert_same(dev2, ref, 'after 3-stream rollout')
    mask = torch.from_numpy((rng.rand(B) < 0.5).astype(np.uint8))
    dev.reset(mask)
    ref.reset(mask)
    _assert_same(dev, ref, 'after masked reset')
    dev.step(acts[0]); ref.step(acts[0])
    _assert_same(dev, ref, 'step after reset')


@pytest.mark.parametrize('label,B,kw', [
    ('config3', 65536, dict(agent_number=50, agent_radius=10, agent_max_speed=40, map_id=0, static_map='maps/random_map_0.npy')),
    ('config4_shard', 32768, dict(agent_number=10, agent_radius=15, agent_max_speed=20, map_id=1, static_map='maps/obstacle_map.npy')),
])
def test_baseline_config_sizes(pkg, hip, label, B, kw):
    """BASELINE configs 3 (65536 envs x 172 agents) and 4 (one GPU's shard: 32768 envs x 24 agents) at full size:
    identical worlds stay identical, walls never change, the explored map only grows and agrees with gt."""
    from drone2d_amd import vec_env, host_init
    p = pkg.Params(planner='NoMove', **kw)
    w = host_init.init_world(pkg.with_defaults(p))
    env = vec_env.VecDrone2DEnv(p, B, backend=hip, worlds=[w] * B)
    gt0 = env.state.gt[0].clone()
    prev = torch.zeros_like(env.state.dmap[0], dtype=torch.bool)
    for t in range(4):
        env.step(0.37 * (t - 1))
        s = env.state
        for name in ('agents', 'gt', 'dmap', 'drone', 'flags', 'hit', 'obs_local', 'kf'):
            x = s.t[name]
            assert bool((x == x[0:1]).all()), f'{label}: {name} diverged across identical envs at step {t + 1}'
        assert torch.equal(s.gt[0] == 1, gt0 == 1)
        ex = s.dmap[0] != 0
        assert bool((prev <= ex).all()) and bool(((s.dmap[0] == 1) <= (s.gt[0] == 1)).all())
        prev = ex
    assert env.N == w['N'] and int(prev.sum()) > 0


CFG5 = dict(agent_number=100, agent_radius=15, agent_max_speed=40, map_size=[6400, 6400], init_pos=[3200, 3200],
            target_list=[[6000, 6000]])


def _cfg5_worlds(pkg, n):
    from drone2d_amd import host_init
    return [host_init.init_world(pkg.with_defaults(pkg.Params(planner='NoMove', map_id=5 + i, **CFG5))) for i in range(n)]


def _cfg5_poses(worlds, t):
    """Drone positions of step t for the 4 worlds: open field, a corner (two border walls in view, static collision),
    and next to an agent of the env's own world (rays hit it, a tracker starts)."""
    from drone2d_amd import _abi as A
    out = []
    for i, w in enumerate(worlds):
        k = (7 * i + 3 * t) % w['N']
        ax, ay = float(w['agents'][A.A_PX, k]), float(w['agents'][A.A_PY, k])
        out.append([[3200., 3200.], [17., 6381.], [int(ax) + 10., int(ay) - 45.], [6383., 12.]][(i + t) % 4])
    return torch.tensor(out, dtype=torch.float64)


@pytest.mark.parametrize('layout', ['rowmajor', 'tiled'])
def test_config5_geometry_vs_oracle(pkg, hip, oracle, layout):
    """(Both device layouts of the grids: the reference's row-major [W][H] and 16 x 16-cell tiles, d2d_cfg.grid_tile.)
    BASELINE config 5's geometry (640 x 640 cells, 640 rays = 10 lane passes per env, 100 agents; the generic kernel
    instantiation, no LDS coverage bitmap, 400 KB grids per env): 4 worlds x 6 steps, device == oracle in every field.
    The oracle itself replays the reference's own trace at this geometry (golden `nomove_cfg5_640`)."""
    from drone2d_amd import vec_env
    worlds = _cfg5_worlds(pkg, 4)
    p = pkg.Params(planner='NoMove', map_id=5, **CFG5)
    ref = vec_env.VecDrone2DEnv(p, 4, backend=oracle, worlds=worlds)
    dev = vec_env.VecDrone2DEnv(p, 4, backend=hip, worlds=worlds, grid_layout=layout)
    assert dev.cfg.R == 640 and dev.cfg.W == 640 and dev.cfg.N == 100 and dev.cfg.grid_tile == (16 if layout == 'tiled' else 0)
    rng = np.random.RandomState(5)
    for t in range(6):
        a = rng.uniform(-1, 1, 4)
        if t in (1, 2, 4):
            xy = _cfg5_poses(worlds, t)
            dev.state.drone[:, :2] = xy.to(dev.device)
            ref.state.drone[:, :2] = xy
        dev.step(a)
        ref.step(a)
        dev.sync()
        for name in FIELDS:
            assert torch.equal(dev.state.logical(name).cpu(), ref.state.t[name]), f'config 5: {name} at step {t + 1}'
    assert int(ref.state.hit.sum()) > 0 and int((ref.state.flags[:, 0] == 1).sum()) > 0    # rays hit agents, a wall collision


def test_config5_replicas_at_shard_scale(pkg, hip):
    """Config 5 at ONE GPU's shard of the 262144-env job: 32768 envs x 100 agents on 640 x 640 cells = 26.8 GB of grids (53.6 GB
    with the reset snapshot) -- the size where a 32-bit index or an allocation limit would show.  Every env equals the 4-env run
    of its world, bit for bit (3 steps, one of them from teleported poses), and a masked reset restores the snapshot."""
    from drone2d_amd import vec_env
    worlds = _cfg5_worlds(pkg, 4)
    p = pkg.Params(planner='NoMove', map_id=5, **CFG5)
    B = 32768
    free = torch.cuda.mem_get_info()[0]
    if free < 70 << 30:
        pytest.skip(f'needs 70 GB of free device memory, {free >> 30} GB are free')
    big = vec_env.VecDrone2DEnv(p, B, backend=hip, worlds=[worlds[i % 4] for i in range(B)])
    small = vec_env.VecDrone2DEnv(p, 4, backend=hip, worlds=worlds)
    assert big.state.gt.numel() == B * 640 * 640 > 2 ** 33                 # byte offsets beyond 32 bits
    rng = np.random.RandomState(6)
    for t in range(3):
        a4 = torch.from_numpy(rng.uniform(-1, 1, 4))
        if t == 1:
            xy = _cfg5_poses(worlds, t).to(big.device)
            big.state.drone[:, :2] = xy.repeat(B // 4, 1)
            small.state.drone[:, :2] = xy
        big.step(a4.repeat(B // 4))
        small.step(a4)
    big.sync()

    def same_as_small(env_big, env_small):
        for name in FIELDS:
            x, y = env_big.state.t[name], env_small.state.t[name].unsqueeze(0)
            for c0 in range(0, B, 4096):                                   # in slices: the comparison of a 13.4 GB field allocates
                xs = x[c0:c0 + 4096]
                assert bool((xs.view(xs.shape[0] // 4, 4, *x.shape[1:]) == y).all()), f'{name} envs {c0}..'
    same_as_small(big, small)
    assert int(small.state.hit.sum()) > 0
    # the LAST envs of the shard (highest addresses) hold what the first hold, and a masked reset of the upper half restores it
    mask = torch.zeros(B, dtype=torch.uint8)
    mask[B // 2:] = 1
    big.reset(mask)
    big.sync()
    for name in ('gt', 'dmap', 'agents', 'drone'):
        assert torch.equal(big.state.t[name][B - 4:], big.init_state.t[name][B - 4:]), name
        assert torch.equal(big.state.t[name][:4], small.state.t[name]), name
    del big
    torch.cuda.empty_cache()


def test_per_step_measurement_noise_rows(pkg, hip, oracle):
    """var_cam != 0 in multi-step calls: step t draws from row t % T of a [T, B, N, 2] noise tensor (the reference draws fresh
    normals every step, utils.py:605) -- d2d_rollout and the persistent d2d_closed_loop, device vs oracle, every field."""
    from drone2d_amd import vec_env
    rng = np.random.RandomState(11)
    dev, ref = _pair(pkg, hip, oracle, 5, agent_number=12, agent_radius=12, agent_max_speed=30, map_id=31, var_cam=2,
                     init_pos=[250, 250])
    T = 9
    noise = rng.standard_normal((4, 5, dev.cfg.N, 2))            # 4 rows for 9 steps: rows wrap around
    acts = rng.uniform(-1, 1, (T, 5))
    for env in (dev, ref):
        env.set_noise(noise)
    # the drones pinned next to an agent of their world, looking at it (yaw 270 looks along +y): rays hit, trackers start
    ag = ref.state.agents
    pin = torch.stack([ag[:, 0, 3].floor() + 10.0, ag[:, 1, 3].floor() - 45.0], dim=1).clamp(30.0, 470.0)
    acts = acts * 0.2
    cd = dev.rollout(acts, pin=pin, collisions=True)
    cr = ref.rollout(acts, pin=pin, collisions=True)
    dev.sync()
    assert torch.equal(cd.cpu(), cr) and int(ref.state.active.sum()) > 0
    _assert_same(dev, ref, 'rollout with noise rows')
    # the same rows are NOT what a single reused row gives (the test would pass trivially otherwise)
    one = vec_env.VecDrone2DEnv(dev.params, 5, backend=oracle, worlds=_worlds(ref))
    one.set_noise(noise[0])
    one.rollout(acts, pin=pin)
    assert not torch.equal(one.state.kf, ref.state.kf)
    # closed loop (persistent kernel on the device): Primitive under a constant gaze
    p = pkg.Params(planner='Primitive', gaze_method='Rotating', agent_number=12, agent_radius=12, agent_max_speed=30, map_id=31,
                   var_cam=2, init_pos=[250, 250], drone_max_speed=40)
    r2 = vec_env.VecDrone2DEnv(p, 4, backend=oracle, planner='Primitive', device_plugins=True, gaze='Rotating')
    d2 = vec_env.VecDrone2DEnv(p, 4, backend=hip, planner='Primitive', device_plugins=True, gaze='Rotating', worlds=_worlds(r2))
    n2 = rng.standard_normal((7, 4, d2.cfg.N, 2))
    for env in (d2, r2):
        env.set_noise(n2)
    r2.closed_loop(25, auto_reset=True)
    for n in (10, 1, 14):                      # cut into several launches: the rows go on where the last call stopped (noise_row0)
        d2.closed_loop(n, auto_reset=True)
    _assert_same(d2, r2, 'closed loop with noise rows')


def test_single_wave_above_64k_of_lds(pkg, hip, oracle):
    """1400 agents: the per-env LDS working set (73 KB) is above the 64 KB a workgroup gets by default -- round 1 refused such a
    configuration (-4); now the wave gets a workgroup of its own with the kernel opted into more dynamic LDS
    (hipFuncAttributeMaxDynamicSharedMemorySize).  Device vs oracle, every field."""
    from drone2d_amd import _lib
    dev, ref = _pair(pkg, hip, oracle, 3, agent_number=1400, agent_radius=4, agent_max_speed=20, map_id=2, map_size=[1000, 1000],
                     init_pos=[500, 500], target_list=[[900, 900]])
    wpb, lds, per_cu, spec = _lib.launch_shape(dev.cfg)
    assert wpb == 1 and lds > 64 * 1024 and per_cu >= 2 and spec == 0
    rng = np.random.RandomState(3)
    for t in range(4):
        a = rng.uniform(-1, 1, 3)
        dev.step(a)
        ref.step(a)
        _assert_same(dev, ref, f'1400 agents step {t + 1}')
    assert int(ref.state.hit.sum()) > 0


def test_full_size_properties(pkg, hip):
    """BASELINE config 2 size (4096 envs x 10 agents): size-independent properties on the device alone:
    (i) a batch of identical worlds stays identical, (ii) wall cells of gt never change, explored cells only
    grow, every explored cell agrees with gt's wall/free classification, (iii) reset is idempotent."""
    from drone2d_amd import vec_env, host_init
    p = pkg.Params(planner='NoMove', agent_number=10, agent_radius=15, agent_max_speed=20, map_id=1)
    w = host_init.init_world(pkg.with_defaults(p))
    B = 4096
    env = vec_env.VecDrone2DEnv(p, B, backend=hip, worlds=[w] * B)
    gt0 = env.state.gt.clone()
    prev_explored = torch.zeros_like(env.state.dmap, dtype=torch.bool)
    rng = np.random.RandomState(0)
    for t in range(20):
        env.step(float(rng.uniform(-1, 1)))
        s = env.state
        for name in ('agents', 'gt', 'dmap', 'drone', 'flags', 'obs_local', 'hit'):
            x = s.t[name]
            assert bool((x == x[0:1]).all()), f'{name} diverged across identical envs at step {t + 1}'
        assert torch.equal(s.gt == 1, gt0 == 1)
        explored = s.dmap != 0
        assert bool((prev_explored <= explored).all())
        assert bool(((s.dmap == 1) <= (s.gt == 1)).all()) and bool(((s.dmap == 2) <= (s.gt != 1)).all())
        prev_explored = explored
    env.reset()
    a = {k: v.clone() for k, v in env.state.t.items()}
    env.reset()
    for k in ('agents', 'gt', 'dmap', 'drone', 'counters'):
        assert torch.equal(a[k], env.state.t[k])


def test_measurement_noise_input(pkg, hip, oracle):
    """var_cam != 0: the host supplies the normal draws; device and oracle agree bit for bit."""
    B, T = 12, 25
    dev, ref = _pair(pkg, hip, oracle, B, agent_number=14, agent_radius=12, agent_max_speed=30, map_id=31, var_cam=2)
    rng = np.random.RandomState(5)
    for t in range(T):
        a = rng.uniform(-1, 1, B)
        z = rng.randn(B, dev.N, 2)
        for env in (dev, ref):
            env.set_noise(z)
            env.step(a)
        _assert_same(dev, ref, f'step {t + 1}')
    assert int(dev.state.active.sum()) > 0


def test_many_agent_raycast_candidates_at_the_cone_edges(pkg, hip, oracle):
    """More than 32 agents on the default geometry: the raycast keeps only agents within their radius of the cone the rays
    span (csrc ray_cull<CONE>, edges from float trigonometry under a margin) and tests up to 64 of them through a per-ray bit
    mask.  72 standing agents per env on rings around the drone, bunched within +-12 degrees of the two edges of the field of
    view, 48 start yaws, the view swinging over them for 12 steps: hit masks and every other field equal the oracle's
    (which tests every agent at every sample, utils.py:658-662)."""
    from drone2d_amd import _abi as A
    B, N = 48, 72
    dev, ref = _pair(pkg, hip, oracle, B, agent_number=N, agent_radius=8, agent_max_speed=5, map_id=21, init_pos=[250, 250],
                     drone_max_speed=40)
    assert dev.cfg.N > 32 and dev.cfg.W == 50 and dev.cfg.R == 50      # default geometry, more than 32 agents: the many-agent kernel
    fov = np.radians(dev.params.drone_view_range)
    ag = ref.state.agents.clone()
    yaw = torch.zeros(B, dtype=torch.float64)
    for e in range(B):
        yaw[e] = (e * 7.5 + 0.25 * (e % 3)) % 360
        pa = 2 * np.pi - np.radians(float(yaw[e]))
        k = 0
        for edge in (pa - fov / 2, pa + fov / 2):
            for off in (-12, -6, -3, -1, 0, 1, 3, 6, 12):
                for dist in (15.0, 40.0, 70.0, 90.0):
                    a = edge + np.radians(off + 0.37 * (e % 5))
                    ag[e, A.A_PX, k] = 250.0 + dist * np.cos(a)
                    ag[e, A.A_PY, k] = 250.0 + dist * np.sin(a)
                    k += 1
        assert k == N
    ag[:, A.A_VX] = 0.0
    ag[:, A.A_VY] = 0.0
    for env in (dev, ref):
        env.state.agents.copy_(ag)
        env.state.drone[:, A.D_YAW] = yaw.to(env.state.drone.device)
    rng = np.random.RandomState(8)
    hits = 0
    for t in range(12):
        a = rng.choice([-1.0, -0.4, 0.3, 1.0], B)
        dev.step(a)
        ref.step(a)
        _assert_same(dev, ref, f'cone edges step {t + 1}')
        hits += int(ref.state.hit.sum())
    per_env = ref.state.hit.sum(1)
    assert hits > 12 * B * 4 and int(per_env.min()) > 0 and int(per_env.max()) < N
    # a crowd: all 72 agents inside the cone, more candidates than the 64 the LDS list holds -- such an env tests every agent at
    # every sample (Geom.ccap); the nearest ring shadows the others, hit masks equal the oracle's again
    ag2 = ref.state.agents.clone()
    yaw2 = ref.state.drone[:, A.D_YAW].clone()
    for e in range(B):
        pa = 2 * np.pi - np.radians(float(yaw2[e]))
        for k in range(N):
            a = pa + (k % 12 - 5.5) / 12.0 * fov * 0.9
            d = 22.0 + 11.0 * (k // 12) + 0.3 * (e % 7)
            ag2[e, A.A_PX, k] = 250.0 + d * np.cos(a)
            ag2[e, A.A_PY, k] = 250.0 + d * np.sin(a)
    for env in (dev, ref):
        env.state.agents.copy_(ag2)
    for t in range(4):
        a = rng.choice([-0.5, 0.0, 0.5], B)
        dev.step(a)
        ref.step(a)
        _assert_same(dev, ref, f'crowd step {t + 1}')
    assert int(ref.state.hit.sum(1).min()) >= 6


def test_crowd_of_forty_in_the_whole_grid_kernel(pkg, hip, oracle):
    """40 agents on the default geometry take the whole-grid kernel for 17-40 agents, whose candidate list holds 32 (Geom.ccap: five
    workgroups per CU instead of four): with all 40 inside the cone of the rays -- more candidates than the list holds -- every
    sample tests every agent, as the reference does; with 33 of them just the same.  Hit masks and every field equal the oracle's."""
    from drone2d_amd import _abi as A
    B, N = 12, 40
    dev, ref = _pair(pkg, hip, oracle, B, agent_number=N, agent_radius=5, agent_max_speed=5, map_id=23, init_pos=[250, 250], drone_max_speed=40)
    assert dev.cfg.N == N and dev.cfg.W == 50
    fov = np.radians(dev.params.drone_view_range)
    ag = ref.state.agents.clone()
    yaw = torch.tensor([(e * 31.0) % 360 for e in range(B)], dtype=torch.float64)
    for e in range(B):
        pa = 2 * np.pi - np.radians(float(yaw[e]))
        inside = N if e % 2 == 0 else 33                       # the other 7 stand behind the drone
        for k in range(N):
            if k < inside:
                a = pa + (k % 10 - 4.5) / 10.0 * fov * 0.85
                d = 24.0 + 14.0 * (k // 10) + 0.4 * (e % 5)
            else:
                a = pa + np.pi + 0.1 * (k - inside)
                d = 60.0
            ag[e, A.A_PX, k] = 250.0 + d * np.cos(a)
            ag[e, A.A_PY, k] = 250.0 + d * np.sin(a)
    ag[:, A.A_VX] = 0.0
    ag[:, A.A_VY] = 0.0
    for env in (dev, ref):
        env.state.agents.copy_(ag)
        env.state.drone[:, A.D_YAW] = yaw.to(env.state.drone.device)
    rng = np.random.RandomState(9)
    for t in range(6):
        a = rng.choice([-0.5, 0.0, 0.5], B)
        dev.step(a)
        ref.step(a)
        _assert_same(dev, ref, f'crowd of forty step {t + 1}')
    assert int(ref.state.hit.sum(1).min()) >= 5


@pytest.mark.parametrize('radius,speed', [(12, 60), (25, 40)])
def test_two_phase_dynamic_grid_with_packed_agents(pkg, hip, oracle, radius, speed):
    """Grids above 256 x 256 cells update the dynamic cells in two phases (csrc dyn_apply<1> / <2>: every agent clears what its
    previous block leaves behind, fence, every agent fetches its new block again and marks) instead of asking a coverage
    structure.  90 fast agents (two lane passes) packed into a 260 x 260 px corner of a 3000 x 3000 px map, so that blocks overlap
    and neighbours keep clearing cells of each other's new blocks -- 3 x 3 blocks (radius 12) and 5 x 5 blocks (radius 25, the
    general loops): the ground truth equals the oracle's literal clear-all-then-mark (utils.py:527-540) for 20 steps."""
    from drone2d_amd import _abi as A
    B, N = 6, 90
    dev, ref = _pair(pkg, hip, oracle, B, agent_number=N, agent_radius=radius, agent_max_speed=speed, map_id=5,
                     map_size=[3000, 3000], init_pos=[400, 400], target_list=[[2500, 2500]])
    assert dev.cfg.W == 300 and dev.cfg.N == N
    rng = np.random.RandomState(radius)
    ag = ref.state.agents.clone()
    ag[:, A.A_PX] = torch.from_numpy(rng.uniform(120, 380, (B, N)))
    ag[:, A.A_PY] = torch.from_numpy(rng.uniform(120, 380, (B, N)))
    for env in (dev, ref):
        env.state.agents.copy_(ag)
    moved = 0
    for t in range(20):
        a = rng.uniform(-1, 1, B)
        before = ref.state.dyn_prev.clone()
        dev.step(a)
        ref.step(a)
        _assert_same(dev, ref, f'packed agents r={radius} step {t + 1}')
        moved += int((before[:, :, :2] != ref.state.dyn_prev[:, :, :2]).any(2).sum())
    assert moved > 20 * B * N // 6                                  # blocks really change cell all the time
    assert int((ref.state.gt == A.DYNAMIC).sum()) > B * 200


def test_gathered_tracker_filters_with_many_agents(pkg, hip, oracle):
    """More than 128 agents (three lane passes): the Kalman filters that have work -- active, or hit this step -- are gathered
    into one list before they run (csrc st_tracker).  150 fast agents, the view sweeping round at full yaw rate for 60 steps so
    that dozens of trackers start, update, coast without measurements and are archived: tracker state, lengths, active flags,
    buffer counters and everything else equal the oracle's at every step."""
    from drone2d_amd import _abi as A
    B, N = 6, 150
    dev, ref = _pair(pkg, hip, oracle, B, agent_number=N, agent_radius=6, agent_max_speed=40, map_id=17, init_pos=[250, 250],
                     drone_max_speed=40)
    assert dev.cfg.N == N
    most, archived = 0, 0
    for t in range(60):
        a = np.full(B, 1.0 if (t // 20) % 2 == 0 else -0.6)
        dev.step(a)
        ref.step(a)
        _assert_same(dev, ref, f'150 agents step {t + 1}')
        most = max(most, int(ref.state.active.sum(1).max()))
        archived = int(ref.state.counters[:, A.C_BUF_N].sum())
    assert most >= 12 and archived > 0, (most, archived)


def test_tracker_state_buffer_ending_on_a_page_boundary(pkg, hip):
    """4096 envs x 32 agents: the tracker state `kf` is 4096 * 32 * 160 B = exactly ten 2 MiB pages, and 32 agents take the kernel
    that reads tracker state from global memory with a lane per element (st_tracker_elem<false>).  Round 3's first version let
    its four idle lanes read elements 20..23 of a record: past the end of the buffer for the last tracker of the last env -- a GPU
    memory fault at exactly such sizes (found by the config 4 profile run: 32768 * 24 * 160 B = sixty pages).  The batch must run
    and equal the 4-env run of its worlds."""
    from drone2d_amd import vec_env
    p = pkg.Params(planner='NoMove', agent_number=32, agent_radius=8, agent_max_speed=30, map_id=70)
    worlds = vec_env.build_worlds(p, 4)
    B = 4096
    big = vec_env.VecDrone2DEnv(p, B, backend=hip, worlds=[worlds[i % 4] for i in range(B)])
    small = vec_env.VecDrone2DEnv(p, 4, backend=hip, worlds=worlds)
    assert big.state.kf.numel() * 8 % (2 << 20) == 0 and big.cfg.N == 32
    # drones next to an agent, looking at it: rays hit, trackers start and update
    ag = small.state.agents
    pin = torch.stack([ag[:, 0, 3].floor() + 8.0, ag[:, 1, 3].floor() - 40.0], dim=1).clamp(30.0, 470.0)
    rng = np.random.RandomState(3)
    for t in range(12):
        a4 = torch.from_numpy(rng.uniform(-0.3, 0.3, 4))
        for env, reps in ((big, B // 4), (small, 1)):
            env.state.drone[:, :2] = pin.repeat(reps, 1).to(env.device)
            env.step(a4.repeat(reps))
    big.sync()
    assert int(small.state.active.sum()) > 0
    for name in FIELDS:
        x = big.state.t[name]
        assert bool((x.view(B // 4, 4, *x.shape[1:]) == small.state.t[name].unsqueeze(0)).all()), name
