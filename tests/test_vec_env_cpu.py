"""Host logic of VecDrone2DEnv, exercised on CPU with the oracle standing in for the HIP library
(test-only injection; the product default is HipBackend and raises without a GPU)."""
import numpy as np
import pytest
import torch


def _mk(pkg, oracle, B, **kw):
    from drone2d_amd import vec_env
    p = pkg.Params(planner='NoMove', agent_number=10, agent_radius=15, agent_max_speed=20, map_id=kw.pop('map_id', 0))
    return vec_env.VecDrone2DEnv(p, B, backend=oracle, **kw)


def test_default_backend_fails_loudly_without_gpu(pkg):
    from drone2d_amd import vec_env, _lib
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(_lib.D2DError):
        vec_env.VecDrone2DEnv(pkg.Params(planner='NoMove'), 2)


def test_batch_envs_are_independent_and_seeded_by_global_id(pkg, oracle):
    """env i of a batch == a single env built with map_id + i; shards (env_offset) agree too."""
    rng = np.random.RandomState(0)
    acts = rng.uniform(-1, 1, (30, 6))
    big = _mk(pkg, oracle, 6)
    shard = _mk(pkg, oracle, 3, env_offset=3)
    singles = [_mk(pkg, oracle, 1, map_id=i) for i in range(6)]
    for t in range(30):
        big.step(acts[t])
        shard.step(acts[t, 3:])
        for i, s in enumerate(singles):
            s.step(acts[t, i:i + 1])
    for name in ('agents', 'gt', 'dmap', 'drone', 'counters', 'flags', 'hit', 'obs_local', 'kf', 'active'):
        for i, s in enumerate(singles):
            assert torch.equal(big.state.t[name][i], s.state.t[name][0]), name
        assert torch.equal(big.state.t[name][3:], shard.state.t[name]), name


def test_reset_restores_the_seeded_world(pkg, oracle):
    env = _mk(pkg, oracle, 4)
    snap = {k: v.clone() for k, v in env.state.t.items()}
    for t in range(12):
        env.step(np.full(4, 0.3))
    mask = torch.tensor([1, 0, 1, 0], dtype=torch.uint8)
    moved = {k: v.clone() for k, v in env.state.t.items()}
    env.reset(mask)
    for name in ('agents', 'gt', 'dmap', 'drone', 'counters', 'kf', 'kf_len', 'active', 'dyn_prev', 'target'):
        assert torch.equal(env.state.t[name][0], snap[name][0]) and torch.equal(env.state.t[name][2], snap[name][2]), name
        assert torch.equal(env.state.t[name][1], moved[name][1]) and torch.equal(env.state.t[name][3], moved[name][3]), name
    env.reset()
    for name in ('agents', 'gt', 'dmap', 'drone', 'counters'):
        assert torch.equal(env.state.t[name], snap[name]), name


def test_rollout_equals_repeated_steps(pkg, oracle):
    rng = np.random.RandomState(1)
    acts = rng.uniform(-1, 1, (25, 5))
    a, b = _mk(pkg, oracle, 5), _mk(pkg, oracle, 5)
    pin = np.tile(np.array([[200.0, 260.0]]), (5, 1))
    coll = a.rollout(acts, pin=pin, collisions=True)
    for t in range(25):
        b.state.drone[:, 0] = 200.0
        b.state.drone[:, 1] = 260.0
        _, _, _, info = b.step(acts[t])
        assert torch.equal(coll[t], info['collision_flag'])
    for name in ('agents', 'gt', 'dmap', 'drone', 'counters', 'flags', 'obs_local'):
        assert torch.equal(a.state.t[name], b.state.t[name]), name


def test_step_outputs_have_reference_shapes(pkg, oracle):
    env = _mk(pkg, oracle, 3)
    obs, rew, done, info = env.step(0.5)
    assert obs['local_map'].shape == (3, 1, 33, 33) and obs['local_map'].dtype == torch.uint8
    assert obs['swep_map'].data_ptr() == obs['local_map'].data_ptr()      # drone_v2.py:252-253: same map twice
    assert obs['yaw_angle'].shape == (3, 1) and obs['yaw_angle'].dtype == torch.float32
    assert float(obs['yaw_angle'][0, 0]) == 274.0 and rew.shape == (3,) and not done.any()
    assert env.episode_stats().shape == (3, 8)


def test_zero_agents_and_parallel_world_build(pkg, oracle):
    from drone2d_amd import vec_env
    p = pkg.Params(planner='NoMove', agent_number=0, map_id=2)
    env = vec_env.VecDrone2DEnv(p, 3, backend=oracle)
    for t in range(5):
        _, _, done, info = env.step(0.25)
    assert env.N == 0 and not done.any() and int((env.state.dmap != 0).sum()) > 0
    q = pkg.Params(planner='NoMove', agent_number=10, agent_radius=15, agent_max_speed=20, map_id=4)
    a = vec_env.build_worlds(q, 70, workers=2)
    b = vec_env.build_worlds(q, 70, workers=0)
    for x, y in zip(a, b):
        assert np.array_equal(x['agents'], y['agents']) and np.array_equal(x['gt'], y['gt'])


def test_noise_rows_continue_across_chunked_calls(pkg, oracle):
    """A run cut into several multi-step calls draws the noise rows the whole run would (d2d_cfg.noise_row0, ABI 7): before, every
    call started again at row 0 and a chunked run replayed the same measurement noise with the period of the chunk."""
    from drone2d_amd import vec_env
    rng = np.random.RandomState(5)
    p = pkg.Params(planner='Primitive', gaze_method='Rotating', agent_number=12, agent_radius=12, agent_max_speed=30, map_id=31,
                   var_cam=2, init_pos=[250, 250], drone_max_speed=40)
    whole = vec_env.VecDrone2DEnv(p, 3, backend=oracle, planner='Primitive', device_plugins=True, gaze='Rotating')
    parts = vec_env.VecDrone2DEnv(p, 3, backend=oracle, planner='Primitive', device_plugins=True, gaze='Rotating')
    noise = rng.standard_normal((7, 3, whole.cfg.N, 2))
    whole.set_noise(noise)
    parts.set_noise(noise)
    whole.closed_loop(24, auto_reset=True)
    for n in (5, 1, 9, 9):
        parts.closed_loop(n, auto_reset=True)
    assert parts.cfg.noise_row0 == 24 % 7
    for name in ('kf', 'active', 'drone', 'dmap', 'counters'):
        assert torch.equal(whole.state.t[name], parts.state.t[name]), name
    assert int(whole.state.active.sum()) > 0
    # a row index outside the table is refused
    parts.cfg.noise_row0 = 7
    with pytest.raises(Exception):
        parts.closed_loop(1, auto_reset=True)


@pytest.mark.parametrize('layout', ['rowmajor', 'tiled'])
def test_a_batch_that_repeats_worlds_loads_like_one_of_copies(pkg, oracle, layout):
    """A world list that names the same dict many times (a sweep's start cells, a bench batch tiled from fewer worlds) is staged once
    per distinct world and gathered on the device: every field equals the load of a list of separate copies, in both grid layouts;
    and the batch steps like it."""
    import copy
    from drone2d_amd import vec_env
    from drone2d_amd.state import distinct_worlds
    p = pkg.Params(planner='NoMove', agent_number=7, agent_radius=10, agent_max_speed=20, map_id=3)
    base = vec_env.build_worlds(p, 3)
    order = [0, 2, 2, 1, 0, 0, 2, 1, 1, 0, 2]
    shared = [base[i] for i in order]
    copies = [copy.deepcopy(base[i]) for i in order]
    d, ix = distinct_worlds(shared)
    assert [id(w) for w in d] == [id(base[0]), id(base[2]), id(base[1])] and ix == [0, 1, 1, 2, 0, 0, 1, 2, 2, 0, 1]
    assert len(distinct_worlds(copies)[0]) == len(order)          # by identity: separate copies stay separate
    a = vec_env.VecDrone2DEnv(p, len(order), backend=oracle, worlds=shared, grid_layout=layout)
    b = vec_env.VecDrone2DEnv(p, len(order), backend=oracle, worlds=copies, grid_layout=layout)
    fields = ('agents', 'agent_unit', 'dyn_prev', 'gt', 'dmap', 'drone', 'target', 'targets', 'counters')
    for name in fields:
        assert torch.equal(a.state.t[name], b.state.t[name]), name
    assert torch.equal(a.tracker_radius, b.tracker_radius)
    for i, k in enumerate(order):
        assert np.array_equal(a.state.logical('gt')[i].numpy(), base[k]['gt'])
    if layout == 'tiled':               # (the oracle steps row-major grids only: the load is what this case checks)
        return
    acts = np.random.RandomState(1).uniform(-1, 1, (12, len(order)))
    for t in range(12):
        a.step(acts[t])
        b.step(acts[t])
    for name in fields + ('flags', 'hit', 'kf', 'active'):
        assert torch.equal(a.state.t[name], b.state.t[name]), name
