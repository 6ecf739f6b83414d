"""Device planner / gaze plugins, CPU side: the host tables are what numpy computes, the oracle's restatement of
Primitive (traj_planner.py:78-233) and Oxford (yaw_planner.py:41-127) reproduces every closed-loop episode
captured from the imported reference -- gaze action, plan result, head waypoint, trajectory length and the
full env state of every step."""
import math

import numpy as np
import pytest
import torch

import replay


def test_numpy_roundings_the_restatement_relies_on():
    """The formulas oracle/d2d_oracle.c and csrc/d2d_plugins.h use for numpy's small matmuls / norm, measured
    against this host's numpy (they depend on its BLAS: skip loudly, not silently, if a host differs)."""
    rng = np.random.RandomState(2)
    fma = getattr(math, 'fma', None)
    if fma is None:
        import ctypes as C
        libm = C.CDLL('libm.so.6')
        libm.fma.restype = C.c_double
        libm.fma.argtypes = [C.c_double] * 3
        fma = libm.fma
    U = np.arange(-40, 40, 0.4 * 40 - 5)
    ts = np.concatenate([np.arange(2, 0, -0.1), np.arange(0, 2, 0.25)])
    for _ in range(4000):
        px, py = float(rng.randint(0, 500)), float(rng.randint(0, 500))
        vx, vy = rng.uniform(-40, 40, 2) if rng.rand() < 0.5 else [float(rng.randint(-40, 40)) + rng.choice([0, 0.5, 0.25, 0.1])] * 2
        ax, ay = rng.choice(U, 2)
        t = float(rng.choice(ts))
        coeff = np.array([[px, vx, ax / 2], [py, vy, ay / 2]])
        pos = np.around(np.array([1, t, t ** 2]) @ coeff.T)                       # traj_planner.py:121,176
        vel = np.array([1, 2 * t]) @ coeff[:, 1:].T                               # :122
        pend = np.around(np.array([1, 2, 2 ** 2]) @ np.array([[px, py], [vx, vy], [ax / 2, ay / 2]]))   # :182
        vend = np.array([1, 2 * 2]) @ np.array([[vx, vy], [ax / 2, ay / 2]])      # :172
        t2 = t ** 2
        assert list(pos) == [np.rint(fma(t2, ax / 2, px + t * vx)), np.rint(fma(t2, ay / 2, py + t * vy))]
        assert list(vel) == [vx + (2 * t) * (ax / 2), vy + (2 * t) * (ay / 2)]
        assert list(pend) == [np.rint((px + 2 * vx) + 4 * (ax / 2)), np.rint((py + 2 * vy) + 4 * (ay / 2))]
        assert list(vend) == [vx + 4 * (ax / 2), vy + 4 * (ay / 2)]
        a, b = rng.uniform(-50, 50, 2)
        assert float(np.linalg.norm(np.array([a, b]))) == math.sqrt(fma(b, b, a * a))


def test_pairwise_plan_is_numpy_sum(pkg):
    from drone2d_amd import device_plugins as DP
    rng = np.random.RandomState(3)
    for W, H in ((50, 50), (64, 37), (100, 80), (9, 9), (90, 90)):
        leaves, prog = DP.pairwise_plan(W * H)
        assert len(prog) == 2 * len(leaves) - 1 and all(l[0] % 8 == 0 for l in leaves)
        assert leaves[0][0] == 0 and all(leaves[k][0] + leaves[k][1] == leaves[k + 1][0] for k in range(len(leaves) - 1))
        ops, start, root = DP.pairwise_levels(len(leaves), prog)
        assert len(ops) == len(leaves) - 1 and start[-1] == len(ops) and root == (2 * len(leaves) - 2 if len(leaves) > 1 else 0)
        done = set(range(len(leaves)))
        for lv in range(len(start) - 1):                       # every level only reads what earlier levels produced
            assert all(a in done and b in done for _, a, b in ops[start[lv]:start[lv + 1]])
            done |= {int(d) for d, _, _ in ops[start[lv]:start[lv + 1]]}
        for _ in range(20):
            v = rng.choice([0, 1], size=(W, H))
            r = np.where(rng.rand(W, H) < 0.1, 1e6, np.where(rng.rand(W, H) < 0.1, 1000.0, np.clip(rng.randint(0, 60, (W, H)) * 0.1, -np.inf, 1)))
            a = v * r
            assert np.sum(a) == DP.pairwise_sum_host(list(a.ravel()), leaves, prog)


def test_perfect_addition_trees_are_butterflies(pkg):
    """What the gaze stage's butterfly over lanes (csrc/d2d_plugins.h: a tree of height h with 2^h blocks, h <= 5) relies on, for
    every map size of the dense path: pw_ntree - 3 pw_nleaf is the tree's height; such a tree adds neighbouring nodes in cell
    order at every level (node 2k + node 2k + 1, left operand first), which is what `v + row_shl(v)` per level computes; and a
    butterfly over zero-padded lanes gives numpy's sum bit for bit."""
    from drone2d_amd import device_plugins as DP
    rng = np.random.RandomState(11)
    seen_perfect = seen_other = 0
    for n in list(range(1, 4097, 7)) + [2500, 3600, 4096, 1200, 2080, 260]:
        leaves, prog = DP.pairwise_plan(n)
        ops, start, root = DP.pairwise_levels(len(leaves), prog)
        nlev, nleaf = len(start) - 1, len(leaves)
        ntree = 2 + len(start) + 3 * len(ops)                   # [n_levels, root, level_start[n_levels + 1], 3 ints per addition]
        assert ntree - 3 * nleaf == nlev
        if nleaf != (1 << nlev) or nlev > 5:
            seen_other += 1
            continue
        seen_perfect += 1
        prev = list(range(nleaf))                               # node ids of the level below, in cell order
        for lv in range(nlev):
            level = [tuple(int(x) for x in o) for o in ops[start[lv]:start[lv + 1]]]
            assert len(level) == len(prev) // 2
            assert [(a, b) for _, a, b in level] == [(prev[2 * k], prev[2 * k + 1]) for k in range(len(level))]
            prev = [d for d, _, _ in level]
        assert prev == [root]
        # the butterfly itself on 32 zero-padded lanes against numpy
        a = np.where(rng.rand(n) < 0.2, rng.choice([1e6, 1000.0, 0.3, 0.7, 1.0], size=n), 0.0)
        lane = [0.0] * 32
        for k, (off, m) in enumerate(leaves):
            lane[k] = float(np.sum(a[off:off + m]))             # a block: numpy's own 8-accumulator sum
        for sh in (1, 2, 4, 8, 16):
            lane = [lane[i] + lane[i + sh] if i + sh < 32 else lane[i] for i in range(32)]
        assert lane[0] == float(np.sum(a))
    assert seen_perfect > 100 and seen_other > 50


def test_acos_window_is_this_hosts_arccos(pkg):
    from drone2d_amd import device_plugins as DP
    rng = np.random.RandomState(4)
    for deg in (90, 120, 60, 360, 200, 30):
        half = math.radians(deg / 2)
        k0, mask = DP.acos_window(half)
        q = np.concatenate([rng.uniform(-1, 1, 5000), math.cos(min(half, math.pi)) + rng.uniform(-1e-14, 1e-14, 5000)])
        q = q[np.abs(q) <= 1]
        want = DP._arccos_le(q, half)
        keys = np.array([DP._key(v) for v in q])
        got = np.where(keys >= k0 + 64, True, False)
        ins = (keys >= k0) & (keys < k0 + 64)
        got[ins] = [(mask >> int(k - k0)) & 1 == 1 for k in keys[ins]]
        assert np.array_equal(got, want), deg
    with pytest.raises(NotImplementedError):
        DP.acos_window(math.radians(90.0))      # a 180-degree view: the edge sits where doubles are far denser than arccos


def test_tables_are_the_references_expressions(pkg):
    from drone2d_amd import device_plugins as DP, host_init
    p = pkg.Params(planner='Primitive', gaze_method='Oxford')
    cfg = host_init.derive_cfg(p, B=1, N=10)
    sc, tb = DP.build_tables(p, cfg)
    assert list(tb['u_space']) == [-40, -29, -18, -7, 4, 15, 26, 37] and sc['n_sample'] == 8 and sc['n_ts'] == 20
    assert tb['traj_t'][0, 0] == np.arange(2, 0, -0.1)[-1] and tb['traj_t'][-1, 0] == 2.0
    assert np.array_equal(tb['yaw_space'], np.arange(-80, 80, 80 / 3))
    t0 = 0.0
    for k in range(10):
        assert tb['tobs_tab'][0, k] == t0
        t0 = t0 + 0.1
    assert tb['tobs_tab'][1, 0] == 5.0 and tb['tobs_tab'][1, 3] == ((5.0 + 0.1) + 0.1) + 0.1
    assert sc['hash_cap'] > sc['node_cap'] and sc['hash_cap'] & (sc['hash_cap'] - 1) == 0


def _closed_loop(pkg, backend, name):
    from drone2d_amd import device_plugins as DP
    gaze = 'Oxford' if 'oxford' in name else 'external'
    R = replay.Replay(pkg, backend, name, kf=True)
    ps = DP.PluginState(R.p, R.cfg, backend.device, R.world['tracker_radius'][None], planner='Primitive', gaze=gaze)
    plan = ps.struct()
    fx = R.fx
    for t in range(R.T):
        s = R.st.struct()
        tag = f'{name} step {t + 1}: '
        if gaze == 'Oxford':
            backend.gaze_stage(R.cfg, s, plan)
            backend.sync()
            assert float(R.st.action[0]) == float(fx['t_action'][t]), tag + 'gaze action'
        else:
            R.st.action.fill_(float(fx['t_action'][t]))
        backend.perceive(R.cfg, s)
        backend.plan_stage(R.cfg, s, plan)
        backend.sync()
        assert int(R.st.plan_ok[0]) == int(fx['t_plan_ok'][t]), tag + 'plan() result'
        assert int(R.st.wp_valid[0]) == int(fx['t_wp_valid'][t]), tag + 'trajectory empty / not'
        if fx['t_wp_valid'][t]:
            assert np.array_equal(R.st.wp[0].cpu().numpy(), fx['t_wp'][t]), tag + 'head waypoint'
        hdr = ps.t['traj_hdr'][0].cpu().numpy()
        assert int(hdr[1] - hdr[0]) == int(fx['t_traj_len'][t]), tag + 'len(trajectory)'
        backend.act(R.cfg, s)
        backend.sync()
        R.compare(t)
    assert int(ps.t['plan_stat'][0, 3]) == 0
    return R, ps


@pytest.mark.parametrize('name', replay.TRACES_CLOSED + replay.TRACES_PLANNED)
def test_oracle_plugins_reproduce_the_reference_episode(pkg, oracle, name):
    _closed_loop(pkg, oracle, name)


def test_oracle_closed_loop_entry_point_and_auto_reset(pkg, oracle):
    """d2d_closed_loop == the four stages called one by one; with auto_reset a finished episode restarts (at its next
    step) from the seeded world with fresh plugin state and replays itself exactly."""
    from drone2d_amd import vec_env
    p = pkg.Params(planner='Primitive', gaze_method='Oxford', agent_number=10, agent_radius=15, agent_max_speed=20,
                   drone_max_speed=40, map_id=1)
    env = vec_env.VecDrone2DEnv(p, 3, backend=oracle, planner='Primitive', device_plugins=True, gaze='Oxford')
    fx = replay.load('readme_oxford_primitive')
    T = len(fx['t_action'])
    yaws = []
    for t in range(T + 40):
        env.closed_loop(1, auto_reset=True)
        k = t if t < T else t - T                 # the step after the terminal one starts the episode over
        assert float(env.state.drone[0, 2]) == float(fx['t_drone'][k][2]), t
        assert np.array_equal(env.state.drone[0, :2].numpy(), fx['t_drone'][k][:2]), t
        assert bool(env.state.flags[0, 3]) == bool(fx['t_done'][k]), t      # the terminal state stays visible
        yaws.append(float(env.state.drone[1, 2]))
    assert int(env.state.counters[0, pkg._abi.C_STEPS]) == 40
    # env 1 is map_id 2: a different world, also deterministic under reset
    env2 = vec_env.VecDrone2DEnv(p, 3, backend=oracle, planner='Primitive', device_plugins=True, gaze='Oxford')
    env2.closed_loop(T + 40, auto_reset=True)
    for name in ('agents', 'gt', 'dmap', 'drone', 'counters', 'kf', 'active', 'flags'):
        assert torch.equal(env.state.t[name], env2.state.t[name]), name
    for name in ('traj_hdr', 'seen_step', 'trk_radius', 'trk_prev'):
        assert torch.equal(env.plugins.t[name], env2.plugins.t[name]), name


def test_constant_gaze_policies_on_the_plugin_path(pkg, oracle):
    """gaze='NoControl' / 'Rotating' with device_plugins: the reference's constant policies (yaw_planner.py:10-16,
    136-142) are a resident action of 0 / 1 -- NoControl + NoMove over 30 closed-loop steps is 30 plain steps at action 0
    (the survivability rollout, cal_difficulty_survivability.py:53-60), Rotating the same at action 1."""
    from drone2d_amd import vec_env
    for gaze, a in (('NoControl', 0.0), ('Rotating', 1.0)):
        p = pkg.Params(planner='NoMove', gaze_method=gaze, agent_number=10, agent_radius=15, agent_max_speed=20, map_id=1)
        env = vec_env.VecDrone2DEnv(p, 3, backend=oracle, planner='NoMove', device_plugins=True, gaze=gaze)
        ref = vec_env.VecDrone2DEnv(p, 3, backend=oracle, planner='NoMove')
        env.closed_loop(30)
        for _ in range(30):
            ref.step(torch.full((3,), a, dtype=torch.float64))
        for name in ('agents', 'gt', 'dmap', 'drone', 'counters', 'kf', 'active', 'flags'):
            assert torch.equal(env.state.t[name], ref.state.t[name]), (gaze, name)
