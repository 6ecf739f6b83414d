"""Randomised closed-loop parity (run with -m gpu): seeded random configurations -- agent count / radius / speed, drone
speed (other primitive sets), pillars, static maps, view cone -- device plugins vs the oracle, every field of the env
and plugin state bit for bit, with auto reset and with freeze."""
import os

import numpy as np
import pytest

from test_gpu_plugins import _assert_same, _pair

pytestmark = pytest.mark.gpu


def _random_cfg(rng):
    kw = dict(agent_number=int(rng.randint(0, 31)), agent_radius=int(rng.choice([-1, 5, 8, 10, 12, 15, 18])),
              agent_max_speed=int(rng.choice([4, 10, 20, 30, 40, 60])), drone_max_speed=int(rng.choice([20, 30, 40, 40, 50])),
              map_id=int(rng.randint(0, 10000)), pillar_number=int(rng.choice([0, 0, 3, 6, 9])),
              drone_view_range=int(rng.choice([60, 90, 90, 120, 360])), drone_view_depth=int(rng.choice([60, 80, 80, 100])))
    if rng.rand() < 0.25:
        kw['static_map'] = str(rng.choice(['maps/obstacle_map.npy', 'maps/shaped_obstacle_map.npy']))
    if rng.rand() < 0.3:
        kw['target_list'] = [[int(rng.randint(40, 460)), int(rng.randint(40, 460))], [int(rng.randint(40, 460)), int(rng.randint(40, 460))]]
    if rng.rand() < 0.3:
        kw['init_pos'] = [int(rng.randint(40, 460)), int(rng.randint(40, 460))]
    if rng.rand() < 0.2:
        kw['max_flight_time'] = 6          # freezing ends episodes early
    return kw


def _family(seed):
    """Which family of configurations a seed belongs to: 0-1 the base family (Oxford + Primitive on the default map), 2 wide (map size,
    drone radius / acceleration / yaw rate), 3 short views, 4 the other plugin combinations, 5 map scale 20 with many agents, 6 the
    default geometry with 41-172 agents (the any-N specialisation: SPEC 3).  Seeds below 70000 keep their historical ranges of 10000;
    a soak (D2D_RANDOM_BASE >= 100000) walks through all seven, 500 seeds each."""
    return min(seed // 10000, 6) if seed < 70000 else (seed // 500) % 7


def _many_cfg(seed, kw):
    """Family 6: the default geometry with more agents than the whole-grid kernels take (41 and up; random_map_0 adds 122)."""
    r2 = np.random.RandomState(700000 + seed)
    kw = dict(kw)
    kw.pop('pillar_number', None)
    kw['agent_radius'] = int(r2.choice([5, 6, 8]))
    kw['agent_number'] = int(r2.choice([41, 48, 64, 65, 80, 100]))
    if r2.rand() < 0.3:
        kw['static_map'] = 'maps/random_map_0.npy'
        kw['agent_number'] = int(r2.choice([10, 30, 50]))
        kw['init_pos'] = [250, 30]
    else:
        kw.pop('static_map', None)
    kw['drone_view_range'] = int(r2.choice([90, 90, 120]))
    kw['drone_view_depth'] = 80
    return kw


def _wide_cfg(seed, kw):
    """Seeds from 20000 on also vary what the first family keeps at its default: the map size (non-square, and large
    enough that the search probes the explored map in HBM instead of its LDS copy), the drone's radius, acceleration
    limit (other primitive sets) and yaw rate."""
    r2 = np.random.RandomState(500000 + seed)
    kw = dict(kw)
    kw.pop('static_map', None)                       # the label maps are 500 x 500
    if r2.rand() < 0.6:
        kw['map_size'] = [int(v) for v in r2.choice([500, 600, 700, 800, 1000], 2)]
    kw['drone_radius'] = int(r2.choice([5, 10, 10, 15]))
    kw['drone_max_acceleration'] = int(r2.choice([20, 40, 40, 60]))
    kw['drone_max_yaw_speed'] = int(r2.choice([40, 80, 80, 120]))
    if _family(seed) == 5:                           # map scale 20 (the reference mixes map_scale and a literal 10), many agents
        kw['map_scale'] = 20
        if r2.rand() < 0.3:
            kw['agent_number'] = int(r2.choice([40, 64, 65, 100]))
            kw['agent_radius'] = int(r2.choice([5, 8]))
            w, h = kw.get('map_size', [500, 500])
            while kw['agent_number'] * (2 * (kw['agent_radius'] + 2)) ** 2 > 0.2 * (w - 40) * (h - 40):
                kw['agent_number'] //= 2
    if _family(seed) >= 3:                           # short views: local map edge 4 * (depth // 10) + 1 < 32
        kw['drone_view_depth'] = int(r2.choice([30, 40, 50, 60]))
    return kw


N_SEEDS = int(os.environ.get('D2D_RANDOM_SEEDS', '48'))      # a soak run sets this higher
SEED_BASE = int(os.environ.get('D2D_RANDOM_BASE', '0'))      # ... and moves on to fresh configurations


# seeds that exposed defects in earlier soaks stay in the default run (1892, 2563: a search whose start node is the goal)
REGRESSION_SEEDS = [1892, 2563, 20003, 30011, 40002, 50001, 60001, 60004, 60007]


@pytest.mark.parametrize('seed', list(range(SEED_BASE, SEED_BASE + N_SEEDS)) + (REGRESSION_SEEDS if SEED_BASE == 0 else []))
def test_random_closed_loop_matches_oracle(pkg, hip, oracle, seed):
    rng = np.random.RandomState(1000 + seed)
    kw = _random_cfg(rng)
    fam = _family(seed)
    if fam == 6:
        kw = _many_cfg(seed, kw)
    elif fam >= 2:
        kw = _wide_cfg(seed, kw)
    B, T, chunk = int(rng.choice([3, 5, 8])), 160, int(rng.choice([5, 9, 16]))
    if fam == 6:
        B, T = 3, 60           # (the oracle's 172-agent steps are what the run waits for)
    try:
        if fam in (4, 5):      # the other plugin combinations of the device path (the persistent kernel's non-split loop,
            #                    the planner under a constant gaze action)
            from drone2d_amd import vec_env
            from test_gpu_vs_oracle import _worlds
            kw = _wide_cfg(seed, kw)
            planner, gaze = [('NoMove', 'Oxford'), ('Primitive', 'Rotating'), ('Primitive', 'NoControl'), ('NoMove', 'Rotating')][seed % 4]
            p = pkg.Params(planner=planner, gaze_method=gaze, **kw)
            ref = vec_env.VecDrone2DEnv(p, B, backend=oracle, planner=planner, device_plugins=True, gaze=gaze)
            dev = vec_env.VecDrone2DEnv(p, B, backend=hip, planner=planner, device_plugins=True, gaze=gaze, worlds=_worlds(ref))
        else:
            dev, ref = _pair(pkg, hip, oracle, B, **kw)
    except NotImplementedError as ex:       # e.g. a view range this host's arccos window cannot describe
        pytest.skip(str(ex))
    mode = dict(auto_reset=True) if seed % 3 else dict(freeze_done=True)
    oracle.lib.d2d_oracle_set_threads(8)
    try:
        for t in range(0, T, chunk):
            dev.closed_loop(chunk, **mode)
            ref.closed_loop(chunk, **mode)
            _assert_same(dev, ref, f'seed {seed} {kw} after step {t + chunk}')
    finally:
        oracle.lib.d2d_oracle_set_threads(1)
