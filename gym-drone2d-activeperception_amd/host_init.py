"""Host-side world construction: what the reference's Drone2DEnv2.__init__ does before the first step.

Follows envs/drone_v2.py:79-112 (seeding, Drone2D start state, `init_obstacles_random_size` :13-66) and
utils.py:508-525 (`OccupancyGridMap.init_obstacles`) so that env `i` built with `map_id = s` starts
bit-identical to the reference built with the same Params: the same `random` / `np.random` (MT19937)
streams are consumed in the same order.  Runs once per env on the host; the result stays resident on the
GPU and doubles as the reset() snapshot (envs/drone_v2.py:259-261 re-runs __init__ from the same seed).

tests/test_host_init.py checks this against tests/golden/init_cases.npz captured from the reference.
"""
import json
import math
import os
import random as _random

import numpy as np
from numpy import array, cos, pi, sin
from numpy.linalg import norm

from . import _abi as A

_MAPS_JSON = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'maps', 'static_maps.json')
_map_cache = {}


def load_static_map(path):
    """Label grid of a static map.  The reference np.load()s `params.static_map` relative to its cwd
    (envs/drone_v2.py:49); here the four shipped maps are package data stored as sparse (x, y, label)
    triplets (maps/static_maps.json) and any other path is loaded as a .npy file."""
    key = os.path.splitext(os.path.basename(str(path)))[0]
    if key in _map_cache:
        return _map_cache[key]
    if os.path.isfile(str(path)):
        grid = np.load(path)
    else:
        with open(_MAPS_JSON) as f:
            db = json.load(f)
        if key not in db:
            raise FileNotFoundError(f'static map {path!r} not found (known: {sorted(db)})')
        grid = np.zeros(db[key]['shape'], dtype=np.int64)
        for x, y, lab in db[key]['xyl']:
            grid[x, y] = lab
    _map_cache[key] = grid
    return grid


def _target_xy(t):
    """planner.set_target: `self.target[:2] = target` broadcasts a scalar (traj_planner.py:24-26)."""
    t = np.asarray(t, dtype=np.float64).ravel()
    return (float(t[0]), float(t[0])) if t.size == 1 else (float(t[0]), float(t[1]))


def init_world(params):
    """Build one env's initial state.  Returns a dict of numpy arrays (fp64 / int32 / uint8)."""
    p = params
    rnd = _random.Random(p.map_id)               # random.seed(map_id)      drone_v2.py:79
    nrs = np.random.RandomState(p.map_id)        # np.random.seed(map_id)   drone_v2.py:80
    W_px, H_px = p.map_size[0], p.map_size[1]
    scale = p.map_scale
    W, H = W_px // scale, H_px // scale
    dx, dy = p.init_position[0], p.init_position[1]
    targets = [_target_xy(t) for t in p.target_list]
    drone_xy = np.array([dx, dy])

    # pillars, drone_v2.py:14-26
    obstacles = []
    while len(obstacles) < p.pillar_number:
        obs = np.array([rnd.randint(50, W_px - 50), rnd.randint(50, H_px - 50), rnd.randint(15, 20)])
        free = True
        for t in p.target_list:
            if norm(t - obs[:-1]) <= p.drone_radius + 20 + obs[-1]:
                free = False
                break
        if norm(drone_xy - obs[:-1]) <= p.drone_radius + 70:
            free = False
        if free:
            obstacles.append(obs)

    # random agents by rejection sampling, drone_v2.py:28-47
    pos, rad, pref, group = [], [], [], []
    n_rand = p.agent_number
    while len(pos) < n_rand:
        x = rnd.uniform(20, W_px - 20)
        y = rnd.uniform(20, H_px - 20)
        r = rnd.uniform(5, 15) if p.agent_radius == -1 else rnd.uniform(p.agent_radius - 2, p.agent_radius + 2)
        k = len(pos)
        pv = -p.agent_max_speed * array([cos(2 * pi * k / n_rand), sin(2 * pi * k / n_rand)])
        new_pos = np.array((x, y))
        free = True
        for q, rq in zip(pos, rad):
            if norm(q - new_pos) <= rq + r:
                free = False
        for ob in obstacles:
            if norm(np.array([ob[0], ob[1]]) - new_pos) <= ob[2] + r + 10:
                free = False
        if norm(new_pos - drone_xy) <= p.drone_radius + 70:
            free = False
        if free:
            pos.append(new_pos)
            rad.append(r)
            pref.append(pv)
            group.append(0)

    # one radius-5 agent per nonzero static-map cell, drone_v2.py:49-66
    label = load_static_map(p.static_map)
    vels = []
    for _ in range(100):
        direction = nrs.rand() * 2 * np.pi
        vels.append([p.agent_max_speed * np.cos(direction), p.agent_max_speed * np.sin(direction)])
    for x in range(label.shape[0]):
        for y in range(label.shape[1]):
            if label[x][y] != 0:
                pos.append(np.array([5 + x * 10, 5 + y * 10], dtype=np.float64))
                rad.append(5)
                pref.append(np.array(vels[label[x][y]], dtype=np.float64))
                group.append(int(label[x][y]))

    N = len(pos)
    agents = np.zeros((A.AF, N), dtype=np.float64)
    unit = np.zeros(N, dtype=np.int32)
    for k in range(N):
        agents[A.A_PX, k], agents[A.A_PY, k] = pos[k][0], pos[k][1]
        agents[A.A_VX, k], agents[A.A_VY, k] = pref[k][0], pref[k][1]
        agents[A.A_R, k] = rad[k]
        agents[A.A_R2, k] = rad[k] ** 2          # Python's own `agent.radius**2` (utils.py:659)
        unit[k] = int(rad[k] // scale)           # utils.py:533

    # ground-truth grid, utils.py:495-525
    gt = np.full((W, H), A.UNOCCUPIED, dtype=np.uint8)
    gt[0, :] = gt[-1, :] = A.OCCUPIED
    gt[:, 0] = gt[:, -1] = A.OCCUPIED
    cx = scale * (np.arange(W) + 0.5)            # get_real_pos, utils.py:542
    cy = scale * (np.arange(H) + 0.5)
    for ob in obstacles:
        d = np.sqrt((cx[:, None] - ob[0]) ** 2 + (cy[None, :] - ob[1]) ** 2)
        gt[d <= ob[2]] = A.OCCUPIED
    dyn_prev = np.zeros((N, 3), dtype=np.int32)
    for k in range(N):
        ax, ay, r = agents[A.A_PX, k], agents[A.A_PY, k], rad[k]
        ci, cj = int(ax // scale), int(ay // scale)
        u0 = int(r // scale) + 2                 # block that contains every cell centre within r
        i0, i1 = max(ci - u0, 0), min(ci + u0 + 1, W)
        j0, j1 = max(cj - u0, 0), min(cj + u0 + 1, H)
        inside = (cx[i0:i1, None] - ax) ** 2 + (cy[None, j0:j1] - ay) ** 2 <= r ** 2
        gt[i0:i1, j0:j1][inside] = A.DYNAMIC     # also over walls, utils.py:521-525
        dyn_prev[k] = (ci, cj, u0)

    T = max(len(targets), 1)
    tl = np.zeros((T, 2), dtype=np.float64)
    for i, t in enumerate(targets):
        tl[i] = t
    drone = np.zeros(A.DF, dtype=np.float64)
    drone[A.D_X], drone[A.D_Y], drone[A.D_YAW] = dx, dy, (-90) % 360     # utils.py:718
    counters = np.zeros(A.CF, dtype=np.int32)
    counters[A.C_SM] = A.SM_WAIT_FOR_GOAL                                 # drone_v2.py:116
    counters[A.C_NTGT] = len(targets)
    tracker_radius = np.full(N, float(p.agent_radius), dtype=np.float64)  # utils.py:184
    tracker_radius[:n_rand] = rad[:n_rand]                                # drone_v2.py:46
    return dict(agents=agents, agent_unit=unit, dyn_prev=dyn_prev, gt=gt,
                dmap=np.zeros((W, H), dtype=np.uint8),                    # utils.py:722 init_num=0
                drone=drone, target=np.array([dx, dy], dtype=np.float64),  # traj_planner.py:22
                targets=tl, counters=counters, group=np.array(group, dtype=np.int64),
                obstacles=np.array(obstacles, dtype=np.int64).reshape(-1, 3),
                tracker_radius=tracker_radius, N=N, W=W, H=H, T=T)


def derive_cfg(params, B, N, T=1, planner_mode=A.PLANNER_EXTERNAL, kf_enabled=True, grid_tile=0):
    """Numeric constants of one batch (include/d2d.h d2d_cfg), evaluated exactly as the reference's
    Python evaluates them."""
    p = params
    c = A.Cfg()
    c.abi_version = A.D2D_ABI_VERSION
    c.B, c.N, c.T = B, N, T
    c.W, c.H = p.map_size[0] // p.map_scale, p.map_size[1] // p.map_scale
    c.R = math.ceil(p.map_size[0] / 10)                       # strip_width = 10, utils.py:570,587
    c.L = 4 * (p.drone_view_depth // p.map_scale) + 1         # drone_v2.py:133
    c.planner_mode = planner_mode
    c.kf_enabled = 1 if kf_enabled else 0
    c.grid_tile = int(grid_tile)
    fov = math.radians(p.drone_view_range)                    # utils.py:575
    c.dt, c.scale = p.dt, p.map_scale
    c.W_px, c.H_px = p.map_size[0], p.map_size[1]
    c.ray_off0 = -fov / 2                                     # utils.py:594
    c.ray_dth = fov / c.R
    c.depth = p.drone_view_depth
    c.drone_radius = p.drone_radius
    c.yaw_rate = p.drone_max_yaw_speed
    c.max_acc = p.drone_max_acceleration
    c.max_steps = p.max_flight_time / p.dt                    # drone_v2.py:89
    c.sigma = p.var_cam
    c.kf_lo_x = 10 + p.agent_radius                           # utils.py:236-237
    c.kf_hi_x = p.map_size[0] - 10 - p.agent_radius
    c.kf_lo_y = 10 + p.agent_radius
    c.kf_hi_y = p.map_size[1] - 10 - p.agent_radius
    return c
