"""--gaze_method plugin surface (reference: yaw_planner.py; registry experiment.py:12-19).

A policy is built from `params` and maps the env's `info` dict to a yaw-rate action in [-1, 1].  The
reference drives its policies "class as instance" (`policy.__init__(policy, params)`,
`policy.plan(policy, info)`, experiment.py:33-34,69); the classes here work both that way and as ordinary
instances because every helper is a module-level function that takes the state holder explicitly.

They consume only what the reference's policies consume from `info`: info['drone'] (.x .y .yaw
.velocity .yaw_range .yaw_depth .map .trackers), info['trajectory'].positions, info['target'].
These run on the host between device steps; they are consumers of the hot path, not part of it.
"""
import math

import numpy as np
from numpy.linalg import norm


def _clip_rate(err_deg, dt, vmax):
    return max(min(err_deg / dt, vmax), -vmax)


def _turn_towards(target_yaw, yaw, dt, vmax):
    """yaw_planner.py:33-39 / 249-253: saturated rate towards target_yaw, reversed across the wrap."""
    rate = _clip_rate(target_yaw - yaw, dt, vmax)
    return (rate if abs(target_yaw - yaw) < 180 else -rate) / vmax


class NoControl:
    """yaw_planner.py:10-16."""

    def __init__(self, params):
        self.params = params

    def plan(self, state):
        return 0


class Rotating:
    """yaw_planner.py:136-142."""

    def __init__(self, params):
        self.params = params

    def plan(self, observation):
        return 1


class LookAhead:
    """Look along the velocity (yaw_planner.py:18-39)."""

    def __init__(self, params):
        self.dt = params.dt
        self.params = params

    def plan(self, state):
        v = state['drone'].velocity
        if v[1] == 0 and v[0] == 0:
            return 0
        target_yaw = math.degrees(math.atan2(-v[1], v[0])) % 360
        return _turn_towards(target_yaw, state['drone'].yaw, self.dt, self.params.drone_max_yaw_speed)


class LookGoal:
    """Look at the first unexplored trajectory point, else the trajectory end (yaw_planner.py:225-254).
    The reference's plan() falls off the end without a return statement, i.e. always yields None; the
    computed rate is returned here only when `strict_reference` is False."""
    strict_reference = True

    def __init__(self, params):
        self.params = params

    def plan(self, observation):
        traj, drone = observation['trajectory'], observation['drone']
        if len(traj) == 0:
            return 0
        x_look, y_look = traj.positions[-1][0], traj.positions[-1][1]
        for p in traj.positions:
            if drone.map.get_grid(p[0], p[1]) == 0:
                x_look, y_look = p[0], p[1]
                break
        target_yaw = math.degrees(math.atan2(-(y_look - drone.y), x_look - drone.x)) % 360
        rate = _turn_towards(target_yaw, drone.yaw, self.params.dt, self.params.drone_max_yaw_speed)
        return None if self.strict_reference else rate


# ---------------------------------------------------------------------------------------------------
# Oxford (yaw_planner.py:41-127)
# ---------------------------------------------------------------------------------------------------
def view_cone_map(params, dim, x0, y0, yaw, yaw_range, yaw_depth):
    """1 where a cell origin lies inside the view cone (yaw_planner.py:67-79)."""
    s = params.map_scale
    x = np.arange(int(dim[0] // s)).reshape(-1, 1) * s
    y = np.arange(int(dim[1] // s)).reshape(1, -1) * s
    c, sn = math.cos(math.radians(yaw)), -math.sin(math.radians(yaw))
    half = math.radians(yaw_range / 2)
    with np.errstate(divide='ignore', invalid='ignore'):
        d2 = (x0 - x) ** 2 + (y0 - y) ** 2
        ang = np.arccos(((x - x0) * c + (y - y0) * sn) / np.sqrt(d2))
        return np.where(np.logical_or(d2 <= 0, np.logical_and(ang <= half, d2 <= yaw_depth ** 2)), 1, 0)


def _oxford_init(self, params):
    self.params = params
    shape = (params.map_size[0] // params.map_scale, params.map_size[1] // params.map_scale)
    self.last_time_observed_map = 5 * np.ones(shape)
    self.swep_map = np.zeros(shape)
    self.dim = params.map_size
    self.dt = params.dt
    self.tau_s, self.tau_c = 3, 0.5
    self.c1, self.c2, self.c3 = 1000000, 1000, 1
    v = params.drone_max_yaw_speed
    self.v_yaw_space = np.arange(-v, v, v / 3)


def _oxford_plan(self, observation):
    drone, traj = observation['drone'], observation['trajectory']
    p, s = self.params, self.params.map_scale
    self.swep_map = np.zeros_like(self.swep_map)
    for i, pos in enumerate(traj.positions):
        self.swep_map[int(pos[0] // s), int(pos[1] // s)] = i * self.dt
    seen = view_cone_map(p, self.dim, drone.x, drone.y, drone.yaw, drone.yaw_range, drone.yaw_depth)
    self.last_time_observed_map = np.where(seen, 0, self.last_time_observed_map + (1 - seen) * self.dt)
    t_obs, sw = self.last_time_observed_map, self.swep_map
    stale = t_obs >= self.tau_c
    reward = np.where((sw > 0) & (sw <= self.tau_s) & stale, self.c1,
                      np.where((sw > self.tau_s) & stale, self.c2, np.clip(self.c3 * t_obs, -np.inf, 1)))
    if len(traj) == 0:
        return 0
    best, best_reward = 0, 0
    hx, hy = traj.positions[0][0], traj.positions[0][1]
    for i, yaw in enumerate(drone.yaw + self.v_yaw_space * self.dt):
        cone = view_cone_map(p, self.dim, hx, hy, yaw % 360, p.drone_view_range, p.drone_view_depth)
        r = np.sum(cone * reward)
        if best_reward < r:
            best, best_reward = i, r
    return self.v_yaw_space[best] / p.drone_max_yaw_speed


class Oxford:
    """Time-since-observed + swept-trajectory reward, 6 yaw-rate candidates (yaw_planner.py:41-127)."""
    __init__ = _oxford_init
    plan = _oxford_plan


# ---------------------------------------------------------------------------------------------------
# Owl (yaw_planner.py:151-222)
# ---------------------------------------------------------------------------------------------------
def angle_between(a, b):
    d = abs(a % 360 - b % 360)
    return np.minimum(d, 360 - d)


def _owl_G(self, theta):
    if angle_between(theta, 0) <= self.theta_h / 2:
        return 0
    return math.radians(angle_between(theta, self.theta_h / 2)) * math.radians(angle_between(theta, -self.theta_h / 2))


def _owl_U(self, theta):
    return self.U_list[np.argmin(angle_between(np.arange(0, 360, 10), theta))]


def _owl_update_U(self, drone, dt):
    dp = drone.velocity * dt
    for i, d_i in enumerate(np.arange(0, 360, 10)):
        d_hat = np.array([math.cos(math.radians(d_i)), math.sin(math.radians(d_i))])
        L = -dp.dot(d_hat) / self.params.drone_view_depth
        L += self.l_hit if angle_between(d_i, -drone.yaw) < self.theta_h / 2 else self.l_miss
        self.U_list[i] = max(min(self.U_list[i] + L, 1), 0)


def _owl_init(self, params):
    self.params = params
    self.dt = 0.8
    self.u = []
    self.lamb = np.array([0.2, 0.9, 1, 0.1, 0])
    v = params.drone_max_yaw_speed
    self.u_space = np.arange(-v, v, v / 10)
    self.theta_h = params.drone_view_range
    self.l_hit, self.l_miss, self.beta = 0.4, -0.05, 1
    self.U_list = np.zeros(36)


def _owl_plan(self, observation):
    if len(self.u) != 0:
        u = self.u[-1]
        self.u.pop()
        return u / self.params.drone_max_yaw_speed
    drone, target = observation['drone'], observation['target']
    trackers = drone.trackers
    _owl_update_U(self, drone, self.dt)
    with np.errstate(divide='ignore', invalid='ignore'):
        d_g = math.degrees(math.atan2(target[1] - drone.y, target[0] - drone.x))
        d_v = math.degrees(math.atan2(*((drone.velocity / norm(drone.velocity))[::-1])))
        here = np.array([drone.x, drone.y])
        d_o = [math.degrees(math.atan2(*((t.mu_upds[-1][:2, 0] - here)[::-1]))) for t in trackers if t.active is True]
        yaws = -(drone.yaw + self.u_space * self.dt)
        costs = np.ones_like(yaws)
        f = np.zeros([yaws.shape[0], 5])
        for i, yaw in enumerate(yaws):
            f[i, 0] = _owl_G(self, yaw - d_g) * (1 - _owl_U(self, d_g))
            f[i, 1] = norm(drone.velocity / 10) ** 2 * _owl_G(self, yaw - d_v) * (1 - _owl_U(self, d_v))
            for d_o_i, t in zip(d_o, trackers):     # pairs the k-th ACTIVE bearing with the k-th slot, as the reference does
                f[i, 2] += self.beta * norm(t.mu_upds[-1][2:, 0]) / norm(t.mu_upds[-1][:2, 0] - here) * _owl_G(self, yaw - d_o_i)
            f[i, 3] = _owl_U(self, yaw)
            f[i, 4] = abs(math.radians(self.u_space[i] * self.dt))
            costs[i] = np.sum(f[i, :].dot(self.lamb))
    idx = np.argmin(costs)
    for _ in range(int(self.dt // self.params.dt) - 1):
        self.u.append(self.u_space[idx])
    return self.u_space[idx] / self.params.drone_max_yaw_speed


class Owl:
    __init__ = _owl_init
    plan = _owl_plan


policy_list = {'LookAhead': LookAhead, 'NoControl': NoControl, 'Oxford': Oxford, 'Rotating': Rotating,
               'Owl': Owl, 'LookGoal': LookGoal}


def register_policy(name, cls):
    policy_list[name] = cls
