"""Drone2DEnv2 — the reference's single-env gym surface (`gym-2d-perception-v2`, envs/drone_v2.py:10-305)
on top of the batched device step.

    env = Drone2DEnv2(params)            # or gym.make('gym-2d-perception-v2', params=params) once registered
    env.reset() -> {}                    # drone_v2.py:259-261
    obs, 0, done, info = env.step(a)     # old-gym 4-tuple, drone_v2.py:257

The world lives in HBM (a VecDrone2DEnv of one env); `step` is d2d_step (NoMove), d2d_perceive -> d2d_plan_stage ->
d2d_act (Primitive on the device) or d2d_perceive -> host planner plugin -> d2d_act (any other class with the
reference interface).  The objects scripts reach into are thin proxies over a host mirror refreshed by one packed
copy per step, and every attribute the reference's scripts mutate writes through to the device:
    env.drone.x / .y / .yaw / .velocity / .radius        (validation_shape.py:79-80, survivability sweeps)
    env.agents[i].position / .pref_velocity / .radius     (validation_speed.py:135-138)
    env.drone.map.grid_map, env.map_gt.grid_map, env.drone.trackers[k].active / .mu_upds / .estimate_pos
"""
import numpy as np
import torch

from . import _abi as A
from .params import with_defaults
from .gaze import DeviceAction
from .planners import planner_list
from .vec_env import VecDrone2DEnv

_NP = {torch.float64: np.float64, torch.float32: np.float32, torch.int32: np.int32, torch.uint8: np.uint8}

try:                                   # gym is optional: the reference pins gym 0.21, this image has none
    import gym as _gym
    if not (hasattr(_gym, '__version__') and hasattr(_gym.spaces.Dict, '__getitem__')):
        raise ImportError('not a real gym (an import-time stub of a harness)')
    _EnvBase = _gym.Env
except Exception:                      # pragma: no cover
    _gym = None
    _EnvBase = object


class Box:
    """Minimal stand-in for gym.spaces.Box when gym is not importable."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low, self.high, self.dtype = low, high, dtype
        self.shape = tuple(shape) if shape is not None else np.shape(low)

    def sample(self):
        return np.random.uniform(self.low, self.high).astype(self.dtype)


class DictSpace:
    def __init__(self, spaces):
        self.spaces = spaces

    def __getitem__(self, k):
        return self.spaces[k]


def _spaces():
    if _gym is not None:
        return _gym.spaces.Box, _gym.spaces.Dict
    return Box, DictSpace


class GridProxy:
    """OccupancyGridMap view (utils.py:494-548): .grid_map, .get_grid, .x_scale, .dim, .width, .height."""

    def __init__(self, env, field, scale, dim):
        self._env, self._field = env, field
        self.x_scale = self.y_scale = scale
        self.dim = dim
        self.width, self.height = dim[0] // scale, dim[1] // scale

    @property
    def grid_map(self):
        return self._env._mirror[self._field]

    @grid_map.setter
    def grid_map(self, value):
        arr = np.ascontiguousarray(value, dtype=np.uint8)
        self._env._mirror[self._field] = arr
        self._env._push(self._field, arr)

    def get_grid(self, x, y):
        if x >= self.dim[0] or x < 0 or y >= self.dim[1] or y < 0:
            return 1
        return self.grid_map[int(x // self.x_scale), int(y // self.y_scale)]

    def get_real_pos(self, i, j):
        return np.array([self.x_scale * (i + 0.5), self.y_scale * (j + 0.5)])


class TrackerProxy:
    """KalmanFilter view (utils.py:172-275): .active, .radius, .mu_upds[-1], .Sigma_upds[-1], .ts, .estimate_pos."""

    def __init__(self, env, k):
        self._env, self._k = env, k

    @property
    def active(self):
        return bool(self._env._mirror['active'][self._k])      # a real bool: Owl tests `active is True`

    @property
    def radius(self):
        return float(self._env._tracker_radius[self._k])

    @property
    def mu_upds(self):
        return [self._env._mirror['kf'][self._k, :4].reshape(4, 1).copy()]

    @property
    def Sigma_upds(self):
        return [self._env._mirror['kf'][self._k, 4:].reshape(4, 4).copy()]

    @property
    def ts(self):
        return range(int(self._env._mirror['kf_len'][self._k]))

    def estimate_pos(self, t):
        kf = self._env._mirror['kf'][self._k]
        return kf[:2] + t * kf[2:4]


class _ArchivedTracker:
    """Entry of info['tracker_buffer']: only len(ts) is consumed (experiment.py:74,93-94)."""

    def __init__(self, n):
        self.ts = range(n)
        self.active = False


class AgentProxy:
    """Agent view (utils.py:461-493).  Under CVM velocity IS pref_velocity (drone_v2.py:178)."""

    def __init__(self, env, k):
        self._env, self._k = env, k
        self.group_id = int(env._group[k])
        self.max_speed = env.params.agent_max_speed

    def _get(self, f0, f1):
        ag = self._env._mirror['agents']
        return np.array([ag[f0, self._k], ag[f1, self._k]])

    def _set(self, f0, f1, v):
        v = np.asarray(v, dtype=np.float64).ravel()
        ag = self._env._mirror['agents']
        ag[f0, self._k], ag[f1, self._k] = v[0], v[1]
        self._env._push('agents', ag)

    position = property(lambda s: s._get(A.A_PX, A.A_PY), lambda s, v: s._set(A.A_PX, A.A_PY, v))
    pref_velocity = property(lambda s: s._get(A.A_VX, A.A_VY), lambda s, v: s._set(A.A_VX, A.A_VY, v))
    velocity = pref_velocity

    @property
    def radius(self):
        return float(self._env._mirror['agents'][A.A_R, self._k])

    @radius.setter
    def radius(self, r):
        env, k = self._env, self._k
        ag = env._mirror['agents']
        ag[A.A_R, k] = r
        ag[A.A_R2, k] = r ** 2
        env._push('agents', ag)
        env._mirror['agent_unit'][k] = int(r // env.params.map_scale)
        env._push('agent_unit', env._mirror['agent_unit'])


class DroneProxy:
    """Drone2D view (utils.py:714-784)."""

    def __init__(self, env):
        self._env = env
        p = env.params
        self.yaw_range, self.yaw_depth = p.drone_view_range, p.drone_view_depth
        self.dt, self.params = p.dt, p
        self.map = GridProxy(env, 'dmap', p.map_scale, p.map_size)
        self.trackers = [TrackerProxy(env, k) for k in range(env._vec.N)]
        self.rays = {}

    def _num(self, i):
        v = float(self._env._mirror['drone'][i])
        return int(v) if v == int(v) and i in (A.D_X, A.D_Y) else v    # x, y are Python ints after step_pos

    def _put(self, i, v):
        self._env._mirror['drone'][i] = float(np.asarray(v).ravel()[0])
        self._env._push('drone', self._env._mirror['drone'])

    x = property(lambda s: s._num(A.D_X), lambda s, v: s._put(A.D_X, v))
    y = property(lambda s: s._num(A.D_Y), lambda s, v: s._put(A.D_Y, v))
    yaw = property(lambda s: s._num(A.D_YAW), lambda s, v: s._put(A.D_YAW, v))

    @property
    def radius(self):
        return self._env.params.drone_radius

    @property
    def velocity(self):
        d = self._env._mirror['drone']
        return np.array([d[A.D_VX], d[A.D_VY]])

    @velocity.setter
    def velocity(self, v):
        v = np.asarray(v, dtype=np.float64).ravel()
        d = self._env._mirror['drone']
        d[A.D_VX], d[A.D_VY] = v[0], v[1]
        self._env._push('drone', d)

    @property
    def acceleration(self):
        d = self._env._mirror['drone']
        return np.array([d[A.D_AX], d[A.D_AY]])

    def get_local_map(self):
        return self._env._mirror['obs_local']


class Drone2DEnv2(_EnvBase):
    """Three step paths, chosen by the class `planner_list[params.planner]` resolves to:
      device NoMove     one fused launch (d2d_step)
      device Primitive  d2d_perceive -> d2d_plan_stage -> d2d_act queued back to back, no host in between
      host plugin       d2d_perceive -> planner.replan_check / plan on the host -> d2d_act
    and ONE packed device-to-host copy per step refreshes the mirror the proxies read (two on the host-plugin path,
    whose planner needs this step's perception)."""
    metadata = {'render.modes': []}
    _MIRROR_STATE = ('agents', 'drone', 'target', 'kf', 'agent_unit', 'counters', 'kf_len', 'newly', 'obs_yaw',
                     'gt', 'dmap', 'active', 'hit', 'flags', 'obs_local')          # 8-byte fields first: aligned views
    _MIRROR_PLUGIN = ('trk_radius', 'traj_hdr')

    def __init__(self, params, device='cuda:0', backend=None):
        self._device, self._backend = device, backend
        self._render_warned = False
        self._build(params)

    # ------------------------------------------------------------------------------------------
    def _build(self, params):
        self.params = with_defaults(params)
        p = self.params
        if p.motion_profile != 'CVM':
            raise NotImplementedError('motion_profile RVO is outside the accelerated hot path (SURVEY.md section 2)')
        planner_cls = planner_list[p.planner]                       # KeyError for an unknown name, as the reference
        on_device = bool(getattr(planner_cls, 'on_device', False))
        self._mode = ('fused' if p.planner == 'NoMove' else 'device') if on_device else 'host'
        want_gaze = on_device and p.gaze_method == 'Oxford'
        plugins = self._mode == 'device' or want_gaze
        self._vec = VecDrone2DEnv(p, 1, device=self._device, backend=self._backend, grid_layout='rowmajor',   # the proxies index [W][H]
                                  planner=p.planner if on_device else 'external', device_plugins=plugins,
                                  gaze=('Oxford' if want_gaze else 'external') if plugins else None)
        self._slot = 0
        self._device_gaze = want_gaze
        self._backend = self._vec.backend
        self._tracker_radius = self._vec.tracker_radius[self._slot].numpy().copy()
        from . import host_init
        self._group = host_init.init_world(p)['group'] if self._vec.N else np.zeros(0, dtype=np.int64)
        # utils.py:605 draws the measurement noise from the global numpy stream the env seeded (drone_v2.py:80) and
        # advanced by the 100 draws of its init (drone_v2.py:51-55): the same stream, kept private to this env
        self._noise_rng = np.random.RandomState(p.map_id)
        self._noise_rng.rand(100)
        self.dt = p.dt
        self.steps = 0
        self.max_steps = p.max_flight_time / p.dt
        self.tracked_agent = 0
        self.tracker_buffer = []
        self.target_list = [list(np.asarray(t).ravel()) for t in p.target_list]
        self.obstacles = []
        self.screen = None                      # scripts assign / read it around render (survivability_calculator.py:38-39)
        self._mirror, self._pull_count = {}, 0
        self._pull()
        self.drone = DroneProxy(self)
        self.agents = [AgentProxy(self, k) for k in range(self._vec.N)]
        self.map_gt = GridProxy(self, 'gt', p.map_scale, p.map_size)
        self.planner = planner_cls(self.drone, p)
        self.state_machine = A.SM_WAIT_FOR_GOAL
        self.fail_count = 0
        BoxT, DictT = _spaces()
        L = self._vec.cfg.L
        self.action_space = BoxT(np.array([-1]), np.array([1]), shape=(1,))
        self.observation_space = DictT({
            'yaw_angle': BoxT(low=np.array([0], dtype=np.float32), high=np.array([360], dtype=np.float32),
                              shape=(1,), dtype=np.float32),
            'local_map': BoxT(low=np.zeros((1, L, L), dtype=np.float32), high=np.float32(4 * np.ones((1, L, L))),
                              shape=(1, L, L), dtype=np.float32),
            'swep_map': BoxT(low=np.zeros((1, L, L), dtype=np.float32), high=np.float32(10 * np.ones((1, L, L))),
                             shape=(1, L, L), dtype=np.float32)})
        self.info = self._info(0, 0, 0)

    def _pull(self, only=None):
        """ONE device-to-host copy: the mirrored fields of this env packed into a byte buffer on the device."""
        vec, slot = self._vec, self._slot
        parts = [(k, vec.state.t[k][slot]) for k in (only or self._MIRROR_STATE)]
        if vec.plugins is not None and only is None:
            parts += [(k, vec.plugins.t[k][slot]) for k in self._MIRROR_PLUGIN]
        flat = torch.cat([t.reshape(-1).view(torch.uint8) for _, t in parts]).cpu().numpy()
        off = 0
        for k, t in parts:
            n = t.numel() * t.element_size()
            self._mirror[k] = np.frombuffer(flat[off:off + n].tobytes(), dtype=_NP[t.dtype]).reshape(tuple(t.shape)).copy()
            off += n
        if 'trk_radius' in self._mirror:
            self._tracker_radius = self._mirror['trk_radius']
        self._pull_count += 1

    def _push(self, field, arr):
        self._vec.state.t[field][self._slot].copy_(torch.from_numpy(np.ascontiguousarray(arr)))

    def _push_plugin(self, field, arr):
        self._vec.plugins.t[field][self._slot].copy_(torch.from_numpy(np.ascontiguousarray(arr)))

    def _info(self, col, dead, frz):
        return {'drone': self.drone, 'trajectory': self.planner.trajectory, 'state_machine': self.state_machine,
                'target': self.planner.target, 'collision_flag': col, 'dead_lock_flag': dead, 'freezing_flag': frz,
                'flight_time': self.steps * self.dt, 'tracker_buffer': self.tracker_buffer}

    # ------------------------------------------------------------------------------------------
    def reset(self):
        self._build(self.params)
        return {}

    def _plan_phase(self):
        """Host planner plugin between the two device halves (drone_v2.py:194-197 + utils.py:733-739).  Returns
        (plan_ok, has_waypoint, waypoint[6]) and pushes the planner's target to the device."""
        # set_target as the device's state machine just did (drone_v2.py:160-163)
        self.planner.target = np.array([self._mirror['target'][0], self._mirror['target'][1], 0., 0.])
        self.planner.replan_check(self.drone)                          # drone_v2.py:194
        ok = bool(self.planner.plan(self.drone, self.dt))              # drone_v2.py:197
        tr = self.planner.trajectory
        wp = np.zeros(6)
        has_wp = len(tr) > 0
        if has_wp:
            wp[0:2] = np.asarray(tr.positions[0], dtype=np.float64).ravel()
            wp[2:4] = np.asarray(tr.velocities[0], dtype=np.float64).ravel()
            wp[4:6] = np.asarray(tr.accelerations[0], dtype=np.float64).ravel()
            tr.pop()                                                   # utils.py:739
        tgt = np.asarray(self.planner.target, dtype=np.float64).ravel()
        self._push('target', tgt[:2])                                  # planners may move the target
        return ok, has_wp, wp

    def _finish_step(self):
        """Refresh the reference-visible attributes from the mirror and build the step's return value."""
        m = self._mirror
        c, f = m['counters'], m['flags']
        self.steps = int(c[A.C_STEPS])
        self.state_machine = int(c[A.C_SM])
        self.fail_count = int(c[A.C_FAIL])
        self.tracked_agent = int(c[A.C_TRACKED])
        left = int(c[A.C_NTGT] - c[A.C_TGT_NEXT])
        self.target_list = self.target_list[-left:] if left > 0 else []
        nbuf, nts = int(c[A.C_BUF_N]), int(c[A.C_BUF_TS])
        self.tracker_buffer = [_ArchivedTracker(nts - (nbuf - 1))] + [_ArchivedTracker(1)] * (nbuf - 1) if nbuf else []
        done = bool(f[A.F_DONE])
        self.info = self._info(int(f[A.F_COLLISION]), int(f[A.F_DEADLOCK]), int(f[A.F_FREEZING]))
        state = {'local_map': m['obs_local'][None], 'swep_map': m['obs_local'][None],       # drone_v2.py:251-255
                 'yaw_angle': np.array([m['obs_yaw']], dtype=np.float32).flatten()}
        return state, 0, done, self.info

    def _perceive(self):
        """d2d_perceive; with measurement noise (var_cam != 0) split after the raycast, because the reference draws
        np.random.randn(2) for the agents the rays hit, in agent order (utils.py:603-605)."""
        vec = self._vec
        if self.params.var_cam == 0 or vec.N == 0:
            vec.backend.perceive(vec.cfg, vec._st)
            return
        vec.backend.run_stages(vec.cfg, vec._st, A.ST_FSM | A.ST_AGENTS | A.ST_RAYCAST)
        self._pull(only=('hit',))
        noise = np.zeros((1, vec.N, 2))
        for k in np.nonzero(self._mirror['hit'])[0]:
            noise[0, k] = self._noise_rng.randn(2)
        vec.set_noise(noise)
        vec.backend.run_stages(vec.cfg, vec._st, A.ST_DYNGRID | A.ST_TRACKER)

    def step(self, a):
        vec = self._vec
        if not (isinstance(a, DeviceAction) and a.fresh_for(self)):    # else: d2d_gaze_stage already wrote the action
            vec._set_action(float(a) if isinstance(a, DeviceAction) else float(np.asarray(a, dtype=np.float64).ravel()[0]))
        if self._mode == 'host':
            self._perceive()
            self._pull()
            ok, has_wp, wp = self._plan_phase()
            vec.set_plan([ok], [has_wp], wp[None])
        elif self._mode == 'device':
            self._perceive()
            vec.backend.plan_stage(vec.cfg, vec._st, vec._plan)
        elif self.params.var_cam != 0 and vec.N:
            self._perceive()
        else:
            vec.backend.step(vec.cfg, vec._st)                         # fused: perceive + act in one launch
            self._pull()
            return self._finish_step()
        vec.backend.act(vec.cfg, vec._st)
        self._pull()
        return self._finish_step()

    def render(self, mode='human'):
        """envs/drone_v2.py:263-303 is display only (pygame).  The reference's Params default to render=True and its
        Experiment.run calls env.render() every step (utils.py:75-77, experiment.py:105-106), so the default
        invocation has to survive it: headless no-op, one warning."""
        if not self._render_warned:
            import warnings
            warnings.warn('Drone2DEnv2.render(): the accelerated env is headless, render() does nothing '
                          '(pass --debug / render=False to silence this)', RuntimeWarning, stacklevel=2)
            self._render_warned = True
        return None


def register():
    """gym registration under the reference's id (envs/__init__.py:5-8).  No-op without gym."""
    if _gym is None:
        return False
    from gym.envs.registration import register as _reg
    try:
        _reg(id='gym-2d-perception-v2', entry_point='drone2d_amd.env:Drone2DEnv2')
    except Exception:
        pass
    return True
