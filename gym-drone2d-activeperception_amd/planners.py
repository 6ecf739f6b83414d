"""--planner plugin surface (reference: traj_planner.py; registry envs/drone_v2.py:70-75).

What the reference's callers touch is kept:
    planner = planner_list[params.planner](drone, params)
    planner.trajectory (.positions / .velocities / .accelerations, len(), pop(), clear()) ; planner.target ;
    planner.set_target(xy) ; planner.plan(drone, dt) -> bool ; planner.replan_check(drone) -> (bool, swep_map)

`Primitive` and `NoMove` ARE the device stages (csrc/d2d_plugins.h `plan_env`, D2D_PLANNER_NOMOVE): the objects
registered here hold no algorithm, they are views of the env's device-resident planner state -- `env.step`
runs `d2d_perceive -> d2d_plan_stage -> d2d_act` without coming back to the host in between.  Any other class
with the reference's interface (the reference's own `traj_planner.Primitive`, a third-party planner) is a host
plugin: register it with `register_planner` and `env.step` calls its replan_check / plan between the two
device halves of the step.  Names this registry does not know are looked up in the reference's `traj_planner`
module when that is importable (i.e. when this package sits inside the reference checkout).
"""
import numpy as np


class HostTrajectory:
    """Plain waypoint lists for host planner plugins (the surface of utils.py:280-294)."""

    def __init__(self):
        self.clear()

    def clear(self):
        self.positions, self.velocities, self.accelerations = [], [], []

    def pop(self):
        for seq in (self.positions, self.velocities, self.accelerations):
            del seq[0]

    def __len__(self):
        return len(self.positions)


class TrajectoryView:
    """The device planner's stored trajectory (`d2d_plan.traj`, `traj_hdr`) behind the reference's trajectory
    surface.  Waypoints are fetched from the device when somebody looks at them (a host gaze policy, a script),
    once per step; `len()` comes from the header the env mirrors every step anyway."""

    def __init__(self, env):
        self._env = env
        self._rows, self._stamp = None, None

    def _load(self):
        env = self._env
        stamp = (env._pull_count, tuple(int(v) for v in env._mirror['traj_hdr']))
        if self._stamp != stamp:
            head, stored = stamp[1]
            self._rows = env._vec.plugins.t['traj'][env._slot, head:stored].cpu().numpy().copy()
            self._stamp = stamp
        return self._rows

    def __len__(self):
        head, stored = self._env._mirror['traj_hdr']
        return int(stored - head)

    @property
    def positions(self):
        return [r[0:2].copy() for r in self._load()]

    @property
    def velocities(self):
        return [r[2:4].copy() for r in self._load()]

    @property
    def accelerations(self):
        return [np.array([0, 0]) for _ in range(len(self))]        # traj_planner.py:214

    def pop(self):
        hdr = self._env._mirror['traj_hdr']
        if hdr[1] - hdr[0] > 0:
            hdr[0] += 1
            self._env._push_plugin('traj_hdr', hdr)

    def clear(self):
        hdr = self._env._mirror['traj_hdr']
        hdr[:] = 0
        self._env._push_plugin('traj_hdr', hdr)


class Planner:
    """Base of the HOST planner plugins: the attributes env.step reads (traj_planner.py:18-26)."""
    on_device = False

    def __init__(self, drone, params):
        self.trajectory = HostTrajectory()
        self.params = params
        self.target = np.array([drone.x, drone.y, 0, 0])

    def set_target(self, target):
        self.target = np.zeros(4)
        self.target[:2] = target

    def plan(self, drone, update_t):
        raise NotImplementedError('No planner implemented!')

    def replan_check(self, drone):
        raise NotImplementedError('No replan checker implemented!')


class _DevicePlanner:
    """A planner that runs as a device stage of env.step; this object only exposes its state."""
    on_device = True

    def __init__(self, drone, params):
        env = getattr(drone, '_env', None)
        if env is None:
            raise TypeError(f'{type(self).__name__} runs on the device inside Drone2DEnv2.step; it is built by the env '
                            '(planner_list[params.planner](env.drone, params)), not for a free-standing drone')
        self._env = env
        self.params = params

    @property
    def target(self):
        t = self._env._mirror['target']
        return np.array([t[0], t[1], 0., 0.])

    @target.setter
    def target(self, value):
        self.set_target(np.asarray(value, dtype=np.float64).ravel()[:2])

    def set_target(self, target):
        t = np.asarray(target, dtype=np.float64).ravel()[:2].copy()
        self._env._mirror['target'] = t
        self._env._push('target', t)

    def plan(self, drone, update_t):
        raise RuntimeError(f'{type(self).__name__}.plan runs inside env.step on the device (d2d_plan_stage)')

    def replan_check(self, drone):
        raise RuntimeError(f'{type(self).__name__}.replan_check runs inside env.step on the device (d2d_plan_stage)')


class NoMove(_DevicePlanner):
    """traj_planner.py:68-76 as D2D_PLANNER_NOMOVE: always succeeds, never moves, target (-1, -1)."""

    def __init__(self, drone, params):
        super().__init__(drone, params)
        self.trajectory = HostTrajectory()


class Primitive(_DevicePlanner):
    """traj_planner.py:78-233 as the device stage `plan_env` (csrc/d2d_plugins.h): replan_check, the
    motion-primitive A* and the head waypoint all run between d2d_perceive and d2d_act."""

    def __init__(self, drone, params):
        super().__init__(drone, params)
        self.trajectory = TrajectoryView(self._env)


class _Registry(dict):
    """Name -> class.  Unknown names fall back to the reference's own module when it is importable."""
    module, what = 'traj_planner', 'planner'

    def __missing__(self, name):
        try:
            mod = __import__(self.module)
            cls = getattr(mod, name)
        except Exception:
            raise KeyError(f'unknown {self.what} {name!r}: known here {sorted(self)}; other names resolve through the '
                           f"reference's {self.module}.py when it is importable") from None
        return cls

    def __contains__(self, name):
        if dict.__contains__(self, name):
            return True
        try:
            self.__missing__(name)
            return True
        except KeyError:
            return False


planner_list = _Registry(Primitive=Primitive, NoMove=NoMove)


def register_planner(name, cls):
    """Plug a planner class with the reference interface (e.g. the reference's own traj_planner classes)."""
    planner_list[name] = cls
