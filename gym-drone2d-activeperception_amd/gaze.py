"""--gaze_method plugin surface (reference: yaw_planner.py; registry experiment.py:12-19).

A policy is built from `params` and maps the env's `info` dict to a yaw-rate action in [-1, 1].  The reference
drives its policies "class as instance" (`policy.__init__(policy, params)`, `policy.plan(policy, info)`,
experiment.py:33-34,69); the classes here work that way and as ordinary instances.

`Oxford` IS the device stage (csrc/d2d_plugins.h `gaze_env`, yaw_planner.py:41-127): `plan(info)` launches
`d2d_gaze_stage` for the env behind `info['drone']` and returns a `DeviceAction` -- the action stays in the env's
action buffer on the device, `env.step` recognises the token and uploads nothing; `float(action)` reads it back for
callers that want the number.  `NoControl` and `Rotating` are the reference's constants; `LookAhead` (the method
main.py:10 and script/train.py:18 select) and `LookGoal` are a dozen scalar operations on what the step already
mirrors to the host (velocity, yaw; the stored trajectory and the drone's map), so they run here on the host with
the same libm calls as the reference (`math.atan2`, `math.degrees`, float `%`).  Any other policy (the reference's
`Owl`, or its `Oxford` as a host object) is a host plugin too: it reads the env through the `info` proxies; names
this registry does not know resolve through the reference's `yaw_planner` module when that is importable.
"""
import math

from .planners import _Registry


def _yaw_rate_towards(heading_deg, yaw_deg, dt, max_rate):
    """Normalised yaw rate that turns `yaw_deg` towards `heading_deg` (both in [0, 360) or not: the reference
    compares them as they are) within one period `dt`, clipped to +-max_rate; the long way round flips the sign
    (yaw_planner.py:34-38 = :251-255)."""
    delta = heading_deg - yaw_deg
    rate = max(min(delta / dt, max_rate), -max_rate)
    if not abs(delta) < 180:
        rate = -rate
    return rate / max_rate


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
    """yaw_planner.py:18-39: look where the drone is flying.  Heading = atan2(-vy, vx) in degrees mod 360 (screen
    y points down); a drone at rest keeps its yaw."""

    def __init__(self, params):
        self.params = params
        self.dt = params.dt

    def plan(self, state):
        vx, vy = state['drone'].velocity[0], state['drone'].velocity[1]
        if vx == 0 and vy == 0:
            return 0
        heading = math.degrees(math.atan2(-vy, vx)) % 360
        return _yaw_rate_towards(heading, state['drone'].yaw, self.dt, self.params.drone_max_yaw_speed)


class LookGoal:
    """yaw_planner.py:225-257: look at the first waypoint of the stored trajectory that lies in a cell the drone has
    not explored yet, else at the last waypoint; no trajectory, no turn."""
    UNEXPLORED = 0                                                     # utils.py grid_type

    def __init__(self, params):
        self.params = params

    def plan(self, observation):
        drone, waypoints = observation['drone'], observation['trajectory'].positions
        if len(waypoints) == 0:
            return 0
        look = next((w for w in waypoints if drone.map.get_grid(w[0], w[1]) == self.UNEXPLORED), waypoints[-1])
        heading = math.degrees(math.atan2(-(look[1] - drone.y), look[0] - drone.x)) % 360
        return _yaw_rate_towards(heading, drone.yaw, self.params.dt, self.params.drone_max_yaw_speed)


class DeviceAction:
    """The gaze action of one env as it sits in the device's action buffer (written by d2d_gaze_stage)."""
    __slots__ = ('_env', '_stamp', '_value')

    def __init__(self, env):
        self._env, self._stamp, self._value = env, env._pull_count, None

    def fresh_for(self, env):
        return env is self._env and self._stamp == env._pull_count

    def __float__(self):
        if self._value is None:
            self._value = float(self._env._vec.state.action[self._env._slot].cpu())
        return self._value

    def __repr__(self):
        return f'DeviceAction({float(self)!r})'


class Oxford:
    """yaw_planner.py:41-127 on the device.  The policy's own state (the time-since-observed map) lives with the env's
    plugin state (`d2d_plan.seen_step`) and starts fresh with every env, as experiment.py:31-34 builds both anew
    per episode."""

    def __init__(self, params):
        self.params = params

    def plan(self, observation):
        env = getattr(observation['drone'], '_env', None)
        if env is None or not env._device_gaze:
            raise TypeError('the device Oxford policy needs an env built with gaze_method="Oxford" and a device planner '
                            "(Primitive / NoMove); for other combinations register the reference's yaw_planner.Oxford")
        env._vec.backend.gaze_stage(env._vec.cfg, env._vec._st, env._vec._plan)
        return DeviceAction(env)


class _PolicyRegistry(_Registry):
    module, what = 'yaw_planner', 'gaze policy'


policy_list = _PolicyRegistry(NoControl=NoControl, Rotating=Rotating, Oxford=Oxford, LookAhead=LookAhead, LookGoal=LookGoal)


def register_policy(name, cls):
    """Plug a gaze policy class with the reference interface (`__init__(params)`, `plan(info)`)."""
    policy_list[name] = cls
