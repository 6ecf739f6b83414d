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
the same libm calls as the reference (`math.atan2`, `math.degrees`, float `%`); so does `Owl` (36 direction scores
and 20 candidate yaw rates per decision, one decision every 0.8 s).  Any other policy (the reference's `Oxford` as a
host object, a user's class) is a host plugin too: it reads the env through the `info` proxies; names this registry
does not know resolve through the reference's `yaw_planner` module when that is importable.
"""
import math

import numpy as np
from numpy.linalg import norm

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


def _apart(a, b):
    """Unsigned angle between two directions given in degrees (scalars or arrays), yaw_planner.py:144-149."""
    d = abs(a % 360 - b % 360)
    return np.minimum(d, 360 - d)


def _owl_unseen(o, theta):
    """0 inside the field of view, else the product of the angles (radians) to its two edges (`G`, yaw_planner.py:170-174).
    `o`: the policy object -- an instance, or the class itself under the reference's class-as-instance use."""
    if _apart(theta, 0) <= o.fov / 2:
        return 0
    return math.radians(_apart(theta, o.fov / 2)) * math.radians(_apart(theta, -o.fov / 2))


def _owl_refresh(o, drone):
    """`update_U`, yaw_planner.py:176-182."""
    moved = drone.velocity * o.dt
    for i, deg in enumerate(np.arange(0, 360, 10)):
        along = np.array([math.cos(math.radians(deg)), math.sin(math.radians(deg))])
        gain = -moved.dot(along) / o.params.drone_view_depth
        gain += o.IN_VIEW if _apart(deg, -drone.yaw) < o.fov / 2 else o.OUT_OF_VIEW
        o.score[i] = max(min(o.score[i] + gain, 1), 0)


def _owl_score(o, theta):
    """`U`, yaw_planner.py:184-186: the score of the ten-degree direction nearest to theta."""
    return o.score[np.argmin(_apart(np.arange(0, 360, 10), theta))]


class Owl:
    """yaw_planner.py:151-222.  Every 0.8 s: refresh the 36 ten-degree direction scores (how recently each direction was in
    view, shifted by the drone's motion), then pick the yaw rate among 20 candidates whose heading after 0.8 s costs least --
    weighted sum of: goal direction out of view, flight direction out of view (times speed squared), tracked agents out of
    view (times speed / distance), the score of the heading itself, and the turn -- and repeat it for the following six steps.
    The operations and their order are the reference's (NaNs of a drone at rest included: all costs NaN -> the first
    candidate).  One reference quirk is kept: the j-th ACTIVE tracker's direction is weighted with the state of tracker j
    (`zip(d_o, trackers)`, :197), whichever tracker that is."""
    HOLD = 0.8
    WEIGHTS = np.array([0.2, 0.9, 1, 0.1, 0])
    IN_VIEW, OUT_OF_VIEW, AGENT_GAIN = 0.4, -0.05, 1

    def __init__(self, params):
        self.params = params
        self.dt = self.HOLD
        top = params.drone_max_yaw_speed
        self.rates = np.arange(-top, top, top / 10)
        self.fov = params.drone_view_range
        self.queue = []
        self.score = np.zeros(36)

    def plan(self, observation):
        top = self.params.drone_max_yaw_speed
        if len(self.queue) != 0:
            return self.queue.pop() / top
        drone, target = observation['drone'], observation['target']
        trackers = drone.trackers
        with np.errstate(invalid='ignore', divide='ignore'):
            _owl_refresh(self, drone)
            here = np.array([drone.x, drone.y])
            to_goal = math.degrees(math.atan2(target[1] - drone.y, target[0] - drone.x))
            to_flight = math.degrees(math.atan2(*((drone.velocity / norm(drone.velocity))[::-1])))
            to_agents = [math.degrees(math.atan2(*((t.mu_upds[-1][:2, 0] - here)[::-1]))) for t in trackers if t.active is True]
            pull = [self.AGENT_GAIN * norm(t.mu_upds[-1][2:, 0]) / norm(t.mu_upds[-1][:2, 0] - here)
                    for _, t in zip(to_agents, trackers)]
            goal_unknown, flight_unknown = 1 - _owl_score(self, to_goal), 1 - _owl_score(self, to_flight)
            speed2 = norm(drone.velocity / 10) ** 2
            headings = -(drone.yaw + self.rates * self.dt)
            terms = np.zeros([headings.shape[0], 5])
            costs = np.ones_like(headings)
            for i, heading in enumerate(headings):
                terms[i, 0] = _owl_unseen(self, heading - to_goal) * goal_unknown
                terms[i, 1] = speed2 * _owl_unseen(self, heading - to_flight) * flight_unknown
                for w, d in zip(pull, to_agents):
                    terms[i, 2] += w * _owl_unseen(self, heading - d)
                terms[i, 3] = _owl_score(self, heading)
                terms[i, 4] = abs(math.radians(self.rates[i] * self.dt))
                costs[i] = np.sum(terms[i, :].dot(self.WEIGHTS))
            pick = self.rates[np.argmin(costs)]
        for _ in range(int(self.dt // self.params.dt) - 1):
            self.queue.append(pick)
        return pick / top


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


policy_list = _PolicyRegistry(NoControl=NoControl, Rotating=Rotating, Oxford=Oxford, LookAhead=LookAhead, LookGoal=LookGoal,
                              Owl=Owl)


def register_policy(name, cls):
    """Plug a gaze policy class with the reference interface (`__init__(params)`, `plan(info)`)."""
    policy_list[name] = cls
