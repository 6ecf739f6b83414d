"""--gaze_method plugin surface (reference: yaw_planner.py; registry experiment.py:12-19).

A policy is built from `params` and maps the env's `info` dict to a yaw-rate action in [-1, 1].  The reference
drives its policies "class as instance" (`policy.__init__(policy, params)`, `policy.plan(policy, info)`,
experiment.py:33-34,69); the classes here work that way and as ordinary instances.

`Oxford` IS the device stage (csrc/d2d_plugins.h `gaze_env`, yaw_planner.py:41-127): `plan(info)` launches
`d2d_gaze_stage` for the env behind `info['drone']` and returns a `DeviceAction` -- the action stays in the env's
action buffer on the device, `env.step` recognises the token and uploads nothing; `float(action)` reads it back for
callers that want the number.  `NoControl` and `Rotating` are the reference's constants.  Any other policy (the
reference's own `LookAhead`, `LookGoal`, `Owl`, or its `Oxford` as a host object) is a host plugin: it reads the env
through the `info` proxies; names this registry does not know resolve through the reference's `yaw_planner` module
when that is importable.
"""
from .planners import _Registry


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


policy_list = _PolicyRegistry(NoControl=NoControl, Rotating=Rotating, Oxford=Oxford)


def register_policy(name, cls):
    """Plug a gaze policy class with the reference interface (`__init__(params)`, `plan(info)`)."""
    policy_list[name] = cls
