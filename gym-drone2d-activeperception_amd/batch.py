"""HostPluginBatch — B reference-style episodes (any --planner / --gaze_method plugin) in lock-step on one
device batch.

The reference's sweeps (`main.py:26-57`, `script/validation_*.py`) run thousands of single-env episodes one
after another.  Here B of them share one VecDrone2DEnv: per step ONE `d2d_perceive` launch for all, the host
planner / gaze plugins of every unfinished episode (they are Python objects with the reference's interface
and see their own env through the same proxies as `Drone2DEnv2`), ONE `d2d_act` launch for all.  Each slot
behaves exactly like a stand-alone `Drone2DEnv2` (tests compare them step for step); finished episodes are
parked (no planner calls, no movement) until `run()` returns.
"""
import numpy as np

from . import _abi as A
from .env import Drone2DEnv2
from .gaze import policy_list
from .host_init import init_world
from .params import with_defaults
from .vec_env import VecDrone2DEnv


class HostPluginBatch:
    def __init__(self, params_list, device='cuda:0', backend=None):
        self.params = [with_defaults(p) for p in params_list]
        for p in self.params:
            if p.gaze_method == 'NoControl':
                p.drone_view_range = 360                                   # experiment.py:28-29
        worlds = [init_world(p) for p in self.params]
        self.vec = VecDrone2DEnv(self.params[0], len(worlds), device=device, backend=backend, planner='external',
                                 worlds=worlds)
        self.envs = [Drone2DEnv2(p, _shared=(self.vec, i)) for i, p in enumerate(self.params)]
        self.policies = []
        for p in self.params:                                              # own instance per episode (the reference
            cls = policy_list[p.gaze_method]                               # keeps policy state on the class)
            pol = cls.__new__(cls)
            cls.__init__(pol, p)
            self.policies.append(pol)
        self.done = np.zeros(len(worlds), dtype=bool)
        self.infos = [e.info for e in self.envs]

    def __len__(self):
        return len(self.envs)

    def step(self):
        """One lock-step step of every unfinished episode: gaze plugins -> perceive -> planner plugins -> act."""
        B = len(self.envs)
        actions = np.zeros(B)
        for i, (e, pol) in enumerate(zip(self.envs, self.policies)):
            if not self.done[i]:
                a = pol.plan(e.info)                                       # experiment.py:69
                actions[i] = 0.0 if a is None else float(np.asarray(a).ravel()[0])
        self.vec.perceive()
        ok, has_wp, wp = np.ones(B, dtype=np.uint8), np.zeros(B, dtype=np.uint8), np.zeros((B, 6))
        for i, e in enumerate(self.envs):
            if not self.done[i]:
                e._pull()
                ok[i], has_wp[i], wp[i] = e._plan_phase()
        self.vec.set_plan(ok, has_wp, wp)
        self.vec.act(actions)
        out = []
        for i, e in enumerate(self.envs):
            if not self.done[i]:
                e._pull()
                obs, rew, d, info = e._finish_step()
                self.infos[i] = info
                self.done[i] = d
                out.append((obs, rew, d, info))
            else:
                out.append(None)
        return out

    def run(self, max_steps=100000):
        """Run every episode to its end (experiment.py:65-70 for each slot).  Returns the final `info` dicts."""
        n = 0
        while not self.done.all() and n < max_steps:
            self.step()
            n += 1
        return self.infos
