"""Episode driver with the reference's Experiment surface (experiment.py:26-106): builds env + gaze policy,
runs one episode, appends one CSV row with the reference's columns.  `main.py` of the reference does
`Experiment(cfg, result_dir).run()`; this is the same object on the accelerated env."""
import csv
import os

import numpy as np

from . import _abi as A
from .env import Drone2DEnv2
from .gaze import policy_list
from .params import with_defaults

CSV_COLUMNS = ['Method', 'Planner', 'Motion Profile', 'Map ID', 'Agent size', 'Number of agents', 'Number of pillars',
               'Agent speed', 'Drone speed', 'Depth variance', 'Initial position', 'Target position', 'Flight time',
               'Grid discovered', 'Agent tracked', 'Agent tracked time', 'Success', 'Static Collision',
               'Dynamic Collision', 'Freezing', 'Dead Lock', 'state machine']


class Experiment:
    def __init__(self, params, dir=None, device='cuda:0', backend=None):
        p = with_defaults(params)
        if p.gaze_method == 'NoControl':
            p.drone_view_range = 360                                   # experiment.py:28-29
        self.params = p
        self.env = Drone2DEnv2(p, device=device, backend=backend)
        self.dt = p.dt
        self.policy = policy_list[p.gaze_method]
        self.policy.__init__(self.policy, p)                           # class as instance, experiment.py:33-34
        self.result_dir = dir
        if dir and not os.path.isfile(dir) and p.record:
            with open(dir, 'w', newline='') as f:
                csv.writer(f).writerow(CSV_COLUMNS)

    def row(self, info):
        """The CSV row of experiment.py:73-103."""
        p = self.params
        n = len(info['tracker_buffer'])
        tracking_time = float(np.array([len(t.ts) * 0.1 for t in info['tracker_buffer']]).sum())
        gm = info['drone'].map.grid_map
        with np.errstate(divide='ignore', invalid='ignore'):
            mean_time = np.float64(tracking_time) / n if n else float('nan')
        return (p.gaze_method, p.planner, p.motion_profile, p.map_id, p.agent_radius, p.agent_number,
                p.pillar_number, p.agent_max_speed, p.drone_max_speed, p.var_cam, p.init_position, p.target_list[0],
                info['flight_time'], gm.shape[0] * gm.shape[1] - np.sum(np.where(gm == 0, 1, 0)), n, mean_time,
                1 if info['state_machine'] == A.SM_GOAL_REACHED else 0, 1 if info['collision_flag'] == 1 else 0,
                1 if info['collision_flag'] == 2 else 0, info['freezing_flag'], info['dead_lock_flag'],
                info['state_machine'])

    def run(self):
        self.env.reset()
        done, info = False, self.env.info
        while not done:
            a = self.policy.plan(self.policy, self.env.info)
            _, _, done, info = self.env.step(0.0 if a is None else a)
        row = self.row(info)
        if self.params.record and self.result_dir:
            with open(self.result_dir, 'a', newline='') as f:
                csv.writer(f).writerow(row)
        return row


class ExperimentBatch:
    """The reference's sweep of episodes (`main.py:26-57`: the same cfg over many map ids, one Experiment and one
    CSV row each) as ONE device batch: env i is the world of `map_id + i`, the gaze policy and the planner run on
    the device (Oxford / Primitive), every env plays exactly one episode (`D2D_DONE_FREEZE`) and `rows()` returns
    the reference's CSV rows.  Everything an episode needs stays on the GPU; the host only reads the rows.

    `LookAhead` (main.py:10's default method) is the one host policy a batch runs: it needs the drone's velocity and yaw
    only, so every step pulls those three numbers per env, evaluates yaw_planner.py:28-39 with the host's libm (the
    reference's `math.atan2`; there is no bit-exact device atan2 here) and uploads the actions -- one launch per step
    instead of one per episode."""

    def __init__(self, params, num_envs, device='cuda:0', backend=None, workers=0):
        from .vec_env import VecDrone2DEnv, build_worlds
        p = with_defaults(params)
        if p.gaze_method not in ('Oxford', 'Rotating', 'NoControl', 'LookAhead') or p.planner not in ('Primitive', 'NoMove'):
            raise NotImplementedError('ExperimentBatch runs the device plugins: gaze_method Oxford / Rotating / NoControl (and '
                                      'LookAhead from the host), planner Primitive / NoMove (use Experiment, one episode at a '
                                      'time, for other host plugin classes)')
        if p.gaze_method == 'NoControl':
            p.drone_view_range = 360                                   # experiment.py:28-29
        self.params = p
        worlds = build_worlds(p, num_envs, workers=workers)
        self.env = VecDrone2DEnv(p, num_envs, device=device, backend=backend, planner=p.planner, worlds=worlds,
                                 device_plugins=True, gaze='external' if p.gaze_method == 'LookAhead' else p.gaze_method)
        self.max_steps = int(np.ceil(p.max_flight_time / p.dt)) + 1           # freezing ends every episode by then

    def _lookahead_actions(self):
        """yaw_planner.LookAhead.plan for every env (gaze.LookAhead, vectorised over the batch on the host)."""
        from .gaze import _yaw_rate_towards
        import math
        p = self.params
        d = self.env.state.drone[:, [A.D_VX, A.D_VY, A.D_YAW]].cpu().numpy()
        out = np.zeros(len(d))
        for e, (vx, vy, yaw) in enumerate(d):
            if vx != 0 or vy != 0:
                out[e] = _yaw_rate_towards(math.degrees(math.atan2(-vy, vx)) % 360, yaw, p.dt, p.drone_max_yaw_speed)
        return out

    def run(self, chunk=None):
        n = self.max_steps
        if self.params.gaze_method == 'LookAhead':
            for t in range(n):
                self.env._set_action(self._lookahead_actions())
                self.env.closed_loop(1, freeze_done=True)
                if t % 16 == 15 and bool(self.env.state.flags[:, A.F_DONE].all()):
                    break
            self.env.sync()
            return self.rows()
        chunk = chunk or n
        for c0 in range(0, n, chunk):
            self.env.closed_loop(min(chunk, n - c0), freeze_done=True)
        self.env.sync()
        return self.rows()

    def rows(self):
        """One tuple per env with the columns of experiment.py:73-103."""
        p, s = self.params, self.env.state
        c = s.counters.cpu().numpy()
        f = s.flags.cpu().numpy()
        disc = (s.dmap != 0).flatten(1).sum(1).cpu().numpy()
        out = []
        for e in range(self.env.num_envs):
            n = int(c[e, A.C_BUF_N])
            tracking_time = float(c[e, A.C_BUF_TS]) * 0.1
            with np.errstate(divide='ignore', invalid='ignore'):
                mean_time = np.float64(tracking_time) / n if n else float('nan')
            sm = int(c[e, A.C_SM])
            out.append((p.gaze_method, p.planner, p.motion_profile, p.map_id + self.env.env_offset + e, p.agent_radius,
                        p.agent_number, p.pillar_number, p.agent_max_speed, p.drone_max_speed, p.var_cam, p.init_position,
                        p.target_list[0], int(c[e, A.C_STEPS]) * p.dt, int(disc[e]), n, mean_time,
                        1 if sm == A.SM_GOAL_REACHED else 0, 1 if f[e, 0] == 1 else 0, 1 if f[e, 0] == 2 else 0,
                        int(f[e, 2]), int(f[e, 1]), sm))
        return out

    def write_csv(self, path):
        new = not os.path.isfile(path)
        with open(path, 'a', newline='') as fh:
            w = csv.writer(fh)
            if new:
                w.writerow(CSV_COLUMNS)
            w.writerows(self.rows())
