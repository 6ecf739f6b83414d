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
