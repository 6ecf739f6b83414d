"""Replays a golden trace (tests/golden/*.npz, captured from the imported reference) through a backend
(the CPU oracle, or the HIP library on the GPU) and compares every recorded output.

Integer outputs (grids, hit mask, flags, counters, local-map observation) must match bit for bit;
float state within `FTOL` absolute (north_star: 1e-5 on float32 state; the fp64 path is far inside)."""
import json
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
FTOL = 0.0
KF_TOL = 1e-6


def load(name):
    """A committed fixture by name, or any trace file by absolute path."""
    return np.load(name if os.path.isabs(name) else os.path.join(GOLD, name + '.npz'))


def params_from(fx, pkg):
    d = json.loads(str(fx['params_json']))
    p = pkg.Params()
    for k, v in d.items():
        setattr(p, k, v)
    return p


class Replay:
    kf_max_dev = 0.0

    def __init__(self, pkg, backend, name, kf=True, copies=1):
        from drone2d_amd import host_init, state
        self.A = pkg._abi
        self.fx = load(name)
        self.name = name
        self.p = params_from(self.fx, pkg)
        self.backend = backend
        world = host_init.init_world(self.p)
        nomove = self.p.planner == 'NoMove'
        self.copies = copies
        self.cfg = host_init.derive_cfg(self.p, B=copies, N=world['N'], T=world['T'],
                                        planner_mode=self.A.PLANNER_NOMOVE if nomove else self.A.PLANNER_EXTERNAL,
                                        kf_enabled=kf)
        self.st = state.BatchState(self.cfg, backend.device)
        self.st.load_worlds([world] * copies)
        self.world = world
        self.T = len(self.fx['t_action'])
        self.t = 0

    def set_inputs(self, t):
        fx, st = self.fx, self.st
        st.action.fill_(float(fx['t_action'][t]))
        if 't_tele' in fx.files:
            st.drone[:, self.A.D_X] = float(fx['t_tele'][t][0])
            st.drone[:, self.A.D_Y] = float(fx['t_tele'][t][1])
        if self.cfg.planner_mode == self.A.PLANNER_EXTERNAL:
            st.plan_ok.fill_(int(fx['t_plan_ok'][t]))
            st.wp_valid.fill_(int(fx['t_wp_valid'][t]))
            st.wp.copy_(torch.from_numpy(np.tile(fx['t_wp'][t], (self.copies, 1))))
        if not self.cfg.kf_enabled:
            st.active.copy_(torch.from_numpy(np.tile(fx['t_active_pre'][t], (self.copies, 1))))

    def compare(self, t, e=0, check_kf=None):
        fx, st, A = self.fx, self.st, self.A
        g = lambda k: st.t[k][e].cpu().numpy()
        tag = f'{self.name} step {t + 1} env {e}: '
        ag = g('agents')
        np.testing.assert_allclose(ag[A.A_PX], fx['t_agent_pos'][t][:, 0], rtol=0, atol=FTOL, err_msg=tag + 'agent x')
        np.testing.assert_allclose(ag[A.A_PY], fx['t_agent_pos'][t][:, 1], rtol=0, atol=FTOL, err_msg=tag + 'agent y')
        np.testing.assert_allclose(ag[A.A_VX], fx['t_agent_pref'][t][:, 0], rtol=0, atol=FTOL, err_msg=tag + 'pref x')
        np.testing.assert_allclose(ag[A.A_VY], fx['t_agent_pref'][t][:, 1], rtol=0, atol=FTOL, err_msg=tag + 'pref y')
        assert np.array_equal(g('hit'), fx['t_hit'][t]), tag + 'hit mask'
        assert np.array_equal(g('dmap'), fx['t_dmap'][t]), tag + 'drone map'
        assert np.array_equal(g('gt'), fx['t_gt'][t]), tag + 'gt grid'
        d = g('drone')
        np.testing.assert_allclose(d[[A.D_X, A.D_Y, A.D_YAW]], fx['t_drone'][t], rtol=0, atol=FTOL, err_msg=tag + 'drone')
        np.testing.assert_allclose(d[[A.D_VX, A.D_VY, A.D_AX, A.D_AY]], fx['t_vel'][t], rtol=0, atol=FTOL, err_msg=tag + 'vel')
        assert np.array_equal(g('flags')[:3], fx['t_flags'][t]), tag + f"flags {g('flags')} vs {fx['t_flags'][t]}"
        assert bool(g('flags')[A.F_DONE]) == bool(fx['t_done'][t]), tag + 'done'
        c = g('counters')
        assert c[A.C_SM] == fx['t_sm'][t], tag + 'state machine'
        assert c[A.C_FAIL] == fx['t_fail'][t], tag + 'fail_count'
        assert c[A.C_STEPS] == fx['t_steps'][t], tag + 'steps'
        assert np.array_equal(g('obs_local'), fx['t_obs_local'][t]), tag + 'obs local_map'
        assert g('obs_yaw') == fx['t_obs_yaw'][t][0], tag + 'obs yaw'
        np.testing.assert_allclose(g('target'), fx['t_target'][t][:2], rtol=0, atol=0, err_msg=tag + 'target')
        assert g('newly') == fx['t_newly'][t], tag + f"newly_tracked {g('newly')} vs {fx['t_newly'][t]}"
        if self.cfg.kf_enabled if check_kf is None else check_kf:
            assert np.array_equal(g('active'), fx['t_active_post'][t]), tag + 'tracker active bits'
            assert np.array_equal(g('kf_len'), fx['t_kf_len'][t]), tag + 'tracker len(ts)'
            kf = g('kf')
            dev = max(float(np.max(np.abs(kf[:, :4] - fx['t_kf_mu'][t]), initial=0.0)),
                      float(np.max(np.abs(kf[:, 4:].reshape(-1, 4, 4) - fx['t_kf_sigma'][t]), initial=0.0)))
            Replay.kf_max_dev = max(Replay.kf_max_dev, dev)   # how far the tracker state really is from numpy's LAPACK
            np.testing.assert_allclose(kf[:, :4], fx['t_kf_mu'][t], rtol=KF_TOL, atol=KF_TOL, err_msg=tag + 'kf mu')
            np.testing.assert_allclose(kf[:, 4:].reshape(-1, 4, 4), fx['t_kf_sigma'][t], rtol=KF_TOL, atol=KF_TOL,
                                       err_msg=tag + 'kf Sigma')
            assert c[A.C_BUF_N] == fx['t_buf_len'][t], tag + f"tracker_buffer len {c[A.C_BUF_N]} vs {fx['t_buf_len'][t]}"

    def run(self, mode='fused', every=1, envs=(0,)):
        for t in range(self.T):
            self.set_inputs(t)
            s = self.st.struct()
            if mode == 'fused':
                self.backend.step(self.cfg, s)
            elif mode == 'split':
                self.backend.perceive(self.cfg, s)
                self.backend.act(self.cfg, s)
            elif mode == 'stages':
                for b in range(8):
                    self.backend.run_stages(self.cfg, s, 1 << b)
            self.backend.sync()
            if t % every == 0 or t == self.T - 1:
                for e in envs:
                    self.compare(t, e)


TRACES_NOMOVE = ['nomove_n10_const', 'nomove_n10_rand_map0', 'nomove_n10_rand_map2', 'nomove_n10_rand_map3',
                 'nomove_n10_rand_map7', 'nomove_teleport_structured', 'surv_pinned_360', 'nomove_obstacle_map',
                 'nomove_shaped_map', 'nomove_random_map_n172', 'nomove_pillars_randr', 'nomove_slow_agents',
                 'nomove_big_map', 'freezing_nomove', 'nomove_short_view_d40',
                 'nomove_cfg5_640']   # BASELINE config 5's geometry: 640 x 640 cells, 640 rays, 100 agents
TRACES_PLANNED = ['readme_oxford_primitive', 'lookahead_primitive_n30_map0', 'lookahead_primitive_n30_map3',
                  'deadlock_primitive']
TRACES_CLOSED = ['closed_oxford_n20_map0', 'closed_oxford_n20_map5', 'closed_oxford_pillars_map2', 'closed_oxford_pillars_map6',
                 'closed_oxford_slow_drone', 'closed_oxford_fast_drone', 'closed_oxford_fov120', 'closed_oxford_two_targets',
                 'closed_oxford_goal_at_start', 'closed_oxford_short_view_d50',
                 # round 3: BASELINE configs 3 (random_map_0, 172 agents) and 4 (obstacle_map, 24 agents), a 1000 x 800 px map with a 120 degree view
                 'closed_oxford_config3', 'closed_oxford_config4', 'closed_oxford_map1000x800',
                 'closed_oxford_cfg5_640']    # config 5's geometry: 640 x 640 cells, 640 rays, 100 agents, Oxford + Primitive
ALL_TRACES = TRACES_NOMOVE + TRACES_PLANNED + TRACES_CLOSED
