"""Drop-in check of the plugin surface with the REFERENCE'S OWN planner and gaze classes (only where the reference
checkout exists, i.e. in the build container; skipped on the GPU box): `traj_planner.Primitive` and
`yaw_planner.Oxford`, imported from /root/reference, drive this repo's env through its proxies and reproduce the
reference episode."""
import os
import sys
import types

import numpy as np
import pytest

from replay import load, params_from

REF = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason='reference checkout not present')


@pytest.fixture(scope='module')
def ref_modules():
    def mk(n):
        m = types.ModuleType(n)
        sys.modules[n] = m
        return m
    saved = {k: sys.modules.get(k) for k in ('pygame', 'cvxpy', 'cvxpy.error', 'gym', 'gym.spaces', 'gym.envs',
                                             'gym.envs.registration', 'utils', 'traj_planner', 'yaw_planner')}
    mk('pygame')
    cv = mk('cvxpy'); ce = mk('cvxpy.error'); ce.SolverError = Exception; cv.error = ce
    if 'gym' not in sys.modules or sys.modules['gym'] is None:
        gym = mk('gym'); gym.__path__ = []
        gym.Env = type('Env', (), {})
        gym.logger = type('L', (), {'set_level': staticmethod(lambda x: None)})
        sp = mk('gym.spaces'); sp.Box = type('Box', (), {'__init__': lambda s, *a, **k: None}); sp.Dict = type('Dict', (), {'__init__': lambda s, d: None})
        gym.spaces = sp
        ge = mk('gym.envs'); ge.__path__ = []
        gr = mk('gym.envs.registration'); gr.register = lambda **k: None
        ge.registration = gr; gym.envs = ge
    import matplotlib
    matplotlib.use('Agg')
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        import traj_planner
        import yaw_planner
        yield traj_planner, yaw_planner
    finally:
        os.chdir(cwd)
        sys.path.remove(REF)
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_reference_primitive_and_oxford_drive_this_env(pkg, oracle, ref_modules):
    traj_planner, yaw_planner = ref_modules
    from drone2d_amd import env as envmod, planners
    fx = load('readme_oxford_primitive')
    p = params_from(fx, pkg)
    own = planners.planner_list['Primitive']
    planners.register_planner('Primitive', traj_planner.Primitive)      # the reference's class, unmodified
    try:
        e = envmod.Drone2DEnv2(p, backend=oracle)
        assert type(e.planner).__module__ == 'traj_planner'
        pol = yaw_planner.Oxford
        pol.__init__(pol, p)
        done, t = False, 0
        while not done and t < 400:
            a = pol.plan(pol, e.info)
            assert abs(float(a) - fx['t_action'][t]) <= 1e-12
            obs, rew, done, info = e.step(a)
            d = fx['t_drone'][t]
            assert (e.drone.x, e.drone.y) == (d[0], d[1]) and np.array_equal(obs['local_map'][0], fx['t_obs_local'][t])
            t += 1
        assert t == 210 and info['state_machine'] == 1 and (e.drone.x, e.drone.y) == (42, 455)
    finally:
        planners.register_planner('Primitive', own)


def test_reference_experiment_driver_runs_unchanged(pkg, oracle, ref_modules):
    """experiment.py of the reference (its Experiment class, unmodified) on this env: gym.make is pointed at
    Drone2DEnv2, as the one-line registration change of INTEGRATION.md does."""
    import gym
    from drone2d_amd import env as envmod
    fx = load('lookahead_primitive_n30_map3')
    p = params_from(fx, pkg)
    gym.make = lambda env_id, params=None: envmod.Drone2DEnv2(params, backend=oracle)
    saved_path = list(sys.path)
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        sys.modules.pop('experiment', None)
        import experiment                       # the reference's driver module
        ex = experiment.Experiment(p, '/tmp/unused.csv')
        ex.run()
        info = ex.env.info
        assert ex.env.steps == len(fx['t_action']) and info['collision_flag'] == fx['t_flags'][-1][0]
        assert np.array_equal(ex.env.drone.map.grid_map, fx['t_dmap'][-1])
    finally:
        os.chdir(cwd)
        sys.path[:] = saved_path
        sys.modules.pop('experiment', None)
