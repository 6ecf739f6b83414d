"""The gym facade driven closed loop on the GPU (HIP library) against the reference traces."""
import pytest

from test_env_facade_cpu import run_closed_loop

pytestmark = pytest.mark.gpu


def test_readme_config_on_device(pkg, hip):
    e, info = run_closed_loop(pkg, hip, 'readme_oxford_primitive', 'Oxford')
    assert (e.drone.x, e.drone.y) == (42, 455) and info['state_machine'] == 1 and e.steps == 210


def test_lookahead_primitive_on_device(pkg, hip):
    run_closed_loop(pkg, hip, 'lookahead_primitive_n30_map3', None)    # recorded gaze actions, device planner


def test_train_py_shaped_params_on_device(pkg, hip):
    from test_env_facade_cpu import test_train_py_shaped_params
    test_train_py_shaped_params(pkg, hip)


def test_default_backend_is_hip(pkg):
    from drone2d_amd import env as envmod, _lib
    e = envmod.Drone2DEnv2(pkg.Params(planner='NoMove', agent_number=5))
    assert isinstance(e._backend, _lib.HipBackend)
    obs, r, done, info = e.step(0.5)
    assert float(obs['yaw_angle'][0]) == 274.0
