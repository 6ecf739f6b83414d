"""Host world construction vs the reference's __init__ (fixtures: tests/golden/init_cases.npz)."""
import json

import numpy as np
import pytest

from replay import load


def _cases():
    fx = load('init_cases')
    return fx, int(fx['n_cfg'])


@pytest.mark.parametrize('i', range(_cases()[1]))
def test_init_world_matches_reference(pkg, i):
    from drone2d_amd import host_init
    A = pkg._abi
    fx, _ = _cases()
    cfg = json.loads(str(fx[f'c{i}_cfg']))
    p = pkg.Params(planner='NoMove', **cfg)
    w = host_init.init_world(p)
    ag = w['agents']
    # seeded streams are identical, so positions / radii / velocities are bit-identical
    assert np.array_equal(ag[A.A_PX], fx[f'c{i}_agent_pos'][:, 0])
    assert np.array_equal(ag[A.A_PY], fx[f'c{i}_agent_pos'][:, 1])
    assert np.array_equal(ag[A.A_VX], fx[f'c{i}_agent_pref'][:, 0])
    assert np.array_equal(ag[A.A_VY], fx[f'c{i}_agent_pref'][:, 1])
    assert np.array_equal(ag[A.A_R], fx[f'c{i}_agent_radius'])
    assert np.array_equal(w['group'], fx[f'c{i}_agent_group'])
    assert np.array_equal(w['gt'], fx[f'c{i}_gt0'])
    assert np.array_equal(w['obstacles'], fx[f'c{i}_obstacles'])
    assert np.array_equal(w['drone'][:3], fx[f'c{i}_drone0'])
    n = min(len(w['tracker_radius']), len(fx[f'c{i}_tracker_radius']))
    assert np.array_equal(w['tracker_radius'][:n], fx[f'c{i}_tracker_radius'][:n])
    # every DYNAMIC cell of the initial grid lies inside its agent's first clear block
    dyn = fx[f'c{i}_dyn_idx0']
    cov = np.zeros_like(w['gt'], dtype=bool)
    for cx, cy, u in w['dyn_prev']:
        cov[max(cx - u, 0):cx + u + 1, max(cy - u, 0):cy + u + 1] = True
    assert cov[dyn[:, 0], dyn[:, 1]].all()
    assert ((w['gt'] == A.DYNAMIC) <= cov).all()


def test_params_surface(pkg):
    p = pkg.Params()
    assert p.render is True and p.record is False            # --debug quirk, utils.py:75-80
    assert p.init_position == [50, 50] and p.target_list == [[50, 460]]
    q = pkg.Params.from_parser(['--agent_number', '20', '--debug', '--planner', 'NoMove', '--map_size', '600', '400'])
    assert q.agent_number == 20 and q.render is False and q.planner == 'NoMove' and q.map_size == [600, 400]
    v = pkg.with_defaults({'agent_number': 3, 'init_pos': [70, 80]})
    assert v.agent_number == 3 and v.init_position == [70, 80] and v.max_flight_time == 80
