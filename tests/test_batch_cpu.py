"""HostPluginBatch: several reference-style episodes in lock-step on one device batch equal the same episodes
run one at a time (and therefore the reference: the single-env facade is pinned by test_env_facade_cpu.py)."""
import numpy as np
import pytest

from replay import load, params_from


def _check(pkg, backend, device):
    from drone2d_amd import batch, runner
    fxs = [load(n) for n in ('readme_oxford_primitive', 'lookahead_primitive_n30_map3')]
    # same N is required inside one batch: three README-config episodes with different seeds + gaze methods
    ps = []
    for mid, gaze in ((1, 'Oxford'), (2, 'LookAhead'), (3, 'Rotating')):
        p = pkg.Params(debug=True, gaze_method=gaze, planner='Primitive', agent_number=10, agent_max_speed=20,
                       agent_radius=15, drone_max_speed=40, map_id=mid)
        p.render = False
        ps.append(p)
    hb = batch.HostPluginBatch(ps, device=device, backend=backend)
    infos = hb.run()
    # slot 0 is the README episode captured from the reference
    fx = fxs[0]
    e0 = hb.envs[0]
    assert e0.steps == len(fx['t_action']) == 210 and (e0.drone.x, e0.drone.y) == (42, 455)
    assert np.array_equal(e0.drone.map.grid_map, fx['t_dmap'][-1]) and infos[0]['state_machine'] == 1
    # every slot equals its stand-alone run
    for i, p in enumerate(ps):
        row = runner.Experiment(p, device=device, backend=backend).run()
        ex = runner.Experiment.__new__(runner.Experiment)
        ex.params = hb.params[i]
        assert ex.row(infos[i])[12:] == row[12:] or np.allclose(np.array(ex.row(infos[i])[12:], dtype=float),
                                                                 np.array(row[12:], dtype=float), equal_nan=True)


def test_host_plugin_batch_cpu(pkg, oracle):
    _check(pkg, oracle, 'cpu')


@pytest.mark.gpu
def test_host_plugin_batch_gpu(pkg, hip):
    _check(pkg, hip, hip.device)
