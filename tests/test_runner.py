"""Experiment runner (experiment.py surface): the CSV row of whole episodes vs the reference's."""
import json

import numpy as np
import pytest

from replay import load


def _rows(backend, device, which):
    from drone2d_amd import runner
    import drone2d_amd as pkg
    fx = load('experiment_rows')
    for i in which:
        kw = json.loads(str(fx[f'r{i}_cfg']))
        p = pkg.Params(debug=True, **kw)
        p.render = False
        row = runner.Experiment(p, device=device, backend=backend).run()
        got = np.array([float(v) for v in row[12:]], dtype=np.float64)
        want = fx[f'r{i}_row']
        assert np.allclose(got, want, rtol=0, atol=1e-9, equal_nan=True), f'case {i}: {got} vs {want}'


def test_csv_rows_match_reference_cpu(pkg, oracle):
    _rows(oracle, 'cpu', [0, 1, 2])   # row 1 is a LookAhead + Primitive episode (main.py's default pair): this package's
    #                                   host LookAhead on the device planner's state


@pytest.mark.gpu
def test_csv_rows_match_reference_gpu(pkg, hip):
    _rows(hip, hip.device, [0, 1, 2])


def _host_gaze(backend, device):
    """LookGoal / LookAhead (host policies of this package on the device planner's trajectory and the drone's map) against
    reference episodes: the action of every step, bit for bit, and the CSV row."""
    from drone2d_amd import runner, gaze, env as envmod
    import drone2d_amd as pkg
    fx = load('host_gaze_rows')
    for i in range(int(fx['n'])):
        kw = json.loads(str(fx[f'r{i}_cfg']))
        p = pkg.Params(debug=True, **kw)
        p.render = False
        env = envmod.Drone2DEnv2(p, device=device, backend=backend)
        pol = gaze.policy_list[kw['gaze_method']](p)
        assert type(pol).__module__.endswith('gaze')                 # this package's class, not a fallback
        acts, done = [], False
        while not done:
            a = pol.plan(env.info)
            acts.append(float(a))
            _, _, done, _ = env.step(a)
        want = fx[f'r{i}_actions']
        assert len(acts) == len(want) and np.array_equal(np.array(acts), want), \
            f'case {i}: first difference at step {int(np.argmax(np.array(acts)[:len(want)] != want[:len(acts)]))}'
        row = runner.Experiment(p, device=device, backend=backend).run()
        got = np.array([float(v) for v in row[12:]], dtype=np.float64)
        assert np.allclose(got, fx[f'r{i}_row'], rtol=0, atol=1e-9, equal_nan=True), f'case {i}: {got} vs {fx[f"r{i}_row"]}'


def test_host_gaze_policies_match_reference_episodes_cpu(pkg, oracle):
    _host_gaze(oracle, 'cpu')


@pytest.mark.gpu
def test_host_gaze_policies_match_reference_episodes_gpu(pkg, hip):
    _host_gaze(hip, hip.device)


def test_csv_file_written(pkg, oracle, tmp_path):
    from drone2d_amd import runner
    p = pkg.Params(debug=False, gaze_method='Rotating', planner='Primitive', agent_number=20, agent_max_speed=40,
                   agent_radius=10, drone_max_speed=40, map_id=2)
    out = tmp_path / 'results.csv'
    runner.Experiment(p, str(out), backend=oracle).run()
    lines = out.read_text().strip().splitlines()
    assert lines[0].split(',')[0] == 'Method' and len(lines) == 2 and lines[1].startswith('Rotating,Primitive,CVM,2,')


def _batch_rows(pkg, backend, device, B=4, case=0):
    """ExperimentBatch (plugins on the device, one frozen episode per env) vs the reference's row for the first map id and
    vs stand-alone Experiment runs (host plugin objects) for the other map ids.  case 0: Oxford + Primitive, all on the
    device; case 1: LookAhead + Primitive, the gaze actions computed on the host for the whole batch every step."""
    from drone2d_amd import runner
    fx = load('experiment_rows')
    kw = json.loads(str(fx[f'r{case}_cfg']))
    p = pkg.Params(debug=True, **kw)
    p.render = False
    eb = runner.ExperimentBatch(p, B, device=device, backend=backend)
    rows = eb.run()
    assert all(int(d) for d in eb.env.state.flags[:, 3].cpu())                      # every episode ended
    got0 = np.array([float(v) for v in rows[0][12:]], dtype=np.float64)
    assert np.allclose(got0, fx[f'r{case}_row'], rtol=0, atol=1e-9, equal_nan=True), (got0, fx[f'r{case}_row'])
    for e in range(1, B):
        q = pkg.Params(debug=True, **dict(kw, map_id=kw['map_id'] + e))
        q.render = False
        want = runner.Experiment(q, device=device, backend=backend).run()
        a = np.array([float(v) for v in rows[e][12:]], dtype=np.float64)
        b = np.array([float(v) for v in want[12:]], dtype=np.float64)
        assert rows[e][3] == want[3] and np.allclose(a, b, rtol=0, atol=1e-9, equal_nan=True), (e, a, b)
    # a second run() call changes nothing: finished envs stay frozen
    before = eb.env.state.drone.clone()
    eb.env.closed_loop(5, freeze_done=True)
    eb.env.sync()
    assert np.array_equal(before.cpu().numpy(), eb.env.state.drone.cpu().numpy())


def _batch_row_rotating(pkg, backend, device):
    """The reference's Rotating + Primitive row (experiment_rows r2) with the planner on the device and the constant
    gaze policy as a resident action."""
    from drone2d_amd import runner
    fx = load('experiment_rows')
    kw = json.loads(str(fx['r2_cfg']))
    assert kw['gaze_method'] == 'Rotating' and kw['planner'] == 'Primitive'
    p = pkg.Params(debug=True, **kw)
    p.render = False
    rows = runner.ExperimentBatch(p, 2, device=device, backend=backend).run()
    got = np.array([float(v) for v in rows[0][12:]], dtype=np.float64)
    assert np.allclose(got, fx['r2_row'], rtol=0, atol=1e-9, equal_nan=True), (got, fx['r2_row'])


def test_experiment_batch_rotating_cpu(pkg, oracle):
    _batch_row_rotating(pkg, oracle, 'cpu')


@pytest.mark.gpu
def test_experiment_batch_rotating_gpu(pkg, hip):
    _batch_row_rotating(pkg, hip, hip.device)


def test_experiment_batch_rows_cpu(pkg, oracle):
    _batch_rows(pkg, oracle, 'cpu', B=3)


@pytest.mark.gpu
def test_experiment_batch_rows_gpu(pkg, hip):
    _batch_rows(pkg, hip, hip.device, B=6)


def test_experiment_batch_lookahead_rows_cpu(pkg, oracle):
    _batch_rows(pkg, oracle, 'cpu', B=3, case=1)


@pytest.mark.gpu
def test_experiment_batch_lookahead_rows_gpu(pkg, hip):
    _batch_rows(pkg, hip, hip.device, B=5, case=1)
