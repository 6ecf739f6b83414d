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
    _rows(oracle, 'cpu', range(3))


@pytest.mark.gpu
def test_csv_rows_match_reference_gpu(pkg, hip):
    _rows(hip, hip.device, [0, 2])


def test_csv_file_written(pkg, oracle, tmp_path):
    from drone2d_amd import runner
    p = pkg.Params(debug=False, gaze_method='Rotating', planner='Primitive', agent_number=20, agent_max_speed=40,
                   agent_radius=10, drone_max_speed=40, map_id=2)
    out = tmp_path / 'results.csv'
    runner.Experiment(p, str(out), backend=oracle).run()
    lines = out.read_text().strip().splitlines()
    assert lines[0].split(',')[0] == 'Method' and len(lines) == 2 and lines[1].startswith('Rotating,Primitive,CVM,2,')
