"""SURVEY 8(e) on real hardware: the HIP backend in MORE THAN ONE PROCESS (run with -m gpu).  A 1-GPU box has one card, so both
ranks use cuda:0 and exchange their episode statistics over gloo; what is exercised is everything else of the multi-rank path --
sharding by global env id, one HIP context per process, the persistent closed loop of two processes side by side on one card, the
gather -- and the result must equal the one-process HIP run env for env.  The ranks are fresh child processes (subprocess): no
process that has initialised the GPU is ever replaced by another program."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
TOTAL, STEPS = 8, 40


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_hip_processes_equal_one(pkg, hip, tmp_path):
    from drone2d_amd import vec_env
    import dist_hip_rank
    port = _free_port()
    procs, outs = [], []
    for r in range(2):
        out = str(tmp_path / f'rank{r}.npz')
        outs.append(out)
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', LOCAL_WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'dist_hip_rank.py'), str(TOTAL), str(STEPS), out], env=env))
    try:
        codes = [p.wait(timeout=240) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert codes == [0, 0], codes
    got = [np.load(o) for o in outs]
    assert [(int(g['lo']), int(g['hi'])) for g in got] == [(0, 4), (4, 8)]
    ref = vec_env.VecDrone2DEnv(dist_hip_rank.params(pkg), TOTAL, backend=hip, planner='Primitive', device_plugins=True, gaze='Oxford')
    for _ in range(2):
        ref.closed_loop(STEPS // 2, auto_reset=True)
    ref.sync()
    want = ref.episode_stats().cpu().numpy()
    for g in got:                                       # every rank holds the full table, ordered by global env id
        assert np.array_equal(g['stats'], want)
    for k in ('drone', 'counters', 'agents', 'dmap', 'gt', 'kf', 'flags', 'action'):
        assert np.array_equal(np.concatenate([got[0][k], got[1][k]]), ref.state.t[k].cpu().numpy()), k
    for k in ('traj_hdr', 'seen_step'):
        assert np.array_equal(np.concatenate([got[0][k], got[1][k]]), ref.plugins.t[k].cpu().numpy()), k
    assert int(ref.state.counters[:, pkg._abi.C_STEPS].sum()) > 0


def test_bench_starts_two_hip_ranks_on_one_card(hip):
    """bench.py --gpus 2 with no launcher on a 1-GPU box (--single-device --dist-backend gloo): its own two ranks, the barrier /
    max-over-ranks timing, the gather, and a line that reports both ranks."""
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--single-device', '--dist-backend', 'gloo', '--envs', '256',
           '--steps', '20', '--warmup', '5', '--prologue', '20', '--leg', 'closed', '--large', '0', '--no-cpu-baseline', '--workers', '0']
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['n_ranks_seen'] == 2 and line['value'] > 0 and line['episode_stats']['envs'] == 512
