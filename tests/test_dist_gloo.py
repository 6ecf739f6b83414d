"""Multi-process path on CPU: world_size 2 over gloo.  Each rank steps its shard (oracle backend standing in
for the HIP library), the shards gather episode statistics, and the result equals the single-process
run env for env."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOTAL, STEPS = 7, 25      # odd on purpose: shards of 4 and 3 envs


def _actions():
    return np.random.RandomState(42).uniform(-1, 1, (STEPS, TOTAL))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    import drone2d_amd as pkg
    from drone2d_amd import dist as d2dist
    from oracle_lib import OracleBackend
    d2dist.init_process_group('gloo')
    p = pkg.Params(planner='NoMove', agent_number=10, agent_radius=15, agent_max_speed=40, map_id=3)
    env = d2dist.make_shard(p, TOTAL, device='cpu', backend=OracleBackend())
    lo, hi = d2dist.shard_range(TOTAL, rank, world)
    assert env.num_envs == hi - lo and env.env_offset == lo
    acts = _actions()
    for t in range(STEPS):
        env.step(acts[t, lo:hi])
    stats = d2dist.gather_episode_stats(env, TOTAL)
    q.put((rank, stats.numpy(), env.state.drone.numpy().copy()))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def _worker_closed(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    import drone2d_amd as pkg
    from drone2d_amd import dist as d2dist
    from oracle_lib import OracleBackend
    d2dist.init_process_group('gloo')
    env = d2dist.make_shard(_closed_params(pkg), 5, device='cpu', backend=OracleBackend(), planner='Primitive',
                            device_plugins=True, gaze='Oxford')
    env.closed_loop(CLOSED_STEPS, auto_reset=True)
    stats = d2dist.gather_episode_stats(env, 5)
    q.put((rank, stats.numpy(), env.state.drone.numpy().copy(), env.plugins.t['traj_hdr'].numpy().copy()))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


CLOSED_STEPS = 230      # past the end of the first episodes: the shards also restart them identically


def _closed_params(pkg):
    return pkg.Params(planner='Primitive', gaze_method='Oxford', agent_number=10, agent_radius=15, agent_max_speed=20,
                      drone_max_speed=40, map_id=1)


def test_closed_loop_shards_equal_one_process(pkg, oracle):
    """The closed loop with the plugins on the device (oracle standing in for the HIP library), 5 envs over 2 ranks:
    env identity is the global env id, so the shards reproduce the single-process batch env for env."""
    from drone2d_amd import vec_env
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29850 + os.getpid() % 100
    procs = [ctx.Process(target=_worker_closed, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    got = {}
    for _ in range(2):
        r, stats, drone, hdr = q.get(timeout=300)
        got[r] = (stats, drone, hdr)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    ref = vec_env.VecDrone2DEnv(_closed_params(pkg), 5, backend=oracle, planner='Primitive', device_plugins=True, gaze='Oxford')
    ref.closed_loop(CLOSED_STEPS, auto_reset=True)
    want = ref.episode_stats().numpy()
    assert np.array_equal(got[0][0], want) and np.array_equal(got[1][0], want)
    assert np.array_equal(np.concatenate([got[0][1], got[1][1]]), ref.state.drone.numpy())
    assert np.array_equal(np.concatenate([got[0][2], got[1][2]]), ref.plugins.t['traj_hdr'].numpy())


def test_two_shards_equal_one_process(pkg, oracle):
    from drone2d_amd import vec_env, dist as d2dist
    assert d2dist.shard_range(7, 0, 2) == (0, 4) and d2dist.shard_range(7, 1, 2) == (4, 7)
    assert [d2dist.shard_range(262144, r, 8) for r in (0, 7)] == [(0, 32768), (229376, 262144)]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29600 + os.getpid() % 200
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    got = {}
    for _ in range(2):
        r, stats, drone = q.get(timeout=180)
        got[r] = (stats, drone)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    p = pkg.Params(planner='NoMove', agent_number=10, agent_radius=15, agent_max_speed=40, map_id=3)
    ref = vec_env.VecDrone2DEnv(p, TOTAL, backend=oracle)
    acts = _actions()
    for t in range(STEPS):
        ref.step(acts[t])
    want = ref.episode_stats().numpy()
    assert np.array_equal(got[0][0], want) and np.array_equal(got[1][0], want)      # every rank holds the full table
    assert np.array_equal(np.concatenate([got[0][1], got[1][1]]), ref.state.drone.numpy())


def test_bench_two_rank_dry_run():
    """bench.py itself, 2 ranks over gloo on the CPU (tests/bench_dry_rank.py puts the oracle behind the package's
    backend): the launch contract of the driver (torch.distributed.run, RANK / WORLD_SIZE from the env, ONE JSON line
    from rank 0), shards by rank, the max-over-ranks clock, the all_gather of episode statistics."""
    import json
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'tests', 'bench_dry_rank.py'), '--gpus', '2', '--steps', '6',
           '--warmup', '2', '--prologue', '4', '--envs', '5', '--workers', '0', '--dist-backend', 'gloo', '--single-device',
           '--leg', 'closed', '--no-cpu-baseline']
    env = dict(os.environ, OMP_NUM_THREADS='1')
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['n_ranks_seen'] == 2 and j['steps'] == 6 and j['warmup'] == 2 and j['scaling'] == 'weak'
    assert j['episode_stats']['envs'] == 10 and j['gather_ms'] is not None and j['value'] > 0
    assert abs(j['value'] - 2 * 5 * 6 / (j['ms_per_step'] * 6e-3)) < 1e-6 * j['value']
    assert j['roofline']['traffic'] is None and 'no PMC pass' in j['roofline']['traffic_source']     # 5 envs: no such profile
    assert j['config']['timed_window']['searches_per_env_per_step'] >= 0
    # the line's contract (keys the driver and the judge read)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline'):
        assert k in j, k
    assert j['unit'] == 'env-steps/s' and j['higher_is_better'] is True and j['vs_baseline'] is None and j['dtype'] == 'f64'
    assert j['data'] == 'synthetic' and 'workload' in j['config'] and 'model' not in j['config']
    r = j['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12
    assert r['algo_bytes_per_env_step'] == 3844.0          # SURVEY 8(d) for config 2, from the run's own N / R / L / cells per agent


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT torch.distributed.run (the way the driver calls N = 1): bench.py starts the two ranks
    itself, relays rank 0's line and checks n_gpus == n_ranks_seen == 2.  (CPU dry run through tests/bench_dry_rank.py.)"""
    import json
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, 'tests', 'bench_dry_rank.py'), '--gpus', '2', '--steps', '4', '--warmup', '2',
           '--prologue', '2', '--envs', '4', '--workers', '0', '--dist-backend', 'gloo', '--single-device', '--leg', 'closed',
           '--no-cpu-baseline']
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env['OMP_NUM_THREADS'] = '1'
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['n_ranks_seen'] == 2 and j['episode_stats']['envs'] == 8 and j['value'] > 0


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    """--gpus 2 under a launcher that started ONE rank would print a 1-rank number labelled n_gpus 2: refused."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0', OMP_NUM_THREADS='1')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'bench_dry_rank.py'), '--gpus', '2', '--steps', '2',
                          '--warmup', '1', '--envs', '2', '--workers', '0', '--leg', 'closed', '--no-cpu-baseline'],
                         capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode != 0 and 'WORLD_SIZE' in out.stderr


def test_bench_parent_returns_when_one_rank_dies_early(tmp_path):
    """A rank other than 0 that dies before its first collective must not leave rank 0 -- and the parent -- waiting for the
    collective's timeout: the parent polls every child, stops the others and fails at once."""
    import subprocess
    import time
    entry = tmp_path / 'rank.py'
    entry.write_text('import os, sys, time\n'
                     'if os.environ["RANK"] == "1":\n'
                     '    sys.exit(7)\n'
                     'time.sleep(600)\n')
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env['D2D_BENCH_ENTRY'] = str(entry)
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], capture_output=True, text=True, timeout=120,
                         env=env, cwd=ROOT)
    assert out.returncode == 1 and time.time() - t0 < 60, (out.returncode, out.stderr[-500:])
    assert 'rank exit codes' in out.stderr


def test_bench_survivability_workload_dry_run():
    """`bench.py --workload survivability` (the reference's survivability table end to end) on the CPU harness: one
    map, the line carries `roofline` per agent count and a cpu_baseline whose table equals the timed one."""
    import json
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, 'tests', 'bench_dry_rank.py'), '--workload', 'survivability', '--maps', '1']
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    j = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert j['config']['table_shape'] == [27, 8, 8, 240] and j['value'] > 0 and j['cpu_baseline']['table_of_map_0_equals_oracle']
    assert [r['agents'] for r in j['roofline']['per_agent_count']] == [10, 20, 30] and j['roofline']['frac'] > 0
