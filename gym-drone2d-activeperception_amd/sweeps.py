"""Survivability sweeps on the batched device step.

The reference's own large-scale consumer of the pure hot path is
script/difficulty_calculator/glob_survivability_calculator.py: for every setting and every start cell it
resets the env, pins the drone (`env.drone.x = x; env.drone.y = y`), calls `env.step(0)` 240 times and
records `info['collision_flag'] == 2` — 8.3 M sequential env-steps for the published table.  Here every
(setting, start cell) pair is one env of a batch and the T steps are ONE `d2d_rollout` launch.

    collision_state = survivability(index)                 # [len(x_range), len(y_range), T/dt], as env_metrics()
    states = survivability_table(map_ids, agent_numbers, agent_sizes, agent_speeds)   # the np.save()d array
"""
import itertools

import numpy as np
import torch

from .params import Params
from .vec_env import VecDrone2DEnv, build_worlds, build_worlds_of


def _params(index):
    p = Params(agent_number=index['agent_number'], agent_radius=index['agent_size'],
               agent_max_speed=index['agent_speed'], motion_profile=index.get('motion_profile', 'CVM'),
               map_id=index['map_id'], gaze_method='NoControl', planner='NoMove', debug=True,
               static_map='maps/empty_map.npy')
    p.render = False
    return p


def start_cells(params, position_step=60):
    """x_range / y_range of glob_survivability_calculator.py:26-27."""
    lo = params.map_scale + params.drone_radius
    xs = list(range(lo, params.map_size[0] - params.map_scale - params.drone_radius, position_step))
    ys = list(range(lo, params.map_size[1] - params.map_scale - params.drone_radius, position_step))
    return xs, ys


def survivability_batch(indices, position_step=60, T=24, device='cuda:0', backend=None, timings=None, streams=None, worlds=None):
    """Collision states of several settings that share agent_number (same N), one launch.
    Returns float64 [len(indices), len(x_range), len(y_range), n_steps] with 1 where the drone pinned at that
    start cell is in dynamic collision at that step (collision_flag == 2).
    `timings`: a dict that collects, per call, the seconds spent building the worlds on the host (`build_s`), in the T-step
    rollout on the device with everything resident (`device_s`, synchronised on both sides) and in the host post-processing
    (`post_s`), with `env_steps` and `launches` (bench.py --workload survivability).
    `worlds`: the seeded worlds of `indices`, one each, when the caller has built them already (survivability_worlds)."""
    import time
    t_build = time.perf_counter()
    plist = [_params(ix) for ix in indices]
    xs, ys = start_cells(plist[0], position_step)
    cells = [(x, y) for x in xs for y in ys]
    n_steps = len(np.arange(0, T, 0.1))
    seeded = worlds if worlds is not None else [build_worlds(p, 1)[0] for p in plist]
    worlds, pins = [], []
    for w in seeded:
        worlds += [w] * len(cells)             # every start cell begins from the same seeded world (env.reset())
        pins += cells
    env = VecDrone2DEnv(plist[0], len(worlds), device=device, backend=backend, planner='NoMove', worlds=worlds)
    actions = torch.zeros((n_steps, len(worlds)), dtype=torch.float64, device=env.device)
    pin = torch.as_tensor(np.asarray(pins, dtype=np.float64), device=env.device)
    ns = streams if streams is not None else (2 if len(worlds) >= 1024 else 1)
    if timings is not None:
        env.sync()
    t_dev = time.perf_counter()
    coll = env.rollout(actions, pin=pin, collisions=True, streams=ns)
    env.sync()
    t_post = time.perf_counter()
    hit = (coll == 2).T.reshape(len(plist), len(xs), len(ys), n_steps).cpu().numpy()
    # The reference files step k under int(t / 0.1) with t = np.arange(0, T, 0.1)[k] (:38-39).  That is not
    # always k (4.3 / 0.1 == 42.999...), so some steps share a slot and some slots are never written;
    # reproduced so the array is interchangeable with the reference's.
    slot = [int(t / 0.1) for t in np.arange(0, T, 0.1)]
    out = np.zeros(hit.shape, dtype=np.float64)
    for k, sl in enumerate(slot):
        out[..., sl] = np.maximum(out[..., sl], hit[..., k])
    if timings is not None:
        rec = dict(N=int(env.cfg.N), envs=len(worlds), steps=n_steps, streams=ns, build_s=t_dev - t_build, device_s=t_post - t_dev,
                   post_s=time.perf_counter() - t_post, env_steps=len(worlds) * n_steps, launches=n_steps * ns)
        timings.setdefault('batches', []).append(rec)
        for k in ('build_s', 'device_s', 'post_s', 'env_steps', 'launches'):
            timings[k] = timings.get(k, 0) + rec[k]
        timings['last_env'] = env        # (bench.py reads the configuration and the cells per agent of the last batch)
    return out


def survivability(index, position_step=60, T=24, device='cuda:0', backend=None):
    """Drop-in for env_metrics(index) of glob_survivability_calculator.py:12-42."""
    return survivability_batch([index], position_step, T, device, backend)[0]


def _table_order(map_ids, agent_numbers, agent_sizes, agent_speeds):
    return [dict(motion_profile='CVM', pillar_number=0, agent_number=n, agent_speed=v, agent_size=r, map_id=m)
            for m in map_ids for (n, r, v) in itertools.product(agent_numbers, agent_sizes, agent_speeds)]


def survivability_worlds(map_ids=range(20), agent_numbers=(10, 20, 30), agent_sizes=(5, 10, 15), agent_speeds=(20, 40, 60), workers=0):
    """The seeded worlds of survivability_table's settings, in its order: the host's share of the table (pure Python, 1-4 ms each),
    optionally over `workers` forked processes -- like vec_env.build_worlds, only BEFORE the process touches the GPU.  Hand the
    list to survivability_table(worlds=...)."""
    return build_worlds_of([_params(ix) for ix in _table_order(map_ids, agent_numbers, agent_sizes, agent_speeds)], workers=workers)


def survivability_table(map_ids=range(20), agent_numbers=(10, 20, 30), agent_sizes=(5, 10, 15),
                        agent_speeds=(20, 40, 60), position_step=60, T=24, device='cuda:0', backend=None, timings=None, streams=None,
                        worlds=None):
    """The array the reference saves as collision_states_*.npy (glob_survivability_calculator.py:44-57), in its
    loop order: map_id outermost, then product(agent_num, agent_size, agent_vel).  `worlds`: survivability_worlds() of the same
    arguments (else the worlds are built here, one after the other)."""
    order = _table_order(map_ids, agent_numbers, agent_sizes, agent_speeds)
    if worlds is not None and len(worlds) != len(order):
        raise ValueError(f'survivability_table: {len(worlds)} worlds for {len(order)} settings')
    result = [None] * len(order)
    for n in agent_numbers:                    # one batch per agent count (a batch shares N)
        sel = [i for i, ix in enumerate(order) if ix['agent_number'] == n]
        got = survivability_batch([order[i] for i in sel], position_step, T, device, backend, timings, streams,
                                  worlds=None if worlds is None else [worlds[i] for i in sel])
        for i, g in zip(sel, got):
            result[i] = g
    return np.array(result)
