#!/usr/bin/env python3
"""Golden-vector generator: runs the REAL reference (imported from /root/reference, with
import-time stubs for the absent gym/pygame/cvxpy) and dumps inputs + expected outputs as
small .npz fixtures next to this script.

Runs ONLY in the build container (the reference is not present on the GPU box).  The .npz files
it writes are data (state vectors, grids, flags); no reference source is stored.

Usage:  python tests/golden/make_golden.py            # regenerate every fixture
"""
import os, sys, types, json, hashlib
import numpy as np

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))


def install_stubs():
    def mk(n):
        m = types.ModuleType(n); sys.modules[n] = m; return m
    mk('pygame')
    cv = mk('cvxpy'); ce = mk('cvxpy.error'); ce.SolverError = Exception; cv.error = ce
    gym = mk('gym'); gym.__path__ = []
    gym.Env = type('Env', (), {})
    gym.logger = type('L', (), {'set_level': staticmethod(lambda x: None)})
    sp = mk('gym.spaces')
    sp.Box = type('Box', (), {'__init__': lambda s, low=None, high=None, shape=None, dtype=np.float32:
                              s.__dict__.update(low=low, high=high, shape=shape, dtype=dtype)})
    sp.Dict = type('Dict', (), {'__init__': lambda s, d: setattr(s, 'spaces', d)})
    gym.spaces = sp
    ge = mk('gym.envs'); ge.__path__ = []
    gr = mk('gym.envs.registration'); gr.register = lambda **k: None
    ge.registration = gr; gym.envs = ge
    sys.path.insert(0, REF); os.chdir(REF)
    import matplotlib; matplotlib.use('Agg')


install_stubs()
import io, contextlib
with contextlib.redirect_stdout(io.StringIO()):
    from utils import Params            # noqa: E402
    from envs.drone_v2 import Drone2DEnv2   # noqa: E402
    import yaw_planner                   # noqa: E402


def make_params(**kw):
    p = Params(debug=True, **kw)
    p.render = False
    return p


def snap_init(env):
    ag = env.agents
    N = len(ag)
    d = dict(
        agent_pos=np.array([a.position for a in ag], dtype=np.float64).reshape(N, 2),
        agent_pref=np.array([a.pref_velocity for a in ag], dtype=np.float64).reshape(N, 2),
        agent_vel=np.array([a.velocity for a in ag], dtype=np.float64).reshape(N, 2),
        agent_radius=np.array([a.radius for a in ag], dtype=np.float64),
        agent_group=np.array([a.group_id for a in ag], dtype=np.int64),
        tracker_radius=np.array([t.radius for t in env.drone.trackers[:max(N, 1)]], dtype=np.float64),
        gt0=env.map_gt.grid_map.copy(),
        dyn_idx0=np.array(env.map_gt.dynamic_idx, dtype=np.int32).reshape(-1, 2),
        obstacles=np.array(env.obstacles, dtype=np.int64).reshape(-1, 3),
        drone0=np.array([env.drone.x, env.drone.y, env.drone.yaw], dtype=np.float64),
    )
    return d


def run_trace(params, T, actions=None, policy=None, teleport=None, stop_on_done=True,
              mutate=None):
    """Step the reference env T times (or until done) recording every input and output."""
    env = Drone2DEnv2(params)
    pol = None
    if policy is not None:
        pol = getattr(yaw_planner, policy)
        pol.__init__(pol, params)
    init = snap_init(env)
    N = len(env.agents)
    rec = {k: [] for k in ['action', 'agent_pos', 'agent_pref', 'gt', 'dmap', 'hit', 'newly',
                           'active_pre', 'active_post', 'drone', 'vel', 'sm', 'fail', 'flags', 'done',
                           'obs_local', 'obs_yaw', 'plan_ok', 'wp_valid', 'wp', 'target', 'tele',
                           'kf_mu', 'kf_sigma', 'kf_len', 'buf_len', 'traj_len', 'steps']}
    # wrap planner.plan to capture its result as seen by step_pos
    plan_cap = {}
    orig_plan = env.planner.plan

    def plan_wrap(drone, dt):
        ok = orig_plan(drone, dt)
        tr = env.planner.trajectory
        plan_cap['ok'] = bool(ok)
        plan_cap['n'] = len(tr)
        if len(tr) > 0:
            plan_cap['wp'] = np.concatenate([np.asarray(tr.positions[0], dtype=np.float64).ravel(),
                                             np.asarray(tr.velocities[0], dtype=np.float64).ravel(),
                                             np.asarray(tr.accelerations[0], dtype=np.float64).ravel()])
        else:
            plan_cap['wp'] = np.zeros(6)
        plan_cap['target'] = np.asarray(env.planner.target, dtype=np.float64).copy()
        return ok
    env.planner.plan = plan_wrap

    for t in range(T):
        if pol is not None:
            a = pol.plan(pol, env.info)
            a = 0.0 if a is None else float(a)
        else:
            a = float(actions[t])
        if teleport is not None:
            pos = teleport(env, t) if callable(teleport) else teleport[t]
            env.drone.x, env.drone.y = pos
            rec['tele'].append(np.array(pos, dtype=np.float64))
        if mutate is not None:
            mutate(env, t)
        active_pre = np.array([tr.active for tr in env.drone.trackers[:N]], dtype=np.uint8)
        tracked_before = env.tracked_agent
        obs, rew, done, info = env.step(a)
        hit = np.zeros(N, dtype=np.uint8)
        for ray in env.drone.rays:
            hit |= ray['hit_list'].numpy().astype(np.uint8)
        rec['action'].append(a)
        rec['agent_pos'].append(np.array([ag.position for ag in env.agents], dtype=np.float64).reshape(N, 2))
        rec['agent_pref'].append(np.array([ag.pref_velocity for ag in env.agents], dtype=np.float64).reshape(N, 2))
        rec['gt'].append(env.map_gt.grid_map.copy())
        rec['dmap'].append(env.drone.map.grid_map.copy())
        rec['hit'].append(hit)
        rec['newly'].append(env.tracked_agent - tracked_before)
        rec['active_pre'].append(active_pre)
        rec['active_post'].append(np.array([tr.active for tr in env.drone.trackers[:N]], dtype=np.uint8))
        rec['drone'].append(np.array([env.drone.x, env.drone.y, float(np.asarray(env.drone.yaw).ravel()[0])], dtype=np.float64))
        rec['vel'].append(np.concatenate([np.asarray(env.drone.velocity, dtype=np.float64).ravel(),
                                          np.asarray(env.drone.acceleration, dtype=np.float64).ravel()]))
        rec['sm'].append(env.state_machine)
        rec['fail'].append(env.fail_count)
        rec['flags'].append(np.array([info['collision_flag'], info['dead_lock_flag'], info['freezing_flag']], dtype=np.uint8))
        rec['done'].append(bool(done))
        rec['obs_local'].append(obs['local_map'][0].copy())
        rec['obs_yaw'].append(obs['yaw_angle'].copy())
        rec['plan_ok'].append(plan_cap['ok'])
        rec['wp_valid'].append(plan_cap['n'] > 0)
        rec['wp'].append(plan_cap['wp'])
        rec['target'].append(plan_cap['target'])
        rec['kf_mu'].append(np.array([tr.mu_upds[-1][:, 0] for tr in env.drone.trackers[:N]], dtype=np.float64).reshape(N, 4))
        rec['kf_sigma'].append(np.array([np.asarray(tr.Sigma_upds[-1], dtype=np.float64) for tr in env.drone.trackers[:N]]).reshape(N, 4, 4))
        rec['kf_len'].append(np.array([len(tr.ts) for tr in env.drone.trackers[:N]], dtype=np.int32))
        rec['buf_len'].append(len(info['tracker_buffer']))
        rec['traj_len'].append(len(env.planner.trajectory))
        rec['steps'].append(env.steps)
        if done and stop_on_done:
            break
    out = dict(init)
    for k, v in rec.items():
        if len(v):
            out['t_' + k] = np.array(v)
    pd = {k: v for k, v in vars(params).items()}
    out['params_json'] = np.array(json.dumps(pd, default=lambda o: list(o) if hasattr(o, '__iter__') else str(o)))
    return out


def save(name, d):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **d)
    print(f'{name}: {os.path.getsize(path)/1024:.1f} KiB, steps={len(d.get("t_action", []))}')


def survivability_cell_sweep(index, position_step=60, T=6.0):
    """The reference's env_metrics (script/difficulty_calculator/glob_survivability_calculator.py:12-42),
    driven verbatim (reset per start cell, pin the drone, step(0), record collision_flag == 2) for a
    shortened horizon T."""
    params = make_params(agent_number=index['agent_number'], agent_radius=index['agent_size'],
                         agent_max_speed=index['agent_speed'], motion_profile='CVM', map_id=index['map_id'],
                         gaze_method='NoControl', planner='NoMove', static_map='maps/empty_map.npy')
    x_range = range(params.map_scale + params.drone_radius, params.map_size[0] - params.map_scale - params.drone_radius, position_step)
    y_range = range(params.map_scale + params.drone_radius, params.map_size[1] - params.map_scale - params.drone_radius, position_step)
    nt = len(np.arange(0, T, 0.1))
    out = np.zeros((len(x_range), len(y_range), nt))
    env = Drone2DEnv2(params)
    for x in x_range:
        for y in y_range:
            env.reset()
            for t in np.arange(0, T, 0.1):
                env.drone.x = x
                env.drone.y = y
                _, _, done, info = env.step(0)
                if info['collision_flag'] == 2:
                    out[int((x - params.map_scale - params.drone_radius) / position_step),
                        int((y - params.map_scale - params.drone_radius) / position_step), int(t / 0.1)] = 1
    return out


def gen_sweep():
    d = {}
    cfgs = [dict(agent_number=10, agent_size=10, agent_speed=40, map_id=0),
            dict(agent_number=10, agent_size=15, agent_speed=60, map_id=3),
            dict(agent_number=20, agent_size=5, agent_speed=20, map_id=1)]
    for i, c in enumerate(cfgs):
        d[f's{i}_cfg'] = np.array(json.dumps(c))
        d[f's{i}_collision_state'] = survivability_cell_sweep(c, T=6.0).astype(np.uint8)
    d['n'] = np.array(len(cfgs))
    save('survivability_sweep', d)


def experiment_row(params, policy_name, actions=None):
    """One episode driven like Experiment.run (experiment.py:65-103) and the values of its CSV row (`actions`: a list
    that receives the policy's output of every step)."""
    from utils import state_machine
    env = Drone2DEnv2(params)
    pol = getattr(yaw_planner, policy_name)
    pol.__init__(pol, params)
    done = False
    while not done:
        a = pol.plan(pol, env.info)
        if actions is not None:
            actions.append(float(a))
        _, _, done, info = env.step(a)
    tracking_time = np.array([len(tr.ts) * 0.1 for tr in info['tracker_buffer']]).sum()
    gm = info['drone'].map.grid_map
    n = len(info['tracker_buffer'])
    return [info['flight_time'], float(gm.shape[0] * gm.shape[1] - np.sum(np.where(gm == 0, 1, 0))), n,
            float(tracking_time / n) if n else float('nan'),
            1 if info['state_machine'] == state_machine['GOAL_REACHED'] else 0,
            1 if info['collision_flag'] == 1 else 0, 1 if info['collision_flag'] == 2 else 0,
            info['freezing_flag'], info['dead_lock_flag'], info['state_machine']]


def gen_rows():
    d = {}
    cases = [('Oxford', dict(gaze_method='Oxford', planner='Primitive', agent_number=10, agent_max_speed=20,
                             agent_radius=15, drone_max_speed=40, map_id=1)),
             ('LookAhead', dict(gaze_method='LookAhead', planner='Primitive', agent_number=30, agent_max_speed=40,
                                agent_radius=10, drone_max_speed=40, map_id=0)),
             ('Rotating', dict(gaze_method='Rotating', planner='Primitive', agent_number=20, agent_max_speed=40,
                               agent_radius=10, drone_max_speed=40, map_id=2))]
    for i, (pol, kw) in enumerate(cases):
        d[f'r{i}_cfg'] = np.array(json.dumps(kw))
        d[f'r{i}_row'] = np.array(experiment_row(make_params(**kw), pol), dtype=np.float64)
    d['n'] = np.array(len(cases))
    save('experiment_rows', d)


def gen_host_gaze():
    """Episodes under the reference's host-side gaze policies LookGoal / LookAhead / Owl (yaw_planner.py:18-39, 225-257,
    151-222) with the Primitive planner: the CSV row and the policy's action of every step."""
    d = {}
    cases = [('LookGoal', dict(gaze_method='LookGoal', planner='Primitive', agent_number=10, agent_max_speed=20,
                               agent_radius=15, drone_max_speed=40, map_id=3)),
             ('LookGoal', dict(gaze_method='LookGoal', planner='Primitive', agent_number=30, agent_max_speed=40,
                               agent_radius=10, drone_max_speed=40, map_id=4)),
             ('LookAhead', dict(gaze_method='LookAhead', planner='Primitive', agent_number=10, agent_max_speed=20,
                                agent_radius=15, drone_max_speed=20, map_id=5)),
             ('Owl', dict(gaze_method='Owl', planner='Primitive', agent_number=10, agent_max_speed=20, agent_radius=15,
                          drone_max_speed=40, map_id=1)),
             ('Owl', dict(gaze_method='Owl', planner='Primitive', agent_number=30, agent_max_speed=40, agent_radius=10,
                          drone_max_speed=40, map_id=6))]
    import warnings
    warnings.simplefilter('ignore')                      # Owl divides by the speed of a drone at rest
    for i, (pol, kw) in enumerate(cases):
        acts = []
        d[f'r{i}_cfg'] = np.array(json.dumps(kw))
        d[f'r{i}_row'] = np.array(experiment_row(make_params(**kw), pol, acts), dtype=np.float64)
        d[f'r{i}_actions'] = np.array(acts, dtype=np.float64)
        print(pol, kw['map_id'], len(acts), 'steps', d[f'r{i}_row'])
    d['n'] = np.array(len(cases))
    save('host_gaze_rows', d)


def gen_flags():
    """Terminal-flag branches of envs/drone_v2.py:222-231: freezing (steps >= max_flight_time / dt) and dead lock
    (10 consecutive planning failures at zero velocity: the target sits inside the border wall)."""
    p = make_params(planner='NoMove', agent_number=3, agent_radius=8, agent_max_speed=10, map_id=21, max_flight_time=2,
                    init_pos=[250, 250])
    save('freezing_nomove', run_trace(p, 25, actions=[0.25] * 25, stop_on_done=False))
    p = make_params(planner='Primitive', gaze_method='LookAhead', agent_number=2, agent_radius=8, agent_max_speed=10, map_id=22,
                    init_pos=[250, 250], target_list=[[4, 4]])
    save('deadlock_primitive', run_trace(p, 14, policy='LookAhead', stop_on_done=False))


def gen_closed():
    """Closed-loop Oxford + Primitive episodes (the planner and the gaze policy are the reference's own objects): pin
    the device plugins -- action, plan result, head waypoint, trajectory length per step."""
    cases = {
        'closed_oxford_n20_map0': dict(agent_number=20, agent_radius=10, agent_max_speed=40, map_id=0),
        'closed_oxford_n20_map5': dict(agent_number=20, agent_radius=10, agent_max_speed=40, map_id=5),
        'closed_oxford_pillars_map2': dict(agent_number=8, agent_radius=12, agent_max_speed=20, map_id=2, pillar_number=8),
        'closed_oxford_pillars_map6': dict(agent_number=15, agent_radius=8, agent_max_speed=30, map_id=6, pillar_number=6),
        'closed_oxford_slow_drone': dict(agent_number=10, agent_radius=15, agent_max_speed=20, map_id=3, drone_max_speed=30),
        'closed_oxford_fast_drone': dict(agent_number=10, agent_radius=10, agent_max_speed=20, map_id=4, drone_max_speed=60),
        'closed_oxford_fov120': dict(agent_number=12, agent_radius=12, agent_max_speed=30, map_id=8, drone_view_range=120,
                                     drone_view_depth=100),
        'closed_oxford_two_targets': dict(agent_number=6, agent_radius=10, agent_max_speed=20, map_id=9,
                                          target_list=[[250, 250], [450, 60]]),
    }
    # the first target lies within the search threshold of the start: the start node is the goal node, Primitive.plan
    # returns True with an EMPTY trajectory (traj_planner.py:158-160, 204-216), the drone stays and the goal test passes
    cases['closed_oxford_goal_at_start'] = dict(agent_number=10, agent_radius=12, agent_max_speed=20, map_id=11,
                                                target_list=[[47, 55], [122, 113]])
    # BASELINE configs 3 and 4 and a non-default map size, closed loop (round 3: until then every closed-loop fixture sat on the
    # 500 x 500 empty map with N <= 20).  Config 3 from the default start (50, 50) dies by collision within a few steps
    # (SURVEY 8d), hence the start below the field of random_map_0's agents.
    cases['closed_oxford_config3'] = dict(agent_number=50, agent_radius=10, agent_max_speed=40, map_id=1, drone_max_speed=40,
                                          static_map='maps/random_map_0.npy', init_pos=[250, 30])
    cases['closed_oxford_config4'] = dict(agent_number=10, agent_radius=15, agent_max_speed=20, map_id=1, drone_max_speed=40,
                                          static_map='maps/obstacle_map.npy')
    cases['closed_oxford_map1000x800'] = dict(agent_number=14, agent_radius=12, agent_max_speed=30, map_id=21, map_size=[1000, 800],
                                              drone_view_range=120, init_pos=[80, 90], target_list=[[900, 700]])
    only = sys.argv[2:] if len(sys.argv) > 2 and sys.argv[1] == 'closed' else None
    for name, kw in cases.items():
        if only and name not in only:
            continue
        p = make_params(gaze_method='Oxford', planner='Primitive', **kw)
        save(name, run_trace(p, 400, policy='Oxford'))


def gen_live(outdir, seed0, count, wide=False, scale20=False):
    """Random closed-loop Oxford + Primitive episodes of the live reference into `outdir` (not committed: the
    build-container-only test tests/test_oracle_vs_live_reference.py replays them through the oracle)."""
    global OUT
    OUT = outdir
    for k in range(count):
        rng = np.random.RandomState(seed0 + k)
        kw = dict(agent_number=int(rng.randint(0, 26)), agent_radius=int(rng.choice([-1, 5, 8, 10, 12, 15, 18])),
                  agent_max_speed=int(rng.choice([4, 10, 20, 30, 40, 60])), drone_max_speed=int(rng.choice([20, 30, 40, 40, 50])),
                  map_id=int(rng.randint(0, 10000)), pillar_number=int(rng.choice([0, 0, 3, 6, 9])),
                  drone_view_range=int(rng.choice([60, 90, 90, 120])), drone_view_depth=int(rng.choice([60, 80, 80, 100])))
        if rng.rand() < 0.25:
            kw['static_map'] = str(rng.choice(['maps/obstacle_map.npy', 'maps/shaped_obstacle_map.npy', 'maps/random_map_0.npy']))   # (random_map_0: round 3)
        if rng.rand() < 0.3:
            kw['target_list'] = [[int(rng.randint(40, 460)), int(rng.randint(40, 460))], [int(rng.randint(40, 460)), int(rng.randint(40, 460))]]
        if rng.rand() < 0.3:
            kw['init_pos'] = [int(rng.randint(40, 460)), int(rng.randint(40, 460))]
        if wide:    # what the family above keeps at its default: map size, drone radius / acceleration / yaw rate, short views,
            #         a first target next to the start (Primitive.plan succeeds with an empty trajectory)
            kw.pop('static_map', None)
            if rng.rand() < 0.5:
                kw['map_size'] = [int(v) for v in rng.choice([500, 600, 700, 800, 1000], 2)]
            kw['drone_radius'] = int(rng.choice([5, 10, 10, 15]))
            kw['drone_max_acceleration'] = int(rng.choice([20, 40, 40, 60]))
            kw['drone_max_yaw_speed'] = int(rng.choice([40, 80, 80, 120]))
            if rng.rand() < 0.5:
                kw['drone_view_depth'] = int(rng.choice([30, 40, 50, 60]))
            if scale20:
                kw['map_scale'] = 20
            if rng.rand() < 0.15:
                x0, y0 = kw.get('init_pos', [50, 50])
                kw['target_list'] = [[x0 + int(rng.randint(-6, 7)), y0 + int(rng.randint(-6, 7))],
                                     [int(rng.randint(40, 460)), int(rng.randint(40, 460))]]
        p = make_params(gaze_method='Oxford', planner='Primitive', **kw)
        save(f'live_oxford_{seed0 + k}', run_trace(p, 120, policy='Oxford'))


def gen_short_view():
    """Short view depths: a 17 x 17 / 21 x 21 local map (L = 4 * (depth // scale) + 1 < 32) and a ray window of a few
    cells -- the geometry the device's row-mapped tile loader mishandled until the random soak found it."""
    rng = np.random.RandomState(777)
    p = make_params(planner='NoMove', agent_number=8, agent_radius=25, agent_max_speed=30, map_id=21)
    p.drone_view_depth = 40
    p.drone_view_range = 120
    T = 90
    tele, cur = [], (250, 250)
    for t in range(T):              # a new pose every sixth step (the drone of a NoMove run stays where it is put)
        if t % 6 == 0:
            cur = (int(rng.randint(15, 485)), int(rng.randint(15, 485)))
        tele.append(cur)
    save('nomove_short_view_d40', run_trace(p, T, actions=rng.uniform(-1, 1, T), teleport=tele, stop_on_done=False))
    p = make_params(gaze_method='Oxford', planner='Primitive', agent_number=10, agent_radius=12, agent_max_speed=20, map_id=22)
    p.drone_view_depth = 50
    save('closed_oxford_short_view_d50', run_trace(p, 300, policy='Oxford'))


def gen_cfg5():
    """BASELINE config 5's geometry through the reference itself: map_size 6400 x 6400 px = 640 x 640 cells, 640 rays
    (rays_number = ceil(map_size[0] / 10), utils.py:570,587), 100 agents.  A handful of NoMove steps with the drone put
    in the open, into a corner next to two border walls, and next to agents (so that rays hit them and trackers start)."""
    p = make_params(planner='NoMove', agent_number=100, agent_radius=15, agent_max_speed=40, map_id=5,
                    map_size=[6400, 6400], init_pos=[3200, 3200], target_list=[[6000, 6000]])

    def tele(env, t):
        if t == 0:
            return (3200, 3200)
        if t == 1:
            return (17, 6381)                      # two border walls inside the view
        a = env.agents[7 if t == 2 else 42].position
        return (int(a[0]) + (10 if t == 2 else -20), int(a[1]) - (45 if t == 2 else 50))   # yaw ~270 looks along +y: the agent is in the cone
    acts = [0.5, -1.0, 0.25, 1.0, 0.0]
    save('nomove_cfg5_640', run_trace(p, len(acts), actions=acts, teleport=tele, stop_on_done=False))


def gen_cfg5_closed():
    """BASELINE config 5's geometry with Oxford + Primitive (round 3: the device plugins took maps of this size from then on): the drone
    starts 130 px from where agent 7 of this seed passes by within the first seconds, so that trackers start and the planner has
    something to avoid.  45 steps of the reference's own closed loop."""
    p = make_params(gaze_method='Oxford', planner='Primitive', agent_number=100, agent_radius=15, agent_max_speed=40, map_id=5,
                    drone_max_speed=40, map_size=[6400, 6400], init_pos=[940, 5000], target_list=[[6000, 6000]])
    save('closed_oxford_cfg5_640', run_trace(p, 45, policy='Oxford'))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'cfg5closed':
        return gen_cfg5_closed()
    if len(sys.argv) > 1 and sys.argv[1] == 'cfg5':
        return gen_cfg5()
    if len(sys.argv) > 1 and sys.argv[1] == 'short_view':
        return gen_short_view()
    if len(sys.argv) > 1 and sys.argv[1] == 'live':
        return gen_live(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), wide=len(sys.argv) > 5 and sys.argv[5] in ('wide', 'wide20'),
                        scale20=len(sys.argv) > 5 and sys.argv[5] == 'wide20')
    if len(sys.argv) > 1 and sys.argv[1] == 'closed':
        return gen_closed()
    if len(sys.argv) > 1 and sys.argv[1] == 'sweep':
        return gen_sweep()
    if len(sys.argv) > 1 and sys.argv[1] == 'rows':
        return gen_rows()
    if len(sys.argv) > 1 and sys.argv[1] == 'host_gaze':
        return gen_host_gaze()
    if len(sys.argv) > 1 and sys.argv[1] == 'flags':
        return gen_flags()
    rng = np.random.RandomState(12345)

    # --- A. NoMove closed loop, BASELINE config-1/2 parameters, constant action (SURVEY section 4 KAT)
    p = make_params(planner='NoMove', agent_number=10, agent_radius=15, agent_max_speed=20, map_id=1)
    save('nomove_n10_const', run_trace(p, 200, actions=[0.5] * 200))

    # --- B. NoMove, random actions, several seeds (short)
    for mid in (0, 2, 3, 7):
        p = make_params(planner='NoMove', agent_number=10, agent_radius=15, agent_max_speed=20, map_id=mid)
        save(f'nomove_n10_rand_map{mid}', run_trace(p, 60, actions=rng.uniform(-1, 1, 60)))

    # --- C. structured yaw (multiples of 4 deg => exact 45-degree rays) with teleports over integer positions
    p = make_params(planner='NoMove', agent_number=10, agent_radius=15, agent_max_speed=20, map_id=4)
    T = 120
    tele = [(int(rng.randint(21, 479)), int(rng.randint(21, 479))) for _ in range(T)]
    acts = rng.choice([-1.0, -0.5, 0.0, 0.5, 1.0, 0.225], size=T)
    save('nomove_teleport_structured', run_trace(p, T, actions=acts, teleport=tele, stop_on_done=False))

    # --- D. survivability-style sweep cell (drone pinned, NoControl => 360 deg view), runs past done
    p = make_params(planner='NoMove', gaze_method='NoControl', agent_number=20, agent_radius=10,
                    agent_max_speed=40, map_id=5)
    p.drone_view_range = 360
    T = 80
    save('surv_pinned_360', run_trace(p, T, actions=[0.0] * T, teleport=[(200, 260)] * T, stop_on_done=False))

    # --- E. static maps => extra radius-5 agents (obstacle_map: +14, shaped: +46, random_map_0: +122)
    p = make_params(planner='NoMove', agent_number=10, agent_radius=15, agent_max_speed=20, map_id=1,
                    static_map='maps/obstacle_map.npy')
    save('nomove_obstacle_map', run_trace(p, 60, actions=rng.uniform(-1, 1, 60), stop_on_done=False))
    p = make_params(planner='NoMove', agent_number=10, agent_radius=10, agent_max_speed=40, map_id=2,
                    static_map='maps/shaped_obstacle_map.npy')
    save('nomove_shaped_map', run_trace(p, 40, actions=rng.uniform(-1, 1, 40), stop_on_done=False))
    p = make_params(planner='NoMove', agent_number=50, agent_radius=10, agent_max_speed=40, map_id=0,
                    static_map='maps/random_map_0.npy')
    save('nomove_random_map_n172', run_trace(p, 25, actions=rng.uniform(-1, 1, 25), stop_on_done=False))

    # --- F. random radii (agent_radius=-1), pillars (static circles), slow agents (stuck rotation branch)
    p = make_params(planner='NoMove', agent_number=15, agent_radius=-1, agent_max_speed=30, map_id=9, pillar_number=4)
    save('nomove_pillars_randr', run_trace(p, 80, actions=rng.uniform(-1, 1, 80), stop_on_done=False))
    p = make_params(planner='NoMove', agent_number=8, agent_radius=12, agent_max_speed=4, map_id=3)
    save('nomove_slow_agents', run_trace(p, 80, actions=rng.uniform(-1, 1, 80), stop_on_done=False))

    # --- G. non-default geometry: 1000x800 px map (100x80 cells, R=100 rays > one wave), depth 120, FOV 120
    p = make_params(planner='NoMove', agent_number=30, agent_radius=12, agent_max_speed=40, map_id=6,
                    map_size=[1000, 800], drone_view_depth=120, drone_view_range=120,
                    init_pos=[300, 400], target_list=[[900, 700]])
    save('nomove_big_map', run_trace(p, 40, actions=rng.uniform(-1, 1, 40), stop_on_done=False))

    # --- H. README config: Oxford gaze + Primitive planner (planner heads + actions recorded as inputs)
    p = make_params(gaze_method='Oxford', planner='Primitive', agent_number=10, agent_max_speed=20,
                    agent_radius=15, drone_max_speed=40, map_id=1)
    save('readme_oxford_primitive', run_trace(p, 800, policy='Oxford'))
    # LookAhead + Primitive on a harder map: exercises brake / replanning / collisions
    for mid in (0, 3):
        p = make_params(gaze_method='LookAhead', planner='Primitive', agent_number=30, agent_max_speed=40,
                        agent_radius=10, drone_max_speed=40, map_id=mid)
        save(f'lookahead_primitive_n30_map{mid}', run_trace(p, 800, policy='LookAhead'))

    # --- I. init-only fixtures (host init restatement): many seeds / settings
    inits = {}
    cfgs = []
    for mid in range(12):
        cfgs.append(dict(agent_number=10, agent_radius=15, agent_max_speed=20, map_id=mid))
    for (n, r, v) in [(10, 5, 20), (20, 10, 40), (30, 15, 60), (30, 5, 60)]:
        cfgs.append(dict(agent_number=n, agent_radius=r, agent_max_speed=v, map_id=11))
    cfgs.append(dict(agent_number=12, agent_radius=-1, agent_max_speed=25, map_id=13, pillar_number=5))
    cfgs.append(dict(agent_number=10, agent_radius=15, agent_max_speed=20, map_id=1, static_map='maps/obstacle_map.npy'))
    cfgs.append(dict(agent_number=5, agent_radius=8, agent_max_speed=20, map_id=2, static_map='maps/shaped_obstacle_map.npy'))
    cfgs.append(dict(agent_number=50, agent_radius=10, agent_max_speed=40, map_id=0, static_map='maps/random_map_0.npy'))
    for i, c in enumerate(cfgs):
        p = make_params(planner='NoMove', **c)
        env = Drone2DEnv2(p)
        s = snap_init(env)
        for k, v in s.items():
            inits[f'c{i}_{k}'] = v
        inits[f'c{i}_cfg'] = np.array(json.dumps(c))
    inits['n_cfg'] = np.array(len(cfgs))
    save('init_cases', inits)

    # --- J. static map label grids as (x, y, label) triplets (data for the package's maps/ directory)
    maps = {}
    for m in ['empty_map', 'obstacle_map', 'shaped_obstacle_map', 'random_map_0']:
        a = np.load(os.path.join(REF, 'maps', m + '.npy'))
        xs, ys = np.nonzero(a)
        maps[m + '_shape'] = np.array(a.shape)
        maps[m + '_xyl'] = np.stack([xs, ys, a[xs, ys]], axis=1).astype(np.int32).reshape(-1, 3)
    save('static_maps', maps)
    gen_sweep()
    gen_rows()
    gen_host_gaze()
    gen_flags()
    gen_closed()


if __name__ == '__main__':
    main()
