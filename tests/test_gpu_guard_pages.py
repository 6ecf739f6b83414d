"""Bounds of the per-env buffers, the way a GPU without an address sanitizer can check them (run with -m gpu): the batch is sized
so that ONE buffer ends exactly on the last byte of its device allocation (a multiple of 2 MiB pages, an allocation of its own),
where a read or write past the end of the last env's record leaves mapped memory and faults instead of landing silently in a
neighbour.  Round 3's out-of-bounds read of the tracker records (idle lanes reading elements 20..23 of the last tracker of the
last env) was found by a profile run at exactly such a size; this test makes those sizes on purpose, for every per-env buffer in
turn.  Each case runs the stepped path (k_stages / k_gaze / k_plan launches) and the persistent closed loop, and the batch must
equal the 4-env run of its worlds."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

PAGE = 2 << 20
ENV_FIELDS = ('agents', 'dyn_prev', 'gt', 'dmap', 'drone', 'target', 'counters', 'active', 'kf', 'kf_len', 'hit', 'newly', 'flags',
              'obs_local', 'obs_yaw', 'action', 'wp')
PLUGIN_FIELDS = ('traj_hdr', 'traj', 'traj_box', 'trk_radius', 'trk_prev', 'seen_step', 'plan_stat')

DEFAULT = dict(agent_number=10, agent_radius=15, agent_max_speed=20)
CASES = [
    # (buffer, where, Params overrides)
    ('agents', 'state', dict(agent_number=32, agent_radius=8, agent_max_speed=30)),      # SPEC 2 kernels (17..40 agents)
    ('dyn_prev', 'state', dict(agent_number=32, agent_radius=8, agent_max_speed=30)),
    ('kf', 'state', dict(agent_number=32, agent_radius=8, agent_max_speed=30)),
    ('kf', 'state', DEFAULT),                                                             # SPEC 1 kernels (<= 16 agents)
    ('kf', 'state', dict(agent_number=50, agent_radius=10, agent_max_speed=40, static_map='maps/random_map_0.npy')),   # SPEC 3 (172 agents)
    ('agents', 'state', dict(agent_number=50, agent_radius=10, agent_max_speed=40, static_map='maps/random_map_0.npy')),
    ('kf_len', 'state', dict(agent_number=32, agent_radius=8, agent_max_speed=30)),
    ('nodes', 'plugins', DEFAULT),
    ('hash', 'plugins', DEFAULT),
    ('traj', 'plugins', DEFAULT),
    ('traj_box', 'plugins', DEFAULT),                                                     # 992 B per env: 65536 envs
    ('seen_step', 'plugins', dict(agent_number=10, agent_radius=15, agent_max_speed=20, map_size=[640, 320], init_pos=[50, 50],
                                  target_list=[[580, 260]])),                             # 64 x 32 cells: the generic kernels
    ('gt', 'state', dict(agent_number=10, agent_radius=15, agent_max_speed=20, map_size=[640, 320], init_pos=[50, 50],
                         target_list=[[580, 260]])),
    ('dmap', 'state', dict(agent_number=10, agent_radius=15, agent_max_speed=20, map_size=[640, 320], init_pos=[50, 50],
                           target_list=[[580, 260]])),
]


def _ends_its_allocation(t):
    """Does tensor `t` end on the last byte of a device segment of the caching allocator (a hipMalloc of its own)?"""
    end = t.data_ptr() + t.numel() * t.element_size()
    for seg in torch.cuda.memory_snapshot():
        if seg['address'] <= t.data_ptr() < seg['address'] + seg['total_size']:
            return end == seg['address'] + seg['total_size']
    return False


@pytest.mark.parametrize('buf,where,kw', CASES, ids=[f"{b}_N{k['agent_number']}{'_map' if 'map_size' in k else ''}{'_r0' if 'static_map' in k else ''}" for b, _, k in CASES])
def test_buffer_ending_on_a_page_boundary(pkg, hip, buf, where, kw):
    from drone2d_amd import vec_env, host_init
    import copy
    p = pkg.Params(planner='Primitive', gaze_method='Oxford', drone_max_speed=40, map_id=11, **kw)
    worlds = []
    for i in range(4):
        q = copy.copy(p)
        q.map_id = 11 + i
        worlds.append(host_init.init_world(pkg.with_defaults(q)))
    mk = lambda n, w: vec_env.VecDrone2DEnv(p, n, backend=hip, planner='Primitive', device_plugins=True, gaze='Oxford', worlds=w)
    small = mk(4, worlds)
    t4 = (small.state.t if where == 'state' else small.plugins.t)[buf]
    per_env = t4.numel() * t4.element_size() // 4
    b0 = PAGE // math.gcd(per_env, PAGE)                  # envs per whole number of pages
    b0 = b0 * 4 // math.gcd(b0, 4)                        # ... and a whole number of replicas of the 4 worlds
    B = b0 * max(1, -(-(24 << 20) // (per_env * b0)))     # >= 24 MiB: more than the remainder of any of the caching allocator's 20 MiB segments
    if B > 200000:
        pytest.skip(f'{buf}: {per_env} B per env needs {B} envs for whole pages')
    big = mk(B, [worlds[i % 4] for i in range(B)])
    # the buffer moves into an allocation of its own: with no cached block to carve it from, a tensor of whole pages (>= 10 MiB) is
    # one fresh device allocation of exactly its size, so its last byte is the allocation's last byte
    holder = big.state.t if where == 'state' else big.plugins.t
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    plugs = []                                            # (free remainders of partly used segments are plugged first)
    for _ in range(32):
        tb = torch.empty_like(holder[buf])
        if _ends_its_allocation(tb):
            break
        plugs.append(tb)
    tb.copy_(holder[buf])
    holder[buf] = tb
    big._st, big._plan = big.state.struct(), big.plugins.struct()
    nbytes = tb.numel() * tb.element_size()
    assert nbytes == per_env * B and nbytes % PAGE == 0
    assert _ends_its_allocation(tb), f'{buf}: the allocator placed the buffer inside a larger segment'
    # drones next to an agent of their world, looking at it: rays hit, trackers start, the planner has something to avoid
    ag = small.state.agents
    pin = torch.stack([ag[:, 0, 3].floor() + 8.0, ag[:, 1, 3].floor() - 40.0], dim=1)
    pin[:, 0].clamp_(30.0, small.cfg.W_px - 30.0)
    pin[:, 1].clamp_(30.0, small.cfg.H_px - 30.0)
    for env, reps in ((big, B // 4), (small, 1)):
        env.state.drone[:, :2] = pin.repeat(reps, 1)
    for t in range(4):                                    # the stepped path: gaze / perceive / plan / act as separate launches
        for env in (big, small):
            env.backend.gaze_stage(env.cfg, env._st, env._plan)
            env.perceive()
            env.backend.plan_stage(env.cfg, env._st, env._plan)
            env.backend.act(env.cfg, env._st)
    a4 = torch.tensor([0.3, -0.2, 0.1, -0.4], dtype=torch.float64)
    for env, reps in ((big, B // 4), (small, 1)):         # and one fused step (k_stages with every stage)
        env.step(a4.repeat(reps))
    for _ in range(2):                                    # the persistent loop
        big.closed_loop(5, auto_reset=True)
        small.closed_loop(5, auto_reset=True)
    big.sync()
    small.sync()
    for src, names in (('state', ENV_FIELDS), ('plugins', PLUGIN_FIELDS)):
        for name in names:
            x = (big.state.t if src == 'state' else big.plugins.t)[name]
            y = (small.state.t if src == 'state' else small.plugins.t)[name]
            for c0 in range(0, B, 8192):
                xs = x[c0:c0 + 8192]
                assert bool((xs.view(xs.shape[0] // 4, 4, *x.shape[1:]) == y.unsqueeze(0)).all()), f'{buf}: {name} differs (envs {c0}..)'
    assert int(small.state.counters[:, pkg._abi.C_STEPS].sum()) > 0
    del big
    torch.cuda.empty_cache()
