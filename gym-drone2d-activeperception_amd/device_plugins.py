"""Host side of the device planner / gaze plugins (include/d2d.h `d2d_plan`).

The device replays the reference's `Primitive` planner (traj_planner.py:78-233) and `Oxford` gaze policy
(yaw_planner.py:41-127) operation for operation.  Every constant those classes derive with numpy / Python
float arithmetic is evaluated HERE, with the very expressions the reference uses, and handed over as small
tables: the device never has to guess how `np.arange`, `t ** 2` or `np.arccos` round.

`PluginState` owns the tables, the per-env plugin state (trajectory, tracker radii, Oxford's seen map) and
the search scratch as torch tensors on the batch's device, and builds the ctypes `d2d_plan` over them.
"""
import math
import struct

import numpy as np
import torch

from . import _abi as A


# ---------------------------------------------------------------------------------------------------
# np.arccos(q) <= half_fov as an exact decision window
# ---------------------------------------------------------------------------------------------------
def _key(x):
    """Monotone integer image of a double (order of the keys == order of the doubles)."""
    b = struct.unpack('<q', struct.pack('<d', float(x)))[0]
    return b ^ ((b >> 63) & 0x7FFFFFFFFFFFFFFF)


def _unkey(k):
    b = k ^ ((k >> 63) & 0x7FFFFFFFFFFFFFFF)
    return struct.unpack('<d', struct.pack('<q', b))[0]


def _arccos_le(q, half):
    """The reference's own expression (yaw_planner.py:77) on an array laid out in full SIMD vectors."""
    q = np.asarray(q, dtype=np.float64)
    rep = np.repeat(q.reshape(-1, 1), 8, axis=1)          # every value fills one 8-lane vector
    with np.errstate(invalid='ignore'):
        return (np.arccos(rep) <= half)[:, 0]


def acos_window(half_fov):
    """(key_lo, mask): decisions of `np.arccos(q) <= half_fov` for the 64 consecutive doubles from the one with
    ordered key `key_lo`; above the window every q <= 1 is inside the cone, below it every q is outside.  Raises
    when that rule does not describe this host's arccos (view ranges whose edge falls where doubles are much
    denser than arccos' own spacing, e.g. exactly 180 degrees)."""
    centre = math.cos(min(half_fov, math.pi))
    k0 = _key(centre) - 32
    qs = np.array([_unkey(k0 + i) for i in range(64)])
    dec = _arccos_le(qs, half_fov)
    mask = 0
    for i, d in enumerate(dec):
        if d:
            mask |= 1 << i
    # validate the rule outside the window on this host
    rng = np.random.RandomState(7)
    probe = np.concatenate([[_unkey(k0 - 1 - i) for i in range(200)], [_unkey(k0 + 64 + i) for i in range(200)],
                            rng.uniform(-1, 1, 20000), centre + rng.uniform(-1e-9, 1e-9, 20000), [1.0, -1.0, 0.0]])
    probe = probe[np.abs(probe) <= 1.0]
    want = _arccos_le(probe, half_fov)
    keys = np.array([_key(v) for v in probe])
    rule = np.where(keys >= k0 + 64, True, np.where(keys < k0, False, False))
    inside = (keys >= k0) & (keys < k0 + 64)
    rule[inside] = [(mask >> int(k - k0)) & 1 == 1 for k in keys[inside]]
    if not np.array_equal(rule, want):
        raise NotImplementedError(f'drone_view_range with half angle {half_fov!r} rad: the edge of the view cone cannot be '
                                  'expressed as a 64-double decision window of this host\'s arccos')
    return k0, mask


# ---------------------------------------------------------------------------------------------------
# np.sum over a contiguous float64 array: numpy's pairwise summation as a block list + an add program
# ---------------------------------------------------------------------------------------------------
def pairwise_plan(n):
    """numpy/_core/src/umath/loops_utils.h.src `pairwise_sum`: blocks of at most 128 elements (8 strided
    accumulators each), split at multiples of 8, partial sums added left + right up the recursion."""
    leaves, prog = [], []

    def rec(off, m):
        if m <= 128:
            leaves.append((off, m))
            prog.append(len(leaves) - 1)
        else:
            m2 = m // 2
            m2 -= m2 % 8
            rec(off, m2)
            rec(off + m2, m - m2)
            prog.append(-1)
    rec(0, n)
    return np.array(leaves, dtype=np.int32).reshape(-1, 2), np.array(prog, dtype=np.int32)


def pairwise_levels(n_leaf, prog):
    """The add program as a tree evaluated level by level: node ids 0..n_leaf-1 are the blocks, every `-1` of the
    program creates the next id = left + right.  Returns (ops [n_leaf - 1][3] = dst, left, right ordered by level,
    level_start [n_levels + 1]): all ops of one level are independent, so the device adds them in parallel while
    every single sum keeps numpy's operand order."""
    st, nodes, height = [], [], {}
    nxt = n_leaf
    for op in prog:
        if op >= 0:
            st.append(int(op))
            height[int(op)] = 0
        else:
            b = st.pop()
            a = st.pop()
            height[nxt] = max(height[a], height[b]) + 1
            nodes.append((height[nxt], nxt, a, b))
            st.append(nxt)
            nxt += 1
    nodes.sort(key=lambda t: (t[0], t[1]))
    ops = np.array([[d, a, b] for _, d, a, b in nodes], dtype=np.int32).reshape(-1, 3)
    nlev = max([h for h, *_ in nodes], default=0)
    start = [0]
    for lv in range(1, nlev + 1):
        start.append(start[-1] + sum(1 for h, *_ in nodes if h == lv))
    return ops, np.array(start, dtype=np.int32), (st[0] if st else 0)


def pairwise_sum_host(a, leaves, prog):
    """Reference implementation of the plan (tests check it against np.sum)."""
    def leaf(off, m):
        v = a[off:off + m]
        if m < 8:
            r = 0.0
            for x in v:
                r += x
            return r
        r = [v[j] for j in range(8)]
        i = 8
        while i < m - (m % 8):
            for j in range(8):
                r[j] += v[i + j]
            i += 8
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        while i < m:
            res += v[i]
            i += 1
        return res
    st = []
    for op in prog:
        if op >= 0:
            st.append(leaf(*leaves[op]))
        else:
            b = st.pop()
            st[-1] = st[-1] + b
    return st[0]


# ---------------------------------------------------------------------------------------------------
# tables
# ---------------------------------------------------------------------------------------------------
def sq_threshold(L):
    """The largest double s with sqrt(s) <= L (math.sqrt is IEEE's correctly rounded square root, hence monotone): `norm(d) <= L` with
    numpy's norm = sqrt(d.d) is exactly `d.d <= sq_threshold(L)`."""
    if not L >= 0.0:
        return -1.0
    t = L * L
    while math.sqrt(t) > L:
        t = math.nextafter(t, -math.inf)
    while math.sqrt(math.nextafter(t, math.inf)) <= L:
        t = math.nextafter(t, math.inf)
    return t


def build_tables(params, cfg, need_acos=True):
    """Scalars + numpy tables of `d2d_plan`, each computed as the reference computes it.  `need_acos`: the arccos decision
    window of the Oxford stage (view ranges it cannot describe only matter when that stage runs)."""
    p = params
    t = {}
    # Primitive.__init__, traj_planner.py:95-107
    if p.drone_max_speed <= 40:
        u_space = np.arange(-p.drone_max_acceleration, p.drone_max_acceleration, 0.4 * p.drone_max_speed - 5)
    else:
        u_space = np.arange(-p.drone_max_acceleration, p.drone_max_acceleration, 4)
    horizon = 2
    sample_num = p.drone_max_speed * horizon // p.map_scale
    ts_check = np.arange(0, horizon, horizon / sample_num)                       # :175
    ts_traj = np.arange(horizon, 0, -p.dt)[::-1]                                 # :212, reversed by :215-216
    t['u_space'] = np.asarray(u_space, dtype=np.float64)
    t['sample_t'] = np.array([[tt, tt ** 2] for tt in ts_check], dtype=np.float64).reshape(-1, 2)      # [1, t, t**2], :176
    t['traj_t'] = np.array([[tt, tt ** 2, 2 * tt] for tt in ts_traj], dtype=np.float64).reshape(-1, 3)  # :121-122
    # Oxford.__init__, yaw_planner.py:46-65
    w = p.drone_max_yaw_speed
    t['yaw_space'] = np.asarray(np.arange(-w, w, w / 3), dtype=np.float64)
    n_calls = int(math.ceil(p.max_flight_time / p.dt)) + 8
    tab = np.zeros((2, n_calls), dtype=np.float64)
    a0, a5 = 0.0, 5.0                                                            # :48 (5 * ones), :95-97
    for k in range(n_calls):
        tab[0, k], tab[1, k] = a0, a5
        a0 = a0 + (1 - 0) * p.dt
        a5 = a5 + (1 - 0) * p.dt
    t['tobs_tab'] = tab
    leaves, prog = pairwise_plan(cfg.W * cfg.H)
    ops, lvl_start, root = pairwise_levels(len(leaves), prog)
    # per block: offset, length, first and last grid row it covers (the device walks rows)
    t['pw_leaf'] = np.array([[o, m, o // cfg.H, (o + m - 1) // cfg.H] for o, m in leaves], dtype=np.int32).reshape(-1, 4)
    t['pw_prog'] = prog
    # the same additions level by level: [n_levels, root, level_start[0..n_levels], then (dst, left, right) per op]
    t['pw_tree'] = np.concatenate([[len(lvl_start) - 1, root], lvl_start, ops.ravel()]).astype(np.int32)
    # first block of every grid row (the blocks a view box can touch are a contiguous range)
    offs = leaves[:, 0]
    t['pw_rowleaf'] = np.array([int(np.searchsorted(offs, i * cfg.H, side='right') - 1) for i in range(cfg.W)], dtype=np.int32)
    half = math.radians(p.drone_view_range / 2)                                  # :72
    key_lo, mask = acos_window(half) if need_acos else (0, 0)
    # successors of one expansion satisfy |v + 2 a| < vmax: at most this many lattice points of u_space
    step = float(u_space[1] - u_space[0]) if len(u_space) > 1 else 1.0
    side = int(2 * p.drone_max_speed / (horizon * step)) + 2
    succ_max = min(side * side, len(u_space) ** 2)
    max_itr = 100
    node_cap = 1 + (max_itr - 1) * succ_max
    hash_cap = 1 << int(math.ceil(math.log2(2 * node_cap + 2)))
    sc = dict(nu=len(u_space), n_sample=len(ts_check), n_ts=len(ts_traj), max_itr=max_itr,
              traj_cap=(max_itr - 1) * len(ts_traj), node_cap=node_cap, hash_cap=hash_cap, n_yaw=len(t['yaw_space']),
              pw_nleaf=len(leaves), pw_nprog=len(prog), tobs_len=n_calls, pw_ntree=len(t['pw_tree']),
              horizon=float(horizon), vmax=float(p.drone_max_speed), safe_dist=float(p.drone_radius + 10),
              goal_tol=10.0, agent_radius=float(p.agent_radius), half_fov=half, yaw_rate_max=float(w),
              vmax_sq=sq_threshold(math.nextafter(float(p.drone_max_speed), -math.inf)) if p.drone_max_speed > 0 else -1.0,   # norm < vmax
              goal_sq=sq_threshold(10.0),
              acos_key_lo=key_lo, acos_mask=mask)
    return sc, t


class PluginState:
    """Tables + per-env plugin state + scratch of one batch, and the ctypes `d2d_plan` over them."""

    def __init__(self, params, cfg, device, tracker_radius, planner='Primitive', gaze='Oxford', tables=None):
        self.cfg = cfg
        self.device = torch.device(device)
        sc, tb = tables if tables is not None else build_tables(params, cfg, need_acos=(gaze == 'Oxford'))
        self.scalars, self.tables_np = sc, tb
        self.planner = A.PLAN_PRIMITIVE if planner == 'Primitive' else A.PLAN_NONE
        self.gaze = A.GAZE_OXFORD if gaze == 'Oxford' else A.GAZE_NONE
        B, N = cfg.B, cfg.N
        dev = self.device
        self.tables = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in tb.items()}
        f64, i32, u8 = torch.float64, torch.int32, torch.uint8
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        self.t = dict(traj=z((B, sc['traj_cap'], 4), f64), traj_hdr=z((B, 2), i32), traj_box=z((B, (sc['traj_cap'] + 63) // 64, 4), f64),
                      trk_radius=z((B, max(N, 1)), f64),
                      trk_prev=z((B, max(N, 1)), u8), trk_lim=z((B, max(N, 1)), f64), seen_step=z((B, cfg.W, cfg.H), i32),
                      nodes=z((B, sc['node_cap'], A.NODE_F), f64), hash=z((B, sc['hash_cap']), i32),
                      launch_args=z((A.LAUNCH_ARGS_BYTES,), u8), plan_stat=z((B, 4), i32))
        r0 = torch.as_tensor(np.asarray(tracker_radius, dtype=np.float64)).reshape(B, -1)
        self.trk_radius0 = z((B, max(N, 1)), f64)
        if N:
            self.trk_radius0[:, :N] = r0.to(dev)
        self.t['trk_radius'].copy_(self.trk_radius0)
        self.tables['trk_radius0'] = self.trk_radius0

    def struct(self):
        s = A.Plan()
        for k in A.PLAN_INT_FIELDS + A.PLAN_F64_FIELDS + ('acos_key_lo', 'acos_mask'):
            if k in self.scalars:
                setattr(s, k, self.scalars[k])
        s.planner, s.gaze = self.planner, self.gaze
        for k in A.PLAN_TABLES:
            setattr(s, k, self.tables[k].data_ptr())
        for k in A.PLAN_STATE:
            setattr(s, k, self.t[k].data_ptr())
        return s

    # host views used by the gym facade / tests
    def trajectory(self, e):
        """Remaining waypoints of env e as (positions [n, 2], velocities [n, 2])."""
        head, stored = (int(v) for v in self.t['traj_hdr'][e].cpu())
        w = self.t['traj'][e, head:stored].cpu().numpy()
        return w[:, :2].copy(), w[:, 2:].copy()
