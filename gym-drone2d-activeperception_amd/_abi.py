"""ctypes mirror of include/d2d.h (the C ABI of the HIP library).  Field order and types must match
the header exactly; tests/test_abi.py cross-checks sizes and constants against the header text."""
import ctypes as C

D2D_ABI_VERSION = 8

UNEXPLORED, OCCUPIED, UNOCCUPIED, DYNAMIC = 0, 1, 2, 3
SM_WAIT_FOR_GOAL, SM_GOAL_REACHED, SM_PLANNING, SM_EXECUTING = 0, 1, 2, 3
PLANNER_EXTERNAL, PLANNER_NOMOVE = 0, 1
PLAN_NONE, PLAN_PRIMITIVE = 0, 1
GAZE_NONE, GAZE_OXFORD = 0, 1
NODE_F = 12

AF = 6
A_PX, A_PY, A_VX, A_VY, A_R, A_R2 = range(6)
DF = 8
D_X, D_Y, D_YAW, D_VX, D_VY, D_AX, D_AY, D_PAD = range(8)
CF = 8
C_STEPS, C_FAIL, C_SM, C_TGT_NEXT, C_NTGT, C_TRACKED, C_BUF_N, C_BUF_TS = range(8)
F_COLLISION, F_DEADLOCK, F_FREEZING, F_DONE = range(4)
KF = 20

ST_FSM, ST_AGENTS, ST_RAYCAST, ST_DYNGRID, ST_TRACKER, ST_CONTROL, ST_COLLIDE, ST_OBS = (1 << i for i in range(8))
ST_PERCEIVE = ST_FSM | ST_AGENTS | ST_RAYCAST | ST_DYNGRID | ST_TRACKER
ST_ACT = ST_CONTROL | ST_COLLIDE | ST_OBS
ST_ALL = ST_PERCEIVE | ST_ACT
ST_SKIP_DONE = 256
DONE_CONTINUE, DONE_RESET, DONE_FREEZE = 0, 1, 2


class Cfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ('abi_version', 'B', 'N', 'W', 'H', 'R', 'L', 'T', 'planner_mode', 'kf_enabled',
                 'noise_rows', 'noise_row0', 'grid_tile', 'reserved2')] + \
               [(n, C.c_double) for n in
                ('dt', 'scale', 'W_px', 'H_px', 'ray_off0', 'ray_dth', 'depth', 'drone_radius', 'yaw_rate',
                 'max_acc', 'max_steps', 'sigma', 'kf_lo_x', 'kf_hi_x', 'kf_lo_y', 'kf_hi_y')]


STATE_FIELDS = ('agents', 'agent_unit', 'dyn_prev', 'gt', 'dmap', 'drone', 'target', 'targets', 'counters',
                'active', 'kf', 'kf_len', 'action', 'plan_ok', 'wp_valid', 'wp', 'noise', 'hit', 'newly',
                'flags', 'obs_local', 'obs_yaw')


class State(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in STATE_FIELDS]


PLAN_INT_FIELDS = ('planner', 'gaze', 'nu', 'n_sample', 'n_ts', 'max_itr', 'traj_cap', 'node_cap', 'hash_cap', 'n_yaw',
                   'pw_nleaf', 'pw_nprog', 'tobs_len', 'pw_ntree')
PLAN_F64_FIELDS = ('horizon', 'vmax', 'safe_dist', 'goal_tol', 'agent_radius', 'half_fov', 'yaw_rate_max', 'vmax_sq', 'goal_sq')
PLAN_TABLES = ('u_space', 'sample_t', 'traj_t', 'yaw_space', 'tobs_tab', 'pw_leaf', 'pw_prog', 'pw_tree', 'pw_rowleaf', 'trk_radius0')
PLAN_STATE = ('traj', 'traj_hdr', 'traj_box', 'trk_radius', 'trk_prev', 'trk_lim', 'seen_step', 'nodes', 'hash', 'launch_args', 'plan_stat')
LAUNCH_ARGS_BYTES = 2048


class Plan(C.Structure):
    """include/d2d.h `d2d_plan`."""
    _fields_ = [(n, C.c_int32) for n in PLAN_INT_FIELDS] + [(n, C.c_double) for n in PLAN_F64_FIELDS] + \
               [('acos_key_lo', C.c_int64), ('acos_mask', C.c_uint64)] + \
               [(n, C.c_void_p) for n in PLAN_TABLES + PLAN_STATE]


def bind(lib, prefix='d2d_'):
    """Declare argtypes / restypes of every entry point of include/d2d.h on a loaded CDLL."""
    P = C.POINTER
    sig = {
        'abi_version': (C.c_int, []),
        'last_error': (C.c_char_p, []),
        'step': (C.c_int, [P(Cfg), P(State), C.c_void_p]),
        'perceive': (C.c_int, [P(Cfg), P(State), C.c_void_p]),
        'act': (C.c_int, [P(Cfg), P(State), C.c_void_p]),
        'run_stages': (C.c_int, [P(Cfg), P(State), C.c_uint32, C.c_void_p]),
        'rollout': (C.c_int, [P(Cfg), P(State), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
        'reset': (C.c_int, [P(Cfg), P(State), P(State), C.c_void_p, C.c_void_p]),
        'tan_array': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
        'gaze_stage': (C.c_int, [P(Cfg), P(State), P(Plan), C.c_void_p]),
        'plan_stage': (C.c_int, [P(Cfg), P(State), P(Plan), C.c_void_p]),
        'closed_loop': (C.c_int, [P(Cfg), P(State), P(Plan), C.c_int32, C.c_int32, P(State), C.c_void_p]),
        'plan_reset': (C.c_int, [P(Cfg), P(Plan), C.c_void_p, C.c_int32, C.c_void_p]),
        'sincos_array': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
        'launch_shape': (C.c_int, [P(Cfg), P(Plan), P(C.c_int32 * 4)]),
    }
    out = {}
    for name, (res, args) in sig.items():
        if name in OPTIONAL and not hasattr(lib, prefix + name):
            continue                       # a diagnostic query older builds of the library lack (A/B runs via D2D_LIB)
        fn = getattr(lib, prefix + name)   # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
        out[name] = fn
    return out


OPTIONAL = ('launch_shape',)
ENTRY_POINTS = ('abi_version', 'last_error', 'step', 'perceive', 'act', 'run_stages', 'rollout', 'reset',
                'tan_array', 'gaze_stage', 'plan_stage', 'closed_loop', 'plan_reset', 'sincos_array', 'launch_shape')
