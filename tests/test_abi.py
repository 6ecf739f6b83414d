"""The C-ABI library loads on a CPU-only box and exports every symbol include/d2d.h declares; the ctypes
mirror agrees with the header.  No compute calls here (no GPU)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = open(os.path.join(ROOT, 'include', 'd2d.h')).read()


def _declared_functions():
    return sorted(set(re.findall(r'^\s*(?:int|const char \*)\s*(d2d_\w+)\s*\(', HDR, flags=re.M)))


def test_header_constants_match_ctypes_mirror(pkg):
    A = pkg._abi
    defs = dict(re.findall(r'#define\s+(D2D_\w+)\s+(\d+)\b', HDR))
    assert int(defs['D2D_ABI_VERSION']) == A.D2D_ABI_VERSION
    for name, val in (('D2D_AF', A.AF), ('D2D_DF', A.DF), ('D2D_CF', A.CF), ('D2D_KF', A.KF),
                      ('D2D_A_R2', A.A_R2), ('D2D_D_YAW', A.D_YAW), ('D2D_C_BUF_TS', A.C_BUF_TS),
                      ('D2D_F_DONE', A.F_DONE), ('D2D_ST_OBS', A.ST_OBS), ('D2D_DYNAMIC', A.DYNAMIC),
                      ('D2D_SM_EXECUTING', A.SM_EXECUTING), ('D2D_PLANNER_NOMOVE', A.PLANNER_NOMOVE)):
        assert int(defs[name]) == val, name
    # struct field order: names in the header appear in the same order as in the ctypes mirror
    body = HDR[HDR.index('typedef struct d2d_state'):HDR.index('} d2d_state;')]
    fields = re.findall(r'\*\s*(\w+);', body)
    assert tuple(fields) == A.STATE_FIELDS
    body = HDR[HDR.index('typedef struct d2d_cfg'):HDR.index('} d2d_cfg;')]
    names = []
    for line in body.splitlines():
        line = line.split('/*')[0]
        m = re.match(r'\s*(int32_t|double)\s+([\w,\s]+);', line)
        if m:
            names += [n.strip() for n in m.group(2).split(',')]
    assert names == [f[0] for f in A.Cfg._fields_]
    assert C.sizeof(A.Cfg) == 14 * 4 + 16 * 8 and C.sizeof(A.State) == len(A.STATE_FIELDS) * 8
    # d2d_plan: scalars in declaration order, then the pointers
    body = HDR[HDR.index('typedef struct d2d_plan {'):HDR.index('} d2d_plan;')]
    names = []
    for line in body.splitlines():
        line = line.split('/*')[0]
        m = re.match(r'\s*(?:const\s+)?(int32_t|double|int64_t|uint64_t|uint8_t|void)\s+(?:D2D_AS\s+)?\*?\s*(\w+);', line)
        if m:
            names.append(m.group(2))
    assert names == [f[0] for f in A.Plan._fields_]
    assert C.sizeof(A.Plan) == 14 * 4 + 9 * 8 + 2 * 8 + (len(A.PLAN_TABLES) + len(A.PLAN_STATE)) * 8 and len(A.PLAN_INT_FIELDS) == 14
    for name, val in (('D2D_NODE_F', A.NODE_F), ('D2D_PLAN_PRIMITIVE', A.PLAN_PRIMITIVE), ('D2D_GAZE_OXFORD', A.GAZE_OXFORD),
                      ('D2D_LAUNCH_ARGS_BYTES', A.LAUNCH_ARGS_BYTES)):
        assert int(defs[name]) == val, name


def test_hip_library_exports_every_declared_symbol(pkg):
    from drone2d_amd import _lib
    lib, fn = _lib.load_library()
    declared = _declared_functions()
    assert declared == sorted('d2d_' + n for n in pkg._abi.ENTRY_POINTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert fn['abi_version']() == pkg._abi.D2D_ABI_VERSION


def test_argument_validation_without_gpu(pkg):
    """Bad arguments are refused before any launch (so this runs on a CPU-only box)."""
    from drone2d_amd import _lib
    A = pkg._abi
    _, fn = _lib.load_library()
    c, s = A.Cfg(), A.State()
    assert fn['step'](C.byref(c), C.byref(s), None) == -2 and b'ABI' in fn['last_error']()
    c.abi_version = A.D2D_ABI_VERSION
    assert fn['step'](C.byref(c), C.byref(s), None) == -1
    c.B, c.N, c.W, c.H, c.R, c.L, c.T = 1, 1, 50, 50, 50, 33, 1
    c.scale, c.depth, c.dt = 1.0, 80.0, 0.1
    assert fn['step'](C.byref(c), C.byref(s), None) == -4 and b'map_scale' in fn['last_error']()
    c.scale = 10.0
    c.W = 40000                                                              # 16-bit cell planes in LDS: at most 32767 cells a side
    assert fn['step'](C.byref(c), C.byref(s), None) == -4 and b'32767' in fn['last_error']()
    c.W = 50
    assert fn['step'](C.byref(c), C.byref(s), None) == -1 and b'null' in fn['last_error']()
    assert fn['tan_array'](None, None, 5, None) == -1
    assert fn['sincos_array'](None, None, None, 5, None) == -1
    assert fn['plan_stage'](C.byref(c), C.byref(s), None, None) == -1        # null state pointers are refused first


def test_stale_squared_thresholds_are_refused_without_gpu(pkg):
    """d2d_plan.vmax_sq / goal_sq are redundant with vmax / goal_tol and the search trusts them: a value that is not THE threshold
    (a plan struct re-used after vmax changed, say) is refused before any launch.  Pointers are dummies: nothing is dereferenced."""
    import ctypes as C
    import math
    from drone2d_amd import _lib, host_init, device_plugins
    A = pkg._abi
    _, fn = _lib.load_library()
    p = pkg.with_defaults(pkg.Params(planner='Primitive', gaze_method='Oxford', agent_number=10, agent_radius=15,
                                     agent_max_speed=20, drone_max_speed=40, map_id=1))
    cfg = host_init.derive_cfg(p, B=4, N=10, T=1, planner_mode=A.PLANNER_EXTERNAL, kf_enabled=True)
    st = A.State()
    for name, typ in A.State._fields_:
        setattr(st, name, 1)
    sc, _ = device_plugins.build_tables(p, cfg)
    plan = A.Plan()
    for name in A.PLAN_TABLES + A.PLAN_STATE:
        setattr(plan, name, 1)
    for k, v in sc.items():
        setattr(plan, k, v)
    plan.planner, plan.gaze = A.PLAN_PRIMITIVE, A.GAZE_NONE
    good = plan.vmax_sq
    assert math.sqrt(good) < 40.0 <= math.sqrt(math.nextafter(good, math.inf))
    for bad in (1600.0, math.nextafter(good, 0.0), -1.0, 900.0):
        plan.vmax_sq = bad
        assert fn['plan_stage'](C.byref(cfg), C.byref(st), C.byref(plan), None) == -1 and b'vmax_sq' in fn['last_error'](), bad
    plan.vmax_sq = good
    plan.goal_sq = 100.0 - 1e-9
    assert fn['plan_stage'](C.byref(cfg), C.byref(st), C.byref(plan), None) == -1 and b'goal_sq' in fn['last_error']()
    plan.goal_sq, plan.vmax, plan.vmax_sq = 0.0, 0.0, 1600.0       # vmax <= 0: only -1 ("nothing passes") or 0 are accepted
    assert fn['plan_stage'](C.byref(cfg), C.byref(st), C.byref(plan), None) == -1 and b'vmax_sq' in fn['last_error']()


def test_oracle_exports_the_same_surface(pkg, oracle):
    for n in pkg._abi.ENTRY_POINTS:
        assert hasattr(oracle.lib, 'd2d_oracle_' + n)


def test_product_never_touches_the_oracle():
    """The package must not import / link / execute anything under oracle/ (no CPU fallback)."""
    pk = os.path.join(ROOT, 'gym-drone2d-activeperception_amd')
    for dp, _, fs in os.walk(pk):
        for f in fs:
            if f.endswith(('.py', '.hip', '.h', '.cpp', '.sh')):
                txt = open(os.path.join(dp, f)).read()
                assert 'liboracle' not in txt and 'oracle_lib' not in txt and 'd2d_oracle_' not in txt, f


def test_header_is_plain_c_for_binders(tmp_path):
    """include/d2d.h compiles as C99 and as C++ without any macro set (D2D_AS expands to nothing: plain pointers), and the
    struct sizes are the ctypes mirror's."""
    import shutil
    import subprocess
    if not shutil.which('gcc'):
        pytest.skip('no gcc')
    from drone2d_amd import _abi as A
    src = tmp_path / 'hdr.c'
    src.write_text('#include "include/d2d.h"\n#include <stdio.h>\n'
                   'int main(void) { printf("%zu %zu %zu\\n", sizeof(d2d_cfg), sizeof(d2d_state), sizeof(d2d_plan)); return 0; }\n')
    exe = tmp_path / 'hdr'
    subprocess.check_call(['gcc', '-std=c99', '-Wall', '-Werror', '-pedantic', '-I', ROOT, str(src), '-o', str(exe)])
    sizes = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert sizes == [C.sizeof(A.Cfg), C.sizeof(A.State), C.sizeof(A.Plan)]
    subprocess.check_call(['g++', '-std=c++11', '-Wall', '-Werror', '-I', ROOT, '-x', 'c++', '-fsyntax-only', str(src)])


def test_bench_workload_keeps_four_workgroups_per_cu(pkg):
    """BASELINE config 2 (4096 envs, 10 agents, default geometry): the persistent closed loop launches 4-wave workgroups and a
    CU's 160 KB of LDS must hold FOUR of them (16 waves per CU = 4096 envs resident on 256 CUs) -- a few hundred bytes more per
    wave in any phase drop it to three and a quarter of the envs queue behind the others.  d2d_launch_shape needs no GPU."""
    from drone2d_amd import _lib, host_init, device_plugins
    import ctypes as C
    p = pkg.with_defaults(pkg.Params(planner='Primitive', gaze_method='Oxford', agent_number=10, agent_radius=15,
                                     agent_max_speed=20, drone_max_speed=40, map_id=1))
    cfg = host_init.derive_cfg(p, B=4096, N=10, T=1, planner_mode=pkg._abi.PLANNER_EXTERNAL, kf_enabled=True)
    wpb, lds, per_cu, spec = _lib.launch_shape(cfg)
    assert (wpb, spec) == (4, 1) and per_cu >= 4, (wpb, lds, per_cu)
    sc, tb = device_plugins.build_tables(p, cfg)
    plan = pkg._abi.Plan()
    for k, v in sc.items():
        setattr(plan, k, v)
    plan.planner, plan.gaze = pkg._abi.PLAN_PRIMITIVE, pkg._abi.GAZE_OXFORD
    plan.launch_args = 1                      # any non-null value: the persistent path is chosen (nothing is dereferenced here)
    wpb, lds, per_cu, spec = _lib.launch_shape(cfg, plan)
    # (round 4: the persistent kernel runs ONE env per workgroup -- a workgroup holds its slots until its slowest wave ends -- so the
    # budget is 10 240 B per wave = 16 one-wave workgroups per CU)
    assert (wpb, spec) == (1, 1) and per_cu >= 16 and lds <= 10240, (wpb, lds, per_cu)
    # config 3's agent count (172) on the default geometry: one or two waves per workgroup, still the specialised kernels
    cfg3 = host_init.derive_cfg(p, B=16, N=172, T=1, planner_mode=pkg._abi.PLANNER_EXTERNAL, kf_enabled=True)
    wpb3, lds3, per_cu3, spec3 = _lib.launch_shape(cfg3)
    assert wpb3 >= 1 and spec3 == 0 and wpb3 * per_cu3 >= 12     # many agents: tiles, not whole grids; 12 waves per CU (50 B per
    #                                                               agent, at most 64 ray candidates; 8 at 98 B per agent)
    # config 4's agent count (24): whole grids like config 2, 16 waves per CU
    cfg4 = host_init.derive_cfg(p, B=16, N=24, T=1, planner_mode=pkg._abi.PLANNER_EXTERNAL, kf_enabled=True)
    wpb4, lds4, per_cu4, spec4 = _lib.launch_shape(cfg4)
    assert (wpb4, spec4) == (4, 1) and per_cu4 >= 4
    # config 5 (100 agents on 640 x 640 cells, 640 rays): the generic kernel within the 10 KB per wave that keep 16 waves on a
    # CU -- the dynamic-grid update of a grid that large needs no LDS structure (two phases)
    p5 = pkg.with_defaults(pkg.Params(planner='NoMove', agent_number=100, agent_radius=15, agent_max_speed=40, drone_max_speed=40,
                                      map_size=[6400, 6400], init_pos=[3200, 3200], target_list=[[6000, 6000]]))
    cfg5 = host_init.derive_cfg(p5, B=16, N=100, T=1, planner_mode=pkg._abi.PLANNER_NOMOVE, kf_enabled=True)
    wpb5, lds5, per_cu5, spec5 = _lib.launch_shape(cfg5)
    assert spec5 == 0 and wpb5 * per_cu5 >= 16 and lds5 <= 4 * 10240, (wpb5, lds5, per_cu5)
