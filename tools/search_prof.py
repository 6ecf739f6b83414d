#!/usr/bin/env python3
"""Diagnostic: per-section shader-clock time of the device A* expansion loop (needs the -DD2D_SEARCH_PROF build:
D2D_OUT=libd2d_prof.so D2D_EXTRA_FLAGS=-DD2D_SEARCH_PROF csrc/build.sh; D2D_LIB=.../libd2d_prof.so)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
import drone2d_amd as pkg  # noqa: E402
from drone2d_amd import _lib, device_plugins as DP  # noqa: E402
import replay  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
be = _lib.HipBackend('cuda:0')
prof = be.lib.d2d_debug_search_prof
prof.argtypes = [C.c_void_p, C.c_int]
TRACE = sys.argv[2] if len(sys.argv) > 2 else 'deadlock_primitive'
R = replay.Replay(pkg, be, TRACE, kf=True, copies=B)
ps = DP.PluginState(R.p, R.cfg, be.device, [R.world['tracker_radius']] * B, planner='Primitive', gaze='external')
plan = ps.struct()
names = ['argmin', 'cur fields', 'speed filter', 'is_free pairs', 'key + probe', 'dedup', 'exists + writes', 'fence', 'TOTAL', 'batches']
for t in range(int(sys.argv[3]) if len(sys.argv) > 3 else 4):
    s = R.st.struct()
    R.st.action.fill_(float(R.fx['t_action'][t]))
    be.perceive(R.cfg, s)
    prof(None, 1)
    be.plan_stage(R.cfg, s, plan)
    be.act(R.cfg, s)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 16)()
    prof(out, 0)
    if out[9] == 0:
        continue
    n = max(1, out[9])
    print(f'step {t + 1} B={B}:', ', '.join(f'{nm} {out[i] / n:.0f}' for i, nm in enumerate(names[:9])), f'(clock ticks per expansion, {out[9]} expansions); per search: setup {out[10]}, path + trajectory {out[11]} ({out[12]} successful)')
