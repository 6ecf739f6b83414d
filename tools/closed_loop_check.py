#!/usr/bin/env python3
"""Closed-loop check of the device plugins against the golden traces (runs on the GPU box, or with
--oracle on the CPU): gaze -> perceive -> plan -> act per step, every recorded value compared."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np  # noqa: E402
import drone2d_amd as pkg  # noqa: E402
from drone2d_amd import device_plugins as DP  # noqa: E402
import replay  # noqa: E402


def run(backend, name, gaze):
    R = replay.Replay(pkg, backend, name, kf=True)
    ps = DP.PluginState(R.p, R.cfg, backend.device, R.world['tracker_radius'][None], planner='Primitive', gaze=gaze)
    plan = ps.struct()
    fx = R.fx
    t0 = time.time()
    for t in range(R.T):
        s = R.st.struct()
        if gaze == 'Oxford':
            backend.gaze_stage(R.cfg, s, plan)
            backend.sync()
            a = float(R.st.action[0])
            if a != float(fx['t_action'][t]):
                return f'{name}: step {t + 1} ACTION {a} != {fx["t_action"][t]}'
        else:
            R.st.action.fill_(float(fx['t_action'][t]))
        backend.perceive(R.cfg, s)
        backend.plan_stage(R.cfg, s, plan)
        backend.sync()
        ok, wv = int(R.st.plan_ok[0]), int(R.st.wp_valid[0])
        wp = R.st.wp[0].cpu().numpy()
        hdr = ps.t['traj_hdr'][0].cpu().numpy()
        tl = int(hdr[1] - hdr[0])
        if ok != int(fx['t_plan_ok'][t]) or wv != int(fx['t_wp_valid'][t]) or (wv and not np.array_equal(wp, fx['t_wp'][t])) \
                or tl != int(fx['t_traj_len'][t]):
            return (f'{name}: step {t + 1} PLAN ok {ok}/{fx["t_plan_ok"][t]} valid {wv}/{fx["t_wp_valid"][t]} wp {wp} / {fx["t_wp"][t]} '
                    f'len {tl}/{fx["t_traj_len"][t]} stat {ps.t["plan_stat"][0].cpu().numpy()}')
        backend.act(R.cfg, s)
        backend.sync()
        try:
            R.compare(t)
        except AssertionError as ex:
            return f'{name}: {str(ex)[:300]}'
    return f'{name}: all {R.T} steps match (searches/expansions/nodes/overflow {ps.t["plan_stat"][0].cpu().numpy()}) {time.time() - t0:.1f}s'


def main():
    if '--oracle' in sys.argv:
        from oracle_lib import OracleBackend
        backend = OracleBackend()
    else:
        from drone2d_amd import _lib
        backend = _lib.HipBackend('cuda:0')
    names = [a for a in sys.argv[1:] if not a.startswith('--')] or replay.TRACES_CLOSED + replay.TRACES_PLANNED
    bad = 0
    for n in names:
        gaze = 'Oxford' if 'oxford' in n else 'replay'
        msg = run(backend, n, gaze)
        print(msg, flush=True)
        bad += 'match' not in msg
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
