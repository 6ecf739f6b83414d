"""Build-container-only pin of the oracle (skipped where /root/reference is absent, i.e. on the GPU box): random
closed-loop Oxford + Primitive episodes are played by the LIVE reference (tests/golden/make_golden.py in a
subprocess, import-time stubs only) and replayed through the oracle's restatement of the step, the planner and the
gaze policy -- gaze action, plan() result, head waypoint, len(trajectory) and the full state of every step."""
import glob
import os
import subprocess
import sys

import pytest

from test_plugins_cpu import _closed_loop

REF = '/root/reference'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason='reference checkout not present')
N_EPISODES = 8


@pytest.fixture(scope='module')
def live_traces(tmp_path_factory):
    out = str(tmp_path_factory.mktemp('live'))
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'tests', 'golden', 'make_golden.py'), 'live', out, '4242', str(N_EPISODES)],
                          stdout=subprocess.DEVNULL)
    files = sorted(glob.glob(os.path.join(out, 'live_oxford_*.npz')))
    assert len(files) == N_EPISODES
    return files


@pytest.mark.parametrize('k', range(N_EPISODES))
def test_oracle_reproduces_live_reference_episode(pkg, oracle, live_traces, k):
    _closed_loop(pkg, oracle, live_traces[k])


N_WIDE = 6


@pytest.fixture(scope='module')
def live_traces_wide(tmp_path_factory):
    """The wider family: map sizes, drone radius / acceleration limit / yaw rate, short views, a target next to the start."""
    out = str(tmp_path_factory.mktemp('livewide'))
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'tests', 'golden', 'make_golden.py'), 'live', out, '9100', str(N_WIDE), 'wide'],
                          stdout=subprocess.DEVNULL)
    files = sorted(glob.glob(os.path.join(out, 'live_oxford_*.npz')))
    assert len(files) == N_WIDE
    return files


@pytest.mark.parametrize('k', range(N_WIDE))
def test_oracle_reproduces_live_reference_episode_wide(pkg, oracle, live_traces_wide, k):
    _closed_loop(pkg, oracle, live_traces_wide[k])
