"""Test harness, not product: one rank of a CPU dry run of bench.py.  The HIP backend of the package is replaced by the
CPU oracle BEFORE bench.main() runs, so that the multi-rank code path of the bench itself (sharding by rank, barriers,
max-over-ranks timing, the all_gather of episode statistics, the JSON line) is exercised without a GPU.  bench.py
itself never touches the oracle outside its cpu_baseline leg."""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import drone2d_amd  # noqa: E402
from drone2d_amd import _lib  # noqa: E402
from oracle_lib import OracleBackend  # noqa: E402

_lib.HipBackend = lambda device='cpu': OracleBackend()
os.environ['D2D_BENCH_ENTRY'] = os.path.abspath(__file__)   # ranks that bench.py starts itself (--gpus N, no launcher) come back here
sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, 'bench.py'), run_name='__main__')
