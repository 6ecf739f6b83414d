"""Test-side loader of the CPU oracle (oracle/liboracle.so).  Test infrastructure: the product package
never imports this."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, 'oracle')


def build_oracle():
    if os.environ.get('D2D_ORACLE_LIB'):     # another build of the restatement, e.g. `make -C oracle asan` (see oracle/Makefile)
        return os.path.abspath(os.environ['D2D_ORACLE_LIB'])
    so = os.path.join(ORACLE_DIR, 'liboracle.so')
    src = os.path.join(ORACLE_DIR, 'd2d_oracle.c')
    hdr = os.path.join(ROOT, 'include', 'd2d.h')
    if (not os.path.isfile(so)) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(['make', '-C', ORACLE_DIR, '-s'])
    return so


class OracleBackend:
    """Same call surface as the product's HipBackend, on host pointers."""
    name = 'oracle'
    device = 'cpu'

    def __init__(self):
        import drone2d_amd
        self.A = drone2d_amd._abi
        self.lib = C.CDLL(build_oracle())
        self.fn = self.A.bind(self.lib, prefix='d2d_oracle_')
        assert self.fn['abi_version']() == self.A.D2D_ABI_VERSION

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError(f'oracle error {rc}: {self.fn["last_error"]().decode()}')

    def run_stages(self, cfg, st, stages):
        self._chk(self.fn['run_stages'](C.byref(cfg), C.byref(st), stages, None))

    def step(self, cfg, st):
        self._chk(self.fn['step'](C.byref(cfg), C.byref(st), None))

    def perceive(self, cfg, st):
        self._chk(self.fn['perceive'](C.byref(cfg), C.byref(st), None))

    def act(self, cfg, st):
        self._chk(self.fn['act'](C.byref(cfg), C.byref(st), None))

    def rollout(self, cfg, st, nsteps, actions, pin=None, coll_out=None, wp_steps=None):
        self._chk(self.fn['rollout'](C.byref(cfg), C.byref(st), nsteps, actions.data_ptr(),
                                     None if wp_steps is None else wp_steps.data_ptr(),
                                     None if pin is None else pin.data_ptr(),
                                     None if coll_out is None else coll_out.data_ptr(), None))

    def reset(self, cfg, st, init, mask=None):
        self._chk(self.fn['reset'](C.byref(cfg), C.byref(st), C.byref(init),
                                   None if mask is None else mask.data_ptr(), None))

    def gaze_stage(self, cfg, st, plan):
        self._chk(self.fn['gaze_stage'](C.byref(cfg), C.byref(st), C.byref(plan), None))

    def plan_stage(self, cfg, st, plan):
        self._chk(self.fn['plan_stage'](C.byref(cfg), C.byref(st), C.byref(plan), None))

    def closed_loop(self, cfg, st, plan, nsteps, on_done=0, init=None):
        self._chk(self.fn['closed_loop'](C.byref(cfg), C.byref(st), C.byref(plan), nsteps, int(on_done),
                                         None if init is None else C.byref(init), None))

    def plan_reset(self, cfg, plan, mask=None, mask_stride=1):
        self._chk(self.fn['plan_reset'](C.byref(cfg), C.byref(plan), None if mask is None else mask.data_ptr(),
                                        mask_stride, None))

    def sincos_array(self, x, s, c):
        self._chk(self.fn['sincos_array'](x.data_ptr(), s.data_ptr(), c.data_ptr(), x.numel(), None))

    def tan_array(self, x, out):
        self._chk(self.fn['tan_array'](x.data_ptr(), out.data_ptr(), x.numel(), None))

    def sync(self):
        pass
