"""Loader of the HIP library (csrc/libd2d_hip.so).  There is NO CPU fallback: if the library is missing
or its ABI does not match, importing the backend raises."""
import ctypes as C
import os

from . import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('D2D_LIB') or os.path.join(_HERE, 'csrc', 'libd2d_hip.so')   # D2D_LIB: A/B builds


class D2DError(RuntimeError):
    pass


def launch_shape(cfg, plan=None, fn=None):
    """(waves per workgroup, LDS bytes per workgroup, workgroups per CU by LDS, specialised kernels?) -- d2d_launch_shape;
    needs no GPU."""
    if fn is None:
        fn = load_library()[1]
    out = (C.c_int32 * 4)()
    rc = fn['launch_shape'](C.byref(cfg), None if plan is None else C.byref(plan), C.byref(out))
    if rc != 0:
        raise D2DError(f'd2d error {rc}: {fn["last_error"]().decode()}')
    return tuple(out)


def load_library(path=LIB_PATH):
    # torch first: its bundled HIP runtime has to be the one this library binds to.  Loaded the other way round the
    # process ends up with two runtimes and every launch fails with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    if not os.path.isfile(path):
        raise D2DError(f'{path} not found: build it with gym-drone2d-activeperception_amd/csrc/build.sh '
                       '(or __graft_entry__.build()); there is no CPU fallback')
    lib = C.CDLL(path)
    fn = A.bind(lib, prefix='d2d_')
    v = fn['abi_version']()
    if v != A.D2D_ABI_VERSION:
        raise D2DError(f'libd2d_hip.so ABI {v} != expected {A.D2D_ABI_VERSION}: rebuild')
    return lib, fn


class HipBackend:
    """Thin call surface over the C ABI; launches go to torch's current HIP stream of `device`."""
    name = 'hip'
    supports_tiled_grids = True      # d2d_cfg.grid_tile = 16 (the CPU oracle keeps the reference's row-major grids)

    def __init__(self, device='cuda:0'):
        import torch
        if not torch.cuda.is_available():
            raise D2DError('HipBackend needs a GPU (torch.cuda.is_available() is False)')
        self.torch = torch
        self.device = torch.device(device)
        self.lib, self.fn = load_library()

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _chk(self, rc):
        if rc != 0:
            raise D2DError(f'd2d error {rc}: {self.fn["last_error"]().decode()}')

    def run_stages(self, cfg, st, stages):
        self._chk(self.fn['run_stages'](C.byref(cfg), C.byref(st), stages, self._stream()))

    def step(self, cfg, st):
        self._chk(self.fn['step'](C.byref(cfg), C.byref(st), self._stream()))

    def perceive(self, cfg, st):
        self._chk(self.fn['perceive'](C.byref(cfg), C.byref(st), self._stream()))

    def act(self, cfg, st):
        self._chk(self.fn['act'](C.byref(cfg), C.byref(st), self._stream()))

    def rollout(self, cfg, st, nsteps, actions, pin=None, coll_out=None, wp_steps=None):
        self._chk(self.fn['rollout'](C.byref(cfg), C.byref(st), nsteps, actions.data_ptr(),
                                     None if wp_steps is None else wp_steps.data_ptr(),
                                     None if pin is None else pin.data_ptr(),
                                     None if coll_out is None else coll_out.data_ptr(), self._stream()))

    def reset(self, cfg, st, init, mask=None):
        self._chk(self.fn['reset'](C.byref(cfg), C.byref(st), C.byref(init),
                                   None if mask is None else mask.data_ptr(), self._stream()))

    def gaze_stage(self, cfg, st, plan):
        self._chk(self.fn['gaze_stage'](C.byref(cfg), C.byref(st), C.byref(plan), self._stream()))

    def plan_stage(self, cfg, st, plan):
        self._chk(self.fn['plan_stage'](C.byref(cfg), C.byref(st), C.byref(plan), self._stream()))

    def closed_loop(self, cfg, st, plan, nsteps, on_done=0, init=None):
        self._chk(self.fn['closed_loop'](C.byref(cfg), C.byref(st), C.byref(plan), nsteps, int(on_done),
                                         None if init is None else C.byref(init), self._stream()))

    def plan_reset(self, cfg, plan, mask=None, mask_stride=1):
        self._chk(self.fn['plan_reset'](C.byref(cfg), C.byref(plan), None if mask is None else mask.data_ptr(),
                                        mask_stride, self._stream()))

    def sincos_array(self, x, s, c):
        self._chk(self.fn['sincos_array'](x.data_ptr(), s.data_ptr(), c.data_ptr(), x.numel(), self._stream()))

    def tan_array(self, x, out):
        self._chk(self.fn['tan_array'](x.data_ptr(), out.data_ptr(), x.numel(), self._stream()))

    def sync(self):
        self.torch.cuda.synchronize(self.device)
