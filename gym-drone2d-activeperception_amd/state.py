"""Device-resident batch state: one torch tensor per field of include/d2d.h `d2d_state`.

torch is plumbing here (allocation, host<->device copies, streams); all arithmetic on these tensors is
done by the HIP library through raw pointers."""
import ctypes as C

import numpy as np
import torch

from . import _abi as A


def _spec(B, N, W, H, L, T):
    f64, i32, u8, f32 = torch.float64, torch.int32, torch.uint8, torch.float32
    return dict(
        agents=((B, A.AF, N), f64), agent_unit=((B, N), i32), dyn_prev=((B, N, 3), i32),
        gt=((B, W, H), u8), dmap=((B, W, H), u8), drone=((B, A.DF), f64), target=((B, 2), f64),
        targets=((B, T, 2), f64), counters=((B, A.CF), i32), active=((B, N), u8),
        kf=((B, N, A.KF), f64), kf_len=((B, N), i32),
        action=((B,), f64), plan_ok=((B,), u8), wp_valid=((B,), u8), wp=((B, 6), f64),
        hit=((B, N), u8), newly=((B,), i32), flags=((B, 4), u8), obs_local=((B, L, L), u8),
        obs_yaw=((B,), f32))


WORLD_FIELDS = ('agents', 'agent_unit', 'dyn_prev', 'gt', 'dmap', 'drone', 'target', 'targets', 'counters',
                'active', 'kf', 'kf_len')


class BatchState:
    """All tensors of one shard.  `struct()` builds the ctypes d2d_state that points at them."""

    def __init__(self, cfg, device):
        self.cfg = cfg
        self.device = torch.device(device)
        self.t = {}
        for name, (shape, dt) in _spec(cfg.B, cfg.N, cfg.W, cfg.H, cfg.L, cfg.T).items():
            self.t[name] = torch.zeros(shape, dtype=dt, device=self.device)
        self.noise = None
        self._dummy = torch.zeros(64, dtype=torch.float64, device=self.device)   # target of empty fields (N == 0)
        self._kf_defaults()

    def _kf_defaults(self):
        kf = self.t['kf']
        kf.zero_()
        for i, v in ((0, 1.0), (5, 1.0), (10, 10.0), (15, 10.0)):   # Sigma = diag(1, 1, 10, 10), utils.py:181
            kf[:, :, 4 + i] = v
        self.t['kf_len'].fill_(1)

    def __getattr__(self, name):
        t = self.__dict__.get('t')
        if t is not None and name in t:
            return t[name]
        raise AttributeError(name)

    def load_worlds(self, worlds):
        """Fill the world fields from a list of host_init.init_world() dicts (one per env)."""
        B = self.cfg.B
        assert len(worlds) == B
        for name in ('agents', 'agent_unit', 'dyn_prev', 'gt', 'dmap', 'drone', 'target', 'targets', 'counters'):
            # in slices of <= 256 MB of host staging: a config-5 shard is 32768 grids of 640 x 640 cells = 13.4 GB per field
            per = max(1, int(np.asarray(worlds[0][name]).nbytes))
            step = max(1, (256 << 20) // per)
            for c0 in range(0, B, step):
                arr = np.stack([w[name] for w in worlds[c0:c0 + step]])
                self.t[name][c0:c0 + step].copy_(torch.from_numpy(arr).to(self.t[name].dtype))
        self.t['active'].zero_()
        self._kf_defaults()

    def clone_world(self):
        """Snapshot of the world fields (reset source)."""
        snap = BatchState.__new__(BatchState)
        snap.cfg, snap.device, snap.noise, snap._dummy = self.cfg, self.device, None, self._dummy
        snap.t = {k: (v.clone() if k in WORLD_FIELDS else v) for k, v in self.t.items()}
        return snap

    def struct(self, use_planner_inputs=True):
        s = A.State()
        for name in A.STATE_FIELDS:
            if name == 'noise':
                s.noise = self.noise.data_ptr() if self.noise is not None else None
            elif name in ('plan_ok', 'wp_valid', 'wp') and not use_planner_inputs:
                setattr(s, name, None)
            else:
                t = self.t[name]
                setattr(s, name, t.data_ptr() if t.numel() else self._dummy.data_ptr())
        return s

    def to(self, device):
        out = BatchState.__new__(BatchState)
        out.cfg, out.device = self.cfg, torch.device(device)
        out._dummy = self._dummy.to(device)
        out.t = {k: v.to(device) for k, v in self.t.items()}
        out.noise = None if self.noise is None else self.noise.to(device)
        return out
