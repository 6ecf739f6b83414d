"""Device-resident batch state: one torch tensor per field of include/d2d.h `d2d_state`.

torch is plumbing here (allocation, host<->device copies, streams); all arithmetic on these tensors is
done by the HIP library through raw pointers."""
import ctypes as C

import numpy as np
import torch

from . import _abi as A


TILE = 16          # d2d_cfg.grid_tile: cells per tile edge of the tiled grid layout (include/d2d.h)


def tile_grid(g, tile=TILE):
    """[..., W, H] row-major grid(s) -> [..., ceil(W / 16) * ceil(H / 16) * 256] in the tiled layout of include/d2d.h
    (d2d_cfg.grid_tile = 16): tiles row-major over (ceil(W / 16), ceil(H / 16)), cells row-major inside a tile; cells of the
    padding (W or H not a multiple of 16) are 0 and never read."""
    g = torch.as_tensor(g)
    W, H = g.shape[-2:]
    Wt, Ht = -(-W // tile), -(-H // tile)
    lead = g.shape[:-2]
    if (Wt * tile, Ht * tile) != (W, H):
        pad = torch.zeros(lead + (Wt * tile, Ht * tile), dtype=g.dtype, device=g.device)
        pad[..., :W, :H] = g
        g = pad
    g = g.reshape(lead + (Wt, tile, Ht, tile)).transpose(-3, -2)
    return g.reshape(lead + (Wt * Ht * tile * tile,)).contiguous()


def untile_grid(t, W, H, tile=TILE):
    """Inverse of tile_grid: [..., Wt * Ht * 256] -> [..., W, H] (a copy)."""
    Wt, Ht = -(-W // tile), -(-H // tile)
    lead = t.shape[:-1]
    g = t.reshape(lead + (Wt, Ht, tile, tile)).transpose(-3, -2).reshape(lead + (Wt * tile, Ht * tile))
    return g[..., :W, :H].contiguous()


def distinct_worlds(worlds):
    """(the distinct dicts of the list in order of first appearance, the position of every entry among them): by identity, not by
    content -- two equal worlds built separately stay two."""
    seen, distinct, index = {}, [], []
    for w in worlds:
        k = seen.get(id(w))
        if k is None:
            k = seen[id(w)] = len(distinct)
            distinct.append(w)
        index.append(k)
    return distinct, index


def _spec(B, N, W, H, L, T, grid_tile=0):
    f64, i32, u8, f32 = torch.float64, torch.int32, torch.uint8, torch.float32
    gshape = (B, (-(-W // grid_tile)) * (-(-H // grid_tile)) * grid_tile * grid_tile) if grid_tile else (B, W, H)
    return dict(
        agents=((B, A.AF, N), f64), agent_unit=((B, N), i32), dyn_prev=((B, N, 3), i32),
        gt=(gshape, u8), dmap=(gshape, u8), drone=((B, A.DF), f64), target=((B, 2), f64),
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
        for name, (shape, dt) in _spec(cfg.B, cfg.N, cfg.W, cfg.H, cfg.L, cfg.T, cfg.grid_tile).items():
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

    def logical(self, name):
        """Field `name` as the reference indexes it: the grids come back [B, W, H] whatever the device layout (a de-tiled copy
        when cfg.grid_tile != 0 -- write through `set_grid`), every other field is the tensor itself."""
        t = self.t[name]
        if name in ('gt', 'dmap') and self.cfg.grid_tile:
            return untile_grid(t, self.cfg.W, self.cfg.H, self.cfg.grid_tile)
        return t

    def set_grid(self, name, grid, envs=slice(None)):
        """Overwrite grid `name` ('gt' / 'dmap') of `envs` from [n, W, H] values given in the reference's indexing."""
        g = torch.as_tensor(grid, dtype=torch.uint8)
        if self.cfg.grid_tile:
            g = tile_grid(g, self.cfg.grid_tile)
        self.t[name][envs] = g.to(self.device)

    def load_worlds(self, worlds):
        """Fill the world fields from a list of host_init.init_world() dicts (one per env).  A list that names the same dict many
        times (a sweep's start cells on one seeded world, a bench batch tiled from fewer distinct worlds) is staged once per
        distinct world and spread over the envs on the device."""
        B = self.cfg.B
        assert len(worlds) == B
        distinct, index = distinct_worlds(worlds)
        U = len(distinct)
        idx = torch.as_tensor(index, dtype=torch.int64, device=self.device) if U < B else None
        for name in ('agents', 'agent_unit', 'dyn_prev', 'gt', 'dmap', 'drone', 'target', 'targets', 'counters'):
            # in slices of <= 256 MB of host staging: a config-5 shard is 32768 grids of 640 x 640 cells = 13.4 GB per field
            per = max(1, int(np.asarray(distinct[0][name]).nbytes))
            step = max(1, (256 << 20) // per)
            dst = self.t[name] if idx is None else torch.empty((U,) + tuple(self.t[name].shape[1:]), dtype=self.t[name].dtype, device=self.device)
            for c0 in range(0, U, step):
                arr = torch.from_numpy(np.stack([w[name] for w in distinct[c0:c0 + step]])).to(self.t[name].dtype)
                if name in ('gt', 'dmap') and self.cfg.grid_tile:
                    arr = tile_grid(arr, self.cfg.grid_tile)
                dst[c0:c0 + step].copy_(arr)
            if idx is not None:
                for c0 in range(0, B, step):                       # (the gather's temporary stays within the same 256 MB)
                    self.t[name][c0:c0 + step] = dst[idx[c0:c0 + step]]
                del dst
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
