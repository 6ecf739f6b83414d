"""VecDrone2DEnv — B independent Drone2D worlds stepped in lock-step on one MI355X.

Batched counterpart of the reference's `Drone2DEnv2` (envs/drone_v2.py:10-261): the same step semantics
per env, state resident in HBM, one fused HIP launch per step.  Env `i` of the batch is the world the
reference builds for `params.map_id + i` (+ `env_offset` of the shard), so results do not depend on how
the batch is sharded across GPUs.

Planner modes
  'NoMove'    traj_planner.py:68-76 runs on the device (one launch per step)
  'external'  the caller supplies plan_ok / wp_valid / wp each step (host planner plugin between
              `perceive()` and `act()`, or replayed plans with the fused `step()`)
  'Primitive' with `device_plugins=True`: traj_planner.py:78-233 runs on the device between the two halves of
              the step; with `gaze='Oxford'` yaw_planner.py:41-127 supplies the action on the device as well
              (`closed_loop()`: gaze -> perceive -> plan -> act, no host round trip)
"""
import numpy as np
import torch

from . import _abi as A
from . import host_init
from .params import with_defaults
from .state import BatchState


def _build_world(args):
    params, map_id = args
    p = with_defaults(params)
    p.map_id = map_id
    return host_init.init_world(p)


def build_worlds(params, num_envs, env_offset=0, workers=0):
    """Host construction of `num_envs` worlds: env i is the reference world for map_id + env_offset + i.
    With `workers` > 0 the (pure Python, GPU-free) construction is spread over forked processes; call
    this before the process touches the GPU."""
    p = with_defaults(params)
    jobs = [(p, p.map_id + env_offset + i) for i in range(num_envs)]
    if workers and num_envs >= 64:
        import multiprocessing as mp
        with mp.get_context('fork').Pool(workers) as pool:
            return pool.map(_build_world, jobs, chunksize=max(1, num_envs // (workers * 8)))
    return [_build_world(j) for j in jobs]


def build_worlds_of(params_list, workers=0):
    """One world per Params of the list (each with its own map_id and agent settings: the survivability sweep's 540 settings).
    `workers` as in build_worlds: forked processes, only before the process touches the GPU."""
    jobs = [(with_defaults(p), p.map_id) for p in params_list]
    if workers and len(jobs) >= 64:
        import multiprocessing as mp
        with mp.get_context('fork').Pool(workers) as pool:
            return pool.map(_build_world, jobs, chunksize=max(1, len(jobs) // (workers * 4)))
    return [_build_world(j) for j in jobs]


class VecDrone2DEnv:
    def __init__(self, params, num_envs, device='cuda:0', planner=None, env_offset=0, backend=None,
                 kf_enabled=True, worlds=None, device_plugins=False, gaze=None, grid_layout=None):
        self.params = with_defaults(params)
        self.num_envs = int(num_envs)
        self.env_offset = int(env_offset)
        planner = planner if planner is not None else self.params.planner
        self.planner_mode = A.PLANNER_NOMOVE if planner == 'NoMove' else A.PLANNER_EXTERNAL
        if backend is None:
            from ._lib import HipBackend       # raises if the HIP library or the GPU is missing
            backend = HipBackend(device)
        self.backend = backend
        self.device = torch.device(backend.device)
        if worlds is None:
            worlds = build_worlds(self.params, self.num_envs, self.env_offset, workers=0)
        N = worlds[0]['N'] if worlds else 0
        T = worlds[0]['T'] if worlds else 1
        if any(w['N'] != N for w in worlds):
            raise ValueError('all envs of a batch must have the same number of agents')
        # device layout of the two grids (include/d2d.h d2d_cfg.grid_tile): 'rowmajor' = the reference's [W][H]; 'tiled' = 16 x 16-cell
        # tiles.  Default: tiled on the HIP backend for grids above 256 x 256 cells (BASELINE config 5: a 3 x 3 block or a 23-byte
        # window row of a 640-cell row-major grid costs a cache line each), row-major otherwise; `state.logical()` / the facade
        # proxies always speak [W][H].
        W, H = self.params.map_size[0] // self.params.map_scale, self.params.map_size[1] // self.params.map_scale
        if grid_layout is None:
            grid_layout = 'tiled' if (W * H > 256 * 256 and getattr(backend, 'supports_tiled_grids', False)) else 'rowmajor'
        if grid_layout not in ('rowmajor', 'tiled'):
            raise ValueError(f'grid_layout {grid_layout!r}: rowmajor or tiled')
        self.cfg = host_init.derive_cfg(self.params, B=self.num_envs, N=N, T=T, planner_mode=self.planner_mode,
                                        kf_enabled=kf_enabled, grid_tile=16 if grid_layout == 'tiled' else 0)
        self.state = BatchState(self.cfg, self.device)
        self.state.load_worlds(worlds)
        if worlds:
            from .state import distinct_worlds
            distinct, index = distinct_worlds(worlds)
            self.tracker_radius = torch.from_numpy(np.stack([w['tracker_radius'] for w in distinct]))[torch.as_tensor(index)]
        else:
            self.tracker_radius = None
        self.init_state = self.state.clone_world()
        self._st = self.state.struct()
        self._init_st = self.init_state.struct()
        self.reward = torch.zeros(self.num_envs, dtype=torch.float32, device=self.device)   # drone_v2.py:257
        self.plugins = None
        if device_plugins:
            from .device_plugins import PluginState
            gaze = gaze if gaze is not None else self.params.gaze_method
            if planner not in ('Primitive', 'NoMove') or gaze not in ('Oxford', 'Rotating', 'NoControl', 'external', None):
                raise NotImplementedError(f'device plugins: planner {planner!r} / gaze {gaze!r} '
                                          '(device: Primitive, NoMove / Oxford, Rotating, NoControl)')
            self.plugins = PluginState(self.params, self.cfg, self.device, self.tracker_radius.numpy(),
                                       planner=planner, gaze=gaze or 'external')
            self._plan = self.plugins.struct()
            # the constant policies of the reference need no kernel: Rotating.plan -> 1 (yaw_planner.py:136-142),
            # NoControl.plan -> 0 (:10-16); the action stays resident
            if gaze == 'Rotating':
                self.state.action.fill_(1.0)
            elif gaze == 'NoControl':
                self.state.action.fill_(0.0)

    # ------------------------------------------------------------------ gym-like surface (batched)
    @property
    def N(self):
        return self.cfg.N

    def reset(self, mask=None):
        """envs/drone_v2.py:259-261: back to the seeded initial world (all envs, or those in `mask`)."""
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
        self.backend.reset(self.cfg, self._st, self._init_st, mask)
        self.reset_plugins(mask)
        return {}

    def _set_action(self, actions):
        a = torch.as_tensor(actions, dtype=torch.float64)
        self.state.action.copy_(a.reshape(-1).expand(self.num_envs) if a.numel() == 1 else a.reshape(self.num_envs),
                                non_blocking=True)

    def set_plan(self, plan_ok, wp_valid, wp):
        """External planner result for this step (traj_planner.py Planner.plan + trajectory head)."""
        self.state.plan_ok.copy_(torch.as_tensor(plan_ok, dtype=torch.uint8).reshape(self.num_envs))
        self.state.wp_valid.copy_(torch.as_tensor(wp_valid, dtype=torch.uint8).reshape(self.num_envs))
        self.state.wp.copy_(torch.as_tensor(wp, dtype=torch.float64).reshape(self.num_envs, 6))

    def set_noise(self, noise):
        """Standard-normal draws for the measurements (utils.py:605); required when var_cam != 0 (the reference takes them
        from np.random in agent order).  [B, N, 2]: the draws of the next step (every step of a multi-step call would reuse
        them); [T, B, N, 2]: a run of multi-step calls (rollout() / closed_loop()) draws row after row, step t of the run from
        row t % T, as the reference draws fresh normals every step -- also when the run is cut into several calls
        (d2d_cfg.noise_row0 is advanced by the steps of every call; set_noise() starts again at row 0)."""
        n = torch.as_tensor(noise, dtype=torch.float64)
        rows = n.shape[0] if n.dim() == 4 else 1
        n = n.reshape(rows, self.num_envs, self.cfg.N, 2).to(self.device).contiguous()
        self.state.noise = n
        self._st.noise = n.data_ptr()
        self.cfg.noise_rows = rows
        self.cfg.noise_row0 = 0

    def _advance_noise(self, nsteps):
        if self.cfg.noise_rows > 1:
            self.cfg.noise_row0 = (self.cfg.noise_row0 + int(nsteps)) % self.cfg.noise_rows

    def step(self, actions):
        """One fused Drone2DEnv2.step for every env.  Returns (obs, reward, done, info) of tensors."""
        self._set_action(actions)
        self.backend.step(self.cfg, self._st)
        return self._result()

    def perceive(self):
        """First half of step() (lines 153-187); a host planner plugin runs after this."""
        self.backend.perceive(self.cfg, self._st)

    def act(self, actions):
        """Second half of step() (lines 198-255)."""
        self._set_action(actions)
        self.backend.act(self.cfg, self._st)
        return self._result()

    def rollout(self, actions, pin=None, collisions=False, streams=1):
        """`actions`: [T, B] gaze actions; T fused steps queued by one call (survivability-style sweeps,
        glob_survivability_calculator.py:31-37).  `pin`: [B, 2] drone position forced before each step.
        `streams` > 1 cuts the batch into that many independent sub-batches whose T-step chains run on their own
        HIP streams (envs are independent, so the chains need no ordering between them; on MI355X two chains of
        2048 envs finish ~16 % sooner than one of 4096 because kernel tails and launch gaps overlap)."""
        actions = torch.as_tensor(actions, dtype=torch.float64, device=self.device).contiguous()
        T = actions.shape[0]
        assert actions.shape == (T, self.num_envs)
        if pin is not None:
            pin = torch.as_tensor(pin, dtype=torch.float64, device=self.device).contiguous()
        coll = torch.empty((T, self.num_envs), dtype=torch.uint8, device=self.device) if collisions else None
        S = max(1, min(int(streams), self.num_envs))
        if S == 1 or self.device.type != 'cuda' or (self.state.noise is not None and self.cfg.noise_rows > 1):
            self.backend.rollout(self.cfg, self._st, T, actions, pin, coll)
            self._advance_noise(T)
            return coll
        # sub-batch i owns envs [lo, hi): its own cfg (B = hi - lo) and state struct (every pointer offset by lo)
        import copy
        cur = torch.cuda.current_stream(self.device)
        bounds = [(self.num_envs * i) // S for i in range(S + 1)]
        subs = []
        for i in range(S):
            lo, hi = bounds[i], bounds[i + 1]
            if hi == lo:
                continue
            cfg = copy.copy(self.cfg)
            cfg.B = hi - lo
            st = A.State()
            for name in A.STATE_FIELDS:
                t = self.state.noise if name == 'noise' else self.state.t.get(name)
                base = getattr(self._st, name)
                stride = 0 if t is None else (t.stride(1) if name == 'noise' else t.stride(0))   # noise: [rows][B][N][2]
                setattr(st, name, None if (t is None or not base) else base + lo * stride * t.element_size())
            # the step-t row of a sub-batch is not contiguous in [T, B]: give each sub-batch its own copies (made on the
            # current stream, like everything the caller queued before this call)
            a_i = actions[:, lo:hi].contiguous()
            p_i = None if pin is None else pin[lo:hi].contiguous()
            # every step writes its whole row of collision flags: no fill, so nothing on the current stream can land
            # after a side stream's first step
            c_i = torch.empty((T, hi - lo), dtype=torch.uint8, device=self.device) if collisions else None
            subs.append((lo, hi, cfg, st, a_i, p_i, c_i, torch.cuda.Stream(self.device)))
        # the side streams start after EVERYTHING queued so far on the current stream, the per-sub-batch copies included
        ready = torch.cuda.Event()
        ready.record(cur)
        for lo, hi, cfg, st, a_i, p_i, c_i, stream in subs:
            stream.wait_event(ready)
            with torch.cuda.stream(stream):
                self.backend.rollout(cfg, st, T, a_i, p_i, c_i)
            for t_ in (a_i, p_i, c_i):       # allocated on the current stream, consumed on `stream`
                if t_ is not None:
                    t_.record_stream(stream)
        for lo, hi, cfg, st, a_i, p_i, c_i, stream in subs:
            cur.wait_stream(stream)
            if collisions:
                coll[:, lo:hi] = c_i
        return coll

    def _result(self):
        s = self.state
        obs = {'local_map': s.obs_local.unsqueeze(1), 'swep_map': s.obs_local.unsqueeze(1),   # drone_v2.py:251-255
               'yaw_angle': s.obs_yaw.unsqueeze(1)}
        done = s.flags[:, A.F_DONE].bool()
        info = {'collision_flag': s.flags[:, A.F_COLLISION], 'dead_lock_flag': s.flags[:, A.F_DEADLOCK],
                'freezing_flag': s.flags[:, A.F_FREEZING], 'state_machine': s.counters[:, A.C_SM],
                'flight_time': s.counters[:, A.C_STEPS].double() * self.cfg.dt,
                'tracked_agent': s.counters[:, A.C_TRACKED], 'newly_tracked': s.newly, 'hit': s.hit}
        return obs, self.reward, done, info

    def closed_loop(self, nsteps=1, auto_reset=False, freeze_done=False):
        """`nsteps` reference-style steps with the plugins on the device: a = Oxford.plan(info); perceive;
        Primitive.replan_check + plan; act (experiment.py:68-70).  `auto_reset`: an env whose episode ended restarts
        from its seeded world with fresh plugin state at its next step.  `freeze_done`: an env whose episode ended
        stays as it ended (one episode per env; `episode_stats()` then holds one CSV row per env)."""
        if self.plugins is None:
            raise RuntimeError('closed_loop() needs device_plugins=True')
        mode = A.DONE_RESET if auto_reset else (A.DONE_FREEZE if freeze_done else A.DONE_CONTINUE)
        self.backend.closed_loop(self.cfg, self._st, self._plan, int(nsteps), mode,
                                 self._init_st if auto_reset else None)
        self._advance_noise(nsteps)
        return self._result()

    def reset_plugins(self, mask=None):
        if self.plugins is not None:
            if mask is not None:
                mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            self.backend.plan_reset(self.cfg, self._plan, mask, 1)

    def sync(self):
        self.backend.sync()

    # ------------------------------------------------------------------ episode statistics (CSV row, experiment.py:73-103)
    def episode_stats(self):
        """Per-env int64 [B, 8]: steps, success, static collision, dynamic collision, freezing, dead lock,
        grid discovered, agents tracked."""
        s = self.state
        c = s.counters.long()
        f = s.flags.long()
        disc = (s.dmap != 0).flatten(1).sum(1)
        return torch.stack([c[:, A.C_STEPS], (c[:, A.C_SM] == A.SM_GOAL_REACHED).long(), (f[:, 0] == 1).long(),
                            (f[:, 0] == 2).long(), f[:, 2], f[:, 1], disc, c[:, A.C_BUF_N]], dim=1)
