/*
 * d2d.h — C ABI of the MI355X-native batched Drone2D step.
 *
 * The reference (smoggy-P/gym-Drone2D-ActivePerception) has no native boundary: its hot path is the
 * Python method Drone2DEnv2.step (envs/drone_v2.py:152-257) and the utils.py objects it mutates.  This
 * header is the boundary a maintainer binds instead (ctypes stub: INTEGRATION.md).  Every entry point
 * names the reference code it replaces.
 *
 * Conventions
 *   - plain C, POD structs, raw pointers; no torch / HIP types in any signature (`stream` is a
 *     hipStream_t passed as void*; NULL = the null stream).
 *   - the caller owns all memory.  libd2d_hip.so expects DEVICE pointers, liboracle (oracle/) expects
 *     HOST pointers with the same layouts.  No hidden allocation, no hidden synchronisation: every
 *     launch is asynchronous on the caller's stream.
 *   - return value: 0 on success, negative on error (-1 bad argument, -2 ABI mismatch, -3 HIP launch
 *     error, -4 unsupported configuration); d2d_last_error() returns a thread-local message.
 *     Nothing throws across the ABI.
 *   - not re-entrant on the same buffers; one host thread (process) per GPU.
 *
 * Layouts (B envs, N agents per env, W x H grid cells, L x L local map, T targets per env)
 *   - all float state is fp64, exactly the reference's Python float / np.float64 state; exported
 *     float32 views are made by the host.
 *   - grids are uint8 [B][W][H], indexed grid[i = x // scale][j = y // scale] as in utils.py:548.
 */
#ifndef D2D_H
#define D2D_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define D2D_ABI_VERSION 8

/* grid cell codes, utils.py:11-16 */
#define D2D_UNEXPLORED 0
#define D2D_OCCUPIED 1
#define D2D_UNOCCUPIED 2
#define D2D_DYNAMIC 3

/* state machine codes, utils.py:24-29 */
#define D2D_SM_WAIT_FOR_GOAL 0
#define D2D_SM_GOAL_REACHED 1
#define D2D_SM_PLANNING 2
#define D2D_SM_EXECUTING 3

/* planner_mode */
#define D2D_PLANNER_EXTERNAL 0 /* plan_ok / wp_valid / wp are inputs (host plugin or replay)          */
#define D2D_PLANNER_NOMOVE 1   /* traj_planner.py:68-76 on device: ok, no waypoint, target=(-1,-1)    */
/* d2d_plan.planner: which planner d2d_plan_stage runs between the two halves of the step */
#define D2D_PLAN_NONE 0      /* plan_ok / wp_valid / wp stay as the caller wrote them                 */
#define D2D_PLAN_PRIMITIVE 1 /* traj_planner.py:78-233: replan_check + motion-primitive A* on device  */
/* d2d_plan.gaze: which gaze policy d2d_gaze_stage runs before the step */
#define D2D_GAZE_NONE 0   /* action stays as the caller wrote it                                      */
#define D2D_GAZE_OXFORD 1 /* yaw_planner.py:41-127 on device                                          */

/* agent field planes of d2d_state.agents: [B][D2D_AF][N] */
#define D2D_AF 6
#define D2D_A_PX 0
#define D2D_A_PY 1
#define D2D_A_VX 2 /* pref_velocity (aliased with velocity under CVM, envs/drone_v2.py:178) */
#define D2D_A_VY 3
#define D2D_A_R 4
#define D2D_A_R2 5 /* radius**2 as the host's Python evaluates it (utils.py:659)            */

/* drone record d2d_state.drone: [B][D2D_DF] */
#define D2D_DF 8
#define D2D_D_X 0
#define D2D_D_Y 1
#define D2D_D_YAW 2
#define D2D_D_VX 3
#define D2D_D_VY 4
#define D2D_D_AX 5
#define D2D_D_AY 6
#define D2D_D_TGT_X 7 /* unused pad kept for 64-B records */

/* per-env int32 counters d2d_state.counters: [B][D2D_CF] */
#define D2D_CF 8
#define D2D_C_STEPS 0
#define D2D_C_FAIL 1
#define D2D_C_SM 2
#define D2D_C_TGT_NEXT 3 /* index of the next entry of the target list (list.pop(0) cursor) */
#define D2D_C_NTGT 4     /* length of the target list                                       */
#define D2D_C_TRACKED 5  /* env.tracked_agent, envs/drone_v2.py:188                        */
#define D2D_C_BUF_N 6    /* len(tracker_buffer)                                            */
#define D2D_C_BUF_TS 7   /* sum(len(tracker.ts)) over tracker_buffer, experiment.py:74     */

/* per-env uint8 flags d2d_state.flags: [B][4] */
#define D2D_F_COLLISION 0
#define D2D_F_DEADLOCK 1
#define D2D_F_FREEZING 2
#define D2D_F_DONE 3

/* Kalman record d2d_state.kf: [B][N][D2D_KF] fp64: mu[4], Sigma[16] row-major */
#define D2D_KF 20

/* stage bits for d2d_run_stages */
#define D2D_ST_FSM 1      /* state machine pre-update, envs/drone_v2.py:153-163      */
#define D2D_ST_AGENTS 2   /* Agent.step under CVM, utils.py:472-493                  */
#define D2D_ST_RAYCAST 4  /* Raycast.castRays, utils.py:593-713                      */
#define D2D_ST_DYNGRID 8  /* OccupancyGridMap.update_dynamic_grid, utils.py:527-540  */
#define D2D_ST_TRACKER 16 /* Drone2D.update_tracker, utils.py:749 -> 242-275         */
#define D2D_ST_CONTROL 32 /* brake / step_pos / step_yaw, utils.py:733-762           */
#define D2D_ST_COLLIDE 64 /* is_collide + flags + done, utils.py:764-778, env:217-235 */
#define D2D_ST_OBS 128    /* get_local_map + obs, utils.py:780-784, env:251-255      */
#define D2D_ST_PERCEIVE (D2D_ST_FSM | D2D_ST_AGENTS | D2D_ST_RAYCAST | D2D_ST_DYNGRID | D2D_ST_TRACKER)
#define D2D_ST_ACT (D2D_ST_CONTROL | D2D_ST_COLLIDE | D2D_ST_OBS)
#define D2D_ST_ALL (D2D_ST_PERCEIVE | D2D_ST_ACT)
#define D2D_ST_SKIP_DONE 256 /* modifier: envs whose flags say "done" are left untouched by the launch */

/* d2d_closed_loop `on_done`: what happens to an env whose step ended its episode (flags[D2D_F_DONE]) */
#define D2D_DONE_CONTINUE 0 /* keeps stepping (the reference's sweeps that ignore `done`)                          */
#define D2D_DONE_RESET 1    /* back to the snapshot with fresh plugin state at the start of its next step          */
#define D2D_DONE_FREEZE 2   /* stays as it ended: one episode per env, the terminal state is the result (main.py)  */

/* Numeric mirror of the reference's Params (utils.py:65-106) plus derived constants. */
/* Address space of the device pointers below.  Empty for every C / C++ user of this header (plain pointers, the ABI);
 * the HIP translation unit sets it to the global address space for its device pass, so that a pointer read out of a
 * struct in memory is known to be a global pointer (`global_*` instead of `flat_*` accesses).  Layout is unaffected. */
#ifndef D2D_AS
#define D2D_AS
#endif

typedef struct d2d_cfg {
  int32_t abi_version; /* D2D_ABI_VERSION */
  int32_t B;           /* envs in this shard */
  int32_t N;           /* agents per env: agent_number + nonzero cells of the static map (drone_v2.py:28-66) */
  int32_t W, H;        /* map_size // map_scale (utils.py:497-498); at most 32767 cells a side (-4 otherwise) */
  int32_t R;           /* rays = ceil(map_size[0] / 10) (utils.py:570,587) */
  int32_t L;           /* local map edge = 4 * (view_depth // map_scale) + 1 (drone_v2.py:133) */
  int32_t T;           /* capacity of the per-env target list */
  int32_t planner_mode;
  int32_t kf_enabled;  /* 1: Kalman trackers run on device (kf, kf_len must be set) */
  int32_t noise_rows;  /* rows of d2d_state.noise, each [B][N][2]: step t of a multi-step call (d2d_rollout, d2d_closed_loop)
                          draws from row (noise_row0 + t) % noise_rows -- the reference draws fresh normals every step
                          (utils.py:605); 0 or 1 = one row used by every step (single-step entry points always use row 0) */
  int32_t noise_row0;  /* the row the FIRST step of a multi-step call draws from (ABI 7): a caller that cuts a run into several
                          calls advances it by the steps of each call, so that the pieces draw the rows the whole run would */
  int32_t grid_tile;   /* layout of `gt` and `dmap` inside one env (ABI 7).  0: row-major [W][H], the reference's indexing
                          (utils.py:548) and the default.  16: 16 x 16-cell tiles of 256 contiguous bytes, tiles row-major over
                          (ceil(W / 16), ceil(H / 16)), cells row-major inside a tile -- cell (i, j) at byte
                          ((i / 16) * ceil(H / 16) + j / 16) * 256 + (i % 16) * 16 + j % 16; every env then takes
                          ceil(W / 16) * ceil(H / 16) * 256 bytes per grid (D2D_GRID_BYTES).  For grids of hundreds of cells a side,
                          where a 3 x 3 block or a 23-byte window row of the row-major layout costs a cache line of its own; the host
                          converts at the boundary (state.py tile_grid / untile_grid).  liboracle takes 0 only. */
  int32_t reserved2;
  double dt;           /* params.dt */
  double scale;        /* params.map_scale (x_scale == y_scale, utils.py:500-501) */
  double W_px, H_px;   /* params.map_size */
  double ray_off0;     /* -FOV/2 with FOV = radians(drone_view_range) (utils.py:575,594) */
  double ray_dth;      /* FOV / R (utils.py:594) */
  double depth;        /* drone_view_depth */
  double drone_radius; /* drone_radius */
  double yaw_rate;     /* drone_max_yaw_speed (drone_v2.py:214) */
  double max_acc;      /* drone_max_acceleration (utils.py:756-760) */
  double max_steps;    /* max_flight_time / dt (drone_v2.py:89) */
  double sigma;        /* var_cam: measurement noise scale (utils.py:605) and KF switch (utils.py:203-205) */
  double kf_lo_x, kf_hi_x, kf_lo_y, kf_hi_y; /* 10 + agent_radius, map_size - 10 - agent_radius (utils.py:236-237) */
} d2d_cfg;

/* bytes of one env's `gt` (or `dmap`) under cfg->grid_tile */
#define D2D_GRID_BYTES(cfg) ((cfg)->grid_tile ? (size_t)(((cfg)->W + 15) / 16) * (size_t)(((cfg)->H + 15) / 16) * 256 : (size_t)(cfg)->W * (size_t)(cfg)->H)

typedef struct d2d_state {
  /* ---- world state (read + written) ---- */
  double D2D_AS *agents;      /* [B][D2D_AF][N] */
  int32_t D2D_AS *agent_unit; /* [B][N]  int(radius // scale) (utils.py:533-534) */
  int32_t D2D_AS *dyn_prev;   /* [B][N][3] cell block (cx, cy, half) that may hold DYNAMIC cells written for
                          this agent last time: replaces the dynamic_idx list (utils.py:506,528-530) */
  uint8_t D2D_AS *gt;         /* [B][W][H] ground-truth grid, env.map_gt.grid_map ([B][D2D_GRID_BYTES] when cfg.grid_tile != 0) */
  uint8_t D2D_AS *dmap;       /* [B][W][H] the drone's explored map, env.drone.map.grid_map (same layout as gt) */
  double D2D_AS *drone;       /* [B][D2D_DF] */
  double D2D_AS *target;      /* [B][2] planner.target[:2] */
  double D2D_AS *targets;     /* [B][T][2] env.target_list */
  int32_t D2D_AS *counters;   /* [B][D2D_CF] */
  uint8_t D2D_AS *active;     /* [B][N] KalmanFilter.active (utils.py:182,273) */
  double D2D_AS *kf;          /* [B][N][D2D_KF] or NULL when !kf_enabled */
  int32_t D2D_AS *kf_len;     /* [B][N] len(tracker.ts) or NULL */
  /* ---- inputs of this step (read only) ---- */
  const double D2D_AS *action;    /* [B] gaze action a in [-1, 1] (drone_v2.py:152,214) */
  const uint8_t D2D_AS *plan_ok;  /* [B] planner.plan() result (drone_v2.py:197); NULL under NOMOVE */
  const uint8_t D2D_AS *wp_valid; /* [B] trajectory non-empty at step_pos (utils.py:734); NULL under NOMOVE */
  const double D2D_AS *wp;        /* [B][6] head waypoint: pos(2), vel(2), acc(2) (utils.py:735-738) */
  const double D2D_AS *noise;     /* [noise_rows][B][N][2] standard normal draws for utils.py:605, or NULL (sigma must be 0) */
  /* ---- outputs of this step (written) ---- */
  uint8_t D2D_AS *hit;        /* [B][N] OR over rays of the per-ray hit lists (utils.py:598-599) */
  int32_t D2D_AS *newly;      /* [B] newly_tracked (utils.py:606-607) */
  uint8_t D2D_AS *flags;      /* [B][4] collision, dead_lock, freezing, done */
  uint8_t D2D_AS *obs_local;  /* [B][L][L] obs['local_map'] (== obs['swep_map'], drone_v2.py:252-253) */
  float D2D_AS *obs_yaw;      /* [B] obs['yaw_angle'] */
} d2d_state;


/* ---------------------------------------------------------------------------------------------
 * Planner / gaze plugins on the device.  The reference calls them as Python
 * objects around and inside step(): `a = policy.plan(info)` before the step (experiment.py:69),
 * `planner.replan_check(drone)` + `planner.plan(drone, dt)` between perception and control
 * (envs/drone_v2.py:194-197).  d2d_plan carries their constants, their per-env state and their scratch;
 * d2d_cfg / d2d_state stay what the hot kernel takes by value.
 *
 * Constant tables are evaluated by the HOST with the very expressions the reference evaluates (numpy
 * arange / Python float arithmetic), so the device only replays IEEE operations:
 *   u_space   [nu]            Primitive.u_space (traj_planner.py:98-101)
 *   sample_t  [n_sample][2]   t, t**2 for t in np.arange(0, 2, 2 / sample_num) (traj_planner.py:175-176)
 *   traj_t    [n_ts][3]       t, t**2, 2 * t for t in np.arange(2, 0, -dt), stored in ASCENDING t (the
 *                             reference builds the list backwards and reverses it, traj_planner.py:210-216)
 *   yaw_space [n_yaw]         Oxford.v_yaw_space = np.arange(-w, w, w / 3) (yaw_planner.py:65)
 *   tobs_tab  [2][tobs_len]   row 0: 0 + dt + dt + ... (k additions), row 1: 5 + dt + dt + ...: the value of a
 *                             cell of Oxford.last_time_observed_map k calls after it was last seen / if it never
 *                             was (yaw_planner.py:48,95-97); the map itself is kept as `seen_step`
 *   pw_leaf   [pw_nleaf][4]   offset, length, first row, last row of the <= 128-element blocks numpy's pairwise summation cuts a
 *   pw_prog   [pw_nprog]      W * H array into, and the order their partial sums are added in (>= 0: push
 *                             block, -1: add the two on top) -- np.sum(view * reward), yaw_planner.py:123.
 *                             pw_tree holds the same additions grouped by tree level (independent within a
 *                             level), pw_rowleaf the block of every grid row's first cell: what the device reads
 *   acos_key_lo / acos_mask   np.arccos(q) <= half_fov for the 64 consecutive doubles q starting at the one whose
 *                             ordered bit pattern is acos_key_lo (bit i of the mask = decision for the i-th);
 *                             every q above the window is inside the cone, every q below outside.  numpy's
 *                             arccos is a SIMD routine that differs from libm by an ulp; the window is how the
 *                             host hands its own arccos to the device (yaw_planner.py:77).
 * ------------------------------------------------------------------------------------------- */
typedef struct d2d_plan {
  int32_t planner;  /* D2D_PLAN_* */
  int32_t gaze;     /* D2D_GAZE_* */
  int32_t nu;
  int32_t n_sample;
  int32_t n_ts;
  int32_t max_itr;  /* 100: the search gives up at itr >= max_itr (traj_planner.py:149) */
  int32_t traj_cap; /* waypoints per env in `traj` (>= (max_itr - 1) * n_ts can never overflow) */
  int32_t node_cap; /* search nodes per env in `nodes` */
  int32_t hash_cap; /* slots per env in `hash`, a power of two > node_cap */
  int32_t n_yaw;
  int32_t pw_nleaf;
  int32_t pw_nprog;
  int32_t tobs_len;
  int32_t pw_ntree;
  double horizon;      /* Primitive.dt = 2 (traj_planner.py:103) */
  double vmax;         /* drone_max_speed (traj_planner.py:172) */
  double safe_dist;    /* drone_radius + 10 (traj_planner.py:32) */
  double goal_tol;     /* search_threshold = 10 (traj_planner.py:106,158) */
  double agent_radius; /* params.agent_radius: KalmanFilter.radius after an archive (utils.py:184) */
  double half_fov;     /* math.radians(drone_view_range / 2) (yaw_planner.py:72) */
  double yaw_rate_max; /* drone_max_yaw_speed (yaw_planner.py:127) */
  double vmax_sq;      /* the largest double s with sqrt(s) < vmax: `norm(v) < vmax` (traj_planner.py:172) as `v.v <= vmax_sq` without a
                          square root per primitive (sqrt is correctly rounded, hence monotone).  Three states: 0 = the library finds it
                          at every search; -1.0 = "nothing passes", the value for vmax <= 0; else it must BE that threshold of `vmax` --
                          every plugin entry point checks and returns -1 on a mismatch (a stale value would silently change searches) */
  double goal_sq;      /* the largest double s with sqrt(s) <= goal_tol (traj_planner.py:158), same use; 0 = as above, else checked
                          against `goal_tol` in the same way (ABI 7) */
  int64_t acos_key_lo;
  uint64_t acos_mask;
  /* ---- constant tables (read only) ---- */
  const double D2D_AS *u_space;
  const double D2D_AS *sample_t;
  const double D2D_AS *traj_t;
  const double D2D_AS *yaw_space;
  const double D2D_AS *tobs_tab;
  const int32_t D2D_AS *pw_leaf;
  const int32_t D2D_AS *pw_prog;
  const int32_t D2D_AS *pw_tree;    /* [pw_ntree] the additions of pw_prog level by level: n_levels, root id, level_start[n_levels + 1],
                                then (dst, left, right) per addition; ids 0..pw_nleaf-1 are the blocks (device only) */
  const int32_t D2D_AS *pw_rowleaf; /* [W] the block that holds the first cell of grid row i (device only) */
  const double D2D_AS *trk_radius0; /* [B][N] tracker radii of the initial world: the reset source of trk_radius */
  /* ---- per-env plugin state (read + written) ---- */
  double D2D_AS *traj;         /* [B][traj_cap][4] planner.trajectory: position(2), velocity(2); accelerations are 0 */
  int32_t D2D_AS *traj_hdr;    /* [B][2] index of the head waypoint, number of waypoints stored (len = stored - head) */
  double D2D_AS *traj_box;     /* [B][ceil(traj_cap / 64)][4] (ABI 8) or NULL: (xmin, ymin, xmax, ymax) of the waypoints stored in slots
                           [64 c, 64 c + 64) of `traj`, written by the planner stage with every trajectory it stores.  Oxford's swept map
                           and replan_check read every REMAINING waypoint at every step (up to 1 800 x 32 B on a 6400 px map); with the
                           boxes they load only the 64-waypoint chunks that can matter -- those that reach into the view box, come
                           within a tracker's safety radius, or cross the cells this step's rays wrote -- and skip the rest, exactly
                           (a box is a bound, never an estimate).  NULL: every stage walks the whole trajectory.  Library-private
                           contents (the oracle ignores the field) */
  double D2D_AS *trk_radius;   /* [B][N] drone.trackers[k].radius (envs/drone_v2.py:46; back to agent_radius on archive) */
  uint8_t D2D_AS *trk_prev;    /* [B][N] tracker.active as the planner stage last saw it (detects the archive) */
  double D2D_AS *trk_lim;      /* [B][N] cache kept by the library: the largest s with sqrt(s) <= drone_radius + trk_radius, i.e.
                           replan_check's `norm(d) <= drone_radius + radius` (traj_planner.py:228) as `d.d <= s`; 0 = not
                           computed yet (the state of a fresh or reset plugin state) */
  int32_t D2D_AS *seen_step;   /* [B][W][H] Oxford: number of the plan() call that last saw the cell, 0 = never */
  /* ---- scratch of the search (contents meaningless between calls) ---- */
  double D2D_AS *nodes;        /* [B][node_cap * D2D_NODE_F] search nodes; the field order inside an env's block is the
                           implementation's (the oracle keeps records, the HIP library planes) */
  int32_t D2D_AS *hash;        /* [B][hash_cap] */
  void *launch_args;    /* >= D2D_LAUNCH_ARGS_BYTES of device memory where the persistent closed-loop launch parks its
                           arguments (NULL: d2d_closed_loop launches every stage of every step separately).  Every
                           d2d_closed_loop call on this d2d_plan rewrites the buffer on its stream: a d2d_plan (like the
                           state buffers) belongs to ONE stream at a time */
  /* ---- diagnostics ---- */
  int32_t D2D_AS *plan_stat;   /* [B][4] searches run, expansions of the last search, nodes of the last search,
                           capacity overflow flag (sticky; a search that overflowed reports failure) */
} d2d_plan;

/* one search node = D2D_NODE_F doubles: position(2), velocity(2), cost, total_cost, acc(2), then parent slot / itr /
 * key / state as raw bits (record offsets of the oracle; scratch, never exchanged between implementations) */
#define D2D_NODE_F 12
#define D2D_LAUNCH_ARGS_BYTES 2048
#define D2D_N_PX 0
#define D2D_N_PY 1
#define D2D_N_VX 2
#define D2D_N_VY 3
#define D2D_N_COST 4
#define D2D_N_TOTAL 5
#define D2D_N_AX 6
#define D2D_N_AY 7
#define D2D_N_LINK 8  /* int32 parent slot, int32 itr */
#define D2D_N_KEY 9   /* int64 packed index (cell x, cell y, round vx, round vy), traj_planner.py:93 */
#define D2D_N_STATE 10 /* int64: 1 open, 2 closed */

/* ---------------------------------------------------------------------------------------------
 * Entry points of libd2d_hip.so (gym-drone2d-activeperception_amd/csrc).  liboracle exports the
 * same set with the prefix d2d_oracle_ and ignores `stream`.
 * ------------------------------------------------------------------------------------------- */

/* ABI version of the loaded library; must equal D2D_ABI_VERSION. */
int d2d_abi_version(void);

/* Last error message of the calling thread ("" if none). */
const char *d2d_last_error(void);

/* One full Drone2DEnv2.step (envs/drone_v2.py:152-257) for every env of the shard, fused in one
 * launch.  Valid when the planner result does not depend on this step's perception: NOMOVE, or
 * EXTERNAL with replayed plan_ok / wp_valid / wp. */
int d2d_step(const d2d_cfg *cfg, const d2d_state *st, void *stream);

/* The first half of step(): lines 153-187 (state machine, agents, raycast, dynamic grid, trackers).
 * A host planner plugin (traj_planner.py Planner.replan_check / plan) runs between the halves. */
int d2d_perceive(const d2d_cfg *cfg, const d2d_state *st, void *stream);

/* The second half: lines 198-255 (brake / follow, yaw, collision, flags, done, observation). */
int d2d_act(const d2d_cfg *cfg, const d2d_state *st, void *stream);

/* Any subset of stages (D2D_ST_* bits) in reference order, one launch: per-stage profiling. */
int d2d_run_stages(const d2d_cfg *cfg, const d2d_state *st, uint32_t stages, void *stream);

/* `nsteps` consecutive fused steps queued back to back on the stream by ONE call (no host round trip
 * between them): the reference's inner loops that call step() back to back with host-independent inputs
 * (glob_survivability_calculator.py:31-37).  actions: [nsteps][B]; optional wp_steps: [nsteps][B][6]
 * planner heads per step (EXTERNAL planner mode; st->plan_ok / st->wp_valid stay as given), or NULL to use
 * st->wp every step; optional pin: [B][2] drone position forced before every step (env.drone.x = x;
 * env.drone.y = y), or NULL; coll_out: [nsteps][B] uint8 collision flag per step, or NULL. */
int d2d_rollout(const d2d_cfg *cfg, const d2d_state *st, int32_t nsteps, const double *actions,
                const double *wp_steps, const double *pin, uint8_t *coll_out, void *stream);

/* reset(): for every env with mask[e] != 0 copy the snapshot `init` (same layouts, same B) over the
 * live state and clear outputs (envs/drone_v2.py:259-261 re-runs __init__; the host ran it once and
 * keeps the result resident).  mask == NULL resets all. */
int d2d_reset(const d2d_cfg *cfg, const d2d_state *st, const d2d_state *init, const uint8_t *mask,
              void *stream);

/* Device restatement of the host libm tan() the reference's math.tan resolves to (utils.py:640):
 * out[i] = tan(in[i]) for |in[i]| <= 25, bit-for-bit glibc 2.35 x86-64 FMA variant.  Test hook. */
int d2d_tan_array(const double *in, double *out, int64_t n, void *stream);


/* policy.plan(info) of the gaze plugin for every env (experiment.py:69), run BEFORE the step on the state the
 * previous step left: writes st->action.  D2D_GAZE_OXFORD: yaw_planner.py:81-127 (view map of the current pose,
 * time-since-observed map, swept-trajectory reward, 6 yaw-rate candidates).  D2D_GAZE_NONE: no-op. */
int d2d_gaze_stage(const d2d_cfg *cfg, const d2d_state *st, const d2d_plan *plan, void *stream);

/* planner.replan_check(drone) + planner.plan(drone, dt) (envs/drone_v2.py:194-197) + the head waypoint
 * step_pos will consume (utils.py:733-739), run BETWEEN d2d_perceive and d2d_act: writes st->plan_ok,
 * st->wp_valid, st->wp (the caller's buffers behind those const pointers) and pops the head.
 * D2D_PLAN_PRIMITIVE: traj_planner.py:125-233.  D2D_PLAN_NONE: no-op. */
int d2d_plan_stage(const d2d_cfg *cfg, const d2d_state *st, const d2d_plan *plan, void *stream);

/* One closed-loop step with the plugins on the device: gaze -> perceive -> plan -> act, queued on `stream`
 * (the reference's `a = policy.plan(info); env.step(a)`, experiment.py:68-70).  `nsteps` of them back to back.
 * `on_done` (D2D_DONE_*): with RESET every env whose PREVIOUS step ended the episode (flags[D2D_F_DONE], so the
 * caller sees the terminal state after the call) is put back to the snapshot `init` with fresh plugin state at the
 * start of its next step, the way the reference's sweeps start the next episode (main.py:26-57); with FREEZE a
 * finished env is left exactly as its last step left it (Experiment.run stops at done, experiment.py:68-72).
 * `init` may be NULL unless on_done == D2D_DONE_RESET. */
int d2d_closed_loop(const d2d_cfg *cfg, const d2d_state *st, const d2d_plan *plan, int32_t nsteps,
                    int32_t on_done, const d2d_state *init, void *stream);

/* Clears the plugin state (trajectory, tracker radii <- plan->trk_radius0, seen map) of the envs with
 * mask[e * mask_stride] != 0 (mask == NULL: all): Experiment.__init__ builds fresh plugin objects per episode
 * (experiment.py:31-34). */
int d2d_plan_reset(const d2d_cfg *cfg, const d2d_plan *plan, const uint8_t *mask, int32_t mask_stride,
                   void *stream);

/* How the library would launch this configuration (no GPU call; for capacity planning and regression tests): fills
 * out[0] = waves (envs) per workgroup, out[1] = LDS bytes per workgroup, out[2] = workgroups of that size a CU's 160 KB of
 * LDS hold, out[3] = 1 if the specialised default-geometry kernels apply (both grids staged whole in LDS), else 0.
 * `plan` == NULL: the fused step (d2d_step / d2d_run_stages); else the persistent closed loop (d2d_closed_loop), out[0] = 0
 * when it would fall back to one launch per stage.  The persistent closed loop runs ONE env per workgroup (out[0] = 1: a workgroup
 * holds its wave slots until its slowest env has finished its steps; a CU then holds out[2] >= 16 of them at the default
 * geometry).  Returns 0, or a negative error like every entry point. */
int d2d_launch_shape(const d2d_cfg *cfg, const d2d_plan *plan, int32_t out[4]);

/* Device restatement of the host libm sin() / cos() the reference's math.sin / math.cos resolve to
 * (yaw_planner.py:71): bit-for-bit glibc 2.35 x86-64 FMA variant for |x| < 105414350.  Test hook. */
int d2d_sincos_array(const double *in, double *sin_out, double *cos_out, int64_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* D2D_H */
