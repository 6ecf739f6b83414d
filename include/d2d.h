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

#define D2D_ABI_VERSION 4

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

/* Numeric mirror of the reference's Params (utils.py:65-106) plus derived constants. */
typedef struct d2d_cfg {
  int32_t abi_version; /* D2D_ABI_VERSION */
  int32_t B;           /* envs in this shard */
  int32_t N;           /* agents per env: agent_number + nonzero cells of the static map (drone_v2.py:28-66) */
  int32_t W, H;        /* map_size // map_scale (utils.py:497-498) */
  int32_t R;           /* rays = ceil(map_size[0] / 10) (utils.py:570,587) */
  int32_t L;           /* local map edge = 4 * (view_depth // map_scale) + 1 (drone_v2.py:133) */
  int32_t T;           /* capacity of the per-env target list */
  int32_t planner_mode;
  int32_t kf_enabled;  /* 1: Kalman trackers run on device (kf, kf_len must be set) */
  int32_t reserved0;
  int32_t reserved1;
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

typedef struct d2d_state {
  /* ---- world state (read + written) ---- */
  double *agents;      /* [B][D2D_AF][N] */
  int32_t *agent_unit; /* [B][N]  int(radius // scale) (utils.py:533-534) */
  int32_t *dyn_prev;   /* [B][N][3] cell block (cx, cy, half) that may hold DYNAMIC cells written for
                          this agent last time: replaces the dynamic_idx list (utils.py:506,528-530) */
  uint8_t *gt;         /* [B][W][H] ground-truth grid, env.map_gt.grid_map */
  uint8_t *dmap;       /* [B][W][H] the drone's explored map, env.drone.map.grid_map */
  double *drone;       /* [B][D2D_DF] */
  double *target;      /* [B][2] planner.target[:2] */
  double *targets;     /* [B][T][2] env.target_list */
  int32_t *counters;   /* [B][D2D_CF] */
  uint8_t *active;     /* [B][N] KalmanFilter.active (utils.py:182,273) */
  double *kf;          /* [B][N][D2D_KF] or NULL when !kf_enabled */
  int32_t *kf_len;     /* [B][N] len(tracker.ts) or NULL */
  /* ---- inputs of this step (read only) ---- */
  const double *action;    /* [B] gaze action a in [-1, 1] (drone_v2.py:152,214) */
  const uint8_t *plan_ok;  /* [B] planner.plan() result (drone_v2.py:197); NULL under NOMOVE */
  const uint8_t *wp_valid; /* [B] trajectory non-empty at step_pos (utils.py:734); NULL under NOMOVE */
  const double *wp;        /* [B][6] head waypoint: pos(2), vel(2), acc(2) (utils.py:735-738) */
  const double *noise;     /* [B][N][2] standard normal draws for utils.py:605, or NULL (sigma must be 0) */
  /* ---- outputs of this step (written) ---- */
  uint8_t *hit;        /* [B][N] OR over rays of the per-ray hit lists (utils.py:598-599) */
  int32_t *newly;      /* [B] newly_tracked (utils.py:606-607) */
  uint8_t *flags;      /* [B][4] collision, dead_lock, freezing, done */
  uint8_t *obs_local;  /* [B][L][L] obs['local_map'] (== obs['swep_map'], drone_v2.py:252-253) */
  float *obs_yaw;      /* [B] obs['yaw_angle'] */
} d2d_state;

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

#ifdef __cplusplus
}
#endif
#endif /* D2D_H */
