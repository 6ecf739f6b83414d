// d2d_hip.hip — the batched Drone2D step for MI355X (gfx950 / CDNA4) and its C ABI (include/d2d.h).
//
// Mapping: ONE WAVEFRONT (64 lanes) PER ENV, 4 envs per 256-thread workgroup, no inter-wave
// communication at all (envs are independent), so there is no __syncthreads() anywhere: every
// hand-off is lane -> lane inside one wave through LDS or the env's own global records.
//   agents / trackers / dynamic grid : lane = agent  (N > 64 loops)
//   raycast                          : lane = ray    (R > 64 loops); agents that can possibly be
//                                      hit are compacted into LDS with __ballot + popcount, the
//                                      ground-truth window the rays can reach is staged in LDS
//   collision                        : lane = probe / agent, __any / __ballot reduction
//   observation                      : lanes sweep the L x L crop, rows contiguous in memory
// There is no dense contraction on this path, hence no MFMA.  Arithmetic is fp64 in the reference's
// own operation order (compiled with -ffp-contract=off; the few fused multiply-adds are the ones the
// reference's runtime performs: libm tan, OpenBLAS dgemv), which is what makes the integer outputs
// bit-exact.  Stage -> reference map: see include/d2d.h D2D_ST_*.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/d2d.h"

#define D2D_TAN_QUAL __device__ __forceinline__
#define D2D_TAN_TBL_QUAL __device__ const
#include "d2d_tan.h"

#define WAVE 64
#define WAVES_PER_BLOCK 4

namespace {

// ------------------------------------------------------------------------------------------------
// small exact helpers
// ------------------------------------------------------------------------------------------------

// Python / numpy `int(v // s)` for integer-valued s > 0: the exact mathematical floor (CPython
// float_floor_div is exact whenever the quotient times s is representable, which it is here).
__device__ __forceinline__ int cell_of(double v, double s) {
  double q = floor(v / s);
  double r = __builtin_fma(-q, s, v);
  if (r < 0.0) q -= 1.0;
  else if (r >= s) q += 1.0;
  return (int)q;
}

// Exact fmod(a, b) for a >= 0, b > 0 (a / b far below 2^53): the true remainder is representable, so
// once the quotient is right the fused multiply-add returns it without rounding.  No loop: a wild
// input (inf / NaN / huge yaw written by a caller) must not be able to hang a wave.
__device__ __forceinline__ double fmod_pos(double a, double b) {
  double n = trunc(a / b);
  double m = __builtin_fma(-n, b, a);
  if (m < 0.0) m = __builtin_fma(-(n - 1.0), b, a);
  else if (m >= b) m = __builtin_fma(-(n + 1.0), b, a);
  return m;
}

// Python float `a % 360.0` (utils.py:743): fmod, then the sign fix-up with one rounded add.
__device__ __forceinline__ double py_mod360(double a) {
  const double m360 = 360.0;
  double m = copysign(fmod_pos(fabs(a), m360), a);
  if (m != 0.0) {
    if (m < 0.0) m += m360;
  } else {
    m = 0.0;
  }
  return m;
}

// lane -> lane hand-off inside one wave (LDS and this env's global records)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ int wave_sum(int v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}

struct LdsView {
  double *cx, *cy, *cr2;  // compacted candidate agents
  int *cidx;
  unsigned char *hit;  // [N]
  unsigned char *gtw;  // staged ground-truth window
};

struct Geom {
  int ncap;   // N rounded up to a multiple of 4
  int reach;  // cells a ray can travel from the drone cell
  int ws;     // window edge = 2 * reach + 1
  int wave_bytes;
};

__host__ __device__ inline Geom make_geom(const d2d_cfg &c) {
  Geom g;
  g.ncap = (c.N + 3) & ~3;
  if (g.ncap < 4) g.ncap = 4;
  g.reach = (int)((c.depth + 1.5 * (c.scale - 1.0)) / c.scale) + 2;
  g.ws = 2 * g.reach + 1;
  g.wave_bytes = (29 * g.ncap + g.ws * g.ws + 15) & ~15;
  return g;
}

struct EnvRegs {  // lane-uniform per-env scalars carried in registers across the fused stages
  double x, y, yaw, vx, vy, ax, ay;
  double tx, ty;
  int steps, fail, sm, tnext, ntgt, tracked, bufn, bufts;
};

// ------------------------------------------------------------------------------------------------
// stages (reference order, see include/d2d.h)
// ------------------------------------------------------------------------------------------------

// envs/drone_v2.py:153-163
__device__ __forceinline__ void st_fsm(const d2d_cfg &c, const d2d_state &s, int e, EnvRegs &r) {
  r.steps += 1;
  if (r.sm == D2D_SM_GOAL_REACHED) r.sm = D2D_SM_WAIT_FOR_GOAL;
  if (r.sm == D2D_SM_WAIT_FOR_GOAL) {
    if (r.tnext < r.ntgt) {
      const double *tl = s.targets + ((size_t)e * c.T + r.tnext) * 2;
      r.tx = tl[0];
      r.ty = tl[1];
      r.tnext += 1;
    }
    r.sm = D2D_SM_PLANNING;
  }
}

// envs/drone_v2.py:176-179 + utils.py:472-493
__device__ __forceinline__ void st_agents(const d2d_cfg &c, const d2d_state &s, int e, int lane) {
  const int N = c.N;
  double *ag = s.agents + (size_t)e * D2D_AF * N;
  const double cs = 0x1.bb67ae8584cabp-1, sn = 0x1.fffffffffffffp-2;  // cos(pi/6), sin(pi/6)
  for (int k = lane; k < N; k += WAVE) {
    const double px = ag[D2D_A_PX * N + k], py = ag[D2D_A_PY * N + k];
    const double velx = ag[D2D_A_VX * N + k], vely = ag[D2D_A_VY * N + k];
    const double rr = ag[D2D_A_R * N + k];
    const double nx = px + velx * c.dt, ny = py + vely * c.dt;
    bool aliased = true;
    double pvx = velx, pvy = vely;
    if (sqrt(velx * velx + vely * vely) <= 5.0) {
      // numpy 2x2 @ 2x1 (OpenBLAS dgemv): fma(M[r][0], v0, M[r][1] * v1); see oracle/d2d_oracle.c
      const double rx = __builtin_fma(cs, velx, (-sn) * vely);
      const double ry = __builtin_fma(sn, velx, cs * vely);
      pvx = rx;
      pvy = ry;
      aliased = false;
    }
    if (nx < c.scale + rr) pvx = fabs(pvx);
    else if (nx > c.W_px - c.scale - rr) pvx = -fabs(pvx);
    if (ny < c.scale + rr) pvy = fabs(pvy);
    else if (ny > c.H_px - c.scale - rr) pvy = -fabs(pvy);
    const double ux = aliased ? pvx : velx, uy = aliased ? pvy : vely;
    ag[D2D_A_PX * N + k] = px + ux * c.dt;
    ag[D2D_A_PY * N + k] = py + uy * c.dt;
    ag[D2D_A_VX * N + k] = pvx;
    ag[D2D_A_VY * N + k] = pvy;
  }
}

// utils.py:612-618
__device__ __forceinline__ double positive_angle(double a) {
  const double two_pi = 0x1.921fb54442d18p+2;  // math.pi * 2
  a = copysign(fmod_pos(fabs(a), two_pi), a);
  if (a < 0.0) a += two_pi;
  return a;
}

// utils.py:593-609, 620-713
__device__ __forceinline__ void st_raycast(const d2d_cfg &c, const d2d_state &s, int e, int lane,
                                           const Geom &g, const LdsView &L, EnvRegs &r) {
  const int N = c.N, W = c.W, H = c.H;
  const double *ag = s.agents + (size_t)e * D2D_AF * N;
  const unsigned char *gt = s.gt + (size_t)e * W * H;
  unsigned char *dm = s.dmap + (size_t)e * W * H;
  const double x0 = r.x, y0 = r.y;
  const double ss = c.scale - 1.0;  // x_step_size, utils.py:621

  // ---- cull: only agents within reach of some ray sample can pass the circle test of :659 ----
  // a sample is < depth + sqrt(2) * ss from the drone, so |agent - drone| <= radius + that bound
  int ncand = 0;
  for (int k0 = 0; k0 < N; k0 += WAVE) {
    const int k = k0 + lane;
    bool cand = false;
    double px = 0, py = 0, r2 = 0;
    if (k < N) {
      px = ag[D2D_A_PX * N + k];
      py = ag[D2D_A_PY * N + k];
      r2 = ag[D2D_A_R2 * N + k];
      const double rr = ag[D2D_A_R * N + k];
      const double lim = fabs(rr) + c.depth + 1.5 * ss + 2.0;
      const double dx = px - x0, dy = py - y0;
      cand = (dx * dx + dy * dy <= lim * lim);
      L.hit[k] = 0;
    }
    const unsigned long long m = __ballot(cand);
    if (cand) {
      const int slot = ncand + __popcll(m & ((1ull << lane) - 1ull));
      L.cx[slot] = px;
      L.cy[slot] = py;
      L.cr2[slot] = r2;
      L.cidx[slot] = k;
    }
    ncand += __popcll(m);
  }

  // ---- stage the ground-truth window the rays can reach (LDS tile, OOB = wall) ----
  const int ci0 = cell_of(x0, c.scale) - g.reach, cj0 = cell_of(y0, c.scale) - g.reach;
  for (int idx = lane; idx < g.ws * g.ws; idx += WAVE) {
    const int wi = idx / g.ws, wj = idx - wi * g.ws;
    const int i = ci0 + wi, j = cj0 + wj;
    L.gtw[idx] = (i >= 0 && i < W && j >= 0 && j < H) ? gt[(size_t)i * H + j] : (unsigned char)D2D_OCCUPIED;
  }
  wave_sync();

  const double rad90 = 0x1.921fb54442d18p+0, rad270 = 0x1.2d97c7f3321d2p+2;  // radians(90), radians(270)
  const double pi_ = 0x1.921fb54442d18p+1;
  const double player_angle = 0x1.921fb54442d18p+2 - r.yaw * 0x1.1df46a2529d39p-6;  // pi*2 - radians(yaw)
  const double depth2 = c.depth * c.depth;

  for (int i0 = 0; i0 < c.R; i0 += WAVE) {
    const int i = i0 + lane;
    bool alive = (i < c.R);
    const double ang = positive_angle(player_angle + (c.ray_off0 + c.ray_dth * (double)i));
    const bool faced_right = (ang < rad90 || ang > rad270);
    const bool faced_up = (ang > pi_);
    double slope = d2d_tan(ang);
    double xs, ys;
    if (fabs(slope) > 1.0) {
      slope = 1.0 / slope;
      ys = faced_up ? -ss : ss;
      xs = ys * slope;
    } else {
      xs = faced_right ? ss : -ss;
      ys = xs * slope;
    }
    double x = x0, y = y0;
    alive = alive && (0.0 < x && x < c.W_px && 0.0 < y && y < c.H_px);
    // every sample advances >= ss along the major axis and stops past `depth`, so the march ends
    // after ceil(depth / ss) + 1 samples; `budget` is a belt-and-braces bound every wave reaches
    int budget = (int)(c.depth / ss) + 4;
    while (__any(alive) && budget-- > 0) {
      if (alive) {
        bool any = false;
        for (int q = 0; q < ncand; ++q) {
          const double dx = L.cx[q] - x, dy = L.cy[q] - y;
          if (dx * dx + dy * dy <= L.cr2[q]) {
            L.hit[L.cidx[q]] = 1;
            any = true;
          }
        }
        if (any) {
          alive = false;
        } else {
          const int ci = cell_of(x, c.scale), cj = cell_of(y, c.scale);
          const int wi = ci - ci0, wj = cj - cj0;
          // the window covers every reachable cell; the guard only protects the LDS tile
          const unsigned char wall =
              (wi >= 0 && wi < g.ws && wj >= 0 && wj < g.ws) ? L.gtw[wi * g.ws + wj] : gt[(size_t)ci * H + cj];
          const double dist = (x - x0) * (x - x0) + (y - y0) * (y - y0);
          if (wall == D2D_OCCUPIED || dist >= depth2) {
            if (wall == D2D_OCCUPIED) dm[(size_t)ci * H + cj] = D2D_OCCUPIED;
            alive = false;
          } else {
            dm[(size_t)ci * H + cj] = D2D_UNOCCUPIED;
            x = x + xs;
            y = y + ys;
            alive = (0.0 < x && x < c.W_px && 0.0 < y && y < c.H_px);
          }
        }
      }
    }
  }
  wave_sync();

  // OR over rays happened in LDS; newly_tracked = #{hit and not active}, utils.py:603-607
  int newly = 0;
  for (int k0 = 0; k0 < N; k0 += WAVE) {
    const int k = k0 + lane;
    bool nw = false;
    if (k < N) {
      const unsigned char h = L.hit[k];
      s.hit[(size_t)e * N + k] = h;
      nw = h && !s.active[(size_t)e * N + k];
    }
    newly += __popcll(__ballot(nw));
  }
  if (lane == 0) s.newly[e] = newly;
  r.tracked += newly;
}

// utils.py:527-540 (dynamic_idx == the DYNAMIC cells, all inside the blocks of dyn_prev)
__device__ __forceinline__ void st_dyngrid(const d2d_cfg &c, const d2d_state &s, int e, int lane) {
  const int N = c.N, W = c.W, H = c.H;
  const double *ag = s.agents + (size_t)e * D2D_AF * N;
  unsigned char *gt = s.gt + (size_t)e * W * H;
  int *prev = s.dyn_prev + (size_t)e * N * 3;
  for (int k = lane; k < N; k += WAVE) {
    const int cx = prev[3 * k], cy = prev[3 * k + 1], u = prev[3 * k + 2];
    const int i1 = min(cx + u + 1, W), j1 = min(cy + u + 1, H);
    for (int i = max(cx - u, 0); i < i1; ++i)
      for (int j = max(cy - u, 0); j < j1; ++j)
        if (gt[(size_t)i * H + j] == D2D_DYNAMIC) gt[(size_t)i * H + j] = D2D_UNOCCUPIED;
  }
  wave_sync();  // every clear lands before any set (the reference clears all, then sets all)
  for (int k = lane; k < N; k += WAVE) {
    const int u = s.agent_unit[(size_t)e * N + k];
    const int cx = cell_of(ag[D2D_A_PX * N + k], c.scale), cy = cell_of(ag[D2D_A_PY * N + k], c.scale);
    const int i1 = min(cx + u + 1, W), j1 = min(cy + u + 1, H);
    for (int i = max(cx - u, 0); i < i1; ++i)
      for (int j = max(cy - u, 0); j < j1; ++j)
        if (gt[(size_t)i * H + j] != D2D_OCCUPIED) gt[(size_t)i * H + j] = D2D_DYNAMIC;
    prev[3 * k] = cx;
    prev[3 * k + 1] = cy;
    prev[3 * k + 2] = u;
  }
}

// ---- Kalman trackers, utils.py:172-275 ----
__device__ __forceinline__ void kf_reset(double *kf) {
#pragma unroll
  for (int i = 0; i < D2D_KF; ++i) kf[i] = 0.0;
  kf[4 + 0] = 1.0;
  kf[4 + 5] = 1.0;
  kf[4 + 10] = 10.0;
  kf[4 + 15] = 10.0;
}

__device__ __forceinline__ void mat4_mul(const double *A, const double *B, double *C) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) acc += A[4 * i + k] * B[4 * k + j];
      C[4 * i + j] = acc;
    }
}

// utils.py:605 + 749-753 + 242-275 (lane = tracker slot)
__device__ __forceinline__ void st_tracker(const d2d_cfg &c, const d2d_state &s, int e, int lane, EnvRegs &r) {
  const int N = c.N;
  const double *ag = s.agents + (size_t)e * D2D_AF * N;
  int arch_n = 0, arch_ts = 0;
  for (int k0 = 0; k0 < N; k0 += WAVE) {
    const int k = k0 + lane;
    if (k < N) {
      const bool has_z = s.hit[(size_t)e * N + k] != 0;
      unsigned char act = s.active[(size_t)e * N + k];
      if (!c.kf_enabled) {
        if (has_z) s.active[(size_t)e * N + k] = 1;
      } else if (act || has_z) {
        double *gk = s.kf + ((size_t)e * N + k) * D2D_KF;
        int len = s.kf_len[(size_t)e * N + k];
        double zx = ag[D2D_A_PX * N + k], zy = ag[D2D_A_PY * N + k];
        if (s.noise) {
          zx = zx + c.sigma * s.noise[((size_t)e * N + k) * 2];
          zy = zy + c.sigma * s.noise[((size_t)e * N + k) * 2 + 1];
        }
        double kf[D2D_KF];
#pragma unroll
        for (int i = 0; i < D2D_KF; ++i) kf[i] = gk[i];
        double *mu = kf, *S = kf + 4;
        if (act) {
          // predict(), utils.py:225-240
          const double F[16] = {1, 0, 0.1, 0, 0, 1, 0, 0.1, 0, 0, 1, 0, 0, 0, 0, 1};
          const double Ft[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0.1, 0, 1, 0, 0, 0.1, 0, 1};
          const double qn = (c.sigma != 0.0) ? 0.1 : 0.001;
          double m2[4], FS[16], P[16];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += F[4 * i + j] * mu[j];
            m2[i] = acc;
          }
          mat4_mul(F, S, FS);
          mat4_mul(FS, Ft, P);
#pragma unroll
          for (int i = 0; i < 4; ++i) P[5 * i] += qn;
#pragma unroll
          for (int i = 0; i < 4; ++i) mu[i] = m2[i];
#pragma unroll
          for (int i = 0; i < 16; ++i) S[i] = P[i];
          len += 1;
          if (P[0] >= 150.0 || !(c.kf_lo_x < m2[0] && m2[0] < c.kf_hi_x) || !(c.kf_lo_y < m2[1] && m2[1] < c.kf_hi_y)) {
            arch_n += 1;  // archived copy -> tracker_buffer
            arch_ts += len;
            kf_reset(kf);
            len = 1;
            act = 0;
          }
          if (has_z) {  // update, utils.py:249-260 (also runs on the freshly reset filter)
            const double a = c.sigma + S[0], b = S[1], cc = S[4], d = c.sigma + S[5];
            const double det = a * d - b * cc;
            const double i00 = d / det, i01 = -b / det, i10 = -cc / det, i11 = a / det;
            double K[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              K[2 * i] = S[4 * i] * i00 + S[4 * i + 1] * i10;
              K[2 * i + 1] = S[4 * i] * i01 + S[4 * i + 1] * i11;
            }
            const double rx = zx - mu[0], ry = zy - mu[1];
            double IKH[16], P2[16];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int j = 0; j < 4; ++j) IKH[4 * i + j] = (i == j ? 1.0 : 0.0) - (j < 2 ? K[2 * i + j] : 0.0);
            mat4_mul(IKH, S, P2);
#pragma unroll
            for (int i = 0; i < 4; ++i) mu[i] = mu[i] + (K[2 * i] * rx + K[2 * i + 1] * ry);
#pragma unroll
            for (int i = 0; i < 16; ++i) S[i] = P2[i];
          }
        } else {  // first sighting, utils.py:263-273
          kf_reset(kf);
          len = 1;
          mu[0] = zx;
          mu[1] = zy;
          act = 1;
        }
#pragma unroll
        for (int i = 0; i < D2D_KF; ++i) gk[i] = kf[i];
        s.kf_len[(size_t)e * N + k] = len;
        s.active[(size_t)e * N + k] = act;
      }
    }
  }
  if (c.kf_enabled) {
    r.bufn += wave_sum(arch_n);
    r.bufts += wave_sum(arch_ts);
  }
}

// envs/drone_v2.py:197-214 with utils.py:733-743, 755-762 (lane-uniform scalar work)
__device__ __forceinline__ void st_control(const d2d_cfg &c, const d2d_state &s, int e, double action, EnvRegs &r) {
  bool ok = true, has_wp = false;
  if (c.planner_mode == D2D_PLANNER_NOMOVE) {
    r.tx = -1.0;  // traj_planner.py:72
    r.ty = -1.0;
  } else {
    ok = s.plan_ok[e] != 0;
    has_wp = s.wp_valid[e] != 0;
  }
  if (!ok) {
    const double n = sqrt(r.vx * r.vx + r.vy * r.vy);
    if (n <= c.max_acc * c.dt) {
      r.vx = 0.0;
      r.vy = 0.0;
    } else {
      r.vx = r.vx - r.vx / n * c.max_acc * c.dt;
      r.vy = r.vy - r.vy / n * c.max_acc * c.dt;
      r.x += r.vx * c.dt;
      r.y += r.vy * c.dt;
    }
    r.sm = D2D_SM_PLANNING;
    r.fail += 1;
  } else {
    r.sm = D2D_SM_EXECUTING;
    r.fail = 0;
  }
  if (has_wp) {
    const double *wp = s.wp + (size_t)e * 6;
    r.ax = wp[4];
    r.ay = wp[5];
    r.vx = wp[2];
    r.vy = wp[3];
    r.x = rint(wp[0]);  // round(): half to even
    r.y = rint(wp[1]);
  }
  r.yaw = py_mod360(r.yaw + action * c.yaw_rate * c.dt);
}

// utils.py:764-778 + envs/drone_v2.py:217-235
__device__ __forceinline__ void st_collide(const d2d_cfg &c, const d2d_state &s, int e, int lane, EnvRegs &r) {
  const int N = c.N, H = c.H;
  const double *ag = s.agents + (size_t)e * D2D_AF * N;
  const unsigned char *gt = s.gt + (size_t)e * c.W * H;
  const double R = c.drone_radius;
  // static: 5 probe points, lane q < 5 (static cells never change, so no ordering with the dyn update)
  bool wallhit = false;
  if (lane < 5) {
    const double ox = (lane == 0) ? -R : (lane == 2 ? R : 0.0);
    const double oy = (lane == 3) ? -R : (lane == 4 ? R : 0.0);
    const double qx = r.x + ox, qy = r.y + oy;
    if (qx >= c.W_px || qx < 0.0 || qy >= c.H_px || qy < 0.0) wallhit = true;
    else wallhit = gt[(size_t)cell_of(qx, c.scale) * H + cell_of(qy, c.scale)] == D2D_OCCUPIED;
  }
  int col = __any(wallhit) ? 1 : 0;
  if (!col) {
    bool dyn = false;
    for (int k = lane; k < N; k += WAVE) {
      const double dx = ag[D2D_A_PX * N + k] - r.x, dy = ag[D2D_A_PY * N + k] - r.y;
      dyn = dyn || (sqrt(dx * dx + dy * dy) < ag[D2D_A_R * N + k] + R);
    }
    if (__any(dyn)) col = 2;
  }
  int dead = 0, frz = 0;
  if (col == 0) {
    const double gx = r.x - r.tx, gy = r.y - r.ty;
    if (sqrt(gx * gx + gy * gy) <= 10.0) r.sm = D2D_SM_GOAL_REACHED;
    const double vn = sqrt(r.vx * r.vx + r.vy * r.vy);
    dead = (r.fail >= 10 && vn == 0.0) ? 1 : 0;
    frz = ((double)r.steps >= c.max_steps && !dead) ? 1 : 0;
  }
  const int done = (col != 0) || dead || frz || (r.sm == D2D_SM_GOAL_REACHED && r.tnext >= r.ntgt);
  if (done && c.kf_enabled) {  // drone_v2.py:232-235
    int an = 0, ats = 0;
    for (int k = lane; k < N; k += WAVE)
      if (s.active[(size_t)e * N + k]) {
        an += 1;
        ats += s.kf_len[(size_t)e * N + k];
      }
    r.bufn += wave_sum(an);
    r.bufts += wave_sum(ats);
  }
  if (lane == 0) {
    unsigned char *f = s.flags + (size_t)e * 4;
    f[D2D_F_COLLISION] = (unsigned char)col;
    f[D2D_F_DEADLOCK] = (unsigned char)dead;
    f[D2D_F_FREEZING] = (unsigned char)frz;
    f[D2D_F_DONE] = (unsigned char)done;
  }
}

// utils.py:780-784 + envs/drone_v2.py:251-255
__device__ __forceinline__ void st_obs(const d2d_cfg &c, const d2d_state &s, int e, int lane, const EnvRegs &r) {
  const int Lm = c.L, edge = (Lm - 1) / 2, W = c.W, H = c.H;
  const unsigned char *dm = s.dmap + (size_t)e * W * H;
  unsigned char *ob = s.obs_local + (size_t)e * Lm * Lm;
  const int ix = cell_of(r.x, c.scale) - edge, iy = cell_of(r.y, c.scale) - edge;
  for (int idx = lane; idx < Lm * Lm; idx += WAVE) {
    const int p = idx / Lm, q = idx - p * Lm;
    const int i = ix + p, j = iy + q;
    ob[idx] = (i >= 0 && i < W && j >= 0 && j < H) ? dm[(size_t)i * H + j] : (unsigned char)0;
  }
  if (lane == 0) s.obs_yaw[e] = (float)r.yaw;
}

__device__ __forceinline__ void load_regs(const d2d_state &s, int e, EnvRegs &r) {
  const double *d = s.drone + (size_t)e * D2D_DF;
  r.x = d[D2D_D_X]; r.y = d[D2D_D_Y]; r.yaw = d[D2D_D_YAW];
  r.vx = d[D2D_D_VX]; r.vy = d[D2D_D_VY]; r.ax = d[D2D_D_AX]; r.ay = d[D2D_D_AY];
  r.tx = s.target[(size_t)e * 2]; r.ty = s.target[(size_t)e * 2 + 1];
  const int *cn = s.counters + (size_t)e * D2D_CF;
  r.steps = cn[D2D_C_STEPS]; r.fail = cn[D2D_C_FAIL]; r.sm = cn[D2D_C_SM]; r.tnext = cn[D2D_C_TGT_NEXT];
  r.ntgt = cn[D2D_C_NTGT]; r.tracked = cn[D2D_C_TRACKED]; r.bufn = cn[D2D_C_BUF_N]; r.bufts = cn[D2D_C_BUF_TS];
}

__device__ __forceinline__ void store_regs(const d2d_state &s, int e, const EnvRegs &r) {
  double *d = s.drone + (size_t)e * D2D_DF;
  d[D2D_D_X] = r.x; d[D2D_D_Y] = r.y; d[D2D_D_YAW] = r.yaw;
  d[D2D_D_VX] = r.vx; d[D2D_D_VY] = r.vy; d[D2D_D_AX] = r.ax; d[D2D_D_AY] = r.ay;
  s.target[(size_t)e * 2] = r.tx; s.target[(size_t)e * 2 + 1] = r.ty;
  int *cn = s.counters + (size_t)e * D2D_CF;
  cn[D2D_C_STEPS] = r.steps; cn[D2D_C_FAIL] = r.fail; cn[D2D_C_SM] = r.sm; cn[D2D_C_TGT_NEXT] = r.tnext;
  cn[D2D_C_TRACKED] = r.tracked; cn[D2D_C_BUF_N] = r.bufn; cn[D2D_C_BUF_TS] = r.bufts;
}

__device__ __forceinline__ void run_env(const d2d_cfg &c, const d2d_state &s, int e, int lane, uint32_t stages,
                                        const Geom &g, const LdsView &L, double action, EnvRegs &r) {
  if (stages & D2D_ST_FSM) st_fsm(c, s, e, r);
  if (stages & D2D_ST_AGENTS) {
    st_agents(c, s, e, lane);
    wave_sync();
  }
  if (stages & D2D_ST_RAYCAST) st_raycast(c, s, e, lane, g, L, r);
  if (stages & D2D_ST_DYNGRID) st_dyngrid(c, s, e, lane);
  if (stages & D2D_ST_TRACKER) {
    wave_sync();
    st_tracker(c, s, e, lane, r);
  }
  if (stages & D2D_ST_CONTROL) st_control(c, s, e, action, r);
  if (stages & D2D_ST_COLLIDE) {
    wave_sync();
    st_collide(c, s, e, lane, r);
  }
  if (stages & D2D_ST_OBS) {
    wave_sync();
    st_obs(c, s, e, lane, r);
  }
}

__device__ __forceinline__ LdsView carve(char *base, const Geom &g) {
  LdsView L;
  L.cx = (double *)base;
  L.cy = L.cx + g.ncap;
  L.cr2 = L.cy + g.ncap;
  L.cidx = (int *)(L.cr2 + g.ncap);
  L.hit = (unsigned char *)(L.cidx + g.ncap);
  L.gtw = L.hit + g.ncap;
  return L;
}

extern __shared__ __attribute__((aligned(16))) char d2d_lds[];

__global__ __launch_bounds__(WAVE *WAVES_PER_BLOCK) void k_stages(d2d_cfg c, d2d_state s, uint32_t stages) {
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  const int e = blockIdx.x * WAVES_PER_BLOCK + wv;
  if (e >= c.B) return;
  const Geom g = make_geom(c);
  const LdsView L = carve(d2d_lds + (size_t)wv * g.wave_bytes, g);
  EnvRegs r;
  load_regs(s, e, r);
  run_env(c, s, e, lane, stages, g, L, s.action[e], r);
  if (lane == 0) store_regs(s, e, r);
}

// `nsteps` fused steps per launch; the env's scalar state stays in registers between steps
__global__ __launch_bounds__(WAVE *WAVES_PER_BLOCK) void k_rollout(d2d_cfg c, d2d_state s, int nsteps,
                                                                   const double *actions, const double *pin,
                                                                   unsigned char *coll_out) {
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  const int e = blockIdx.x * WAVES_PER_BLOCK + wv;
  if (e >= c.B) return;
  const Geom g = make_geom(c);
  const LdsView L = carve(d2d_lds + (size_t)wv * g.wave_bytes, g);
  EnvRegs r;
  load_regs(s, e, r);
  for (int t = 0; t < nsteps; ++t) {
    if (pin) {
      r.x = pin[(size_t)e * 2];
      r.y = pin[(size_t)e * 2 + 1];
    }
    run_env(c, s, e, lane, D2D_ST_ALL, g, L, actions[(size_t)t * c.B + e], r);
    wave_sync();
    if (coll_out && lane == 0) coll_out[(size_t)t * c.B + e] = s.flags[(size_t)e * 4 + D2D_F_COLLISION];
  }
  if (lane == 0) store_regs(s, e, r);
}

// reset(): masked copy of the snapshot over the live state, one wave per env
__global__ __launch_bounds__(WAVE *WAVES_PER_BLOCK) void k_reset(d2d_cfg c, d2d_state s, d2d_state init,
                                                                 const unsigned char *mask) {
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  const int e = blockIdx.x * WAVES_PER_BLOCK + wv;
  if (e >= c.B) return;
  if (mask && !mask[e]) return;
  const size_t N = c.N, WH = (size_t)c.W * c.H, LL = (size_t)c.L * c.L;
  for (size_t i = lane; i < D2D_AF * N; i += WAVE) s.agents[e * D2D_AF * N + i] = init.agents[e * D2D_AF * N + i];
  for (size_t i = lane; i < N; i += WAVE) {
    s.agent_unit[e * N + i] = init.agent_unit[e * N + i];
    s.active[e * N + i] = init.active[e * N + i];
    s.hit[e * N + i] = 0;
    if (s.kf_len && init.kf_len) s.kf_len[e * N + i] = init.kf_len[e * N + i];
  }
  for (size_t i = lane; i < 3 * N; i += WAVE) s.dyn_prev[e * 3 * N + i] = init.dyn_prev[e * 3 * N + i];
  if (s.kf && init.kf)
    for (size_t i = lane; i < D2D_KF * N; i += WAVE) s.kf[e * D2D_KF * N + i] = init.kf[e * D2D_KF * N + i];
  if ((WH & 3) == 0) {  // grids are 4-byte aligned per env when W*H % 4 == 0
    const uint32_t *a = (const uint32_t *)(init.gt + e * WH), *b = (const uint32_t *)(init.dmap + e * WH);
    uint32_t *x = (uint32_t *)(s.gt + e * WH), *y = (uint32_t *)(s.dmap + e * WH);
    for (size_t i = lane; i < WH / 4; i += WAVE) {
      x[i] = a[i];
      y[i] = b[i];
    }
  } else {
    for (size_t i = lane; i < WH; i += WAVE) {
      s.gt[e * WH + i] = init.gt[e * WH + i];
      s.dmap[e * WH + i] = init.dmap[e * WH + i];
    }
  }
  for (size_t i = lane; i < LL; i += WAVE) s.obs_local[e * LL + i] = 0;
  for (size_t i = lane; i < (size_t)c.T * 2; i += WAVE) s.targets[e * c.T * 2 + i] = init.targets[e * c.T * 2 + i];
  if (lane < D2D_DF) s.drone[(size_t)e * D2D_DF + lane] = init.drone[(size_t)e * D2D_DF + lane];
  if (lane < D2D_CF) s.counters[(size_t)e * D2D_CF + lane] = init.counters[(size_t)e * D2D_CF + lane];
  if (lane < 2) s.target[(size_t)e * 2 + lane] = init.target[(size_t)e * 2 + lane];
  if (lane < 4) s.flags[(size_t)e * 4 + lane] = 0;
  if (lane == 0) {
    s.newly[e] = 0;
    s.obs_yaw[e] = 0.f;
  }
}

__global__ void k_tan(const double *in, double *out, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = d2d_tan(in[i]);
}

// ------------------------------------------------------------------------------------------------
// host side of the C ABI
// ------------------------------------------------------------------------------------------------
thread_local char g_err[256] = "";

int fail(int code, const char *msg) {
  snprintf(g_err, sizeof g_err, "%s", msg);
  return code;
}

int check(const d2d_cfg *c, const d2d_state *s) {
  if (!c || !s) return fail(-1, "null cfg/state");
  if (c->abi_version != D2D_ABI_VERSION) return fail(-2, "ABI version mismatch");
  if (c->B < 0 || c->N < 0 || c->W <= 0 || c->H <= 0 || c->R <= 0 || c->L <= 0 || (c->L & 1) == 0 || c->T <= 0)
    return fail(-1, "bad dimensions");
  if (!(c->scale >= 2.0) || c->scale != (double)(long long)c->scale)
    return fail(-4, "map_scale must be an integer >= 2 (scale 1 never advances a ray, utils.py:621)");
  if (!(c->depth > 0) || !(c->dt > 0)) return fail(-1, "bad depth / dt");
  if (c->kf_enabled && (!s->kf || !s->kf_len)) return fail(-1, "kf_enabled without kf buffers");
  if (c->sigma != 0.0 && c->kf_enabled && !s->noise) return fail(-1, "var_cam != 0 needs the noise input");
  if (!s->agents || !s->agent_unit || !s->dyn_prev || !s->gt || !s->dmap || !s->drone || !s->target || !s->targets ||
      !s->counters || !s->active || !s->hit || !s->newly || !s->flags || !s->obs_local || !s->obs_yaw)
    return fail(-1, "null state pointer");
  const Geom g = make_geom(*c);
  if ((size_t)g.wave_bytes * WAVES_PER_BLOCK > 64 * 1024) return fail(-4, "N / view depth too large for the LDS tile");
  return 0;
}

int launch_stages(const d2d_cfg *c, const d2d_state *s, uint32_t stages, void *stream) {
  int rc = check(c, s);
  if (rc) return rc;
  if (!s->action && (stages & D2D_ST_CONTROL)) return fail(-1, "null action");
  if ((stages & D2D_ST_CONTROL) && c->planner_mode == D2D_PLANNER_EXTERNAL && (!s->plan_ok || !s->wp_valid || !s->wp))
    return fail(-1, "external planner mode needs plan_ok / wp_valid / wp");
  if (c->B == 0) return 0;
  const Geom g = make_geom(*c);
  const dim3 grid((c->B + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), block(WAVE * WAVES_PER_BLOCK);
  d2d_state st = *s;
  if (!st.action) st.action = (const double *)st.drone;  // never dereferenced meaningfully without CONTROL
  hipLaunchKernelGGL(k_stages, grid, block, (size_t)g.wave_bytes * WAVES_PER_BLOCK, (hipStream_t)stream, *c, st, stages);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail(-3, hipGetErrorString(err));
  return 0;
}

}  // namespace

extern "C" {

int d2d_abi_version(void) { return D2D_ABI_VERSION; }
const char *d2d_last_error(void) { return g_err; }

int d2d_run_stages(const d2d_cfg *c, const d2d_state *s, uint32_t stages, void *stream) {
  return launch_stages(c, s, stages, stream);
}
int d2d_step(const d2d_cfg *c, const d2d_state *s, void *stream) { return launch_stages(c, s, D2D_ST_ALL, stream); }
int d2d_perceive(const d2d_cfg *c, const d2d_state *s, void *stream) {
  return launch_stages(c, s, D2D_ST_PERCEIVE, stream);
}
int d2d_act(const d2d_cfg *c, const d2d_state *s, void *stream) { return launch_stages(c, s, D2D_ST_ACT, stream); }

int d2d_rollout(const d2d_cfg *c, const d2d_state *s, int32_t nsteps, const double *actions, const double *pin,
                uint8_t *coll_out, void *stream) {
  int rc = check(c, s);
  if (rc) return rc;
  if (!actions || nsteps < 0) return fail(-1, "rollout: bad arguments");
  if (c->planner_mode == D2D_PLANNER_EXTERNAL && (!s->plan_ok || !s->wp_valid || !s->wp))
    return fail(-1, "external planner mode needs plan_ok / wp_valid / wp");
  if (c->B == 0 || nsteps == 0) return 0;
  const Geom g = make_geom(*c);
  const dim3 grid((c->B + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), block(WAVE * WAVES_PER_BLOCK);
  hipLaunchKernelGGL(k_rollout, grid, block, (size_t)g.wave_bytes * WAVES_PER_BLOCK, (hipStream_t)stream, *c, *s,
                     (int)nsteps, actions, pin, coll_out);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail(-3, hipGetErrorString(err));
  return 0;
}

int d2d_reset(const d2d_cfg *c, const d2d_state *s, const d2d_state *init, const uint8_t *mask, void *stream) {
  int rc = check(c, s);
  if (rc) return rc;
  if (!init || !init->agents || !init->agent_unit || !init->dyn_prev || !init->gt || !init->dmap || !init->drone ||
      !init->target || !init->targets || !init->counters || !init->active)
    return fail(-1, "reset: incomplete snapshot");
  if (c->B == 0) return 0;
  const dim3 grid((c->B + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), block(WAVE * WAVES_PER_BLOCK);
  hipLaunchKernelGGL(k_reset, grid, block, 0, (hipStream_t)stream, *c, *s, *init, mask);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail(-3, hipGetErrorString(err));
  return 0;
}

int d2d_tan_array(const double *in, double *out, int64_t n, void *stream) {
  if (n < 0 || (n > 0 && (!in || !out))) return fail(-1, "tan_array: bad arguments");
  if (n == 0) return 0;
  const int bs = 256;
  hipLaunchKernelGGL(k_tan, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, (hipStream_t)stream, in, out,
                     (long long)n);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail(-3, hipGetErrorString(err));
  return 0;
}

}  // extern "C"
