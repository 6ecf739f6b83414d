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
#ifndef D2D_MIN_WAVES
#define D2D_MIN_WAVES 3  // waves per SIMD the register allocator must leave room for
#endif

#ifdef D2D_STAMPS
// Diagnostic build only (tools/stage_stamps.py): per-env shader-clock stamps at stage boundaries.
__device__ unsigned long long *d2d_stamp_buf = nullptr;
#define D2D_STAMP(idx)                                                          \
  do {                                                                          \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                 \
    if (d2d_stamp_buf && lane == 0) d2d_stamp_buf[(size_t)e * 16 + (idx)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define D2D_STAMP(idx) do { } while (0)
#endif

namespace {

// ------------------------------------------------------------------------------------------------
// small exact helpers
// ------------------------------------------------------------------------------------------------

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

// lane -> lane hand-off inside one wave through GLOBAL memory (this env's own records): drains the
// wave's outstanding stores and refreshes the CU's L1.  Expensive (a full memory round trip); the
// fused step needs it once, before the observation crop re-reads the cells the rays just wrote.
__device__ __forceinline__ void wave_sync_global() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// lane -> lane hand-off through LDS only: LDS operations of one wave execute in order, so it is enough
// to keep the compiler from moving LDS accesses across this point and to drain lgkmcnt.
__device__ __forceinline__ void wave_sync_lds() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int wave_sum(int v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}

// idx / d and idx % d for small non-negative idx without an integer division
struct FastDiv {
  float inv;
  int d;
  __device__ __forceinline__ explicit FastDiv(int d_) : inv(1.0f / (float)d_), d(d_) {}
  __device__ __forceinline__ void divmod(int idx, int &q, int &r) const {
    q = (int)(((float)idx + 0.5f) * inv);
    r = idx - q * d;
  }
};

struct LdsView {
  double *ax, *ay, *ar, *ar2;  // all agents of the env after this step's move   [ncap]
  double *cx, *cy, *cr2, *crr;  // compacted ray candidates (centre, r^2, r)      [ncap]
  int *cidx;                   // candidate -> agent index                        [ncap]
  int *ncx, *ncy, *nu;         // new dynamic block of every agent (cell, half)   [ncap]
  unsigned int *bm;            // bitmap of cells covered by some agent's new block [bmw]
  unsigned char *hit;          // per-agent hit flag (OR over rays)               [ncap]
  unsigned char *gtw;          // staged ground-truth window                      [ws * ws]
};

struct Geom {
  int ncap;   // N rounded up to a multiple of 4
  int reach;  // cells a ray can travel from the drone cell
  int ws;     // window edge = 2 * reach + 1
  int smax;   // samples after which every ray has stopped: sample k is >= k * ss from the drone
  int klo;    // samples 0..klo are < depth from the drone whatever the slope (k * ss * sqrt(2) < depth)
  int bmw;    // dwords of the per-env cell bitmap kept in LDS (0: grid too large, loop over agents instead)
  int wave_bytes;
};

__host__ __device__ inline Geom make_geom(const d2d_cfg &c) {
  Geom g;
  g.ncap = (c.N + 3) & ~3;
  if (g.ncap < 4) g.ncap = 4;
  g.reach = (int)((c.depth + 1.5 * (c.scale - 1.0)) / c.scale) + 2;
  g.ws = 2 * g.reach + 1;
  const double ss = c.scale - 1.0;
  g.smax = (int)(c.depth / ss) + 2;
  g.klo = (int)(c.depth / (ss * 1.4142136)) - ((c.depth / (ss * 1.4142136)) == (double)(int)(c.depth / (ss * 1.4142136)) ? 1 : 0);
  if (g.klo < -1) g.klo = -1;
  g.bmw = (c.W * c.H + 31) / 32;
  if (g.bmw > 2048) g.bmw = 0;  // 8 KB per wave at most (256 x 256 cells)
  g.wave_bytes = (81 * g.ncap + 4 * g.bmw + g.ws * g.ws + 15) & ~15;
  return g;
}

struct EnvRegs {  // lane-uniform per-env scalars carried in registers across the fused stages
  double x, y, yaw, vx, vy, ax, ay;
  double tx, ty;
  int steps, fail, sm, tnext, ntgt, tracked, bufn, bufts;
};

struct Consts {  // per-launch derived constants
  double inv_scale;
};

// Python / numpy `int(v // s)` for integer-valued s > 0: the exact mathematical floor.  floor(v * (1/s))
// is within one of it; the fused remainder v - q s (exact when q is right) settles which.
__device__ __forceinline__ int cell_fast(double v, double s, double inv_s) {
  double q = floor(v * inv_s);
  const double r = __builtin_fma(-q, s, v);
  if (r < 0.0) q -= 1.0;
  else if (r >= s) q += 1.0;
  return (int)q;
}

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

// envs/drone_v2.py:176-179 + utils.py:472-493; lane = agent.  With `move` false the agents are only
// staged into LDS (a launch that does not contain the AGENTS stage).
__device__ __forceinline__ void st_agents(const d2d_cfg &c, const d2d_state &s, int e, int lane, const LdsView &L,
                                          const Consts &k_, bool move) {
  const int N = c.N;
  double *ag = s.agents + (size_t)e * D2D_AF * N;
  const double cs = 0x1.bb67ae8584cabp-1, sn = 0x1.fffffffffffffp-2;  // cos(pi/6), sin(pi/6)
  for (int k = lane; k < N; k += WAVE) {
    double px = ag[D2D_A_PX * N + k], py = ag[D2D_A_PY * N + k];
    const double rr = ag[D2D_A_R * N + k], r2 = ag[D2D_A_R2 * N + k];
    const int u = s.agent_unit[(size_t)e * N + k];
    if (move) {
      const double velx = ag[D2D_A_VX * N + k], vely = ag[D2D_A_VY * N + k];
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
      px = px + ux * c.dt;
      py = py + uy * c.dt;
      ag[D2D_A_PX * N + k] = px;
      ag[D2D_A_PY * N + k] = py;
      ag[D2D_A_VX * N + k] = pvx;
      ag[D2D_A_VY * N + k] = pvy;
    }
    L.ax[k] = px;
    L.ay[k] = py;
    L.ar[k] = rr;
    L.ar2[k] = r2;
    L.ncx[k] = cell_fast(px, c.scale, k_.inv_scale);
    L.ncy[k] = cell_fast(py, c.scale, k_.inv_scale);
    L.nu[k] = u;
  }
}

// utils.py:612-618
__device__ __forceinline__ double positive_angle(double a) {
  const double two_pi = 0x1.921fb54442d18p+2;  // math.pi * 2
  a = copysign(fmod_pos(fabs(a), two_pi), a);
  if (a < 0.0) a += two_pi;
  return a;
}

// floor(v / s) for 0 < v < 2^31 and integer s >= 2: floor(v / s) == floor(floor(v) / s), and floor(v) fits an
// int, so the rest is 24-bit integer work (full-rate VALU) instead of fp64.
struct CellDiv {
  float inv;
  int s;
  __device__ __forceinline__ explicit CellDiv(double scale) : inv(1.0f / (float)scale), s((int)scale) {}
  __device__ __forceinline__ int operator()(double v) const {
    const int vi = (int)v;  // truncation == floor for v >= 0
    int q = (int)((float)vi * inv);
    const int r = vi - q * s;
    q += (r >= s) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
  }
};

// utils.py:593-609, 620-713; lane = ray.  (x0, y0, yaw0) is the pose BEFORE this step's control.
//
// The reference marches each ray sample by sample until it stops.  Here every lane runs the same fixed
// number of samples (g.smax: after that many every ray is past `depth`), positions are the same iterated
// sums, and the per-sample decisions are predicated on `alive` instead of steering control flow, so the
// LDS lookups of consecutive samples overlap and no lane waits for the slowest ray.  Two exact shortcuts:
//  * candidates: an agent can only be hit by ray i if its centre is within radius of the ray's LINE and
//    not behind the drone; that conservative test runs once per (ray, candidate) and leaves a bit mask
//    (almost always empty), only the set bits get the reference's exact per-sample circle test;
//  * depth: sample k <= klo is nearer than `depth` for any slope, so `dist >= depth^2` is only evaluated
//    for the last few samples.
template <bool UNROLL>
__device__ __forceinline__ void st_raycast(const d2d_cfg &c, const d2d_state &s, int e, int lane, const Geom &g,
                                           const LdsView &L, const Consts &k_, double x0, double y0, double yaw0,
                                           EnvRegs &r) {
  const int N = c.N, W = c.W, H = c.H;
  const unsigned char *gt = s.gt + (size_t)e * W * H;
  unsigned char *dm = s.dmap + (size_t)e * W * H;
  const double ss = c.scale - 1.0;  // x_step_size, utils.py:621
  const CellDiv cell(c.scale);

  // ---- stage the ground-truth window the rays can reach (LDS tile, OOB = wall) ----
  const int ci0 = cell_fast(x0, c.scale, k_.inv_scale) - g.reach, cj0 = cell_fast(y0, c.scale, k_.inv_scale) - g.reach;
  {
    const FastDiv fd(g.ws);
    for (int idx = lane; idx < g.ws * g.ws; idx += WAVE) {
      int wi, wj;
      fd.divmod(idx, wi, wj);
      const int i = ci0 + wi, j = cj0 + wj;
#ifdef D2D_ABL_NOSTAGE
      L.gtw[idx] = (i > 0 && i < W - 1 && j > 0 && j < H - 1) ? 2 : 1;
#else
      L.gtw[idx] = (i >= 0 && i < W && j >= 0 && j < H) ? gt[(size_t)i * H + j] : (unsigned char)D2D_OCCUPIED;
#endif
    }
  }

  // ---- cull: only agents within reach of some ray sample can pass the circle test of :659 ----
  // a sample is < depth + sqrt(2) * ss from the drone, so |agent - drone| <= radius + that bound
  int ncand = 0;
  for (int k0 = 0; k0 < N; k0 += WAVE) {
    const int k = k0 + lane;
    bool cand = false;
    double px = 0, py = 0, r2 = 0, rr = 0;
    if (k < N) {
      px = L.ax[k];
      py = L.ay[k];
      r2 = L.ar2[k];
      rr = fabs(L.ar[k]);
      const double lim = rr + c.depth + 1.5 * ss + 2.0;
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
      L.crr[slot] = rr;
      L.cidx[slot] = k;
    }
    ncand += __popcll(m);
  }
#ifdef D2D_ABL_NOCAND
  ncand = 0;
#endif
  wave_sync_lds();
  D2D_STAMP(4);

  const double rad90 = 0x1.921fb54442d18p+0, rad270 = 0x1.2d97c7f3321d2p+2;  // radians(90), radians(270)
  const double pi_ = 0x1.921fb54442d18p+1;
  const double player_angle = 0x1.921fb54442d18p+2 - yaw0 * 0x1.1df46a2529d39p-6;  // pi*2 - radians(yaw)
  const double depth2 = c.depth * c.depth;
  const bool mask_path = ncand <= 32;

  for (int i0 = 0; i0 < c.R; i0 += WAVE) {
    const int i = i0 + lane;
    const double ang = positive_angle(player_angle + (c.ray_off0 + c.ray_dth * (double)i));
    const bool faced_right = (ang < rad90 || ang > rad270);
    const bool faced_up = (ang > pi_);
#ifdef D2D_ABL_NOTAN
    double slope = ang * 0.3;
#else
    double slope = d2d_tan(ang);
#endif
    double xs, ys;
    if (fabs(slope) > 1.0) {
      slope = 1.0 / slope;
      ys = faced_up ? -ss : ss;
      xs = ys * slope;
    } else {
      xs = faced_right ? ss : -ss;
      ys = xs * slope;
    }

    // per-ray candidate mask (conservative; the exact test stays per sample)
    unsigned int cmask = 0;
    if (mask_path) {
      const double len2 = xs * xs + ys * ys, l1 = fabs(xs) + fabs(ys);
      for (int q = 0; q < ncand; ++q) {
        const double ex = L.cx[q] - x0, ey = L.cy[q] - y0, rq = L.crr[q] + 1e-6;
        const double cr = ex * ys - ey * xs, dt = ex * xs + ey * ys;
        const bool near_line = cr * cr <= rq * rq * len2 * (1.0 + 1e-9);
        const bool ahead = dt + rq * l1 >= 0.0;
        cmask |= (near_line && ahead) ? (1u << q) : 0u;
      }
    }

    D2D_STAMP(5);
    double x = x0, y = y0;
    bool alive = (i < c.R) && (0.0 < x && x < c.W_px && 0.0 < y && y < c.H_px);
    const int smax = g.smax, klo = g.klo;
    auto sample = [&](int k) {
      // exact circle tests (utils.py:658-662): every candidate that can matter, no early-out among agents
      bool any = false;
      if (mask_path) {
        unsigned int m = alive ? cmask : 0u;
        while (m) {
          const int q = __ffs((int)m) - 1;
          m &= m - 1;
          const double dx = L.cx[q] - x, dy = L.cy[q] - y;
          if (dx * dx + dy * dy <= L.cr2[q]) {
            L.hit[L.cidx[q]] = 1;
            any = true;
          }
        }
      } else if (alive) {
        for (int q = 0; q < ncand; ++q) {
          const double dx = L.cx[q] - x, dy = L.cy[q] - y;
          if (dx * dx + dy * dy <= L.cr2[q]) {
            L.hit[L.cidx[q]] = 1;
            any = true;
          }
        }
      }
      // the cell of this sample and its ground-truth value from the LDS tile.  Unconditional and clamped (a
      // live sample always lies inside the tile: the previous sample was nearer than `depth`); it must not
      // become a select between an LDS and a global pointer (flat load + vmcnt(0) wait per sample).
      const double xc = alive ? x : x0, yc = alive ? y : y0;
      const int ci = cell(xc), cj = cell(yc);
      const int wi = min(max(ci - ci0, 0), g.ws - 1), wj = min(max(cj - cj0, 0), g.ws - 1);
      const unsigned char wall = L.gtw[wi * g.ws + wj];
      bool far = false;
      if (k > klo) far = ((x - x0) * (x - x0) + (y - y0) * (y - y0) >= depth2);
      const bool stop = (wall == D2D_OCCUPIED) || far;
      if (alive && !any) {
#ifndef D2D_ABL_NOSTORE
        if (!stop) dm[(size_t)ci * H + cj] = D2D_UNOCCUPIED;
        else if (wall == D2D_OCCUPIED) dm[(size_t)ci * H + cj] = D2D_OCCUPIED;
#endif
      }
      alive = alive && !any && !stop;
      x = x + xs;
      y = y + ys;
      alive = alive && (0.0 < x && x < c.W_px && 0.0 < y && y < c.H_px);
    };
#ifndef D2D_ABL_NOMARCH
    if (UNROLL) {
#pragma unroll
      for (int k = 0; k < 10; ++k) sample(k);
    } else {
      for (int k = 0; k < smax; ++k) sample(k);
    }
#endif
  }
  wave_sync_lds();
  D2D_STAMP(6);

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

// utils.py:527-540, lane = agent.  The reference clears every cell of dynamic_idx (== every DYNAMIC cell,
// all of which lie in the blocks of dyn_prev) and then marks every agent's new block.  Written here as
// ONE order-independent pass: the final value of a cell depends only on (static or not, covered by some
// new block or not), so a lane may observe another lane's already-final value instead of the old one
// without changing the outcome -- no clear/set ordering, no memory fence.
// Is cell (i, j) inside the new block of ANY agent?  Straight loop, no early exit: the LDS reads are
// broadcasts and pipeline.  Only used when the grid is too large for the LDS bitmap.
__device__ __forceinline__ bool dyn_covered_loop(const LdsView &L, int N, int i, int j) {
  bool cov = false;
  for (int a = 0; a < N; ++a) cov = cov || (abs(i - L.ncx[a]) <= L.nu[a] && abs(j - L.ncy[a]) <= L.nu[a]);
  return cov;
}

__device__ __forceinline__ void st_dyngrid(const d2d_cfg &c, const d2d_state &s, int e, int lane, const Geom &g,
                                           const LdsView &L) {
  const int N = c.N, W = c.W, H = c.H;
  unsigned char *__restrict__ gt = s.gt + (size_t)e * W * H;
  int *prev = s.dyn_prev + (size_t)e * N * 3;
  const bool use_bm = g.bmw > 0;
  if (use_bm) {  // coverage bitmap: every agent ORs the cells of its new block
    for (int w = lane; w < g.bmw; w += WAVE) L.bm[w] = 0u;
    wave_sync_lds();
    for (int k = lane; k < N; k += WAVE) {
      const int cx = L.ncx[k], cy = L.ncy[k], u = L.nu[k];
      const int i1 = min(cx + u + 1, W), j1 = min(cy + u + 1, H);
      for (int i = max(cx - u, 0); i < i1; ++i)
        for (int j = max(cy - u, 0); j < j1; ++j) {
          const int bit = i * H + j;
          atomicOr(&L.bm[bit >> 5], 1u << (bit & 31));
        }
    }
    wave_sync_lds();
  }
  for (int k = lane; k < N; k += WAVE) {
    const int pcx = prev[3 * k], pcy = prev[3 * k + 1], pu = prev[3 * k + 2];
    const int ncx = L.ncx[k], ncy = L.ncy[k], nu = L.nu[k];
    if (pu <= 1 && nu <= 1) {
      // common case (radius < 2 cells): all 18 cell reads are issued back to back, then the few writes
      unsigned char pv[9], nv[9];
      bool pcov[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const int di = q / 3 - 1, dj = q % 3 - 1;
        const int i = pcx + di, j = pcy + dj;
        const bool ok = abs(di) <= pu && abs(dj) <= pu && i >= 0 && i < W && j >= 0 && j < H;
        pv[q] = ok ? gt[(size_t)i * H + j] : (unsigned char)D2D_OCCUPIED;
        const int bit = ok ? i * H + j : 0;
        pcov[q] = use_bm ? ((L.bm[bit >> 5] >> (bit & 31)) & 1u) != 0u : false;
        const int i2 = ncx + di, j2 = ncy + dj;
        const bool ok2 = abs(di) <= nu && abs(dj) <= nu && i2 >= 0 && i2 < W && j2 >= 0 && j2 < H;
        nv[q] = ok2 ? gt[(size_t)i2 * H + j2] : (unsigned char)D2D_OCCUPIED;
      }
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const int di = q / 3 - 1, dj = q % 3 - 1;
        if (pv[q] == D2D_DYNAMIC) {
          const bool cov = use_bm ? pcov[q] : dyn_covered_loop(L, N, pcx + di, pcy + dj);
          if (!cov) gt[(size_t)(pcx + di) * H + (pcy + dj)] = D2D_UNOCCUPIED;
        }
        if (nv[q] != D2D_OCCUPIED && nv[q] != D2D_DYNAMIC) gt[(size_t)(ncx + di) * H + (ncy + dj)] = D2D_DYNAMIC;
      }
    } else {
      const int i1 = min(pcx + pu + 1, W), j1 = min(pcy + pu + 1, H);
      for (int i = max(pcx - pu, 0); i < i1; ++i)
        for (int j = max(pcy - pu, 0); j < j1; ++j)
          if (gt[(size_t)i * H + j] == D2D_DYNAMIC) {
            const int bit = i * H + j;
            const bool cov = use_bm ? ((L.bm[bit >> 5] >> (bit & 31)) & 1u) != 0u : dyn_covered_loop(L, N, i, j);
            if (!cov) gt[(size_t)i * H + j] = D2D_UNOCCUPIED;
          }
      const int i3 = min(ncx + nu + 1, W), j3 = min(ncy + nu + 1, H);
      for (int i = max(ncx - nu, 0); i < i3; ++i)
        for (int j = max(ncy - nu, 0); j < j3; ++j) {
          const unsigned char v = gt[(size_t)i * H + j];
          if (v != D2D_OCCUPIED && v != D2D_DYNAMIC) gt[(size_t)i * H + j] = D2D_DYNAMIC;
        }
    }
    prev[3 * k] = ncx;
    prev[3 * k + 1] = ncy;
    prev[3 * k + 2] = nu;
  }
}

// ---- Kalman trackers, utils.py:172-275; lane = tracker slot ----
// F = [[1,0,.1,0],[0,1,0,.1],[0,0,1,0],[0,0,0,1]] and H = [I2 0] are constant, so the dense products
// of the reference collapse: multiplying by an exact 0 or 1 and adding an exact 0 do not round, hence the
// sparse expressions below give the same values as the oracle's dense loops (tests compare bit for bit).
__device__ __forceinline__ void st_tracker(const d2d_cfg &c, const d2d_state &s, int e, int lane, const LdsView &L,
                                           EnvRegs &r) {
  const int N = c.N;
  int arch_n = 0, arch_ts = 0;
  for (int k0 = 0; k0 < N; k0 += WAVE) {
    const int k = k0 + lane;
    if (k < N) {
      const bool has_z = L.hit[k] != 0;
      unsigned char act = s.active[(size_t)e * N + k];
      if (!c.kf_enabled) {
        if (has_z) s.active[(size_t)e * N + k] = 1;
      } else if (act || has_z) {
        double *gk = s.kf + ((size_t)e * N + k) * D2D_KF;
        int len = 1;
        double zx = L.ax[k], zy = L.ay[k];
        if (s.noise) {
          zx = zx + c.sigma * s.noise[((size_t)e * N + k) * 2];
          zy = zy + c.sigma * s.noise[((size_t)e * N + k) * 2 + 1];
        }
        double m0, m1, m2, m3;
        double S[16];
        if (act) {
          len = s.kf_len[(size_t)e * N + k];
          const double u0 = gk[0], u1 = gk[1], u2 = gk[2], u3 = gk[3];
          double T[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) T[i] = gk[4 + i];
          // predict(), utils.py:225-240
          const double qn = (c.sigma != 0.0) ? 0.1 : 0.001;
          m0 = u0 + 0.1 * u2;
          m1 = u1 + 0.1 * u3;
          m2 = u2;
          m3 = u3;
          double FS[16];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            FS[j] = T[j] + 0.1 * T[8 + j];
            FS[4 + j] = T[4 + j] + 0.1 * T[12 + j];
            FS[8 + j] = T[8 + j];
            FS[12 + j] = T[12 + j];
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            S[4 * i + 0] = FS[4 * i + 0] + FS[4 * i + 2] * 0.1;
            S[4 * i + 1] = FS[4 * i + 1] + FS[4 * i + 3] * 0.1;
            S[4 * i + 2] = FS[4 * i + 2];
            S[4 * i + 3] = FS[4 * i + 3];
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) S[5 * i] += qn;
          len += 1;
          if (S[0] >= 150.0 || !(c.kf_lo_x < m0 && m0 < c.kf_hi_x) || !(c.kf_lo_y < m1 && m1 < c.kf_hi_y)) {
            arch_n += 1;  // archived copy -> tracker_buffer
            arch_ts += len;
            m0 = m1 = m2 = m3 = 0.0;
#pragma unroll
            for (int i = 0; i < 16; ++i) S[i] = 0.0;
            S[0] = 1.0;
            S[5] = 1.0;
            S[10] = 10.0;
            S[15] = 10.0;
            len = 1;
            act = 0;
          }
          if (has_z) {  // update, utils.py:249-260 (also runs on the freshly reset filter)
            const double a = c.sigma + S[0], b = S[1], cc = S[4], d = c.sigma + S[5];
            const double det = a * d - b * cc;
            const double i00 = d / det, i01 = -b / det, i10 = -cc / det, i11 = a / det;
            double K0[4], K1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              K0[i] = S[4 * i] * i00 + S[4 * i + 1] * i10;
              K1[i] = S[4 * i] * i01 + S[4 * i + 1] * i11;
            }
            const double rx = zx - m0, ry = zy - m1;
            double P[16];
            const double e00 = 1.0 - K0[0], e01 = 0.0 - K1[0], e10 = 0.0 - K0[1], e11 = 1.0 - K1[1];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              P[j] = e00 * S[j] + e01 * S[4 + j];
              P[4 + j] = e10 * S[j] + e11 * S[4 + j];
              P[8 + j] = ((0.0 - K0[2]) * S[j] + (0.0 - K1[2]) * S[4 + j]) + S[8 + j];
              P[12 + j] = ((0.0 - K0[3]) * S[j] + (0.0 - K1[3]) * S[4 + j]) + S[12 + j];
            }
            m0 = m0 + (K0[0] * rx + K1[0] * ry);
            m1 = m1 + (K0[1] * rx + K1[1] * ry);
            m2 = m2 + (K0[2] * rx + K1[2] * ry);
            m3 = m3 + (K0[3] * rx + K1[3] * ry);
#pragma unroll
            for (int i = 0; i < 16; ++i) S[i] = P[i];
          }
        } else {  // first sighting, utils.py:263-273
          m0 = zx;
          m1 = zy;
          m2 = 0.0;
          m3 = 0.0;
#pragma unroll
          for (int i = 0; i < 16; ++i) S[i] = 0.0;
          S[0] = 1.0;
          S[5] = 1.0;
          S[10] = 10.0;
          S[15] = 10.0;
          len = 1;
          act = 1;
        }
        gk[0] = m0;
        gk[1] = m1;
        gk[2] = m2;
        gk[3] = m3;
#pragma unroll
        for (int i = 0; i < 16; ++i) gk[4 + i] = S[i];
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
__device__ __forceinline__ void st_collide(const d2d_cfg &c, const d2d_state &s, int e, int lane, const LdsView &L,
                                           const Consts &k_, EnvRegs &r) {
  const int N = c.N, H = c.H;
  const unsigned char *gt = s.gt + (size_t)e * c.W * H;
  const double R = c.drone_radius;
  // static: 5 probe points, lane q < 5 (static cells never change, so no ordering with the dyn update)
  bool wallhit = false;
  if (lane < 5) {
    const double ox = (lane == 0) ? -R : (lane == 2 ? R : 0.0);
    const double oy = (lane == 3) ? -R : (lane == 4 ? R : 0.0);
    const double qx = r.x + ox, qy = r.y + oy;
    if (qx >= c.W_px || qx < 0.0 || qy >= c.H_px || qy < 0.0) wallhit = true;
    else wallhit = gt[(size_t)cell_fast(qx, c.scale, k_.inv_scale) * H + cell_fast(qy, c.scale, k_.inv_scale)] == D2D_OCCUPIED;
  }
  int col = __any(wallhit) ? 1 : 0;
  if (!col) {
    bool dyn = false;
    for (int k = lane; k < N; k += WAVE) {
      const double dx = L.ax[k] - r.x, dy = L.ay[k] - r.y;
      dyn = dyn || (sqrt(dx * dx + dy * dy) < L.ar[k] + R);
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

// utils.py:780-784 + envs/drone_v2.py:251-255.  Lanes sweep the L x L crop (rows of the map are
// contiguous in memory); the loads of a chunk are all issued before its stores so the crop costs a
// couple of memory round trips instead of one per 64 bytes.
__device__ __forceinline__ void st_obs(const d2d_cfg &c, const d2d_state &s, int e, int lane, const Consts &k_,
                                       const EnvRegs &r) {
  const int Lm = c.L, edge = (Lm - 1) / 2, W = c.W, H = c.H, n = Lm * Lm;
  const unsigned char *__restrict__ dm = s.dmap + (size_t)e * W * H;
  unsigned char *__restrict__ ob = s.obs_local + (size_t)e * n;
  const int ix = cell_fast(r.x, c.scale, k_.inv_scale) - edge, iy = cell_fast(r.y, c.scale, k_.inv_scale) - edge;
  const FastDiv fd(Lm);
  constexpr int CH = 9;  // 9 x 64 = 576 cells per chunk: L = 33 takes two chunks
  for (int base = 0; base < n; base += CH * WAVE) {
    unsigned char v[CH];
#pragma unroll
    for (int t = 0; t < CH; ++t) {
      const int idx = base + t * WAVE + lane;
      int p, q;
      fd.divmod(idx, p, q);
      const int i = ix + p, j = iy + q;
      v[t] = (idx < n && i >= 0 && i < W && j >= 0 && j < H) ? dm[(size_t)i * H + j] : (unsigned char)0;
    }
#pragma unroll
    for (int t = 0; t < CH; ++t) {
      const int idx = base + t * WAVE + lane;
      if (idx < n) ob[idx] = v[t];
    }
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

// One env-step (or any subset of its stages) by one wave.  Stage results are identical to running the
// stages in reference order; the ORDER OF EXECUTION differs where that is free: the control stage only
// consumes inputs (plan, action) and the previous pose, so it runs first and its memory latency overlaps
// with everything else, while the raycast keeps using the pose from before it (x0, y0, yaw0).
__device__ __forceinline__ void run_env(const d2d_cfg &c, const d2d_state &s, int e, int lane, uint32_t stages,
                                        const Geom &g, const LdsView &L, double action, EnvRegs &r) {
  Consts k_;
  k_.inv_scale = 1.0 / c.scale;
  D2D_STAMP(1);
  if (stages & D2D_ST_FSM) st_fsm(c, s, e, r);
  const double x0 = r.x, y0 = r.y, yaw0 = r.yaw;
  if (stages & D2D_ST_CONTROL) st_control(c, s, e, action, r);
  D2D_STAMP(2);
  const uint32_t needs_agents = D2D_ST_AGENTS | D2D_ST_RAYCAST | D2D_ST_DYNGRID | D2D_ST_TRACKER | D2D_ST_COLLIDE;
  if (stages & needs_agents) {
    st_agents(c, s, e, lane, L, k_, (stages & D2D_ST_AGENTS) != 0);
    wave_sync_lds();
  }
  D2D_STAMP(3);
  if (stages & D2D_ST_RAYCAST) {
    if (g.smax == 10 && g.klo == 6) st_raycast<true>(c, s, e, lane, g, L, k_, x0, y0, yaw0, r);  // depth 80, scale 10
    else st_raycast<false>(c, s, e, lane, g, L, k_, x0, y0, yaw0, r);
  }
  D2D_STAMP(7);
  if (stages & D2D_ST_DYNGRID) st_dyngrid(c, s, e, lane, g, L);
  D2D_STAMP(8);
  if (stages & D2D_ST_TRACKER) {
    if (!(stages & D2D_ST_RAYCAST)) {  // hit mask of an earlier launch: stage it where the raycast leaves it
      for (int k = lane; k < c.N; k += WAVE) L.hit[k] = s.hit[(size_t)e * c.N + k];
      wave_sync_lds();
    }
    st_tracker(c, s, e, lane, L, r);
  }
  D2D_STAMP(9);
  if (stages & D2D_ST_COLLIDE) st_collide(c, s, e, lane, L, k_, r);
  D2D_STAMP(10);
  if (stages & D2D_ST_OBS) {
    if (stages & D2D_ST_RAYCAST) wave_sync_global();  // the crop re-reads cells the rays just wrote
    D2D_STAMP(11);
    st_obs(c, s, e, lane, k_, r);
  }
  D2D_STAMP(12);
}

__device__ __forceinline__ LdsView carve(char *base, const Geom &g) {
  LdsView L;
  L.ax = (double *)base;
  L.ay = L.ax + g.ncap;
  L.ar = L.ay + g.ncap;
  L.ar2 = L.ar + g.ncap;
  L.cx = L.ar2 + g.ncap;
  L.cy = L.cx + g.ncap;
  L.cr2 = L.cy + g.ncap;
  L.crr = L.cr2 + g.ncap;
  L.cidx = (int *)(L.crr + g.ncap);
  L.ncx = L.cidx + g.ncap;
  L.ncy = L.ncx + g.ncap;
  L.nu = L.ncy + g.ncap;
  L.bm = (unsigned int *)(L.nu + g.ncap);
  L.hit = (unsigned char *)(L.bm + g.bmw);
  L.gtw = L.hit + g.ncap;
  return L;
}

extern __shared__ __attribute__((aligned(16))) char d2d_lds[];

__global__ __launch_bounds__(WAVE *WAVES_PER_BLOCK, D2D_MIN_WAVES) void k_stages(d2d_cfg c, d2d_state s, uint32_t stages) {
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  const int e = blockIdx.x * WAVES_PER_BLOCK + wv;
  if (e >= c.B) return;
  const Geom g = make_geom(c);
  const LdsView L = carve(d2d_lds + (size_t)wv * g.wave_bytes, g);
  EnvRegs r;
#ifdef D2D_STAMPS
  if (d2d_stamp_buf && lane == 0) d2d_stamp_buf[(size_t)e * 16 + 0] = __builtin_amdgcn_s_memtime();
#endif
  load_regs(s, e, r);
  run_env(c, s, e, lane, stages, g, L, s.action[e], r);
  if (lane == 0) store_regs(s, e, r);
  D2D_STAMP(13);
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
    wave_sync_global();  // the next step re-reads grid cells / records other lanes of this wave wrote
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

#ifdef D2D_STAMPS
int d2d_debug_set_stamps(unsigned long long *buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(d2d_stamp_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : -3;
}
#endif

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
