// d2d_hip.hip — the batched Drone2D step for MI355X (gfx950 / CDNA4) and its C ABI (include/d2d.h).
//
// Mapping: ONE WAVEFRONT (64 lanes) PER ENV, 4 envs per 256-thread workgroup, no inter-wave
// communication at all (envs are independent), so there is no __syncthreads() anywhere: every
// hand-off is lane -> lane inside one wave through LDS or the env's own global records.
//   agents / trackers / dynamic grid : lane = agent  (N > 64 loops)
//   raycast                          : lane = ray    (R > 64 loops); agents that can possibly be
//                                      hit are compacted into LDS with __ballot + popcount, the
//                                      ground-truth window the rays can reach is an LDS tile
//   collision                        : lane = probe / agent, __any / __ballot reduction
//   observation                      : an LDS tile of the drone's map that the rays patch in place
// Memory schedule of one env-step, default geometry with N <= 40 (SPEC 1 / 2, Geom.full) -- ONE batch of loads, then fire-and-forget
// stores, no fence:
//   batch    everything is addressed by the env index alone: pose, counters, inputs, agents, tracker flags; the tracker states
//            and BOTH 50 x 50 grids go to LDS whole by LDS-DMA (global_load_lds, no VGPRs).  Rays, collision probes, the
//            dynamic-grid update (coverage marks kept in the ground-truth copy) and the observation crop read the copies;
//            the per-ray tan / candidate work runs while the DMA is in flight
//   stores   agents, drone map, grid, trackers, flags, observation
// Other configurations (more agents, other maps) keep two batches: batch 2 = the ground-truth window tile the rays can reach
// and the drone-map crop tile (byte loads, all in flight before the first LDS write), dynamic-grid cells, collision probes,
// all addressed by batch-1 data.  Their kernels (SPEC 3 = default geometry with any number of agents, SPEC 0 = anything) read
// cfg / state through the kernarg segment (StagesKArgs), cull ray candidates against the cone of the rays, and update the
// dynamic cells of grids above 256 x 256 cells in two phases (dyn_apply).
// What bounds the kernel is instruction issue (~1400 VALU + ~950 SALU wave-instructions per env-step at config 2, fp64
// heavy), not bytes: see DESIGN.md section 3 for the measurements behind each choice.
// The planner / gaze plugins (d2d_plugins.h) and the persistent closed loop k_closed (every wave loops over the steps
// of its own env: gaze -> perceive -> plan -> act) follow the step kernel below.
// There is no dense contraction on this path, hence no MFMA.  Arithmetic is fp64 in the reference's
// own operation order (compiled with -ffp-contract=off; the few fused multiply-adds are the ones the
// reference's runtime performs: libm tan, OpenBLAS dgemv), which is what makes the integer outputs
// bit-exact.  Stage -> reference map: see include/d2d.h D2D_ST_*.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

// Device pass: the pointers of d2d_state / d2d_plan are global-address-space pointers.  The persistent kernel reads them out
// of a struct in memory; as plain (generic) pointers every access through them is a `flat_*` instruction, which counts on
// vmcnt AND lgkmcnt -- each LDS hand-off wait then also waits for all global traffic in flight -- and takes a 64-bit VGPR
// address.  (Kernel ARGUMENTS are inferred global by the compiler; loaded pointers are not.)
#if defined(__HIP_DEVICE_COMPILE__)
#define D2D_AS __attribute__((address_space(1)))
#endif
#include "../../include/d2d.h"

#define D2D_TAN_QUAL __device__ __forceinline__
#define D2D_TAN_TBL_QUAL __device__ const
#include "d2d_tan.h"

#define WAVE 64
#define WAVES_PER_BLOCK 4
#ifndef D2D_MIN_WAVES
#define D2D_MIN_WAVES 4  // waves per SIMD the register allocator must leave room for (4096 envs = 4 waves per SIMD)
#endif

#ifdef D2D_CHAIN_PROF
// Diagnostic build only (tools/chain_prof.py): shader clocks per phase of the persistent loop, [B][8]: gaze, perceive, planner's
// every-step part, search, act, then the sections of the gaze stage (GZ), summed over the steps of a launch; [B][16]
__device__ unsigned long long *d2d_phase_buf = nullptr;
#define D2D_PHASE_ADD(idx, t0)                                                                                        \
  do {                                                                                                                \
    if (d2d_phase_buf && (threadIdx.x & (WAVE - 1)) == 0) d2d_phase_buf[(size_t)e * 16 + (idx)] += __builtin_amdgcn_s_memtime() - (t0); \
  } while (0)
// sections of the gaze stage (drains the memory counters: shares, not absolute times)
#define GZ(idx)                                                                                  \
  do {                                                                                           \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                  \
    const unsigned long long gz_now = __builtin_amdgcn_s_memtime();                              \
    if (d2d_phase_buf && lane == 0) d2d_phase_buf[(size_t)e * 16 + 5 + (idx)] += gz_now - gz_t;  \
    gz_t = gz_now;                                                                               \
  } while (0)
#else
#define GZ(idx) do { } while (0)
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

// Exact fmod(a, b) for a >= 0, b > 0 (a / b far below 2^52): the true remainder is representable, so once
// the quotient is right the fused multiply-add returns it without rounding.  The quotient estimate uses a
// multiplication by 1/b (no fp64 division); it can be off by one, which the two fix-ups absorb.  No loop: a
// wild input (inf / NaN / huge yaw written by a caller) must not be able to hang a wave.
__device__ __forceinline__ double fmod_pos(double a, double b, double inv_b) {
  double n = trunc(a * inv_b);
  double m = __builtin_fma(-n, b, a);
  if (m < 0.0) m = __builtin_fma(-(n - 1.0), b, a);
  else if (m >= b) m = __builtin_fma(-(n + 1.0), b, a);
  return m;
}

// Python float `a % 360.0` (utils.py:743): fmod, then the sign fix-up with one rounded add.
__device__ __forceinline__ double py_mod360(double a) {
  const double m360 = 360.0;
  double m = copysign(fmod_pos(fabs(a), m360, 0x1.6c16c16c16c17p-9 /* 1/360 */), a);
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
  // any 0 <= idx < 2^24 (cell indices of a large grid: the float quotient can be one off), with the fix-up
  __device__ __forceinline__ void divmod_big(int idx, int &q, int &r) const {
    q = (int)(((float)idx + 0.5f) * inv);
    r = idx - q * d;
    if (r < 0) {
      q -= 1;
      r += d;
    } else if (r >= d) {
      q += 1;
      r -= d;
    }
  }
};

struct LdsView {
  double *ax, *ay, *ar, *ar2;    // all agents of the env after this step's move   [ncap]
  double *cx, *cy, *cr2, *crr;   // compacted ray candidates (centre, r^2, r)      [ccap]
  double *kf;                    // tracker state staged in batch 1 by DMA         [ncap][20] (if g.kf_lds)
  int *cidx;                     // candidate -> agent index                        [ccap]
  int *klen;                     // len(tracker.ts)                                 [ncap]
  short *ncx, *ncy, *nu;         // new dynamic block of every agent (cell, half)   [ncap]  (grids are at most 32767 cells
  short *pcx, *pcy, *pu;         // block written last time (dyn_prev)              [ncap]   wide: check())
  unsigned int *bm;              // bitmap of cells covered by some agent's new block [bmw]
  unsigned int *gtw;             // ground-truth window tile, bytes                 [ws][ws]
  unsigned int *dmt;             // drone-map crop tile, bytes                      [L][L]
  unsigned char *hit, *act;      // per-agent hit flag (OR over rays), tracker.active [ncap]
};

struct Geom {
  int ncap;   // N rounded up to a multiple of 4
  int ccap;   // ray candidates kept in LDS: min(ncap, 64).  An env with more candidates than that (a crowd inside the cone of
              // the rays) tests every agent at every sample instead, as the reference does.
  int reach;  // cells a ray can travel from the drone cell
  int ws;     // window edge = 2 * reach + 1
  int wdw;    // dwords of the window tile
  int ldw;    // dwords of the crop tile
  int smax;   // samples after which every ray has stopped: sample k is >= k * ss from the drone
  int klo;    // samples 0..klo are < depth from the drone whatever the slope (k * ss * sqrt(2) < depth)
  int bmw;    // dwords of the per-env cell bitmap kept in LDS (0: none)
  int dyn2p;  // grid too large for a bitmap (above 256 x 256 cells): the dynamic-grid update runs in two phases instead (dyn_clear,
              // dyn_mark) and needs no coverage structure at all
  int kf_lds; // tracker state staged in LDS (fits the 64 KB workgroup budget)
  int full;   // both grids staged WHOLE in LDS (the specialised 50 x 50 geometry): gtw / dmt are the full copies, no tiles, no bitmap
  int wave_bytes;
};

// `wpb`: waves (envs) per workgroup; the LDS budget of a workgroup is 64 KB
__host__ __device__ inline Geom make_geom(const d2d_cfg &c, int wpb, int ncap_fixed = 0, bool full = false) {
  Geom g;
  g.full = full ? 1 : 0;
  g.ncap = (c.N + 3) & ~3;
  if (g.ncap < 4) g.ncap = 4;
  if (ncap_fixed) g.ncap = ncap_fixed;
  g.ccap = g.ncap < 64 ? g.ncap : 64;
  // The whole-grid kernel for 17-40 agents addresses its candidates through a 32-bit mask anyway: 32 candidate slots instead of 40
  // bring its wave to 8 176 B of LDS -- FIVE 4-wave workgroups per CU instead of four (its 93 VGPRs since the four-lane Kalman stage
  // allow five waves per SIMD); an env with 33-40 candidates inside the cone tests every agent at every sample, as one with more
  // than `ccap` always did.
  if (full && g.ccap > 32) g.ccap = 32;
  g.reach = (int)((c.depth + 1.5 * (c.scale - 1.0)) / c.scale) + 2;
  g.ws = 2 * g.reach + 1;
  g.wdw = (g.ws * g.ws + 3) / 4;   // dwords of the window tile
  g.ldw = (c.L * c.L + 3) / 4;     // dwords of the crop tile
  const double ss = c.scale - 1.0;
  g.smax = (int)(c.depth / ss) + 2;
  const double kl = c.depth / (ss * 1.4142136);
  g.klo = (int)kl - ((kl == (double)(int)kl) ? 1 : 0);
  if (g.klo < -1) g.klo = -1;
  g.bmw = (c.W * c.H + 31) / 32;
  g.dyn2p = 0;
  if (g.bmw > 2048) {  // a bitmap above 8 KB per wave (256 x 256 cells): the two-phase update, no LDS
    g.dyn2p = 1;
    g.bmw = 0;
  }
  if (full) {  // whole grids: W * H bytes each (rounded to 16), the dynamic-grid coverage marks live in the gt copy itself
    g.wdw = ((c.W * c.H + 15) & ~15) / 4;
    g.ldw = g.wdw;
    g.bmw = 0;
    g.dyn2p = 0;
  }
  // per agent: 4 doubles, klen, 6 shorts, hit + act; per candidate: 4 doubles + index
  const int base = (32 + 4 + 12 + 2) * g.ncap + 36 * g.ccap + 4 * g.bmw + 4 * g.wdw + 4 * g.ldw;
  // tracker state staged in LDS only while the wave stays within the 10 KB that keep 16 waves on a CU: beyond that the
  // occupancy it would cost is worth more than the staging (the trackers then come straight from global memory)
#ifndef D2D_LDS_WAVE_BUDGET
#define D2D_LDS_WAVE_BUDGET 10240
#endif
  g.kf_lds = (c.kf_enabled && ((base + 160 * g.ncap + 15) & ~15) <= D2D_LDS_WAVE_BUDGET) ? 1 : 0;
  // The whole-grid kernel for <= 16 agents (SPEC 1) runs its trackers with a lane per ELEMENT of a tracker's state (st_tracker_elem):
  // 83 VGPRs, and with the tracker block left in global memory 6 400 B of LDS per wave -- five waves per SIMD instead of four
  // (the fused step at 65 536 envs: 239 -> 218 us; the 2 560-byte block was what kept a launch at 16 waves per CU).
  if (full && g.ncap <= 16) g.kf_lds = 0;
  (void)wpb;
  g.wave_bytes = (base + (g.kf_lds ? 160 * g.ncap : 0) + 15) & ~15;
  return g;
}

__device__ __forceinline__ LdsView carve(char *base, const Geom &g, int Lm) {
  LdsView L;
  L.ax = (double *)base;
  L.ay = L.ax + g.ncap;
  L.ar = L.ay + g.ncap;
  L.ar2 = L.ar + g.ncap;
  L.cx = L.ar2 + g.ncap;
  L.cy = L.cx + g.ccap;
  L.cr2 = L.cy + g.ccap;
  L.crr = L.cr2 + g.ccap;
  L.kf = L.crr + g.ccap;
  L.cidx = (int *)(L.kf + (g.kf_lds ? 20 * g.ncap : 0));
  L.klen = L.cidx + g.ccap;
  L.bm = (unsigned int *)(L.klen + g.ncap);
  L.gtw = L.bm + g.bmw;
  L.dmt = L.gtw + g.wdw;
  L.ncx = (short *)(L.dmt + g.ldw);  // ncap is a multiple of 4: every plane below stays 8-byte aligned
  L.ncy = L.ncx + g.ncap;
  L.nu = L.ncy + g.ncap;
  L.pcx = L.nu + g.ncap;
  L.pcy = L.pcx + g.ncap;
  L.pu = L.pcy + g.ncap;
  L.hit = (unsigned char *)(L.pu + g.ncap);
  L.act = L.hit + g.ncap;
  return L;
}

struct EnvRegs {  // lane-uniform per-env scalars carried in registers across the fused stages
  double x, y, yaw, vx, vy, ax, ay;
  double tx, ty;
  int steps, fail, sm, tnext, ntgt, tracked, bufn, bufts;
  int done = -1;  // flags[D2D_F_DONE] as the collision stage of THIS call wrote it; -1: that stage did not run
};

// Python / numpy `int(v // s)` for integer-valued s > 0: the exact mathematical floor.  floor(v * (1/s))
// is within one of it; the fused remainder v - q s (exact when q is right) settles which.
__device__ __forceinline__ int cell_fast(double v, double s, double inv_s) {
  const double q = floor(v * inv_s);
  const double r = __builtin_fma(-q, s, v);
  // selects, not branches (ten of these sit in one collision test): at most one of the two corrections applies
  return (int)q + ((r >= s) ? 1 : 0) - ((r < 0.0) ? 1 : 0);
}

// floor(v / s) for 0 < v < 2^24 and integer s >= 2: floor(v / s) == floor(floor(v) / s), and floor(v) fits
// an int, so the rest is 24-bit integer work (full-rate VALU) instead of fp64.
struct CellDiv {
  float inv;
  int s;
  bool by10;  // scale 10 on a map below 81920 px: floor(n / 10) == (n * 52429) >> 19 exactly, in 32-bit arithmetic
  __device__ __forceinline__ CellDiv(double scale, double max_px)
      : inv(1.0f / (float)scale), s((int)scale), by10(scale == 10.0 && max_px <= 81919.0) {}
  __device__ __forceinline__ int operator()(double v) const {
    const int vi = __double2int_rz(v);  // truncation == floor for v >= 0; saturates / 0 for wild or NaN inputs
    // a live sample lies inside the map (0 <= vi < 81920 < 2^24, product < 2^32): a full-rate 24-bit multiply; a dead lane's
    // garbage is masked to 24 bits by the instruction and ignored by the caller
    if (by10) return (int)(__umul24((unsigned int)vi, 52429u) >> 19);
    int q = (int)((float)vi * inv);
    const int r = vi - q * s;
    q += (r >= s) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
  }
};

// Byte offset of cell (i, j) inside one env's grid (d2d_cfg.grid_tile).  Row-major [W][H] (the reference's indexing, utils.py:548,
// and the ABI default), or -- TILED -- 16 x 16-cell tiles of 256 contiguous bytes, tiles row-major over (ceil(W / 16), ceil(H / 16)),
// cells row-major inside a tile: on grids of hundreds of cells a side every 3 x 3 block, 23-byte window row and 33-byte crop row of
// the row-major layout costs a cache line of its own (640-byte row stride); tiled, a 3 x 3 block lies in one or two lines, a 23 x 23
// window in at most nine tiles.  0 <= i < W, 0 <= j < H.
template <bool TILED>
struct GridIx {
  static constexpr bool tiled = TILED;
  int H, Ht;
  __device__ __forceinline__ int operator()(int i, int j) const {
    if constexpr (TILED) return ((((i >> 4) * Ht + (j >> 4)) << 8) | ((i & 15) << 4) | (j & 15));
    else return i * H + j;
  }
};
__host__ __device__ inline size_t grid_bytes(const d2d_cfg &c) {
  return c.grid_tile ? (size_t)((c.W + 15) >> 4) * (size_t)((c.H + 15) >> 4) * 256 : (size_t)c.W * c.H;
}
// run-time form for the few grid reads of the plugin stages
__device__ __forceinline__ int grid_ix(const d2d_cfg &c, int i, int j) {
  return c.grid_tile ? ((((i >> 4) * ((c.H + 15) >> 4) + (j >> 4)) << 8) | ((i & 15) << 4) | (j & 15)) : i * c.H + j;
}

// ------------------------------------------------------------------------------------------------
// LDS tiles of a uint8 grid
// ------------------------------------------------------------------------------------------------
// A tile is the `rows` x `cols` block of cells from cell (i0, j0), one byte per cell, row-major, cells outside
// the grid replaced by `fill`.  (An LDS-DMA variant with dword-aligned rows was measured: its per-dword
// address arithmetic costs more wave instructions than these byte loads, and the kernel is bound by
// instruction issue, not by VGPRs or bytes.)
struct Tile {
  int i0, j0, rows, cols;
  __device__ __forceinline__ int byte_index(int r, int q) const { return r * cols + q; }
};

__device__ __forceinline__ Tile make_tile(int i0, int j0, int rows, int cols) {
  Tile t;
  t.i0 = i0; t.j0 = j0; t.rows = rows; t.cols = cols;
  return t;
}

// Two tiles at once, two rows per wave instruction: lanes 0-31 take row 2t, lanes 32-63 row 2t + 1, lane & 31 is
// the column (tile A: cols <= 32; tile B: cols <= 33, its column 32 is swept by a last pass with lane = row).
// MAXA / MAXB bound the row pairs held in registers (12 and 17: 23- and 33-row tiles).
template <typename GX>
__device__ __forceinline__ void tile_rows2(const GX &gx, const Tile &ta, unsigned char *la, const unsigned char *__restrict__ ga, int W,
                                           int H, int lane, unsigned char fa, const Tile &tb, unsigned char *lb,
                                           const unsigned char *__restrict__ gb, unsigned char fb) {
  constexpr int MAXA = 12, MAXB = 17;
  const int half = lane >> 5, col = lane & 31;
  unsigned char va[MAXA], vb[MAXB], vc = 0;
  {
    const int j = ta.j0 + col, jc = min(max(j, 0), H - 1);
    const bool jok = col < ta.cols && j >= 0 && j < H;
#pragma unroll
    for (int t = 0; t < MAXA; ++t) {
      const int r = 2 * t + half, i = ta.i0 + r;
      const unsigned char g = ga[gx(min(max(i, 0), W - 1), jc)];
      va[t] = (jok && i >= 0 && i < W) ? g : fa;
    }
  }
  {
    const int j = tb.j0 + col, jc = min(max(j, 0), H - 1);
    const bool jok = col < tb.cols && j >= 0 && j < H;
#pragma unroll
    for (int t = 0; t < MAXB; ++t) {
      const int r = 2 * t + half, i = tb.i0 + r;
      const unsigned char g = gb[gx(min(max(i, 0), W - 1), jc)];
      vb[t] = (jok && i >= 0 && i < W) ? g : fb;
    }
    if (tb.cols > 32) {  // column 32 of every row, lane = row
      const int i = tb.i0 + lane, j32 = tb.j0 + 32;
      const unsigned char g = gb[gx(min(max(i, 0), W - 1), min(max(j32, 0), H - 1))];
      vc = (i >= 0 && i < W && j32 >= 0 && j32 < H) ? g : fb;
    }
  }
#pragma unroll
  for (int t = 0; t < MAXA; ++t) {
    const int r = 2 * t + half;
    if (r < ta.rows && col < ta.cols) la[r * ta.cols + col] = va[t];
  }
#pragma unroll
  for (int t = 0; t < MAXB; ++t) {
    const int r = 2 * t + half;
    if (r < tb.rows && col < tb.cols) lb[r * tb.cols + col] = vb[t];  // tiles narrower than 32 columns: crop edge L < 32
  }
  if (tb.cols > 32 && lane < tb.rows) lb[lane * tb.cols + 32] = vc;
}

// loads of up to CH x 64 cells are issued back to back, then written to LDS
template <int CH, typename GX>
__device__ __forceinline__ void tile_load(const GX &gx, const Tile &t, unsigned char *lds, const unsigned char *__restrict__ grid, int W,
                                          int H, int lane, unsigned char fill) {
  const FastDiv fd(t.cols);
  const int n = t.rows * t.cols;
  for (int base = 0; base < n; base += CH * WAVE) {
    unsigned char v[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int idx = base + u * WAVE + lane;
      int r, q;
      fd.divmod(idx, r, q);
      const int i = t.i0 + r, j = t.j0 + q;
      const unsigned char g = grid[gx(min(max(i, 0), W - 1), min(max(j, 0), H - 1))];
      v[u] = (i >= 0 && i < W && j >= 0 && j < H) ? g : fill;
    }
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int idx = base + u * WAVE + lane;
      if (idx < n) lds[idx] = v[u];
    }
  }
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

// Planner result + gaze action of this step, loaded in batch 1
struct StepIn {
  double action;
  double wp[6];
  bool ok, has_wp;
};

// The planner's result as it is loaded (load_inputs_raw) and as the control stage reads it (finish_inputs): two steps so that the
// caller can put everything else it requests from memory between them.
struct StepInRaw {
  unsigned char okb, wvb;
  double w[6];
  bool use;
};

__device__ __forceinline__ void load_inputs_raw(const d2d_cfg &c, const d2d_state &s, int e, bool control, StepInRaw &raw) {
  // No branch around the loads (a branch ends the block: its loads are waited for before it is left, a round trip ahead of
  // everything requested after it): without a planner the same loads read bytes of the env's own state and are ignored.
  raw.use = control && c.planner_mode != D2D_PLANNER_NOMOVE;
  const unsigned char *some_byte = s.flags + (size_t)e * 4;
  const unsigned char *pok = raw.use ? s.plan_ok + e : some_byte, *pwv = raw.use ? s.wp_valid + e : some_byte;
  const double *wp = raw.use ? s.wp + (size_t)e * 6 : s.drone + (size_t)e * D2D_DF;  // (D2D_DF >= 6 doubles)
  raw.okb = *pok;
  raw.wvb = *pwv;
#pragma unroll
  for (int i = 0; i < 6; ++i) raw.w[i] = wp[i];
}

__device__ __forceinline__ void finish_inputs(const StepInRaw &raw, double action, StepIn &in) {
  in.action = action;
  in.ok = raw.use ? raw.okb != 0 : true;
  in.has_wp = raw.use ? raw.wvb != 0 : false;
#pragma unroll
  for (int i = 0; i < 6; ++i) in.wp[i] = raw.w[i];  // (read only where has_wp holds)
}

// the same in one step, with a branch around the loads: a launch of k_stages (run_env without `load_r`)
__device__ __forceinline__ void load_inputs(const d2d_cfg &c, const d2d_state &s, int e, double action, bool control,
                                            StepIn &in) {
  in.action = action;
  in.ok = true;
  in.has_wp = false;
#pragma unroll
  for (int i = 0; i < 6; ++i) in.wp[i] = 0.0;
  if (control && c.planner_mode != D2D_PLANNER_NOMOVE) {
    in.ok = s.plan_ok[e] != 0;
    in.has_wp = s.wp_valid[e] != 0;
    const double *wp = s.wp + (size_t)e * 6;
#pragma unroll
    for (int i = 0; i < 6; ++i) in.wp[i] = wp[i];
  }
}

// envs/drone_v2.py:197-214 with utils.py:733-743, 755-762 (lane-uniform scalar work)
__device__ __forceinline__ void st_control(const d2d_cfg &c, const StepIn &in, EnvRegs &r) {
  if (c.planner_mode == D2D_PLANNER_NOMOVE) {
    r.tx = -1.0;  // traj_planner.py:72
    r.ty = -1.0;
  }
  if (!in.ok) {
    const double n = sqrt(__builtin_fma(r.vy, r.vy, r.vx * r.vx));  // numpy norm: sqrt(ddot) = sqrt(fma(y, y, x * x))
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
  if (in.has_wp) {
    r.ax = in.wp[4];
    r.ay = in.wp[5];
    r.vx = in.wp[2];
    r.vy = in.wp[3];
    r.x = rint(in.wp[0]);  // round(): half to even
    r.y = rint(in.wp[1]);
  }
  r.yaw = py_mod360(r.yaw + in.action * c.yaw_rate * c.dt);
}

// envs/drone_v2.py:176-179 + utils.py:472-493; lane = agent.  Batch-1 loads of everything per-agent
// (agent planes, unit, previous dynamic block, tracker active / len / state) are issued together; the
// moved agents and the staged tracker data land in LDS for the later stages.  With `move` false the
// agents are only staged (a launch without the AGENTS stage).
// `light`: only the collision test / the trackers follow (no rays, no dynamic grid): positions, radii and tracker flags.
struct AgentIn {  // what the agents stage reads of one agent
  double px, py, velx, vely, rr, r2;
  int u, p0, p1, p2, klen;
  unsigned char act;
};

// Loads of agent `k` (clamped into [0, N): callers pass any lane), WITHOUT a branch around them -- a block that ends behind its
// loads is left only when they have arrived, a round trip ahead of whatever the caller requests next.  An env without agents reads
// element 0 of its own pose / counters / flags instead (never used).
// ANY_LANE false: the plain form for a caller that only comes here with k < N (k_stages' loop over the agents).
template <bool ANY_LANE = true>
__device__ __forceinline__ void agent_load(const d2d_cfg &c, const d2d_state &s, int e, int k, bool want_trk, AgentIn &a,
                                           bool light = false) {
  const int N = c.N;
  const bool has = !ANY_LANE || N > 0, klen_on = has && want_trk && c.kf_enabled;
  const int kk = ANY_LANE ? min(k, max(N, 1) - 1) : k;
  const double *__restrict__ ag = has ? s.agents + (size_t)e * D2D_AF * N : s.drone + (size_t)e * D2D_DF;
  const int *some_int = s.counters + (size_t)e * D2D_CF;
  const int *__restrict__ prev = has ? s.dyn_prev + (size_t)e * N * 3 : some_int;
  const int *unitp = has ? s.agent_unit + (size_t)e * N : some_int;
  const int *klp = (!ANY_LANE || klen_on) ? s.kf_len + (size_t)e * N : some_int;
  const unsigned char *actp = has ? s.active + (size_t)e * N : s.flags + (size_t)e * 4;
  a.px = ag[D2D_A_PX * N + kk];
  a.py = ag[D2D_A_PY * N + kk];
  a.rr = ag[D2D_A_R * N + kk];
  a.act = actp[kk];
  a.velx = a.vely = a.r2 = 0.0;
  a.u = a.p0 = a.p1 = a.p2 = 0;
  if (!light) {  // (a constant in the persistent loop's phases; a branch -- cheap -- in a k_stages launch of the light stages alone)
    a.velx = ag[D2D_A_VX * N + kk];
    a.vely = ag[D2D_A_VY * N + kk];
    a.r2 = ag[D2D_A_R2 * N + kk];
    a.u = unitp[kk];
    a.p0 = prev[3 * kk];
    a.p1 = prev[3 * kk + 1];
    a.p2 = prev[3 * kk + 2];
  }
  if constexpr (ANY_LANE) {
    const int kl = klp[klen_on ? kk : 0];  // (the stand-in pointer holds D2D_CF ints, not N: only its element 0 may be touched)
    a.klen = klen_on ? kl : 1;
  } else {
    a.klen = 1;
    if (klen_on) a.klen = klp[kk];
  }
}

// envs/drone_v2.py:176-179 + utils.py:472-493 for agent k (loaded by agent_load): moved (`move`), written back and staged in LDS
// for the later stages.  `light`: only the collision test / the trackers follow (no rays, no dynamic grid): positions, radii and
// tracker flags.
__device__ __forceinline__ void agent_apply(const d2d_cfg &c, const d2d_state &s, int e, int k, const LdsView &L, double inv_scale,
                                            bool move, bool light, const AgentIn &a) {
  const int N = c.N;
  double *__restrict__ ag = s.agents + (size_t)e * D2D_AF * N;
  const double cs = 0x1.bb67ae8584cabp-1, sn = 0x1.fffffffffffffp-2;  // cos(pi/6), sin(pi/6)
  double px = a.px, py = a.py;
  const double velx = a.velx, vely = a.vely, rr = a.rr;
  if (light) {
    L.ax[k] = px;
    L.ay[k] = py;
    L.ar[k] = rr;
    L.act[k] = a.act;
    L.klen[k] = a.klen;
    return;
  }
  if (move) {
    const double nx = px + velx * c.dt, ny = py + vely * c.dt;
    bool aliased = true;
    double pvx = velx, pvy = vely;
    // norm(v) <= 5 (utils.py:476), numpy's norm = sqrt(fma(vy, vy, vx * vx)) (OpenBLAS ddot).  sqrt is correctly
    // rounded and monotonic, and sqrt(s) rounds to <= 5 exactly for s <= nextafter(25) = 0x1.9000000000001p+4
    // (checked on the host), so the fp64 sqrt is not needed.
    if (__builtin_fma(vely, vely, velx * velx) <= 0x1.9000000000001p+4) {
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
  L.ar2[k] = a.r2;
  L.ncx[k] = cell_fast(px, c.scale, inv_scale);
  L.ncy[k] = cell_fast(py, c.scale, inv_scale);
  L.nu[k] = a.u;
  L.pcx[k] = a.p0;
  L.pcy[k] = a.p1;
  L.pu[k] = a.p2;
  L.act[k] = a.act;
  L.klen[k] = a.klen;
}

// lane = agent, from agent `k_first` on (the caller may have done the first 64 itself, loads and stores apart).  Batch-1 loads of
// everything per-agent (agent planes, unit, previous dynamic block, tracker active / len) are issued together.  With `move` false
// the agents are only staged (a launch without the AGENTS stage).
__device__ __forceinline__ void st_agents(const d2d_cfg &c, const d2d_state &s, int e, int lane, const Geom &g,
                                          const LdsView &L, double inv_scale, bool move, bool want_trk, bool light = false,
                                          int k_first = 0) {
  for (int k = k_first + lane; k < c.N; k += WAVE) {
    AgentIn a;
    agent_load(c, s, e, k, want_trk, a, light);
    agent_apply(c, s, e, k, L, inv_scale, move, light, a);
  }
}

// The agents stage as one loop, loads and stores together: a launch of k_stages (run_env without `load_r`).
__device__ __forceinline__ void st_agents_plain(const d2d_cfg &c, const d2d_state &s, int e, int lane, const Geom &g,
                                          const LdsView &L, double inv_scale, bool move, bool want_trk, bool light = false) {
  const int N = c.N;
  double *__restrict__ ag = s.agents + (size_t)e * D2D_AF * N;
  const int *__restrict__ prev = s.dyn_prev + (size_t)e * N * 3;
  const double cs = 0x1.bb67ae8584cabp-1, sn = 0x1.fffffffffffffp-2;  // cos(pi/6), sin(pi/6)
  for (int k = lane; k < N; k += WAVE) {
    double px = ag[D2D_A_PX * N + k], py = ag[D2D_A_PY * N + k];
    const double velx = ag[D2D_A_VX * N + k], vely = ag[D2D_A_VY * N + k];
    const double rr = ag[D2D_A_R * N + k];
    const unsigned char act = s.active[(size_t)e * N + k];
    if (light) {
      int kl = 1;
      if (want_trk && c.kf_enabled) kl = s.kf_len[(size_t)e * N + k];
      L.ax[k] = px;
      L.ay[k] = py;
      L.ar[k] = rr;
      L.act[k] = act;
      L.klen[k] = kl;
      continue;
    }
    const double r2 = ag[D2D_A_R2 * N + k];
    const int u = s.agent_unit[(size_t)e * N + k];
    const int p0 = prev[3 * k], p1 = prev[3 * k + 1], p2 = prev[3 * k + 2];
    int klen = 1;
    if (want_trk && c.kf_enabled) klen = s.kf_len[(size_t)e * N + k];
    if (move) {
      const double nx = px + velx * c.dt, ny = py + vely * c.dt;
      bool aliased = true;
      double pvx = velx, pvy = vely;
      // norm(v) <= 5 (utils.py:476), numpy's norm = sqrt(fma(vy, vy, vx * vx)) (OpenBLAS ddot).  sqrt is correctly
      // rounded and monotonic, and sqrt(s) rounds to <= 5 exactly for s <= nextafter(25) = 0x1.9000000000001p+4
      // (checked on the host), so the fp64 sqrt is not needed.
      if (__builtin_fma(vely, vely, velx * velx) <= 0x1.9000000000001p+4) {
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
    L.ncx[k] = cell_fast(px, c.scale, inv_scale);
    L.ncy[k] = cell_fast(py, c.scale, inv_scale);
    L.nu[k] = u;
    L.pcx[k] = p0;
    L.pcy[k] = p1;
    L.pu[k] = p2;
    L.act[k] = act;
    L.klen[k] = klen;
  }
}

// Tracker state of the env -> LDS by DMA (the env's [N][20] block is contiguous: 16-byte pieces, no VGPRs).
// The caller's vmcnt(0) before the first tile read also covers this.
__device__ __forceinline__ void kf_stage(const d2d_cfg &c, const d2d_state &s, int e, int lane, const LdsView &L) {
  const int n16 = c.N * (D2D_KF * 8 / 16);
  const char *src = (const char *)(s.kf + (size_t)e * c.N * D2D_KF);
  for (int base = 0; base < n16; base += WAVE) {
    const int idx = base + lane;
    if (idx < n16)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (size_t)idx * 16),
                                       (__attribute__((address_space(3))) void *)((char *)L.kf + (size_t)base * 16), 16, 0, 0);
  }
}

// utils.py:612-618
__device__ __forceinline__ double positive_angle(double a) {
  const double two_pi = 0x1.921fb54442d18p+2;  // math.pi * 2
  a = copysign(fmod_pos(fabs(a), two_pi, 0x1.45f306dc9c883p-3 /* 1/(2 pi) */), a);
  if (a < 0.0) a += two_pi;
  return a;
}

// candidate compaction for the raycast (utils.py:658-662 tests every agent; only these can pass).
// CONE (the kernels for many agents): besides the disc of radius depth + ..., an agent must come within its radius of the
// CONE the rays span -- samples lie on the rays, the rays inside the convex cone between the first and the last one, and a
// point farther than r outside either edge's half-plane is farther than r from the cone.  The edge directions come from the
// fast float sine / cosine (|error| < 1e-5 of a direction = 1e-3 px at the end of a ray) under a margin of 0.05 px; the
// exact per-sample tests decide among the survivors as before.  A view of 90 degrees keeps about a third of the disc's agents.
template <bool CONE>
__device__ __forceinline__ int ray_cull(const d2d_cfg &c, int lane, const LdsView &L, double x0, double y0, double yaw0,
                                        int ccap) {
  const int N = c.N;
  const double ss = c.scale - 1.0;
  int ncand = 0;
  double lox = 0, loy = 0, hix = 0, hiy = 0;
  bool cone = false;
  if constexpr (CONE) {
    const double span = c.ray_dth * (double)(c.R - 1);
    const double pa = 0x1.921fb54442d18p+2 - yaw0 * 0x1.1df46a2529d39p-6;  // as ray_setup
    // convex (below 166 degrees), first ray clockwise of the last; a yaw far outside [0, 360) (or NaN) keeps every agent of the
    // disc: the float angle would be too coarse
    cone = span > 0.0 && span < 2.9 && fabs(pa) < 64.0;
    const float tlo = (float)(pa + c.ray_off0), thi = (float)(pa + c.ray_off0 + span);
    lox = (double)__cosf(tlo); loy = (double)__sinf(tlo);
    hix = (double)__cosf(thi); hiy = (double)__sinf(thi);
  }
  for (int k0 = 0; k0 < N; k0 += WAVE) {
    const int k = k0 + lane;
    bool cand = false;
    double px = 0, py = 0, r2 = 0, rr = 0;
    if (k < N) {
      px = L.ax[k];
      py = L.ay[k];
      r2 = L.ar2[k];
      rr = fabs(L.ar[k]);
      // a sample is < depth + sqrt(2) * ss from the drone, so |agent - drone| <= radius + that bound
      const double lim = rr + c.depth + 1.5 * ss + 2.0;
      const double dx = px - x0, dy = py - y0;
      cand = (dx * dx + dy * dy <= lim * lim);
      if constexpr (CONE) {
        // outward distance from the edge lines: -cross(lo, d) and -cross(d, hi)
        const double olo = loy * dx - lox * dy, ohi = dy * hix - dx * hiy, reach = rr + 0.05;
        cand = cand && (!cone || (olo <= reach && ohi <= reach));
      }
      L.hit[k] = 0;
    }
    const unsigned long long m = __ballot(cand);
    const int slot = ncand + __popcll(m & ((1ull << lane) - 1ull));
    if (cand && slot < ccap) {  // beyond the capacity the count alone matters: such an env tests every agent (Geom.ccap)
      L.cx[slot] = px;
      L.cy[slot] = py;
      L.cr2[slot] = r2;
      L.crr[slot] = rr;
      L.cidx[slot] = k;
    }
    ncand += __popcll(m);
  }
  return ncand;
}

// Per-ray setup (utils.py:594,626-650): direction steps and the conservative candidate mask.
// `M`: the mask type = how many candidates of an env the mask path takes (32, or 64 where an env can have more than 32
// agents: at BASELINE config 3's 172 agents a fifth of the env-steps have 33..52 candidates, and without a mask every sample
// of every ray would test all of them).
template <typename M>
struct RayT {
  double xs, ys;
  M cmask;
};
template <typename M>
__device__ __forceinline__ int mask_first(M m) {  // index of the lowest set bit
  if constexpr (sizeof(M) == 8) return __ffsll((long long)m) - 1;
  else return __ffs((int)m) - 1;
}

template <typename M>
__device__ __forceinline__ RayT<M> ray_setup(const d2d_cfg &c, const LdsView &L, int i, int ncand, double x0, double y0,
                                             double yaw0) {
  const double ss = c.scale - 1.0;  // x_step_size, utils.py:621
  const double rad90 = 0x1.921fb54442d18p+0, rad270 = 0x1.2d97c7f3321d2p+2;  // radians(90), radians(270)
  const double pi_ = 0x1.921fb54442d18p+1;
  const double player_angle = 0x1.921fb54442d18p+2 - yaw0 * 0x1.1df46a2529d39p-6;  // pi*2 - radians(yaw)
  const double ang = positive_angle(player_angle + (c.ray_off0 + c.ray_dth * (double)i));
  const bool faced_right = (ang < rad90 || ang > rad270);
  const bool faced_up = (ang > pi_);
#ifdef D2D_ABL_NOTAN
  double slope = ang * 0.3;
#else
  double slope = d2d_tan(ang);
#endif
  RayT<M> ry;
  if (fabs(slope) > 1.0) {
    slope = 1.0 / slope;
    ry.ys = faced_up ? -ss : ss;
    ry.xs = ry.ys * slope;
  } else {
    ry.xs = faced_right ? ss : -ss;
    ry.ys = ry.xs * slope;
  }
  // An agent can only be hit by this ray if its centre is within radius of the ray's LINE and not behind
  // the drone.  Conservative (margins cover the rounding of the iterated sample positions); the exact
  // per-sample circle test of the reference is then applied to the set bits only.
  ry.cmask = 0;
  if (ncand <= (int)(8 * sizeof(M))) {
    const double len2 = ry.xs * ry.xs + ry.ys * ry.ys, l1 = fabs(ry.xs) + fabs(ry.ys);
    for (int q = 0; q < ncand; ++q) {
      const double ex = L.cx[q] - x0, ey = L.cy[q] - y0, rq = L.crr[q] + 1e-6;
      const double cr = ex * ry.ys - ey * ry.xs, dt = ex * ry.xs + ey * ry.ys;
      const bool near_line = cr * cr <= rq * rq * len2 * (1.0 + 1e-9);
      const bool ahead = dt + rq * l1 >= 0.0;
      ry.cmask |= (near_line && ahead) ? (M)((M)1 << q) : (M)0;
    }
  }
  return ry;
}

// utils.py:620-713 for one ray (one lane).  Every lane runs the same fixed number of samples (g.smax: after
// that many every ray is past `depth`); positions are the reference's iterated sums; the per-sample
// decisions are predicated on `alive` instead of steering control flow, so consecutive samples' LDS
// lookups overlap and no lane waits for the slowest ray.  `dist >= depth^2` is only evaluated for samples
// k > klo (earlier ones are nearer than `depth` for any slope).
// GENERAL: some ray of the wave has more than one candidate (or the env more than 32): the per-sample LDS
// candidate loops are compiled in.  The common instantiation tests only the register-held first candidate.
// FULL: L.gtw / L.dmt are copies of the WHOLE grids (Geom.full), indexed by the cell itself -- no window / crop arithmetic.
template <bool GENERAL, bool FULL, typename M, typename GX>
__device__ __forceinline__ void ray_march(const GX &gx, const d2d_cfg &c, const Geom &g, const LdsView &L, const RayT<M> &ry, bool active,
                                          int ncand, double x0, double y0, const Tile &wt, const Tile &ct, bool patch,
                                          unsigned char *__restrict__ dm) {
  const int H = c.H;
  const CellDiv cell(c.scale, fmax(c.W_px, c.H_px));
  const double depth2 = c.depth * c.depth;
  const bool mask_path = ncand <= (int)(8 * sizeof(M));
  const unsigned char *gtw = (const unsigned char *)L.gtw;
  unsigned char *dmt = (unsigned char *)L.dmt;
  // sample 0 (the drone's own position, the same for every ray) was decided once by the caller
  double x = x0 + ry.xs, y = y0 + ry.ys;
  bool alive = active && (0.0 < x && x < c.W_px && 0.0 < y && y < c.H_px);
  const int klo = g.klo;
  // The first candidate of this ray's mask (almost always the only one) is tested from registers on every
  // sample, branch-free; further candidates (rare) go through the LDS loop.
  const bool has1 = mask_path && ry.cmask != (M)0;
  const int q1 = has1 ? mask_first(ry.cmask) : 0;
  const double c1x = L.cx[q1], c1y = L.cy[q1], c1r2 = has1 ? L.cr2[q1] : -1.0;
  const int c1i = L.cidx[q1];
  const M rest = mask_path ? (M)(ry.cmask & (ry.cmask - (M)1)) : (M)0;
  const bool some1 = __any(has1);
  auto sample = [&](auto far_tag) {
    constexpr bool FAR = decltype(far_tag)::value;
    // exact circle tests (utils.py:658-662): every candidate that can matter, no early-out among agents
    bool any = false;
    if (some1) {  // wave-uniform: no ray of this pass has a candidate at all (the usual case on a sparse map) -> nothing to test
      const double dx = c1x - x, dy = c1y - y;
      any = alive && (dx * dx + dy * dy <= c1r2);
      if (any) L.hit[c1i] = 1;
    }
    if (GENERAL && mask_path) {
      M m = alive ? rest : (M)0;
      while (m) {
        const int q = mask_first(m);
        m &= m - (M)1;
        const double dx = L.cx[q] - x, dy = L.cy[q] - y;
        if (dx * dx + dy * dy <= L.cr2[q]) {
          L.hit[L.cidx[q]] = 1;
          any = true;
        }
      }
    } else if (GENERAL && alive) {
      if (ncand <= g.ccap) {
        for (int q = 0; q < ncand; ++q) {
          const double dx = L.cx[q] - x, dy = L.cy[q] - y;
          if (dx * dx + dy * dy <= L.cr2[q]) {
            L.hit[L.cidx[q]] = 1;
            any = true;
          }
        }
      } else {  // a crowd: more candidates than the LDS list holds -- every agent, as utils.py:658-662 does
        for (int k = 0; k < c.N; ++k) {
          const double dx = L.ax[k] - x, dy = L.ay[k] - y;
          if (dx * dx + dy * dy <= L.ar2[k]) {
            L.hit[k] = 1;
            any = true;
          }
        }
      }
    }
    // The cell of this sample and its ground-truth value from the LDS tile.  Unconditional and clamped (a
    // live sample always lies inside the tile: the previous sample was nearer than `depth`; a dead lane's
    // garbage position just reads some tile byte that is then ignored); it must not become a select between
    // an LDS and a global pointer (flat load + vmcnt(0) wait per sample).
    const int ci = cell(x), cj = cell(y);
    unsigned int gi = 0;  // unsigned: a 32-bit byte offset on the env's scalar base pointer, no 64-bit address arithmetic per sample
    unsigned char wall;
    if constexpr (FULL) {
      gi = alive ? (unsigned int)(ci * H + cj) : 0u;  // a live sample lies inside the map (0 < x < W_px): the index is the cell's own
      wall = gtw[gi];
    } else {
      const int wr = min(max(ci - wt.i0, 0), wt.rows - 1), wq = min(max(cj - wt.j0, 0), wt.cols - 1);
      wall = gtw[wt.byte_index(wr, wq)];
    }
    bool far = false;
    if (FAR) far = ((x - x0) * (x - x0) + (y - y0) * (y - y0) >= depth2);
    const bool stop = (wall == D2D_OCCUPIED) || far;
    const bool write = alive && !any && (!stop || wall == D2D_OCCUPIED);
    if (write) {
      const unsigned char v = stop ? (unsigned char)D2D_OCCUPIED : (unsigned char)D2D_UNOCCUPIED;
      if constexpr (FULL) {
        dm[gi] = v;
        if (patch) dmt[gi] = v;  // the copy the observation crop is cut from
      } else {
#ifndef D2D_ABL_NOSTORE
        dm[(unsigned int)gx(ci, cj)] = v;
#endif
        const unsigned int pr = (unsigned int)(ci - ct.i0), pq = (unsigned int)(cj - ct.j0);  // observation tile in step
        if (patch && pr < (unsigned int)ct.rows && pq < (unsigned int)ct.cols) dmt[pr * ct.cols + pq] = v;
      }
    }
    alive = alive && !any && !stop;
    x = x + ry.xs;
    y = y + ry.ys;
    alive = alive && (0.0 < x && x < c.W_px && 0.0 < y && y < c.H_px);
  };
#ifndef D2D_ABL_NOMARCH
  // not unrolled (a 10x body overflows the instruction cache); samples 0..klo cannot be past `depth`
  const int k1 = min(klo + 1, g.smax);
#pragma unroll 1
  for (int k = 1; k < k1; ++k) sample(std::false_type{});
#pragma unroll 1
  for (int k = max(k1, 1); k < g.smax; ++k) sample(std::true_type{});
#endif
}

// utils.py:527-540, lane = agent.  The reference clears every cell of dynamic_idx (== every DYNAMIC cell,
// all of which lie in the blocks of dyn_prev) and then marks every agent's new block.  Written here as
// ONE order-independent pass: the final value of a cell depends only on (static or not, covered by some
// new block or not), so a lane may observe another lane's already-final value instead of the old one
// without changing the outcome -- no clear/set ordering, no memory fence.  Coverage comes from an LDS
// bitmap every agent ORs its new block into.
struct DynCells {  // the common case (blocks of at most 3 x 3 cells), reduced to what the update needs
  unsigned int pclr;   // bit q: previous-block cell q holds DYNAMIC and is not inside this agent's own new block
  unsigned int nfree;  // bit q: new-block cell q is neither static nor already DYNAMIC
};

// The LDS bitmap of the cells covered by some agent's new block (grids up to 256 x 256 cells; Geom.bmw == 0: none).
__device__ __forceinline__ void dyn_bitmap(const d2d_cfg &c, int lane, const Geom &g, const LdsView &L) {
  const int N = c.N, W = c.W, H = c.H;
  if (g.bmw == 0) return;
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
}

// 3 x 3 cells (bit q = (di + 1) * 3 + (dj + 1)) of a block of half-width u in {0, 1} that lie inside the grid
__device__ __forceinline__ unsigned int block_valid9(int cx, int cy, int u, int W, int H) {
  if (u == 0) return 0x010u;
  unsigned int rows = 0x2u | (cx - 1 >= 0 ? 0x1u : 0u) | (cx + 1 < W ? 0x4u : 0u);
  unsigned int cols = 0x2u | (cy - 1 >= 0 ? 0x1u : 0u) | (cy + 1 < H ? 0x4u : 0u);
  const unsigned int r9 = ((rows & 1u) ? 0x007u : 0u) | ((rows & 2u) ? 0x038u : 0u) | ((rows & 4u) ? 0x1C0u : 0u);
  return r9 & (cols | (cols << 3) | (cols << 6));
}

// PREV / NEW: which of the two blocks are fetched (both for the one-pass update; the two-phase update of large grids fetches the
// previous block before its first phase and the new block, again, after the fence between the phases).
template <bool PREV = true, bool NEW = true, typename GX>
__device__ __forceinline__ void dyn_load(const GX &gx, const d2d_cfg &c, const unsigned char *__restrict__ gt, int k, const LdsView &L,
                                         DynCells &dc) {
  const int W = c.W, H = c.H;
  const int pcx = L.pcx[k], pcy = L.pcy[k], pu = L.pu[k], ncx = L.ncx[k], ncy = L.ncy[k], nu = L.nu[k];
  unsigned char pv[9], nv[9];
  const bool small = pu <= 1 && nu <= 1;
  const bool interior = pcx >= 1 && pcx <= W - 2 && pcy >= 1 && pcy <= H - 2 && ncx >= 1 && ncx <= W - 2 && ncy >= 1 &&
                        ncy <= H - 2;
  if (__all(interior || !small)) {
    // every 3 x 3 block of the wave lies inside the grid (agents keep a radius away from the border): one
    // address per block, the nine cells at constant offsets
    if constexpr (GX::tiled) {
      const int pi = interior ? pcx : 1, pj = interior ? pcy : 1, ni = interior ? ncx : 1, nj = interior ? ncy : 1;
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        pv[q] = PREV ? gt[gx(pi + q / 3 - 1, pj + q % 3 - 1)] : (unsigned char)0;
        nv[q] = NEW ? gt[gx(ni + q / 3 - 1, nj + q % 3 - 1)] : (unsigned char)D2D_OCCUPIED;
      }
    } else {
      const unsigned char *pp = gt + (interior ? pcx * H + pcy : H + 1), *np = gt + (interior ? ncx * H + ncy : H + 1);
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const int o = (q / 3 - 1) * H + (q % 3 - 1);
        pv[q] = PREV ? pp[o] : (unsigned char)0;
        nv[q] = NEW ? np[o] : (unsigned char)D2D_OCCUPIED;
      }
    }
  } else {
#pragma unroll
    for (int q = 0; q < 9; ++q) {  // clamped (always valid) addresses
      const int di = q / 3 - 1, dj = q % 3 - 1;
      pv[q] = PREV ? gt[gx(min(max(pcx + di, 0), W - 1), min(max(pcy + dj, 0), H - 1))] : (unsigned char)0;
      nv[q] = NEW ? gt[gx(min(max(ncx + di, 0), W - 1), min(max(ncy + dj, 0), H - 1))] : (unsigned char)D2D_OCCUPIED;
    }
  }
  unsigned int pdyn = 0, nfree = 0;
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    pdyn |= (pv[q] == D2D_DYNAMIC) ? (1u << q) : 0u;
    nfree |= (nv[q] != D2D_OCCUPIED && nv[q] != D2D_DYNAMIC) ? (1u << q) : 0u;
  }
  // previous cells inside this agent's own new block stay DYNAMIC whatever the others do: only the rest
  // (none unless the agent changed cell) needs the coverage bitmap
  unsigned int own = 0;
  {
    const int dx = ncx - pcx, dy = ncy - pcy;  // prev cell (di, dj) is in the new block iff |di - dx| <= nu, |dj - dy| <= nu
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const int di = q / 3 - 1, dj = q % 3 - 1;
      own |= (abs(di - dx) <= nu && abs(dj - dy) <= nu) ? (1u << q) : 0u;
    }
  }
  dc.pclr = small ? (pdyn & block_valid9(pcx, pcy, pu, W, H) & ~own) : 0u;
  dc.nfree = small ? (nfree & block_valid9(ncx, ncy, nu, W, H)) : 0u;
}

// PHASE 0: the one-pass update (coverage from the LDS bitmap).
// PHASE 1 / 2: the two-phase update of grids too large for a bitmap (Geom.dyn2p).  Phase 1: every agent clears the cells of its
// previous block that hold DYNAMIC and lie outside its OWN new block -- whether or not another agent's new block covers them.
// After a fence, phase 2: every agent fetches its new block again and marks what is neither static nor DYNAMIC, which restores
// whatever a neighbour cleared in phase 1.  Same final grid as the reference's clear-everything-then-mark (a cell that is not
// static ends DYNAMIC iff some new block covers it, and a cell that held DYNAMIC and is covered by nobody lies in a previous
// block whose owner clears it), with no coverage structure: no LDS, no inserts, no lookups.
template <int PHASE, typename GX>
__device__ __forceinline__ void dyn_apply(const GX &gx, const d2d_cfg &c, const d2d_state &s, int e, int k, const Geom &g,
                                          const LdsView &L, unsigned char *__restrict__ gt, const DynCells &dc) {
  const int N = c.N, W = c.W, H = c.H;
  int *prev = s.dyn_prev + (size_t)e * N * 3;
  const int pcx = L.pcx[k], pcy = L.pcy[k], pu = L.pu[k], ncx = L.ncx[k], ncy = L.ncy[k], nu = L.nu[k];
  auto covered = [&](int i, int j) {
    if (PHASE == 1) return abs(i - ncx) <= nu && abs(j - ncy) <= nu;  // only the agent's own new block holds a clear back
    const int bit = i * H + j;
    return ((L.bm[bit >> 5] >> (bit & 31)) & 1u) != 0u;
  };
  if (pu <= 1 && nu <= 1) {
    // only set bits cost anything: none for an agent that stayed in its cell, a handful when it moved on
    if (PHASE != 2) {
      unsigned int m = dc.pclr;  // already without the cells of the own new block
      while (m) {
        const int q = __ffs((int)m) - 1;
        m &= m - 1;
        const int i = pcx + q / 3 - 1, j = pcy + q % 3 - 1;
        if (PHASE == 1 || !covered(i, j)) gt[gx(i, j)] = D2D_UNOCCUPIED;
      }
    }
    if (PHASE != 1) {
      unsigned int m = dc.nfree;
      while (m) {
        const int q = __ffs((int)m) - 1;
        m &= m - 1;
        gt[gx(ncx + q / 3 - 1, ncy + q % 3 - 1)] = D2D_DYNAMIC;
      }
    }
  } else {
    if (PHASE != 2) {
      const int i1 = min(pcx + pu + 1, W), j1 = min(pcy + pu + 1, H);
      for (int i = max(pcx - pu, 0); i < i1; ++i)
        for (int j = max(pcy - pu, 0); j < j1; ++j)
          if (gt[gx(i, j)] == D2D_DYNAMIC && !covered(i, j)) gt[gx(i, j)] = D2D_UNOCCUPIED;
    }
    if (PHASE != 1) {
      const int i3 = min(ncx + nu + 1, W), j3 = min(ncy + nu + 1, H);
      for (int i = max(ncx - nu, 0); i < i3; ++i)
        for (int j = max(ncy - nu, 0); j < j3; ++j) {
          const unsigned char v = gt[gx(i, j)];
          if (v != D2D_OCCUPIED && v != D2D_DYNAMIC) gt[gx(i, j)] = D2D_DYNAMIC;
        }
    }
  }
  if (PHASE != 1) {
    if (pcx != ncx) prev[3 * k] = ncx;
    if (pcy != ncy) prev[3 * k + 1] = ncy;
    if (pu != nu) prev[3 * k + 2] = nu;
  }
}


// ---- whole-grid staging (Geom.full: the specialised 50 x 50 geometry) ----
// A grid of one env (W * H bytes, a multiple of 4, dword aligned because every env's offset is) -> LDS by DMA, four bytes
// per lane per instruction (the env's block is not 16-byte aligned in general); no VGPRs, the caller's vmcnt(0) covers it.
__device__ __forceinline__ void grid_stage(const unsigned char *__restrict__ src, unsigned int *dst, int nbytes, int lane) {
  const int nd = nbytes >> 2;
  for (int base = 0; base < nd; base += WAVE) {
    const int idx = base + lane;
    if (idx < nd)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (size_t)idx * 4),
                                       (__attribute__((address_space(3))) void *)((char *)dst + (size_t)base * 4), 4, 0, 0);
  }
}

// utils.py:527-540 on the LDS copy of the ground truth, lane = agent.  Same order-independent rule as dyn_apply (final
// value of a cell = static ? unchanged : covered by some new block ? DYNAMIC : was DYNAMIC ? UNOCCUPIED : unchanged), with the
// coverage kept IN the copy: every agent first ORs a mark (bit 7) into the cells of its new block, then a previous cell
// that holds DYNAMIC without a mark is cleared and a new cell that is neither static nor DYNAMIC is set -- in global memory
// only (the copy is scratch from here on: nothing reads it after this stage).
__device__ __forceinline__ void dyn_full(const d2d_cfg &c, const d2d_state &s, int e, int lane, const LdsView &L,
                                         unsigned char *__restrict__ gt) {
  const int N = c.N, W = c.W, H = c.H;
  unsigned char *g8 = (unsigned char *)L.gtw;
  int *prev = s.dyn_prev + (size_t)e * N * 3;
  bool small = true;
  for (int k = lane; k < N; k += WAVE) small = small && (L.pu[k] <= 1 && L.nu[k] <= 1);
  if (__all(small) && N * 9 <= 3 * WAVE) {
    // few agents, blocks of at most 3 x 3 cells: lane = (agent, cell of its 3 x 3 neighbourhood), N * 9 pairs over the lanes
    const int npair = N * 9;
    const FastDiv fd9(9);
    for (int p0 = 0; p0 < npair; p0 += WAVE) {
      const int pi = min(p0 + lane, npair - 1);
      int k, q;
      fd9.divmod(pi, k, q);
      const int di = ((q * 11) >> 5) - 1, dj = q - 3 * ((q * 11) >> 5) - 1;  // q / 3 - 1, q % 3 - 1 for q < 9
      const int ni = L.ncx[k] + di, nj = L.ncy[k] + dj, nu = L.nu[k];
      const bool nval = (int)(p0 + lane < npair) & (int)(abs(di) <= nu) & (abs(dj) <= nu) & ((unsigned int)ni < (unsigned int)W) &
                        ((unsigned int)nj < (unsigned int)H);
      if (nval) g8[ni * H + nj] = g8[ni * H + nj] | 0x80;
    }
    wave_sync_lds();
    for (int p0 = 0; p0 < npair; p0 += WAVE) {
      const int pi = min(p0 + lane, npair - 1);
      int k, q;
      fd9.divmod(pi, k, q);
      const int di = ((q * 11) >> 5) - 1, dj = q - 3 * ((q * 11) >> 5) - 1;
      const bool on = p0 + lane < npair;
      const int pcx = L.pcx[k], pcy = L.pcy[k], pu = L.pu[k], ncx = L.ncx[k], ncy = L.ncy[k], nu = L.nu[k];
      const int qi = pcx + di, qj = pcy + dj, ni = ncx + di, nj = ncy + dj;
      const bool pval = (int)on & (int)(abs(di) <= pu) & (int)(abs(dj) <= pu) & ((unsigned int)qi < (unsigned int)W) & ((unsigned int)qj < (unsigned int)H);
      const bool nval = (int)on & (int)(abs(di) <= nu) & (int)(abs(dj) <= nu) & ((unsigned int)ni < (unsigned int)W) & ((unsigned int)nj < (unsigned int)H);
      const int pidx = min(max(qi, 0), W - 1) * H + min(max(qj, 0), H - 1), nidx = min(max(ni, 0), W - 1) * H + min(max(nj, 0), H - 1);
      const unsigned char pv = g8[pidx], nv = g8[nidx] & 0x7f;
      if (pval & (pv == D2D_DYNAMIC)) gt[pidx] = D2D_UNOCCUPIED;                      // DYNAMIC and unmarked
      if (nval & (nv != D2D_OCCUPIED) & (nv != D2D_DYNAMIC)) gt[nidx] = D2D_DYNAMIC;
      if (on & (q == 4)) {  // the centre cell's lane keeps the agent's record
        if (pcx != ncx) prev[3 * k] = ncx;
        if (pcy != ncy) prev[3 * k + 1] = ncy;
        if (pu != nu) prev[3 * k + 2] = nu;
      }
    }
    return;
  }
  if (__all(small)) {
    // many agents (the lanes are busy with lane = agent): 3 x 3 blocks, fixed trip counts, all reads of a lane in flight together
    for (int k0 = 0; k0 < N; k0 += WAVE) {
      const int k = k0 + lane, kc = min(k, N - 1);
      const bool on = k < N;
      const int ncx = L.ncx[kc], ncy = L.ncy[kc], nu = L.nu[kc];
      const unsigned int nval = on ? block_valid9(ncx, ncy, nu, W, H) : 0u;
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const int di = q / 3 - 1, dj = q % 3 - 1;
        if ((nval >> q) & 1u) {
          const int idx = (ncx + di) * H + (ncy + dj);
          g8[idx] = g8[idx] | 0x80;
        }
      }
    }
    wave_sync_lds();
    for (int k0 = 0; k0 < N; k0 += WAVE) {
      const int k = k0 + lane, kc = min(k, N - 1);
      const bool on = k < N;
      const int pcx = L.pcx[kc], pcy = L.pcy[kc], pu = L.pu[kc], ncx = L.ncx[kc], ncy = L.ncy[kc], nu = L.nu[kc];
      const unsigned int pval = on ? block_valid9(pcx, pcy, pu, W, H) : 0u, nval = on ? block_valid9(ncx, ncy, nu, W, H) : 0u;
      unsigned char pv[9], nv[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) {  // clamped (always valid) addresses, all reads in flight together
        const int di = q / 3 - 1, dj = q % 3 - 1;
        pv[q] = g8[min(max(pcx + di, 0), W - 1) * H + min(max(pcy + dj, 0), H - 1)];
        nv[q] = g8[min(max(ncx + di, 0), W - 1) * H + min(max(ncy + dj, 0), H - 1)];
      }
      unsigned int pclr = 0, nset = 0;
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        pclr |= (pv[q] == D2D_DYNAMIC) ? (1u << q) : 0u;  // DYNAMIC and unmarked
        const unsigned char o = nv[q] & 0x7f;
        nset |= (o != D2D_OCCUPIED && o != D2D_DYNAMIC) ? (1u << q) : 0u;
      }
      unsigned int m = pclr & pval;
      while (m) {
        const int q = __ffs((int)m) - 1;
        m &= m - 1;
        gt[(pcx + q / 3 - 1) * H + (pcy + q % 3 - 1)] = D2D_UNOCCUPIED;
      }
      m = nset & nval;
      while (m) {
        const int q = __ffs((int)m) - 1;
        m &= m - 1;
        gt[(ncx + q / 3 - 1) * H + (ncy + q % 3 - 1)] = D2D_DYNAMIC;
      }
      if (on) {
        if (pcx != ncx) prev[3 * k] = ncx;
        if (pcy != ncy) prev[3 * k + 1] = ncy;
        if (pu != nu) prev[3 * k + 2] = nu;
      }
    }
    return;
  }
  for (int k = lane; k < N; k += WAVE) {  // any block size: plain loops
    const int ncx = L.ncx[k], ncy = L.ncy[k], nu = L.nu[k];
    const int i1 = min(ncx + nu + 1, W), j1 = min(ncy + nu + 1, H);
    for (int i = max(ncx - nu, 0); i < i1; ++i)
      for (int j = max(ncy - nu, 0); j < j1; ++j) g8[i * H + j] = g8[i * H + j] | 0x80;
  }
  wave_sync_lds();
  for (int k = lane; k < N; k += WAVE) {
    const int pcx = L.pcx[k], pcy = L.pcy[k], pu = L.pu[k], ncx = L.ncx[k], ncy = L.ncy[k], nu = L.nu[k];
    const int i1 = min(pcx + pu + 1, W), j1 = min(pcy + pu + 1, H);
    for (int i = max(pcx - pu, 0); i < i1; ++i)
      for (int j = max(pcy - pu, 0); j < j1; ++j)
        if (g8[i * H + j] == D2D_DYNAMIC) gt[i * H + j] = D2D_UNOCCUPIED;
    const int i3 = min(ncx + nu + 1, W), j3 = min(ncy + nu + 1, H);
    for (int i = max(ncx - nu, 0); i < i3; ++i)
      for (int j = max(ncy - nu, 0); j < j3; ++j) {
        const unsigned char o = g8[i * H + j] & 0x7f;
        if (o != D2D_OCCUPIED && o != D2D_DYNAMIC) gt[i * H + j] = D2D_DYNAMIC;
      }
    if (pcx != ncx) prev[3 * k] = ncx;
    if (pcy != ncy) prev[3 * k + 1] = ncy;
    if (pu != nu) prev[3 * k + 2] = nu;
  }
}

// utils.py:780-784 + envs/drone_v2.py:251-255 from the LDS copy of the whole explored map (which the rays patched): the
// L x L crop around cell (ci, cj), zero outside the map.  Two crop rows per pass (lanes 0-31 / 32-63 = the first 32 columns of
// rows 2t / 2t + 1), further columns by lane = row: indices advance by constants, no division.
__device__ __forceinline__ void obs_full(const d2d_cfg &c, const d2d_state &s, int e, int lane, const LdsView &L, int ci, int cj,
                                         const EnvRegs &r) {
  const int W = c.W, H = c.H, Lm = c.L, edge = (c.L - 1) / 2;
  const unsigned char *d8 = (const unsigned char *)L.dmt;
  unsigned char *__restrict__ ob = s.obs_local + (size_t)e * Lm * Lm;
  const int i0 = ci - edge, j0 = cj - edge;
  const int half = lane >> 5, col = lane & 31;
  const int j = j0 + col;
  const bool jok = col < Lm && j >= 0 && j < H;
  const int jc = min(max(j, 0), H - 1);
  for (int t = 0; 2 * t < Lm; ++t) {
    const int rr = 2 * t + half, i = i0 + rr;
    const unsigned char v = d8[min(max(i, 0), W - 1) * H + jc];
    if (rr < Lm && col < Lm) ob[rr * Lm + col] = (jok && i >= 0 && i < W) ? v : (unsigned char)0;
  }
  for (int c0 = 32; c0 < Lm; ++c0) {  // column c0 of every row, lane = row (Lm <= 64 rows checked by the caller)
    const int i = i0 + lane, jj = j0 + c0;
    const unsigned char v = d8[min(max(i, 0), W - 1) * H + min(max(jj, 0), H - 1)];
    if (lane < Lm) ob[lane * Lm + c0] = (i >= 0 && i < W && jj >= 0 && jj < H) ? v : (unsigned char)0;
  }
  if (lane == 0) s.obs_yaw[e] = (float)r.yaw;
}

// ---- Kalman trackers, utils.py:172-275; lane = tracker slot ----
// F = [[1,0,.1,0],[0,1,0,.1],[0,0,1,0],[0,0,0,1]] and H = [I2 0] are constant, so the dense products
// of the reference collapse: multiplying by an exact 0 or 1 and adding an exact 0 do not round, hence the
// sparse expressions below give the same values as the oracle's dense loops (tests compare bit for bit).
template <bool KF_LDS>
__device__ __forceinline__ void st_tracker(const d2d_cfg &c, const d2d_state &s, int e, int lane, const Geom &g,
                                           const LdsView &L, EnvRegs &r, size_t noise_off) {
  const int N = c.N;
  int arch_n = 0, arch_ts = 0;
  // More than 128 agents: the few trackers that have anything to do (active, or hit this step) are spread over all lane passes,
  // and every pass with one of them pays for the whole filter.  Their indices are gathered first -- into the ray candidates'
  // planes, which nothing reads after the raycast (room for 16 * ccap indices) -- and the filter runs over that list: one pass
  // instead of three at BASELINE config 3's 172 agents.
  short *need_list = (short *)L.cx;
  const bool gather = c.kf_enabled && N > 2 * WAVE && N <= 16 * g.ccap;  // three passes or more (two: the gathering costs what it saves)
  int nwork = N;
  if (gather) {
    nwork = 0;
    for (int k0 = 0; k0 < N; k0 += WAVE) {
      const int k = k0 + lane;
      const bool need = k < N && (L.act[k] != 0 || L.hit[k] != 0);
      const unsigned long long m = __ballot(need);
      if (need) need_list[nwork + __popcll(m & ((1ull << lane) - 1ull))] = (short)k;
      nwork += __popcll(m);
    }
    wave_sync_lds();
  }
  for (int q0 = 0; q0 < nwork; q0 += WAVE) {
    const int q = q0 + lane;
    const int k = gather ? (q < nwork ? (int)need_list[q] : N) : q;
    if (k < N) {
      const bool has_z = L.hit[k] != 0;
      unsigned char act = L.act[k];
      if (!c.kf_enabled) {
        if (has_z && !act) s.active[(size_t)e * N + k] = 1;
      } else if (act || has_z) {
        double *__restrict__ gk = s.kf + ((size_t)e * N + k) * D2D_KF;
        int len = 1;
        double zx = L.ax[k], zy = L.ay[k];
        if (s.noise) {
          zx = zx + c.sigma * s.noise[noise_off + ((size_t)e * N + k) * 2];
          zy = zy + c.sigma * s.noise[noise_off + ((size_t)e * N + k) * 2 + 1];
        }
        double m0, m1, m2, m3;
        double S[16];
        if (act) {
          len = L.klen[k];
          if (KF_LDS) {
            const double *lk = L.kf + k * D2D_KF;
            m0 = lk[0]; m1 = lk[1]; m2 = lk[2]; m3 = lk[3];
#pragma unroll
            for (int i = 0; i < 16; ++i) S[i] = lk[4 + i];
          } else {
            m0 = gk[0]; m1 = gk[1]; m2 = gk[2]; m3 = gk[3];
#pragma unroll
            for (int i = 0; i < 16; ++i) S[i] = gk[4 + i];
          }
          // predict(), utils.py:225-240, in place (each line only reads entries not yet overwritten):
          // mu <- F mu ; S <- F S ; S <- S F^T ; S += Q
          const double qn = (c.sigma != 0.0) ? 0.1 : 0.001;
          m0 = m0 + 0.1 * m2;
          m1 = m1 + 0.1 * m3;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            S[j] = S[j] + 0.1 * S[8 + j];
            S[4 + j] = S[4 + j] + 0.1 * S[12 + j];
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            S[4 * i + 0] = S[4 * i + 0] + S[4 * i + 2] * 0.1;
            S[4 * i + 1] = S[4 * i + 1] + S[4 * i + 3] * 0.1;
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
            const double idet = 1.0 / det;  // inv(S) through one reciprocal (same in oracle/d2d_oracle.c)
            const double i00 = d * idet, i01 = -b * idet, i10 = -cc * idet, i11 = a * idet;
            double K0[4], K1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              K0[i] = S[4 * i] * i00 + S[4 * i + 1] * i10;
              K1[i] = S[4 * i] * i01 + S[4 * i + 1] * i11;
            }
            const double rx = zx - m0, ry = zy - m1;
            const double e00 = 1.0 - K0[0], e01 = 0.0 - K1[0], e10 = 0.0 - K0[1], e11 = 1.0 - K1[1];
#pragma unroll
            for (int j = 0; j < 4; ++j) {  // S <- (I - K H) S, rows 2,3 first (they read rows 0,1), then 0,1
              const double s0 = S[j], s1 = S[4 + j];
              S[8 + j] = ((0.0 - K0[2]) * s0 + (0.0 - K1[2]) * s1) + S[8 + j];
              S[12 + j] = ((0.0 - K0[3]) * s0 + (0.0 - K1[3]) * s1) + S[12 + j];
              S[j] = e00 * s0 + e01 * s1;
              S[4 + j] = e10 * s0 + e11 * s1;
            }
            m0 = m0 + (K0[0] * rx + K1[0] * ry);
            m1 = m1 + (K0[1] * rx + K1[1] * ry);
            m2 = m2 + (K0[2] * rx + K1[2] * ry);
            m3 = m3 + (K0[3] * rx + K1[3] * ry);
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
        L.act[k] = act;
        L.klen[k] = len;
      }
    }
  }
  if (c.kf_enabled) {
    r.bufn += wave_sum(arch_n);
    r.bufts += wave_sum(arch_ts);
  }
}

// The same stage with lane = ONE ELEMENT of one tracker's state (mu[4], Sigma[16]: 20 lanes per tracker, three trackers per pass):
// the kernel for few agents (SPEC 1), where one to three trackers have anything to do in a step.  With a lane per tracker the
// stage keeps a whole filter in registers -- 16 + 8 + 8 doubles: the register peak of the kernel (111 VGPRs with it, 82 without)
// -- and issues a filter's ~170 instructions for one active lane.  Here every lane forms its own element, in two steps through
// LDS: the predicted element (utils.py:225-240), then -- after the archive test, which every lane of the tracker evaluates on the
// same predicted values -- the updated one (:249-260).  Each element is the expression the sequential code evaluates for it, operand
// for operand (F and H are 0 / 1 / 0.1: the dense products collapse without changing a rounding), so the state stays bit-identical.
__device__ __forceinline__ void st_tracker_elem(const d2d_cfg &c, const d2d_state &s, int e, int lane, const Geom &g,
                                                const LdsView &L, EnvRegs &r, size_t noise_off) {
  const int N = c.N;
  if (!c.kf_enabled) {
    for (int k = lane; k < N; k += WAVE)
      if (L.hit[k] != 0 && L.act[k] == 0) s.active[(size_t)e * N + k] = 1;
    return;
  }
  // the trackers with anything to do (active, or hit this step), in index order
  short *list = (short *)L.cidx;   // [<= 2 * ccap] (nothing reads the candidate planes after the raycast)
  double *pbuf = L.cx;             // [3][20] predicted elements of the three trackers of a pass (cx .. crr: 32 * ccap bytes >= 512)
  int nwork = 0;
  for (int k0 = 0; k0 < N; k0 += WAVE) {
    const int k = k0 + lane;
    const bool need = k < N && (L.act[k] != 0 || L.hit[k] != 0);
    const unsigned long long m = __ballot(need);
    if (need) list[nwork + __popcll(m & ((1ull << lane) - 1ull))] = (short)k;
    nwork += __popcll(m);
  }
  if (nwork == 0) return;
  wave_sync_lds();
  // lanes 60..63 are idle: they mirror element 0 of the third tracker (an element index of 20..23 would read past the tracker's
  // record -- past the END of the kf buffer for the last tracker of the last env: a fault when that buffer ends on a page boundary)
  const int slot = lane >= 40 ? 2 : (lane >= 20 ? 1 : 0), idx = lane < 60 ? lane - 20 * slot : 0;
  const bool is_mu = idx < 4;
  const int ei = is_mu ? idx : (idx - 4) >> 2, ej = is_mu ? 0 : (idx - 4) & 3;       // mu[ei] or Sigma[ei][ej]
  const double qn = (c.sigma != 0.0) ? 0.1 : 0.001;
  const double init_el = is_mu ? 0.0 : (ei != ej ? 0.0 : (ei < 2 ? 1.0 : 10.0));    // KalmanFilter.__init__: mu 0, Sigma diag(1, 1, 10, 10)
  int arch_n = 0, arch_ts = 0;
  for (int q0 = 0; q0 < nwork; q0 += 3) {
    const int q = q0 + slot;
    const bool on = lane < 60 && q < nwork;
    const int k = on ? (int)list[q] : (int)list[q0];
    const bool has_z = L.hit[k] != 0, act = L.act[k] != 0;
    double *__restrict__ gk = s.kf + ((size_t)e * N + k) * D2D_KF;
    auto old = [&](int el) -> double { return gk[el]; };   // the tracker's record in global memory (four loads per lane in flight)
    // ---- predict (only meaningful for an active tracker; computed by all, selected below) ----
    double pe;
    if (is_mu) {
      const double m_i = old(ei), m_v = old(ei < 2 ? ei + 2 : ei);
      pe = ei < 2 ? m_i + 0.1 * m_v : m_i;
    } else {
      const int lo_i = ei < 2 ? ei + 2 : ei, hi_j = ej < 2 ? ej + 2 : ej;
      const double s_ij = old(4 + 4 * ei + ej), s_lj = old(4 + 4 * lo_i + ej), s_ih = old(4 + 4 * ei + hi_j), s_lh = old(4 + 4 * lo_i + hi_j);
      const double r_ij = ei < 2 ? s_ij + 0.1 * s_lj : s_ij;   // Sigma <- F Sigma: rows 0, 1 += 0.1 * rows 2, 3
      const double r_ih = ei < 2 ? s_ih + 0.1 * s_lh : s_ih;
      pe = ej < 2 ? r_ij + r_ih * 0.1 : r_ij;                  // Sigma <- Sigma F^T: columns 0, 1 += columns 2, 3 * 0.1
      if (ei == ej) pe += qn;                                  // + Q
    }
    wave_sync_lds();  // (the previous pass has read its pbuf)
    if (lane < 60) pbuf[lane] = pe;
    wave_sync_lds();
    const double *pb = pbuf + 20 * slot;
    const double pm0 = pb[0], pm1 = pb[1];
    const bool archive = act && (pb[4] >= 150.0 || !(c.kf_lo_x < pm0 && pm0 < c.kf_hi_x) || !(c.kf_lo_y < pm1 && pm1 < c.kf_hi_y));
    const bool reset = archive || !act;   // the state the update (if any) starts from is a fresh filter's
    auto cur = [&](int el, double fresh) -> double { const double v = pb[el]; return reset ? fresh : v; };
    // the tracker's own len / flags, by the lane of its element 0
    const int klen = L.klen[k];
    int len = act ? klen + 1 : 1;
    if (archive) len = 1;
    double zx = L.ax[k], zy = L.ay[k];
    if (s.noise) {
      zx = zx + c.sigma * s.noise[noise_off + ((size_t)e * N + k) * 2];
      zy = zy + c.sigma * s.noise[noise_off + ((size_t)e * N + k) * 2 + 1];
    }
    double out;
    if (!act) {  // first sighting, utils.py:263-273 (has_z holds: the tracker is on the list)
      out = is_mu ? (idx == 0 ? zx : (idx == 1 ? zy : 0.0)) : init_el;
    } else if (!has_z) {
      out = reset ? init_el : pe;
    } else {  // update, utils.py:249-260 (also runs on the freshly reset filter)
      const double c00 = cur(4, 1.0), c01 = cur(5, 0.0), c10 = cur(8, 0.0), c11 = cur(9, 1.0);
      const double a = c.sigma + c00, b = c01, cc = c10, d = c.sigma + c11;
      const double det = a * d - b * cc;
      const double idet = 1.0 / det;  // inv(S) through one reciprocal (same in oracle/d2d_oracle.c)
      const double i00 = d * idet, i01 = -b * idet, i10 = -cc * idet, i11 = a * idet;
      // the gain's row of this element: K[ei] = Sigma[ei][0..1] inv(S)
      const double ci0 = cur(4 + 4 * ei, ei == 0 ? 1.0 : 0.0), ci1 = cur(4 + 4 * ei + 1, ei == 1 ? 1.0 : 0.0);
      const double K0 = ci0 * i00 + ci1 * i10, K1 = ci0 * i01 + ci1 * i11;
      if (is_mu) {
        const double cm0 = cur(0, 0.0), cm1 = cur(1, 0.0), cmi = cur(ei, 0.0);
        const double rx = zx - cm0, ry = zy - cm1;
        out = cmi + (K0 * rx + K1 * ry);
      } else {
        const double s0 = cur(4 + ej, ej == 0 ? 1.0 : 0.0), s1 = cur(8 + ej, ej == 1 ? 1.0 : 0.0);
        const double cij = cur(idx, init_el);
        const double ka = ei == 0 ? 1.0 - K0 : 0.0 - K0, kb = ei == 1 ? 1.0 - K1 : 0.0 - K1;   // (I - K H)[ei][0..1]
        const double t2 = ka * s0 + kb * s1;
        out = ei < 2 ? t2 : t2 + cij;
      }
    }
    if (on) {
      gk[idx] = out;
      if (idx == 0) {
        const unsigned char nact = archive ? 0 : 1;   // predict() re-initialised the filter (utils.py:238): inactive, even if update() then corrects it
        if (archive) {
          arch_n += 1;
          arch_ts += klen + 1;
        }
        s.kf_len[(size_t)e * N + k] = len;
        s.active[(size_t)e * N + k] = nact;
        L.act[k] = nact;
        L.klen[k] = len;
      }
    }
  }
  r.bufn += wave_sum(arch_n);
  r.bufts += wave_sum(arch_ts);
}

// Value of lane Q (0..3) of the lane's quad: a DPP quad_perm broadcast, no LDS round trip.  Every lane of the quad must be
// executing (the callers keep the whole wave in step and predicate their stores instead).
template <int Q>
__device__ __forceinline__ double quad_bcast_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, Q * 0x55, 0xf, 0xf, false);  // quad_perm:[Q,Q,Q,Q]
  hi = __builtin_amdgcn_update_dpp(hi, hi, Q * 0x55, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// The same stage with FOUR LANES PER TRACKER, lane = one COLUMN j of the tracker's state (mu[j] and Sigma[0..3][j]), 16 trackers
// per pass over the list of those that have anything to do (active, or hit this step): the form for more than 16 agents.
// With a lane per tracker the stage holds a whole filter in registers -- 16 + 8 + 8 doubles, the register peak of every kernel
// and phase that contains it (121 VGPRs against 82 without it; in the persistent loop the gaze + perceive phase saved and
// restored 40 callee-saved registers per call for it: 11 KB of scratch writes per env-step) -- and issues the filter's ~170
// instructions for the few lanes whose tracker is active.  The filter is column-local except for three hand-offs inside the quad
// (DPP quad_perm broadcasts, no LDS): predict() adds column j + 2 to column j < 2 (the lane loads both), the archive test reads
// Sigma[0][0], mu[0], mu[1], and update() needs columns 0 and 1 for the gain.  A lane then holds its own column (4 doubles), the
// two broadcast columns (8) and one row of the gain at a time.  Every element is the expression the sequential code evaluates for
// it, operand for operand (F and H are 0 / 1 / 0.1: the dense products collapse without changing a rounding): bit-identical state.
template <bool KF_LDS>
__device__ __forceinline__ void st_tracker_quad(const d2d_cfg &c, const d2d_state &s, int e, int lane, const Geom &g,
                                                const LdsView &L, EnvRegs &r, size_t noise_off) {
  const int N = c.N;
  if (!c.kf_enabled) {
    for (int k = lane; k < N; k += WAVE)
      if (L.hit[k] != 0 && L.act[k] == 0) s.active[(size_t)e * N + k] = 1;
    return;
  }
  // the trackers with anything to do, in index order -- into the ray candidates' planes, which nothing reads after the raycast
  // (cx .. crr are contiguous: room for 16 * ccap >= N indices)
  short *list = (short *)L.cx;
  int nwork = 0;
  for (int k0 = 0; k0 < N; k0 += WAVE) {
    const int k = k0 + lane;
    const bool need = k < N && (L.act[k] != 0 || L.hit[k] != 0);
    const unsigned long long m = __ballot(need);
    if (need) list[nwork + __popcll(m & ((1ull << lane) - 1ull))] = (short)k;
    nwork += __popcll(m);
  }
  if (nwork == 0) return;
  wave_sync_lds();
  const int slot = lane >> 2, j = lane & 3, jp = j < 2 ? j + 2 : j;   // the column predict() adds to column j < 2
  const double qn = (c.sigma != 0.0) ? 0.1 : 0.001;
  int arch_n = 0, arch_ts = 0;
  for (int q0 = 0; q0 < nwork; q0 += WAVE / 4) {
    const bool on = q0 + slot < nwork;
    const int k = (int)list[on ? q0 + slot : q0];
    const bool has_z = L.hit[k] != 0, act = L.act[k] != 0;
    double *__restrict__ gk = s.kf + ((size_t)e * N + k) * D2D_KF;
    // ten loads per lane in flight: mu[j], mu[jp], columns j and jp (two instantiations: a run-time choice between an LDS and a
    // global pointer would become flat loads)
    auto old = [&](int el) -> double {
      if constexpr (KF_LDS) return L.kf[k * D2D_KF + el];
      else return gk[el];
    };
    const double mj = old(j), mp = old(jp);
    double A[4], Bc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      A[i] = old(4 + 4 * i + j);
      Bc[i] = old(4 + 4 * i + jp);
    }
    const int klen = L.klen[k];
    double zx = L.ax[k], zy = L.ay[k];
    if (s.noise) {
      zx = zx + c.sigma * s.noise[noise_off + ((size_t)e * N + k) * 2];
      zy = zy + c.sigma * s.noise[noise_off + ((size_t)e * N + k) * 2 + 1];
    }
    // ---- predict(), utils.py:225-240: mu <- F mu ; S <- F S ; S <- S F^T ; S += Q (computed by all, selected below) ----
    double m = j < 2 ? mj + 0.1 * mp : mj;
    A[0] = A[0] + 0.1 * A[2];     // F S on the own column ...
    A[1] = A[1] + 0.1 * A[3];
    Bc[0] = Bc[0] + 0.1 * Bc[2];  // ... and on the column S F^T adds to it
    Bc[1] = Bc[1] + 0.1 * Bc[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const double t = A[i] + Bc[i] * 0.1;
      A[i] = j < 2 ? t : A[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const double t = A[i] + qn;
      A[i] = i == j ? t : A[i];
    }
    int len = klen + 1;
    const double p00 = quad_bcast_f64<0>(A[0]), pm0 = quad_bcast_f64<0>(m), pm1 = quad_bcast_f64<1>(m);
    const bool archive = act && (p00 >= 150.0 || !(c.kf_lo_x < pm0 && pm0 < c.kf_hi_x) || !(c.kf_lo_y < pm1 && pm1 < c.kf_hi_y));
    if (archive && on && j == 0) {
      arch_n += 1;  // archived copy -> tracker_buffer
      arch_ts += len;
    }
    // the state update() (if any) starts from: the prediction, or a fresh filter's (KalmanFilter.__init__: mu 0, Sigma diag(1, 1, 10, 10))
    // after an archive / before the first sighting
    const bool fresh = archive || !act;
    if (fresh) len = 1;
    m = fresh ? 0.0 : m;
#pragma unroll
    for (int i = 0; i < 4; ++i) A[i] = fresh ? (i == j ? (i < 2 ? 1.0 : 10.0) : 0.0) : A[i];
    double om = m, O[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) O[i] = A[i];
    {  // update(), utils.py:249-260 (also runs on the freshly reset filter); computed by all, selected below
      double C0[4], C1[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        C0[i] = quad_bcast_f64<0>(A[i]);
        C1[i] = quad_bcast_f64<1>(A[i]);
      }
      const double m0 = quad_bcast_f64<0>(m), m1 = quad_bcast_f64<1>(m);
      const double a = c.sigma + C0[0], b = C1[0], cc = C0[1], d = c.sigma + C1[1];
      const double det = a * d - b * cc;
      const double idet = 1.0 / det;  // inv(S) through one reciprocal (same in oracle/d2d_oracle.c)
      const double i00 = d * idet, i01 = -b * idet, i10 = -cc * idet, i11 = a * idet;
      const double rx = zx - m0, ry = zy - m1;
      const double s0 = A[0], s1 = A[1];
      double Kj0 = 0.0, Kj1 = 0.0;  // the gain's row j, for mu[j]
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const double K0 = C0[i] * i00 + C1[i] * i10, K1 = C0[i] * i01 + C1[i] * i11;
        Kj0 = i == j ? K0 : Kj0;
        Kj1 = i == j ? K1 : Kj1;
        // S <- (I - K H) S on the own column: rows 2, 3 add to themselves, rows 0, 1 are replaced
        if (i >= 2) O[i] = ((0.0 - K0) * s0 + (0.0 - K1) * s1) + A[i];
        else if (i == 0) O[i] = (1.0 - K0) * s0 + (0.0 - K1) * s1;
        else O[i] = (0.0 - K0) * s0 + (1.0 - K1) * s1;
      }
      om = m + (Kj0 * rx + Kj1 * ry);
    }
    if (!act) {  // first sighting, utils.py:263-273 (has_z holds: the tracker is on the list)
      om = j == 0 ? zx : (j == 1 ? zy : 0.0);
#pragma unroll
      for (int i = 0; i < 4; ++i) O[i] = A[i];
    } else if (!has_z) {
      om = m;
#pragma unroll
      for (int i = 0; i < 4; ++i) O[i] = A[i];
    }
    if (on) {
      gk[j] = om;
#pragma unroll
      for (int i = 0; i < 4; ++i) gk[4 + 4 * i + j] = O[i];
      if (j == 0) {
        const unsigned char nact = (act && archive) ? 0 : 1;  // predict() re-initialised the filter (utils.py:238): inactive, even if update() then corrects it
        s.kf_len[(size_t)e * N + k] = len;
        s.active[(size_t)e * N + k] = nact;
        L.act[k] = nact;
        L.klen[k] = len;
      }
    }
  }
  r.bufn += wave_sum(arch_n);
  r.bufts += wave_sum(arch_ts);
}

// utils.py:764-778 + envs/drone_v2.py:217-235.  `probe_wall`: the lane's batch-2 static probe (lane < 5).
__device__ __forceinline__ void st_collide(const d2d_cfg &c, const d2d_state &s, int e, int lane, const LdsView &L,
                                           bool probe_wall, EnvRegs &r) {
  const int N = c.N;
  const double R = c.drone_radius;
  int col = __any(probe_wall) ? 1 : 0;
  if (!col) {
    bool dyn = false;
    for (int k = lane; k < N; k += WAVE) {
      // norm(d) < r + R (utils.py:774): decided by d.d against (r + R)^2 unless they agree to 1e-14, where the
      // correctly rounded sqrt of the reference decides
      const double dx = L.ax[k] - r.x, dy = L.ay[k] - r.y, t = L.ar[k] + R;
      const double d2 = __builtin_fma(dy, dy, dx * dx), t2 = t * t;  // numpy norm's dot product
      bool hit = d2 < t2 && t > 0.0;
      if (fabs(d2 - t2) <= 1e-14 * t2) hit = sqrt(d2) < t;
      dyn = dyn || hit;
    }
    if (__any(dyn)) col = 2;
  }
  int dead = 0, frz = 0;
  if (col == 0) {
    const double gx = r.x - r.tx, gy = r.y - r.ty;
    // norm(d) <= 10 (drone_v2.py:223) with numpy's norm = sqrt(fma(dy, dy, dx * dx)): sqrt is correctly rounded and monotone,
    // and sqrt(s) <= 10 exactly for s <= 0x1.9000000000001p+6 (100 + 1 ulp; checked on the host), so no square root is taken.
    // norm(v) == 0 (:224) <=> the dot product itself is 0 (sqrt(s) == 0 only for s == 0).
    if (__builtin_fma(gy, gy, gx * gx) <= 0x1.9000000000001p+6) r.sm = D2D_SM_GOAL_REACHED;
    dead = (r.fail >= 10 && __builtin_fma(r.vy, r.vy, r.vx * r.vx) == 0.0) ? 1 : 0;
    frz = ((double)r.steps >= c.max_steps && !dead) ? 1 : 0;
  }
  const int done = (col != 0) || dead || frz || (r.sm == D2D_SM_GOAL_REACHED && r.tnext >= r.ntgt);
  if (done && c.kf_enabled) {  // drone_v2.py:232-235
    int an = 0, ats = 0;
    for (int k = lane; k < N; k += WAVE) {
      if (L.act[k] != 0) {
        an += 1;
        ats += L.klen[k];
      }
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
  r.done = done;
}

// utils.py:780-784 + envs/drone_v2.py:251-255: the crop is the LDS tile (loaded before the rays ran, zero
// outside the map, patched by the rays), copied out as it is.
__device__ __forceinline__ void st_obs(const d2d_cfg &c, const d2d_state &s, int e, int lane, const LdsView &L,
                                       const EnvRegs &r) {
  const int n = c.L * c.L;
  unsigned char *__restrict__ ob = s.obs_local + (size_t)e * n;
  const unsigned char *dmt = (const unsigned char *)L.dmt;
  for (int idx = lane; idx < n; idx += WAVE) ob[idx] = dmt[idx];
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

// One env-step (or any subset of its stages) by one wave.  Stage RESULTS are those of running the stages
// in reference order; the ORDER OF EXECUTION differs where that is free:
//  * the control stage consumes only inputs (plan, action) and the previous pose, so it runs first and the
//    raycast keeps using the pose from before it (x0, y0, yaw0);
//  * all loads addressed by batch-1 data (window, crop, grid cells, probes) are issued together, and the
//    per-ray tan / candidate work runs while they are in flight.
// FULL (Geom.full, the specialised 50 x 50 geometry): both grids are staged WHOLE in LDS by DMA in batch 1 -- there is no
// batch 2 at all: rays, collision probes, the dynamic-grid update and the observation crop read the copies.
// WIDE: 64 ray candidates on the mask path (spec_wide).  CONE: candidates culled against the cone of the rays (spec_cone).
template <bool FULL, bool WIDE, bool CONE, bool TILED = false>
__device__ __forceinline__ void run_env(const d2d_cfg &c, const d2d_state &s, int e, int lane, uint32_t stages,
                                        const Geom &g, const LdsView &L, double action, EnvRegs &r, size_t noise_off = 0,
                                        bool load_r = false, const StepIn *given = nullptr) {
  using RayMask = std::conditional_t<WIDE, unsigned long long, unsigned int>;
  const int N = c.N, W = c.W, H = c.H;
  const double inv_scale = 1.0 / c.scale;
  const bool do_ray = stages & D2D_ST_RAYCAST, do_dyn = stages & D2D_ST_DYNGRID, do_trk = stages & D2D_ST_TRACKER;
  const bool do_col = stages & D2D_ST_COLLIDE, do_obs = stages & D2D_ST_OBS, do_ctl = stages & D2D_ST_CONTROL;
  const GridIx<TILED> gx = {H, (H + 15) >> 4};
  const size_t gbytes = TILED ? (size_t)((W + 15) >> 4) * (size_t)((H + 15) >> 4) * 256 : (size_t)W * H;
  unsigned char *__restrict__ gt = s.gt + (size_t)e * gbytes;
  unsigned char *__restrict__ dm = s.dmap + (size_t)e * gbytes;

  // ---------------- batch 1 ----------------
  // the five collision probes alone read global memory directly (staging the grid for them too -- one round trip fewer in the act
  // phase -- measured: no gain, 2.5 KB more traffic per env-step)
  const bool gt_staged = FULL && (do_ray || do_dyn);
  const uint32_t needs_agents = D2D_ST_AGENTS | D2D_ST_RAYCAST | D2D_ST_DYNGRID | D2D_ST_TRACKER | D2D_ST_COLLIDE;
  // a launch / phase whose agents are only read (collision test, trackers): positions, radii and tracker flags are all it stages
  const bool agents_light = (stages & needs_agents) && !(stages & (D2D_ST_AGENTS | D2D_ST_RAYCAST | D2D_ST_DYNGRID));
  StepInRaw in_raw;
  AgentIn ag0;
  const bool agents_any = (stages & needs_agents) != 0;
  // `load_r` (the act phase of the persistent loop; the stage mask is a constant there): EVERYTHING the phase reads that does not
  // hang on another load is requested before anything looks at a loaded value -- the planner's result (unless the caller hands it
  // over in registers: `given`), the first 64 agents (written to LDS further down), the grid copies, the tracker block, and last
  // the env's registers (they come back through lane reads that wait where the loads stand).  Left alone, the scheduler pulls the
  // control stage's arithmetic up between the groups of loads and each group waits for the one before it: a round trip apiece,
  // five in the act phase.  The perceive phase and a launch of k_stages keep the plain order: the first holds too many registers
  // across the grid copies this way (13 instead of 6 callee-saved registers saved per call), the second has a run-time stage mask
  // (unconditional loads and pointer selects cost it 3 %).
  if (load_r) {
    in_raw.use = false;
    if (do_ctl && !given) load_inputs_raw(c, s, e, true, in_raw);  // (the stage mask is a constant here: no branch at run time)
    agent_load(c, s, e, lane, do_trk || do_col, ag0, agents_light);  // (every lane: no branch -- see agent_load)
  }
  if constexpr (FULL) {
    if (gt_staged) grid_stage(gt, L.gtw, W * H, lane);
    if (do_obs) grid_stage(dm, L.dmt, W * H, lane);
  }
  if (load_r) {
    if (do_trk && c.kf_enabled && g.kf_lds) kf_stage(c, s, e, lane, L);
    load_regs(s, e, r);
    __builtin_amdgcn_sched_barrier(0);
  }
  StepIn in;
  if (given) {  // the planner's result, handed over in registers (ph_plan_act)
    in = *given;
    in.action = action;
  } else if (load_r) {
    finish_inputs(in_raw, action, in);
  } else {
    load_inputs(c, s, e, action, do_ctl, in);
  }
  D2D_STAMP(1);
  if (stages & D2D_ST_FSM) st_fsm(c, s, e, r);
  const double x0 = r.x, y0 = r.y, yaw0 = r.yaw;
  if (do_ctl) st_control(c, in, r);
  D2D_STAMP(2);
  if (!load_r && do_trk && c.kf_enabled && g.kf_lds) kf_stage(c, s, e, lane, L);
  if (agents_any) {
    const bool move = (stages & D2D_ST_AGENTS) != 0;
    if (load_r) {
      if (lane < N) agent_apply(c, s, e, lane, L, inv_scale, move, agents_light, ag0);
      if (N > WAVE) st_agents(c, s, e, lane, g, L, inv_scale, move, do_trk || do_col, agents_light, WAVE);
    } else {
      st_agents_plain(c, s, e, lane, g, L, inv_scale, move, do_trk || do_col, agents_light);
    }
    if (do_trk && !do_ray)  // hit mask of an earlier launch: stage it where the raycast leaves it
      for (int k = lane; k < N; k += WAVE) L.hit[k] = s.hit[(size_t)e * N + k];
    wave_sync_lds();
  }
  D2D_STAMP(3);

  // ---------------- batch 2 (addresses from batch-1 data; tile path only) ----------------
  const int ocx = cell_fast(x0, c.scale, inv_scale), ocy = cell_fast(y0, c.scale, inv_scale);
  const int ncx_d = cell_fast(r.x, c.scale, inv_scale), ncy_d = cell_fast(r.y, c.scale, inv_scale);  // the drone's cell after control
  const int edge = (c.L - 1) / 2;
  const Tile wt = make_tile(ocx - g.reach, ocy - g.reach, g.ws, g.ws);
  const Tile ct = make_tile(ncx_d - edge, ncy_d - edge, c.L, c.L);
  bool probe_wall = false;
  const bool dyn_fast = do_dyn;  // lane = agent, 64 agents per pass: the first pass's cells are fetched early (below), behind the raycast
  DynCells dc;
  dc.pclr = dc.nfree = 0;
  if (do_col && lane < 5 && !gt_staged) {  // utils.py:766-771: static cells never change, so the probes can be read now
    const double R = c.drone_radius;
    const double ox = (lane == 0) ? -R : (lane == 2 ? R : 0.0);
    const double oy = (lane == 3) ? -R : (lane == 4 ? R : 0.0);
    const double qx = r.x + ox, qy = r.y + oy;
    const bool oob = (qx >= c.W_px || qx < 0.0 || qy >= c.H_px || qy < 0.0);
    const int pi = min(max(cell_fast(qx, c.scale, inv_scale), 0), W - 1), pj = min(max(cell_fast(qy, c.scale, inv_scale), 0), H - 1);
    probe_wall = oob || gt[gx(pi, pj)] == D2D_OCCUPIED;
  }
  if constexpr (!FULL) {
    if (do_ray && do_obs && wt.cols <= 32 && wt.rows <= 24 && ct.cols <= 33 && ct.rows <= 34) {
      // default geometry (23 x 23 window, 33 x 33 crop): lanes map to (row parity, column), so a cell costs an
      // add and a compare instead of a division; all loads of both tiles are in flight before the first LDS write
      tile_rows2(gx, wt, (unsigned char *)L.gtw, gt, W, H, lane, (unsigned char)D2D_OCCUPIED,
                 ct, (unsigned char *)L.dmt, dm, (unsigned char)0);
    } else {
      if (do_ray) tile_load<9>(gx, wt, (unsigned char *)L.gtw, gt, W, H, lane, (unsigned char)D2D_OCCUPIED);
      if (do_obs) tile_load<9>(gx, ct, (unsigned char *)L.dmt, dm, W, H, lane, (unsigned char)0);
    }
    if (do_dyn) dyn_bitmap(c, lane, g, L);
    if (dyn_fast && lane < N) {
      if (g.dyn2p) dyn_load<true, false>(gx, c, gt, lane, L, dc);
      else dyn_load(gx, c, gt, lane, L, dc);
    }
  }

  // ---------------- raycast: setup while the loads are in flight ----------------
  int ncand = 0;
  if (do_ray) {
    ncand = ray_cull<CONE>(c, lane, L, x0, y0, yaw0, g.ccap);
#ifdef D2D_ABL_NOCAND
    ncand = 0;
#endif
  }
  wave_sync_lds();
  D2D_STAMP(4);
  // FULL: everything that reads the staged copies waits here once (the tracker DMA is covered too)
  auto landed = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wave_sync_lds();
    if constexpr (FULL) {
      if (do_col && lane < 5 && gt_staged) {  // utils.py:766-771 from the ground-truth copy, BEFORE the dynamic-grid stage marks it
        const double R = c.drone_radius;
        const double ox = (lane == 0) ? -R : (lane == 2 ? R : 0.0);
        const double oy = (lane == 3) ? -R : (lane == 4 ? R : 0.0);
        const double qx = r.x + ox, qy = r.y + oy;
        const bool oob = (qx >= c.W_px || qx < 0.0 || qy >= c.H_px || qy < 0.0);
        const int pi = min(max(cell_fast(qx, c.scale, inv_scale), 0), W - 1), pj = min(max(cell_fast(qy, c.scale, inv_scale), 0), H - 1);
        probe_wall = oob || ((const unsigned char *)L.gtw)[pi * H + pj] == D2D_OCCUPIED;
      }
    }
  };
  int newly = 0;
  if (do_ray) {
    bool go = false;  // do the rays continue past sample 0?
    for (int i0 = 0; i0 < c.R; i0 += WAVE) {
      const int i = i0 + lane;
      const RayT<RayMask> ry = ray_setup<RayMask>(c, L, i, ncand, x0, y0, yaw0);
      if (i0 == 0) {  // the tracker DMA and the tiles / grids have to be in LDS before the first sample reads / patches them
        landed();
        D2D_STAMP(5);
        // Sample 0 of every ray is the drone's own position (utils.py:641-642), so its outcome is shared: an
        // agent covering it stops every ray before the map is touched (:658-664), else its cell is recorded.
        go = (0.0 < x0 && x0 < c.W_px && 0.0 < y0 && y0 < c.H_px);
        bool cover = false;
        if (ncand <= g.ccap) {
          for (int q = lane; q < ncand; q += WAVE) {
            const double dx = L.cx[q] - x0, dy = L.cy[q] - y0;
            if (go && dx * dx + dy * dy <= L.cr2[q]) {
              L.hit[L.cidx[q]] = 1;
              cover = true;
            }
          }
        } else {  // a crowd (Geom.ccap): the list is incomplete, every agent is asked
          for (int k = lane; k < N; k += WAVE) {
            const double dx = L.ax[k] - x0, dy = L.ay[k] - y0;
            if (go && dx * dx + dy * dy <= L.ar2[k]) {
              L.hit[k] = 1;
              cover = true;
            }
          }
        }
        go = go && !__any(cover);
        if (go) {  // dist == 0 < depth^2: only a wall stops the rays here
          const int g0 = FULL ? ocx * H + ocy : wt.byte_index(g.reach, g.reach);  // `go`: the drone is inside the map
          const unsigned char w0 = ((const unsigned char *)L.gtw)[g0];
          const unsigned char v0 = (w0 == D2D_OCCUPIED) ? (unsigned char)D2D_OCCUPIED : (unsigned char)D2D_UNOCCUPIED;
          if (lane == 0) {
#ifndef D2D_ABL_NOSTORE
            dm[gx(ocx, ocy)] = v0;
#endif
            if constexpr (FULL) {
              if (do_obs) ((unsigned char *)L.dmt)[g0] = v0;
            } else {
              const unsigned int pr = (unsigned int)(ocx - ct.i0), pq = (unsigned int)(ocy - ct.j0);
              if (do_obs && pr < (unsigned int)ct.rows && pq < (unsigned int)ct.cols) ((unsigned char *)L.dmt)[pr * ct.cols + pq] = v0;
            }
          }
          go = (w0 != D2D_OCCUPIED);
        }
      }
      const bool general = ncand > (int)(8 * sizeof(RayMask)) || __any((ry.cmask & (ry.cmask - (RayMask)1)) != (RayMask)0);
      if (general) ray_march<true, FULL>(gx, c, g, L, ry, go && i < c.R, ncand, x0, y0, wt, ct, do_obs, dm);
      else ray_march<false, FULL>(gx, c, g, L, ry, go && i < c.R, ncand, x0, y0, wt, ct, do_obs, dm);
    }
    wave_sync_lds();
    D2D_STAMP(6);
    // OR over rays happened in LDS; newly_tracked = #{hit and not active}, utils.py:603-607
    for (int k0 = 0; k0 < N; k0 += WAVE) {
      const int k = k0 + lane;
      bool nw = false;
      if (k < N) {
        const unsigned char h = L.hit[k];
        s.hit[(size_t)e * N + k] = h;
        nw = h && !L.act[k];
      }
      newly += __popcll(__ballot(nw));
    }
    if (lane == 0) s.newly[e] = newly;
    r.tracked += newly;
  } else if (FULL ? (do_obs || do_trk || do_dyn || do_col) : (do_obs || do_trk)) {
    landed();
  }
  D2D_STAMP(7);
#ifndef D2D_ABL_NOTRK
  if (do_trk) {  // two instantiations: a run-time choice between an LDS and a global pointer would become flat loads
    // <= 16 agents on whole grids (SPEC 1): lane = element of a tracker's state, the record read from global memory.  With 17 to 40
    // agents the lane-per-tracker form fetches every filter in ONE round trip where the per-element form needs one per pass of three
    // trackers (config 4's step: 141 us against 160 us per 32768 envs): kept there.
    if (FULL && g.ncap <= 16) {
      st_tracker_elem(c, s, e, lane, g, L, r, noise_off);
    } else {
#ifdef D2D_TRK_LANE
      if (g.kf_lds) st_tracker<true>(c, s, e, lane, g, L, r, noise_off);
      else st_tracker<false>(c, s, e, lane, g, L, r, noise_off);
#else
      if (g.kf_lds) st_tracker_quad<true>(c, s, e, lane, g, L, r, noise_off);
      else st_tracker_quad<false>(c, s, e, lane, g, L, r, noise_off);
#endif
    }
  }
#endif
  D2D_STAMP(8);
#ifndef D2D_ABL_NODYN
  if (do_dyn) {
    if constexpr (FULL) {
      dyn_full(c, s, e, lane, L, gt);
    } else if (g.dyn2p) {  // grids above 256 x 256 cells: clear, fence, fetch again and mark (dyn_apply)
      if (lane < N) dyn_apply<1>(gx, c, s, e, lane, g, L, gt, dc);
      for (int k = WAVE + lane; k < N; k += WAVE) {
        DynCells dk;
        dyn_load<true, false>(gx, c, gt, k, L, dk);
        dyn_apply<1>(gx, c, s, e, k, g, L, gt, dk);
      }
      wave_sync_global();
      for (int k = lane; k < N; k += WAVE) {
        DynCells dk;
        dyn_load<false, true>(gx, c, gt, k, L, dk);
        dyn_apply<2>(gx, c, s, e, k, g, L, gt, dk);
      }
    } else {
      if (lane < N) dyn_apply<0>(gx, c, s, e, lane, g, L, gt, dc);
      // more than 64 agents: every further pass fetches the 18 cells of its 64 agents together (one round trip), then applies.
      // A later pass may read cells an earlier one has already written: the rule is order-independent (see dyn_apply).
      for (int k = WAVE + lane; k < N; k += WAVE) {
        DynCells dk;
        dyn_load(gx, c, gt, k, L, dk);
        dyn_apply<0>(gx, c, s, e, k, g, L, gt, dk);
      }
    }
  }
#endif
  D2D_STAMP(9);
#ifndef D2D_ABL_NOCOL
  if (do_col) st_collide(c, s, e, lane, L, probe_wall, r);
#endif
  D2D_STAMP(10);
  D2D_STAMP(11);
#ifndef D2D_ABL_NOOBS
  if (do_obs) {
    wave_sync_lds();
    if constexpr (FULL) obs_full(c, s, e, lane, L, ncx_d, ncy_d, r);
    else st_obs(c, s, e, lane, L, r);
  }
#endif
  D2D_STAMP(12);
}

// ------------------------------------------------------------------------------------------------
// Specialisation for the reference's default geometry (utils.py:66-72: 500 x 500 px map, scale 10, depth 80,
// FOV 90 => 50 x 50 cells, 50 rays, 33 x 33 crop; N <= 16 agents).  The kernel is bound by instruction issue
// and by SGPR pressure: every field of the by-value config is a live scalar the compiler cannot
// rematerialise, and past ~100 of them it spills to VGPR lanes (v_readlane on every use).  Overwriting the
// kernel's own copy of the config with the literals the host has verified turns them into immediates:
// constant-folded tile sizes and LDS offsets, no spills.  Any other config takes the generic instantiation.
// SPEC 1: N <= 16 agent slots, SPEC 2: N <= 40 (the default map plus the 14 obstacle_map agents, the reference's sweeps of up
// to 30 agents; 40 is where both grids whole + the agent planes still leave four workgroups per CU in every phase), SPEC 3: the
// default geometry with any N (LDS capacity and waves per workgroup stay run-time values)
__host__ __device__ constexpr int spec_ncap(int spec) { return spec == 1 ? 16 : (spec == 2 ? 40 : 0); }
// Both grids staged whole in LDS (Geom.full) only for the instantiation with few agents (N <= 16: the per-wave working set
// then stays below 10 KB, four 4-wave workgroups per CU).  With more agents the per-agent planes already fill the LDS and
// 5 KB more per wave cost occupancy in the persistent loop -- measured on BASELINE config 4 (24 agents): 4.9e7 env-steps/s
// with whole grids against 5.5e7 with the window / crop tiles; config 3 (172 agents): 8 -> 6 waves per CU, the step 35 % slower.
__host__ __device__ constexpr bool spec_full(int spec) { return spec == 1 || spec == 2; }
// 64 ray candidates on the mask path instead of 32: the default geometry with more than 32 agents (BASELINE config 3: 172 agents
// on 500 x 500 px).  Not the generic kernel: it is at its scalar-register limit (the per-lane predicates of the march live in
// SGPR pairs) and the wider mask costs config 5 -- 100 agents on 6400 x 6400 px, hardly ever a candidate -- a quarter of its
// raycast time in spilled scalars.
__host__ __device__ constexpr bool spec_wide(int spec) { return spec == 3 || spec == 0 || spec == 4; }
// SPEC 4: the generic kernel on tiled grids (d2d_cfg.grid_tile, GridIx)
__host__ __device__ constexpr bool spec_tiled(int spec) { return spec == 4; }
__host__ __device__ constexpr bool spec_generic(int spec) { return spec == 0 || spec == 4; }
// ray candidates culled against the cone of the rays (ray_cull<true>): where an env has enough agents for it to pay
__host__ __device__ constexpr bool spec_cone(int spec) { return spec != 1; }
__host__ __device__ inline bool spec_default_matches(const d2d_cfg &c) {
  return c.W == 50 && c.H == 50 && c.R == 50 && c.L == 33 && c.dt == 0.1 && c.scale == 10.0 &&
         c.W_px == 500.0 && c.H_px == 500.0 && c.ray_off0 == -0x1.921fb54442d18p-1 && c.ray_dth == 0x1.015bf9217271ap-5 &&
         c.depth == 80.0 && c.drone_radius == 10.0 && c.yaw_rate == 80.0 && c.max_acc == 40.0 && c.max_steps == 800.0 &&
         c.sigma == 0.0 && c.grid_tile == 0;
}

__device__ __forceinline__ void spec_default_apply(d2d_cfg &c) {
  c.W = 50; c.H = 50; c.R = 50; c.L = 33;
  c.dt = 0.1; c.scale = 10.0; c.W_px = 500.0; c.H_px = 500.0;
  c.ray_off0 = -0x1.921fb54442d18p-1; c.ray_dth = 0x1.015bf9217271ap-5;
  c.depth = 80.0; c.drone_radius = 10.0; c.yaw_rate = 80.0; c.max_acc = 40.0; c.max_steps = 800.0; c.sigma = 0.0;
  c.grid_tile = 0;
}

extern __shared__ __attribute__((aligned(16))) char d2d_lds[];

// The kernel's own argument block as it lies in the kernarg segment (constant memory).  The generic kernel reads the
// configuration and the state's pointers THROUGH it, field by field where they are used, instead of from the by-value
// parameters: those are all fetched on entry and stay live to the end -- some 130 scalar registers of arguments against the
// 102 there are, i.e. hundreds of spill reloads (v_readlane) inside the ray and agent loops.  The specialised kernels fold
// the configuration into constants and do not need this.
struct StagesKArgs {
  d2d_cfg c;
  d2d_state s;
  uint32_t stages;
  const double *pin;
  unsigned char *coll_out;
};

// The ONE parameter of the kernel is that block, by value: the kernarg segment then holds exactly a StagesKArgs at offset 0 by
// construction (a separate parameter list would have to be kept in step with the struct by hand).
template <int SPEC>
__global__ __launch_bounds__(WAVE *WAVES_PER_BLOCK, D2D_MIN_WAVES) void k_stages(StagesKArgs args) {
#if defined(__HIP_DEVICE_COMPILE__)
  const StagesKArgs *ka = (const StagesKArgs *)(const __attribute__((address_space(4))) StagesKArgs *)__builtin_amdgcn_kernarg_segment_ptr();
#else
  const StagesKArgs *ka = nullptr;  // the host pass only parses this
#endif
  const d2d_cfg &c_in = args.c;
  const d2d_state &s_in = args.s;
  const uint32_t stages = args.stages;
  const double *pin = args.pin;
  unsigned char *coll_out = args.coll_out;
  d2d_cfg c_folded;
  if (!spec_generic(SPEC)) {
    c_folded = c_in;
    spec_default_apply(c_folded);
  }
  const d2d_cfg &c = !spec_generic(SPEC) ? c_folded : ka->c;
  const d2d_state &s = (SPEC == 1 || SPEC == 2) ? s_in : ka->s;  // the many-agent kernel folds cfg, but its state pointers spill too
  // wave-uniform by construction; readfirstlane tells the compiler, so every per-env base pointer and LDS
  // base lives in SGPRs and loads take the scalar-base + 32-bit-offset form
  const int lane = threadIdx.x & (WAVE - 1), wpb = (SPEC == 1 || SPEC == 2) ? WAVES_PER_BLOCK : (int)(blockDim.x / WAVE);
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
  const int e = blockIdx.x * wpb + wv;
  if (e >= c.B) return;
  if ((stages & D2D_ST_SKIP_DONE) && s.flags[(size_t)e * 4 + D2D_F_DONE] != 0) return;
  const Geom g = make_geom(c, wpb, spec_ncap(SPEC), spec_full(SPEC));
  const LdsView L = carve(d2d_lds + (size_t)wv * g.wave_bytes, g, c.L);
  EnvRegs r;
#ifdef D2D_STAMPS
  if (d2d_stamp_buf && lane == 0) d2d_stamp_buf[(size_t)e * 16 + 0] = __builtin_amdgcn_s_memtime();
#endif
  load_regs(s, e, r);
#ifdef D2D_NO_PIN
  pin = nullptr;
  coll_out = nullptr;
#endif
  if (pin) {  // d2d_rollout: env.drone.x = x; env.drone.y = y before the step
    r.x = pin[(size_t)e * 2];
    r.y = pin[(size_t)e * 2 + 1];
  }
  run_env<spec_full(SPEC), spec_wide(SPEC), spec_cone(SPEC), spec_tiled(SPEC)>(c, s, e, lane, stages, g, L, s.action[e], r);
  if (lane == 0) {
    store_regs(s, e, r);
    if (coll_out) coll_out[e] = s.flags[(size_t)e * 4 + D2D_F_COLLISION];
  }
  D2D_STAMP(13);
}

// reset(): copy of the snapshot over the live state of env e by one wave
__device__ __forceinline__ void reset_env(const d2d_cfg &c, const d2d_state &s, const d2d_state &init, size_t e, int lane) {
  const size_t N = c.N, WH = grid_bytes(c), LL = (size_t)c.L * c.L;
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
  if (lane < D2D_DF) s.drone[e * D2D_DF + lane] = init.drone[e * D2D_DF + lane];
  if (lane < D2D_CF) s.counters[e * D2D_CF + lane] = init.counters[e * D2D_CF + lane];
  if (lane < 2) s.target[e * 2 + lane] = init.target[e * 2 + lane];
  if (lane < 4) s.flags[e * 4 + lane] = 0;
  if (lane == 0) {
    s.newly[e] = 0;
    s.obs_yaw[e] = 0.f;
  }
}

// masked reset, one wave per env
__global__ __launch_bounds__(WAVE *WAVES_PER_BLOCK) void k_reset(d2d_cfg c, d2d_state s, d2d_state init,
                                                                 const unsigned char *mask, int mask_stride) {
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  const int e = blockIdx.x * WAVES_PER_BLOCK + wv;
  if (e >= c.B) return;
  if (mask && !mask[(size_t)e * mask_stride]) return;
  reset_env(c, s, init, (size_t)e, lane);
}

__global__ void k_tan(const double *in, double *out, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = d2d_tan(in[i]);
}

#include "d2d_plugins.h"

// ------------------------------------------------------------------------------------------------
// The closed loop as ONE persistent launch: every wave runs `nsteps` reference-style steps of its own env,
//   [reset if the previous step ended the episode] -> Oxford.plan -> perceive -> Primitive.replan_check / plan -> act,
// with only wave-local fences between the phases.  Envs are independent, so nothing has to wait at a kernel
// boundary for the slowest env of the batch: a 99-expansion search (up to ~1 ms) costs that env's wave its own
// time instead of stalling 4095 others every step, and a step costs no launches at all.
// Instantiated like the step kernel (SPEC 0-3); a configuration whose phases do not fit the LDS budget of one workgroup
// takes the launch-per-stage path.
// ------------------------------------------------------------------------------------------------
// The launch arguments (four structs of pointers) are parked once in device memory (d2d_plan.launch_args) and every
// phase is a NON-INLINED function that reads what it needs through scalar loads: inlined into one loop body the
// by-value arguments all stay live across the loop (474 SGPR + 419 VGPR spills, measured); as calls each phase
// gets the register allocation it has as a kernel of its own.
struct ClosedArgs {
  d2d_cfg c;
  d2d_state s;
  d2d_plan p;
  d2d_state init;
  int on_done, nsteps;
};
static_assert(sizeof(ClosedArgs) <= D2D_LAUNCH_ARGS_BYTES, "d2d_plan.launch_args (D2D_LAUNCH_ARGS_BYTES) is too small for the parked launch arguments");

__global__ void k_closed_args(ClosedArgs *dst, d2d_cfg c, d2d_state s, d2d_plan p, d2d_state init, int on_done, int nsteps) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    dst->c = c;
    dst->s = s;
    dst->p = p;
    dst->init = init;
    dst->on_done = on_done;
    dst->nsteps = nsteps;
  }
}

template <int SPEC>
__host__ __device__ inline int closed_wave_bytes(const d2d_cfg &c, const d2d_plan &p, int wpb) {
  int b = make_geom(c, wpb, spec_ncap(SPEC), spec_full(SPEC)).wave_bytes;
  const int pb = plan_wave_bytes(c.N, p.nu, p.n_sample, c.W, c.H), gb = p.gaze == D2D_GAZE_OXFORD ? gaze_geom(c, p).wave_bytes : 0;
  b = b > pb ? b : pb;
  b = b > gb ? b : gb;
  return (b + 15) & ~15;
}


// function arguments arrive in VGPRs; these are wave-uniform by construction, say so -- and say that the bundle lives in
// device memory nobody writes during the launch (constant address space): its fields then come through scalar loads
// instead of per-lane `flat_load`s
#if defined(__HIP_DEVICE_COMPILE__)
typedef const ClosedArgs __attribute__((address_space(4))) *ArgsPtr;
#else
typedef const ClosedArgs *ArgsPtr;  // the host pass only parses the device functions
#endif

// What a phase starts from: the bundle's address, the env index, the LDS offset of the wave -- wave-uniform by construction, said so
// with readfirstlane -- and the lane index.  A phase that is INLINED into the persistent kernel (D2D_PH_INLINE / the search) launders
// them through an empty asm: nothing it computes is then loop-invariant in the caller's step loop (hoisted out of the loop, such
// values stay live across the other phases and spill), and every phase keeps the live ranges it has as a function of its own.
struct PhaseIn {
  ArgsPtr a;
  int e, off, lane;
};
template <bool LAUNDER>
__device__ __forceinline__ PhaseIn phase_enter(const ClosedArgs *ap, int e_, int lds_off_) {
  const unsigned long long v = (unsigned long long)ap;
  unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)v), hi = __builtin_amdgcn_readfirstlane((unsigned int)(v >> 32));
  int e = __builtin_amdgcn_readfirstlane(e_), off = __builtin_amdgcn_readfirstlane(lds_off_);
  int lane = threadIdx.x & (WAVE - 1);
  if constexpr (LAUNDER) asm volatile("; phase" : "+s"(lo), "+s"(hi), "+s"(e), "+s"(off), "+v"(lane));
  PhaseIn in;
  in.a = (ArgsPtr)(((unsigned long long)hi << 32) | lo);
  in.e = e;
  in.off = off;
  in.lane = lane;
  return in;
}
// Round 4: EVERY phase is inlined into the persistent kernel (a kernel has no callee-saved registers: as functions the phases saved
// and restored up to 48 VGPRs per call -- 11 KB of scratch written and read per env-step in the gaze + perceive phase of the
// many-agent kernels, 12 KB per search).  Two things make that work where rounds 1-3 measured spills: the laundering above, and
// building WITHOUT MachineLICM (csrc/build.sh: -mllvm -disable-machine-licm) -- that pass hoists the 64-bit constants of every phase
// (polynomial coefficients of tan / sin / cos, thresholds) to the kernel's entry, where they are live across the whole step loop;
// a 64-bit literal is two moves and not rematerialisable, so the allocator SPILLS constants (18 at entry, ~130 reloads inside the
// loop, and scalar ones into VGPR lanes).  Without the pass the inlined kernel has no scratch at all.  -DD2D_PH_CALLS builds the
// called form (the search stays inlined), e.g. for A/B runs.
#ifndef D2D_PH_CALLS
#define D2D_PH_INLINE 1
#endif
#ifdef D2D_PH_INLINE
#define D2D_PH_ATTR __forceinline__
constexpr bool kPhaseLaunder = true;
#else
#define D2D_PH_ATTR __attribute__((noinline))
constexpr bool kPhaseLaunder = false;
#endif

// The planner stage as two calls: the part every step runs (small: few registers to save), and the search, called
// only when the trajectory is empty (a few percent of the steps).
template <int SPEC>
__device__ D2D_PH_ATTR int ph_plan_quick(const ClosedArgs *ap, int e_, int lds_off_) {
  const PhaseIn in_ = phase_enter<kPhaseLaunder>(ap, e_, lds_off_);
  const ArgsPtr a = in_.a;
  const int e = in_.e, lane = in_.lane;
  char *base = d2d_lds + in_.off;
  d2d_cfg c = a->c;
  if (!spec_generic(SPEC)) spec_default_apply(c);
  if constexpr (SPEC == 1 || SPEC == 2) {  // folds the size of the search's cost mirror (search_lds_nodes)
    constexpr int cap = spec_ncap(SPEC);
    __builtin_assume(c.N <= cap);
  }
  const bool need = plan_env_quick(c, a->s, a->p, e, lane, base);
  wave_sync_global();
  return need ? 1 : 0;
}

// The search is INLINED into the persistent kernel (round 4): as a function it needs all 128 VGPRs of the kernel's budget, i.e. all
// 48 callee-saved ones, and saved + restored them around every call -- 12 KB of scratch written and read per search, 15 KB of
// corrected HBM traffic per env-step on BASELINE config 3 (0.42 searches per env-step) -- for a caller that keeps nothing but
// wave-uniform values in scalar registers.  A kernel has no callee-saved registers.  What the inlined body reads (the argument
// bundle's address, the env index, the LDS offset, the lane index) is laundered through an empty asm at the call site, so that no
// value of the search is loop-invariant in the caller's step loop: hoisted out of it, such values stay live across the other
// phases' calls and spill.  Same-call A/B against the called form: config 2 +3 % (600 / 300 and the driver's 20-step window),
// config 3 +4 %, config 4 +2.4 %.  -DD2D_SEARCH_CALL builds the called form.
#ifndef D2D_SEARCH_CALL
#define D2D_SEARCH_INLINE 1
#define D2D_SEARCH_ATTR __forceinline__
#else
#define D2D_SEARCH_ATTR __attribute__((noinline))
#endif
template <int SPEC>
__device__ D2D_SEARCH_ATTR void ph_plan_search(const ClosedArgs *ap, int e_, int lds_off_) {
#ifdef D2D_SEARCH_INLINE
  const PhaseIn in_ = phase_enter<true>(ap, e_, lds_off_);
#else
  const PhaseIn in_ = phase_enter<false>(ap, e_, lds_off_);
#endif
  const ArgsPtr a = in_.a;
  const int e = in_.e, lane = in_.lane;
  char *base = d2d_lds + in_.off;
  d2d_cfg c = a->c;
  if (!spec_generic(SPEC)) spec_default_apply(c);
  if constexpr (SPEC == 1 || SPEC == 2) {  // folds the size of the search's cost mirror (search_lds_nodes)
    constexpr int cap = spec_ncap(SPEC);
    __builtin_assume(c.N <= cap);
  }
  plan_env_search(c, a->s, a->p, e, lane, base);
  wave_sync_global();
}

// The planner's every-step part AND the act phase in one call, for the steps whose trajectory is kept (96 % of them): the
// planner's result reaches the control stage in registers -- no store -> fence -> load between the two, one call, one scalar-load
// round trip for the arguments, one fence fewer per step.  Returns -1 when Primitive.plan has to search (the caller runs
// ph_plan_search and then ph_stages<ACT>), else the episode flag the collision stage wrote.
// `walls_ok`: the env's "every remaining waypoint passed the wall test and only rays have written the map since" flag
// (plan_env_quick), carried by the caller across the steps of a launch.
template <int SPEC>
__device__ D2D_PH_ATTR int ph_plan_act(const ClosedArgs *ap, int e_, int lds_off_, int &walls_ok) {
  const PhaseIn in_ = phase_enter<kPhaseLaunder>(ap, e_, lds_off_);
  const ArgsPtr a = in_.a;
  const int e = in_.e, lane = in_.lane;
  char *base = d2d_lds + in_.off;
  d2d_cfg c = a->c;
  if (!spec_generic(SPEC)) spec_default_apply(c);
  if constexpr (SPEC == 1 || SPEC == 2) {  // folds the size of the search's cost mirror (search_lds_nodes)
    constexpr int cap = spec_ncap(SPEC);
    __builtin_assume(c.N <= cap);
  }
  double4 w_head;
  int wk = __builtin_amdgcn_readfirstlane(walls_ok);
  if constexpr (kPhaseLaunder) asm volatile("; phase" : "+s"(wk));
  const bool need = plan_env_quick(c, a->s, a->p, e, lane, base, &w_head, &wk);
  walls_ok = __builtin_amdgcn_readfirstlane(wk);
  if (need) {
    wave_sync_global();
    return -1;
  }
  // (nothing the act phase loads was written above: the planner's stores are its own state and the result handed over here;
  // its LDS staging has drained before the act phase's copies land in the same bytes)
  wave_sync_lds();
  StepIn in;
  in.action = 0.0;
  in.ok = true;
  in.has_wp = true;
  in.wp[0] = w_head.x; in.wp[1] = w_head.y; in.wp[2] = w_head.z; in.wp[3] = w_head.w; in.wp[4] = 0.0; in.wp[5] = 0.0;
  const int wpb = (int)(blockDim.x / WAVE);
  const Geom g = make_geom(c, wpb, spec_ncap(SPEC), spec_full(SPEC));
  const LdsView L = carve(base, g, c.L);
  EnvRegs r;
  run_env<spec_full(SPEC), spec_wide(SPEC), spec_cone(SPEC), spec_tiled(SPEC)>(c, a->s, e, lane, D2D_ST_ACT, g, L, a->s.action[e], r, 0, true, &in);
  if (lane == 0) store_regs(a->s, e, r);
  wave_sync_global();
  return r.done;
}

// gaze + the stages that follow it in one call (one set of callee-saved registers, one fence fewer per step)
// `done_`: the env's episode flag as the caller knows it (the act phase of the step before returns it) -- no load, no round trip,
// before the gaze stage can ask for anything else.  Returns the flag as this call's collision stage wrote it (-1: it did not run).
template <int SPEC, uint32_t STAGES>
__device__ D2D_PH_ATTR int ph_gaze_stages(const ClosedArgs *ap, int e_, int lds_off_, int t_, int done_) {
  const PhaseIn in_ = phase_enter<kPhaseLaunder>(ap, e_, lds_off_);
  const ArgsPtr a = in_.a;
  const int e = in_.e, lane = in_.lane;
  int tstep = __builtin_amdgcn_readfirstlane(t_), known_done = __builtin_amdgcn_readfirstlane(done_);
  if constexpr (kPhaseLaunder) asm volatile("; phase" : "+s"(tstep), "+s"(known_done));
  char *base = d2d_lds + in_.off;
  d2d_cfg c = a->c;
  if (!spec_generic(SPEC)) spec_default_apply(c);
#ifdef D2D_CHAIN_PROF
  const unsigned long long pp0 = __builtin_amdgcn_s_memtime();
#endif
  gaze_env(c, a->s, a->p, a->init, a->on_done == D2D_DONE_RESET, e, lane, base, known_done);
  wave_sync_global();
#ifdef D2D_CHAIN_PROF
  D2D_PHASE_ADD(0, pp0);
  const unsigned long long pp1 = __builtin_amdgcn_s_memtime();
#endif
  const int wpb = (int)(blockDim.x / WAVE);
  const Geom g = make_geom(c, wpb, spec_ncap(SPEC), spec_full(SPEC));
  const LdsView L = carve(base, g, c.L);
  EnvRegs r;
  load_regs(a->s, e, r);
  // this step's row of the measurement noise (d2d_cfg.noise_rows; utils.py:605 draws fresh normals every step)
  const size_t noise_off = c.noise_rows > 1 ? (size_t)((c.noise_row0 + tstep) % c.noise_rows) * c.B * c.N * 2 : 0;
  // (the plain order of loads, not run_env's one-batch form: with the agents' first pass held in registers across the grid copies
  // this phase saves 13 callee-saved registers per call instead of 6 -- 1.8 KB of scratch writes per env-step for +0.6 %)
  run_env<spec_full(SPEC), spec_wide(SPEC), spec_cone(SPEC), spec_tiled(SPEC)>(c, a->s, e, lane, STAGES, g, L, a->s.action[e], r, noise_off, false);
  if (lane == 0) store_regs(a->s, e, r);
  wave_sync_global();
#ifdef D2D_CHAIN_PROF
  D2D_PHASE_ADD(1, pp1);
#endif
  return r.done;
}

template <int SPEC, uint32_t STAGES>
__device__ D2D_PH_ATTR int ph_stages(const ClosedArgs *ap, int e_, int lds_off_) {
  const PhaseIn in_ = phase_enter<kPhaseLaunder>(ap, e_, lds_off_);
  const ArgsPtr a = in_.a;
  const int e = in_.e, lane = in_.lane;
  char *base = d2d_lds + in_.off;
  d2d_cfg c = a->c;
  if (!spec_generic(SPEC)) spec_default_apply(c);
  const int wpb = (int)(blockDim.x / WAVE);
  const Geom g = make_geom(c, wpb, spec_ncap(SPEC), spec_full(SPEC));
  const LdsView L = carve(base, g, c.L);
  EnvRegs r;
  run_env<spec_full(SPEC), spec_wide(SPEC), spec_cone(SPEC), spec_tiled(SPEC)>(c, a->s, e, lane, STAGES, g, L, a->s.action[e], r, 0, true);
  if (lane == 0) store_regs(a->s, e, r);
  wave_sync_global();
  return r.done;
}

template <int SPEC>
__global__ __launch_bounds__(WAVE *WAVES_PER_BLOCK, D2D_MIN_WAVES) void k_closed(const ClosedArgs *__restrict__ a) {
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
  const int wpb = (int)(blockDim.x / WAVE);
  const int e = blockIdx.x * wpb + wv;
  if (e >= a->c.B) return;
  d2d_cfg c = a->c;
  if (!spec_generic(SPEC)) spec_default_apply(c);
  const int off = wv * closed_wave_bytes<SPEC>(c, a->p, wpb);
  const bool split = a->p.planner == D2D_PLAN_PRIMITIVE;
  const int nsteps = a->nsteps;
  const bool freeze = a->on_done == D2D_DONE_FREEZE;
  int nsearch = 0;
  int walls_ok = 0;  // (plan_env_quick: nothing is known about the trajectory's waypoints at the start of a launch)
#ifdef D2D_CHAIN_PROF
  // Diagnostic build only (tools/chain_prof.py): how long this env's chain of steps ran (shader clocks) and how much of it
  // its searches took -- the launch lasts as long as the longest chain.
  const unsigned long long cp0 = __builtin_amdgcn_s_memtime();
  unsigned long long cps = 0;
#endif
  // the env's episode flag, carried in a register from the collision stage that writes it to the gaze stage that asks for it
  int flag_done = a->s.flags[(size_t)e * 4 + D2D_F_DONE] != 0 ? 1 : 0;
#pragma unroll 1
  for (int t = 0; t < nsteps; ++t) {
    if (freeze && flag_done != 0) break;  // one episode per env: it stays as it ended
    if (split) {
      // An env that searches often is the one the launch ends up waiting for (the worlds repeat, so it stays that env):
      // past one search per 32 steps of this launch all its phases issue first on their SIMD.  Its chain is
      // latency-bound, so the three waves it shares the SIMD with give up little.  (+4 % at 4096 envs; thresholds 16-64
      // and levels 1-2 measure the same, level 3 -- the search's own -- less.)
      if (nsearch * 32 > t + 16) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
      ph_gaze_stages<SPEC, D2D_ST_PERCEIVE>(a, e, off, t, flag_done);
#ifdef D2D_CHAIN_PROF
      const unsigned long long q0 = __builtin_amdgcn_s_memtime();
#endif
      // the planner's every-step part and, unless it has to search, the act phase behind it in the same call (ph_plan_act)
      const int pa = __builtin_amdgcn_readfirstlane(ph_plan_act<SPEC>(a, e, off, walls_ok));
#ifdef D2D_CHAIN_PROF
      D2D_PHASE_ADD(2, q0);  // (planner every-step part + act of the steps without a search)
#endif
      if (pa < 0) {
#ifdef D2D_CHAIN_PROF
        const unsigned long long s0 = __builtin_amdgcn_s_memtime();
#endif
        ph_plan_search<SPEC>(a, e, off);
        nsearch += 1;
        walls_ok = 0;  // a new trajectory (or none): its waypoints have not been through replan_check yet
#ifdef D2D_CHAIN_PROF
        cps += __builtin_amdgcn_s_memtime() - s0;
        D2D_PHASE_ADD(3, s0);
        const unsigned long long a0 = __builtin_amdgcn_s_memtime();
#endif
        flag_done = __builtin_amdgcn_readfirstlane(ph_stages<SPEC, D2D_ST_ACT>(a, e, off));
#ifdef D2D_CHAIN_PROF
        D2D_PHASE_ADD(4, a0);  // (act of the steps with a search)
#endif
      } else {
        flag_done = pa;
      }
    } else {
      flag_done = __builtin_amdgcn_readfirstlane(ph_gaze_stages<SPEC, D2D_ST_ALL>(a, e, off, t, flag_done));
    }
  }
#ifdef D2D_CHAIN_PROF
  if ((threadIdx.x & (WAVE - 1)) == 0 && a->p.plan_stat) {
    const unsigned long long tot = __builtin_amdgcn_s_memtime() - cp0;
    a->p.plan_stat[(size_t)e * 4 + 1] = (int)(tot >> 4);   // 16-clock units
    a->p.plan_stat[(size_t)e * 4 + 2] = (int)(cps >> 4);
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// host side of the C ABI
// ------------------------------------------------------------------------------------------------
thread_local char g_err[256] = "";

int fail(int code, const char *msg) {
  snprintf(g_err, sizeof g_err, "%s", msg);
  return code;
}

// LDS of one workgroup.  A CU has 160 KB; occupancy (waves per CU) is what the per-wave working set allows whatever the
// workgroup size, so workgroups stay within 64 KB (several of them per CU) -- except that a SINGLE wave whose working set
// exceeds 64 KB (many agents, deep views, large plugin tables) gets a workgroup of its own with up to the whole 160 KB,
// which the kernel has to opt into (hipFuncAttributeMaxDynamicSharedMemorySize).
constexpr size_t LDS_SOFT = 64 * 1024, LDS_HARD = 160 * 1024;

template <typename K>
void lds_optin(K kernel, size_t bytes) {
  if (bytes > LDS_SOFT) (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// does this configuration take the specialised kernels (default geometry: grids staged whole in LDS)?
bool spec_path(const d2d_cfg &c) {
#ifndef D2D_NO_SPEC
  return spec_default_matches(c);
#else
  return false;
#endif
}

// Envs (waves) per workgroup -- 4, 2 or 1 -- for a per-wave working set of bytes(wpb): the choice that puts the most waves on a
// CU (workgroups of wpb * bytes within 64 KB, or a single wave with up to 160 KB, as many of them as 160 KB hold; 16 waves is
// the register limit anyway).  Ties go to the larger workgroup.  0 = does not fit at all.  E.g. 16.9 KB per wave: 2 per
// workgroup gives 4 workgroups = 8 waves, 1 per workgroup 9.
template <typename F>
int best_wpb(F bytes) {
  int best = 0, best_waves = 0;
  for (int wpb = WAVES_PER_BLOCK; wpb >= 1; wpb >>= 1) {
    const size_t wg = (size_t)bytes(wpb) * wpb;
    if (wg == 0 || wg > (wpb == 1 ? LDS_HARD : LDS_SOFT)) continue;
    int waves = (int)(LDS_HARD / ((wg + 1023) & ~(size_t)1023)) * wpb;  // allocation granularity: be conservative
    if (waves > 16) waves = 16;
    if (waves > best_waves) {
      best = wpb;
      best_waves = waves;
    }
  }
  return best;
}

int pick_wpb(const d2d_cfg &c) {
  const bool full = spec_path(c) && spec_full(c.N <= spec_ncap(1) ? 1 : (c.N <= spec_ncap(2) ? 2 : 3));
  return best_wpb([&](int wpb) { return make_geom(c, wpb, 0, full).wave_bytes; });
}

int check(const d2d_cfg *c, const d2d_state *s) {
  if (!c || !s) return fail(-1, "null cfg/state");
  if (c->abi_version != D2D_ABI_VERSION) return fail(-2, "ABI version mismatch");
  if (c->B < 0 || c->N < 0 || c->W <= 0 || c->H <= 0 || c->R <= 0 || c->L <= 0 || (c->L & 1) == 0 || c->T <= 0)
    return fail(-1, "bad dimensions");
  if (c->noise_row0 < 0 || c->noise_row0 >= (c->noise_rows > 1 ? c->noise_rows : 1)) return fail(-1, "noise_row0 outside [0, noise_rows)");
  if (c->grid_tile != 0 && c->grid_tile != 16) return fail(-1, "grid_tile must be 0 (row-major [W][H]) or 16 (16 x 16-cell tiles)");
  if (c->W > 32767 || c->H > 32767) return fail(-4, "grids of more than 32767 cells a side are not supported (16-bit cell planes in LDS)");
  if (!(c->scale >= 2.0) || c->scale != (double)(long long)c->scale)
    return fail(-4, "map_scale must be an integer >= 2 (scale 1 never advances a ray, utils.py:621)");
  if (!(c->depth > 0) || !(c->dt > 0)) return fail(-1, "bad depth / dt");
  if (c->kf_enabled && (!s->kf || !s->kf_len)) return fail(-1, "kf_enabled without kf buffers");
  if (c->sigma != 0.0 && c->kf_enabled && c->N > 0 && !s->noise) return fail(-1, "var_cam != 0 needs the noise input");
  if (!s->agents || !s->agent_unit || !s->dyn_prev || !s->gt || !s->dmap || !s->drone || !s->target || !s->targets ||
      !s->counters || !s->active || !s->hit || !s->newly || !s->flags || !s->obs_local || !s->obs_yaw)
    return fail(-1, "null state pointer");
  if (pick_wpb(*c) == 0) return fail(-4, "N / view depth too large for the per-env LDS working set");
  return 0;
}

int launch_stages(const d2d_cfg *c, const d2d_state *s, uint32_t stages, void *stream, const double *pin = nullptr,
                  unsigned char *coll_out = nullptr) {
  int rc = check(c, s);
  if (rc) return rc;
  if (!s->action && (stages & D2D_ST_CONTROL)) return fail(-1, "null action");
  if ((stages & D2D_ST_CONTROL) && c->planner_mode == D2D_PLANNER_EXTERNAL && (!s->plan_ok || !s->wp_valid || !s->wp))
    return fail(-1, "external planner mode needs plan_ok / wp_valid / wp");
  if (c->B == 0) return 0;
  d2d_state st = *s;
  if (!st.action) st.action = (const double D2D_AS *)st.drone;  // never dereferenced meaningfully without CONTROL
  StagesKArgs ka;
  memset(&ka, 0, sizeof ka);
  ka.c = *c;
  ka.s = st;
  ka.stages = stages;
  ka.pin = pin;
  ka.coll_out = coll_out;
  if (spec_path(*c) && c->N > spec_ncap(2)) {
    const int wpb = pick_wpb(*c);
    const Geom g = make_geom(*c, wpb, 0, spec_full(3));
    const dim3 grid((c->B + wpb - 1) / wpb), block(WAVE * wpb);
    lds_optin(k_stages<3>, (size_t)g.wave_bytes * wpb);
    hipLaunchKernelGGL(k_stages<3>, grid, block, (size_t)g.wave_bytes * wpb, (hipStream_t)stream, ka);
  } else if (spec_path(*c)) {
    const int spec = c->N <= spec_ncap(1) ? 1 : 2;
    const Geom g = make_geom(*c, WAVES_PER_BLOCK, spec_ncap(spec), spec_full(spec));
    const dim3 grid((c->B + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), block(WAVE * WAVES_PER_BLOCK);
    const size_t lds = (size_t)g.wave_bytes * WAVES_PER_BLOCK;
    if (spec == 1) hipLaunchKernelGGL(k_stages<1>, grid, block, lds, (hipStream_t)stream, ka);
    else hipLaunchKernelGGL(k_stages<2>, grid, block, lds, (hipStream_t)stream, ka);
  } else {
    const int wpb = pick_wpb(*c);
    const Geom g = make_geom(*c, wpb);
    const dim3 grid((c->B + wpb - 1) / wpb), block(WAVE * wpb);
    if (c->grid_tile) {
      lds_optin(k_stages<4>, (size_t)g.wave_bytes * wpb);
      hipLaunchKernelGGL(k_stages<4>, grid, block, (size_t)g.wave_bytes * wpb, (hipStream_t)stream, ka);
    } else {
      lds_optin(k_stages<0>, (size_t)g.wave_bytes * wpb);
      hipLaunchKernelGGL(k_stages<0>, grid, block, (size_t)g.wave_bytes * wpb, (hipStream_t)stream, ka);
    }
  }
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail(-3, hipGetErrorString(err));
  return 0;
}

int reset_launch(const d2d_cfg *c, const d2d_state *s, const d2d_state *init, const uint8_t *mask, int mask_stride,
                 void *stream) {
  int rc = check(c, s);
  if (rc) return rc;
  if (!init || !init->agents || !init->agent_unit || !init->dyn_prev || !init->gt || !init->dmap || !init->drone ||
      !init->target || !init->targets || !init->counters || !init->active)
    return fail(-1, "reset: incomplete snapshot");
  if (c->B == 0) return 0;
  const dim3 grid((c->B + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), block(WAVE * WAVES_PER_BLOCK);
  hipLaunchKernelGGL(k_reset, grid, block, 0, (hipStream_t)stream, *c, *s, *init, mask, mask_stride);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail(-3, hipGetErrorString(err));
  return 0;
}

int plan_check(const d2d_cfg *c, const d2d_state *s, const d2d_plan *p) {
  int rc = check(c, s);
  if (rc) return rc;
  if (!p) return fail(-1, "null plan");
  if (p->planner == D2D_PLAN_PRIMITIVE || p->gaze == D2D_GAZE_OXFORD) {
    if (!p->traj || !p->traj_hdr) return fail(-1, "plan: null trajectory buffers");
    if (!c->kf_enabled) return fail(-4, "device plugins need the Kalman trackers on the device (kf_enabled)");
  }
  if (p->planner == D2D_PLAN_PRIMITIVE) {
    if (!p->u_space || !p->sample_t || !p->traj_t || !p->trk_radius || !p->trk_prev || !p->trk_lim || !p->nodes || !p->hash || !p->plan_stat)
      return fail(-1, "plan: null planner pointer");
    if (p->hash_cap <= p->node_cap || (p->hash_cap & (p->hash_cap - 1)))
      return fail(-1, "plan: hash_cap must be a power of two > node_cap");
    if (p->nu <= 0 || p->n_sample <= 0 || p->n_ts <= 0 || p->traj_cap < p->n_ts || p->node_cap < 2)
      return fail(-1, "plan: bad planner dimensions");
    if (!s->plan_ok || !s->wp_valid || !s->wp) return fail(-1, "plan: plan_ok / wp_valid / wp buffers missing");
    // the squared thresholds a caller hands over must be THE thresholds of vmax / goal_tol: the search trusts them
    if (p->vmax_sq != 0.0) {
      const bool none = !(p->vmax > 0.0);  // nothing is `< vmax`: the caller says so with -1
      const bool ok = none ? p->vmax_sq == -1.0
                           : (p->vmax_sq > 0.0 && sqrt(p->vmax_sq) < p->vmax && sqrt(nextafter(p->vmax_sq, INFINITY)) >= p->vmax);
      if (!ok) return fail(-1, "plan: vmax_sq is not the largest s with sqrt(s) < vmax (0 = let the library find it; -1 when vmax <= 0)");
    }
    if (p->goal_sq != 0.0) {
      const bool ok = p->goal_tol >= 0.0 && p->goal_sq >= 0.0 && sqrt(p->goal_sq) <= p->goal_tol && sqrt(nextafter(p->goal_sq, INFINITY)) > p->goal_tol;
      if (!ok) return fail(-1, "plan: goal_sq is not the largest s with sqrt(s) <= goal_tol (0 = let the library find it)");
    }
    if ((size_t)plan_wave_bytes(c->N, p->nu, p->n_sample, c->W, c->H) > LDS_HARD) return fail(-4, "plan: too many agents for the tracker staging in LDS");
  }
  if (p->gaze == D2D_GAZE_OXFORD) {
    if (!p->yaw_space || !p->tobs_tab || !p->pw_leaf || !p->pw_tree || !p->pw_rowleaf || !p->seen_step) return fail(-1, "plan: null gaze pointer");
    if (!s->action) return fail(-1, "plan: null action buffer");
    if (p->n_yaw <= 0 || p->n_yaw > 7) return fail(-4, "gaze: at most 7 yaw-rate candidates");
    if (p->pw_nleaf <= 0 || p->pw_nprog != 2 * p->pw_nleaf - 1 || p->pw_ntree < 3) return fail(-1, "gaze: bad pairwise-sum program");
    if ((size_t)gaze_geom(*c, *p).wave_bytes > LDS_HARD)
      return fail(-4, "gaze: map / view depth too large for the per-env LDS working set (above 4096 cells a view deeper than 13 cells "
                      "needs the whole pairwise-sum plan of the map in LDS)");
    if ((long long)c->W * c->H >= (1ll << 24)) return fail(-4, "gaze: maps of 2^24 cells and more are not supported");
    if (gaze_geom(*c, *p).bbn > 64 || c->W * c->H < 64)
      return fail(-4, "gaze: view depth above 29 cells or a map below 64 cells");
    if (c->max_steps + 2 > (double)p->tobs_len) return fail(-1, "gaze: tobs_tab shorter than the longest episode");
  }
  return 0;
}

// `init` != NULL: envs whose flags say "done" are first put back to the snapshot, plugin state included
int gaze_launch(const d2d_cfg *c, const d2d_state *s, const d2d_plan *p, const d2d_state *init, bool skip_done, void *stream) {
  if ((p->gaze != D2D_GAZE_OXFORD && !init) || c->B == 0) return 0;
  const size_t wb = p->gaze == D2D_GAZE_OXFORD ? (size_t)gaze_geom(*c, *p).wave_bytes : 0;
  int wpb = WAVES_PER_BLOCK;
  while (wpb > 1 && wb * wpb > LDS_SOFT) wpb >>= 1;
  const dim3 grid((c->B + wpb - 1) / wpb), block(WAVE * wpb);
  const size_t lds = wb * wpb;
  lds_optin(k_gaze, lds);
  hipLaunchKernelGGL(k_gaze, grid, block, lds, (hipStream_t)stream, *c, *s, *p, init ? *init : *s, init ? 1 : (skip_done ? 2 : 0));
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail(-3, hipGetErrorString(err));
  return 0;
}

int plan_launch(const d2d_cfg *c, const d2d_state *s, const d2d_plan *p, bool skip_done, void *stream) {
  if (p->planner != D2D_PLAN_PRIMITIVE || c->B == 0) return 0;
  const size_t wb = (size_t)plan_wave_bytes(c->N, p->nu, p->n_sample, c->W, c->H);
  int wpb = WAVES_PER_BLOCK;
  while (wpb > 1 && wb * wpb > LDS_SOFT) wpb >>= 1;
  const dim3 grid((c->B + wpb - 1) / wpb), block(WAVE * wpb);
  const size_t lds = wb * wpb;
  lds_optin(k_plan, lds);
  hipLaunchKernelGGL(k_plan, grid, block, lds, (hipStream_t)stream, *c, *s, *p, skip_done ? 1 : 0);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail(-3, hipGetErrorString(err));
  return 0;
}

int plan_reset_launch(const d2d_cfg *c, const d2d_plan *p, const uint8_t *mask, int mask_stride, void *stream) {
  if (c->B == 0) return 0;
  const dim3 grid((c->B + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), block(WAVE * WAVES_PER_BLOCK);
  hipLaunchKernelGGL(k_plan_reset, grid, block, 0, (hipStream_t)stream, *c, *p, mask, mask_stride);
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

#ifdef D2D_CHAIN_PROF
int d2d_debug_set_phase_buf(unsigned long long *buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(d2d_phase_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : -3;
}
#endif

#ifdef D2D_SEARCH_PROF
int d2d_debug_search_prof(unsigned long long *out16, int reset) {
  if (reset) {
    unsigned long long z[16] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(d2d_search_prof), z, sizeof z) == hipSuccess ? 0 : -3;
  }
  return hipMemcpyFromSymbol(out16, HIP_SYMBOL(d2d_search_prof), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -3;
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

int d2d_rollout(const d2d_cfg *c, const d2d_state *s, int32_t nsteps, const double *actions, const double *wp_steps,
                const double *pin, uint8_t *coll_out, void *stream) {
  // nsteps fused steps queued back to back on the stream (one d2d_step-sized launch each, ~40 us of GPU work
  // against ~5 us of launch cost, so a device-side step loop buys nothing and costs registers)
  int rc = check(c, s);
  if (rc) return rc;
  if (!actions || nsteps < 0) return fail(-1, "rollout: bad arguments");
  d2d_state st = *s;
  for (int32_t t = 0; t < nsteps; ++t) {
    st.action = (const double D2D_AS *)(actions + (size_t)t * c->B);
    if (wp_steps) st.wp = (const double D2D_AS *)(wp_steps + (size_t)t * c->B * 6);
    if (s->noise && c->noise_rows > 1) st.noise = s->noise + (size_t)((c->noise_row0 + t) % c->noise_rows) * c->B * c->N * 2;
    rc = launch_stages(c, &st, D2D_ST_ALL, stream, pin, coll_out ? coll_out + (size_t)t * c->B : nullptr);
    if (rc) return rc;
  }
  return 0;
}

int d2d_reset(const d2d_cfg *c, const d2d_state *s, const d2d_state *init, const uint8_t *mask, void *stream) {
  return reset_launch(c, s, init, mask, 1, stream);
}

int d2d_gaze_stage(const d2d_cfg *c, const d2d_state *s, const d2d_plan *p, void *stream) {
  int rc = plan_check(c, s, p);
  if (rc) return rc;
  return gaze_launch(c, s, p, nullptr, false, stream);
}

int d2d_plan_stage(const d2d_cfg *c, const d2d_state *s, const d2d_plan *p, void *stream) {
  int rc = plan_check(c, s, p);
  if (rc) return rc;
  return plan_launch(c, s, p, false, stream);
}

int d2d_plan_reset(const d2d_cfg *c, const d2d_plan *p, const uint8_t *mask, int32_t mask_stride, void *stream) {
  if (!c || !p) return fail(-1, "null cfg/plan");
  if (c->abi_version != D2D_ABI_VERSION) return fail(-2, "ABI version mismatch");
  return plan_reset_launch(c, p, mask, mask_stride, stream);
}

int d2d_closed_loop(const d2d_cfg *c, const d2d_state *s, const d2d_plan *p, int32_t nsteps, int32_t on_done,
                    const d2d_state *init, void *stream) {
  // [reset of the envs whose previous step ended the episode +] gaze -> perceive -> plan -> act per step
  int rc = plan_check(c, s, p);
  if (rc) return rc;
  if (nsteps < 0) return fail(-1, "closed_loop: bad step count");
  if (on_done < D2D_DONE_CONTINUE || on_done > D2D_DONE_FREEZE) return fail(-1, "closed_loop: bad on_done");
  const bool auto_reset = on_done == D2D_DONE_RESET;
  if (auto_reset && (!init || !init->agents || !init->agent_unit || !init->dyn_prev || !init->gt || !init->dmap ||
                     !init->drone || !init->target || !init->targets || !init->counters || !init->active))
    return fail(-1, "closed_loop: D2D_DONE_RESET needs the snapshot");
  if (c->B == 0 || nsteps == 0) return 0;
#ifndef D2D_NO_PERSISTENT
  if (c->planner_mode == D2D_PLANNER_EXTERNAL && p->launch_args && (p->planner == D2D_PLAN_PRIMITIVE || p->gaze == D2D_GAZE_OXFORD)) {
    // one persistent launch: every wave loops over the steps of its own env.  Specialisation as for the step kernel;
    // as many envs (waves) per workgroup as the largest phase's LDS working set allows
    const int spec = !spec_path(*c) ? 0 : (c->N <= spec_ncap(1) ? 1 : (c->N <= spec_ncap(2) ? 2 : 3));
    auto bytes = [&](int wpb) {
      return spec == 0 ? closed_wave_bytes<0>(*c, *p, wpb) : spec == 1 ? closed_wave_bytes<1>(*c, *p, wpb)
           : spec == 2 ? closed_wave_bytes<2>(*c, *p, wpb) : closed_wave_bytes<3>(*c, *p, wpb);
    };
    // ONE wave (env) per workgroup: the waves of a workgroup never talk to each other, but a workgroup holds its wave slots and its
    // LDS until its SLOWEST wave has finished -- and the chains of a launch differ widely (a mean chain is 0.3-0.6 of the longest:
    // searches, resets).  With four envs per workgroup, three slots of every workgroup with a heavy env idle until it ends; with one
    // they go to the next env at once.  Same-call, 600 / 300: config 4 (32 768 envs) 7.98e7 -> 9.35e7, config 5 2.43e7 -> 2.74e7,
    // config 2 at 65 536 envs 1.25e8 -> 1.32e8, at 16 384 1.08e8 -> 1.14e8; at 4096 envs (one round: every env has its slot from
    // the start) 8.44e7 = 8.42e7.  (A k_stages launch does the same work in every wave and keeps four.)
    const int wpb = (size_t)bytes(1) <= LDS_HARD ? 1 : 0;
    if (wpb >= 1) {
      const dim3 grid((c->B + wpb - 1) / wpb), block(WAVE * wpb);
      const size_t lds = (size_t)bytes(wpb) * wpb;
      ClosedArgs *dev = (ClosedArgs *)p->launch_args;
      hipLaunchKernelGGL(k_closed_args, dim3(1), dim3(64), 0, (hipStream_t)stream, dev, *c, *s, *p, auto_reset ? *init : *s,
                         (int)on_done, (int)nsteps);
      if (spec == 0) lds_optin(k_closed<0>, lds), lds_optin(k_closed<4>, lds);
      if (spec == 3) lds_optin(k_closed<3>, lds);
      switch (spec) {
        case 0:
          if (c->grid_tile) hipLaunchKernelGGL(k_closed<4>, grid, block, lds, (hipStream_t)stream, (const ClosedArgs *)dev);
          else hipLaunchKernelGGL(k_closed<0>, grid, block, lds, (hipStream_t)stream, (const ClosedArgs *)dev);
          break;
        case 1: hipLaunchKernelGGL(k_closed<1>, grid, block, lds, (hipStream_t)stream, (const ClosedArgs *)dev); break;
        case 2: hipLaunchKernelGGL(k_closed<2>, grid, block, lds, (hipStream_t)stream, (const ClosedArgs *)dev); break;
        default: hipLaunchKernelGGL(k_closed<3>, grid, block, lds, (hipStream_t)stream, (const ClosedArgs *)dev); break;
      }
      hipError_t err = hipGetLastError();
      if (err != hipSuccess) return fail(-3, hipGetErrorString(err));
      return 0;
    }
  }
#endif
  // any other configuration: one launch per stage per step (4 per step, 2 when the planner stage is not on the device)
  const bool split = p->planner == D2D_PLAN_PRIMITIVE;
  const uint32_t skip = on_done == D2D_DONE_FREEZE ? D2D_ST_SKIP_DONE : 0;
  for (int32_t t = 0; t < nsteps; ++t) {
    if ((rc = gaze_launch(c, s, p, auto_reset ? init : nullptr, skip != 0, stream))) return rc;
    d2d_state sn = *s;  // this step's row of the measurement noise
    if (s->noise && c->noise_rows > 1) sn.noise = s->noise + (size_t)((c->noise_row0 + t) % c->noise_rows) * c->B * c->N * 2;
    if (split) {
      if ((rc = launch_stages(c, &sn, D2D_ST_PERCEIVE | skip, stream))) return rc;
      if ((rc = plan_launch(c, s, p, skip != 0, stream))) return rc;
      if ((rc = launch_stages(c, s, D2D_ST_ACT | skip, stream))) return rc;
    } else if ((rc = launch_stages(c, &sn, D2D_ST_ALL | skip, stream))) {
      return rc;
    }
  }
  return 0;
}

int d2d_launch_shape(const d2d_cfg *c, const d2d_plan *p, int32_t out[4]) {
  if (!c || !out) return fail(-1, "launch_shape: null argument");
  if (c->abi_version != D2D_ABI_VERSION) return fail(-2, "ABI version mismatch");
  const int spec = !spec_path(*c) ? 0 : (c->N <= spec_ncap(1) ? 1 : (c->N <= spec_ncap(2) ? 2 : 3));
  const bool full = spec_full(spec);
  int wpb = (spec == 1 || spec == 2) ? WAVES_PER_BLOCK : pick_wpb(*c);
  size_t wb = 0;
  if (!p) {
    wb = wpb ? (size_t)make_geom(*c, wpb, spec_ncap(spec), full).wave_bytes : 0;
  } else {
    auto bytes = [&](int w) {
      return spec == 0 ? closed_wave_bytes<0>(*c, *p, w) : spec == 1 ? closed_wave_bytes<1>(*c, *p, w)
           : spec == 2 ? closed_wave_bytes<2>(*c, *p, w) : closed_wave_bytes<3>(*c, *p, w);
    };
    wpb = (size_t)bytes(1) <= LDS_HARD ? 1 : 0;   // the persistent kernel runs one env per workgroup (d2d_closed_loop)
    wb = wpb >= 1 ? (size_t)bytes(wpb) : 0;
    if (!p->launch_args || c->planner_mode != D2D_PLANNER_EXTERNAL) wpb = 0;
  }
  out[0] = wpb;
  out[1] = (int32_t)(wb * (size_t)(wpb > 0 ? wpb : 0));
  out[2] = out[1] > 0 ? (int32_t)(LDS_HARD / (size_t)out[1]) : 0;
  out[3] = full ? 1 : 0;
  return 0;
}

int d2d_sincos_array(const double *in, double *so, double *co, int64_t n, void *stream) {
  if (n < 0 || (n > 0 && (!in || !so || !co))) return fail(-1, "sincos_array: bad arguments");
  if (n == 0) return 0;
  const int bs = 256;
  hipLaunchKernelGGL(k_sincos, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, (hipStream_t)stream, in, so, co,
                     (long long)n);
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
