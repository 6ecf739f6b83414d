// d2d_plugins.h — the reference's planner / gaze plugins as device stages (include/d2d.h d2d_plan; SURVEY
// section 8 rows f2, f3).  Included by d2d_hip.hip inside its anonymous namespace.
//
//   k_plan   Primitive.replan_check + Primitive.plan (traj_planner.py:125-233) + the head waypoint step_pos consumes
//   k_gaze   Oxford.plan (yaw_planner.py:81-127)
//
// Mapping: ONE WAVEFRONT PER ENV, like the fused step: envs never talk to each other, every hand-off is lane ->
// lane inside one wave (LDS, or the env's own global scratch behind wave_sync_global()).
//   search            lane = motion primitive (8 x 8 accelerations = one wave); the open set is an append-only node
//                     array in the env's scratch (Python dict order = insertion order = slot order) with the costs of
//                     the first 512 nodes mirrored in LDS (closed = +inf): min() is a scan of that mirror + a
//                     lexicographic (cost, slot) reduction 64 -> 8 -> 1 through LDS; the dict lookup is an
//                     open-addressing hash table; the (primitive, collision sample) pairs of the few primitives that
//                     pass the speed limit are spread over the lanes, their probes read an LDS copy of the explored
//                     map; the successors of one expansion are de-duplicated among the lanes (ballot + LDS) and
//                     inserted together, which gives the same final dict as the reference's one-by-one loop
//   replan_check      lane = trajectory waypoint, __any over the wave
//   view maps         lane = cell of the 2 * depth bounding box around the viewpoint (cells outside cannot be seen);
//                     for the six candidates only the cells inside the map and the view disk, compacted by ballot
//   np.sum            numpy's pairwise order: lane = (candidate, accumulator 0..7) walks the box cells of one
//                     <= 128-element block, ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) by three shuffles, the blocks'
//                     sums added level by level of the recursion tree (host table pw_tree)
// What bounds both stages is a chain of short dependent steps of ONE wave, so the code is written
// for latency: loads issued together before their first use, selects instead of short-circuit conditions (a branch
// per LDS read makes the wave wait for every read on its own), constant tables staged in LDS.
// Arithmetic is fp64 in the reference's operation order with numpy's measured roundings (oracle/d2d_oracle.c lists
// them); sin / cos are csrc/d2d_sincos.h, arccos(q) <= half_fov is the host's decision window.

#define D2D_SINCOS_QUAL __device__ __forceinline__
#define D2D_SINCOS_TBL_QUAL __device__ const
#include "d2d_sincos.h"

__device__ __forceinline__ double norm2(double x, double y) { return sqrt(__builtin_fma(y, y, x * x)); }

__device__ __forceinline__ double shfl_f64(double v, int src) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl(lo, src, WAVE);
  hi = __shfl(hi, src, WAVE);
  return __hiloint2double(hi, lo);
}

// Value of lane + N (N = 1, 2, 4) inside the lane's row of 16: a DPP row shift, no LDS round trip.  Only used where
// the source lane lies in the same row (groups of 8 neighbouring lanes).
template <int N>
__device__ __forceinline__ double row_shl_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x100 + N, 0xf, 0xf, false);  // row_shl:N
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x100 + N, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// Value of lane - N inside the lane's row of 16 (DPP row_shr:N); lanes without a source keep their own value.
template <int N>
__device__ __forceinline__ int row_shr_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x110 + N, 0xf, 0xf, false); }
template <int N>
__device__ __forceinline__ double row_shr_f64(double v) {
  return __hiloint2double(row_shr_i32<N>(__double2hiint(v)), row_shr_i32<N>(__double2loint(v)));
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// Lexicographic (cost, slot) minimum over the wave, returned wave-uniform, in two plain reductions: the minimum cost
// (never NaN: a lane's candidate starts at +inf and is only replaced through `<`), then the smallest slot among the lanes
// that hold it.  Four DPP steps fold every row of 16 lanes into its last lane, readlanes fold the rows.
// (+inf, INT_MAX) is "no candidate".
__device__ __forceinline__ int wave_argmin(double t, int idx) {
  double m = t;
  m = __builtin_fmin(m, row_shr_f64<1>(m));
  m = __builtin_fmin(m, row_shr_f64<2>(m));
  m = __builtin_fmin(m, row_shr_f64<4>(m));
  m = __builtin_fmin(m, row_shr_f64<8>(m));
  const double wmin = __builtin_fmin(__builtin_fmin(readlane_f64(m, 15), readlane_f64(m, 31)),
                                     __builtin_fmin(readlane_f64(m, 47), readlane_f64(m, 63)));
  // one lane holds the minimum (ties between total costs are rare): its slot is a lane read, no second reduction
  const unsigned long long eq = __ballot(t == wmin);
  if (__popcll(eq) == 1) return __builtin_amdgcn_readlane(idx, __ffsll((long long)eq) - 1);
  int c = (t == wmin) ? idx : 0x7fffffff;
  c = min(c, row_shr_i32<1>(c));
  c = min(c, row_shr_i32<2>(c));
  c = min(c, row_shr_i32<4>(c));
  c = min(c, row_shr_i32<8>(c));
  return min(min(__builtin_amdgcn_readlane(c, 15), __builtin_amdgcn_readlane(c, 31)),
             min(__builtin_amdgcn_readlane(c, 47), __builtin_amdgcn_readlane(c, 63)));
}

// Minimum / maximum over the wave, returned wave-uniform (DPP row shifts + four lane reads, no LDS round trips; never NaN inputs)
__device__ __forceinline__ double wave_fmin(double m) {
  m = __builtin_fmin(m, row_shr_f64<1>(m));
  m = __builtin_fmin(m, row_shr_f64<2>(m));
  m = __builtin_fmin(m, row_shr_f64<4>(m));
  m = __builtin_fmin(m, row_shr_f64<8>(m));
  return __builtin_fmin(__builtin_fmin(readlane_f64(m, 15), readlane_f64(m, 31)), __builtin_fmin(readlane_f64(m, 47), readlane_f64(m, 63)));
}
__device__ __forceinline__ double wave_fmax(double m) {
  m = __builtin_fmax(m, row_shr_f64<1>(m));
  m = __builtin_fmax(m, row_shr_f64<2>(m));
  m = __builtin_fmax(m, row_shr_f64<4>(m));
  m = __builtin_fmax(m, row_shr_f64<8>(m));
  return __builtin_fmax(__builtin_fmax(readlane_f64(m, 15), readlane_f64(m, 31)), __builtin_fmax(readlane_f64(m, 47), readlane_f64(m, 63)));
}

struct TrkView {  // active trackers of the env, compacted into LDS
  double *mx, *my, *vx, *vy;
  double *lim;  // norm(d) <= L rewritten as d.d <= T(L), see sq_threshold: replan_check's limits, then (when a search follows) plan's
  int n;
};

// The largest double s with sqrt(s) <= L (sqrt correctly rounded, hence monotone): `norm(d) <= L` with numpy's
// norm = sqrt(fma(dy, dy, dx * dx)) is exactly `fma(dy, dy, dx * dx) <= sq_threshold(L)`, without the square root in
// the inner loops.  L * L is within a few ulp of the answer; the walk is bounded.
__device__ __forceinline__ double sq_threshold(double L) {
  if (!(L >= 0.0)) return -1.0;  // nothing is <= a negative or NaN limit
  double t = L * L;
  for (int k = 0; k < 8 && sqrt(t) > L; ++k) t = __longlong_as_double(__double_as_longlong(t) - 1);
  for (int k = 0; k < 8; ++k) {
    const double u = __longlong_as_double(__double_as_longlong(t) + 1);
    if (!(sqrt(u) <= L)) break;
    t = u;
  }
  return t;
}

// The tracker part of Planner.is_free (traj_planner.py:52-58): is (x, y) at time t within the safety radius of an active
// tracker's predicted position?  Four trackers per round, loads first: one tracker per iteration pays the LDS latency
// once per tracker (the trip count is dynamic, the compiler does not pipeline it).  The planes are padded to a multiple
// of four entries.
template <int WIDTH>
__device__ __forceinline__ bool plan_hits_round(const TrkView &T, int q, double x, double y, double t) {
  double mx[WIDTH], my[WIDTH], vx[WIDTH], vy[WIDTH], lim[WIDTH];
#pragma unroll
  for (int u = 0; u < WIDTH; ++u) {
    mx[u] = T.mx[q + u]; my[u] = T.my[q + u]; vx[u] = T.vx[q + u]; vy[u] = T.vy[q + u]; lim[u] = T.lim[q + u];
  }
  bool hit = false;
#pragma unroll
  for (int u = 0; u < WIDTH; ++u) {
    const double ex = mx[u] + t * vx[u], ey = my[u] + t * vy[u];  // estimate_pos, utils.py:220-223
    const double dx = x - ex, dy = y - ey;
    hit = hit | ((q + u < T.n) & (__builtin_fma(dy, dy, dx * dx) <= lim[u]));
  }
  return hit;
}

__device__ __forceinline__ bool plan_hits_tracker(const TrkView &T, double x, double y, double t) {
  if (T.n <= 2) return T.n > 0 ? plan_hits_round<2>(T, 0, x, y, t) : false;  // the usual case: one or two active trackers
  bool hit = false;
  for (int q = 0; q < T.n; q += 4) hit = hit | plan_hits_round<4>(T, q, x, y, t);
  return hit;
}

// Planner.is_free, traj_planner.py:28-59.  The five probes (x -+ d, y), (x, y), (x, y -+ d) of get_grid
// (utils.py:545-548) share three column and three row indices; loads are unconditional (clamped) and in flight together.
template <typename DM>
__device__ __forceinline__ bool plan_is_free(const d2d_cfg &c, const d2d_plan &p, DM dm,
                                             const TrkView &T, double x, double y, double t, double inv_scale) {
  if (x != x || y != y) return false;
  const double d = p.safe_dist;
  const double xl = x - d, xr = x + d, yl = y - d, yr = y + d;
  const int W1 = c.W - 1, H1 = c.H - 1;
  const int i0 = min(max(cell_fast(xl, c.scale, inv_scale), 0), W1), i1 = min(max(cell_fast(x, c.scale, inv_scale), 0), W1),
            i2 = min(max(cell_fast(xr, c.scale, inv_scale), 0), W1);
  const int j0 = min(max(cell_fast(yl, c.scale, inv_scale), 0), H1), j1 = min(max(cell_fast(y, c.scale, inv_scale), 0), H1),
            j2 = min(max(cell_fast(yr, c.scale, inv_scale), 0), H1);
  const unsigned char v0 = dm[grid_ix(c, i0, j1)], v1 = dm[grid_ix(c, i1, j1)], v2 = dm[grid_ix(c, i2, j1)], v3 = dm[grid_ix(c, i1, j0)],
                      v4 = dm[grid_ix(c, i1, j2)];
  // out of the map = wall (get_grid): x >= W_px or x < 0 or y >= H_px or y < 0, per probe
  const bool ox0 = (xl >= c.W_px) | (xl < 0.0), ox1 = (x >= c.W_px) | (x < 0.0), ox2 = (xr >= c.W_px) | (xr < 0.0);
  const bool oy0 = (yl >= c.H_px) | (yl < 0.0), oy1 = (y >= c.H_px) | (y < 0.0), oy2 = (yr >= c.H_px) | (yr < 0.0);
  const bool wall = (ox0 | oy1 | (v0 == D2D_OCCUPIED)) | (ox1 | oy1 | (v1 == D2D_OCCUPIED)) | (ox2 | oy1 | (v2 == D2D_OCCUPIED)) |
                    (ox1 | oy0 | (v3 == D2D_OCCUPIED)) | (ox1 | oy2 | (v4 == D2D_OCCUPIED));
  return !wall && !plan_hits_tracker(T, x, y, t);
}

// The wall part of is_free for the planner's collision samples on the default grid scale: the samples are integer-valued
// (np.around), the safety distance is an integer, so the five probes are integer arithmetic -- two conversions, then
// 24-bit multiplies for floor(v / 10) == (v * 52429) >> 19 (exact below 81920) -- instead of six fp64 floor divisions.
// `xi`, `yi`: the sample, already known to lie within +-2^22; `di`: the safety distance; Wpx / Hpx: the map in pixels.
template <typename DM>
__device__ __forceinline__ bool plan_wall_int(const d2d_cfg &c, DM dm, int xi, int yi, int di, int Wpx, int Hpx) {
  const int xl = xi - di, xr = xi + di, yl = yi - di, yr = yi + di;
  // out of the map = wall: v < 0 or v >= size, one unsigned compare
  const bool ox0 = (unsigned int)xl >= (unsigned int)Wpx, ox1 = (unsigned int)xi >= (unsigned int)Wpx, ox2 = (unsigned int)xr >= (unsigned int)Wpx;
  const bool oy0 = (unsigned int)yl >= (unsigned int)Hpx, oy1 = (unsigned int)yi >= (unsigned int)Hpx, oy2 = (unsigned int)yr >= (unsigned int)Hpx;
  const int Wm = Wpx - 1, Hm = Hpx - 1;
#define D2D_BY10(v, m) ((int)(__umul24((unsigned int)min(max((v), 0), (m)), 52429u) >> 19))
  const int W1 = c.W - 1, H1 = c.H - 1;  // an out-of-range probe is a wall whatever it reads: only its address must be valid
  const int i0 = min(D2D_BY10(xl, Wm), W1), i1 = min(D2D_BY10(xi, Wm), W1), i2 = min(D2D_BY10(xr, Wm), W1);
  const int j0 = min(D2D_BY10(yl, Hm), H1), j1 = min(D2D_BY10(yi, Hm), H1), j2 = min(D2D_BY10(yr, Hm), H1);
#undef D2D_BY10
  const unsigned char v0 = dm[grid_ix(c, i0, j1)], v1 = dm[grid_ix(c, i1, j1)], v2 = dm[grid_ix(c, i2, j1)], v3 = dm[grid_ix(c, i1, j0)],
                      v4 = dm[grid_ix(c, i1, j2)];
  return (ox0 | oy1 | (v0 == D2D_OCCUPIED)) | (ox1 | oy1 | (v1 == D2D_OCCUPIED)) | (ox2 | oy1 | (v2 == D2D_OCCUPIED)) |
         (ox1 | oy0 | (v3 == D2D_OCCUPIED)) | (ox1 | oy2 | (v4 == D2D_OCCUPIED));
}


// Primitive_Node.get_index, traj_planner.py:93: (round(x) // 10, round(y) // 10, round(vx), round(vy)); the floor
// division of the rounded coordinate is exact in fp64 (cell_fast), no 64-bit integer division
__device__ __forceinline__ long long node_key(double px, double py, double vx, double vy) {
  const int a = cell_fast(rint(px), 10.0, 0.1), b = cell_fast(rint(py), 10.0, 0.1);
  const int cc = (int)rint(vx), d = (int)rint(vy);
  const unsigned int hi = ((unsigned int)(a + 32768) << 16) | ((unsigned int)(b + 32768) & 0xffffu);
  const unsigned int lo = ((unsigned int)(cc + 32768) << 16) | ((unsigned int)(d + 32768) & 0xffffu);
  return (long long)(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ unsigned int key_hash(long long k) {
  return (unsigned int)(((unsigned long long)k * 0x9E3779B97F4A7C15ull) >> 32);
}

// Scratch layout of one env's search (private to this file): ONE 96-byte record per node (D2D_NODE_F doubles, 16-byte aligned)
//   +0 position(2)   +2 velocity(2)   +4 cost, key   +6 meta (int32 parent slot | int32 itr << 2 | state: 1 open, 2 closed), total_cost
//   +8 acceleration(2)   +10 unused
// so that a pop reads its node with four loads from one base pointer, a dict hit with two, and an insert writes five 16-byte
// pieces (round 2 kept eleven planes: eleven base pointers -- scalar registers the loop did not have -- and one store per field).
#define NR_POS 0
#define NR_VEL 2
#define NR_COST 4   // (cost, key)
#define NR_META 6   // (meta, total_cost)
#define NR_ACC 8
struct NodeRecs {
  double *b;
  __device__ __forceinline__ double *rec(int i) const { return b + (size_t)i * D2D_NODE_F; }
};
__device__ __forceinline__ double2 ld2(const double *q) { return *(const double2 *)q; }
__device__ __forceinline__ void st2(double *q, double a, double b) { *(double2 *)q = make_double2(a, b); }
__device__ __forceinline__ double meta_pack(int parent, int itr, int state) { return __hiloint2double((itr << 2) | state, parent); }
__device__ __forceinline__ int meta_parent(double m) { return __double2loint(m); }
__device__ __forceinline__ int meta_itr(double m) { return __double2hiint(m) >> 2; }
__device__ __forceinline__ int meta_state(double m) { return __double2hiint(m) & 3; }

// chunks a remaining trajectory has to span before the every-step walks consult the boxes
#ifndef D2D_TRAJ_PRUNE_CHUNKS
#define D2D_TRAJ_PRUNE_CHUNKS 4
#endif
// d2d_plan.traj_box of env e, or null when the caller gave none (then every walk visits every chunk)
__device__ __forceinline__ double *traj_boxes(const d2d_plan &p, int e) {
  return p.traj_box ? (double *)p.traj_box + (size_t)e * ((p.traj_cap + WAVE - 1) / WAVE) * 4 : nullptr;
}

// The dict of a search in LDS: open addressing over LH_N 16-bit entries, entry = 5 fingerprint bits of the key's hash << 11 |
// slot + 1 (0 = empty).  A key that is not in the dict -- most successors -- is settled without leaving LDS; a fingerprint match
// is confirmed against the node's key in its record (the same load brings its cost and state).  Holds LH_MAX keys / slots below
// 2047; a search that outgrows it (large primitive sets) moves its dict to the hash table in global memory, once, and goes on there.
#define LH_N 1024
#define LH_MAX 896

#ifdef D2D_SEARCH_PROF
// Diagnostic build only (tools/search_prof.py): shader-clock time of env 0's search per section of the expansion loop
__device__ unsigned long long d2d_search_prof[16];
#define SP_T(var)                                              \
  do {                                                         \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
    var = __builtin_amdgcn_s_memtime();                        \
  } while (0)
// sections accumulate in (scalar) registers and reach memory once per search: a read-modify-write of the global
// counters per section would put an L2 round trip into every measured interval
#define SP_ADD(idx, t0, t1) \
  do {                      \
    spacc[idx] += (t1) - (t0); \
  } while (0)
#define SP_FLUSH()                                                            \
  do {                                                                        \
    if (e == 0 && lane == 0)                                                  \
      for (int spi = 0; spi < 16; ++spi) d2d_search_prof[spi] += spacc[spi]; \
  } while (0)
#else
#define SP_T(var) do { } while (0)
#define SP_ADD(idx, t0, t1) do { } while (0)
#define SP_FLUSH() do { } while (0)
#endif

struct SearchLds {
  int *chain;       // [128] path slots; during the search: the 64-bucket table of the successors' keys (de-duplication)
  double *us;       // [nu] u_space
  double *st;       // [n_sample][2] t, t**2
  double *pc;       // [nu * nu] (x_acc**2 + y_acc**2) / 100 of every primitive (traj_planner.py:184): one division per search, not per expansion
  double *rv;       // [64] accelerations x of the primitives that pass the speed limit, by rank; de-duplication values (candidate costs)
  long long *rk;    // [64] ... accelerations y; de-duplication keys; occupancy rows while the wall rows are built
  int *ri;          // [64] de-duplication lanes
  int *misc;        // [4] misc[0] = number of active trackers staged by the quick part of the stage
  double *tot;      // [ntot] total_cost of the first nodes, +inf once closed (the min() scan reads these)
  int ntot;         // search_lds_nodes(N)
  unsigned long long *prow;  // [64] wall rows: bit j of row i = "a collision sample in cell (i, j) touches a wall" (Planner.is_free's five
                             // probes folded into one), or null when the grid has more than 64 cells a side
  unsigned short *lh;        // [LH_N] the dict (see LH_N)
};

// nodes whose total_cost is mirrored in LDS: 512 (a capped search makes about 800); with more than 64 agents the six-plane
// tracker staging grows and 256 keep the planner stage within three 4-wave workgroups per CU (BASELINE config 3: 13.4 KB per wave)
__host__ __device__ inline int search_lds_nodes(int N) { return N > 64 ? 256 : 512; }
__host__ __device__ inline bool search_wall_rows(int W, int H) { return W <= 64 && H <= 64; }

// Primitive.plan's search (traj_planner.py:128-218) by one wave.  Returns the number of waypoints written, -1 = failure.
// 0 is a success: the start node is the goal node (target within the search threshold of the start), the reference
// returns True with an empty trajectory (traj_planner.py:158-160, 204-216).
__device__ int plan_search(const d2d_cfg &c, const d2d_state &s, const d2d_plan &p, int e, int lane, const TrkView &T,
                           const SearchLds &S, const unsigned char *__restrict__ dm, double inv_scale) {
  const double H = p.horizon;
  const NodeRecs nd = {p.nodes + (size_t)e * p.node_cap * D2D_NODE_F};
  int *__restrict__ tab = p.hash + (size_t)e * p.hash_cap;
  double *__restrict__ traj = p.traj + (size_t)e * p.traj_cap * 4;
  int *stat = p.plan_stat + (size_t)e * 4;
  int *chain = S.chain;
  const unsigned int hmask = (unsigned int)p.hash_cap - 1u;
  const double tx = s.target[(size_t)e * 2], ty = s.target[(size_t)e * 2 + 1];
  const double *dr = s.drone + (size_t)e * D2D_DF;

  unsigned long long spa = 0, spb = 0;
  (void)spa; (void)spb;
#ifdef D2D_SEARCH_PROF
  unsigned long long spacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  SP_T(spa);
  for (int i = lane; i < p.nu; i += WAVE) S.us[i] = p.u_space[i];
  {
    const FastDiv fdu(p.nu);
    for (int i = lane; i < p.nu * p.nu; i += WAVE) {
      int ia, ja;
      fdu.divmod(i, ia, ja);
      const double ax = p.u_space[ia], ay = p.u_space[ja];
      S.pc[i] = (ax * ax + ay * ay) / 100;  // traj_planner.py:184
    }
  }
  for (int i = lane; i < 2 * p.n_sample; i += WAVE) S.st[i] = p.sample_t[i];
  {
    uint4 *lh128 = (uint4 *)S.lh;  // (16-byte aligned: plan_carve)
    for (int i = lane; i < LH_N / 8; i += WAVE) lh128[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  // integer probes for the collision samples: grid scale 10, map below 81920 px, integer safety distance (the samples are
  // integer-valued: np.around)
  const bool int_walls = c.scale == 10.0 && c.W_px <= 81919.0 && c.H_px <= 81919.0 && c.W_px == floor(c.W_px) &&
                         c.H_px == floor(c.H_px) && c.W_px >= 1.0 && c.H_px >= 1.0 && p.safe_dist >= 0.0 &&
                         p.safe_dist <= 1048576.0 && p.safe_dist == floor(p.safe_dist);
  const int safe_i = int_walls ? (int)p.safe_dist : 0, wpx_i = int_walls ? (int)c.W_px : 1, hpx_i = int_walls ? (int)c.H_px : 1;
  // The five probes of Planner.is_free (traj_planner.py:33-50: (x -+ d, y), (x, y), (x, y -+ d), out of the map = wall) for an
  // integer sample, a safety distance of k whole cells and a map of whole cells: the probes are the cells (i -+ k, j), (i, j),
  // (i, j -+ k) of the sample's cell (i, j), a probe leaves the map iff i < k, i >= W - k, j < k or j >= H - k.  Their OR is one bit
  // of a row word built once per search from the explored map: row i = occ[i] | occ[i - k] | occ[i + k] | occ[i] << k | occ[i] >> k |
  // border.  One LDS read per sample instead of five and a fifth of the arithmetic.
  const int kcell = safe_i / 10;
  const bool rows = S.prow != nullptr && int_walls && c.W <= 64 && c.H <= 64 && c.W_px == 10.0 * (double)c.W &&
                    c.H_px == 10.0 * (double)c.H && safe_i == 10 * kcell && 2 * kcell <= min(c.W, c.H);
  if (rows) {
    int olo = 0, ohi = 0;  // lane i: the occupancy bits of grid row i (bit j = explored map holds OCCUPIED at (i, j))
    unsigned long long *orow = (unsigned long long *)S.rk;
    const int nb = c.W * c.H;
    if (c.grid_tile == 0 && (nb & 3) == 0 && (((size_t)dm) & 3) == 0) {
      // row-major map, dword aligned: ten dwords per lane in flight (one round trip for a 50 x 50 map), the few OCCUPIED bytes found by
      // an exact zero-byte test on x ^ 0x01010101 and ORed into their rows in LDS -- most dwords hold none and cost six instructions
      orow[lane] = 0ull;
      const unsigned int *src = (const unsigned int *)dm;
      const FastDiv fdh(c.H);
      for (int d0 = 0; d0 < nb / 4; d0 += 10 * WAVE) {
        unsigned int x[10];
#pragma unroll
        for (int u = 0; u < 10; ++u) {
          const int idx = d0 + u * WAVE + lane;
          x[u] = src[min(idx, nb / 4 - 1)];
        }
#pragma unroll
        for (int u = 0; u < 10; ++u) {
          const int idx = d0 + u * WAVE + lane;
          const unsigned int y = x[u] ^ 0x01010101u;  // a zero byte where the cell holds OCCUPIED (1)
          unsigned int z = ~(((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y | 0x7f7f7f7fu);  // bit 7 of exactly the zero bytes
          if (idx >= nb / 4) z = 0u;
          while (z) {
            const int b = (__ffs((int)z) - 1) >> 3;
            z &= z - 1u;
            int i, j;
            fdh.divmod_big(4 * idx + b, i, j);
            atomicOr(&orow[i], 1ull << j);
          }
        }
      }
      wave_sync_lds();
      const unsigned long long o = orow[lane];
      olo = (int)(unsigned int)o;
      ohi = (int)(unsigned int)(o >> 32);
      wave_sync_lds();
    } else {
      const int jc = min(lane, c.H - 1);
      for (int i0 = 0; i0 < c.W; i0 += 8) {
        unsigned char v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = dm[grid_ix(c, min(i0 + u, c.W - 1), jc)];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (i0 + u < c.W) {  // wave-uniform
            const unsigned long long m = __ballot(lane < c.H && v[u] == D2D_OCCUPIED);
            olo = lane == i0 + u ? (int)(unsigned int)m : olo;
            ohi = lane == i0 + u ? (int)(unsigned int)(m >> 32) : ohi;
          }
        }
      }
    }
    const unsigned long long occ = ((unsigned long long)(unsigned int)ohi << 32) | (unsigned int)olo;
    orow[lane] = occ;
    wave_sync_lds();
    const unsigned long long up = orow[max(lane - kcell, 0)], dn = orow[min(lane + kcell, WAVE - 1)];
    const unsigned long long edge = kcell > 0 ? ((1ull << kcell) - 1ull) : 0ull;
    const unsigned long long edge_hi = kcell > 0 ? (edge << (c.H - kcell)) : 0ull;
    unsigned long long pr = occ | up | dn | (occ << kcell) | (occ >> kcell) | edge | edge_hi;
    if (lane < kcell || lane >= c.W - kcell) pr = ~0ull;
    wave_sync_lds();  // the occupancy rows have been read: rk is free again
    S.prow[lane] = pr;
  }
  const double kInf = __longlong_as_double(0x7ff0000000000000ll);
  // norm(v_end) < vmax  <=>  v.v <= (largest s with sqrt(s) < vmax)  -- sq_threshold of the double below vmax;
  // norm(p - target) <= goal_tol  <=>  d.d <= sq_threshold(goal_tol): no square root per expansion
  // (the host hands both over, d2d_plan.vmax_sq / goal_sq; a caller that leaves them 0 pays four square roots per search)
  const double vmax2 = p.vmax_sq != 0.0 ? p.vmax_sq : (p.vmax > 0.0 ? sq_threshold(__longlong_as_double(__double_as_longlong(p.vmax) - 1)) : -1.0);
  const double goal2 = p.goal_sq != 0.0 ? p.goal_sq : sq_threshold(p.goal_tol);
  wave_sync_lds();
  if (lane == 0) {
    const double x = dr[D2D_D_X], y = dr[D2D_D_Y], vx = dr[D2D_D_VX], vy = dr[D2D_D_VY];
    const double tot0 = 0.0 + 0.5 * norm2(x - tx, y - ty) + 0.1 * norm2(vx, vy);  // traj_planner.py:88
    const long long k = node_key(x, y, vx, vy);
    double *r0 = nd.rec(0);
    st2(r0 + NR_POS, x, y);
    st2(r0 + NR_VEL, vx, vy);
    st2(r0 + NR_COST, 0.0, __longlong_as_double(k));
    st2(r0 + NR_META, meta_pack(-1, 0, 1), tot0);
    st2(r0 + NR_ACC, 0.0, 0.0);
    const unsigned int h32 = key_hash(k);
    S.lh[h32 >> 22] = (unsigned short)((((h32 >> 17) & 0x1fu) << 11) | 1u);
    S.tot[0] = tot0;
  }
  wave_sync_global();
  SP_T(spb);
  SP_ADD(10, spa, spb);
  // Two successors of ONE node differ in x_acc or y_acc, hence by 2 H |delta acc| / 2 >= H * (spacing of u_space) in an end velocity,
  // hence -- once that exceeds 1 (+ slack for the roundings) -- in round(v): their dict keys differ, and the successors of an
  // expansion need no de-duplication among themselves.  (u_space = arange(-a, a, 0.4 vmax - 5) or step 4: always, for drone speeds
  // of 14 and more.)
  bool nodup;
  double amax = 0.0;  // the largest |acceleration| (for the reach bound below), from the staged copy like the spacing: lane reads +
                      // DPP reductions instead of one dependent uniform load per acceleration (eight round trips per search)
  {
    double dmin = 1e300;
    for (int i = lane; i < p.nu; i += WAVE) {  // (staged above, behind a hand-off)
      const double ui = S.us[i];
      if (i + 1 < p.nu) dmin = fmin(dmin, S.us[i + 1] - ui);
      amax = fmax(amax, fabs(ui));
    }
    dmin = wave_fmin(dmin);
    amax = wave_fmax(amax);
    nodup = H * dmin > 1.0 + 1e-6;
  }
  int nn = 1, open_n = 1, goal = -1, itr = 0, expansions = 0;
  bool overflow = false;
  bool lds_dict = true;  // the dict lives in LDS (S.lh); false once the search has outgrown it: then in `tab`
  const int nprim = p.nu * p.nu;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const FastDiv fd_nu(p.nu), fd_ns(p.n_sample);
  // |sample| <= |p| + T |v| + T^2 |a| / 2 for every sample time T < horizon: one bound per expansion instead of a test per sample
  const double reach_a = 0.5 * H * H * amax + 2.0;
  // the pairs of one primitive never straddle two rounds of 64 when n_sample divides 64: its free samples are one bit field
  const bool ns_pow2 = (p.n_sample & (p.n_sample - 1)) == 0 && p.n_sample <= WAVE;
  const bool one_batch = false;  // (the lane's accelerations kept in registers across the search: measured, no gain -- they spill)
  double ax1 = 0.0, ay1 = 0.0;
  if (one_batch) {
    int ia, ja;
    fd_nu.divmod(lane < nprim ? lane : 0, ia, ja);
    ax1 = S.us[ia];
    ay1 = S.us[ja];
  }
  for (;;) {
    itr += 1;
    if (open_n == 0 || itr >= p.max_itr) break;
    unsigned long long sp0 = 0, sp1 = 0, sp2 = 0, sp3 = 0, sp4 = 0, sp5 = 0, sp6 = 0, sp7 = 0, sp8 = 0;
    (void)sp0; (void)sp1; (void)sp2; (void)sp3; (void)sp4; (void)sp5; (void)sp6; (void)sp7; (void)sp8;
    SP_T(sp0);
    // ---- min(open_set, key=total_cost): first minimal entry in insertion (= slot) order ----
    // Branch-free on purpose: with short-circuit conditions the compiler waits for every LDS read on its own.
    // "no candidate" is (+inf, INT_MAX), so a plain lexicographic (cost, slot) compare needs no validity tests.
    double best = kInf;
    int bidx = 0x7fffffff;
    bool fenced = false;
    {
      // the first S.ntot nodes from their LDS mirror (closed = +inf; an open node with an infinite or
      // NaN cost is told apart by the state in its record below, which such a search then falls back to)
      const int nl = min(nn, S.ntot);
      for (int s0 = 0; s0 < nl; s0 += 4 * WAVE) {
        double t4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t4[u] = S.tot[min(s0 + u * WAVE + lane, S.ntot - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int si = s0 + u * WAVE + lane;
          const bool take = (si < nl) & (t4[u] < best);  // ascending slots per lane: strict < keeps the earliest
          best = take ? t4[u] : best;
          bidx = take ? si : bidx;
        }
      }
      // The stores of the previous expansion (node records) are drained HERE, not at its end: the scan of the LDS
      // mirror above does not need them, so it runs while they are in flight.
      if (nn > nl) {
        wave_sync_global();
        fenced = true;
      }
      for (int s0 = nl; s0 < nn; s0 += WAVE) {  // beyond the mirror: state and cost come together
        const int si = s0 + lane;
        const double2 mt = ld2(nd.rec(min(si, nn - 1)) + NR_META);
        const bool take = (si < nn) & (meta_state(mt.x) == 1) & (mt.y < best);
        best = take ? mt.y : best;
        bidx = take ? si : bidx;
      }
    }
    bidx = wave_argmin(best, bidx);  // first minimal entry in slot order
    if (!fenced) wave_sync_global();  // before the popped node's record and the records behind dict hits are read
    if (__builtin_amdgcn_readfirstlane(bidx) == 0x7fffffff) {
      // every open node has a non-finite cost (wild inputs): the literal scan over the states decides
      int fb = 0x7fffffff;
      for (int s0 = 0; s0 < nn; s0 += WAVE) {
        const int si = s0 + lane;
        if (si < nn && fb == 0x7fffffff && meta_state(nd.rec(si)[NR_META]) == 1) fb = si;
      }
      for (int o = 32; o > 0; o >>= 1) fb = min(fb, __shfl_xor(fb, o, WAVE));
      bidx = fb;
    }
    const int cur = __builtin_amdgcn_readfirstlane(bidx);
    SP_T(sp1);
    SP_ADD(0, sp0, sp1);
    const double *rc = nd.rec(cur);
    const double2 cpos = ld2(rc + NR_POS), cvel = ld2(rc + NR_VEL);
    const double ccost = rc[NR_COST];
    const int citr = meta_itr(rc[NR_META]);
    const double px = cpos.x, py = cpos.y, vx = cvel.x, vy = cvel.y;
    // every collision sample of this expansion fits an int (also false for NaN / infinite state)
    const bool fits_all = int_walls && (fabs(px) + H * fabs(vx) + reach_a <= 4194304.0) && (fabs(py) + H * fabs(vy) + reach_a <= 4194304.0);
    SP_T(sp2);
    SP_ADD(1, sp1, sp2);
    if (__builtin_fma(py - ty, py - ty, (px - tx) * (px - tx)) <= goal2) {  // :158
      goal = cur;
      break;
    }
    if (lane == 0) {
      ((int *)(nd.rec(cur) + NR_META))[1] = (citr << 2) | 2;
      if (cur < S.ntot) S.tot[cur] = kInf;
    }
    open_n -= 1;
    expansions += 1;
    // ---- expand: lane = primitive, batches of 64 in generation order (x_acc outer, y_acc inner) ----
    for (int p0 = 0; p0 < nprim && !overflow; p0 += WAVE) {
      const int pi = p0 + lane;
      bool ok = pi < nprim;
      double ax, ay;
      if (one_batch) {  // the lane's primitive never changes: its accelerations stay in registers
        ax = ax1;
        ay = ay1;
      } else {
        int ia, ja;
        fd_nu.divmod(ok ? pi : 0, ia, ja);  // no integer divisions in the loop: ~25 instructions each
        ax = S.us[ia];
        ay = S.us[ja];
      }
      const double hx = ax / 2, hy = ay / 2;
      const double vex = vx + (2 * H) * hx, vey = vy + (2 * H) * hy;  // :172,183
      ok = ok && (__builtin_fma(vey, vey, vex * vex) <= vmax2);
      // :175-180.  Few primitives pass the speed limit (about ten of 64), so the collision samples are spread over
      // the lanes as (primitive, sample) pairs instead of one sample round per iteration: the reference's early
      // `break` only skips work, a successor needs ALL its samples free.
      const unsigned long long vm = __ballot(ok);
      const int myrank = __popcll(vm & lt_mask);
      SP_T(sp3);
      SP_ADD(2, sp2, sp3);
      {
        // the accelerations of the primitives that passed, by rank: the pairs fetch theirs with one LDS read
        const int nv = __popcll(vm);
        double *rax = S.rv, *ray = (double *)S.rk;
        if (ok) {
          rax[myrank] = hx;
          ray[myrank] = hy;
        }
        wave_sync_lds();
        const int npair = nv * p.n_sample;
        const int my_lo = myrank * p.n_sample;  // this primitive's pairs: [my_lo, my_lo + n_sample)
        int nfree = 0;
        // two rounds of pairs in flight (about ten primitives x eight samples = 80 pairs); which samples are free comes back as
        // ballots: every primitive counts the set bits of its own pairs -- no LDS counters, no hand-off
        for (int q0 = 0; q0 < npair; q0 += 2 * WAVE) {
          int pr[2], si[2];
          bool in[2], fr[2];
          double sxv[2], syv[2], tgv[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int q = q0 + u * WAVE + lane;
            in[u] = q < npair;
            fd_ns.divmod(in[u] ? q : 0, pr[u], si[u]);
          }
          double shx[2], shy[2], tt[2], tt2[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            shx[u] = rax[pr[u]]; shy[u] = ray[pr[u]]; tt[u] = S.st[2 * si[u]]; tt2[u] = S.st[2 * si[u] + 1];
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const double sx = rint(__builtin_fma(tt2[u], shx[u], px + tt[u] * vx)), sy = rint(__builtin_fma(tt2[u], shy[u], py + tt[u] * vy));
            const double tg = tt[u] + (double)citr * H;
            sxv[u] = sx; syv[u] = sy; tgv[u] = tg;
          }
          if (fits_all) {  // wave-uniform: integer probes
            // walls and trackers of both rounds without a branch between them (`&`, not `&&`): a short-circuit costs an exec-mask
            // branch per round and keeps the second round's loads from overlapping the first one's arithmetic
            bool wall[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int xi = (int)sxv[u], yi = (int)syv[u];
              if (rows) {
                // outside the map = wall whatever the row says: only the row index has to stay inside the table
                const bool inside = ((unsigned int)xi < (unsigned int)wpx_i) & ((unsigned int)yi < (unsigned int)hpx_i);
                unsigned int ci = __umul24((unsigned int)xi, 52429u) >> 19;
                ci = ci < 63u ? ci : 63u;
                const unsigned int cj = (__umul24((unsigned int)yi, 52429u) >> 19) & 63u;
                wall[u] = !inside | (((S.prow[ci] >> cj) & 1ull) != 0ull);
              } else {
                wall[u] = plan_wall_int(c, dm, xi, yi, safe_i, wpx_i, hpx_i);
              }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) fr[u] = !wall[u] & !plan_hits_tracker(T, sxv[u], syv[u], tgv[u]);
          } else {
#pragma unroll
            for (int u = 0; u < 2; ++u) fr[u] = plan_is_free(c, p, dm, T, sxv[u], syv[u], tgv[u], inv_scale);
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const unsigned long long fm = __ballot(in[u] & fr[u]);
            const int qb = q0 + u * WAVE;  // this round holds the pairs [qb, qb + 64)
            if (ns_pow2) {
              const unsigned int sh = (unsigned int)(my_lo - qb);
              if (sh < (unsigned int)WAVE) nfree = __popcll(p.n_sample >= WAVE ? fm : ((fm >> sh) & ((1ull << p.n_sample) - 1ull)));
            } else {
              const int lo = max(my_lo, qb) - qb, hi = min(my_lo + p.n_sample, qb + WAVE) - qb;
              if (hi > lo) {
                const unsigned long long w = fm >> lo;
                nfree += __popcll(hi - lo >= WAVE ? w : (w & ((1ull << (hi - lo)) - 1ull)));
              }
            }
          }
        }
        if (ok) ok = nfree == p.n_sample;
      }
      SP_T(sp4);
      SP_ADD(3, sp3, sp4);
      const double ex = rint((px + H * vx) + (H * H) * hx), ey = rint((py + H * vy) + (H * H) * hy);  // :182
      const double cost = ccost + S.pc[ok ? pi : 0] + 10;                                             // :184, the term from its table
      const long long key = node_key(ex, ey, vex, vey);
      const unsigned int h32 = key_hash(key);
      // ---- :192-202 for all successors of the batch at once ----
      int slot = -1;
      int slot_state = 0;  // state and cost of the node found, fetched in the same round trip as its key
      double slot_cost = 0.0;
      unsigned int h = 0;  // ends at the bucket with the key, or at the empty bucket a new key goes into
      if (lds_dict) {
        h = h32 >> 22;
        const unsigned int fp = (h32 >> 17) & 0x1fu;
        if (ok) {
          for (int guard = 0; guard < LH_N; ++guard) {
            const unsigned int ent = S.lh[h];
            if (ent == 0u) break;
            if ((ent >> 11) == fp) {
              const int sl = (int)(ent & 0x7ffu) - 1;
              const double *r2 = nd.rec(sl);
              const double2 ck = ld2(r2 + NR_COST);
              const double m2 = r2[NR_META];
              if (__double_as_longlong(ck.y) == key) {
                slot = sl;
                slot_state = meta_state(m2);
                slot_cost = ck.x;
                break;
              }
            }
            h = (h + 1u) & (LH_N - 1u);
          }
        }
      } else {
        h = h32 & hmask;
        if (ok) {
          for (int guard = 0; guard < p.hash_cap; ++guard) {
            const int sv = __hip_atomic_load(&tab[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // written by atomics
            if (sv == 0) break;
            const double *r2 = nd.rec(sv - 1);
            const double2 ck = ld2(r2 + NR_COST);
            const double m2 = r2[NR_META];
            if (__double_as_longlong(ck.y) == key) {
              slot = sv - 1;
              slot_state = meta_state(m2);
              slot_cost = ck.x;
              break;
            }
            h = (h + 1) & hmask;
          }
        }
      }
      SP_T(sp5);
      SP_ADD(4, sp4, sp5);
      // the successors among themselves (rank order = lane order = generation order): the first lane of a key owns its
      // place in the dict, the cheapest (earliest on ties) its value.  Repeats within one expansion are rare:
      // every successor puts its key into a 64-bucket LDS table (the path buffer, free at this point) by compare-and-swap:
      // finding its own key there = a repeat; another key = next bucket.  One or two LDS round trips for ten keys, and exact:
      // the resolution below runs only when two successors really share a key (or a key equals the empty marker, 0)
      const unsigned long long m = __ballot(ok);
      const int nok = __popcll(m), rank = __popcll(m & lt_mask);
      unsigned long long dupm = 0;
      if (!nodup) {
        unsigned long long *bk = (unsigned long long *)chain;
        bk[lane] = 0ull;  // LDS operations of one wave execute in order: the swaps below see the cleared table
        const unsigned long long ukey = (unsigned long long)key;
        unsigned int hb = ((unsigned int)ukey + (unsigned int)(ukey >> 32) * 0x9E3779B1u) >> 26;
        const bool odd = ok & (ukey == 0ull);
        bool rep = false, pending = ok & !odd;
        for (int trip = 0; trip < WAVE && __any(pending); ++trip) {  // <= 64 keys in 64 buckets: always terminates
          unsigned long long old = 0ull;
          if (pending) old = atomicCAS(&bk[hb], 0ull, ukey);
          const bool same = pending & (old == ukey);
          rep = rep | same;
          pending = pending & (old != 0ull) & !same;
          hb = (hb + 1u) & 63u;
        }
        dupm = __ballot(rep | odd);
      }
      int leader = lane, wlane = lane;
      double wcost = cost;
      if (dupm != 0ull) {  // rare: a plain loop keeps the register pressure of the common path low
        wave_sync_lds();   // (the accelerations by rank in rv / rk have been consumed)
        if (ok) {
          S.rk[rank] = key;
          S.rv[rank] = cost;
          S.ri[rank] = lane;
        }
        wave_sync_lds();
        leader = lane;
        wlane = -1;
        wcost = 0.0;
#pragma unroll 1
        for (int j = 0; j < nok; ++j) {
          const bool same = ok & (S.rk[j] == key);
          const int lj = S.ri[j];
          const double cj = S.rv[j];
          leader = (same & (lj < leader)) ? lj : leader;
          const bool better = same & ((wlane < 0) | (cj < wcost));
          wcost = better ? cj : wcost;
          wlane = better ? lj : wlane;
        }
      }
      if (!nodup) wave_sync_lds();
      SP_T(sp6);
      SP_ADD(5, sp5, sp6);
      const bool is_leader = ok && leader == lane;
      const bool exists = slot >= 0;
      const bool closed = exists & (slot_state == 2);
      const double ecost = slot_cost;
      const unsigned long long newm = __ballot(is_leader && !exists);
      const int nnew = __popcll(newm);
      if (nn + nnew > p.node_cap) {
        overflow = true;
        break;
      }
      int myslot = slot;
      if (is_leader && !exists) myslot = nn + __popcll(newm & lt_mask);
      const int gslot = dupm != 0ull ? __shfl(myslot, leader, WAVE) : myslot;  // the slot of my group (no repeats: my own)
      const bool write = ok && wlane == lane && (!exists || (!closed && ecost > wcost));
      if (write) {
        double *rw = nd.rec(gslot);
        const double tot = cost + 0.5 * norm2(ex - tx, ey - ty) + 0.1 * norm2(vex, vey);
        st2(rw + NR_POS, ex, ey);
        st2(rw + NR_VEL, vex, vey);
        st2(rw + NR_COST, cost, __longlong_as_double(key));
        st2(rw + NR_META, meta_pack(cur, citr + 1, 1), tot);
        st2(rw + NR_ACC, ax, ay);
        if (gslot < S.ntot) S.tot[gslot] = tot;
      }
      // ---- the new keys into the dict ----
      if (lds_dict && (nn + nnew > LH_MAX)) {
        // The search has outgrown the LDS dict (large primitive sets): every node so far goes into the hash table in global
        // memory, once, and the dict lives there from here on.  (The new nodes of this batch follow below like all later ones.)
        for (int i = lane; i < p.hash_cap; i += WAVE) tab[i] = 0;
        wave_sync_global();
        for (int s0 = 0; s0 < nn; s0 += WAVE) {
          const int si = s0 + lane;
          if (si < nn) {
            const long long k2 = __double_as_longlong(nd.rec(si)[NR_COST + 1]);
            unsigned int hh = key_hash(k2) & hmask;
            for (int guard = 0; guard < p.hash_cap; ++guard) {
              if (atomicCAS(&tab[hh], 0, si + 1) == 0) break;
              hh = (hh + 1) & hmask;
            }
          }
        }
        wave_sync_global();
        lds_dict = false;
        h = h32 & hmask;  // the probes of this batch ended in the LDS dict: the inserts below start at the key's home bucket
      }
      if (is_leader && !exists) {
        if (lds_dict) {
          // the probe ended at an empty entry `h`: two entries share a dword, another successor may be taking either of them right
          // now -- compare-and-swap on the dword, on from the next entry when this one has been taken
          const unsigned int ent = (((h32 >> 17) & 0x1fu) << 11) | (unsigned int)(myslot + 1);
          unsigned int *lh32 = (unsigned int *)S.lh;
          for (int guard = 0; guard < 4 * LH_N; ++guard) {
            const unsigned int sh = (h & 1u) * 16u;
            const unsigned int w = lh32[h >> 1];
            if (((w >> sh) & 0xffffu) != 0u) {
              h = (h + 1u) & (LH_N - 1u);
              continue;
            }
            if (atomicCAS(&lh32[h >> 1], w, w | (ent << sh)) == w) break;
          }
        } else {
          unsigned int hh = h;
          for (int guard = 0; guard < p.hash_cap; ++guard) {
            if (atomicCAS(&tab[hh], 0, myslot + 1) == 0) break;
            hh = (hh + 1) & hmask;
          }
        }
      }
      nn += nnew;
      open_n += nnew;
      SP_T(sp7);
      SP_ADD(6, sp6, sp7);
      // a further batch of this expansion probes what this one wrote: full hand-off; after the last batch only LDS (cost mirror,
      // dict) has to be in order -- the global hand-off waits until the next argmin has scanned the mirror (see there)
      if (p0 + WAVE < nprim) wave_sync_global();
      else wave_sync_lds();
      SP_T(sp8);
      SP_ADD(7, sp7, sp8);
      SP_ADD(8, sp0, sp8);
#ifdef D2D_SEARCH_PROF
      spacc[9] += 1;
#endif
    }
    if (overflow) break;
  }
  if (lane == 0) {
    stat[0] += 1;
    stat[1] = expansions;
    stat[2] = nn;
    if (overflow) stat[3] = 1;
  }
  if (goal < 0 || overflow) {
    SP_FLUSH();
    return -1;
  }
  SP_T(spa);
  // ---- :207-216: waypoints of every primitive on the path, start side first ----
  int depth = 0;
  for (int q = goal; q != 0 && depth <= 128; q = meta_parent(nd.rec(q)[NR_META])) depth += 1;  // bounded: a wave must always terminate
  if (depth * p.n_ts > p.traj_cap || depth > 128) {
    if (lane == 0) stat[3] = 1;
    return -1;
  }
  if (lane == 0) {
    int q = goal;
    for (int lvl = depth - 1; lvl >= 0; --lvl) {
      chain[lvl] = q;
      q = meta_parent(nd.rec(q)[NR_META]);
    }
  }
  wave_sync_lds();
  const int total = depth * p.n_ts;
  double *tbox = traj_boxes(p, e);
  for (int w0 = 0; w0 < total; w0 += WAVE) {
    const int w = w0 + lane;
    double ox = 0.0, oy = 0.0;
    if (w < total) {
      const int lvl = w / p.n_ts, mi = w - lvl * p.n_ts;
      const int q = chain[lvl];
      const double *rq = nd.rec(q);
      const int par = meta_parent(rq[NR_META]);
      const double2 acc = ld2(rq + NR_ACC);
      const double hx = acc.x / 2, hy = acc.y / 2;
      const double t = p.traj_t[3 * mi], t2 = p.traj_t[3 * mi + 1], tt = p.traj_t[3 * mi + 2];
      const double2 pp = ld2(nd.rec(par) + NR_POS), pv = ld2(nd.rec(par) + NR_VEL);
      double *o = traj + (size_t)w * 4;
      ox = rint(__builtin_fma(t2, hx, pp.x + t * pv.x));  // :121
      oy = rint(__builtin_fma(t2, hy, pp.y + t * pv.y));
      o[0] = ox;
      o[1] = oy;
      o[2] = pv.x + tt * hx;                                // :122
      o[3] = pv.y + tt * hy;
    }
    if (tbox) {  // the bounding box of this chunk of 64 waypoints (d2d_plan.traj_box): what lets the every-step walks skip it
      const bool on = w < total;
      const double xlo = wave_fmin(on ? ox : 1e300), ylo = wave_fmin(on ? oy : 1e300);
      const double xhi = wave_fmax(on ? ox : -1e300), yhi = wave_fmax(on ? oy : -1e300);
      if (lane == 0) {
        st2(tbox + 4 * (w0 / WAVE), xlo, ylo);
        st2(tbox + 4 * (w0 / WAVE) + 2, xhi, yhi);
      }
    }
  }
  SP_T(spb);
  SP_ADD(11, spa, spb);
#ifdef D2D_SEARCH_PROF
  spacc[12] += 1;
#endif
  SP_FLUSH();
  return total;
}

// LDS per wave of k_plan: 5 planes of ncap doubles (active trackers) + the search's hand-off arrays, cost mirror, wall rows, dict
__host__ __device__ inline int plan_wave_bytes(int N, int nu, int n_sample, int W, int H) {
  const int ncap = ((N > 0 ? N : 1) + 3) & ~3;
  const int nu4 = ((nu + 3) & ~3) + ((nu * nu + 3) & ~3), ns4 = (2 * n_sample + 3) & ~3;  // u_space + the primitives' cost terms
  const int rowb = search_wall_rows(W, H) ? 8 * 64 : 0;
  return 5 * 8 * ncap + 8 * (nu4 + ns4) + 8 * 64 + 8 * 64 + 4 * 64 + 128 * 4 + 16 + 8 * search_lds_nodes(N) + rowb + 2 * LH_N;
}

// replan_check + plan + head waypoint of env e by one wave; `base`: plan_wave_bytes() bytes of LDS
__device__ __forceinline__ void plan_carve(const d2d_cfg &c, const d2d_plan &p, char *base, TrkView &T, SearchLds &S) {
  const int N = c.N, ncap = ((N > 0 ? N : 1) + 3) & ~3;
  T.mx = (double *)base;
  T.my = T.mx + ncap;
  T.vx = T.my + ncap;
  T.vy = T.vx + ncap;
  T.lim = T.vy + ncap;
  S.us = T.lim + ncap;
  S.pc = S.us + ((p.nu + 3) & ~3);
  S.st = S.pc + ((p.nu * p.nu + 3) & ~3);
  S.rv = S.st + ((2 * p.n_sample + 3) & ~3);
  S.rk = (long long *)(S.rv + 64);
  S.ri = (int *)(S.rk + 64);
  S.chain = S.ri + 64;
  S.misc = S.chain + 128;
  S.tot = (double *)(S.misc + 4);
  S.ntot = search_lds_nodes(N);
  unsigned long long *after = (unsigned long long *)(S.tot + S.ntot);
  S.prow = search_wall_rows(c.W, c.H) ? after : nullptr;
  S.lh = (unsigned short *)(after + (search_wall_rows(c.W, c.H) ? 64 : 0));
}

// Writes the head step_pos will consume (utils.py:733-739) and the planner's result; pops the head.
// `w_head`: the head waypoint (x, y, vx, vy) where the caller holds it already (lane 0), else null: read here -- both halves before
// the first store (the stores could alias the loads for all the compiler knows: four dependent round trips otherwise).
__device__ __forceinline__ void plan_emit(const d2d_state &s, const d2d_plan &p, int e, int lane, int head, int stored, int ok,
                                          const double4 *w_head = nullptr) {
  if (lane == 0) {
    const double *traj = p.traj + (size_t)e * p.traj_cap * 4;
    int *hdr = p.traj_hdr + (size_t)e * 2;
    unsigned char *plan_ok = (unsigned char *)s.plan_ok, *wp_valid = (unsigned char *)s.wp_valid;
    double *wp = (double *)s.wp + (size_t)e * 6;
    plan_ok[e] = (unsigned char)ok;
    if (stored - head > 0) {
      double2 wa, wb;
      if (w_head) {
        wa = make_double2(w_head->x, w_head->y);
        wb = make_double2(w_head->z, w_head->w);
      } else {
        const double *w = traj + (size_t)head * 4;
        wa = ld2(w);
        wb = ld2(w + 2);
      }
      wp[0] = wa.x; wp[1] = wa.y; wp[2] = wb.x; wp[3] = wb.y; wp[4] = 0.0; wp[5] = 0.0;
      wp_valid[e] = 1;
      head += 1;
    } else {
      wp_valid[e] = 0;
      for (int i = 0; i < 6; ++i) wp[i] = 0.0;
    }
    hdr[0] = head;
    hdr[1] = stored;
  }
}

// The part of the planner stage every step runs: tracker bookkeeping (the active ones staged in LDS, where a search
// that follows finds them), replan_check, and -- when the trajectory is still there -- its head.  Returns true when
// the trajectory is empty, i.e. Primitive.plan has to search (plan_env_search).  Kept apart from the search so that,
// as a called function in the persistent loop, the common path does not pay the search's register saves.
// `w_head_out` (optional): where the trajectory is kept, the head waypoint plan_emit stored (x, y, vx, vy), wave-uniform -- the
// act phase of the persistent loop takes it from there instead of reading the planner's result back from memory.
// `walls_ok` (optional; the persistent loop's own flag of this env): in = every remaining waypoint passed replan_check's wall test
// at the step before AND nothing but this step's rays has written the explored map since (so only waypoints inside the cells the rays
// can reach need the test again); out = the same for the next step.  Null: the test walks every waypoint (a stand-alone launch: the
// host may have written the map between two calls).
__device__ __forceinline__ bool plan_env_quick(const d2d_cfg &c, const d2d_state &s, const d2d_plan &p, int e, int lane, char *base,
                                               double4 *w_head_out = nullptr, int *walls_ok = nullptr) {
  const int N = c.N;
  TrkView T;
  SearchLds S;
  plan_carve(c, p, base, T, S);
  const double inv_scale = 1.0 / c.scale;
  const unsigned char *__restrict__ dm = s.dmap + (size_t)e * grid_bytes(c);
  // the trajectory's header goes out with the trackers' loads below (it addresses the waypoints: a round trip of its own otherwise)
  // (as a per-lane load, made uniform only behind the trackers' loads: a uniform load is waited for where it stands)
  int *hdr = p.traj_hdr + (size_t)e * 2;
  const int hdr_l = hdr[lane & 1];
  // ---- trackers: archive bookkeeping (utils.py:184,238) and the active ones into LDS ----
  int nact = 0;
  for (int k0 = 0; k0 < N; k0 += WAVE) {
    const int k = k0 + lane;
    bool act = false;
    double m0 = 0, m1 = 0, m2 = 0, m3 = 0, rad = 0, lim2 = 0;
    if (k < N) {
      // one batch of loads: the tracker's mean is fetched whether or not it is active (a second, dependent round trip otherwise)
      const double *mu = s.kf + ((size_t)e * N + k) * D2D_KF;
      m0 = mu[0]; m1 = mu[1]; m2 = mu[2]; m3 = mu[3];
      act = s.active[(size_t)e * N + k] != 0;
      const bool prev = p.trk_prev[(size_t)e * N + k] != 0;
      rad = p.trk_radius[(size_t)e * N + k];
      const double lim_in = p.trk_lim[(size_t)e * N + k];
      lim2 = lim_in;
      if (prev && !act) {
        rad = p.agent_radius;
        p.trk_radius[(size_t)e * N + k] = rad;
        lim2 = 0.0;
      }
      // replan_check's `norm(d) <= drone_radius + radius` as `d.d <= lim2`: the threshold (three square roots to find) only
      // changes with the tracker's radius, so it is kept in the plugin state; 0 = not computed for this radius yet
      if (act && !(lim2 > 0.0)) lim2 = sq_threshold(c.drone_radius + rad);
      if (lim2 != lim_in) p.trk_lim[(size_t)e * N + k] = lim2;
      if (prev != act) p.trk_prev[(size_t)e * N + k] = act ? 1 : 0;
    }
    const unsigned long long am = __ballot(act);
    if (act) {
      const int q = nact + __popcll(am & ((1ull << lane) - 1ull));
      T.mx[q] = m0; T.my[q] = m1; T.vx[q] = m2; T.vy[q] = m3;
      T.lim[q] = lim2;  // traj_planner.py:228, cached threshold (see above); a search replaces it with its own (below)
    }
    nact += __popcll(am);
  }
  T.n = nact;
  wave_sync_lds();
  int head = __builtin_amdgcn_readlane(hdr_l, 0), stored = __builtin_amdgcn_readlane(hdr_l, 1);
  // ---- replan_check, traj_planner.py:220-233 ----
  double *__restrict__ traj = p.traj + (size_t)e * p.traj_cap * 4;
  double4 w_first = make_double4(0.0, 0.0, 0.0, 0.0);  // lane 0: the head waypoint, what plan_emit hands to step_pos
  {
    const int n = stored - head;
    bool bad = false;
    // The walk, in chunks of 64 slots of the trajectory buffer.  A chunk is loaded only if its bounding box (d2d_plan.traj_box) says
    // it can matter: it holds the head; or one of its waypoints can come within an active tracker's radius (the tracker's positions
    // over the chunk's time span are a segment: box against box); or it can hold a waypoint on a cell the explored map has gained
    // since the last full test -- inside a persistent launch only this step's rays write that map, within `reach` cells of the
    // drone's cell.  Everything a box excludes is excluded exactly; the per-waypoint tests below are the reference's.
    const int c_lo = head >> 6, c_hi = n > 0 ? (stored - 1) >> 6 : c_lo - 1;
    const double *tbox = traj_boxes(p, e);
    const bool chunked = tbox != nullptr && c_hi - c_lo < 64 && c_hi - c_lo >= D2D_TRAJ_PRUNE_CHUNKS;  // (short trajectories: as they are)
    // (the swept value is uint8: beyond 2 560 waypoints it wraps to 0 and a waypoint skips the wall test for ten steps -- such a
    // trajectory takes the full walk)
    const bool walls_known = walls_ok != nullptr && *walls_ok != 0 && n < 2560;
    unsigned long long cmask = 0ull;
    bool walls_full = true;
    if (__builtin_expect(chunked && n > 0, 0)) {
      const double2 dpos = ld2(s.drone + (size_t)e * D2D_DF);  // the pose this step's rays were cast from (the control stage comes later)
      bool need = false;
      if (c_lo + lane <= c_hi) {
        const int cc = c_lo + lane;
        const double2 lo = ld2(tbox + 4 * cc), hi = ld2(tbox + 4 * cc + 2);
        need = cc == c_lo;
        if (!walls_known) {
          need = true;
        } else {
          const int reach = (int)((c.depth + 1.5 * (c.scale - 1.0)) / c.scale) + 2;  // Geom.reach: the cells a ray can travel
          const int ocx = cell_fast(dpos.x, c.scale, inv_scale), ocy = cell_fast(dpos.y, c.scale, inv_scale);
          const double wx0 = (double)(ocx - reach) * c.scale, wx1 = (double)(ocx + reach + 1) * c.scale;
          const double wy0 = (double)(ocy - reach) * c.scale, wy1 = (double)(ocy + reach + 1) * c.scale;
          need = need | ((hi.x >= wx0) & (lo.x < wx1) & (hi.y >= wy0) & (lo.y < wy1));
        }
        const int a_lo = max(cc << 6, head), a_hi = min((cc << 6) + 63, stored - 1);
        const double t_lo = (double)(a_lo - head) * c.dt, t_hi = (double)(a_hi - head) * c.dt;
        for (int q = 0; q < nact; ++q) {
          // estimate_pos(t) = m + t v is monotone in t on each axis: its values over the chunk lie between those at the two ends
          const double mxq = T.mx[q], myq = T.my[q], vxq = T.vx[q], vyq = T.vy[q];
          const double ex0 = mxq + t_lo * vxq, ex1 = mxq + t_hi * vxq, ey0 = myq + t_lo * vyq, ey1 = myq + t_hi * vyq;
          const double rad = sqrt(T.lim[q]) * (1.0 + 1e-9) + 1e-9;   // d.d <= lim needs |dx| <= sqrt(lim) and |dy| <= sqrt(lim)
          const double exl = fmin(ex0, ex1), exh = fmax(ex0, ex1), eyl = fmin(ey0, ey1), eyh = fmax(ey0, ey1);
          need = need | ((hi.x + rad >= exl) & (lo.x - rad <= exh) & (hi.y + rad >= eyl) & (lo.y - rad <= eyh));
        }
      }
      cmask = __ballot(need);
      walls_full = !walls_known;
    }
    auto visit = [&](int a, bool first) {
      if (a >= head && a < stored) {
        const int i = a - head;
        const double *w = traj + (size_t)a * 4;
        const double2 wxy = ld2(w), wv = ld2(w + 2);  // (x, y), (vx, vy)
        const double4 w4 = make_double4(wxy.x, wxy.y, wv.x, wv.y);
        if (first) w_first = w4;
        const double wx = w4.x, wy = w4.y;
        const double ti = (double)i * c.dt;
        const int ci = cell_fast(wx, c.scale, inv_scale), cj = cell_fast(wy, c.scale, inv_scale);
        const bool in = ci >= 0 && ci < c.W && cj >= 0 && cj < c.H;
        // swep_map is uint8: the stored value is trunc(i * dt); a wall under a non-zero stored value forces the replan
        const int sv = ((int)ti) & 0xff;
        if (in && sv > 0 && dm[grid_ix(c, min(max(ci, 0), c.W - 1), min(max(cj, 0), c.H - 1))] == D2D_OCCUPIED) bad = true;
        for (int q = 0; q < nact; q += 4) {  // four trackers per round, loads first (see plan_is_free)
          double mx[4], my[4], vx[4], vy[4], lim[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            mx[u] = T.mx[q + u]; my[u] = T.my[q + u]; vx[u] = T.vx[q + u]; vy[u] = T.vy[q + u]; lim[u] = T.lim[q + u];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const double ex = mx[u] + ti * vx[u], ey = my[u] + ti * vy[u];
            const double dx = ex - wx, dy = ey - wy;
            bad = bad | ((q + u < nact) & (__builtin_fma(dy, dy, dx * dx) <= lim[u]));
          }
        }
      }
    };
    if (__builtin_expect(chunked, 0)) {
      for (unsigned long long m = cmask; m; m &= m - 1ull) {
        const int cc = c_lo + __ffsll((long long)m) - 1;
        visit((cc << 6) + lane, cc == c_lo);
      }
    } else {
      for (int a0 = head; a0 < stored; a0 += WAVE) visit(a0 + lane, a0 == head);   // (lane 0 of the first pass holds the head)
    }
    // chunked: the head waypoint sits in lane head % 64 of the first chunk, plan_emit takes it from lane 0
    if (__builtin_expect(chunked && n > 0, 0)) {
      const int hl = head & 63;
      w_first = make_double4(readlane_f64(w_first.x, hl), readlane_f64(w_first.y, hl), readlane_f64(w_first.z, hl), readlane_f64(w_first.w, hl));
    }
    const bool replan = __any(bad);
    if (replan) head = stored = 0;
    // (n < 2560: every waypoint the NEXT step tests -- one index lower -- then had a non-zero swept value at this step too, i.e. was
    // tested now; a longer trajectory holds waypoints whose uint8 value has wrapped to 0)
    if (walls_ok) *walls_ok = (!replan && n > 0 && n < 2560 && (walls_full || walls_known)) ? 1 : 0;
  }
  if (stored - head == 0) {  // Primitive.plan has to search: the trackers (and their count) wait in LDS
    // ... with the limit of Planner.is_free in the one limit plane (traj_planner.py:58: drone_radius + radius + 5 + var_cam, as
    // its squared threshold): the same compaction walk as above, a few percent of the steps
    int q0 = 0;
    for (int k0 = 0; k0 < N; k0 += WAVE) {
      const int k = k0 + lane;
      const bool act = k < N && s.active[(size_t)e * N + k] != 0;
      const unsigned long long am = __ballot(act);
      if (act) T.lim[q0 + __popcll(am & ((1ull << lane) - 1ull))] = sq_threshold(c.drone_radius + p.trk_radius[(size_t)e * N + k] + 5 + c.sigma);
      q0 += __popcll(am);
    }
    if (lane == 0) {
      hdr[0] = 0;
      hdr[1] = 0;
      S.misc[0] = nact;
    }
    wave_sync_lds();
    return true;
  }
  plan_emit(s, p, e, lane, head, stored, 1, &w_first);  // traj_planner.py:128-129: a non-empty trajectory is kept
  if (w_head_out)
    *w_head_out = make_double4(readlane_f64(w_first.x, 0), readlane_f64(w_first.y, 0), readlane_f64(w_first.z, 0), readlane_f64(w_first.w, 0));
  return false;
}

// Primitive.plan's search (traj_planner.py:125-218) for an env whose plan_env_quick returned true.
__device__ __forceinline__ void plan_env_search(const d2d_cfg &c, const d2d_state &s, const d2d_plan &p, int e, int lane, char *base) {
  TrkView T;
  SearchLds S;
  plan_carve(c, p, base, T, S);
  T.n = S.misc[0];  // the trackers are still in LDS, the limit plane holds plan's thresholds (plan_env_quick)
  const double inv_scale = 1.0 / c.scale;
  const unsigned char *__restrict__ dm = s.dmap + (size_t)e * grid_bytes(c);
  // A search is a long chain of short dependent steps and, in the persistent loop, what the slowest env of a launch
  // spends its time on; its wave shares the SIMD with three others that mostly run throughput phases.  Raised issue
  // priority lets it go first whenever it is ready (0.78 ms -> its stand-alone 0.46 ms is the range at stake).
  __builtin_amdgcn_s_setprio(3);
  const int found = plan_search(c, s, p, e, lane, T, S, dm, inv_scale);
  __builtin_amdgcn_s_setprio(0);
  wave_sync_global();
  plan_emit(s, p, e, lane, 0, found > 0 ? found : 0, found >= 0 ? 1 : 0);
}

// replan_check + plan + head waypoint of env e by one wave; `base`: plan_wave_bytes() bytes of LDS
__device__ __forceinline__ void plan_env(const d2d_cfg &c, const d2d_state &s, const d2d_plan &p, int e, int lane, char *base) {
  if (plan_env_quick(c, s, p, e, lane, base)) plan_env_search(c, s, p, e, lane, base);
}

__global__ __launch_bounds__(WAVE *WAVES_PER_BLOCK) void k_plan(d2d_cfg c, d2d_state s, d2d_plan p, int skip_done) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
  const int e = blockIdx.x * (int)(blockDim.x / WAVE) + wv;
  if (e >= c.B) return;
  if (skip_done && s.flags[(size_t)e * 4 + D2D_F_DONE] != 0) return;
  plan_env(c, s, p, e, lane, d2d_lds + (size_t)wv * plan_wave_bytes(c.N, p.nu, p.n_sample, c.W, c.H));
}

// ------------------------------------------------------------------------------------------------
// Oxford
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ long long dkey(double x) {
  const long long b = __double_as_longlong(x);
  return b ^ ((b >> 63) & 0x7FFFFFFFFFFFFFFFll);
}

// np.arccos(q) <= view_angle (yaw_planner.py:77) through the host's decision window
__device__ __forceinline__ bool acos_le(const d2d_plan &p, double q) {
  if (!(fabs(q) <= 1.0)) return false;
  const long long k = dkey(q);
  if (k >= p.acos_key_lo + 64) return true;
  if (k < p.acos_key_lo) return false;
  return ((p.acos_mask >> (int)(k - p.acos_key_lo)) & 1ull) != 0ull;
}

// Oxford.get_view_map (yaw_planner.py:67-79) for the cell whose origin is (x, y).
// The literal test is arccos(dot / sqrt(d2)) <= half_fov (through the host's window) and d2 <= depth^2.  For a
// view cone narrower than 180 degrees (cos_half > 0) the window sits at q ~ cos_half > 0, so a cell is settled
// without the square root and the division whenever dot^2 and cos_half^2 d2 differ by more than 1e-12 relative
// (the window is 64 ulp ~ 7e-15 wide and the roundings of both sides stay below 1e-15): only cells within
// 1e-12 of the cone's edge or of its axis take the literal path.  `quick` is 0 when the shortcut does not apply.
struct ViewCone {
  double cy, sy;   // view direction: cos(radians(yaw)), -sin(radians(yaw))
};

__device__ __forceinline__ bool view_cell(const d2d_plan &p, double depth2, double quick, double x0, double y0,
                                          const ViewCone &v, double x, double y) {
  const double a = x0 - x, b = y0 - y;
  const double d2 = a * a + b * b;
  if (d2 <= 0.0) return true;
  if (!(d2 <= depth2)) return false;
  const double dot = (x - x0) * v.cy + (y - y0) * v.sy;
  if (quick > 0.0) {
    const double lhs = dot * dot, rhs = quick * d2;  // quick = cos_half^2
    // inside for sure: well above the edge AND q well below 1 (dead ahead q can round above 1, where arccos is NaN
    // and the reference's comparison false: those cells take the literal path too)
    if (dot > 0.0 && lhs > rhs * (1.0 + 1e-12) && lhs < d2 * (1.0 - 1e-12)) return true;
    if (dot <= 0.0 || lhs < rhs * (1.0 - 1e-12)) return false;
  }
  return acos_le(p, dot / sqrt(d2));
}

// plugin state of env e back to "fresh objects" (experiment.py:31-34)
__device__ __forceinline__ void plan_reset_env(const d2d_cfg &c, const d2d_plan &p, size_t e, int lane) {
  const size_t N = (size_t)(c.N > 0 ? c.N : 1), WH = (size_t)c.W * c.H;
  if (p.traj_hdr && lane < 2) p.traj_hdr[e * 2 + lane] = 0;
  for (size_t k = lane; k < N; k += WAVE) {
    if (p.trk_radius && p.trk_radius0) p.trk_radius[e * N + k] = p.trk_radius0[e * N + k];
    if (p.trk_prev) p.trk_prev[e * N + k] = 0;
    if (p.trk_lim) p.trk_lim[e * N + k] = 0.0;
  }
  if (p.seen_step)
    for (size_t i = lane; i < WH; i += WAVE) p.seen_step[e * WH + i] = 0;
}


// Cells whose ORIGIN can lie in the sector {|p - c| <= depth, angle(p - c, dir) <= w}: a box [i_lo, i_hi] x [j_lo, j_hi], one
// cell of margin on every side (float arithmetic on purpose: the box only has to CONTAIN the sector; the exact per-cell tests
// decide).  `cw`, `sw`: cos / sin of w; (dx, dy): the unit view direction.  Clipped to `lo` / `hi` bounds given by the caller.
struct CellBox {
  int i_lo, i_hi, j_lo, j_hi;
};
__device__ __forceinline__ CellBox sector_box(double cx, double cy, double depth, double inv_scale, double ddx, double ddy,
                                              float cw, float sw) {
  const float dx = (float)ddx, dy = (float)ddy;
  // edge directions dir rotated by +-w, the apex (0), and the axis directions that lie inside the sector
  const float ex0 = dx * cw - dy * sw, ey0 = dy * cw + dx * sw, ex1 = dx * cw + dy * sw, ey1 = dy * cw - dx * sw;
  const float m = cw - 0.02f;  // an axis is inside when its cosine with dir is >= cos w (slack towards "inside")
  float xmax = fmaxf(0.f, fmaxf(ex0, ex1)), xmin = fminf(0.f, fminf(ex0, ex1));
  float ymax = fmaxf(0.f, fmaxf(ey0, ey1)), ymin = fminf(0.f, fminf(ey0, ey1));
  xmax = dx >= m ? 1.f : xmax;
  xmin = -dx >= m ? -1.f : xmin;
  ymax = dy >= m ? 1.f : ymax;
  ymin = -dy >= m ? -1.f : ymin;
  const float d = (float)depth * 1.001f + 0.01f, is = (float)inv_scale, fx = (float)cx, fy = (float)cy;
  CellBox b;
  b.i_lo = (int)floorf((fx + d * xmin) * is) - 1;
  b.i_hi = (int)floorf((fx + d * xmax) * is) + 1;
  b.j_lo = (int)floorf((fy + d * ymin) * is) - 1;
  b.j_hi = (int)floorf((fy + d * ymax) * is) + 1;
  return b;
}

#define D2D_TOBS_LDS 16
// Maps above this many cells: the pairwise-summation plan of np.sum over W x H (one block per <= 128 cells + the tree of their
// additions) no longer fits the wave's LDS.  Only blocks that hold a non-zero term matter (x + 0.0 == x, every term is >= +0.0) and
// those lie inside the view box: the SPARSE path finds them by walking numpy's recursion down from the root for the box's cells and
// adds their sums up the same recursion -- no table of the W x H plan on the device at all (gaze_env).
#ifndef D2D_GAZE_DENSE_CELLS
#define D2D_GAZE_DENSE_CELLS 4096
#endif
#ifndef D2D_GAZE_HOT
#define D2D_GAZE_HOT 64     // blocks with a non-zero term the sparse path holds (two per box row: a block has >= 64 cells)
#endif
struct GazeGeom {
  int bbn;    // cells per axis of the bounding box of a view disk
  int ncell;  // bbn * bbn
  int sparse; // the sparse pairwise path (maps above D2D_GAZE_DENSE_CELLS cells)
  int nleaf;  // block records kept in LDS: pw_nleaf, or D2D_GAZE_HOT
  int ntree;  // ints of the addition tree kept in LDS: pw_ntree, or 0
  int nnode;  // sums per candidate: blocks + additions, or D2D_GAZE_HOT
  int wave_bytes;
};

__host__ __device__ inline GazeGeom gaze_geom(const d2d_cfg &c, const d2d_plan &p) {
  GazeGeom g;
  g.bbn = 2 * ((int)(c.depth / c.scale) + 1) + 3;
  g.ncell = g.bbn * g.bbn;
  g.sparse = (c.W * c.H > D2D_GAZE_DENSE_CELLS) ? 1 : 0;
  // the sparse path holds two blocks per box row, D2D_GAZE_HOT in all: a view box deeper than that (above 13 cells of view depth)
  // takes the dense plan whatever the map's size -- while it fits the wave's LDS (plan_check)
  if (g.sparse && 2 * g.bbn > D2D_GAZE_HOT) g.sparse = 0;
  g.nleaf = g.sparse ? D2D_GAZE_HOT : p.pw_nleaf;
  g.ntree = g.sparse ? 2 * D2D_GAZE_HOT : p.pw_ntree;  // sparse: depth and path of every hot block in numpy's recursion
  g.nnode = g.sparse ? D2D_GAZE_HOT : 2 * p.pw_nleaf - 1;
  // int swept index + double reward + candidate bits per box cell, block sums + add stacks per candidate, the plan
  const int sums = 8 * p.n_yaw * g.nnode, live = 4 * g.ncell;  // the live-cell list shares the sums' space
  const int bytes = 4 * g.ncell + 8 * g.ncell + ((g.ncell + 7) & ~7) + (((sums > live ? sums : live) + 7) & ~7) + 8 * 16 +
                    4 * (4 * g.nleaf + g.ntree) + 16 + 8 * D2D_TOBS_LDS;  // + row / column masks of the non-zero terms, table head
  g.wave_bytes = (bytes + 15) & ~15;
  return g;
}

// Reset-if-done + Oxford.plan of env e by one wave; `base`: gaze_geom().wave_bytes bytes of LDS.
// `auto_reset`: an env whose previous step ended its episode (flags[D2D_F_DONE]) first goes back to the snapshot
// `init` with fresh plugin state -- the next episode of the reference's sweeps (main.py:26-57).
// The stage is a chain of small dependent loads (table rows of sin / cos, the seen map, the pairwise plan) rather
// than arithmetic, so loads are batched: one sin / cos pass for all seven view directions, every lane's seen-map
// cells fetched before the first is used, the pairwise plan staged in LDS.
// `known_done`: the env's episode flag where the caller holds it (the persistent loop), -1: read here.
__device__ __forceinline__ void gaze_env(const d2d_cfg &c, const d2d_state &s, const d2d_plan &p, const d2d_state &init,
                                         int auto_reset, int e, int lane, char *base, int known_done = -1) {
  // ---- one batch of loads: everything the stage needs that does not hang on another load (the episode flag, the pose, the step
  //      count, the trajectory header, the table heads, the pairwise plan, the candidates' yaw rates) is requested before the first
  //      of them is looked at -- one round trip where the straightforward order makes seven dependent ones ----
  const bool oxford = p.gaze == D2D_GAZE_OXFORD;
  const GazeGeom g = gaze_geom(c, p);
  const double *dr = s.drone + (size_t)e * D2D_DF;
  const int *hdr = p.traj_hdr + (size_t)e * 2;
  const int was_done = !auto_reset ? 0 : (known_done >= 0 ? known_done : (int)s.flags[(size_t)e * 4 + D2D_F_DONE]);
  double x0 = dr[D2D_D_X], y0 = dr[D2D_D_Y], yaw = dr[D2D_D_YAW];
  int steps = s.counters[(size_t)e * D2D_CF + D2D_C_STEPS];
  int head = 0, stored = 0;
  double ys_l = 0.0, tob_l = 0.0;  // lane a < n_yaw (and a + 8, a + 16 ...): yaw_space[a]; lane < D2D_TOBS_LDS: its entry of the table of times
  if (oxford) {
    head = hdr[0];
    stored = hdr[1];
    ys_l = p.yaw_space[min(lane & 7, max(p.n_yaw, 1) - 1)];  // (lanes 8..15 hold 0..7's again: the view directions' sines)
    tob_l = p.tobs_tab[min(lane, p.tobs_len - 1)];
  }
  if (was_done != 0) {
    reset_env(c, s, init, (size_t)e, lane);
    plan_reset_env(c, p, (size_t)e, lane);
    wave_sync_global();
    x0 = dr[D2D_D_X]; y0 = dr[D2D_D_Y]; yaw = dr[D2D_D_YAW];  // (the fence above: these are fresh loads)
    steps = s.counters[(size_t)e * D2D_CF + D2D_C_STEPS];
    if (oxford) {
      head = hdr[0];
      stored = hdr[1];
    }
  }
  if (!oxford) return;
#ifdef D2D_CHAIN_PROF
  unsigned long long gz_t = __builtin_amdgcn_s_memtime();
#endif
  double *rew = (double *)base;                                   // [ncell]
  const int nnode = g.nnode;                                      // blocks + their pairwise sums up to the root (sparse: the hot blocks)
  double *lsum = rew + g.ncell;                                   // [n_yaw][nnode]
  const int lsum_doubles = max(p.n_yaw * nnode, (g.ncell + 1) / 2);
  double *stk = lsum + lsum_doubles;                              // [8][2] view directions
  double *tobl = stk + 16;                                        // [D2D_TOBS_LDS] head of tobs_tab row 0
  int *swi = (int *)(tobl + D2D_TOBS_LDS);                        // [ncell]
  int *swl = (int *)lsum;                                         // [ncell] the live cells of the box: done before the sums start
  int *pwl = swi + g.ncell;                                       // [pw_nleaf][4] + [pw_ntree]: the pairwise plan
  int *pwp = pwl + 4 * g.nleaf;
  unsigned char *cm = (unsigned char *)(pwp + g.ntree);           // [ncell]
  int *rng = (int *)(cm + ((g.ncell + 7) & ~7));                  // [2] first / last grid row with a non-zero term of any sum
  const int W = c.W, H = c.H;
  const double deg2rad = 0x1.1df46a2529d39p-6;                    // math.radians
  const double depth2 = c.depth * c.depth;
  const double inv_scale = 1.0 / c.scale;
  const int call = steps + 1;  // one plan() per step, before it
  const int n = stored - head;
  int *__restrict__ seen = p.seen_step + (size_t)e * W * H;
  double *act = (double *)s.action;
  if (call >= p.tobs_len) {  // stepping past the longest episode (D2D_DONE_CONTINUE): the table ends, the yaw is held
    if (lane == 0) act[e] = 0.0;
    return;
  }
  // the head of the trajectory (the point the candidates look from), requested now, used after the seen pass
  const double *__restrict__ traj = p.traj + (size_t)e * p.traj_cap * 4;
  const double2 hxy = *(const double2 *)(traj + (size_t)(n > 0 ? head : 0) * 4);
  // the table head (the entries below 1: reward = the value itself) into LDS (the pairwise plan -- a few hundred bytes -- only where
  // the exact sums run: below, behind the quick decision)
  if (lane < D2D_TOBS_LDS) tobl[lane] = tob_l;
  const FastDiv fdb(g.bbn);
  // shortcut of view_cell: only for cones narrower than 180 degrees whose edge is well inside (0, 1); cos(half_fov)
  // is the double in the middle of the host's arccos window
  double quick = 0.0;
  {
    const long long k = p.acos_key_lo + 32;
    const double ch = __longlong_as_double(k ^ ((k >> 63) & 0x7FFFFFFFFFFFFFFFll));
    if (p.half_fov < 1.5 && ch > 0.05) quick = ch * ch;
  }
  // ---- the seven view directions in one pass: lane a < n_yaw = candidate a, lane 7 = the current pose (:71, :114) ----
  double *vdir = stk;  // [8][2], free until the add stacks are used
#ifdef D2D_SINCOS_TWO_PASS
  if (lane < 8) {
    double cyv = 0.0, syv = 0.0;
    if (lane < p.n_yaw || lane == 7) {
      // Drone2D.__init__ takes `yaw % 360` for the candidates (utils.py:718); the drone's own yaw already is
      const double ty = (lane == 7) ? yaw : py_mod360(yaw + ys_l * c.dt);
      cyv = d2d_cos(ty * deg2rad);
      syv = -d2d_sin(ty * deg2rad);
    }
    vdir[2 * lane] = cyv;
    vdir[2 * lane + 1] = syv;
  }
#else
  // lanes 0..7 the cosines, lanes 8..15 the (negated) sines of the same eight angles: one walk through the sin / cos code for both
  // (d2d_sin_or_cos: each lane does what d2d_cos / d2d_sin do for its argument, bit for bit)
  if (lane < 16) {
    const int a = lane & 7;
    double v = 0.0;
    if (a < p.n_yaw || a == 7) {
      // Drone2D.__init__ takes `yaw % 360` for the candidates (utils.py:718); the drone's own yaw already is
      const double ty = (a == 7) ? yaw : py_mod360(yaw + ys_l * c.dt);
      const double r = d2d_sin_or_cos(ty * deg2rad, lane < 8);
      v = lane < 8 ? r : -r;
    }
    vdir[2 * a + (lane >> 3)] = v;
  }
#endif
  // the largest yaw step of a candidate, degrees: a maximum over the lanes that hold the rates (exact in any order)
  double span_deg;
  {
    double m = lane < p.n_yaw ? fabs(ys_l) * c.dt : 0.0;  // (n_yaw <= 7: plan_check)
    m = fmax(m, row_shl_f64<1>(m));
    m = fmax(m, row_shl_f64<2>(m));
    m = fmax(m, row_shl_f64<4>(m));
    span_deg = readlane_f64(m, 0);
  }
  wave_sync_lds();
  // the drone's own direction stays in registers; the candidates' (used once per live cell, below) are read from LDS where they are
  // used: eight directions in registers were 32 VGPRs live across the stage -- its register peak
  ViewCone cone7;
  cone7.cy = vdir[14];
  cone7.sy = vdir[15];
  GZ(0);  // tables, the seven view directions
  // ---- t_i: cells the current pose sees (yaw_planner.py:93-97); only the box around the drone can be seen ----
  // cos / sin of the pre-test sectors (float, with degrees of slack): the drone's own view, and the sector that holds every
  // candidate's view (half_fov + the largest yaw step)
  const float w_own = (float)p.half_fov + 2.0f * 0.0174533f, w_all = (float)p.half_fov + ((float)span_deg + 3.0f) * 0.0174533f;
  const bool boxes = w_all < 1.518f && span_deg <= 90.0;  // else: the whole disk's box, no sector pre-test
  {
    const int bi = (int)floor((x0 - c.depth) * inv_scale) - 1, bj = (int)floor((y0 - c.depth) * inv_scale) - 1;
    CellBox b;
    b.i_lo = bi; b.i_hi = bi + g.bbn - 1; b.j_lo = bj; b.j_hi = bj + g.bbn - 1;
    if (boxes) {
      const CellBox sb = sector_box(x0, y0, c.depth, inv_scale, cone7.cy, cone7.sy, __cosf(w_own), __sinf(w_own));
      b.i_lo = max(b.i_lo, sb.i_lo); b.i_hi = min(b.i_hi, sb.i_hi); b.j_lo = max(b.j_lo, sb.j_lo); b.j_hi = min(b.j_hi, sb.j_hi);
    }
    b.i_lo = max(b.i_lo, 0); b.i_hi = min(b.i_hi, W - 1); b.j_lo = max(b.j_lo, 0); b.j_hi = min(b.j_hi, H - 1);
    const int nc = b.j_hi - b.j_lo + 1, nsub = (b.i_hi - b.i_lo + 1) * nc;
    if (nc > 0 && nsub > 0) {
      const FastDiv fds(nc);
      for (int q0 = 0; q0 < nsub; q0 += WAVE) {
        const int q = q0 + lane;
        int r, cc;
        fds.divmod(q, r, cc);
        const int i = b.i_lo + r, j = b.j_lo + cc;
        if (q < nsub) {
          if (view_cell(p, depth2, quick, x0, y0, cone7, (double)i * c.scale, (double)j * c.scale)) seen[i * H + j] = call;
        }
      }
    }
  }
#if defined(D2D_GAZE_ABL) && D2D_GAZE_ABL == 1
  if (lane == 0) act[e] = 0.0;
  return;
#endif
  GZ(1);  // seen pass
  if (n == 0) {  // :118-119
    if (lane == 0) act[e] = 0.0;
    return;
  }
  const double hx = hxy.x, hy = hxy.y;
  const int bi = (int)floor((hx - c.depth) * inv_scale) - 1, bj = (int)floor((hy - c.depth) * inv_scale) - 1;
  // ---- v_i: the swept map inside the box (last write wins = largest waypoint index), :88-90 ----
  for (int q = lane; q < g.ncell; q += WAVE) swi[q] = -1;
  wave_sync_lds();
  {
    // Only waypoints whose cell lies in the box count.  The trajectory is walked in chunks of 64 slots; a chunk whose bounding box
    // (d2d_plan.traj_box, written with the trajectory) does not reach into the box's pixel range holds none and is not loaded: on a
    // 6400 px map the remaining trajectory is up to 1 800 waypoints, the box holds the next few dozen.
    const int c_lo = head >> 6, c_hi = (stored - 1) >> 6;
    const double *tbox = traj_boxes(p, e);
    unsigned long long cmask = 0ull;
    // (short trajectories -- the 500 px maps: a few chunks -- are walked as they are: looking at the boxes costs what it saves)
    const bool chunked = tbox != nullptr && c_hi - c_lo < 64 && c_hi - c_lo >= D2D_TRAJ_PRUNE_CHUNKS;
    if (__builtin_expect(chunked, 0)) {
      bool need = false;
      if (c_lo + lane <= c_hi) {
        const double2 lo = ld2(tbox + 4 * (c_lo + lane)), hi = ld2(tbox + 4 * (c_lo + lane) + 2);
        const double bx0 = (double)bi * c.scale, bx1 = (double)(bi + g.bbn) * c.scale;
        const double by0 = (double)bj * c.scale, by1 = (double)(bj + g.bbn) * c.scale;
        need = (hi.x >= bx0) & (lo.x < bx1) & (hi.y >= by0) & (lo.y < by1);   // bi <= floor(x / scale) < bi + bbn for some waypoint
      }
      cmask = __ballot(need);
    }
    auto visit = [&](int a) {  // slot a of the trajectory buffer
      if (a >= head && a < stored) {
        const int i = a - head;
        const double2 wxy = ld2(traj + (size_t)a * 4);
        const int ci = cell_fast(wxy.x, c.scale, inv_scale), cj = cell_fast(wxy.y, c.scale, inv_scale);
        const int r = ci - bi, cc = cj - bj;
        if (ci >= 0 && ci < W && cj >= 0 && cj < H && r >= 0 && r < g.bbn && cc >= 0 && cc < g.bbn) atomicMax(&swi[r * g.bbn + cc], i);
      }
    };
    if (__builtin_expect(chunked, 0)) {
      for (unsigned long long m = cmask; m; m &= m - 1ull) visit(((c_lo + __ffsll((long long)m) - 1) << 6) + lane);
    } else {
      for (int a0 = head; a0 < stored; a0 += WAVE) visit(a0 + lane);
    }
  }
  wave_sync_global();  // LDS hand-off of the swept map AND the seen map the lanes wrote above
  GZ(2);  // swept map + fence
#if defined(D2D_GAZE_ABL) && D2D_GAZE_ABL == 2
  if (lane == 0) act[e] = 0.0;
  return;
#endif
  // ---- reward (:109-111) and the candidates' view bits.  Only box cells inside the map AND inside the view disk of
  //      the head (d2 <= depth^2, about half of the box) can carry a view bit; every other cell contributes 0 to all
  //      six sums whatever its reward.  Those live cells are compacted first (ballot + prefix count into the `swl`
  //      list), so the expensive part runs on ~4 full passes instead of 7 sparse ones ----
  // Live cells = box cells inside the map, inside the head's view disk AND inside the sector that holds every candidate's view
  // (`w_all` around the drone's current direction); only the box of that sector is walked.  cos^2 of the sector's half angle
  // (double from a float cosine, shrunk by 2 %) for the per-cell pre-test.
  double wide2 = 0.0;
  if (boxes) {
    const double cwd = (double)__cosf(w_all) * 0.98;
    wide2 = cwd * cwd;
  }
  for (int k = lane; k < (g.ncell + 3) / 4; k += WAVE) ((unsigned int *)cm)[k] = 0u;  // no view bits anywhere else in the box
  int nlive = 0;
  {
    CellBox b;
    b.i_lo = bi; b.i_hi = bi + g.bbn - 1; b.j_lo = bj; b.j_hi = bj + g.bbn - 1;
    if (boxes) {
      const CellBox sb = sector_box(hx, hy, c.depth, inv_scale, cone7.cy, cone7.sy, __cosf(w_all), __sinf(w_all));
      b.i_lo = max(b.i_lo, sb.i_lo); b.i_hi = min(b.i_hi, sb.i_hi); b.j_lo = max(b.j_lo, sb.j_lo); b.j_hi = min(b.j_hi, sb.j_hi);
    }
    b.i_lo = max(b.i_lo, 0); b.i_hi = min(b.i_hi, W - 1); b.j_lo = max(b.j_lo, 0); b.j_hi = min(b.j_hi, H - 1);
    const int nc = b.j_hi - b.j_lo + 1, nsub = (nc > 0 && b.i_hi >= b.i_lo) ? (b.i_hi - b.i_lo + 1) * nc : 0;
    const FastDiv fds(nc > 0 ? nc : 1);
    for (int q0 = 0; q0 < nsub; q0 += WAVE) {
      const int qs = q0 + lane;
      int r, cc;
      fds.divmod(qs, r, cc);
      const int i = b.i_lo + r, j = b.j_lo + cc;
      const int q = (i - bi) * g.bbn + (j - bj);  // index in the disk's box: rows of the grid in order, ascending with qs
      bool live = false;
      if (qs < nsub) {
        const double ca = hx - (double)i * c.scale, cb = hy - (double)j * c.scale;
        const double d2 = ca * ca + cb * cb;
        const double dm = (-ca) * cone7.cy + (-cb) * cone7.sy;
        const bool sector = (wide2 <= 0.0) | (d2 <= 0.0) | ((dm > 0.0) & (dm * dm >= wide2 * d2));
        live = (d2 <= depth2) & sector;
      }
      const unsigned long long lm = __ballot(live);
      if (live) swl[nlive + __popcll(lm & ((1ull << lane) - 1ull))] = q;
      nlive += __popcll(lm);
    }
  }
  if (lane == 0) {
    rng[0] = 0;   // box rows / box columns (bit = index in the disk's box, two words each: bbn <= 64) that hold a non-zero term
    rng[1] = 0;
    rng[2] = 0;
    rng[3] = 0;
  }
  wave_sync_lds();
#if defined(D2D_GAZE_ABL) && D2D_GAZE_ABL == 3
  if (lane == 0) act[e] = 0.0;
  return;
#endif
  GZ(3);  // live-cell compaction
#ifndef D2D_GAZE_EXACT_ONLY
  double qa[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};  // this lane's part of every candidate's sum, in ANY order (the quick decision below)
#endif
  bool any_hot = false;  // this lane has met a cell that adds a non-zero term to some candidate's sum
  for (int l0 = 0; l0 < nlive; l0 += 4 * WAVE) {
    // four live cells per lane: their seen-map entries are fetched together, then their table rows, then the arithmetic
    int qq[4], sn[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      qq[u] = swl[min(l0 + u * WAVE + lane, nlive - 1)];
      int r, cc;
      fdb.divmod(qq[u], r, cc);
      sn[u] = seen[(bi + r) * H + (bj + cc)];
    }
    double tobs[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      // time since the cell was seen: k calls ago -> 0 + dt + dt + ... (host table).  The reward needs the value itself only
      // while it is below 1 (the first entries, kept in LDS); a cell never seen starts at 5 (yaw_planner.py:48): stale, reward 1
      const int k = call - min(sn[u], call);
      tobs[u] = sn[u] > 0 ? tobl[min(k, D2D_TOBS_LDS - 1)] : 5.0;
    }
    if (__any(tobl[D2D_TOBS_LDS - 1] < 1.0)) {  // a time step below 1 / 64: the literal table
#pragma unroll
      for (int u = 0; u < 4; ++u) tobs[u] = sn[u] > 0 ? p.tobs_tab[call - min(sn[u], call)] : p.tobs_tab[p.tobs_len + call];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (l0 + u * WAVE + lane < nlive) {
        const int q = qq[u];
        int r, cc;
        fdb.divmod(q, r, cc);
        const int i = bi + r, j = bj + cc;
        const int si = swi[q];
        const double sw = si >= 0 ? (double)si * c.dt : 0.0;
        const bool stale = tobs[u] >= 0.5;
        double rw = (1.0 * tobs[u] < 1.0) ? 1.0 * tobs[u] : 1.0;  // np.clip(c3 * t, -inf, 1)
        rw = ((sw > 3.0) & stale) ? 1000.0 : rw;
        rw = ((sw > 0.0) & (sw <= 3.0) & stale) ? 1000000.0 : rw;
        // the six candidates look from the same point: distance terms once per cell, direction terms per candidate
        const double x = (double)i * c.scale, y = (double)j * c.scale;
        const double ca = hx - x, cb = hy - y;
        const double d2 = ca * ca + cb * cb;
        unsigned int bits = 0, rare = 0;
        if (d2 <= 0.0) {
          bits = (1u << p.n_yaw) - 1u;
        } else {
          const double dxv = x - hx, dyv = y - hy;
          const double rhs = quick * d2, hi = rhs * (1.0 + 1e-12), lo = rhs * (1.0 - 1e-12), top = d2 * (1.0 - 1e-12);
#pragma unroll
          for (int a = 0; a < 7; ++a) {
            if (a < p.n_yaw) {
              const double dot = dxv * vdir[2 * a] + dyv * vdir[2 * a + 1];
              // dot^2 with dot's sign (hi, lo, top are > 0): `dot > 0 and dot^2 > hi` is `sq > hi`, `dot <= 0 or dot^2 < lo` is `sq < lo`
              // -- three compares per candidate instead of five; a degenerate lo == 0 only sends dot == 0 to the literal test below
              const double sq = dot * fabs(dot);
              const bool inside = (quick > 0.0) & (sq > hi) & (sq < top);
              const bool outside = (quick > 0.0) & (sq < lo);
              bits |= inside ? (1u << a) : 0u;
              rare |= (inside | outside) ? 0u : (1u << a);
            }
          }
          if (rare) {  // within 1e-12 of a cone's edge or axis (or a cone the shortcut does not cover): the literal test
            const double rs = sqrt(d2);
            for (unsigned int m = rare; m; m &= m - 1) {
              const int a = __ffs((int)m) - 1;
              const double dot = dxv * vdir[2 * a] + dyv * vdir[2 * a + 1];
              bits |= acos_le(p, dot / rs) ? (1u << a) : 0u;
            }
          }
        }
        // A cell adds view * reward to a sum: nothing unless it is seen by some candidate AND its reward is non-zero (rewards
        // are >= +0.0, and x + 0.0 == x), and the cells the drone sees right now have reward 0 -- most of every candidate's
        // view.  Only the rows that hold such a cell take part in the sums below.
        const bool hot = (bits != 0u) & (rw != 0.0);
        rew[q] = rw;
        cm[q] = hot ? (unsigned char)bits : (unsigned char)0;
#ifndef D2D_GAZE_EXACT_ONLY
        {
          const int hbits = hot ? (int)bits : 0;
#pragma unroll
          for (int a = 0; a < 7; ++a) {  // + view * reward, view in {0.0, 1.0} (as in the block sums below)
            const double view = __hiloint2double(((hbits << (31 - a)) >> 31) & 0x3ff00000, 0);
            qa[a] = __builtin_fma(rw, view, qa[a]);
          }
        }
#endif
        any_hot = any_hot | hot;
      }
    }
  }
  wave_sync_lds();
#if defined(D2D_GAZE_ABL) && D2D_GAZE_ABL == 4
  if (lane == 0) act[e] = 0.0;
  return;
#endif
  // ---- np.sum(view * reward) per candidate in numpy's pairwise order (:123) ----
  // lane = one accumulator chain (block, r) of the blocks the box touches, carrying the partial sums of ALL six
  // candidates: the chain's elements off + r + 8 k that lie in the box (at most ~12 at the default geometry: a block spans
  // <= 4 grid rows, a box row holds <= 3 cells of a residue) are read once (view bits + reward) and added to the candidates whose bit
  // is set.  The eight chains of a block are eight neighbouring lanes: ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) by three
  // shuffles per candidate.  (Block offsets are multiples of 8, so g % 8 == r.)  ~800 instructions instead of the
  // ~2000 of a (candidate, r) mapping whose lanes walk every slot of every block.
  GZ(4);  // rewards + candidate bits
  const FastDiv fdh(H);
  if (!__any(any_hot)) {  // every sum is 0: `max_reward < 0` never holds, the first candidate stays (yaw_planner.py:116-125)
    if (lane == 0) act[e] = ys_l / p.yaw_rate_max;
    return;
  }
#ifndef D2D_GAZE_EXACT_ONLY
  // ---- the quick decision (round 4).  The argmax over the candidates needs numpy's pairwise sums (below: a third of the stage) only
  //      when two of them are close.  Every candidate's sum is first formed in ANY order (per-lane partial sums above, LDS atomic
  //      adds here).  A sum of n non-negative terms carries a relative error of at most (n - 1) u in any order of addition (u = 2^-53,
  //      n < 4096 cells of the box: below 5e-13), numpy's pairwise order included.  So a candidate more than 1e-9 M below the
  //      largest quick sum M is STRICTLY below it in numpy's sums too -- it cannot be the argmax -- and
  //        * one contender left (four steps in five): it is the reference's choice;
  //        * several contenders that see exactly the same cells with a non-zero term (one step in five: neighbouring yaw rates whose
  //          views differ only in cells the drone sees right now): their numpy sums are the same terms at the same positions,
  //          bit-identical, and the reference's strict `<` keeps the first of them;
  //        * anything else (3 % of the steps): the exact sums below decide, as before.
  //      Measured on 100 000 plan() calls of the oracle's closed loop (profiles/r04/NOTES.md).  -DD2D_GAZE_EXACT_ONLY builds the
  //      stage without this block (csrc/libd2d_hip_exact.so: the suite runs the exact path on every step through it). ----
  {
    double *qs = stk;  // [8]: the view directions kept there have been consumed by the reward pass
    if (lane < 8) qs[lane] = 0.0;
#pragma unroll
    for (int a = 0; a < 7; ++a)
      if (qa[a] != 0.0) atomicAdd(&qs[a], qa[a]);  // (LDS operations of one wave execute in order: behind the clear above)
    wave_sync_lds();
    const double q_l = qs[min(lane, 7)];
    const bool cand = lane < p.n_yaw;
    const double M = wave_fmax(cand ? q_l : 0.0);  // > 0: some cell carries a non-zero term
    const unsigned int C = (unsigned int)__ballot(cand & (q_l >= M - 1e-9 * M));
    bool bad = false;
    if (__popc(C) > 1) {
      for (int l = lane; l < nlive; l += WAVE) {
        const unsigned int t = (unsigned int)cm[swl[l]] & C;
        bad = bad | ((t != 0u) & (t != C));
      }
    }
    if (!__any(bad)) {
      const double ys_best = shfl_f64(ys_l, __ffs((int)C) - 1);
      if (lane == 0) act[e] = ys_best / p.yaw_rate_max;  // :127
      return;
    }
    wave_sync_lds();  // (the exact path re-uses `stk`)
  }
#endif
  // the box rows / columns that hold a non-zero term (`rng`, cleared above), from the view bits the reward pass left: only the exact
  // sums need them, i.e. the few steps the quick decision does not settle
  // the pairwise plan of the map (block table + addition tree) into LDS: needed from here on only
  if (!g.sparse) {
    for (int k = lane; k < 4 * p.pw_nleaf; k += WAVE) pwl[k] = p.pw_leaf[k];
    for (int k = lane; k < p.pw_ntree; k += WAVE) pwp[k] = p.pw_tree[k];
  }
  // (and the reward plane outside the cells with a view bit is cleared: the sums multiply by the view bit, and 0 x a stale NaN / inf of
  // another phase's bytes would not be 0)
  for (int k = lane; k < g.ncell; k += WAVE)
    if (cm[k] == 0) rew[k] = 0.0;
  for (int l = lane; l < nlive; l += WAVE) {
    const int q = swl[l];
    if (cm[q] != 0) {
      int r, cc;
      fdb.divmod(q, r, cc);
      atomicOr((unsigned int *)&rng[r < 32 ? 0 : 1], 1u << (r & 31));
      atomicOr((unsigned int *)&rng[cc < 32 ? 2 : 3], 1u << (cc & 31));
    }
  }
  wave_sync_lds();
  const unsigned long long hrows = ((unsigned long long)(unsigned int)rng[1] << 32) | (unsigned int)rng[0];
  const unsigned long long hcols = ((unsigned long long)(unsigned int)rng[3] << 32) | (unsigned int)rng[2];
  // rows / columns of the grid that hold a non-zero term (they lie inside the box and inside the map)
  const int row_lo = bi + __ffsll((long long)hrows) - 1, row_hi = bi + 63 - __clzll((long long)hrows);
  const int jlo = bj + __ffsll((long long)hcols) - 1, jhi = bj + 64 - __clzll((long long)hcols);
  for (int k = lane; k < p.n_yaw * nnode; k += WAVE) lsum[k] = 0.0;  // blocks without a non-zero term sum to 0
  // the blocks that cover a row with a non-zero term, compacted in order: lane = block (their row ranges are in LDS)
  int *hlist = swi;  // the swept map is not needed any more
  int nhl = 0;
  if (g.sparse) {
    // The blocks that can hold a non-zero term: for every box row with one, the block of its first and of its last column with one
    // (a block has >= 64 cells, the columns span < 64: there is none in between).  numpy's recursion (loops_utils.h.src pairwise_sum:
    // n <= 128 is a block, else split at n / 2 rounded down to a multiple of 8) is walked down from the root for that cell: the block
    // [off, off + m), its depth and its path (one bit per split, 1 = right half).  Lane = (row, first / last), in cell order.
    const int r = lane >> 1, i = row_lo + r;
    const bool on = i <= row_hi && ((hrows >> (i - bi)) & 1ull) != 0ull;
    const int gcell = i * H + ((lane & 1) ? jhi - 1 : jlo);
    int off = 0, m = W * H, depth = 0, path = 0;
    while (__any(on && m > 128)) {
      if (m > 128) {
        int m2 = m >> 1;
        m2 -= m2 & 7;
        const bool right = gcell >= off + m2;
        off = right ? off + m2 : off;
        m = right ? m - m2 : m2;
        path = (path << 1) | (right ? 1 : 0);
        depth += 1;
      }
    }
    // the same block twice (first and last column in one block, or a block that spans two rows): lanes are in cell order, so equal
    // blocks are neighbours among the lanes that are on
    const unsigned long long om = __ballot(on);
    const unsigned long long below = om & ((1ull << lane) - 1ull);
    const int prev = below ? 63 - __clzll((long long)below) : lane;
    const int off_prev = __shfl(off, prev, WAVE);
    const bool isnew = on && (below == 0ull || off_prev != off);
    const unsigned long long nm = __ballot(isnew);
    nhl = __popcll(nm);  // <= D2D_GAZE_HOT: 2 lanes per row, at most 32 rows (plan_check: bbn <= 32 on such maps)
    if (isnew) {
      const int k = __popcll(nm & ((1ull << lane) - 1ull));
      int i_first, i_last, dummy;
      fdh.divmod_big(off, i_first, dummy);
      fdh.divmod_big(off + m - 1, i_last, dummy);
      pwl[4 * k] = off;
      pwl[4 * k + 1] = m;
      pwl[4 * k + 2] = i_first;
      pwl[4 * k + 3] = i_last;
      pwp[k] = depth;
      pwp[D2D_GAZE_HOT + k] = path;
      hlist[k] = k;
    }
  }
  for (int l0 = 0; !g.sparse && l0 < p.pw_nleaf; l0 += WAVE) {
    const int lf = min(l0 + lane, p.pw_nleaf - 1);
    const int i_first = pwl[4 * lf + 2], i_last = pwl[4 * lf + 3];
    const int lo = max(max(i_first, row_lo) - bi, 0), hi = min(min(i_last, row_hi) - bi, 63);
    const bool hotl = (l0 + lane < p.pw_nleaf) && lo <= hi && ((hrows >> lo) & ((hi - lo >= 63) ? ~0ull : ((1ull << (hi - lo + 1)) - 1ull))) != 0ull;
    const unsigned long long hm = __ballot(hotl);
    if (hotl) hlist[nhl + __popcll(hm & ((1ull << lane) - 1ull))] = lf;
    nhl += __popcll(hm);
  }
  wave_sync_lds();
  GZ(5);  // hot blocks
  const int nchain = 8 * nhl;
  const int per_row = (min(jhi - jlo, g.bbn) + 7 + 7) >> 3;  // cells of one residue among the columns with a non-zero term, at most
  const int max_span = 127 / H + 2;      // grid rows a block of <= 128 cells can touch
  // the walk for NA candidates (6 at the reference's yaw space, 7 at most): view bits of candidates >= n_yaw are never set
  auto walk = [&](auto na_tag) {
    constexpr int NA = decltype(na_tag)::value;
    for (int c0 = 0; c0 < nchain; c0 += WAVE) {
      const int ch = c0 + lane;
      const bool live = ch < nchain;
      const int lf = hlist[min(ch, nchain - 1) / 8], r_of = lane & 7;
      const int off = pwl[4 * lf], m = pwl[4 * lf + 1], i_first = pwl[4 * lf + 2], i_last = pwl[4 * lf + 3];
      const int gend = off + (m - (m & 7));  // the last m % 8 elements of a block are added after its fold
      double acc[NA];
#pragma unroll
      for (int a = 0; a < NA; ++a) acc[a] = 0.0;
      // uniform bounds, predicated body: every lane makes the same number of trips
      const int i_lo = max(i_first, row_lo);
      // rows to walk: the largest span among the blocks of this pass (wave-uniform), usually 3 of the possible 4
      const int my_span = live ? min(i_last, row_hi) - i_lo + 1 : 0;
      int span = 1;
      for (int k = 2; k <= max_span; ++k) span = __any(my_span >= k) ? k : span;
      for (int di = 0; di < span; ++di) {
        const int i = i_lo + di;
        const bool row_on = live & (i <= min(i_last, row_hi));
        const int glo = max(off, i * H + jlo), ghi = min(gend, i * H + jhi);
        const int rowbase = (i - bi) * g.bbn - i * H - bj;  // box index = rowbase + g
        const int g0 = glo + ((r_of - glo) & 7);
        for (int k = 0; k < per_row; ++k) {
          const int gq = g0 + 8 * k;
          const bool on = row_on & (gq < ghi);
          const int q = min(max(rowbase + gq, 0), g.ncell - 1);
          const int bits = on ? (int)cm[q] : 0;
          const double rw = rew[q];  // finite everywhere in the box (zeroed above): 0 * rw == +0.0
#pragma unroll
          for (int a = 0; a < NA; ++a) {
            // acc + view * reward with view in {0.0, 1.0}: the product is exact (the reward itself, or +0.0 which changes nothing),
            // so the fused multiply-add rounds once, exactly like the reference's add -- three instructions per candidate
            // (bit field, and, fma) instead of five (64-bit mask, two ands, add)
            const double view = __hiloint2double(((bits << (31 - a)) >> 31) & 0x3ff00000, 0);
            acc[a] = __builtin_fma(rw, view, acc[a]);
          }
        }
      }
#pragma unroll
      for (int a = 0; a < NA; ++a) {  // ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7))
        double v = acc[a] + row_shl_f64<1>(acc[a]);
        v = v + row_shl_f64<2>(v);
        v = v + row_shl_f64<4>(v);
        if ((m & 7) != 0 && r_of == 0 && live)
          for (int k = m - (m & 7); k < m; ++k) {  // numpy adds the block's last m % 8 elements one by one after the fold
            int gi, gj;
            fdh.divmod_big(off + k, gi, gj);
            const int r = gi - bi, cc = gj - bj;
            double x = 0.0;
            if (r >= 0 && r < g.bbn && cc >= 0 && cc < g.bbn && ((cm[r * g.bbn + cc] >> a) & 1)) x = rew[r * g.bbn + cc];
            v += x;
          }
        if (r_of == 0 && live && a < p.n_yaw) lsum[a * nnode + lf] = v;
      }
    }
  };
  if (p.n_yaw <= 6) walk(std::integral_constant<int, 6>{});
  else walk(std::integral_constant<int, 7>{});
  wave_sync_lds();
  GZ(6);  // pairwise block sums
#if defined(D2D_GAZE_ABL) && D2D_GAZE_ABL == 5
  if (lane == 0) act[e] = 0.0;
  return;
#endif
  // ---- the blocks' sums added in the recursion's order, level by level: the additions of one level are independent
  //      (lane = (candidate, addition)), every single one keeps numpy's left + right; argmax below (:116-125) ----
  if (g.sparse) {
    // lane = hot block, in cell order = the order of the recursion's leaves.  Deepest level first: two blocks whose paths differ in
    // the last bit only are the two halves of one split -- left + right, numpy's operand order; a half without a partner has a
    // sibling that sums to +0.0, and x + 0.0 == x == 0.0 + x for the x >= +0.0 there are: it moves up unchanged.
    bool alive = lane < nhl;
    int depth = alive ? pwp[lane] : 0, path = alive ? pwp[D2D_GAZE_HOT + lane] : 0;
    double val[7];
#pragma unroll
    for (int a = 0; a < 7; ++a) val[a] = (alive && a < p.n_yaw) ? lsum[a * nnode + lane] : 0.0;
    int maxd = depth;
    for (int o = 32; o > 0; o >>= 1) maxd = max(maxd, __shfl_xor(maxd, o, WAVE));
    for (int D = maxd; D >= 1; --D) {
      const unsigned long long am = __ballot(alive);
      const unsigned long long above = lane < 63 ? (am >> (lane + 1)) : 0ull;
      const int nx = above ? lane + 1 + (__ffsll((long long)above) - 1) : lane;
      const int dn = __shfl(depth, nx, WAVE), pn = __shfl(path, nx, WAVE);
      const bool adv = alive && depth == D;
      const bool left = adv && above != 0ull && dn == D && (pn >> 1) == (path >> 1) && (path & 1) == 0 && (pn & 1) == 1;
      const unsigned long long lm = __ballot(left);
      if (__builtin_amdgcn_readfirstlane((int)(lm != 0ull))) {
#pragma unroll
        for (int a = 0; a < 7; ++a) {
          if (a < p.n_yaw) {
            const double vn = shfl_f64(val[a], nx);
            if (left) val[a] = val[a] + vn;
          }
        }
      }
      // the right half of a pair is done: it is the lane whose previous live lane added it
      const unsigned long long below = am & ((1ull << lane) - 1ull);
      const int pv = below ? 63 - __clzll((long long)below) : 0;
      const bool right = alive && below != 0ull && ((lm >> pv) & 1ull) != 0ull;
      if (adv) {
        depth -= 1;
        path >>= 1;
      }
      if (right) alive = false;
    }
    // the root is the first hot block's lane
    int best = 0;
    double max_reward = 0.0;
    for (int a = 0; a < p.n_yaw; ++a) {
      double r = 0.0;
#pragma unroll
      for (int b = 0; b < 7; ++b) r = (b == a) ? val[b] : r;
      r = shfl_f64(r, 0);
      if (max_reward < r) {
        best = a;
        max_reward = r;
      }
    }
    const double ys_best = shfl_f64(ys_l, best);  // (every lane takes part: a lane read inside `lane == 0` would find its source masked off)
    if (lane == 0) act[e] = ys_best / p.yaw_rate_max;  // :127
    return;
  }
  // A tree of height h with 2^h blocks (at most 32) is the perfect one, its blocks in cell order: the 50 x 50 map's (32 blocks of 78 /
  // 79 cells), every map whose recursion halves evenly.  Its additions are those of a butterfly over neighbouring lanes: lane = block
  // (+ 32 for the odd candidates), four DPP row shifts and one lane read per candidate pair -- no LDS round trip per level, no table of
  // additions.  Lanes past the last block hold +0.0 (x + 0.0 == x for the x >= +0.0 there are).
  const int nlev_tree = p.pw_ntree - 3 * p.pw_nleaf;  // pw_tree = [n_levels, root, level_start[n_levels + 1], 3 ints per addition]
  if (nlev_tree >= 0 && nlev_tree <= 5 && p.pw_nleaf == (1 << nlev_tree)) {
    const int leaf = lane & 31, half = lane >> 5;
    double v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int a = 2 * j + half;
      const bool on = (a < p.n_yaw) & (leaf < p.pw_nleaf);
      const double x = lsum[min(a, p.n_yaw - 1) * nnode + min(leaf, p.pw_nleaf - 1)];
      v[j] = on ? x : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // left + right at every level: the lower lane is the left operand
      double x = v[j];
      x = x + row_shl_f64<1>(x);
      x = x + row_shl_f64<2>(x);
      x = x + row_shl_f64<4>(x);
      x = x + row_shl_f64<8>(x);
      v[j] = x;
    }
    int best = 0;
    double max_reward = 0.0;
#pragma unroll
    for (int a = 0; a < 7; ++a) {
      if (a < p.n_yaw) {
        const int l0 = 32 * (a & 1);
        const double r = readlane_f64(v[a >> 1], l0) + readlane_f64(v[a >> 1], l0 + 16);  // blocks 0..15 + blocks 16..31
        if (max_reward < r) {  // :116-125
          best = a;
          max_reward = r;
        }
      }
    }
    const double ys_best = shfl_f64(ys_l, best);
    if (lane == 0) act[e] = ys_best / p.yaw_rate_max;  // :127
    GZ(7);  // tree + argmax
    return;
  }
  {
    const int nlev = pwp[0];
    const int *ops = pwp + 3 + nlev;
    // the level starts once (lane lv holds level_start[lv]: a lane read per level instead of an LDS round trip), and item k of a
    // level = (addition k / n_yaw, candidate k % n_yaw): one divisor for all levels, set up outside the loop
    const int lstart_l = pwp[2 + min(lane, nlev)];
    const FastDiv fdy(p.n_yaw);
    for (int lv = 0; lv < nlev; ++lv) {
      const int o0 = __builtin_amdgcn_readlane(lstart_l, lv), cnt = __builtin_amdgcn_readlane(lstart_l, lv + 1) - o0;
      for (int k = lane; k < p.n_yaw * cnt; k += WAVE) {
        int a, ko;
        fdy.divmod(k, ko, a);
        const int o = o0 + ko;
        const int dst = ops[3 * o], lft = ops[3 * o + 1], rgt = ops[3 * o + 2];
        lsum[a * nnode + dst] = lsum[a * nnode + lft] + lsum[a * nnode + rgt];
      }
      wave_sync_lds();
    }
  }
  double total = 0.0;
  if (lane < p.n_yaw) total = lsum[lane * nnode + pwp[1]];
  int best = 0;
  double max_reward = 0.0;
  for (int a = 0; a < p.n_yaw; ++a) {
    const double r = shfl_f64(total, a);
    if (max_reward < r) {
      best = a;
      max_reward = r;
    }
  }
  const double ys_best = shfl_f64(ys_l, best);
  if (lane == 0) act[e] = ys_best / p.yaw_rate_max;  // :127
  GZ(7);  // tree + argmax
}

// mode: 0 plain, 1 reset the envs that are done first, 2 leave the envs that are done untouched
__global__ __launch_bounds__(WAVE *WAVES_PER_BLOCK) void k_gaze(d2d_cfg c, d2d_state s, d2d_plan p, d2d_state init, int mode) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
  const int e = blockIdx.x * (int)(blockDim.x / WAVE) + wv;
  if (e >= c.B) return;
  if (mode == 2 && s.flags[(size_t)e * 4 + D2D_F_DONE] != 0) return;
  gaze_env(c, s, p, init, mode == 1, e, lane, d2d_lds + (size_t)wv * gaze_geom(c, p).wave_bytes);
}

__global__ __launch_bounds__(WAVE *WAVES_PER_BLOCK) void k_plan_reset(d2d_cfg c, d2d_plan p, const unsigned char *mask,
                                                                      int mask_stride) {
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  const int e = blockIdx.x * WAVES_PER_BLOCK + wv;
  if (e >= c.B) return;
  if (mask && !mask[(size_t)e * mask_stride]) return;
  plan_reset_env(c, p, (size_t)e, lane);
}

__global__ void k_sincos(const double *in, double *so, double *co, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    // through the routine the gaze stage uses, with sines and cosines mixed among the lanes of a wave as they are there
    const int odd = (int)(threadIdx.x & 1);
    const double first = d2d_sin_or_cos(in[i], odd), second = d2d_sin_or_cos(in[i], odd ^ 1);
    so[i] = odd ? second : first;
    co[i] = odd ? first : second;
  }
}
