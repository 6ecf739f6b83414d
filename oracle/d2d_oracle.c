/*
 * d2d_oracle.c — CPU restatement of the reference's Drone2DEnv2.step hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker for the HIP path, never the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so.  Nothing under
 * gym-drone2d-activeperception_amd/ imports, links or executes it.
 *
 * Pinned against the reference itself: the tests/golden .npz fixtures were produced by importing the reference's
 * Python (tests/golden/make_golden.py) and tests/test_oracle_golden.py replays every trace through this
 * file, requiring grids / hit masks / flags / counters / observations AND the fp64 agent and drone state to be
 * bit-exact (tolerance 0); only the Kalman tracker state is compared to 1e-6 (the reference computes it through
 * BLAS / LAPACK, whose summation order is not reproducible).
 *
 * Scalar, one env after another, one ray after another, in the reference's own order of operations.
 * Plain C, libm for tan/sqrt/fmod/floor (the same libm the reference's math.tan resolves to).
 * Build: oracle/Makefile (-O2 -ffp-contract=off: no fused multiply-adds the reference does not have).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/d2d.h"

static __thread char g_err[256];

static int fail(int code, const char *msg) {
  snprintf(g_err, sizeof g_err, "%s", msg);
  return code;
}

static int g_threads = 1;
/* number of host threads run_stages spreads envs over (default 1 = scalar) */
void d2d_oracle_set_threads(int n) { g_threads = n > 0 ? n : 1; }

int d2d_oracle_abi_version(void) { return D2D_ABI_VERSION; }
const char *d2d_oracle_last_error(void) { return g_err; }

/* Python / numpy float floor division a // b for b > 0 (CPython float_floor_div, numpy npy_divmod). */
static double py_floordiv(double a, double b) {
  double mod = fmod(a, b);
  double div = (a - mod) / b;
  if (mod != 0.0) {
    if ((b < 0.0) != (mod < 0.0)) div -= 1.0;
  }
  if (div != 0.0) {
    double fl = floor(div);
    if (div - fl > 0.5) fl += 1.0;
    return fl;
  }
  return copysign(0.0, a / b);
}

/* Python float modulo a % b for b > 0 (utils.py:743 `% 360`). */
static double py_mod(double a, double b) {
  double mod = fmod(a, b);
  if (mod != 0.0) {
    if ((b < 0.0) != (mod < 0.0)) mod += b;
  } else {
    mod = copysign(0.0, b);
  }
  return mod;
}

static int cell_of(double v, double scale) { return (int)py_floordiv(v, scale); }

/* numpy.linalg.norm of a 2-vector: sqrt(x.dot(x)), and the bundled OpenBLAS ddot evaluates the two-element dot
 * product as fma(x1, x1, x0 * x0) on FMA hardware (measured against numpy: 20000/20000 bit-identical; the unfused
 * sum matches only 92 %).  Call sites: utils.py:476,756,760,774, envs/drone_v2.py:223-224, traj_planner.py:58,88,158,172,228. */
static double norm2(double x, double y) { return sqrt(fma(y, y, x * x)); }

/* Test diagnostics (tests/golden/live_sweep.py): the smallest |distance - threshold| any tracker test of the planner has seen.
 * The Kalman state matches numpy's LAPACK to 1e-6 only (one reciprocal here, `inv` there), so a plan could differ from the
 * reference's where a tracker distance sits within 1e-6 of its threshold: the sweep records how close it ever gets. */
static double g_min_margin = 1e300;
static void note_margin(double dist, double lim) {
  const double m = fabs(dist - lim);
  if (m < g_min_margin) {
#pragma omp critical(d2d_margin)
    if (m < g_min_margin) g_min_margin = m;
  }
}
double d2d_oracle_debug_min_margin(int reset) {
  const double m = g_min_margin;
  if (reset) g_min_margin = 1e300;
  return m;
}

typedef struct env_view {
  const d2d_cfg *c;
  int e;
  double *ag;      /* [AF][N] */
  int32_t *unit;   /* [N] */
  int32_t *prev;   /* [N][3] */
  uint8_t *gt, *dm;
  double *dr;      /* [DF] */
  double *tgt;     /* [2] */
  double *tlist;   /* [T][2] */
  int32_t *cnt;    /* [CF] */
  uint8_t *active; /* [N] */
  double *kf;      /* [N][KF] */
  int32_t *kflen;  /* [N] */
  uint8_t *hit;
  int32_t *newly;
  uint8_t *flags;
  uint8_t *obs;
  float *obs_yaw;
} env_view;

static env_view view(const d2d_cfg *c, const d2d_state *s, int e) {
  env_view v;
  size_t N = (size_t)c->N, WH = (size_t)c->W * c->H;
  v.c = c;
  v.e = e;
  v.ag = s->agents + (size_t)e * D2D_AF * N;
  v.unit = s->agent_unit + (size_t)e * N;
  v.prev = s->dyn_prev + (size_t)e * N * 3;
  v.gt = s->gt + (size_t)e * WH;
  v.dm = s->dmap + (size_t)e * WH;
  v.dr = s->drone + (size_t)e * D2D_DF;
  v.tgt = s->target + (size_t)e * 2;
  v.tlist = s->targets + (size_t)e * c->T * 2;
  v.cnt = s->counters + (size_t)e * D2D_CF;
  v.active = s->active + (size_t)e * N;
  v.kf = s->kf ? s->kf + (size_t)e * N * D2D_KF : 0;
  v.kflen = s->kf_len ? s->kf_len + (size_t)e * N : 0;
  v.hit = s->hit + (size_t)e * N;
  v.newly = s->newly + e;
  v.flags = s->flags + (size_t)e * 4;
  v.obs = s->obs_local + (size_t)e * c->L * c->L;
  v.obs_yaw = s->obs_yaw + e;
  return v;
}

/* envs/drone_v2.py:153-163 */
static void st_fsm(env_view *v) {
  v->cnt[D2D_C_STEPS] += 1;
  if (v->cnt[D2D_C_SM] == D2D_SM_GOAL_REACHED) v->cnt[D2D_C_SM] = D2D_SM_WAIT_FOR_GOAL;
  if (v->cnt[D2D_C_SM] == D2D_SM_WAIT_FOR_GOAL) {
    int k = v->cnt[D2D_C_TGT_NEXT];
    if (k < v->cnt[D2D_C_NTGT]) { /* the reference raises IndexError on an empty list; callers stop at done */
      v->tgt[0] = v->tlist[2 * k];
      v->tgt[1] = v->tlist[2 * k + 1];
      v->cnt[D2D_C_TGT_NEXT] = k + 1;
    }
    v->cnt[D2D_C_SM] = D2D_SM_PLANNING;
  }
}

/* envs/drone_v2.py:176-179 + utils.py:472-493 (CVM: velocity IS pref_velocity, the same ndarray) */
static void st_agents(env_view *v) {
  const d2d_cfg *c = v->c;
  int N = c->N;
  double *px = v->ag + D2D_A_PX * N, *py = v->ag + D2D_A_PY * N, *vx = v->ag + D2D_A_VX * N,
         *vy = v->ag + D2D_A_VY * N, *rr = v->ag + D2D_A_R * N;
  const double cs = 0x1.bb67ae8584cabp-1 /* cos(pi/6) */, sn = 0x1.fffffffffffffp-2 /* sin(pi/6) */;
  for (int k = 0; k < N; ++k) {
    double velx = vx[k], vely = vy[k]; /* agent.velocity (alias of pref) */
    double nx = px[k] + velx * c->dt, ny = py[k] + vely * c->dt;
    int aliased = 1;
    double pvx = velx, pvy = vely;
    if (norm2(velx, vely) <= 5.0) {
      /* pref_velocity rebound to a fresh array: Rot(+30 deg) @ pref; velocity keeps the old array */
      /* numpy's 2x2 @ 2x1 goes through the bundled OpenBLAS dgemv, which evaluates each row as
       * fma(M[r][0], v0, M[r][1] * v1) on FMA hardware (measured: 40000/40000 rows bit-identical;
       * the unfused form matches only 72 %).  Only reached by agents slower than 5 px/s. */
      double rx = fma(cs, velx, (-sn) * vely);
      double ry = fma(sn, velx, cs * vely);
      pvx = rx;
      pvy = ry;
      aliased = 0;
    }
    if (nx < c->scale + rr[k]) pvx = fabs(pvx);
    else if (nx > c->W_px - c->scale - rr[k]) pvx = -fabs(pvx);
    if (ny < c->scale + rr[k]) pvy = fabs(pvy);
    else if (ny > c->H_px - c->scale - rr[k]) pvy = -fabs(pvy);
    double ux = aliased ? pvx : velx, uy = aliased ? pvy : vely; /* what self.velocity holds at :493 */
    px[k] = px[k] + ux * c->dt;
    py[k] = py[k] + uy * c->dt;
    vx[k] = pvx;
    vy[k] = pvy;
  }
}

/* utils.py:612-618 */
static double positive_angle(double a) {
  const double two_pi = M_PI * 2;
  a = copysign(fmod(fabs(a), two_pi), a);
  if (a < 0) a += two_pi;
  return a;
}

/* utils.py:593-609, 620-713 */
static void st_raycast(env_view *v) {
  const d2d_cfg *c = v->c;
  int N = c->N, H = c->H;
  const double *px = v->ag + D2D_A_PX * N, *py = v->ag + D2D_A_PY * N, *r2 = v->ag + D2D_A_R2 * N;
  const double rad90 = 90 * (M_PI / 180.0), rad270 = 270 * (M_PI / 180.0);
  const double ss = c->scale - 1; /* x_step_size, utils.py:621 */
  const double x0 = v->dr[D2D_D_X], y0 = v->dr[D2D_D_Y];
  const double player_angle = M_PI * 2 - v->dr[D2D_D_YAW] * (M_PI / 180.0);
  memset(v->hit, 0, (size_t)N);
  for (int i = 0; i < c->R; ++i) {
    double ang = positive_angle(player_angle + (c->ray_off0 + c->ray_dth * i));
    int faced_right = (ang < rad90 || ang > rad270);
    int faced_up = (ang > M_PI);
    double slope = tan(ang);
    double xs, ys;
    if (fabs(slope) > 1) {
      slope = 1 / slope;
      ys = faced_up ? -ss : ss;
      xs = ys * slope;
    } else {
      xs = faced_right ? ss : -ss;
      ys = xs * slope;
    }
    double x = x0, y = y0;
    while (0 < x && x < c->W_px && 0 < y && y < c->H_px) {
      int ci = cell_of(x, c->scale), cj = cell_of(y, c->scale);
      int any = 0;
      for (int k = 0; k < N; ++k) {
        double dx = px[k] - x, dy = py[k] - y;
        if (dx * dx + dy * dy <= r2[k]) {
          v->hit[k] = 1;
          any = 1;
        }
      }
      if (any) break;
      uint8_t wall = v->gt[(size_t)ci * H + cj];
      double dist = (x - x0) * (x - x0) + (y - y0) * (y - y0);
      if (wall == D2D_OCCUPIED || dist >= c->depth * c->depth) {
        if (wall == D2D_OCCUPIED) v->dm[(size_t)ci * H + cj] = D2D_OCCUPIED;
        break;
      }
      v->dm[(size_t)ci * H + cj] = D2D_UNOCCUPIED;
      x = x + xs;
      y = y + ys;
    }
  }
  int newly = 0;
  for (int k = 0; k < N; ++k)
    if (v->hit[k] && !v->active[k]) ++newly;
  *v->newly = newly;
  v->cnt[D2D_C_TRACKED] += newly;
}

/* utils.py:527-540.  dynamic_idx == the set of DYNAMIC cells, all of which lie inside the blocks kept
 * in dyn_prev (host init makes the first block wide enough for the circle rasterisation of :521-525). */
static void st_dyngrid(env_view *v) {
  const d2d_cfg *c = v->c;
  int N = c->N, W = c->W, H = c->H;
  const double *px = v->ag + D2D_A_PX * N, *py = v->ag + D2D_A_PY * N;
  for (int k = 0; k < N; ++k) {
    int cx = v->prev[3 * k], cy = v->prev[3 * k + 1], u = v->prev[3 * k + 2];
    for (int i = (cx - u > 0 ? cx - u : 0); i < (cx + u + 1 < W ? cx + u + 1 : W); ++i)
      for (int j = (cy - u > 0 ? cy - u : 0); j < (cy + u + 1 < H ? cy + u + 1 : H); ++j)
        if (v->gt[(size_t)i * H + j] == D2D_DYNAMIC) v->gt[(size_t)i * H + j] = D2D_UNOCCUPIED;
  }
  for (int k = 0; k < N; ++k) {
    int u = v->unit[k];
    int cx = cell_of(px[k], c->scale), cy = cell_of(py[k], c->scale);
    for (int i = (cx - u > 0 ? cx - u : 0); i < (cx + u + 1 < W ? cx + u + 1 : W); ++i)
      for (int j = (cy - u > 0 ? cy - u : 0); j < (cy + u + 1 < H ? cy + u + 1 : H); ++j)
        if (v->gt[(size_t)i * H + j] != D2D_OCCUPIED) v->gt[(size_t)i * H + j] = D2D_DYNAMIC;
    v->prev[3 * k] = cx;
    v->prev[3 * k + 1] = cy;
    v->prev[3 * k + 2] = u;
  }
}

/* ---- Kalman trackers, utils.py:172-275 (4-state constant velocity, F hard-codes 0.1) ---- */
static void kf_reset(double *kf, int32_t *len) { /* KalmanFilter.__init__ defaults, utils.py:181-198 */
  memset(kf, 0, sizeof(double) * D2D_KF);
  kf[4 + 0] = 1;
  kf[4 + 5] = 1;
  kf[4 + 10] = 10;
  kf[4 + 15] = 10;
  *len = 1;
}

static void mat4_mul(const double *A, const double *B, double *C) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += A[4 * i + k] * B[4 * k + j];
      C[4 * i + j] = s;
    }
}

/* predict(), utils.py:225-240; returns 1 when the filter was archived + reset */
static int kf_predict(env_view *v, int k) {
  const d2d_cfg *c = v->c;
  double *mu = v->kf + (size_t)k * D2D_KF, *S = mu + 4;
  static const double F[16] = {1, 0, 0.1, 0, 0, 1, 0, 0.1, 0, 0, 1, 0, 0, 0, 0, 1};
  static const double Ft[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0.1, 0, 1, 0, 0, 0.1, 0, 1};
  double q = (c->sigma != 0) ? 0.1 : 0.001;
  double m2[4], FS[16], P[16];
  for (int i = 0; i < 4; ++i) {
    double s = 0;
    for (int j = 0; j < 4; ++j) s += F[4 * i + j] * mu[j];
    m2[i] = s;
  }
  mat4_mul(F, S, FS);
  mat4_mul(FS, Ft, P);
  for (int i = 0; i < 4; ++i) P[5 * i] += q;
  memcpy(mu, m2, sizeof m2);
  memcpy(S, P, sizeof P);
  v->kflen[k] += 1;
  if (P[0] >= 150 || !(c->kf_lo_x < m2[0] && m2[0] < c->kf_hi_x) || !(c->kf_lo_y < m2[1] && m2[1] < c->kf_hi_y)) {
    v->cnt[D2D_C_BUF_N] += 1; /* achieved_filter -> tracker_buffer (drone_v2.py:187) */
    v->cnt[D2D_C_BUF_TS] += v->kflen[k];
    kf_reset(mu, &v->kflen[k]);
    v->active[k] = 0;
    return 1;
  }
  return 0;
}

/* update(z), utils.py:242-275 */
static void kf_update(env_view *v, int k, int has_z, double zx, double zy) {
  const d2d_cfg *c = v->c;
  double *mu = v->kf + (size_t)k * D2D_KF, *S = mu + 4;
  if (v->active[k]) {
    kf_predict(v, k);
    if (has_z) {
      /* S2 = Sigma_z + H Sigma H^T ; K = Sigma H^T inv(S2) ; mu += K (z - H mu) ; Sigma = (I - K H) Sigma */
      double a = c->sigma + S[0], b = S[1], cc = S[4], d = c->sigma + S[5];
      double det = a * d - b * cc;
      double idet = 1.0 / det;
      double i00 = d * idet, i01 = -b * idet, i10 = -cc * idet, i11 = a * idet;
      double K[8];
      for (int i = 0; i < 4; ++i) {
        K[2 * i] = S[4 * i] * i00 + S[4 * i + 1] * i10;
        K[2 * i + 1] = S[4 * i] * i01 + S[4 * i + 1] * i11;
      }
      double rx = zx - mu[0], ry = zy - mu[1];
      double IKH[16], P[16];
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) IKH[4 * i + j] = (i == j ? 1.0 : 0.0) - (j < 2 ? K[2 * i + j] : 0.0);
      mat4_mul(IKH, S, P);
      for (int i = 0; i < 4; ++i) mu[i] = mu[i] + (K[2 * i] * rx + K[2 * i + 1] * ry);
      memcpy(S, P, sizeof P);
    }
  } else if (has_z) {
    kf_reset(mu, &v->kflen[k]);
    mu[0] = zx;
    mu[1] = zy;
    v->active[k] = 1;
  }
}

/* utils.py:605 (measurement) + utils.py:749-753 */
static void st_tracker(env_view *v, const d2d_state *s) {
  const d2d_cfg *c = v->c;
  int N = c->N;
  const double *px = v->ag + D2D_A_PX * N, *py = v->ag + D2D_A_PY * N;
  const double *noise = s->noise ? s->noise + (size_t)v->e * N * 2 : 0;
  for (int k = 0; k < N; ++k) {
    if (c->kf_enabled) {
      double zx = px[k], zy = py[k];
      if (noise) {
        zx = px[k] + c->sigma * noise[2 * k];
        zy = py[k] + c->sigma * noise[2 * k + 1];
      }
      kf_update(v, k, v->hit[k], zx, zy);
    } else if (v->hit[k]) {
      v->active[k] = 1; /* without the filter only activation is modelled */
    }
  }
}

static uint8_t get_grid(const env_view *v, double x, double y) { /* utils.py:545-548 */
  const d2d_cfg *c = v->c;
  if (x >= c->W_px || x < 0 || y >= c->H_px || y < 0) return 1;
  return v->gt[(size_t)cell_of(x, c->scale) * c->H + cell_of(y, c->scale)];
}

/* envs/drone_v2.py:197-214 with utils.py:733-743, 755-762 */
static void st_control(env_view *v, const d2d_state *s) {
  const d2d_cfg *c = v->c;
  int ok = 1, has_wp = 0;
  if (c->planner_mode == D2D_PLANNER_NOMOVE) {
    v->tgt[0] = -1; /* traj_planner.py:72 */
    v->tgt[1] = -1;
  } else {
    ok = s->plan_ok[v->e] != 0;
    has_wp = s->wp_valid[v->e] != 0;
  }
  double *d = v->dr;
  if (!ok) {
    double n = norm2(d[D2D_D_VX], d[D2D_D_VY]);
    if (n <= c->max_acc * c->dt) {
      d[D2D_D_VX] = 0;
      d[D2D_D_VY] = 0;
    } else {
      d[D2D_D_VX] = d[D2D_D_VX] - d[D2D_D_VX] / n * c->max_acc * c->dt;
      d[D2D_D_VY] = d[D2D_D_VY] - d[D2D_D_VY] / n * c->max_acc * c->dt;
      d[D2D_D_X] += d[D2D_D_VX] * c->dt;
      d[D2D_D_Y] += d[D2D_D_VY] * c->dt;
    }
    v->cnt[D2D_C_SM] = D2D_SM_PLANNING;
    v->cnt[D2D_C_FAIL] += 1;
  } else {
    v->cnt[D2D_C_SM] = D2D_SM_EXECUTING;
    v->cnt[D2D_C_FAIL] = 0;
  }
  if (has_wp) {
    const double *wp = s->wp + (size_t)v->e * 6;
    d[D2D_D_AX] = wp[4];
    d[D2D_D_AY] = wp[5];
    d[D2D_D_VX] = wp[2];
    d[D2D_D_VY] = wp[3];
    d[D2D_D_X] = rint(wp[0]); /* round(): half to even */
    d[D2D_D_Y] = rint(wp[1]);
  }
  double a = s->action[v->e];
  d[D2D_D_YAW] = py_mod(d[D2D_D_YAW] + a * c->yaw_rate * c->dt, 360.0);
}

/* utils.py:764-778 + envs/drone_v2.py:217-235 */
static void st_collide(env_view *v) {
  const d2d_cfg *c = v->c;
  int N = c->N;
  const double *px = v->ag + D2D_A_PX * N, *py = v->ag + D2D_A_PY * N, *rr = v->ag + D2D_A_R * N;
  double x = v->dr[D2D_D_X], y = v->dr[D2D_D_Y], R = c->drone_radius;
  const double off[5][2] = {{-R, 0}, {0, 0}, {R, 0}, {0, -R}, {0, R}};
  int col = 0;
  for (int q = 0; q < 5 && !col; ++q)
    if (get_grid(v, x + off[q][0], y + off[q][1]) == 1) col = 1;
  if (!col)
    for (int k = 0; k < N; ++k) {
      double dx = px[k] - x, dy = py[k] - y;
      if (norm2(dx, dy) < rr[k] + R) {
        col = 2;
        break;
      }
    }
  int dead = 0, frz = 0;
  if (col == 0) {
    double gx = x - v->tgt[0], gy = y - v->tgt[1];
    if (norm2(gx, gy) <= 10) v->cnt[D2D_C_SM] = D2D_SM_GOAL_REACHED;
    double vn = norm2(v->dr[D2D_D_VX], v->dr[D2D_D_VY]);
    dead = (v->cnt[D2D_C_FAIL] >= 10 && vn == 0) ? 1 : 0;
    frz = ((double)v->cnt[D2D_C_STEPS] >= c->max_steps && !dead) ? 1 : 0;
  }
  int done = (col != 0) || dead || frz ||
             (v->cnt[D2D_C_SM] == D2D_SM_GOAL_REACHED && v->cnt[D2D_C_TGT_NEXT] >= v->cnt[D2D_C_NTGT]);
  if (done && c->kf_enabled) /* drone_v2.py:232-235 */
    for (int k = 0; k < N; ++k)
      if (v->active[k]) {
        v->cnt[D2D_C_BUF_N] += 1;
        v->cnt[D2D_C_BUF_TS] += v->kflen[k];
      }
  v->flags[D2D_F_COLLISION] = (uint8_t)col;
  v->flags[D2D_F_DEADLOCK] = (uint8_t)dead;
  v->flags[D2D_F_FREEZING] = (uint8_t)frz;
  v->flags[D2D_F_DONE] = (uint8_t)done;
}

/* utils.py:780-784 + envs/drone_v2.py:251-255 */
static void st_obs(env_view *v) {
  const d2d_cfg *c = v->c;
  int L = c->L, edge = (L - 1) / 2;
  int ix = cell_of(v->dr[D2D_D_X], c->scale), iy = cell_of(v->dr[D2D_D_Y], c->scale);
  for (int p = 0; p < L; ++p)
    for (int q = 0; q < L; ++q) {
      int i = ix - edge + p, j = iy - edge + q;
      v->obs[p * L + q] = (i >= 0 && i < c->W && j >= 0 && j < c->H) ? v->dm[(size_t)i * c->H + j] : 0;
    }
  *v->obs_yaw = (float)v->dr[D2D_D_YAW];
}

static int check(const d2d_cfg *c, const d2d_state *s) {
  if (!c || !s) return fail(-1, "null cfg/state");
  if (c->abi_version != D2D_ABI_VERSION) return fail(-2, "ABI version mismatch");
  if (c->B < 0 || c->N < 0 || c->W <= 0 || c->H <= 0 || c->R <= 0 || c->L <= 0 || (c->L & 1) == 0)
    return fail(-1, "bad dimensions");
  if (c->noise_row0 < 0 || c->noise_row0 >= (c->noise_rows > 1 ? c->noise_rows : 1)) return fail(-1, "noise_row0 outside [0, noise_rows)");
  if (c->grid_tile != 0) return fail(-4, "the oracle keeps the reference's row-major grids (grid_tile must be 0)");
  if (!(c->scale >= 2)) return fail(-4, "map_scale < 2: the reference's ray march never advances (utils.py:621)");
  if (c->kf_enabled && (!s->kf || !s->kf_len)) return fail(-1, "kf_enabled without kf buffers");
  if (c->planner_mode == D2D_PLANNER_EXTERNAL && (!s->plan_ok || !s->wp_valid || !s->wp))
    return fail(-1, "external planner mode needs plan_ok / wp_valid / wp");
  return 0;
}

int d2d_oracle_run_stages(const d2d_cfg *c, const d2d_state *s, uint32_t stages, void *stream) {
  (void)stream;
  int rc = check(c, s);
  if (rc) return rc;
  /* envs are independent: the cpu_baseline leg of bench.py may spread them over host threads */
#pragma omp parallel for schedule(static) num_threads(g_threads > 0 ? g_threads : 1)
  for (int e = 0; e < c->B; ++e) {
    env_view v = view(c, s, e);
    if ((stages & D2D_ST_SKIP_DONE) && v.flags[D2D_F_DONE]) continue;
    if (stages & D2D_ST_FSM) st_fsm(&v);
    if (stages & D2D_ST_AGENTS) st_agents(&v);
    if (stages & D2D_ST_RAYCAST) st_raycast(&v);
    if (stages & D2D_ST_DYNGRID) st_dyngrid(&v);
    if (stages & D2D_ST_TRACKER) st_tracker(&v, s);
    if (stages & D2D_ST_CONTROL) st_control(&v, s);
    if (stages & D2D_ST_COLLIDE) st_collide(&v);
    if (stages & D2D_ST_OBS) st_obs(&v);
  }
  return 0;
}

int d2d_oracle_step(const d2d_cfg *c, const d2d_state *s, void *stream) {
  return d2d_oracle_run_stages(c, s, D2D_ST_ALL, stream);
}
int d2d_oracle_perceive(const d2d_cfg *c, const d2d_state *s, void *stream) {
  return d2d_oracle_run_stages(c, s, D2D_ST_PERCEIVE, stream);
}
int d2d_oracle_act(const d2d_cfg *c, const d2d_state *s, void *stream) {
  return d2d_oracle_run_stages(c, s, D2D_ST_ACT, stream);
}

int d2d_oracle_rollout(const d2d_cfg *c, const d2d_state *s, int32_t nsteps, const double *actions,
                       const double *wp_steps, const double *pin, uint8_t *coll_out, void *stream) {
  (void)stream;
  int rc = check(c, s);
  if (rc) return rc;
  if (!actions || nsteps < 0) return fail(-1, "rollout: bad arguments");
  d2d_state t = *s;
  for (int k = 0; k < nsteps; ++k) {
    t.action = actions + (size_t)k * c->B;
    if (wp_steps) t.wp = wp_steps + (size_t)k * c->B * 6;
    if (s->noise && c->noise_rows > 1) t.noise = s->noise + (size_t)((c->noise_row0 + k) % c->noise_rows) * c->B * c->N * 2;
    if (pin)
      for (int e = 0; e < c->B; ++e) {
        s->drone[(size_t)e * D2D_DF + D2D_D_X] = pin[2 * e];
        s->drone[(size_t)e * D2D_DF + D2D_D_Y] = pin[2 * e + 1];
      }
    rc = d2d_oracle_run_stages(c, &t, D2D_ST_ALL, 0);
    if (rc) return rc;
    if (coll_out)
      for (int e = 0; e < c->B; ++e) coll_out[(size_t)k * c->B + e] = s->flags[(size_t)e * 4 + D2D_F_COLLISION];
  }
  return 0;
}

int d2d_oracle_reset(const d2d_cfg *c, const d2d_state *s, const d2d_state *init, const uint8_t *mask,
                     void *stream) {
  (void)stream;
  int rc = check(c, s);
  if (rc) return rc;
  if (!init) return fail(-1, "reset: null snapshot");
  size_t N = (size_t)c->N, WH = (size_t)c->W * c->H, LL = (size_t)c->L * c->L;
  for (int e = 0; e < c->B; ++e) {
    if (mask && !mask[e]) continue;
    memcpy(s->agents + e * D2D_AF * N, init->agents + e * D2D_AF * N, sizeof(double) * D2D_AF * N);
    memcpy(s->agent_unit + e * N, init->agent_unit + e * N, sizeof(int32_t) * N);
    memcpy(s->dyn_prev + e * N * 3, init->dyn_prev + e * N * 3, sizeof(int32_t) * N * 3);
    memcpy(s->gt + e * WH, init->gt + e * WH, WH);
    memcpy(s->dmap + e * WH, init->dmap + e * WH, WH);
    memcpy(s->drone + (size_t)e * D2D_DF, init->drone + (size_t)e * D2D_DF, sizeof(double) * D2D_DF);
    memcpy(s->target + (size_t)e * 2, init->target + (size_t)e * 2, sizeof(double) * 2);
    memcpy(s->targets + (size_t)e * c->T * 2, init->targets + (size_t)e * c->T * 2, sizeof(double) * c->T * 2);
    memcpy(s->counters + (size_t)e * D2D_CF, init->counters + (size_t)e * D2D_CF, sizeof(int32_t) * D2D_CF);
    memcpy(s->active + e * N, init->active + e * N, N);
    if (s->kf && init->kf) memcpy(s->kf + e * N * D2D_KF, init->kf + e * N * D2D_KF, sizeof(double) * N * D2D_KF);
    if (s->kf_len && init->kf_len) memcpy(s->kf_len + e * N, init->kf_len + e * N, sizeof(int32_t) * N);
    memset(s->hit + e * N, 0, N);
    s->newly[e] = 0;
    memset(s->flags + (size_t)e * 4, 0, 4);
    memset(s->obs_local + e * LL, 0, LL);
    s->obs_yaw[e] = 0;
  }
  return 0;
}

int d2d_oracle_tan_array(const double *in, double *out, int64_t n, void *stream) {
  (void)stream;
  for (int64_t i = 0; i < n; ++i) out[i] = tan(in[i]);
  return 0;
}

/* =================================================================================================
 * Planner / gaze plugins: Primitive (traj_planner.py:78-233) and Oxford
 * (yaw_planner.py:41-127), restated scalar and literal.  Same status as the rest of this file: TEST
 * INFRASTRUCTURE, pinned by the reference's own episodes (the Primitive traces under tests/golden record the action,
 * plan result, head waypoint and trajectory length of every step of the imported reference).
 *
 * numpy roundings that matter here, each measured against numpy 2.2.6 + its bundled OpenBLAS in the build
 * container (100000 / 100000 bit-identical, tools in DESIGN.md section 4):
 *   np.array([1, t, t**2]) @ coeff.T            = fma(t**2, a/2, p + t * v)            (traj_planner.py:121,176)
 *   np.array([1, 2*t]) @ coeff[:, 1:].T         = v + (2*t) * (a/2)                    (traj_planner.py:122)
 *   np.array([1, H, H**2]) @ [[p],[v],[a/2]]    = (p + 2 * v) + 4 * (a/2)              (traj_planner.py:182)
 *   np.array([1, 2*H]) @ [[v],[a/2]]            = v + 4 * (a/2)                        (traj_planner.py:172,183)
 *   norm(2-vector)                              = sqrt(fma(y, y, x * x))               (norm2 above)
 *   np.sum(W x H float64)                       = pairwise summation (d2d_plan.pw_leaf / pw_prog)
 *   np.arccos(q) <= half_fov                    = d2d_plan.acos_key_lo / acos_mask (numpy's arccos is a SIMD
 *                                                 routine, not libm's; the host tabulates its decisions)
 * ================================================================================================= */

static int g_skip_done = 0; /* set by closed_loop(D2D_DONE_FREEZE) around the plugin stages: finished envs are left alone */

static uint8_t dm_get_grid(const env_view *v, double x, double y) { /* drone.map.get_grid, utils.py:545-548 */
  const d2d_cfg *c = v->c;
  if (x >= c->W_px || x < 0 || y >= c->H_px || y < 0) return 1;
  return v->dm[(size_t)cell_of(x, c->scale) * c->H + cell_of(y, c->scale)];
}

typedef struct plan_view {
  const d2d_plan *p;
  double *traj;      /* [traj_cap][4] */
  int32_t *hdr;      /* head, stored */
  double *trk_radius;
  uint8_t *trk_prev;
  int32_t *seen;
  double *nodes;
  int32_t *hash;
  int32_t *stat;
} plan_view;

static plan_view pview(const d2d_cfg *c, const d2d_plan *p, int e) {
  plan_view q;
  size_t N = (size_t)(c->N > 0 ? c->N : 1);
  q.p = p;
  q.traj = p->traj + (size_t)e * p->traj_cap * 4;
  q.hdr = p->traj_hdr + (size_t)e * 2;
  q.trk_radius = p->trk_radius + (size_t)e * N;
  q.trk_prev = p->trk_prev + (size_t)e * N;
  q.seen = p->seen_step + (size_t)e * c->W * c->H;
  q.nodes = p->nodes + (size_t)e * p->node_cap * D2D_NODE_F;
  q.hash = p->hash + (size_t)e * p->hash_cap;
  q.stat = p->plan_stat + (size_t)e * 4;
  return q;
}

/* Planner.is_free, traj_planner.py:28-59 */
static int is_free(const env_view *v, const plan_view *q, double x, double y, double t) {
  const d2d_cfg *c = v->c;
  if (isnan(x) || isnan(y)) return 0;
  const double d = q->p->safe_dist;
  if (dm_get_grid(v, x - d, y) == 1) return 0;
  if (dm_get_grid(v, x, y) == 1) return 0;
  if (dm_get_grid(v, x + d, y) == 1) return 0;
  if (dm_get_grid(v, x, y - d) == 1) return 0;
  if (dm_get_grid(v, x, y + d) == 1) return 0;
  for (int k = 0; k < c->N; ++k)
    if (v->active[k]) {
      const double *mu = v->kf + (size_t)k * D2D_KF;
      double ex = mu[0] + t * mu[2], ey = mu[1] + t * mu[3]; /* estimate_pos, utils.py:220-223 */
      note_margin(norm2(x - ex, y - ey), c->drone_radius + q->trk_radius[k] + 5 + c->sigma);
      if (norm2(x - ex, y - ey) <= c->drone_radius + q->trk_radius[k] + 5 + c->sigma) return 0;
    }
  return 1;
}

static int64_t py_ifloordiv(int64_t a, int64_t b) { /* Python int // for b > 0 */
  int64_t d = a / b;
  if ((a % b != 0) && (a < 0)) --d;
  return d;
}

/* Primitive_Node.get_index, traj_planner.py:93: (round(x) // 10, round(y) // 10, round(vx), round(vy)) */
static int64_t node_key(double px, double py, double vx, double vy) {
  int64_t a = py_ifloordiv((int64_t)rint(px), 10), b = py_ifloordiv((int64_t)rint(py), 10);
  int64_t cc = (int64_t)rint(vx), d = (int64_t)rint(vy);
  return (int64_t)((((uint64_t)(a + 32768) & 0xffff) << 48) | (((uint64_t)(b + 32768) & 0xffff) << 32) |
                   (((uint64_t)(cc + 32768) & 0xffff) << 16) | ((uint64_t)(d + 32768) & 0xffff));
}

static void node_set_i64(double *slot, int64_t x) { memcpy(slot, &x, 8); }
static int64_t node_get_i64(const double *slot) {
  int64_t x;
  memcpy(&x, slot, 8);
  return x;
}

static uint32_t key_hash(int64_t k) {
  uint64_t x = (uint64_t)k * 0x9E3779B97F4A7C15ull;
  return (uint32_t)(x >> 32);
}

/* slot of `key` in the env's table, or -1 */
static int hash_find(const plan_view *q, int64_t key) {
  uint32_t m = (uint32_t)q->p->hash_cap - 1, h = key_hash(key) & m;
  for (;;) {
    int32_t s = q->hash[h];
    if (s == 0) return -1;
    if (node_get_i64(q->nodes + (size_t)(s - 1) * D2D_NODE_F + D2D_N_KEY) == key) return s - 1;
    h = (h + 1) & m;
  }
}

static void hash_put(const plan_view *q, int64_t key, int slot) {
  uint32_t m = (uint32_t)q->p->hash_cap - 1, h = key_hash(key) & m;
  while (q->hash[h] != 0) h = (h + 1) & m;
  q->hash[h] = slot + 1;
}

static void node_write(const env_view *v, double *nd, double px, double py, double vx, double vy, double cost, double ax,
                       double ay, int parent, int itr) {
  nd[D2D_N_PX] = px;
  nd[D2D_N_PY] = py;
  nd[D2D_N_VX] = vx;
  nd[D2D_N_VY] = vy;
  nd[D2D_N_COST] = cost;
  /* traj_planner.py:88 */
  nd[D2D_N_TOTAL] = cost + 0.5 * norm2(px - v->tgt[0], py - v->tgt[1]) + 0.1 * norm2(vx, vy);
  nd[D2D_N_AX] = ax;
  nd[D2D_N_AY] = ay;
  int32_t link[2] = {parent, itr};
  memcpy(nd + D2D_N_LINK, link, 8);
  node_set_i64(nd + D2D_N_KEY, node_key(px, py, vx, vy));
  node_set_i64(nd + D2D_N_STATE, 1);
}

/* Primitive.plan, traj_planner.py:125-218.  The open set is a Python dict: iteration order = order of first
 * insertion of each key, a replaced value keeps its place; min() returns the first minimal entry.  Slots are
 * appended in insertion order and closed in place, which is exactly that order. */
static int primitive_search(const env_view *v, const plan_view *q) {
  const d2d_plan *p = q->p;
  const double H = p->horizon;
  memset(q->hash, 0, sizeof(int32_t) * (size_t)p->hash_cap);
  int nn = 0, open_n = 0, goal = -1, itr = 0, expansions = 0, overflow = 0;
  node_write(v, q->nodes, v->dr[D2D_D_X], v->dr[D2D_D_Y], v->dr[D2D_D_VX], v->dr[D2D_D_VY], 0.0, 0.0, 0.0, -1, 0);
  hash_put(q, node_get_i64(q->nodes + D2D_N_KEY), 0);
  nn = 1;
  open_n = 1;
  for (;;) {
    itr += 1;
    if (open_n == 0 || itr >= p->max_itr) break;
    int cur = -1;
    double best = 0;
    for (int s = 0; s < nn; ++s) {
      const double *nd = q->nodes + (size_t)s * D2D_NODE_F;
      if (node_get_i64(nd + D2D_N_STATE) != 1) continue;
      if (cur < 0 || nd[D2D_N_TOTAL] < best) {
        cur = s;
        best = nd[D2D_N_TOTAL];
      }
    }
    double *cn = q->nodes + (size_t)cur * D2D_NODE_F;
    const double px = cn[D2D_N_PX], py = cn[D2D_N_PY], vx = cn[D2D_N_VX], vy = cn[D2D_N_VY], ccost = cn[D2D_N_COST];
    int32_t link[2];
    memcpy(link, cn + D2D_N_LINK, 8);
    const int citr = link[1];
    if (norm2(px - v->tgt[0], py - v->tgt[1]) <= p->goal_tol) {
      goal = cur;
      break;
    }
    node_set_i64(cn + D2D_N_STATE, 2);
    open_n -= 1;
    expansions += 1;
    for (int ia = 0; ia < p->nu && !overflow; ++ia)
      for (int ja = 0; ja < p->nu; ++ja) {
        const double ax = p->u_space[ia], ay = p->u_space[ja];
        const double hx = ax / 2, hy = ay / 2;
        const double vex = vx + (2 * H) * hx, vey = vy + (2 * H) * hy; /* :172,183 */
        if (!(norm2(vex, vey) < p->vmax)) continue;
        int ok = 1;
        for (int sidx = 0; sidx < p->n_sample; ++sidx) { /* :175-180 */
          const double t = p->sample_t[2 * sidx], t2 = p->sample_t[2 * sidx + 1];
          const double sx = rint(fma(t2, hx, px + t * vx)), sy = rint(fma(t2, hy, py + t * vy));
          if (!is_free(v, q, sx, sy, t + citr * H)) {
            ok = 0;
            break;
          }
        }
        if (!ok) continue;
        const double ex = rint((px + H * vx) + (H * H) * hx), ey = rint((py + H * vy) + (H * H) * hy); /* :182 */
        const double cost = ccost + (ax * ax + ay * ay) / 100 + 10;                                   /* :184 */
        double tmp[D2D_NODE_F];
        node_write(v, tmp, ex, ey, vex, vey, cost, ax, ay, cur, citr + 1);
        /* :192-202, applied successor by successor in generation order */
        const int64_t key = node_get_i64(tmp + D2D_N_KEY);
        const int s = hash_find(q, key);
        if (s >= 0) {
          double *nd = q->nodes + (size_t)s * D2D_NODE_F;
          if (node_get_i64(nd + D2D_N_STATE) == 2) continue;
          if (nd[D2D_N_COST] > cost) memcpy(nd, tmp, sizeof tmp);
        } else {
          if (nn >= p->node_cap) {
            overflow = 1;
            break;
          }
          memcpy(q->nodes + (size_t)nn * D2D_NODE_F, tmp, sizeof tmp);
          hash_put(q, key, nn);
          nn += 1;
          open_n += 1;
        }
      }
    if (overflow) break;
  }
  q->stat[0] += 1;
  q->stat[1] = expansions;
  q->stat[2] = nn;
  if (overflow) q->stat[3] = 1;
  q->hdr[0] = 0;
  q->hdr[1] = 0;
  if (goal < 0 || overflow) return 0;
  /* :207-216: waypoints of every primitive from the goal back to the start, then reversed */
  int depth = 0;
  for (int s = goal; s != 0;) {
    int32_t link[2];
    memcpy(link, q->nodes + (size_t)s * D2D_NODE_F + D2D_N_LINK, 8);
    s = link[0];
    depth += 1;
  }
  if (depth * p->n_ts > p->traj_cap) {
    q->stat[3] = 1;
    return 0;
  }
  int s = goal;
  for (int lvl = depth - 1; lvl >= 0; --lvl) {
    const double *nd = q->nodes + (size_t)s * D2D_NODE_F;
    int32_t link[2];
    memcpy(link, nd + D2D_N_LINK, 8);
    const double *pn = q->nodes + (size_t)link[0] * D2D_NODE_F;
    const double hx = nd[D2D_N_AX] / 2, hy = nd[D2D_N_AY] / 2;
    for (int m = 0; m < p->n_ts; ++m) {
      const double t = p->traj_t[3 * m], t2 = p->traj_t[3 * m + 1], tt = p->traj_t[3 * m + 2];
      double *w = q->traj + ((size_t)lvl * p->n_ts + m) * 4;
      w[0] = rint(fma(t2, hx, pn[D2D_N_PX] + t * pn[D2D_N_VX])); /* :121 */
      w[1] = rint(fma(t2, hy, pn[D2D_N_PY] + t * pn[D2D_N_VY]));
      w[2] = pn[D2D_N_VX] + tt * hx;                             /* :122 */
      w[3] = pn[D2D_N_VY] + tt * hy;
    }
    s = link[0];
  }
  q->hdr[1] = depth * p->n_ts;
  return 1;
}

/* Primitive.replan_check, traj_planner.py:220-233; returns 1 when the trajectory was cleared */
static int replan_check(const env_view *v, const plan_view *q) {
  const d2d_cfg *c = v->c;
  const int head = q->hdr[0], n = q->hdr[1] - q->hdr[0];
  int clear = 0, swept_wall = 0;
  for (int i = 0; i < n && !clear; ++i) {
    const double *w = q->traj + (size_t)(head + i) * 4;
    const double ti = i * c->dt;
    const int ci = cell_of(w[0], c->scale), cj = cell_of(w[1], c->scale);
    /* swep_map is uint8: the stored value is trunc(i * dt); np.sum(where(occ == 1) * swep) > 0 asks for one
     * trajectory cell with a non-zero stored value on a wall (later visits only raise the value) */
    if (ci >= 0 && ci < c->W && cj >= 0 && cj < c->H && (uint8_t)ti > 0 && v->dm[(size_t)ci * c->H + cj] == D2D_OCCUPIED)
      swept_wall = 1;
    for (int k = 0; k < c->N; ++k)
      if (v->active[k]) {
        const double *mu = v->kf + (size_t)k * D2D_KF;
        double ex = mu[0] + ti * mu[2], ey = mu[1] + ti * mu[3];
        note_margin(norm2(ex - w[0], ey - w[1]), c->drone_radius + q->trk_radius[k]);
        if (norm2(ex - w[0], ey - w[1]) <= c->drone_radius + q->trk_radius[k]) {
          clear = 1;
          break;
        }
      }
  }
  if (clear || swept_wall) {
    q->hdr[0] = 0;
    q->hdr[1] = 0;
    return 1;
  }
  return 0;
}

static int plan_check(const d2d_cfg *c, const d2d_state *s, const d2d_plan *p) {
  int rc = check(c, s);
  if (rc) return rc;
  if (!p) return fail(-1, "null plan");
  if (p->planner == D2D_PLAN_PRIMITIVE || p->gaze == D2D_GAZE_OXFORD) {
    if (!p->traj || !p->traj_hdr) return fail(-1, "plan: null trajectory buffers");
    if (!c->kf_enabled) return fail(-4, "device plugins need the Kalman trackers on the device (kf_enabled)");
  }
  if (p->planner == D2D_PLAN_PRIMITIVE) {
    if (!p->u_space || !p->sample_t || !p->traj_t || !p->trk_radius || !p->trk_prev || !p->nodes || !p->hash || !p->plan_stat)
      return fail(-1, "plan: null planner pointer");
    if (p->hash_cap <= p->node_cap || (p->hash_cap & (p->hash_cap - 1))) return fail(-1, "plan: hash_cap must be a power of two > node_cap");
    if (!s->plan_ok || !s->wp_valid || !s->wp) return fail(-1, "plan: plan_ok / wp_valid / wp buffers missing");
  }
  if (p->gaze == D2D_GAZE_OXFORD) {
    if (!p->yaw_space || !p->tobs_tab || !p->pw_leaf || !p->pw_prog || !p->seen_step) return fail(-1, "plan: null gaze pointer");
    if (!s->action) return fail(-1, "plan: null action buffer");
  }
  return 0;
}

int d2d_oracle_plan_stage(const d2d_cfg *c, const d2d_state *s, const d2d_plan *p, void *stream) {
  (void)stream;
  int rc = plan_check(c, s, p);
  if (rc) return rc;
  if (p->planner != D2D_PLAN_PRIMITIVE) return 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads > 0 ? g_threads : 1)
  for (int e = 0; e < c->B; ++e) {
    env_view v = view(c, s, e);
    plan_view q = pview(c, p, e);
    if (g_skip_done && v.flags[D2D_F_DONE]) continue;
    /* KalmanFilter.__init__ on archive puts tracker.radius back to params.agent_radius (utils.py:184,238) */
    for (int k = 0; k < c->N; ++k) {
      if (q.trk_prev[k] && !v.active[k]) q.trk_radius[k] = p->agent_radius;
      q.trk_prev[k] = v.active[k];
    }
    replan_check(&v, &q);                                    /* drone_v2.py:194 */
    int ok = 1;
    if (q.hdr[1] - q.hdr[0] == 0) ok = primitive_search(&v, &q); /* drone_v2.py:197, traj_planner.py:128-129 */
    uint8_t *plan_ok = (uint8_t *)s->plan_ok, *wp_valid = (uint8_t *)s->wp_valid;
    double *wp = (double *)s->wp + (size_t)e * 6;
    plan_ok[e] = (uint8_t)ok;
    if (q.hdr[1] - q.hdr[0] > 0) { /* step_pos, utils.py:733-739 */
      const double *w = q.traj + (size_t)q.hdr[0] * 4;
      wp[0] = w[0]; wp[1] = w[1]; wp[2] = w[2]; wp[3] = w[3]; wp[4] = 0; wp[5] = 0;
      wp_valid[e] = 1;
      q.hdr[0] += 1;
    } else {
      wp_valid[e] = 0;
      for (int i = 0; i < 6; ++i) wp[i] = 0;
    }
  }
  return 0;
}

/* ---- Oxford, yaw_planner.py:41-127 ---- */
static int64_t dkey(double x) {
  int64_t b;
  memcpy(&b, &x, 8);
  return b ^ ((b >> 63) & 0x7FFFFFFFFFFFFFFFll);
}

static int acos_le(const d2d_plan *p, double q) { /* np.arccos(q) <= view_angle, yaw_planner.py:77 */
  if (!(fabs(q) <= 1.0)) return 0;                 /* arccos is NaN there and the comparison false */
  int64_t k = dkey(q);
  if (k >= p->acos_key_lo + 64) return 1;
  if (k < p->acos_key_lo) return 0;
  return (int)((p->acos_mask >> (k - p->acos_key_lo)) & 1);
}

/* Oxford.get_view_map, yaw_planner.py:67-79, for cell (i, j) */
static int view_cell(const d2d_cfg *c, const d2d_plan *p, double x0, double y0, double cy, double sy, int i, int j) {
  const double x = (double)i * c->scale, y = (double)j * c->scale;
  const double a = x0 - x, b = y0 - y;
  const double d2 = a * a + b * b;
  if (d2 <= 0) return 1;
  const double q = ((x - x0) * cy + (y - y0) * sy) / sqrt(d2);
  return acos_le(p, q) && d2 <= c->depth * c->depth;
}

static double pw_leaf_sum(const double *a, int m) {
  if (m < 8) {
    double r = 0;
    for (int i = 0; i < m; ++i) r += a[i];
    return r;
  }
  double r[8];
  for (int j = 0; j < 8; ++j) r[j] = a[j];
  int i = 8;
  for (; i < m - (m % 8); i += 8)
    for (int j = 0; j < 8; ++j) r[j] += a[i + j];
  double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < m; ++i) res += a[i];
  return res;
}

static double pw_sum(const d2d_plan *p, const double *a) {
  double st[64];
  int sp = 0;
  for (int k = 0; k < p->pw_nprog; ++k) {
    int op = p->pw_prog[k];
    if (op >= 0) st[sp++] = pw_leaf_sum(a + p->pw_leaf[4 * op], p->pw_leaf[4 * op + 1]);
    else {
      sp -= 1;
      st[sp - 1] = st[sp - 1] + st[sp];
    }
  }
  return st[0];
}

int d2d_oracle_gaze_stage(const d2d_cfg *c, const d2d_state *s, const d2d_plan *p, void *stream) {
  (void)stream;
  int rc = plan_check(c, s, p);
  if (rc) return rc;
  if (p->gaze != D2D_GAZE_OXFORD) return 0;
  const double deg2rad = M_PI / 180.0; /* math.radians */
  const int WH = c->W * c->H;
#pragma omp parallel for schedule(static) num_threads(g_threads > 0 ? g_threads : 1)
  for (int e = 0; e < c->B; ++e) {
    env_view v = view(c, s, e);
    plan_view q = pview(c, p, e);
    if (g_skip_done && v.flags[D2D_F_DONE]) continue;
    double *sw = (double *)malloc(sizeof(double) * WH * 2), *rew = sw + WH;
    const int call = v.cnt[D2D_C_STEPS] + 1; /* one plan() per step, before it (experiment.py:68-70) */
    if (call >= p->tobs_len) { /* only reachable when stepping goes on past the longest episode (D2D_DONE_CONTINUE): the
                                  time-since-observed table ends there; the policy then holds the yaw, as on the device */
      ((double *)s->action)[e] = 0;
      free(sw);
      continue;
    }
    const int head = q.hdr[0], n = q.hdr[1] - q.hdr[0];
    for (int i = 0; i < WH; ++i) sw[i] = 0;
    for (int i = 0; i < n; ++i) { /* :88-90 */
      const double *w = q.traj + (size_t)(head + i) * 4;
      const int ci = cell_of(w[0], c->scale), cj = cell_of(w[1], c->scale);
      if (ci >= 0 && ci < c->W && cj >= 0 && cj < c->H) sw[ci * c->H + cj] = i * c->dt;
    }
    const double x0 = v.dr[D2D_D_X], y0 = v.dr[D2D_D_Y], yaw = v.dr[D2D_D_YAW];
    {
      const double cy = cos(yaw * deg2rad), sy = -sin(yaw * deg2rad); /* :71 */
      for (int i = 0; i < c->W; ++i)
        for (int j = 0; j < c->H; ++j)
          if (view_cell(c, p, x0, y0, cy, sy, i, j)) q.seen[i * c->H + j] = call; /* :93-97 */
    }
    for (int i = 0; i < WH; ++i) { /* :109-111 */
      const double tobs = q.seen[i] > 0 ? p->tobs_tab[call - q.seen[i]] : p->tobs_tab[p->tobs_len + call];
      const double w = sw[i];
      if (w > 0 && w <= 3 && tobs >= 0.5) rew[i] = 1000000;
      else if (w > 3 && tobs >= 0.5) rew[i] = 1000;
      else rew[i] = (1 * tobs < 1) ? 1 * tobs : 1; /* np.clip(c3 * t, -inf, 1) */
    }
    double *act = (double *)s->action;
    if (n == 0) { /* :118-119 */
      act[e] = 0;
      free(sw);
      continue;
    }
    const double hx = q.traj[(size_t)head * 4], hy = q.traj[(size_t)head * 4 + 1];
    int best = 0;
    double max_reward = 0;
    double *prod = sw; /* the swept map is not needed any more */
    for (int a = 0; a < p->n_yaw; ++a) { /* :121-125 */
      const double ty = py_mod(yaw + p->yaw_space[a] * c->dt, 360.0);
      const double cy = cos(ty * deg2rad), sy = -sin(ty * deg2rad);
      for (int i = 0; i < c->W; ++i)
        for (int j = 0; j < c->H; ++j) prod[i * c->H + j] = view_cell(c, p, hx, hy, cy, sy, i, j) ? rew[i * c->H + j] : 0.0;
      const double r = pw_sum(p, prod);
      if (max_reward < r) {
        best = a;
        max_reward = r;
      }
    }
    act[e] = p->yaw_space[best] / p->yaw_rate_max; /* :127 */
    free(sw);
  }
  return 0;
}

int d2d_oracle_plan_reset(const d2d_cfg *c, const d2d_plan *p, const uint8_t *mask, int32_t mask_stride, void *stream) {
  (void)stream;
  if (!c || !p) return fail(-1, "null cfg/plan");
  size_t N = (size_t)(c->N > 0 ? c->N : 1);
  for (int e = 0; e < c->B; ++e) {
    if (mask && !mask[(size_t)e * mask_stride]) continue;
    if (p->traj_hdr) p->traj_hdr[2 * e] = p->traj_hdr[2 * e + 1] = 0;
    if (p->trk_radius && p->trk_radius0) memcpy(p->trk_radius + e * N, p->trk_radius0 + e * N, sizeof(double) * N);
    if (p->trk_prev) memset(p->trk_prev + e * N, 0, N);
    if (p->seen_step) memset(p->seen_step + (size_t)e * c->W * c->H, 0, sizeof(int32_t) * (size_t)c->W * c->H);
  }
  return 0;
}

int d2d_oracle_closed_loop(const d2d_cfg *c, const d2d_state *s, const d2d_plan *p, int32_t nsteps, int32_t on_done,
                           const d2d_state *init, void *stream) {
  (void)stream;
  int rc = plan_check(c, s, p);
  if (rc) return rc;
  const int auto_reset = on_done == D2D_DONE_RESET;
  const uint32_t skip = on_done == D2D_DONE_FREEZE ? D2D_ST_SKIP_DONE : 0;
  if (auto_reset && !init) return fail(-1, "closed_loop: D2D_DONE_RESET needs the snapshot");
  g_skip_done = skip != 0;
  for (int t = 0; t < nsteps && !rc; ++t) {
    if (auto_reset) { /* the next episode starts from the seeded world with fresh plugin objects (main.py:26-57) */
      uint8_t *done = (uint8_t *)malloc((size_t)c->B);
      for (int e = 0; e < c->B; ++e) done[e] = s->flags[(size_t)e * 4 + D2D_F_DONE];
      rc = d2d_oracle_reset(c, s, init, done, 0);
      if (!rc) rc = d2d_oracle_plan_reset(c, p, done, 1, 0);
      free(done);
      if (rc) break;
    }
    if ((rc = d2d_oracle_gaze_stage(c, s, p, 0))) break;
    d2d_state sn = *s; /* this step's row of the measurement noise */
    if (s->noise && c->noise_rows > 1) sn.noise = s->noise + (size_t)((c->noise_row0 + t) % c->noise_rows) * c->B * c->N * 2;
    if ((rc = d2d_oracle_run_stages(c, &sn, D2D_ST_PERCEIVE | skip, 0))) break;
    if ((rc = d2d_oracle_plan_stage(c, s, p, 0))) break;
    if ((rc = d2d_oracle_run_stages(c, s, D2D_ST_ACT | skip, 0))) break;
  }
  g_skip_done = 0;
  return rc;
}

/* the oracle launches nothing: one env at a time on the host */
int d2d_oracle_launch_shape(const d2d_cfg *c, const d2d_plan *p, int32_t out[4]) {
  (void)c; (void)p;
  out[0] = 1; out[1] = 0; out[2] = 0; out[3] = 0;
  return 0;
}

int d2d_oracle_sincos_array(const double *in, double *so, double *co, int64_t n, void *stream) {
  (void)stream;
  for (int64_t i = 0; i < n; ++i) {
    so[i] = sin(in[i]);
    co[i] = cos(in[i]);
  }
  return 0;
}
