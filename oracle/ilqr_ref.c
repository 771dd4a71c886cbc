/*
 * ORACLE (test infrastructure; never linked into or called by the product path).
 *
 * Plain-C float64 restatement of the reference contingency planner's numerical core:
 *   gen_dist_field                    planners/ilqr/utils.py:5-22, common/geometry.py:70-78
 *   init_(warm_start_)cost_tree       planners/mind/trajectory_tree.py:19-124 (per-node 256x256 fields,
 *                                     MATERIALISED exactly like the reference -- this is also the
 *                                     honest CPU baseline bench.py times), common/geometry.py:27-30
 *   PotentialField                    planners/ilqr/potential.py:62-264 (banker's rounding, border
 *                                     windows as written, 3x3 smoothing, Bernstein-2 patch)
 *   StatePotential/Constraint/Control planners/ilqr/potential.py:4-59
 *   TreeCost                          planners/ilqr/cost.py:326-446
 *   bicycle f, f_x, f_u               planners/mind/trajectory_tree.py:153-177 (closed-form Jacobian)
 *   iLQR.fit & friends                planners/ilqr/solver.py:80-421 (Q1 Jacobian at post-state, Q2
 *                                     root aliasing V[-1], Q9, Q10, Q19 reproduced)
 *   sin / cos / tan                   mind_amd/csrc/mind_trig.h (shared with the kernel: where numpy / glibc would round the last
 *                                     bit differently in 1-2 % of the arguments, kernel and oracle do not)
 * Pinned by tests/golden/ilqr.npz captured from the imported reference (tools/gen_golden.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/mind_hip.h"
/* sin / cos / tan: the routine k_ilqr uses (<= 0.75 / 0.9 ulp), so that kernel and oracle agree to the bit on every platform; the
 * oracle stays pinned to the reference (numpy's functions) by the goldens at 1e-9 */
#include "../mind_amd/csrc/mind_trig.h"
/* test hooks (tests/test_trig.py) */
void oracle_sincos(double x, double *s, double *c) { mind_sincos(x, s, c); }
void oracle_tan_cos(double x, double *t, double *c) { mind_tan_cos(x, t, c); }
#ifdef ORACLE_LIBM_TRIG
/* libilqr_oracle_libm.so (Makefile): the same oracle with the C library's sin / cos / tan, as numpy calls them in the reference.  Kernel ==
 * oracle "to the bit" rests on the header both include; this build is the independent witness that a defect of that header cannot hide on both
 * sides: tests/test_oracle_ilqr.py holds it to the reference goldens and to the shared-header build. */
static inline void libm_sincos(double x, double *s, double *c) { *s = sin(x); *c = cos(x); }
static inline void libm_tan_cos(double x, double *t, double *c) { *t = tan(x); *c = cos(x); }
#define mind_sincos libm_sincos
#define mind_tan_cos libm_tan_cos
#define mind_tan tan
#endif

#define NS 6
#define NU 2

typedef struct {
  const mind_ilqr_cfg *cfg;
  int M;
  const int32_t *parent;
  int *child_start, *child_list; /* children in creation (key) order */
  double *prob;                  /* (double)(float)prob */
  double **field;                /* per node [H*W] or shared */
  double *gx, *gy;               /* grid coordinates (xx[0,:], yy[:,0]) */
  double off[2];
  int W, H;
  double x0[NS];
  double target_vel;
  double mu, delta;
} solver_t;

/* numpy add.reduce for contiguous doubles (pairwise summation, blocks of 8 / 128) */
static double np_pairwise_sum(const double *a, long n) {
  if (n < 8) {
    double res = 0.;
    for (long i = 0; i < n; i++) res += a[i];
    return res;
  } else if (n <= 128) {
    double r[8];
    long i;
    for (i = 0; i < 8; i++) r[i] = a[i];
    for (i = 8; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; j++) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
  } else {
    long n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
  }
}

/* ---- grid + distance field (ilqr/utils.py:5-22) ---- */
static void make_grid(const mind_ilqr_cfg *cfg, const double *x0, double *gx, double *gy, double *off) {
  const int W = cfg->grid_w, H = cfg->grid_h;
  const double fsx = (double)(W - 1) * cfg->grid_res, fsy = (double)(H - 1) * cfg->grid_res;
  off[0] = x0[0] - 0.5 * fsx;
  off[1] = x0[1] - 0.5 * fsy;
  const double sx = fsx / (double)(W - 1), sy = fsy / (double)(H - 1);
  for (int i = 0; i < W; i++) gx[i] = (double)i * sx + 0.0;
  gx[W - 1] = fsx;
  for (int i = 0; i < H; i++) gy[i] = (double)i * sy + 0.0;
  gy[H - 1] = fsy;
  for (int i = 0; i < W; i++) gx[i] += off[0];
  for (int i = 0; i < H; i++) gy[i] += off[1];
}

static double seg_dist(double px, double py, const double *a, const double *b) {
  const double lvx = b[0] - a[0], lvy = b[1] - a[1];
  const double len2 = lvx * lvx + lvy * lvy;
  double t = ((px - a[0]) * lvx + (py - a[1]) * lvy) / len2;
  t = t < 0 ? 0 : (t > 1 ? 1 : t);
  const double qx = a[0] + t * lvx, qy = a[1] + t * lvy;
  const double dx = px - qx, dy = py - qy;
  return sqrt(dx * dx + dy * dy);
}

static void lane_dist_field(const double *gx, const double *gy, int W, int H, const double *lane, int P,
                            double *out /* [H*W] squared distance */) {
  for (int r = 0; r < H; r++)
    for (int c = 0; c < W; c++) {
      double d = INFINITY;
      for (int j = 0; j < P - 1; j++) {
        const double dj = seg_dist(gx[c], gy[r], lane + 2 * j, lane + 2 * j + 2);
        d = dj < d ? dj : d; /* np.minimum */
      }
      out[r * W + c] = d * d;
    }
}

/* ---- PotentialField (potential.py:62-264) ---- */
static long round_half_even(double v) { return (long)nearbyint(v); } /* default rounding mode = RNE == Python round() */

static void smooth_grid(const double *F, int W, int H, int xi, int yi, double s[3][3]) {
  double g[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#define FF(r, c) F[(size_t)(r) * W + (c)]
  if (xi == 0 && yi == 0) {
    for (int r = 0; r < 2; r++) for (int c = 0; c < 2; c++) g[1 + r][1 + c] = FF(r, c);
  } else if (xi == 0 && yi == H - 1) {
    for (int r = 0; r < 2; r++) for (int c = 0; c < 2; c++) g[1 + r][c] = FF(H - 2 + r, c);
  } else if (xi == W - 1 && yi == 0) {
    for (int r = 0; r < 2; r++) for (int c = 0; c < 2; c++) g[r][1 + c] = FF(r, W - 2 + c);
  } else if (xi == W - 1 && yi == H - 1) {
    for (int r = 0; r < 2; r++) for (int c = 0; c < 2; c++) g[r][c] = FF(H - 2 + r, W - 2 + c);
  } else if (xi == 0) {
    for (int r = 0; r < 3; r++) for (int c = 0; c < 2; c++) g[r][c] = FF(yi - 1 + r, c);
  } else if (xi == W - 1) {
    for (int r = 0; r < 3; r++) for (int c = 0; c < 2; c++) g[r][1 + c] = FF(yi - 1 + r, W - 2 + c);
  } else if (yi == 0) {
    for (int r = 0; r < 2; r++) for (int c = 0; c < 3; c++) g[r][c] = FF(r, xi - 1 + c);
  } else if (yi == H - 1) {
    for (int r = 0; r < 2; r++) for (int c = 0; c < 3; c++) g[1 + r][c] = FF(H - 2 + r, xi - 1 + c);
  } else {
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) g[r][c] = FF(yi - 1 + r, xi - 1 + c);
  }
#undef FF
  /* np.mean of a 2x2 block = pairwise add.reduce over 4 elements (n<8: sequential) / 4 */
  s[0][0] = (((0. + g[0][0]) + g[0][1]) + g[1][0] + g[1][1]) / 4.0;
  s[0][2] = (((0. + g[0][1]) + g[0][2]) + g[1][1] + g[1][2]) / 4.0;
  s[2][0] = (((0. + g[1][0]) + g[1][1]) + g[2][0] + g[2][1]) / 4.0;
  s[2][2] = (((0. + g[1][1]) + g[1][2]) + g[2][1] + g[2][2]) / 4.0;
  s[0][1] = ((0. + g[0][1]) + g[1][1]) / 2.0;
  s[1][0] = ((0. + g[1][0]) + g[1][1]) / 2.0;
  s[1][2] = ((0. + g[1][1]) + g[1][2]) / 2.0;
  s[2][1] = ((0. + g[1][1]) + g[2][1]) / 2.0;
  s[1][1] = g[1][1];
}

typedef struct { double val, gx, gy, hxx, hyy, hxy; } field_eval_t;

static void field_eval(const solver_t *S, const double *F, double px, double py, field_eval_t *o) {
  const double res = S->cfg->grid_res;
  long xi = round_half_even((px - S->off[0]) / res);
  long yi = round_half_even((py - S->off[1]) / res);
  xi = xi < 0 ? 0 : (xi > S->W - 1 ? S->W - 1 : xi);
  yi = yi < 0 ? 0 : (yi > S->H - 1 ? S->H - 1 : yi);
  double g[3][3];
  smooth_grid(F, S->W, S->H, (int)xi, (int)yi, g);
  const double u = (px - S->gx[xi]) / res + 0.5;
  const double v = (py - S->gy[yi]) / res + 0.5;
  const double u1 = 1 - u, v1 = 1 - v;
  o->val = u1 * u1 * v1 * v1 * g[0][0] + u1 * u1 * 2.0 * v1 * v * g[1][0] + u1 * u1 * v * v * g[2][0] +
           2.0 * u1 * u * v1 * v1 * g[0][1] + 2.0 * u1 * u * 2.0 * v1 * v * g[1][1] + 2.0 * u1 * u * v * v * g[2][1] +
           u * u * v1 * v1 * g[0][2] + u * u * 2.0 * v1 * v * g[1][2] + u * u * v * v * g[2][2];
  const double a = -2.0 + 2.0 * u, b = 2.0 * (1.0 - 2.0 * u), c = u * 2.0;
  const double w1 = 1.0 - v;
  o->gx = 1.0 / res * (a * w1 * w1 * g[0][0] + a * 2.0 * w1 * v * g[1][0] + a * v * v * g[2][0] +
                       b * w1 * w1 * g[0][1] + b * 2.0 * w1 * v * g[1][1] + b * v * v * g[2][1] +
                       c * w1 * w1 * g[0][2] + c * 2.0 * w1 * v * g[1][2] + c * v * v * g[2][2]);
  const double av = -2.0 + 2.0 * v, bv = 2.0 * (1.0 - 2.0 * v), cv = 2.0 * v;
  const double z1 = 1.0 - u;
  o->gy = 1.0 / res * (z1 * z1 * av * g[0][0] + z1 * z1 * bv * g[1][0] + z1 * z1 * cv * g[2][0] +
                       2.0 * z1 * u * av * g[0][1] + 2.0 * z1 * u * bv * g[1][1] + 2.0 * z1 * u * cv * g[2][1] +
                       u * u * av * g[0][2] + u * u * bv * g[1][2] + u * u * cv * g[2][2]);
  const double r2 = 1.0 / (res * res);
  o->hxx = r2 * (2.0 * w1 * w1 * g[0][0] + 2.0 * w1 * 2.0 * v * g[1][0] + 2.0 * v * v * g[2][0] +
                 -4.0 * w1 * w1 * g[0][1] + -4.0 * w1 * 2.0 * v * g[1][1] + -4.0 * v * v * g[2][1] +
                 2.0 * w1 * w1 * g[0][2] + 2.0 * w1 * 2.0 * v * g[1][2] + 2.0 * v * v * g[2][2]);
  o->hyy = r2 * (2.0 * z1 * z1 * g[0][0] + -4.0 * z1 * z1 * g[1][0] + 2.0 * z1 * z1 * g[2][0] +
                 2.0 * z1 * 2.0 * u * g[0][1] + -4.0 * z1 * 2.0 * u * g[1][1] + 2.0 * z1 * 2.0 * u * g[2][1] +
                 2.0 * u * u * g[0][2] + -4.0 * u * u * g[1][2] + 2.0 * u * u * g[2][2]);
  o->hxy = r2 * (a * av * g[0][0] + a * bv * g[1][0] + a * cv * g[2][0] +
                 b * av * g[0][1] + b * bv * g[1][1] + b * cv * g[2][1] +
                 2.0 * u * av * g[0][2] + 2.0 * u * bv * g[1][2] + 2.0 * u * cv * g[2][2]);
}

/* ---- per-node cost (cost.py:341-446 with the potentials of trajectory_tree.py:44-50,110-118) ---- */
static double node_cost(const solver_t *S, int i, const double *x, const double *u, double *lx, double *lxx,
                        double *lu, double *luu) {
  const mind_ilqr_cfg *c = S->cfg;
  const double p = S->prob[i];
  field_eval_t fe;
  field_eval(S, S->field[i], x[0], x[1], &fe);
  double cost = 0;
  cost += fe.val;
  /* StatePotential */
  double xd[NS] = {0, 0, S->target_vel, 0, 0, 0};
  double sp = 0, sc = 0;
  for (int k = 0; k < NS; k++) {
    const double w = c->w_des_state[k] * p, d = x[k] - xd[k];
    sp += d * w * d;
  }
  cost += sp;
  for (int k = 0; k < NS; k++) {
    const double w = c->w_state_con[k] * p;
    const double d = fmax(x[k] - c->state_upper[k], 0) + fmax(c->state_lower[k] - x[k], 0);
    sc += d * w * d;
  }
  cost += sc;
  double cp = 0;
  for (int k = 0; k < NU; k++) cp += u[k] * (c->w_ctrl[k] * p) * u[k];
  cost += cp;
  if (lx) {
    memset(lx, 0, NS * sizeof(double));
    memset(lxx, 0, NS * NS * sizeof(double));
    lx[0] += fe.gx;
    lx[1] += fe.gy;
    lxx[0] += fe.hxx; lxx[1] += fe.hxy; lxx[NS] += fe.hxy; lxx[NS + 1] += fe.hyy;
    for (int k = 0; k < NS; k++) {
      const double w = c->w_des_state[k] * p;
      lx[k] += 2.0 * (w * (x[k] - xd[k]));
      lxx[k * NS + k] += 2.0 * w;
    }
    for (int k = 0; k < NS; k++) {
      const double w = c->w_state_con[k] * p;
      if (x[k] > c->state_upper[k]) { lx[k] += 2.0 * w * (x[k] - c->state_upper[k]); lxx[k * NS + k] += 2.0 * w; }
      else if (x[k] < c->state_lower[k]) { lx[k] += 2.0 * w * (x[k] - c->state_lower[k]); lxx[k * NS + k] += 2.0 * w; }
    }
    for (int k = 0; k < NU; k++) {
      const double w = c->w_ctrl[k] * p;
      lu[k] = 2.0 * (w * u[k]);
      luu[k * NU + k] = 2.0 * w;
    }
    luu[1] = luu[2] = 0;
  }
  return cost;
}

/* ---- dynamics (trajectory_tree.py:168-175) ---- */
static void dyn_f(const mind_ilqr_cfg *c, const double *x, const double *u, double *o) {
  const double dt = c->dt, wb = c->wheelbase;
  double s3, c3;
  mind_sincos(x[3], &s3, &c3);
  o[0] = x[0] + x[2] * c3 * dt;
  o[1] = x[1] + x[2] * s3 * dt;
  o[2] = x[2] + x[4] * dt;
  o[3] = x[3] + x[2] / wb * mind_tan(x[5]) * dt;
  o[4] = x[4] + u[0] * dt;
  o[5] = x[5] + u[1] * dt;
}
static void dyn_fx(const mind_ilqr_cfg *c, const double *x, double *J) {
  const double dt = c->dt, wb = c->wheelbase;
  memset(J, 0, NS * NS * sizeof(double));
  for (int k = 0; k < NS; k++) J[k * NS + k] = 1.0;
  double s3, c3, t5, c5;
  mind_sincos(x[3], &s3, &c3);
  mind_tan_cos(x[5], &t5, &c5);
  J[0 * NS + 2] = c3 * dt;
  J[0 * NS + 3] = -x[2] * s3 * dt;
  J[1 * NS + 2] = s3 * dt;
  J[1 * NS + 3] = x[2] * c3 * dt;
  J[2 * NS + 4] = dt;
  J[3 * NS + 2] = t5 / wb * dt;
  J[3 * NS + 5] = x[2] / wb / (c5 * c5) * dt;
}

typedef struct {
  double *xs, *us, *Fx, *L, *Lx, *Lu, *Lxx, *Luu, *k, *K, *Vx, *Vxx, *xs_new, *us_new;
  double J_opt;
} work_t;

static void forward_rollout(const solver_t *S, work_t *w) {
  /* stack order is irrelevant for the values (each node depends on its parent only); parents have
     smaller keys than children, so key order is a valid topological order (solver.py:283-330). */
  for (int c = 0; c < S->M; c++) {
    const double *px = S->parent[c] < 0 ? S->x0 : w->xs + S->parent[c] * NS;
    double *x = w->xs + c * NS;
    dyn_f(S->cfg, px, w->us + c * NU, x);
    dyn_fx(S->cfg, x, w->Fx + c * NS * NS); /* Jacobian at the POST state (Q1) */
    w->L[c] = node_cost(S, c, x, w->us + c * NU, w->Lx + c * NS, w->Lxx + c * NS * NS, w->Lu + c * NU, w->Luu + c * NU * NU);
  }
  w->J_opt = np_pairwise_sum(w->L, S->M);
}

/* LAPACK-style 2x2 solve with partial pivoting; returns 1 if singular */
static int solve2(const double *A, const double *b, int nrhs, double *x) {
  double a00 = A[0], a01 = A[1], a10 = A[2], a11 = A[3];
  int swap = fabs(a10) > fabs(a00);
  if (swap) { double t; t = a00; a00 = a10; a10 = t; t = a01; a01 = a11; a11 = t; }
  if (a00 == 0.0) return 1;
  const double l = a10 / a00;
  const double u11 = a11 - l * a01;
  if (u11 == 0.0) return 1;
  for (int r = 0; r < nrhs; r++) {
    double b0 = b[r], b1 = b[nrhs + r];
    if (swap) { double t = b0; b0 = b1; b1 = t; }
    b1 = b1 - l * b0;
    const double x1 = b1 / u11;
    const double x0 = (b0 - a01 * x1) / a00;
    x[r] = x0;
    x[nrhs + r] = x1;
  }
  return 0;
}

static int gains(const solver_t *S, work_t *w, int key) {
  const double *fx = w->Fx + key * NS * NS, *lx = w->Lx + key * NS, *lu = w->Lu + key * NU;
  const double *lxx = w->Lxx + key * NS * NS, *luu = w->Luu + key * NU * NU;
  double *Vx = w->Vx + key * NS, *Vxx = w->Vxx + key * NS * NS;
  const double dt = S->cfg->dt;
  double Qx[NS], Qu[NU], Qxx[NS * NS], Qux[NU * NS], Quu[NU * NU];
  /* f_u is zero except [4,0] = [5,1] = dt */
  for (int i = 0; i < NS; i++) {
    double s = 0;
    for (int r = 0; r < NS; r++) s += fx[r * NS + i] * Vx[r];
    Qx[i] = lx[i] + s;
  }
  Qu[0] = lu[0] + dt * Vx[4];
  Qu[1] = lu[1] + dt * Vx[5];
  double T[NS * NS]; /* f_x^T V_xx */
  for (int i = 0; i < NS; i++)
    for (int j = 0; j < NS; j++) {
      double s = 0;
      for (int r = 0; r < NS; r++) s += fx[r * NS + i] * Vxx[r * NS + j];
      T[i * NS + j] = s;
    }
  for (int i = 0; i < NS; i++)
    for (int j = 0; j < NS; j++) {
      double s = 0;
      for (int r = 0; r < NS; r++) s += T[i * NS + r] * fx[r * NS + j];
      Qxx[i * NS + j] = lxx[i * NS + j] + s;
    }
  /* (V_xx + mu I) rows 4 and 5 scaled by dt = f_u^T (V_xx + reg) */
  double R[NU * NS];
  for (int j = 0; j < NS; j++) {
    R[0 * NS + j] = dt * (Vxx[4 * NS + j] + (j == 4 ? S->mu : 0.0));
    R[1 * NS + j] = dt * (Vxx[5 * NS + j] + (j == 5 ? S->mu : 0.0));
  }
  for (int a = 0; a < NU; a++)
    for (int j = 0; j < NS; j++) {
      double s = 0;
      for (int r = 0; r < NS; r++) s += R[a * NS + r] * fx[r * NS + j];
      Qux[a * NS + j] = 0.0 + s;
    }
  for (int a = 0; a < NU; a++)
    for (int b = 0; b < NU; b++) Quu[a * NU + b] = luu[a * NU + b] + R[a * NS + 4 + b] * dt;
  double kk[NU], KK[NU * NS];
  if (solve2(Quu, Qu, 1, kk)) return 1;
  if (solve2(Quu, Qux, NS, KK)) return 1;
  for (int a = 0; a < NU; a++) kk[a] = -kk[a];
  for (int a = 0; a < NU * NS; a++) KK[a] = -KK[a];
  memcpy(w->k + key * NU, kk, sizeof(kk));
  memcpy(w->K + key * NU * NS, KK, sizeof(KK));
  /* V_x = Q_x + K^T Quu k ; += K^T Q_u + Q_ux^T k */
  double Quuk[NU] = {Quu[0] * kk[0] + Quu[1] * kk[1], Quu[2] * kk[0] + Quu[3] * kk[1]};
  for (int i = 0; i < NS; i++) {
    double v = Qx[i] + (KK[0 * NS + i] * Quuk[0] + KK[1 * NS + i] * Quuk[1]);
    v += (KK[0 * NS + i] * Qu[0] + KK[1 * NS + i] * Qu[1]) + (Qux[0 * NS + i] * kk[0] + Qux[1 * NS + i] * kk[1]);
    Vx[i] = v;
  }
  double QuuK[NU * NS];
  for (int j = 0; j < NS; j++) {
    QuuK[0 * NS + j] = Quu[0] * KK[0 * NS + j] + Quu[1] * KK[1 * NS + j];
    QuuK[1 * NS + j] = Quu[2] * KK[0 * NS + j] + Quu[3] * KK[1 * NS + j];
  }
  double Vn[NS * NS];
  for (int i = 0; i < NS; i++)
    for (int j = 0; j < NS; j++) {
      double v = Qxx[i * NS + j] + (KK[0 * NS + i] * QuuK[0 * NS + j] + KK[1 * NS + i] * QuuK[1 * NS + j]);
      v += (KK[0 * NS + i] * Qux[0 * NS + j] + KK[1 * NS + i] * Qux[1 * NS + j]) +
           (Qux[0 * NS + i] * KK[0 * NS + j] + Qux[1 * NS + i] * KK[1 * NS + j]);
      Vn[i * NS + j] = v;
    }
  for (int i = 0; i < NS; i++)
    for (int j = 0; j < NS; j++) Vxx[i * NS + j] = 0.5 * (Vn[i * NS + j] + Vn[j * NS + i]);
  return 0;
}

/* post-order recursion of solver.py:344-350: children in key order; parent accumulates V */
static int backward_rec(const solver_t *S, work_t *w, int key) {
  const int lo = key < 0 ? 0 : S->child_start[key], hi = key < 0 ? 1 : S->child_start[key + 1];
  for (int q = lo; q < hi; q++) {
    const int c = key < 0 ? 0 : S->child_list[q];
    if (backward_rec(S, w, c)) return 1;
    if (gains(S, w, c)) return 1;
    /* root key -1 aliases numpy index -1 = last node (Q2): harmless, reproduced */
    const int tgt = key < 0 ? S->M - 1 : key;
    for (int i = 0; i < NS; i++) w->Vx[tgt * NS + i] += w->Vx[c * NS + i];
    for (int i = 0; i < NS * NS; i++) w->Vxx[tgt * NS * NS + i] += w->Vxx[c * NS * NS + i];
  }
  return 0;
}

static int backward_pass(const solver_t *S, work_t *w) {
  memset(w->Vx, 0, S->M * NS * sizeof(double));
  memset(w->Vxx, 0, S->M * NS * NS * sizeof(double));
  memset(w->k, 0, S->M * NU * sizeof(double));
  memset(w->K, 0, S->M * NU * NS * sizeof(double));
  return backward_rec(S, w, -1);
}

static double line_search(const solver_t *S, work_t *w, double alpha) {
  for (int c = 0; c < S->M; c++) {
    const int p = S->parent[c];
    double *un = w->us_new + c * NU, *xn = w->xs_new + c * NS;
    if (p < 0) {
      for (int a = 0; a < NU; a++) un[a] = w->us[c * NU + a] + alpha * w->k[c * NU + a];
      dyn_f(S->cfg, S->x0, un, xn);
    } else {
      for (int a = 0; a < NU; a++) {
        double s = 0;
        for (int j = 0; j < NS; j++) s += w->K[(c * NU + a) * NS + j] * (w->xs_new[p * NS + j] - w->xs[p * NS + j]);
        un[a] = w->us[c * NU + a] + alpha * w->k[c * NU + a] + s;
      }
      dyn_f(S->cfg, w->xs_new + p * NS, un, xn);
    }
  }
  double J = 0; /* python sum(): sequential from 0 */
  for (int c = 0; c < S->M; c++) J += node_cost(S, c, w->xs_new + c * NS, w->us_new + c * NU, 0, 0, 0, 0);
  return J;
}

static int build_children(solver_t *S) {
  S->child_start = (int *)calloc(S->M + 1, sizeof(int));
  S->child_list = (int *)calloc(S->M > 0 ? S->M : 1, sizeof(int));
  for (int c = 1; c < S->M; c++) {
    if (S->parent[c] < 0 || S->parent[c] >= c) return 1;
    S->child_start[S->parent[c] + 1]++;
  }
  for (int i = 0; i < S->M; i++) S->child_start[i + 1] += S->child_start[i];
  int *fill = (int *)calloc(S->M, sizeof(int));
  for (int c = 1; c < S->M; c++) S->child_list[S->child_start[S->parent[c]] + fill[S->parent[c]]++] = c;
  free(fill);
  return 0;
}

/* materialise the per-node fields like trajectory_tree.py:78-108 (use_exo) / :41-42 (warm start) */
static void build_fields(solver_t *S, const mind_cost_tree *t, const double *quad, int use_exo) {
  const mind_ilqr_cfg *c = S->cfg;
  const int W = S->W, H = S->H, a = t->n_agents;
  for (int i = 0; i < S->M; i++) {
    double *F = S->field[i] = (double *)malloc((size_t)W * H * sizeof(double));
    const float pf = t->prob[i];
    const double wp = (double)(float)((float)c->w_tgt * pf); /* python float * f32 array -> f32 */
    if (!use_exo) {
      for (long q = 0; q < (long)W * H; q++) F[q] = wp * quad[q];
      continue;
    }
    const float *mean = t->agent_mean + (size_t)i * a * 2;
    const float *cov = t->agent_cov + (size_t)i * a;
    const double ego_cov = (double)(float)(cov[0] + (float)c->w_ego_cov_offset);
    for (int r = 0; r < H; r++)
      for (int cc = 0; cc < W; cc++) {
        const double px = S->gx[cc], py = S->gy[r];
        double dx = px - (double)mean[0], dy = py - (double)mean[1];
        double ego = sqrt(dx * dx + dy * dy) - ego_cov;
        ego = ego > 0.0 ? ego : 0.0;
        double covf = 0.0;
        for (int e = 1; e < a; e++) {
          const double ec = (double)(float)(cov[e] + (float)c->w_exo_cov_offset);
          dx = px - (double)mean[2 * e];
          dy = py - (double)mean[2 * e + 1];
          double v = ec - sqrt(dx * dx + dy * dy);
          v = v > 0.0 ? v : 0.0;
          if (v > 0) v += c->w_exo_cost_offset;
          covf += v;
        }
        F[(size_t)r * W + cc] = (wp * quad[(size_t)r * W + cc] + c->w_exo * covf) + c->w_ego * ego;
      }
  }
}

/* per-iteration trace of the last oracle_ilqr_solve (same rows as mind_last_ilqr_trace, include/mind_hip.h): mu of the backward pass,
 * J_opt when the line search starts, accepted alpha index (-1 rejected, -2 LinAlgError), J of the accepted candidate */
#define ORACLE_TRACE_CAP 256
static double oracle_trace[ORACLE_TRACE_CAP][4];
static int oracle_trace_rows;
int oracle_ilqr_last_trace(double *out, int cap_rows) {
  const int n = oracle_trace_rows < cap_rows ? oracle_trace_rows : cap_rows;
  if (n > 0) memcpy(out, oracle_trace, (size_t)n * 4 * sizeof(double));
  return oracle_trace_rows;
}

int oracle_ilqr_solve(const mind_ilqr_cfg *cfg, const mind_cost_tree *t, const double *x0, const double *lane,
                      int P, double target_vel, int use_exo, const double *us_init, double *xs, double *us,
                      mind_ilqr_stats *st, double *J_trace /* [max_iter] or NULL */) {
  solver_t S;
  memset(&S, 0, sizeof(S));
  S.cfg = cfg;
  S.M = t->n_nodes;
  S.parent = t->parent;
  S.W = cfg->grid_w;
  S.H = cfg->grid_h;
  S.target_vel = target_vel;
  memcpy(S.x0, x0, sizeof(S.x0));
  if (build_children(&S)) return -1;
  S.gx = (double *)malloc(S.W * sizeof(double));
  S.gy = (double *)malloc(S.H * sizeof(double));
  make_grid(cfg, x0, S.gx, S.gy, S.off);
  double *quad = (double *)malloc((size_t)S.W * S.H * sizeof(double));
  lane_dist_field(S.gx, S.gy, S.W, S.H, lane, P, quad);
  S.prob = (double *)malloc(S.M * sizeof(double));
  for (int i = 0; i < S.M; i++) S.prob[i] = (double)t->prob[i];
  S.field = (double **)calloc(S.M, sizeof(double *));
  build_fields(&S, t, quad, use_exo);

  const int M = S.M;
  work_t w;
  memset(&w, 0, sizeof(w));
#define AL(n) (double *)calloc((size_t)(n), sizeof(double))
  w.xs = AL(M * NS); w.us = AL(M * NU); w.Fx = AL(M * NS * NS); w.L = AL(M); w.Lx = AL(M * NS); w.Lu = AL(M * NU);
  w.Lxx = AL(M * NS * NS); w.Luu = AL(M * NU * NU); w.k = AL(M * NU); w.K = AL(M * NU * NS); w.Vx = AL(M * NS);
  w.Vxx = AL(M * NS * NS); w.xs_new = AL(M * NS); w.us_new = AL(M * NU);
  if (us_init) memcpy(w.us, us_init, M * NU * sizeof(double));
  S.mu = 1.0;
  S.delta = 2.0;
  double alphas[10];
  for (int j = 0; j < 10; j++) alphas[j] = pow(1.1, -(double)(j * j));
  int accepted = 1, converged = 0, it;
  oracle_trace_rows = 0;
  for (it = 0; it < cfg->max_iter; it++) {
    if (accepted) { forward_rollout(&S, &w); accepted = 0; }
    if (J_trace) J_trace[it] = w.J_opt;
    double *tr = it < ORACLE_TRACE_CAP ? oracle_trace[it] : NULL;
    if (tr) { tr[0] = S.mu; tr[1] = w.J_opt; tr[2] = -1.0; tr[3] = w.J_opt; oracle_trace_rows = it + 1; }
    if (backward_pass(&S, &w)) { if (tr) tr[2] = -2.0; continue; } /* LinAlgError: continue without raising mu (Q9) */
    for (int j = 0; j < 10; j++) {
      const double Jn = line_search(&S, &w, alphas[j]);
      if (Jn < w.J_opt) {
        if (tr) { tr[2] = (double)j; tr[3] = Jn; }
        if (fabs((w.J_opt - Jn) / w.J_opt) < 1e-6) converged = 1;
        accepted = 1;
        memcpy(w.xs, w.xs_new, M * NS * sizeof(double));
        memcpy(w.us, w.us_new, M * NU * sizeof(double));
        S.delta = fmin(1.0, S.delta) / 2.0;
        S.mu *= S.delta;
        if (S.mu <= 1e-6) S.mu = 0.0;
        break;
      }
    }
    if (converged) break;
    if (!accepted) {
      S.delta = fmax(1.0, S.delta) * 2.0;
      S.mu = fmax(1e-6, S.mu * S.delta);
      if (S.mu >= 1e10) break;
    }
  }
  memcpy(xs, w.xs, M * NS * sizeof(double));
  memcpy(us, w.us, M * NU * sizeof(double));
  if (st) { st->iterations = it < cfg->max_iter ? it + 1 : it; st->converged = converged; st->J = w.J_opt; st->mu = S.mu; }
  for (int i = 0; i < M; i++) free(S.field[i]);
  free(S.field); free(S.prob); free(quad); free(S.gx); free(S.gy); free(S.child_start); free(S.child_list);
  free(w.xs); free(w.us); free(w.Fx); free(w.L); free(w.Lx); free(w.Lu); free(w.Lxx); free(w.Luu); free(w.k);
  free(w.K); free(w.Vx); free(w.Vxx); free(w.xs_new); free(w.us_new);
  return 0;
}

/* PotentialField value/gradient/Hessian on an arbitrary field (golden test G4):
 * F [H*W] row-major [y][x]; out = (val, gx, gy, hxx, hyy, hxy) */
int oracle_field_eval(const double *F, int W, int H, double res, const double *origin_xy, double px, double py,
                      double *out) {
  mind_ilqr_cfg cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.grid_res = res; cfg.grid_w = W; cfg.grid_h = H;
  solver_t S;
  memset(&S, 0, sizeof(S));
  S.cfg = &cfg; S.W = W; S.H = H;
  S.gx = (double *)malloc(W * sizeof(double));
  S.gy = (double *)malloc(H * sizeof(double));
  /* caller passes x0 such that offset = origin: x0 = origin + 0.5 * field_size */
  double x0[2] = {origin_xy[0] + 0.5 * ((double)(W - 1) * res), origin_xy[1] + 0.5 * ((double)(H - 1) * res)};
  make_grid(&cfg, x0, S.gx, S.gy, S.off);
  field_eval_t fe;
  field_eval(&S, F, px, py, &fe);
  out[0] = fe.val; out[1] = fe.gx; out[2] = fe.gy; out[3] = fe.hxx; out[4] = fe.hyy; out[5] = fe.hxy;
  free(S.gx); free(S.gy);
  return 0;
}

/* the lane distance field alone (golden test G5 spot values) */
int oracle_lane_field(const mind_ilqr_cfg *cfg, const double *x0, const double *lane, int P, double *quad,
                      double *gx, double *gy, double *off) {
  make_grid(cfg, x0, gx, gy, off);
  lane_dist_field(gx, gy, cfg->grid_w, cfg->grid_h, lane, P, quad);
  return 0;
}

/* per-node costs l(x_i, u_i, i) for given xs/us (debug / golden G5) */
int oracle_node_costs(const mind_ilqr_cfg *cfg, const mind_cost_tree *t, const double *x0, const double *lane, int P,
                      double target_vel, int use_exo, const double *xs, const double *us, double *L) {
  solver_t S;
  memset(&S, 0, sizeof(S));
  S.cfg = cfg; S.M = t->n_nodes; S.parent = t->parent; S.W = cfg->grid_w; S.H = cfg->grid_h; S.target_vel = target_vel;
  memcpy(S.x0, x0, sizeof(S.x0));
  S.gx = (double *)malloc(S.W * sizeof(double));
  S.gy = (double *)malloc(S.H * sizeof(double));
  make_grid(cfg, x0, S.gx, S.gy, S.off);
  double *quad = (double *)malloc((size_t)S.W * S.H * sizeof(double));
  lane_dist_field(S.gx, S.gy, S.W, S.H, lane, P, quad);
  S.prob = (double *)malloc(S.M * sizeof(double));
  for (int i = 0; i < S.M; i++) S.prob[i] = (double)t->prob[i];
  S.field = (double **)calloc(S.M, sizeof(double *));
  build_fields(&S, t, quad, use_exo);
  for (int i = 0; i < S.M; i++) L[i] = node_cost(&S, i, xs + i * NS, us + i * NU, 0, 0, 0, 0);
  for (int i = 0; i < S.M; i++) free(S.field[i]);
  free(S.field); free(S.prob); free(quad); free(S.gx); free(S.gy);
  return 0;
}

/* the materialised per-node cost fields alone (what trajectory_tree.py:41-42 / :78-108 hands to
 * PotentialField), for tests of the generic planners/ilqr surface: fields [M,H,W], gx [W], gy [H], off [2] */
int oracle_node_fields(const mind_ilqr_cfg *cfg, const mind_cost_tree *t, const double *x0, const double *lane, int P,
                       int use_exo, double *fields, double *gx, double *gy, double *off) {
  solver_t S;
  memset(&S, 0, sizeof(S));
  S.cfg = cfg; S.M = t->n_nodes; S.parent = t->parent; S.W = cfg->grid_w; S.H = cfg->grid_h;
  S.gx = gx; S.gy = gy;
  make_grid(cfg, x0, S.gx, S.gy, S.off);
  off[0] = S.off[0]; off[1] = S.off[1];
  double *quad = (double *)malloc((size_t)S.W * S.H * sizeof(double));
  lane_dist_field(S.gx, S.gy, S.W, S.H, lane, P, quad);
  S.field = (double **)calloc(S.M, sizeof(double *));
  build_fields(&S, t, quad, use_exo);
  for (int i = 0; i < S.M; i++) {
    memcpy(fields + (size_t)i * S.W * S.H, S.field[i], (size_t)S.W * S.H * sizeof(double));
    free(S.field[i]);
  }
  free(S.field); free(quad);
  return 0;
}

/* per-node cost AND derivatives at given xs/us (cost.py:341-446):
 * out [M, 47] = { l, l_x[6], l_u[2], l_xx[36], diag l_uu[2] } */
int oracle_node_derivs(const mind_ilqr_cfg *cfg, const mind_cost_tree *t, const double *x0, const double *lane, int P,
                       double target_vel, int use_exo, const double *xs, const double *us, double *out) {
  solver_t S;
  memset(&S, 0, sizeof(S));
  S.cfg = cfg; S.M = t->n_nodes; S.parent = t->parent; S.W = cfg->grid_w; S.H = cfg->grid_h; S.target_vel = target_vel;
  memcpy(S.x0, x0, sizeof(S.x0));
  S.gx = (double *)malloc(S.W * sizeof(double));
  S.gy = (double *)malloc(S.H * sizeof(double));
  make_grid(cfg, x0, S.gx, S.gy, S.off);
  double *quad = (double *)malloc((size_t)S.W * S.H * sizeof(double));
  lane_dist_field(S.gx, S.gy, S.W, S.H, lane, P, quad);
  S.prob = (double *)malloc(S.M * sizeof(double));
  for (int i = 0; i < S.M; i++) S.prob[i] = (double)t->prob[i];
  S.field = (double **)calloc(S.M, sizeof(double *));
  build_fields(&S, t, quad, use_exo);
  for (int i = 0; i < S.M; i++) {
    double *o = out + (size_t)i * 47, luu[4];
    o[0] = node_cost(&S, i, xs + i * NS, us + i * NU, o + 1, o + 9, o + 7, luu);
    o[45] = luu[0]; o[46] = luu[3];
  }
  for (int i = 0; i < S.M; i++) free(S.field[i]);
  free(S.field); free(S.prob); free(quad); free(S.gx); free(S.gy);
  return 0;
}
