// Tree-iLQR contingency solves for gfx950: float64, one workgroup per cost tree, the whole
// iLQR.fit loop (rollout / Riccati sweep / line search / LM schedule) in ONE persistent launch.
//
// Reference semantics: planners/ilqr/solver.py:80-421 (tree-aware iLQR), planners/ilqr/cost.py:326-446,
// planners/ilqr/potential.py (PotentialField + quadratic potentials), planners/mind/trajectory_tree.py:
// 19-177 (cost-tree construction, bicycle model).  Quirks Q1 (Jacobian at the post state), Q9, Q10,
// Q19 are reproduced; see DESIGN.md "k8-k12".
//
// Design:
//  * the reference materialises a 256x256 float64 cost field PER TRAJECTORY NODE (75% of its time);
//    here only the target-lane distance field (shared by all nodes/trees of a plan) is materialised
//    once per call (k_lane_field, 512 KB); the exo/ego terms of the 3x3 cell window a query touches
//    are evaluated analytically at query time -- identical values, ~10^4 x less work.
//  * tree levels are processed synchronously: nodes of one depth in parallel waves; inside a wave
//    the 9-cell x agents distance work and the 6x6 Riccati algebra are spread over the 64 lanes.
//  * all 10 line-search step sizes are rolled out concurrently, then the FIRST improving one is
//    taken, which is what the reference's sequential search returns.
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

#define IL_NS 6
#define IL_NU 2
#define IL_NA 10      // line-search candidates
#define IL_THREADS 512
#define IL_WAVES (IL_THREADS / 64)

struct IlqrTreeDev {
  int M, n_agents, n_levels, pad;
  const int *parent;        // [M]
  const int *level_start;   // [n_levels+1]
  const int *level_nodes;   // [M] nodes sorted by depth (ties: key)
  const int *child_start;   // [M+1]
  const int *child_list;    // [M-1] children in key order
  const float *prob;        // [M]
  const float *mean;        // [M,a,2]
  const float *cov;         // [M,a]
  // workspace (doubles)
  double *xs, *us, *Fx, *L, *Lx, *Lxx, *k, *K, *Vx, *Vxx;   // [M,*]
  double *xs_new, *us_new, *L_new;                            // [NA,M,*]
  // outputs
  double *stats;            // [4]: iterations, converged, J, mu
};

struct IlqrConst {
  double dt, wb;
  double w_des[6], w_con[6], lb[6], ub[6], w_ctrl[2];
  double w_tgt, w_ego, w_ego_off, w_exo, w_exo_off, w_exo_cost;
  double res, off_x, off_y, target_vel;
  double x0[6];
  int W, H, max_iter, use_exo;
  double alphas[IL_NA];     // 1.1 ** (-j*j), j = 0..9 (solver.py:125), computed on the host
  const double *gx, *gy;    // [W], [H] grid coordinates (numpy linspace + offset, built on the host)
  const double *quad;       // [H*W] squared distance to the target lane
};

#define IL_WFENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

__device__ __forceinline__ double il_shfl_down(double v, int d) {
  return __shfl_down(v, d, 64);
}

// ---- lane distance field: d(c)^2 = min over segments (ilqr/utils.py:5-22, geometry.py:70-78) ----
__global__ void k_lane_field(const double *__restrict__ gx, const double *__restrict__ gy, int W, int H,
                             const double *__restrict__ lane, int P, double *__restrict__ quad) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= W * H) return;
  const double px = gx[idx % W], py = gy[idx / W];
  double d = INFINITY;
  for (int j = 0; j < P - 1; ++j) {
    const double ax = lane[2 * j], ay = lane[2 * j + 1];
    const double lvx = lane[2 * j + 2] - ax, lvy = lane[2 * j + 3] - ay;
    const double len2 = lvx * lvx + lvy * lvy;
    double t = ((px - ax) * lvx + (py - ay) * lvy) / len2;
    t = t < 0 ? 0 : (t > 1 ? 1 : t);
    const double dx = px - (ax + t * lvx), dy = py - (ay + t * lvy);
    const double dj = sqrt(dx * dx + dy * dy);
    d = dj < d ? dj : d;
  }
  quad[idx] = d * d;
}

// ---- 3x3 window placement, reproducing potential.py:126-144 exactly (Q4: borders are NOT zero
//      padding).  For window entry (r,c) returns the source field cell or -1 for "stays zero". ----
__device__ __forceinline__ void il_window_src(int xi, int yi, int W, int H, int r, int c, int &sy, int &sx) {
  sy = -1; sx = -1;
  const bool x0 = xi == 0, xL = xi == W - 1, y0 = yi == 0, yL = yi == H - 1;
  if (x0 && y0) { if (r >= 1 && c >= 1) { sy = r - 1; sx = c - 1; } }
  else if (x0 && yL) { if (r >= 1 && c <= 1) { sy = H - 2 + (r - 1); sx = c; } }
  else if (xL && y0) { if (r <= 1 && c >= 1) { sy = r; sx = W - 2 + (c - 1); } }
  else if (xL && yL) { if (r <= 1 && c <= 1) { sy = H - 2 + r; sx = W - 2 + c; } }
  else if (x0) { if (c <= 1) { sy = yi - 1 + r; sx = c; } }
  else if (xL) { if (c >= 1) { sy = yi - 1 + r; sx = W - 2 + (c - 1); } }
  else if (y0) { if (r <= 1) { sy = r; sx = xi - 1 + c; } }
  else if (yL) { if (r >= 1) { sy = H - 2 + (r - 1); sx = xi - 1 + c; } }
  else { sy = yi - 1 + r; sx = xi - 1 + c; }
}

struct FieldOut { double val, gx, gy, hxx, hyy, hxy; };

// Cooperative (one wave) evaluation of node `i`'s potential field at (px, py).
// cells[9] (LDS, per wave) receives the raw window; every lane returns the same FieldOut.
__device__ __forceinline__ void il_field(const IlqrConst &C, const IlqrTreeDev &T, int i, double px, double py,
                                         double *scr /* >= 64+9 doubles, per wave */, bool want_deriv, FieldOut &o) {
  const int lane = threadIdx.x & 63;
  long xi = (long)rint((px - C.off_x) / C.res);
  long yi = (long)rint((py - C.off_y) / C.res);
  xi = xi < 0 ? 0 : (xi > C.W - 1 ? C.W - 1 : xi);
  yi = yi < 0 ? 0 : (yi > C.H - 1 ? C.H - 1 : yi);
  const float pf = T.prob[i];
  const double wp = (double)((float)C.w_tgt * pf);
  // lanes 0..62: cell = lane % 9, agent slot = lane / 9 (7 slots)
  const int cell = lane % 9, slot = lane / 9;
  int sy, sx;
  il_window_src((int)xi, (int)yi, C.W, C.H, cell / 3, cell % 3, sy, sx);
  double part = 0.0;
  if (lane < 63 && sy >= 0 && C.use_exo) {
    const double cx = C.gx[sx], cy = C.gy[sy];
    const float *mean = T.mean + (size_t)i * T.n_agents * 2;
    const float *cov = T.cov + (size_t)i * T.n_agents;
    for (int e = 1 + slot; e < T.n_agents; e += 7) {
      const double ec = (double)(cov[e] + (float)C.w_exo_off);
      const double dx = cx - (double)mean[2 * e], dy = cy - (double)mean[2 * e + 1];
      double v = ec - sqrt(dx * dx + dy * dy);
      v = v > 0.0 ? v : 0.0;
      if (v > 0.0) v += C.w_exo_cost;
      part += v;
    }
  }
  IL_WFENCE();
  scr[lane] = part;
  IL_WFENCE();
  if (lane < 9) {
    double cell_v = 0.0;
    il_window_src((int)xi, (int)yi, C.W, C.H, lane / 3, lane % 3, sy, sx);
    if (sy >= 0) {
      double covf = 0.0;
      for (int s = 0; s < 7; ++s) covf += scr[lane + 9 * s];
      const double q = C.quad[(size_t)sy * C.W + sx];
      if (C.use_exo) {
        const float *mean = T.mean + (size_t)i * T.n_agents * 2;
        const float *cov = T.cov + (size_t)i * T.n_agents;
        const double ego_cov = (double)(cov[0] + (float)C.w_ego_off);
        const double dx = C.gx[sx] - (double)mean[0], dy = C.gy[sy] - (double)mean[1];
        double ego = sqrt(dx * dx + dy * dy) - ego_cov;
        ego = ego > 0.0 ? ego : 0.0;
        cell_v = (wp * q + C.w_exo * covf) + C.w_ego * ego;
      } else {
        cell_v = wp * q;
      }
    }
    scr[64 + lane] = cell_v;
  }
  IL_WFENCE();
  double g[3][3], s[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) g[r][c] = scr[64 + r * 3 + c];
  s[0][0] = (((g[0][0] + g[0][1]) + g[1][0]) + g[1][1]) / 4.0;
  s[0][2] = (((g[0][1] + g[0][2]) + g[1][1]) + g[1][2]) / 4.0;
  s[2][0] = (((g[1][0] + g[1][1]) + g[2][0]) + g[2][1]) / 4.0;
  s[2][2] = (((g[1][1] + g[1][2]) + g[2][1]) + g[2][2]) / 4.0;
  s[0][1] = (g[0][1] + g[1][1]) / 2.0;
  s[1][0] = (g[1][0] + g[1][1]) / 2.0;
  s[1][2] = (g[1][1] + g[1][2]) / 2.0;
  s[2][1] = (g[1][1] + g[2][1]) / 2.0;
  s[1][1] = g[1][1];
  const double res = C.res;
  const double u = (px - C.gx[xi]) / res + 0.5;
  const double v = (py - C.gy[yi]) / res + 0.5;
  const double u1 = 1 - u, v1 = 1 - v;
  o.val = u1 * u1 * v1 * v1 * s[0][0] + u1 * u1 * 2.0 * v1 * v * s[1][0] + u1 * u1 * v * v * s[2][0] +
          2.0 * u1 * u * v1 * v1 * s[0][1] + 2.0 * u1 * u * 2.0 * v1 * v * s[1][1] + 2.0 * u1 * u * v * v * s[2][1] +
          u * u * v1 * v1 * s[0][2] + u * u * 2.0 * v1 * v * s[1][2] + u * u * v * v * s[2][2];
  if (want_deriv) {
    const double a = -2.0 + 2.0 * u, b = 2.0 * (1.0 - 2.0 * u), c = u * 2.0;
    o.gx = 1.0 / res * (a * v1 * v1 * s[0][0] + a * 2.0 * v1 * v * s[1][0] + a * v * v * s[2][0] +
                        b * v1 * v1 * s[0][1] + b * 2.0 * v1 * v * s[1][1] + b * v * v * s[2][1] +
                        c * v1 * v1 * s[0][2] + c * 2.0 * v1 * v * s[1][2] + c * v * v * s[2][2]);
    const double av = -2.0 + 2.0 * v, bv = 2.0 * (1.0 - 2.0 * v), cv = 2.0 * v;
    o.gy = 1.0 / res * (u1 * u1 * av * s[0][0] + u1 * u1 * bv * s[1][0] + u1 * u1 * cv * s[2][0] +
                        2.0 * u1 * u * av * s[0][1] + 2.0 * u1 * u * bv * s[1][1] + 2.0 * u1 * u * cv * s[2][1] +
                        u * u * av * s[0][2] + u * u * bv * s[1][2] + u * u * cv * s[2][2]);
    const double r2 = 1.0 / (res * res);
    o.hxx = r2 * (2.0 * v1 * v1 * s[0][0] + 2.0 * v1 * 2.0 * v * s[1][0] + 2.0 * v * v * s[2][0] +
                  -4.0 * v1 * v1 * s[0][1] + -4.0 * v1 * 2.0 * v * s[1][1] + -4.0 * v * v * s[2][1] +
                  2.0 * v1 * v1 * s[0][2] + 2.0 * v1 * 2.0 * v * s[1][2] + 2.0 * v * v * s[2][2]);
    o.hyy = r2 * (2.0 * u1 * u1 * s[0][0] + -4.0 * u1 * u1 * s[1][0] + 2.0 * u1 * u1 * s[2][0] +
                  2.0 * u1 * 2.0 * u * s[0][1] + -4.0 * u1 * 2.0 * u * s[1][1] + 2.0 * u1 * 2.0 * u * s[2][1] +
                  2.0 * u * u * s[0][2] + -4.0 * u * u * s[1][2] + 2.0 * u * u * s[2][2]);
    o.hxy = r2 * (a * av * s[0][0] + a * bv * s[1][0] + a * cv * s[2][0] +
                  b * av * s[0][1] + b * bv * s[1][1] + b * cv * s[2][1] +
                  2.0 * u * av * s[0][2] + 2.0 * u * bv * s[1][2] + 2.0 * u * cv * s[2][2]);
  }
}

__device__ __forceinline__ void il_dyn(const IlqrConst &C, const double *x, const double *u, double *o) {
  o[0] = x[0] + x[2] * cos(x[3]) * C.dt;
  o[1] = x[1] + x[2] * sin(x[3]) * C.dt;
  o[2] = x[2] + x[4] * C.dt;
  o[3] = x[3] + x[2] / C.wb * tan(x[5]) * C.dt;
  o[4] = x[4] + u[0] * C.dt;
  o[5] = x[5] + u[1] * C.dt;
}

// quadratic potentials (potential.py:4-59) + field value; derivatives optional (lane 0 writes)
__device__ __forceinline__ double il_node_cost(const IlqrConst &C, double p, const double *x, const double *u,
                                               const FieldOut &fe, double *lx, double *lxx, double *lu_diag /*[4]: lu0, lu1, luu00, luu11*/) {
  double cost = 0.0;
  cost += fe.val;
  double sp = 0.0, sc = 0.0, cp = 0.0;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const double xd = (k == 2) ? C.target_vel : 0.0;
    const double w = C.w_des[k] * p, d = x[k] - xd;
    sp += d * w * d;
  }
  cost += sp;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const double w = C.w_con[k] * p;
    const double d = fmax(x[k] - C.ub[k], 0.0) + fmax(C.lb[k] - x[k], 0.0);
    sc += d * w * d;
  }
  cost += sc;
#pragma unroll
  for (int k = 0; k < 2; ++k) cp += u[k] * (C.w_ctrl[k] * p) * u[k];
  cost += cp;
  if (lx) {
#pragma unroll
    for (int k = 0; k < 36; ++k) lxx[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) lx[k] = 0.0;
    lx[0] += fe.gx; lx[1] += fe.gy;
    lxx[0] += fe.hxx; lxx[1] += fe.hxy; lxx[6] += fe.hxy; lxx[7] += fe.hyy;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const double xd = (k == 2) ? C.target_vel : 0.0;
      const double w = C.w_des[k] * p;
      lx[k] += 2.0 * (w * (x[k] - xd));
      lxx[k * 6 + k] += 2.0 * w;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const double w = C.w_con[k] * p;
      if (x[k] > C.ub[k]) { lx[k] += 2.0 * w * (x[k] - C.ub[k]); lxx[k * 6 + k] += 2.0 * w; }
      else if (x[k] < C.lb[k]) { lx[k] += 2.0 * w * (x[k] - C.lb[k]); lxx[k * 6 + k] += 2.0 * w; }
    }
    lu_diag[0] = 2.0 * ((C.w_ctrl[0] * p) * u[0]);
    lu_diag[1] = 2.0 * ((C.w_ctrl[1] * p) * u[1]);
    lu_diag[2] = 2.0 * (C.w_ctrl[0] * p);
    lu_diag[3] = 2.0 * (C.w_ctrl[1] * p);
  }
  return cost;
}

// numpy add.reduce order (pairwise, blocks of 8/128) so that J_opt equals the reference's L.sum()
__device__ double il_np_sum(const double *a, long n) {
  if (n < 8) {
    double r = 0.0;
    for (long i = 0; i < n; ++i) r += a[i];
    return r;
  } else if (n <= 128) {
    double r[8];
    long i;
    for (i = 0; i < 8; ++i) r[i] = a[i];
    for (i = 8; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  long n2 = n / 2;
  n2 -= n2 % 8;
  return il_np_sum(a, n2) + il_np_sum(a + n2, n - n2);
}

// Riccati step for one node, cooperative over one wave (solver.py:352-421).  Vx/Vxx of `key` hold the
// children sums on entry and the node's value function on exit.  Returns (wave-uniform) 1 if singular.
__device__ __forceinline__ int il_gains(const IlqrConst &C, const IlqrTreeDev &T, int key, double mu, double *scr) {
  const int lane = threadIdx.x & 63;
  const int i = lane / 6, j = lane % 6;   // lanes 0..35 <-> (i,j)
  double *fx = scr, *Vxx = scr + 36, *Tm = scr + 72, *Qxx = scr + 108, *Qux = scr + 144, *Kk = scr + 156;
  double *Qx = scr + 170, *Vx = scr + 176, *misc = scr + 182;   // misc: Qu[2], Quu[4], k[2], flag
  IL_WFENCE();
  if (lane < 36) { fx[lane] = T.Fx[(size_t)key * 36 + lane]; Vxx[lane] = T.Vxx[(size_t)key * 36 + lane]; }
  if (lane < 6) Vx[lane] = T.Vx[(size_t)key * 6 + lane];
  IL_WFENCE();
  const double dt = C.dt;
  if (lane < 36) {
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < 6; ++r) s += fx[r * 6 + i] * Vxx[r * 6 + j];
    Tm[lane] = s;
  }
  if (lane >= 36 && lane < 42) {  // Q_x
    const int a = lane - 36;
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < 6; ++r) s += fx[r * 6 + a] * Vx[r];
    Qx[a] = T.Lx[(size_t)key * 6 + a] + s;
  }
  if (lane >= 42 && lane < 54) {  // Q_ux = f_u^T (V_xx + mu I) f_x ; f_u^T picks rows 4,5 scaled by dt
    const int a = (lane - 42) / 6, jj = (lane - 42) % 6;
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const double R = dt * (Vxx[(4 + a) * 6 + r] + ((r == 4 + a) ? mu : 0.0));
      s += R * fx[r * 6 + jj];
    }
    Qux[a * 6 + jj] = 0.0 + s;
  }
  if (lane >= 54 && lane < 58) {  // Q_uu
    const int a = (lane - 54) / 2, b = (lane - 54) % 2;
    const double R = dt * (Vxx[(4 + a) * 6 + 4 + b] + ((a == b) ? mu : 0.0));
    const double luu = (a == b) ? 2.0 * (C.w_ctrl[a] * (double)T.prob[key]) : 0.0;
    misc[2 + a * 2 + b] = luu + R * dt;
  }
  if (lane >= 58 && lane < 60) {  // Q_u
    const int a = lane - 58;
    const double lu = 2.0 * ((C.w_ctrl[a] * (double)T.prob[key]) * T.us[(size_t)key * 2 + a]);
    misc[a] = lu + dt * Vx[4 + a];
  }
  IL_WFENCE();
  if (lane < 36) {
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < 6; ++r) s += Tm[i * 6 + r] * fx[r * 6 + j];
    Qxx[lane] = T.Lxx[(size_t)key * 36 + lane] + s;
  }
  // 2x2 solves with partial pivoting (LAPACK dgesv order): lanes 0..6 each own one right-hand side
  int singular = 0;
  {
    double a00 = misc[2], a01 = misc[3], a10 = misc[4], a11 = misc[5];
    const bool swp = fabs(a10) > fabs(a00);
    if (swp) { double t = a00; a00 = a10; a10 = t; t = a01; a01 = a11; a11 = t; }
    const double l = a10 / a00;
    const double u11 = a11 - l * a01;
    singular = (a00 == 0.0) || (u11 == 0.0);
    if (lane < 7 && !singular) {
      double b0 = lane < 6 ? Qux[lane] : misc[0];
      double b1 = lane < 6 ? Qux[6 + lane] : misc[1];
      if (swp) { double t = b0; b0 = b1; b1 = t; }
      b1 = b1 - l * b0;
      const double x1 = b1 / u11;
      const double x0 = (b0 - a01 * x1) / a00;
      if (lane < 6) { Kk[lane] = -x0; Kk[6 + lane] = -x1; }
      else { misc[6] = -x0; misc[7] = -x1; }
    }
  }
  IL_WFENCE();
  if (singular) return 1;
  const double k0 = misc[6], k1 = misc[7];
  const double q00 = misc[2], q01 = misc[3], q10 = misc[4], q11 = misc[5];
  if (lane < 36) {
    const double QuuK0j = q00 * Kk[j] + q01 * Kk[6 + j];
    const double QuuK1j = q10 * Kk[j] + q11 * Kk[6 + j];
    double v = Qxx[lane] + (Kk[i] * QuuK0j + Kk[6 + i] * QuuK1j);
    v += (Kk[i] * Qux[j] + Kk[6 + i] * Qux[6 + j]) + (Qux[i] * Kk[j] + Qux[6 + i] * Kk[6 + j]);
    Tm[lane] = v;
  }
  if (lane >= 36 && lane < 42) {
    const int a = lane - 36;
    const double Quuk0 = q00 * k0 + q01 * k1, Quuk1 = q10 * k0 + q11 * k1;
    double v = Qx[a] + (Kk[a] * Quuk0 + Kk[6 + a] * Quuk1);
    v += (Kk[a] * misc[0] + Kk[6 + a] * misc[1]) + (Qux[a] * k0 + Qux[6 + a] * k1);
    T.Vx[(size_t)key * 6 + a] = v;
  }
  if (lane >= 42 && lane < 54) T.K[(size_t)key * 12 + (lane - 42)] = Kk[lane - 42];
  if (lane == 54) { T.k[(size_t)key * 2] = k0; T.k[(size_t)key * 2 + 1] = k1; }
  IL_WFENCE();
  if (lane < 36) T.Vxx[(size_t)key * 36 + lane] = 0.5 * (Tm[i * 6 + j] + Tm[j * 6 + i]);
  return 0;
}

__global__ __launch_bounds__(IL_THREADS) void k_ilqr(const IlqrTreeDev *__restrict__ trees, IlqrConst C) {
  const IlqrTreeDev T = trees[blockIdx.x];
  __shared__ double scr_all[IL_WAVES][192];
  __shared__ double Jnew[IL_NA];
  __shared__ double sh_mu, sh_delta, sh_J;
  __shared__ int sh_accepted, sh_converged, sh_stop, sh_sing, sh_pick;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double *scr = scr_all[wave];
  const int M = T.M;
  if (tid == 0) { sh_mu = 1.0; sh_delta = 2.0; sh_accepted = 1; sh_converged = 0; sh_stop = 0; sh_J = 0.0; }
  __syncthreads();
  int it = 0;
  for (it = 0; it < C.max_iter; ++it) {
    // ---------------- forward rollout with derivatives (solver.py:255-330) ----------------
    if (sh_accepted) {
      for (int d = 0; d < T.n_levels; ++d) {
        const int lo = T.level_start[d], hi = T.level_start[d + 1];
        for (int q = lo + wave; q < hi; q += IL_WAVES) {
          const int c = T.level_nodes[q];
          const int p = T.parent[c];
          double xp[6], u[2], x[6];
#pragma unroll
          for (int k = 0; k < 6; ++k) xp[k] = p < 0 ? C.x0[k] : T.xs[(size_t)p * 6 + k];
          u[0] = T.us[(size_t)c * 2]; u[1] = T.us[(size_t)c * 2 + 1];
          il_dyn(C, xp, u, x);
          FieldOut fe;
          il_field(C, T, c, x[0], x[1], scr, true, fe);
          if (lane == 0) {
            double lx[6], lxx[36], lud[4];
            const double Lc = il_node_cost(C, (double)T.prob[c], x, u, fe, lx, lxx, lud);
            T.L[c] = Lc;
#pragma unroll
            for (int k = 0; k < 6; ++k) { T.xs[(size_t)c * 6 + k] = x[k]; T.Lx[(size_t)c * 6 + k] = lx[k]; }
#pragma unroll
            for (int k = 0; k < 36; ++k) T.Lxx[(size_t)c * 36 + k] = lxx[k];
            // f_x at the POST state (Q1)
            double J[36];
#pragma unroll
            for (int k = 0; k < 36; ++k) J[k] = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) J[k * 6 + k] = 1.0;
            J[0 * 6 + 2] = cos(x[3]) * C.dt;
            J[0 * 6 + 3] = -x[2] * sin(x[3]) * C.dt;
            J[1 * 6 + 2] = sin(x[3]) * C.dt;
            J[1 * 6 + 3] = x[2] * cos(x[3]) * C.dt;
            J[2 * 6 + 4] = C.dt;
            J[3 * 6 + 2] = tan(x[5]) / C.wb * C.dt;
            J[3 * 6 + 5] = x[2] / C.wb / (cos(x[5]) * cos(x[5])) * C.dt;
#pragma unroll
            for (int k = 0; k < 36; ++k) T.Fx[(size_t)c * 36 + k] = J[k];
          }
        }
        __threadfence_block();
        __syncthreads();
      }
      if (tid == 0) { sh_J = il_np_sum(T.L, M); sh_accepted = 0; }
      __syncthreads();
    }
    // ---------------- backward pass (solver.py:332-373), deepest level first ----------------
    for (int q = tid; q < M * 6; q += IL_THREADS) T.Vx[q] = 0.0;
    for (int q = tid; q < M * 36; q += IL_THREADS) T.Vxx[q] = 0.0;
    if (tid == 0) sh_sing = 0;
    __threadfence_block();
    __syncthreads();
    const double mu = sh_mu;
    for (int d = T.n_levels - 1; d >= 0; --d) {
      const int lo = T.level_start[d], hi = T.level_start[d + 1];
      for (int q = lo + wave; q < hi; q += IL_WAVES) {
        const int c = T.level_nodes[q];
        // V[c] <- sum of children's value functions (children in key order, from zero)
        if (lane < 42) {
          double acc = 0.0;
          for (int e = T.child_start[c]; e < T.child_start[c + 1]; ++e) {
            const int ch = T.child_list[e];
            acc += lane < 36 ? T.Vxx[(size_t)ch * 36 + lane] : T.Vx[(size_t)ch * 6 + (lane - 36)];
          }
          if (lane < 36) T.Vxx[(size_t)c * 36 + lane] = acc; else T.Vx[(size_t)c * 6 + (lane - 36)] = acc;
        }
        __threadfence_block();
        if (il_gains(C, T, c, mu, scr) && lane == 0) sh_sing = 1;
      }
      __threadfence_block();
      __syncthreads();
    }
    if (sh_sing) { __syncthreads(); continue; }   // LinAlgError: retry without raising mu (Q9)
    // ---------------- line search: all 10 alphas rolled out concurrently (solver.py:180-253) -------
    for (int d = 0; d < T.n_levels; ++d) {
      const int lo = T.level_start[d], hi = T.level_start[d + 1];
      const int items = (hi - lo) * IL_NA;
      for (int w = wave; w < items; w += IL_WAVES) {
        const int a = w % IL_NA, c = T.level_nodes[lo + w / IL_NA];
        const int p = T.parent[c];
        const double alpha = C.alphas[a];
        double *xn = T.xs_new + ((size_t)a * M + c) * 6, *un = T.us_new + ((size_t)a * M + c) * 2;
        double x[6], u[2], xp[6];
        if (p < 0) {
#pragma unroll
          for (int k = 0; k < 6; ++k) xp[k] = C.x0[k];
          u[0] = T.us[(size_t)c * 2] + alpha * T.k[(size_t)c * 2];
          u[1] = T.us[(size_t)c * 2 + 1] + alpha * T.k[(size_t)c * 2 + 1];
        } else {
          const double *xpn = T.xs_new + ((size_t)a * M + p) * 6;
#pragma unroll
          for (int k = 0; k < 6; ++k) xp[k] = xpn[k];
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            double s = 0.0;
#pragma unroll
            for (int jj = 0; jj < 6; ++jj) s += T.K[((size_t)c * 2 + b) * 6 + jj] * (xp[jj] - T.xs[(size_t)p * 6 + jj]);
            u[b] = T.us[(size_t)c * 2 + b] + alpha * T.k[(size_t)c * 2 + b] + s;
          }
        }
        il_dyn(C, xp, u, x);
        FieldOut fe;
        il_field(C, T, c, x[0], x[1], scr, false, fe);
        if (lane == 0) {
          const double Lc = il_node_cost(C, (double)T.prob[c], x, u, fe, nullptr, nullptr, nullptr);
          T.L_new[(size_t)a * M + c] = Lc;
#pragma unroll
          for (int k = 0; k < 6; ++k) xn[k] = x[k];
          un[0] = u[0]; un[1] = u[1];
        }
      }
      __threadfence_block();
      __syncthreads();
    }
    if (tid < IL_NA) {
      double J = 0.0;   // python sum(): sequential
      const double *Ln = T.L_new + (size_t)tid * M;
      for (int c = 0; c < M; ++c) J += Ln[c];
      Jnew[tid] = J;
    }
    __syncthreads();
    if (tid == 0) {
      int pick = -1;
      for (int a = 0; a < IL_NA; ++a)
        if (Jnew[a] < sh_J) { pick = a; break; }
      sh_pick = pick;
      if (pick >= 0) {
        if (fabs((sh_J - Jnew[pick]) / sh_J) < 1e-6) sh_converged = 1;
        sh_accepted = 1;
        sh_delta = fmin(1.0, sh_delta) / 2.0;
        sh_mu *= sh_delta;
        if (sh_mu <= 1e-6) sh_mu = 0.0;
      } else {
        sh_delta = fmax(1.0, sh_delta) * 2.0;
        sh_mu = fmax(1e-6, sh_mu * sh_delta);
        if (sh_mu >= 1e10) sh_stop = 1;
      }
    }
    __syncthreads();
    if (sh_pick >= 0) {
      const double *xn = T.xs_new + (size_t)sh_pick * M * 6, *un = T.us_new + (size_t)sh_pick * M * 2;
      for (int q = tid; q < M * 6; q += IL_THREADS) T.xs[q] = xn[q];
      for (int q = tid; q < M * 2; q += IL_THREADS) T.us[q] = un[q];
      __threadfence_block();
    }
    __syncthreads();
    if (sh_converged || sh_stop) break;
  }
  if (tid == 0) {
    T.stats[0] = (double)(it < C.max_iter ? it + 1 : it);
    T.stats[1] = (double)sh_converged;
    T.stats[2] = sh_J;
    T.stats[3] = sh_mu;
  }
}
