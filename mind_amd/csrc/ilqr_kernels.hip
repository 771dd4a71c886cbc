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
#include "mind_trig.h"      // sin / cos / tan shared with oracle/ilqr_ref.c (same operations => same bits)

#define IL_NS 6
#define IL_NU 2
#define IL_NA 10      // line-search candidates
#define IL_THREADS 512
#define IL_WAVES (IL_THREADS / 64)
#ifndef IL_SPEC_OVERSUB
#define IL_SPEC_OVERSUB 3   // (slot, segment) items per wave tolerated at the widest segment level
#endif
#define IL_SPEC 4     // Levenberg-Marquardt values evaluated speculatively per step INSIDE one workgroup (see k_ilqr)
#define IL_SLOTS 12   // ... and by the follower workgroups of a tree (k_ilqr<GEN, 2>: one Levenberg-Marquardt value per workgroup); per-slot arrays
                      // hold max(workgroups per tree, IL_SPEC) sets in that mode (mind_hip.hip: what the launch can use)
#define IL_MAXA 128   // agents per scene staged in LDS (cfg4: 64, stress: 128)
#define IL_REL 15     // relevant-agent list length per node
#define IL_RA 64      // doubles per node in T.relag: (1 + IL_REL) records x 4
#define IL_SCR 288    // doubles of per-wave LDS scratch (Riccati operands 0..190, constants 192..204, dump slots 208..287)
#define IL_CST 192    // scr[IL_CST..+6) = {1,0,0,0,0,0}, scr[IL_CST+6..+12) = {dt,0,0,0,0,0}
#define IL_LSUM 2048   // doubles of LDS used to stage the cost sums
#define IL_RMARGIN 3.0  // [m] the list is exact for queries within this distance of the nominal state

// Pointer into HBM whose accesses are emitted as GLOBAL memory instructions.  The tree / constant structs are read from
// memory, where the compiler cannot infer an address space: plain pointers would become generic (flat) accesses, which
// count against the LDS counter as well -- every wait for an LDS operand would then also wait for the global loads that
// were issued ahead on purpose (the one-node-ahead prefetches of the Riccati sweep and of the chain rollout).
#define IL_AS1 __attribute__((address_space(1)))
typedef double il_d2 __attribute__((ext_vector_type(2)));   // built-in vectors: loadable through address-space pointers
typedef float il_f2 __attribute__((ext_vector_type(2)));
template <class T> struct GP {
  T *p;
  GP() = default;
  __host__ __device__ GP(T *q) : p(q) {}
  template <class U> __host__ __device__ GP(const GP<U> &o) : p(o.p) {}
  __device__ __forceinline__ T IL_AS1 *g() const { return (T IL_AS1 *)p; }
  __device__ __forceinline__ T IL_AS1 &operator[](size_t i) const { return g()[i]; }
  __device__ __forceinline__ T IL_AS1 &operator*() const { return *g(); }
  template <class V> __device__ __forceinline__ V IL_AS1 *as() const { return (V IL_AS1 *)p; }
  __host__ __device__ __forceinline__ GP operator+(size_t o) const { return GP(p + o); }
  __host__ __device__ __forceinline__ GP &operator+=(size_t o) { p += o; return *this; }
  __host__ __device__ __forceinline__ explicit operator bool() const { return p != nullptr; }
};
// Control block of one cost tree whose Levenberg-Marquardt slots are spread over workgroups (k_ilqr<GEN, 2>): workgroup 0 (the master) runs
// iLQR.fit as a single workgroup does and evaluates slot 0 of every pass itself; workgroup s > 0 (a follower) waits for the master's command and
// evaluates slot s -- backward sweep at the mu the LM schedule reaches after s rejections, line search, candidate costs -- from the same nominal
// trajectory.  One exchange per pass: the master publishes {mu, delta, slot count, phase} (release, then gen + 1), a follower publishes its ten
// candidate costs and its singular flag (release, then done[s] = gen).  The master looks at a follower's result only when every earlier slot
// was rejected, so a pass whose slot 0 is accepted never waits.  Zeroed by the host with the upload.
struct IlSlotCtl {
  unsigned gen, cmd, alive, pad;         // cmd = slots | phase << 8 | exit << 16; alive = bit s set once workgroup s runs
  double mu, delta;
  unsigned done[IL_SLOTS];
  unsigned sing[IL_SLOTS];
  double Jnew[IL_SLOTS][IL_NA];
  // the derivative speculator (il_speculate): sreq = generation of the pass whose first candidate it shall differentiate, sdone = the one it has
  unsigned sreq, sdone, spad0, spad1;
};

struct IlqrTreeDev {
  int M, n_agents, n_levels, pad;
  GP<const int> parent;        // [M]
  GP<const int> level_start;   // [n_levels+1]
  GP<const int> level_nodes;   // [M] nodes sorted by depth (ties: key)
  GP<const int> child_start;   // [M+1]
  GP<const int> child_list;    // [M-1] children in key order
  // chain segments: maximal single-child paths; one wave walks a segment without workgroup barriers
  int n_segs, n_slevels, max_level_segs, pad2;
  GP<const int> seg_start;     // [n_segs+1] into seg_nodes (root -> leaf order inside a segment)
  GP<const int> seg_nodes;     // [M]
  GP<const int> slevel_start;  // [n_slevels+1] into slevel_segs (segments grouped by depth in the segment tree)
  GP<const int> slevel_segs;   // [n_segs]
  GP<const int> seg_rec;       // [n_segs][16]: s0, s1, last node, first node, node before the last, the last node's child count, its child_list start, -, six children
  // forward steps of the line search: a step = the segments of one segment level cut into chunks of at most `ilqr_chunk` nodes
  // (host), so that the cost pass of step s - 1 runs on the idle waves while the state chains of step s advance
  int n_fsteps, padf;
  GP<const int> fstep_start;   // [n_fsteps+1] into fstep_q0 / fstep_q1
  GP<const int> fstep_q0;      // per item a record of 8 ints: first position in seg_nodes, one past the last, the first two nodes, the first node's parent
  GP<const int> fstep_q1;      // per item: one past the last position
  GP<const int> fstep_nstart;  // [n_fsteps+1] into fstep_nodes
  GP<const int> fstep_nodes;   // [M] nodes in step order
  GP<const float> prob;        // [M]
  GP<const float> mean;        // [M,a,2]
  GP<const float> cov;         // [M,a]
  // workspace (doubles)
  GP<double> xs, us, Fx, L, Lx, Lxx;                        // [M,*]
  GP<double> k, K, Vx, Vxx;                                   // [IL_SPEC][M,*]  (one set per speculative mu)
  GP<double> xs_new, us_new, L_new;                            // [IL_SPEC][NA,M,*]
  GP<int> rel;                 // [M] number of exo agents near the nominal state (<= IL_REL), or -1 = overflow
  GP<double> relag;            // [M, IL_RA] compact staged records {mean_x, mean_y, sigma+offset, threshold}: entry 0 =
                            //   ego, entries 1..rel[c] = the relevant exo agents in ascending agent order
  // generic mode (planners/ilqr surface: arbitrary materialised fields + per-node diagonal weights)
  GP<const double> field;      // [M, H*W] cost_field of every node's PotentialField, or null
  GP<const double> node_w;     // [M, IL_NW]: w_des[6] w_con[6] lb[6] ub[6] w_ctrl[2] des[6], or null
  // outputs
  // second set of what the derivative pass writes (k_ilqr<GEN, 2> with a derivative speculator; dset == 0 otherwise): the host lays both
  // sets out alike, `dset` doubles (relag, Fx, L, Lx, Lxx) and `drel` ints (rel) apart -- two words instead of six more pointers in the
  // kernel's scalar registers.  Inside the master Fx .. rel are always the set of the nominal trajectory (il_flip_sets moves them and negates
  // the distances); followers and the speculator hold the launch's pointers and take the master's word (cmd bit 24) for which is which
  long long dset, drel;
  // small launches: where the host reads the results (page-locked, mapped staging) -- written by the kernel at its end; null = copies follow
  GP<double> h_xs, h_us, h_stats;
  // ... and a word the kernel sets to h_gen (system scope, behind those results) when this tree's are complete: the host may price a candidate
  // tree while the launch still works on the others (null: wide trees, whose launch can abort)
  unsigned *h_done;
  unsigned h_gen, pad_h;
  GP<double> stats;            // [IL_NSTAT]: iterations, converged, J, mu, phase cycles, profile slots
  // per-iteration trace of the fit (mind_last_ilqr_trace): IL_TRACE_W doubles per reference iteration {mu the backward pass ran with,
  // J of the nominal trajectory, accepted alpha index (-1: step rejected, -2: singular Q_uu), J of the accepted candidate}
  GP<double> trace;            // [phases 2][trace_cap][IL_TRACE_W], or null
  int trace_cap, padt;
  IlSlotCtl *ctl;              // slots spread over workgroups (k_ilqr<GEN, 2>), or null
};

struct IlqrConst {
  double dt, wb;
  double w_des[6], w_con[6], lb[6], ub[6], w_ctrl[2];
  double w_tgt, w_ego, w_ego_off, w_exo, w_exo_off, w_exo_cost;
  double res, off_x, off_y, target_vel;
  double x0[6];
  int W, H, max_iter, use_exo;
  double alphas[IL_NA];     // 1.1 ** (-j*j), j = 0..9 (solver.py:125), computed on the host
  GP<const double> gx, gy;    // [W], [H] grid coordinates (numpy linspace + offset, built on the host)
  // lin != 0: gx[i] == ((i == W-1 ? fsx : i * stepx) + off_x) bit for bit (checked on the host), so the kernels
  // compute cell centres instead of loading them
  int lin, pad_;
  double stepx, stepy, fsx, fsy;
  double in_x0, in_x1, in_y0, in_y1;   // conservative "window fully inside the grid" box for the list fast path
  GP<const double> quad;       // [H*W] squared distance to the target lane
};

// Ordering point for lane-to-lane exchange through LDS inside ONE wave.  The LDS pipeline executes a wave's DS
// instructions in order, so a ds_read issued after a ds_write sees the written data without waiting for the
// write to retire; only the compiler must be kept from reordering the accesses (it inserts the s_waitcnt a
// register use needs by itself).  -DIL_HW_FENCE restores the explicit wait.
#ifdef IL_HW_FENCE
#define IL_WFENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define IL_WFENCE() asm volatile("" ::: "memory")
#endif
#define IL_TRACE_W 4
#define IL_NSTAT 25   // doubles per tree in T.stats: 4 results + 4 phase cycle counters + 16 profile slots + passes

// fine-grained cycle attribution (diagnostic build only: -DIL_PROFILE); slots: 0-1 chain-rollout node (stage,
// u+dynamics+store), 5 nodes; 2-4 cost-pass chunk (stage+loads, field, cost+store), 15 chunks; 6-9 Riccati
// node (products, Qxx, solve, value update), 10 nodes; 11-13 derivative block (setup, tasks, assemble), 14 blocks
#ifdef IL_PROFILE
#define IL_PROF_ARG , long long *prof
#define IL_PROF_PASS , prof
#define IL_PT0() long long tp_ = clock64()
#define IL_PT(i) do { const long long n_ = clock64(); prof[i] += n_ - tp_; tp_ = n_; } while (0)
#define IL_PCNT(i) prof[i] += 1
#else
#define IL_PROF_ARG
#define IL_PROF_PASS
#define IL_PT0() do {} while (0)
#define IL_PT(i) do {} while (0)
#define IL_PCNT(i) do {} while (0)
#endif

__device__ __forceinline__ double il_shfl_down(double v, int d) {
  return __shfl_down(v, d, 64);
}

__device__ __forceinline__ double il_gx(const IlqrConst &C, int i) {
  return C.lin ? (i == C.W - 1 ? C.fsx : (double)i * C.stepx) + C.off_x : C.gx[i];
}
__device__ __forceinline__ double il_gy(const IlqrConst &C, int i) {
  return C.lin ? (i == C.H - 1 ? C.fsy : (double)i * C.stepy) + C.off_y : C.gy[i];
}

// ---- lane distance field: d(c)^2 = min over segments (ilqr/utils.py:5-22, geometry.py:70-78) ----
__global__ void k_lane_field(const double *__restrict__ gx, const double *__restrict__ gy, int W, int H,
                             const double *__restrict__ lane, int P, double *__restrict__ quad, int squared = 1) {
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
  quad[idx] = squared ? d * d : d;
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

// doubles of the per-wave agent table (derivative pass): all agents x 4
__host__ __device__ static inline int il_ag_doubles(int n_agents) { return 4 * n_agents; }

// Quadratic-potential parameters of one trajectory node (potential.py:4-59).  GEN = false: the planner's
// uniform structure, config weight x node probability (trajectory_tree.py:38-46, product in float64 as
// numpy computes it); GEN = true: arbitrary diagonal weights per node, read from T.node_w.
#define IL_NW 32
template <bool GEN> struct IlNodeW {
  const IlqrConst &C;
  const double *w;
  double p;
  __device__ __forceinline__ double wdes(int k) const { return GEN ? w[k] : C.w_des[k] * p; }
  __device__ __forceinline__ double wcon(int k) const { return GEN ? w[6 + k] : C.w_con[k] * p; }
  __device__ __forceinline__ double lb(int k) const { return GEN ? w[12 + k] : C.lb[k]; }
  __device__ __forceinline__ double ub(int k) const { return GEN ? w[18 + k] : C.ub[k]; }
  __device__ __forceinline__ double wctrl(int k) const { return GEN ? w[24 + k] : C.w_ctrl[k] * p; }
  __device__ __forceinline__ double des(int k) const { return GEN ? w[26 + k] : (k == 2 ? C.target_vel : 0.0); }
};

// Stage node i's agents into the wave's LDS table ag[e] = {mean_x, mean_y, sigma_e + offset (f32 add,
// as the reference computes it), early-out threshold}: entry 0 = ego (offset w_ego_cov_offset).
__device__ __forceinline__ void il_stage_agents(const IlqrConst &C, const IlqrTreeDev &T, int i, double *ag) {
  const int lane = threadIdx.x & 63;
  auto mean = T.mean + (size_t)i * T.n_agents * 2;
  auto cov = T.cov + (size_t)i * T.n_agents;
  IL_WFENCE();
  for (int e = lane; e < T.n_agents; e += 64) {
    const double ec = (double)(cov[e] + (float)(e == 0 ? C.w_ego_off : C.w_exo_off));
    ag[4 * e + 0] = (double)mean[2 * e];
    ag[4 * e + 1] = (double)mean[2 * e + 1];
    ag[4 * e + 2] = ec;
    ag[4 * e + 3] = ec * ec * 1.000000001;
  }
  IL_WFENCE();
}

// Cooperative (one wave) evaluation of node `i`'s potential field at (px, py).
// cells[9] (LDS, per wave) receives the raw window; every lane returns the same FieldOut.
template <bool GEN>
__device__ __forceinline__ void il_field(const IlqrConst &C, const IlqrTreeDev &T, int i, double px, double py,
                                         double *scr /* >= 64+9 doubles, per wave */, const double *ag, bool want_deriv, FieldOut &o) {
  const int lane = threadIdx.x & 63;
  long xi = (long)rint((px - C.off_x) / C.res);
  long yi = (long)rint((py - C.off_y) / C.res);
  xi = xi < 0 ? 0 : (xi > C.W - 1 ? C.W - 1 : xi);
  yi = yi < 0 ? 0 : (yi > C.H - 1 ? C.H - 1 : yi);
  const float pf = GEN ? 0.f : T.prob[i];
  const double wp = (double)((float)C.w_tgt * pf);
  // lanes 0..62: cell = lane % 9, agent slot = lane / 9 (7 slots)
  const int cell = lane % 9, slot = lane / 9;
  int sy, sx;
  il_window_src((int)xi, (int)yi, C.W, C.H, cell / 3, cell % 3, sy, sx);
  double part = 0.0;
  if (lane < 63 && sy >= 0 && C.use_exo) {
    const double cx = C.gx[sx], cy = C.gy[sy];
    for (int e = 1 + slot; e < T.n_agents; e += 7) {
      const double ec = ag[4 * e + 2];
      const double dx = cx - ag[4 * e], dy = cy - ag[4 * e + 1];
      const double d2 = dx * dx + dy * dy;
      if (d2 > ag[4 * e + 3]) continue;
      double v = ec - sqrt(d2);
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
    if (sy >= 0 && GEN) {
      cell_v = T.field[((size_t)i * C.H + sy) * C.W + sx];
    } else if (sy >= 0) {
      double covf = 0.0;
      for (int s = 0; s < 7; ++s) covf += scr[lane + 9 * s];
      const double q = C.quad[(size_t)sy * C.W + sx];
      if (C.use_exo) {
        const double ego_cov = ag[2];
        const double dx = C.gx[sx] - ag[0], dy = C.gy[sy] - ag[1];
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
  double s3, c3;
  mind_sincos(x[3], &s3, &c3);
  o[0] = x[0] + x[2] * c3 * C.dt;
  o[1] = x[1] + x[2] * s3 * C.dt;
  o[2] = x[2] + x[4] * C.dt;
  o[3] = x[3] + x[2] / C.wb * mind_tan(x[5]) * C.dt;
  o[4] = x[4] + u[0] * C.dt;
  o[5] = x[5] + u[1] * C.dt;
}

// quadratic potentials (potential.py:4-59) + field value
template <bool GEN>
__device__ __forceinline__ double il_node_cost(const IlNodeW<GEN> &NW, const double *x, const double *u, const FieldOut &fe) {
  double cost = 0.0;
  cost += fe.val;
  double sp = 0.0, sc = 0.0, cp = 0.0;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const double w = NW.wdes(k), d = x[k] - NW.des(k);
    sp += d * w * d;
  }
  cost += sp;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const double w = NW.wcon(k);
    const double d = fmax(x[k] - NW.ub(k), 0.0) + fmax(NW.lb(k) - x[k], 0.0);
    sc += d * w * d;
  }
  cost += sc;
#pragma unroll
  for (int k = 0; k < 2; ++k) cp += u[k] * NW.wctrl(k) * u[k];
  cost += cp;
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

// The same value for a tree too big for one thread to walk: the recursion above splits [0, n) into leaves of <= 128 elements (in order) and adds
// their sums pairwise -- every leaf sum is independent, so the workgroup's threads take a leaf each (the leaf's eight strided accumulators
// and its tail exactly as above) and one thread then folds the partial sums along the same recursion.  tab: >= 2 x leaves doubles of LDS.
__device__ inline int il_np_leaves(long lo, long n, double *tab, int idx) {          // leaf table: tab[2 i] = offset, tab[2 i + 1] = length
  if (n <= 128) { tab[2 * idx] = (double)lo; tab[2 * idx + 1] = (double)n; return idx + 1; }
  long n2 = n / 2;
  n2 -= n2 % 8;
  idx = il_np_leaves(lo, n2, tab, idx);
  return il_np_leaves(lo + n2, n - n2, tab, idx);
}
__device__ inline double il_np_fold(long n, const double *part, int &idx) {
  if (n <= 128) return part[idx++];
  long n2 = n / 2;
  n2 -= n2 % 8;
  const double l = il_np_fold(n2, part, idx);
  return l + il_np_fold(n - n2, part, idx);
}
// all threads of the workgroup; the result is valid in thread 0.  n <= 128 * (IL_LSUM / 3) elements (the host keeps M below that).
__device__ __forceinline__ double il_np_sum_wg(GP<const double> a, long n, double *tab) {
  __shared__ int sh_leaves;
  if (threadIdx.x == 0) sh_leaves = il_np_leaves(0, n, tab, 0);
  __syncthreads();
  const int nl = sh_leaves;
  double *part = tab + 2 * (size_t)nl;
  for (int i = threadIdx.x; i < nl; i += IL_THREADS) {
    const long lo = (long)tab[2 * i], len = (long)tab[2 * i + 1];
    const auto q = a + (size_t)lo;
    double res;
    if (len < 8) {
      res = 0.0;
      for (long k = 0; k < len; ++k) res += q[k];
    } else {
      double r[8];
      long k;
      for (k = 0; k < 8; ++k) r[k] = q[k];
      for (k = 8; k < len - (len % 8); k += 8)
        for (int j = 0; j < 8; ++j) r[j] += q[k + j];
      res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
      for (; k < len; ++k) res += q[k];
    }
    part[i] = res;
  }
  __syncthreads();
  double out = 0.0;
  if (threadIdx.x == 0) { int idx = 0; out = il_np_fold(n, part, idx); }
  __syncthreads();
  return out;
}

// J(candidate) = python sum() over the nodes in node order, for `rows` candidates whose node costs lie in rows of M doubles at Lg: strictly
// sequential per candidate, so thread r < rows walks row r -- but through LDS tiles the whole workgroup loads (coalesced, the next tile
// requested while this one is added) instead of one dependent global round trip per eight nodes.  The result is valid in threads < rows.
__device__ __forceinline__ double il_seq_sums_tiled(GP<const double> Lg, int M, int rows, double *lsum) {
  const int tid = threadIdx.x;
  const int tile = IL_LSUM / rows;                      // nodes per tile
  const int per = (rows * tile + IL_THREADS - 1) / IL_THREADS;      // elements per thread and tile (<= 4: IL_LSUM / IL_THREADS)
  double pre[IL_LSUM / IL_THREADS];
  auto request = [&](int n0) {
    const int nn = M - n0 < tile ? M - n0 : tile;
#pragma unroll
    for (int e = 0; e < IL_LSUM / IL_THREADS; ++e) {
      const int q = tid + e * IL_THREADS;
      const int r = q / nn, c = q - r * nn;
      pre[e] = (e < per && q < rows * nn) ? Lg[(size_t)r * M + n0 + c] : 0.0;
    }
  };
  double J = 0.0;
  request(0);
  for (int n0 = 0; n0 < M; n0 += tile) {
    const int nn = M - n0 < tile ? M - n0 : tile;
#pragma unroll
    for (int e = 0; e < IL_LSUM / IL_THREADS; ++e) {
      const int q = tid + e * IL_THREADS;
      const int r = q / nn, c = q - r * nn;
      if (e < per && q < rows * nn) lsum[r * tile + c] = pre[e];
    }
    __syncthreads();
    if (n0 + tile < M) request(n0 + tile);
    if (tid < rows) {
      const double *Ln = lsum + (size_t)tid * tile;
      int c = 0;
      for (; c + 8 <= nn; c += 8) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = Ln[c + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) J += v[k];
      }
      for (; c < nn; ++c) J += Ln[c];
    }
    __syncthreads();
  }
  return J;
}

// ---- Riccati sweep of one chain segment, one wave (solver.py:344-421) -----------------------------------------------------
// Per-wave LDS scratch (doubles): fx 0 | Vxx 36 | Tm 72 | Vr 108 (dt (V_xx[4+a][r] + mu [r == 4+a]), a < 2) | Vxr 120 (dt V_x[4+a]) |
// U 124 (2 x 7: Q_ux rows with Q_u in column 6) | KK 138 (2 x 7: K rows with k in column 6) | Quu 152 | Vx 176 | constants IL_CST |
// dump slots IL_DMP.. (lanes without a result write there: the node body has no branches).
#define IL_U 124
#define IL_KK 138
#define IL_QUU 152
#define IL_DMP 208
// The node step is ONE straight-line instruction stream for all 64 lanes: roles differ only in the LDS addresses and constants of a
// per-lane plan that is built once per segment (role-dependent CODE is serialised by the SIMT hardware, and the branches around
// it keep the compiler from overlapping the LDS reads, the 2x2 solves and the products).  Arithmetic per entry is the oracle's:
//   stage 1, every lane  s = sum_r A_r B_r  (six terms, r ascending), result = addend + s:
//     T = F_x^T V_xx (36 lanes), Q_x = l_x + F_x^T V_x (6), Q_ux = f_u^T (V_xx + mu I) F_x (12; f_u^T picks rows 4, 5 scaled by dt:
//     the scaled, regularised rows Vr are written by the lanes that produce V_xx), Q_uu = l_uu + f_u^T (V_xx + mu I) f_u (4),
//     Q_u = l_u + f_u^T V_x (2); the terms the oracle multiplies by an exact 0 / 1 constant are kept as such (B = {dt,0,..} / {1,0,..});
//   stage 2  Q_xx = l_xx + T F_x; 2x2 solves with partial pivoting (LAPACK dgesv order), every lane owns one right-hand side;
//   value update V = Q + K^T Q_uu K + K^T Q_u. + Q_.u K (V_xx lanes and V_x lanes in one stream), symmetrisation, 0 + V.
// On entry scr holds the children-summed value function (V_xx at 36, V_x at 176); on exit the value function of the segment's
// head node.  Per-node operands (F_x, l_xx, l_x, controls, control weights) are requested one node ahead, the node index two
// ahead.  Returns (wave-uniform) 1 if a Q_uu is singular.
template <bool GEN>
__device__ __forceinline__ int il_backward_segment(const IlqrConst &C, const IlqrTreeDev &T, const IlqrTreeDev &Ts, int s0, int s1,
                                                   int c_last, int c_prev, double mu, double *scr IL_PROF_ARG, const unsigned *watch = nullptr,
                                                   unsigned watch_gen = 0u) {
  const int lane = threadIdx.x & 63;
  const int i = lane / 6, j = lane % 6;
  const bool mat = lane < 36, vxl = lane >= 36 && lane < 42;
  const double dt = C.dt;
  // ---- per-lane plan
  int a_off, a_str, b_off, b_str, dst1, kind = 0, asel = 0;     // kind: 1 l_x, 2 l_uu diagonal, 3 l_u
  if (mat) { a_off = i; a_str = 6; b_off = 36 + j; b_str = 6; dst1 = 72 + lane; }
  else if (vxl) { a_off = lane - 36; a_str = 6; b_off = 176; b_str = 1; dst1 = IL_DMP + lane; kind = 1; }
  else if (lane < 54) { const int a = (lane - 42) / 6, jj = (lane - 42) % 6; a_off = 108 + a * 6; a_str = 1; b_off = jj; b_str = 6; dst1 = IL_U + a * 7 + jj; }
  else if (lane < 58) { const int a = (lane - 54) / 2, b = (lane - 54) % 2; a_off = 108 + a * 6 + 4 + b; a_str = 0; b_off = IL_CST + 6; b_str = 1;
                        dst1 = IL_QUU + (lane - 54); kind = a == b ? 2 : 0; asel = a; }
  else if (lane < 60) { const int a = lane - 58; a_off = 120 + a; a_str = 0; b_off = IL_CST; b_str = 1; dst1 = IL_U + a * 7 + 6; kind = 3; asel = a; }
  else { a_off = 0; a_str = 0; b_off = IL_CST + 1; b_str = 0; dst1 = IL_DMP + lane; }
  int t_off = 72 + (mat ? i * 6 : 0), f_off = mat ? j : 0;                 // stage 2
  int rho = lane < 7 ? lane : 6, rho_w = lane < 7 ? IL_KK + lane : IL_DMP + lane;   // solve: right-hand side / result column
  int jx = mat ? j : 6, ix = mat ? i : (vxl ? lane - 36 : 0);              // value update: columns of K / Q_ux (6 = k / Q_u)
  int tm_w = mat ? 72 + lane : IL_DMP + lane, tm_r = mat ? 72 + j * 6 + i : IL_DMP + lane;
  int v_dst = mat ? 36 + lane : (vxl ? 176 + (lane - 36) : IL_DMP + lane);
  int vr_dst = (mat && i >= 4) ? 108 + (i - 4) * 6 + j : ((vxl && lane >= 40) ? 120 + (lane - 40) : IL_DMP + lane);
  int fx_dst = mat ? lane : IL_DMP + lane;
  double madd = (mat && i >= 4 && i == j) ? mu : 0.0;
  // per-lane global pointers (VGPRs: the tree struct's base pointers would be re-read from spilled SGPRs in every node):
  // this lane's entry of F_x / l_xx / l_x, the controls, the node weights, where this lane's gains go
  const int lfx = mat ? lane : 35, llx = vxl ? lane - 36 : 0;
  const double IL_AS1 *pFx = T.Fx.g() + lfx, *pLxx = T.Lxx.g() + lfx, *pLx = T.Lx.g() + llx, *pUs = T.us.g();
  const float IL_AS1 *pProb = T.prob.g();
  const double IL_AS1 *pNw = GEN ? T.node_w.g() + 24 : nullptr;
  const int IL_AS1 *pSeg = T.seg_nodes.g();
  double IL_AS1 *pGa = lane < 6 ? Ts.K.g() + lane : Ts.k.g();        // K column `lane` (rows 6 apart) | k
  int ga_str = lane < 6 ? 12 : 2, ga_row = lane < 6 ? 6 : 1;
  // the plan is opaque to the optimiser: it must stay in registers over the node loop instead of being rebuilt per node
  asm volatile("" : "+v"(a_off), "+v"(a_str), "+v"(b_off), "+v"(b_str), "+v"(dst1), "+v"(t_off), "+v"(f_off), "+v"(rho), "+v"(rho_w));
  long long mk1 = kind == 1 ? -1ll : 0ll, mk2 = kind == 2 ? -1ll : 0ll, mk3 = kind == 3 ? -1ll : 0ll;     // addend selection masks
  asm volatile("" : "+v"(jx), "+v"(ix), "+v"(tm_w), "+v"(tm_r), "+v"(v_dst), "+v"(vr_dst), "+v"(fx_dst), "+v"(madd), "+v"(asel));
  asm volatile("" : "+v"(mk1), "+v"(mk2), "+v"(mk3));
  asm volatile("" : "+v"(pFx), "+v"(pLxx), "+v"(pLx), "+v"(pUs), "+v"(pProb), "+v"(pSeg), "+v"(pGa), "+v"(ga_str), "+v"(ga_row));
  // ---- regularised, scaled rows of the entry value function
  {
    const double vin = scr[v_dst];
    IL_WFENCE();
    scr[vr_dst] = dt * (vin + madd);
    IL_WFENCE();
  }
  struct Pre { double fx, lxx, lx, u0, u1, w0, w1; };
  auto request = [&](int c, Pre &P) {
    P.fx = pFx[(size_t)c * 36];
    P.lxx = pLxx[(size_t)c * 36];
    P.lx = pLx[(size_t)c * 6];
    const il_d2 u = *(const il_d2 IL_AS1 *)(pUs + (size_t)c * 2);
    P.u0 = u.x; P.u1 = u.y;
    if (GEN) { P.w0 = pNw[(size_t)c * IL_NW]; P.w1 = pNw[(size_t)c * IL_NW + 1]; }
    else { const double pp = (double)pProb[c]; P.w0 = C.w_ctrl[0] * pp; P.w1 = C.w_ctrl[1] * pp; }
  };
  int r = s1 - 1;
  int c = c_last;                               // (the segment's last node and the one before it come with the segment record)
  int cn_v = c_prev;                            // node indices travel two nodes ahead, in a VGPR until they are needed
  Pre P, N;
  request(c, P);
  // gains of the node before (loads and stores share one in-order counter: a store issued at the end of a node would be waited for at
  // the top of the next one, so it is issued behind the next node's loads instead and has a whole node to retire)
  double g0 = 0.0, g1 = 0.0;
  int gc = c;
  // (a follower: the master's command word, read one node ahead -- when it has moved on, this slot's result is of no use to anybody: 2)
  unsigned wnow = watch_gen;
  for (; r >= s0; --r) {
    IL_PT0();
    if (watch) {
      if ((unsigned)__builtin_amdgcn_readfirstlane((int)wnow) != watch_gen) return 2;
      wnow = __hip_atomic_load(watch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int cn = __builtin_amdgcn_readfirstlane(cn_v);
    cn_v = pSeg[r - 2 >= s0 ? r - 2 : s0];
    request(cn, N);
    if (lane < 7 && r != s1 - 1) {      // gains to global memory: K columns (lanes 0..5), k (lane 6)
      double IL_AS1 *g = pGa + (size_t)gc * ga_str;
      g[0] = g0; g[ga_row] = g1;
    }
    // ---- stage 1
    scr[fx_dst] = P.fx;
    const double wsel = asel ? P.w1 : P.w0, usel = asel ? P.u1 : P.u0;
    double ad2 = 2.0 * wsel, ad3 = 2.0 * (wsel * usel);
    // every lane computes both, then picks by bit masks: no branches in the body
    const double addend = __longlong_as_double((__double_as_longlong(P.lx) & mk1) | (__double_as_longlong(ad2) & mk2) | (__double_as_longlong(ad3) & mk3));
    IL_WFENCE();
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 6; ++q) s += scr[a_off + q * a_str] * scr[b_off + q * b_str];
    const double r1 = addend + s;
    IL_WFENCE();                      // all operands read before the results overwrite Tm / U / Quu
    scr[dst1] = r1;
    IL_WFENCE();
    IL_PT(6);
    // ---- stage 2: Q_xx (V_xx lanes; the others compute a value nobody reads)
    double qb = 0.0;
#pragma unroll
    for (int q = 0; q < 6; ++q) qb += scr[t_off + q] * scr[f_off + q * 6];
    qb = P.lxx + qb;
    asm volatile("" : "+v"(qb));
    qb = mat ? qb : r1;                // V_x lanes: Q_x
    IL_PT(7);
    // ---- 2x2 solves: every lane factorises Q_uu, lanes 0..6 own the right-hand sides (Q_ux columns, Q_u)
    const double q00 = scr[IL_QUU], q01 = scr[IL_QUU + 1], q10 = scr[IL_QUU + 2], q11 = scr[IL_QUU + 3];
    double a00 = q00, a01 = q01, a10 = q10, a11 = q11;
    double b0 = scr[IL_U + rho], b1 = scr[IL_U + 7 + rho];
    const bool swp = fabs(a10) > fabs(a00);
    if (swp) { double t = a00; a00 = a10; a10 = t; t = a01; a01 = a11; a11 = t; t = b0; b0 = b1; b1 = t; }
    const double l = a10 / a00;
    const double u11 = a11 - l * a01;
    const int singular = __builtin_amdgcn_readfirstlane((int)((a00 == 0.0) || (u11 == 0.0)));
    b1 = b1 - l * b0;
    const double x1 = b1 / u11;
    const double x0 = (b0 - a01 * x1) / a00;
    if (singular) return 1;             // (the slot is unusable: nobody reads its gains)
    IL_WFENCE();
    scr[rho_w] = -x0; scr[rho_w + 7] = -x1;
    IL_WFENCE();
    IL_PT(8);
    // ---- value update
    const double Kj0 = scr[IL_KK + jx], Kj1 = scr[IL_KK + 7 + jx], Uj0 = scr[IL_U + jx], Uj1 = scr[IL_U + 7 + jx];
    const double Ki0 = scr[IL_KK + ix], Ki1 = scr[IL_KK + 7 + ix], Ui0 = scr[IL_U + ix], Ui1 = scr[IL_U + 7 + ix];
    const double QuuK0 = q00 * Kj0 + q01 * Kj1;
    const double QuuK1 = q10 * Kj0 + q11 * Kj1;
    double v = qb + (Ki0 * QuuK0 + Ki1 * QuuK1);
    v += (Ki0 * Uj0 + Ki1 * Uj1) + (Ui0 * Kj0 + Ui1 * Kj1);
    IL_WFENCE();
    scr[tm_w] = v;
    IL_WFENCE();
    const double vt = scr[tm_r];
    double vs = 0.5 * (v + vt);
    asm volatile("" : "+v"(vs));
    const double vnew = 0.0 + (mat ? vs : v);          // parent accumulates 0 + V[child] (solver.py:349-350)
    IL_WFENCE();
    scr[v_dst] = vnew;
    scr[vr_dst] = dt * (vnew + madd);
    IL_WFENCE();
    g0 = -x0; g1 = -x1; gc = c;
    IL_PT(9); IL_PCNT(10);
    P = N; c = cn;
  }
  if (lane < 7) {
    double IL_AS1 *g = pGa + (size_t)gc * ga_str;
    g[0] = g0; g[ga_row] = g1;
  }
  return 0;
}

// x' = f(x, u) (trajectory_tree.py:168-175)
__device__ __forceinline__ void il_dyn_sc(const IlqrConst &C, const double *x, const double *u, double *o) {
  double s3, c3;
  mind_sincos(x[3], &s3, &c3);
  o[0] = x[0] + x[2] * c3 * C.dt;
  o[1] = x[1] + x[2] * s3 * C.dt;
  o[2] = x[2] + x[4] * C.dt;
  o[3] = x[3] + x[2] / C.wb * mind_tan(x[5]) * C.dt;
  o[4] = x[4] + u[0] * C.dt;
  o[5] = x[5] + u[1] * C.dt;
}

// ---- line search (solver.py:202-240), two phases ---------------------------------------------------------
// Only the state recursion x_c = f(x_parent, u_c(x_parent)) is serial along a chain; the cost of a candidate
// node depends on nothing but its own (x, u).  Phase 1 rolls the states of all 10 step sizes down the chain
// segments (cheap per node), phase 2 evaluates the M x 10 x slots node costs with one LANE per (node, alpha).

// Per-node operands of the chain rollout (gains, nominal control and state), loaded one node ahead into
// registers: the addresses are wave-uniform, every lane holds the full set.
struct IlKv { double K[12], k[2], us[2], xs[6]; };

// Phase 1: states/controls of ALL 10 line-search candidates along the chain pieces of one forward step, IL_PACK pieces per wave:
// lane = (piece g = lane / 10, candidate a = lane % 10).  The recursion is lane-wise (a lane carries its own state), so every piece
// of a level -- and every speculative mu slot of it -- rides in the lanes of ONE instruction stream instead of a wave each: two
// busy waves on one SIMD share its float64 pipe and run at 2.4 k cycles per node instead of 1.5 k (tools/micro/ric_bench.hip), and
// the demo trees have five branches per level on four SIMDs.  Item `w * IL_PACK + g` of the R = ni x slots items of the step:
// slot = item / ni, piece = lo + item % ni = positions [fstep_q0, fstep_q1) of T.seg_nodes; the first node of a piece continues
// from its parent's candidate state in T.xs_new.  Lanes without an item (and lanes whose piece is shorter than the wave's longest)
// shadow a valid node with frozen state and do not store.  init != 0: nominal rollout (alpha = 0, gains are zero).  Writes
// T.xs_new / T.us_new of the item's slot.  Base pointers live in VGPRs, the node index travels two nodes ahead, the node's operands
// one node ahead (loads and stores share one in-order counter: a node's results are stored at the top of the NEXT node, behind
// that node's operand loads).
#define IL_PACK 6
__device__ __forceinline__ int il_rollout_packed(const IlqrConst &C, const IlqrTreeDev &T, int lo, int ni, int R, int w, int init, int slot0 IL_PROF_ARG,
                                                 const unsigned *watch = nullptr, unsigned watch_gen = 0u) {
  const int lane = threadIdx.x & 63;
  const int M = T.M;
  const int a = lane % IL_NA, g = lane / IL_NA;
  const int item = w * IL_PACK + g;
  const bool valid = g < IL_PACK && item < R;
  const int ic = valid ? item : w * IL_PACK;
  const int sl_ = ic / ni, it = lo + (ic - sl_ * ni), slot = slot0 + sl_;      // (slot0: the slot a follower workgroup evaluates)
  const double alpha = init ? 0.0 : C.alphas[a];
  const int IL_AS1 *pSeg = T.seg_nodes.g();
  const double IL_AS1 *pK = T.K.g() + (size_t)slot * M * 12, *pk = T.k.g() + (size_t)slot * M * 2, *pUs = T.us.g(), *pXs = T.xs.g();
  double IL_AS1 *pXn = T.xs_new.g() + ((size_t)slot * IL_NA + a) * M * 6, *pUn = T.us_new.g() + ((size_t)slot * IL_NA + a) * M * 2;
  asm volatile("" : "+v"(pSeg), "+v"(pK), "+v"(pk), "+v"(pUs), "+v"(pXs), "+v"(pXn), "+v"(pUn));
  const int IL_AS1 *rec = T.fstep_q0.g() + (size_t)it * 8;       // {q0, q1, first node, second node, the first node's parent}
  int q = rec[0];
  const int s1 = rec[1];
  int c = rec[2];
  int cn = rec[3];
  const int p0 = rec[4];                    // -1: node 0, the only child of the x0 root (checked on the host)
  int nmax = s1 - q;                      // the wave runs as many nodes as its longest piece has
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(nmax, d); nmax = o > nmax ? o : nmax; }
  nmax = __builtin_amdgcn_readfirstlane(nmax);
  auto prefetch = [&](int c, IlKv &P) {
    const auto qK = (const il_d2 IL_AS1 *)(pK + (size_t)c * 12);
    const auto qx = (const il_d2 IL_AS1 *)(pXs + (size_t)c * 6);
#pragma unroll
    for (int e = 0; e < 6; ++e) { const il_d2 v = qK[e]; P.K[2 * e] = v.x; P.K[2 * e + 1] = v.y; }
#pragma unroll
    for (int e = 0; e < 3; ++e) { const il_d2 v = qx[e]; P.xs[2 * e] = v.x; P.xs[2 * e + 1] = v.y; }
    const il_d2 vk = *(const il_d2 IL_AS1 *)(pk + (size_t)c * 2);
    const il_d2 vu = *(const il_d2 IL_AS1 *)(pUs + (size_t)c * 2);
    P.k[0] = vk.x; P.k[1] = vk.y; P.us[0] = vu.x; P.us[1] = vu.y;
  };
  double xp[6], xo[6];
  {
    const size_t pp = (size_t)(p0 < 0 ? 0 : p0) * 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) { const double vn = pXn[pp + k], vs = pXs[pp + k]; xp[k] = p0 < 0 ? C.x0[k] : vn; xo[k] = p0 < 0 ? C.x0[k] : vs; }
  }
  IlKv PA, PB;
  prefetch(c, PA);
  double su[2] = {0.0, 0.0};
  int sc = c;
  bool sact = false;                      // the previous node of this lane was one of its piece (its state is to be stored)
  auto store = [&]() {
    if (sact) {
      const auto xn = (il_d2 IL_AS1 *)(pXn + (size_t)sc * 6);
      xn[0] = il_d2{xp[0], xp[1]}; xn[1] = il_d2{xp[2], xp[3]}; xn[2] = il_d2{xp[4], xp[5]};
      *(il_d2 IL_AS1 *)(pUn + (size_t)sc * 2) = il_d2{su[0], su[1]};
    }
  };
  // Two nodes per loop trip (the operand sets swap roles instead of being copied).
  auto node = [&](const IlKv &Pc, IlKv &Pn) {
    IL_PT0();
    const int cnn = pSeg[q + 2 < s1 ? q + 2 : s1 - 1];
    prefetch(cn, Pn);
    store();
    IL_PT(0);
    double u[2], x[6];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      double sm = 0.0;
#pragma unroll
      for (int jj = 0; jj < 6; ++jj) sm += Pc.K[b * 6 + jj] * (xp[jj] - xo[jj]);
      const double t = Pc.us[b] + alpha * Pc.k[b];
      u[b] = c == 0 ? t : t + sm;
    }
    il_dyn_sc(C, xp, u, x);
    const bool act = valid && q < s1;
    if (act) {
#pragma unroll
      for (int k = 0; k < 6; ++k) { xp[k] = x[k]; xo[k] = Pc.xs[k]; }
      su[0] = u[0]; su[1] = u[1];
      sc = c;
    }
    sact = act;
    c = cn; cn = cnn;
    ++q;
    IL_PT(1); IL_PCNT(5);
  };
  unsigned wnow = watch_gen;
  for (int n = 0; n < nmax; n += 2) {
    if (watch) {          // (a follower whose master has moved on leaves: see il_backward_segment)
      if ((unsigned)__builtin_amdgcn_readfirstlane((int)wnow) != watch_gen) return 1;
      wnow = __hip_atomic_load(watch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    node(PA, PB);
    if (n + 1 < nmax) node(PB, PA);
  }
  store();
  return 0;
}

// 3x3 window as separable axes: source column of window column c, source row of window row r, -1 = the
// cell stays zero (every case of il_window_src is a product of a row and a column condition).
__device__ __forceinline__ void il_window_axes(int xi, int yi, int W, int H, int *sxc, int *syr) {
#pragma unroll
  for (int k = 0; k < 3; ++k) { sxc[k] = -1; syr[k] = -1; }
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      int sy, sx;
      il_window_src(xi, yi, W, H, r, c, sy, sx);
      if (sy >= 0) { syr[r] = sy; sxc[c] = sx; }
    }
}

// one exo agent's contribution to the 9 window cells (trajectory_tree.py:94-101), accumulated in agent order
#define IL_EXO_TERM(AX, AY, EC, TH)                                                             \
  do {                                                                                          \
    double dx2_[3], dy2_[3];                                                                    \
    _Pragma("unroll") for (int k_ = 0; k_ < 3; ++k_) {                                          \
      const double dx_ = colx[k_] - (AX), dy_ = rowy[k_] - (AY);                                \
      dx2_[k_] = dx_ * dx_; dy2_[k_] = dy_ * dy_;                                               \
    }                                                                                           \
    /* the smallest of the 9 squared distances decides whether ANY cell can be reached */       \
    const double mx_ = fmin(dx2_[0], fmin(dx2_[1], dx2_[2])), my_ = fmin(dy2_[0], fmin(dy2_[1], dy2_[2])); \
    if (!__any(mx_ + my_ <= (TH))) break;                                                        \
    _Pragma("unroll") for (int r_ = 0; r_ < 3; ++r_)                                            \
      _Pragma("unroll") for (int c_ = 0; c_ < 3; ++c_) {                                        \
        const double d2_ = dx2_[c_] + dy2_[r_];                                                 \
        if (d2_ <= (TH)) {                      /* else max(ec - sqrt(d2), 0) == 0 exactly */   \
          double v_ = (EC) - sqrt(d2_);                                                         \
          v_ = v_ > 0.0 ? v_ : 0.0;                                                             \
          if (v_ > 0.0) v_ += C.w_exo_cost;                                                     \
          cov[r_ * 3 + c_] += v_;                                                               \
        }                                                                                       \
      }                                                                                         \
  } while (0)

// Phase 2: node costs of all candidates, one lane per (slot, node, alpha); a wave takes chunks of 6 nodes
// x 10 alphas and stages the 6 nodes' compact agent records (T.relag) in its LDS.  The exo sum runs over
// the records in ascending agent order (the oracle's summation order with provably-zero terms dropped).
// A candidate outside the validity region of its node's list (or a node whose list overflowed) walks all
// agents from global memory instead.
// `nodes` / `cnt`: the nodes to evaluate (a forward step's list); `wave` = this wave's position in the chunk deal (the caller rotates
// it so that the waves without a state chain this step come first).
template <bool GEN>
__device__ __forceinline__ void il_cost_pass(const IlqrConst &C, const IlqrTreeDev &T, int nuse, double *recs, int wave, int nwaves,
                                             GP<const int> nodes, int cnt, int slot0 IL_PROF_ARG, const unsigned *watch = nullptr, unsigned watch_gen = 0u) {
  const int lane = threadIdx.x & 63;
  const int M = T.M, P = nuse * cnt;
  const int nchunk = (P + 5) / 6;
  const int a = lane % IL_NA, pl = lane / IL_NA;
  unsigned wnow = watch_gen;
  for (int ch = wave; ch < nchunk; ch += nwaves) {
    IL_PT0();
    if (watch) {          // (a follower whose master has moved on skips the chunks that are left)
      if ((unsigned)__builtin_amdgcn_readfirstlane((int)wnow) != watch_gen) return;
      wnow = __hip_atomic_load(watch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int pi = ch * 6 + pl;
    const bool valid = lane < 60 && pi < P;
    const int pic = pi < P ? pi : P - 1;
    const int slot = slot0 + pic / cnt, c = nodes[pic % cnt];
    // issue every independent load first: the records of the chunk's 6 nodes, the candidate's state/control,
    // the node's probability / list length / nominal position
    double rr[6];
    if (C.use_exo) {
#pragma unroll
      for (int p = 0; p < 6; ++p) {
        const int pp = ch * 6 + p < P ? ch * 6 + p : P - 1;
        rr[p] = T.relag[(size_t)nodes[pp % cnt] * IL_RA + lane];
      }
    }
    const size_t eo = (size_t)(slot * IL_NA + a) * M + c;
    double x[6], u[2];
    {
      const auto px = (T.xs_new + eo * 6).as<const il_d2>();
      const il_d2 v0 = px[0], v1 = px[1], v2 = px[2], vu = *(T.us_new + eo * 2).as<const il_d2>();
      x[0] = v0.x; x[1] = v0.y; x[2] = v1.x; x[3] = v1.y; x[4] = v2.x; x[5] = v2.y; u[0] = vu.x; u[1] = vu.y;
    }
    const float pf = GEN ? 0.f : T.prob[c];
    int cnt = 0;
    bool full = false;
    double nx = 0.0, ny = 0.0;
    if (C.use_exo) {
      cnt = T.rel[c];
      const il_d2 vn = *(T.xs + (size_t)c * 6).as<const il_d2>();
      nx = vn.x; ny = vn.y;
      IL_WFENCE();
#pragma unroll
      for (int p = 0; p < 6; ++p) recs[p * IL_RA + lane] = rr[p];
      IL_WFENCE();
    }
    if (C.use_exo) {
      // the list was built around the nominal state; it is exact while the query stays within IL_RMARGIN of
      // it and its window inside the grid (no index clamping)
      const double ddx = x[0] - nx, ddy = x[1] - ny;
      const bool ok = x[0] > C.in_x0 && x[0] < C.in_x1 && x[1] > C.in_y0 && x[1] < C.in_y1 &&
                      ddx * ddx + ddy * ddy < IL_RMARGIN * IL_RMARGIN;
      full = cnt < 0 || !ok;
    }
    long xi = (long)rint((x[0] - C.off_x) / C.res);
    long yi = (long)rint((x[1] - C.off_y) / C.res);
    xi = xi < 0 ? 0 : (xi > C.W - 1 ? C.W - 1 : xi);
    yi = yi < 0 ? 0 : (yi > C.H - 1 ? C.H - 1 : yi);
    int sxc[3], syr[3];
    il_window_axes((int)xi, (int)yi, C.W, C.H, sxc, syr);
    double g[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) {
        const int sy = syr[r] < 0 ? 0 : syr[r], sx = sxc[cc] < 0 ? 0 : sxc[cc];
        g[r * 3 + cc] = GEN ? T.field[((size_t)c * C.H + sy) * C.W + sx] : C.quad[(size_t)sy * C.W + sx];
      }
    IL_PT(2);
    if (!GEN) {
      const double wp = (double)((float)C.w_tgt * pf);
      if (C.use_exo) {
        double colx[3], rowy[3], cov[9];
#pragma unroll
        for (int k = 0; k < 3; ++k) { colx[k] = il_gx(C, sxc[k] < 0 ? 0 : sxc[k]); rowy[k] = il_gy(C, syr[k] < 0 ? 0 : syr[k]); }
#pragma unroll
        for (int k = 0; k < 9; ++k) cov[k] = 0.0;
        const double *rec = recs + pl * IL_RA;
        double ex, ey, ego_cov;
        if (!full) {
          for (int e = 1; e <= cnt; ++e) {
            const double ax = rec[4 * e], ay = rec[4 * e + 1], ec = rec[4 * e + 2], th = rec[4 * e + 3];
            IL_EXO_TERM(ax, ay, ec, th);
          }
          ex = rec[0]; ey = rec[1]; ego_cov = rec[2];
        } else {
          auto mean = T.mean + (size_t)c * T.n_agents * 2;
          auto cv = T.cov + (size_t)c * T.n_agents;
          for (int e = 1; e < T.n_agents; ++e) {
            const double ax = (double)mean[2 * e], ay = (double)mean[2 * e + 1];
            const double ec = (double)(cv[e] + (float)C.w_exo_off), th = ec * ec * 1.000000001;
            IL_EXO_TERM(ax, ay, ec, th);
          }
          ex = (double)mean[0]; ey = (double)mean[1]; ego_cov = (double)(cv[0] + (float)C.w_ego_off);
        }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) {
            const double dx = colx[cc] - ex, dy = rowy[r] - ey;
            double ego = sqrt(dx * dx + dy * dy) - ego_cov;
            ego = ego > 0.0 ? ego : 0.0;
            g[r * 3 + cc] = (wp * g[r * 3 + cc] + C.w_exo * cov[r * 3 + cc]) + C.w_ego * ego;
          }
      } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) g[k] = wp * g[k];
      }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int cc = 0; cc < 3; ++cc)
        if (syr[r] < 0 || sxc[cc] < 0) g[r * 3 + cc] = 0.0;
    IL_PT(3);
    double s00, s02, s20, s22, s01, s10, s12, s21, s11;
    s00 = (((g[0] + g[1]) + g[3]) + g[4]) / 4.0;
    s02 = (((g[1] + g[2]) + g[4]) + g[5]) / 4.0;
    s20 = (((g[3] + g[4]) + g[6]) + g[7]) / 4.0;
    s22 = (((g[4] + g[5]) + g[7]) + g[8]) / 4.0;
    s01 = (g[1] + g[4]) / 2.0;
    s10 = (g[3] + g[4]) / 2.0;
    s12 = (g[4] + g[5]) / 2.0;
    s21 = (g[4] + g[7]) / 2.0;
    s11 = g[4];
    const double uu = (x[0] - il_gx(C, (int)xi)) / C.res + 0.5;
    const double vv = (x[1] - il_gy(C, (int)yi)) / C.res + 0.5;
    const double u1 = 1 - uu, v1 = 1 - vv;
    FieldOut fe;
    fe.val = u1 * u1 * v1 * v1 * s00 + u1 * u1 * 2.0 * v1 * vv * s10 + u1 * u1 * vv * vv * s20 +
             2.0 * u1 * uu * v1 * v1 * s01 + 2.0 * u1 * uu * 2.0 * v1 * vv * s11 + 2.0 * u1 * uu * vv * vv * s21 +
             uu * uu * v1 * v1 * s02 + uu * uu * 2.0 * v1 * vv * s12 + uu * uu * vv * vv * s22;
    if (valid) {
      const IlNodeW<GEN> NW{C, GEN ? (T.node_w + (size_t)c * IL_NW).p : nullptr, (double)pf};
      T.L_new[eo] = il_node_cost<GEN>(NW, x, u, fe);
    }
    IL_PT(4); IL_PCNT(15);
  }
}

// Cost value / gradient / Hessian and the dynamics Jacobian of node c at (x, u), one matrix entry per
// lane (cost.py:341-446 over the node's potentials; f_x at the POST state, Q1).  Every destination is
// optional.  One wave; `ag` must already hold the node's agents when C.use_exo.
template <bool GEN>
__device__ __forceinline__ void il_node_derivs(const IlqrConst &C, const IlqrTreeDev &T, int c, const double *x, const double *u,
                                               double *scr, const double *ag, double *dLxx, double *dFx, double *dLx, double *dL) {
  const int lane = threadIdx.x & 63;
  const IlNodeW<GEN> NW{C, GEN ? (T.node_w + (size_t)c * IL_NW).p : nullptr, GEN ? 0.0 : (double)T.prob[c]};
  FieldOut fe;
  il_field<GEN>(C, T, c, x[0], x[1], scr, ag, true, fe);
  double s3, c3;
  mind_sincos(x[3], &s3, &c3);
  double c5, t5;
  mind_tan_cos(x[5], &t5, &c5);
  if (lane < 36) {
    const int i = lane / 6, j = lane % 6;
    // l_xx = field Hessian (xy block) + 2 W_des + 2 W_con on violated bounds
    double h = 0.0;
    if (i == 0 && j == 0) h += fe.hxx;
    if ((i == 0 && j == 1) || (i == 1 && j == 0)) h += fe.hxy;
    if (i == 1 && j == 1) h += fe.hyy;
    if (i == j) {
      h += 2.0 * NW.wdes(i);
      if (x[i] > NW.ub(i) || x[i] < NW.lb(i)) h += 2.0 * NW.wcon(i);
    }
    if (dLxx) dLxx[lane] = h;
    double J = i == j ? 1.0 : 0.0;
    if (i == 0 && j == 2) J = c3 * C.dt;
    if (i == 0 && j == 3) J = -x[2] * s3 * C.dt;
    if (i == 1 && j == 2) J = s3 * C.dt;
    if (i == 1 && j == 3) J = x[2] * c3 * C.dt;
    if (i == 2 && j == 4) J = C.dt;
    if (i == 3 && j == 2) J = t5 / C.wb * C.dt;
    if (i == 3 && j == 5) J = x[2] / C.wb / (c5 * c5) * C.dt;
    if (dFx) dFx[lane] = J;
  } else if (lane < 42) {
    const int k = lane - 36;
    double g = 0.0;
    if (k == 0) g += fe.gx;
    if (k == 1) g += fe.gy;
    g += 2.0 * (NW.wdes(k) * (x[k] - NW.des(k)));
    const double w = NW.wcon(k);
    if (x[k] > NW.ub(k)) g += 2.0 * w * (x[k] - NW.ub(k));
    else if (x[k] < NW.lb(k)) g += 2.0 * w * (x[k] - NW.lb(k));
    if (dLx) dLx[k] = g;
  } else if (lane == 42) {
    if (dL) *dL = il_node_cost<GEN>(NW, x, u, fe);
  }
}

// Derivatives at the accepted iterate (solver.py:285-294,308-320; xs/us are already known: the accepted
// line-search candidate IS the next nominal rollout), one LANE per node, blocks of up to 64 nodes.  The block's
// agent arrays are staged agent-major in LDS; eleven independent tasks are spread over the waves -- the nine
// window cells (exo sum over ALL agents in ascending order, as the oracle), the relevant-agent records for
// the next line search, the dynamics Jacobian -- then wave 0 assembles value / gradient / Hessian per node.
// Only the entries of F_x / l_xx that ever change are written (the rest is set once at kernel start).
#define IL_DSTG 12288   // floats of LDS staging: 3 x agents x (nodes per block + 1)
#define IL_RECS (64 * IL_RA)   // doubles of LDS for compact records: 64 nodes (derivative pass) >= 8 waves x 6 nodes (cost pass)
template <bool GEN>
__device__ __forceinline__ void il_deriv_pass(const IlqrConst &C, const IlqrTreeDev &T, double *gcell, float *stg, double *drec,
                                              unsigned *dmask, int wg, int G, bool &staged IL_PROF_ARG) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int M = T.M, A = T.n_agents;
  const bool exo = !GEN && C.use_exo;
  int nb = 64;
  if (exo && 3 * A * (nb + 1) > IL_DSTG) nb = IL_DSTG / (3 * A) - 1;
  const int nbp = nb + 1;                       // padded row: conflict-free transposed writes
  float *smx = stg, *smy = stg + A * nbp, *scv = stg + 2 * A * nbp;
  for (int n0 = wg * nb; n0 < M; n0 += G * nb) {
    IL_PT0();
    const int nn = M - n0 < nb ? M - n0 : nb;
    // the agents' means / sigmas of this workgroup's block do not change during a fit and nobody else writes the staging region: a
    // workgroup with ONE block (every demo-size tree) stages them in its first derivative pass only
    const bool one_block = wg * nb + G * nb >= M;
    if (exo && !(staged && one_block)) {
      for (int q = tid; q < nn * A; q += IL_THREADS) {
        const int n = q / A, e = q - n * A;
        const il_f2 m = T.mean.as<const il_f2>()[(size_t)(n0 + n) * A + e];
        smx[e * nbp + n] = m.x; smy[e * nbp + n] = m.y; scv[e * nbp + n] = T.cov[(size_t)(n0 + n) * A + e];
      }
      __syncthreads();
      staged = one_block;
    }
    const bool valid = lane < nn;
    const int n = valid ? lane : nn - 1;
    const int c = n0 + n;
    double x[6], u[2];
    {
      const auto px = (T.xs + (size_t)c * 6).as<const il_d2>();
      const il_d2 v0 = px[0], v1 = px[1], v2 = px[2], vu = *(T.us + (size_t)c * 2).as<const il_d2>();
      x[0] = v0.x; x[1] = v0.y; x[2] = v1.x; x[3] = v1.y; x[4] = v2.x; x[5] = v2.y; u[0] = vu.x; u[1] = vu.y;
    }
    const float pf = GEN ? 0.f : T.prob[c];
    long xi = (long)rint((x[0] - C.off_x) / C.res);
    long yi = (long)rint((x[1] - C.off_y) / C.res);
    xi = xi < 0 ? 0 : (xi > C.W - 1 ? C.W - 1 : xi);
    yi = yi < 0 ? 0 : (yi > C.H - 1 ? C.H - 1 : yi);
    int sxc[3], syr[3];
    il_window_axes((int)xi, (int)yi, C.W, C.H, sxc, syr);
    IL_PT(11);
    // ---- compact records of the agents near the nominal state (for the next line search AND for this pass's
    //      window cells): |mu_e - x| < (sigma_e + offset) + margin + window reach, ascending agent order.
    //      The waves share the agents: hit bits meet in an LDS mask per node, a hit's slot is the number of
    //      lower set bits.
    int cnt = 0;
    bool full = false;
    if (exo) {
      if (wave < 4) dmask[n * 4 + wave] = 0u;
      __syncthreads();
      for (int e = 1 + wave; e < A; e += IL_WAVES) {
        const double ax = (double)smx[e * nbp + n], ay = (double)smy[e * nbp + n];
        const double ec = (double)(scv[e * nbp + n] + (float)C.w_exo_off);
        const double dx = ax - x[0], dy = ay - x[1];
        const double rr = ec + IL_RMARGIN + 1.0;      // 1.0 > 1.5 cells * sqrt(2) * 0.4 m
        if (valid && dx * dx + dy * dy < rr * rr) atomicOr(&dmask[n * 4 + (e >> 5)], 1u << (e & 31));
      }
      __syncthreads();
      const unsigned m0 = dmask[n * 4], m1 = dmask[n * 4 + 1], m2 = dmask[n * 4 + 2], m3 = dmask[n * 4 + 3];
      cnt = __popc(m0) + __popc(m1) + __popc(m2) + __popc(m3);
      if (cnt > IL_REL) cnt = -1;
      auto ra = T.relag + (size_t)c * IL_RA;
      double *rl = drec + (size_t)n * IL_RA;
      for (int e = 1 + wave; e < A; e += IL_WAVES) {
        const unsigned w_ = e >> 5, bit = 1u << (e & 31);
        const unsigned mw = w_ == 0 ? m0 : (w_ == 1 ? m1 : (w_ == 2 ? m2 : m3));
        if (cnt >= 0 && (mw & bit)) {
          const int slot = (w_ > 0 ? __popc(m0) : 0) + (w_ > 1 ? __popc(m1) : 0) + (w_ > 2 ? __popc(m2) : 0) + __popc(mw & (bit - 1u));
          const double ax = (double)smx[e * nbp + n], ay = (double)smy[e * nbp + n];
          const double ec = (double)(scv[e * nbp + n] + (float)C.w_exo_off);
          const double th = ec * ec * 1.000000001;
          double *dl = rl + 4 * (1 + slot);
          dl[0] = ax; dl[1] = ay; dl[2] = ec; dl[3] = th;
          if (valid) { const GP<double> dg = ra + 4 * (1 + slot); dg[0] = ax; dg[1] = ay; dg[2] = ec; dg[3] = th; }
        }
      }
      if (wave == 0) {
        const double ec0 = (double)(scv[n] + (float)C.w_ego_off);
        rl[0] = (double)smx[n]; rl[1] = (double)smy[n]; rl[2] = ec0; rl[3] = ec0 * ec0 * 1.000000001;
        if (valid) {
          ra[0] = rl[0]; ra[1] = rl[1]; ra[2] = rl[2]; ra[3] = rl[3];
          T.rel[c] = cnt;
        }
      }
      // the list covers the window only when the window sits around x (no index clamping)
      full = cnt < 0 || !(x[0] > C.in_x0 && x[0] < C.in_x1 && x[1] > C.in_y0 && x[1] < C.in_y1);
      __syncthreads();
    }
    for (int task = wave; task < 10; task += IL_WAVES) {
      if (task < 9) {
        // ---- one window cell of every node of the block (trajectory_tree.py:78-108)
        const int r = task / 3, cc = task - 3 * r;
        const int sy0 = r == 0 ? syr[0] : (r == 1 ? syr[1] : syr[2]);
        const int sx0 = cc == 0 ? sxc[0] : (cc == 1 ? sxc[1] : sxc[2]);
        const int sy = sy0 < 0 ? 0 : sy0, sx = sx0 < 0 ? 0 : sx0;
        double v;
        if (GEN) {
          v = T.field[((size_t)c * C.H + sy) * C.W + sx];
        } else {
          const double q = C.quad[(size_t)sy * C.W + sx];
          const double wp = (double)((float)C.w_tgt * pf);
          if (exo) {
            const double cx = il_gx(C, sx), cy = il_gy(C, sy);
            double covf = 0.0;
            // exo sum in ascending agent order (the oracle's order; provably-zero terms dropped): over the node's
            // records, or over all agents when the list overflowed / the window was clamped to the grid border
            const int ne = full ? A - 1 : cnt;
            const double *rl = drec + (size_t)n * IL_RA;
            for (int e0 = 1; e0 <= ne; e0 += 4) {
              double ec[4], d2[4];
              bool need = false;
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const int e = e0 + k <= ne ? e0 + k : (ne > 0 ? ne : 1);
                double ax, ay;
                if (full) {
                  ax = (double)smx[e * nbp + n]; ay = (double)smy[e * nbp + n];
                  ec[k] = (double)(scv[e * nbp + n] + (float)C.w_exo_off);
                } else {
                  ax = rl[4 * e]; ay = rl[4 * e + 1]; ec[k] = rl[4 * e + 2];
                }
                const double dx = cx - ax, dy = cy - ay;
                d2[k] = dx * dx + dy * dy;
                if (e0 + k > ne || d2[k] > ec[k] * ec[k] * 1.000000001) d2[k] = -1.0;   // max(ec - sqrt(d2), 0) == 0 exactly
                need |= d2[k] >= 0.0;
              }
              if (__any(need)) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                  double t = d2[k] >= 0.0 ? ec[k] - sqrt(d2[k]) : 0.0;
                  t = t > 0.0 ? t : 0.0;
                  if (t > 0.0) t += C.w_exo_cost;
                  covf += t;
                }
              }
            }
            const double ego_cov = (double)(scv[n] + (float)C.w_ego_off);
            const double dx = cx - (double)smx[n], dy = cy - (double)smy[n];
            double ego = sqrt(dx * dx + dy * dy) - ego_cov;
            ego = ego > 0.0 ? ego : 0.0;
            v = (wp * q + C.w_exo * covf) + C.w_ego * ego;
          } else {
            v = wp * q;
          }
        }
        gcell[n * 9 + task] = (sy0 >= 0 && sx0 >= 0) ? v : 0.0;
      } else {
        // ---- f_x at the POST state (Q1); constant entries were written at kernel start
        double s3, c3;
        mind_sincos(x[3], &s3, &c3);
        double c5, t5;
        mind_tan_cos(x[5], &t5, &c5);
        if (valid) {
          auto F = T.Fx + (size_t)c * 36;
          F[2] = c3 * C.dt;
          F[3] = -x[2] * s3 * C.dt;
          F[8] = s3 * C.dt;
          F[9] = x[2] * c3 * C.dt;
          F[20] = t5 / C.wb * C.dt;
          F[23] = x[2] / C.wb / (c5 * c5) * C.dt;
        }
      }
    }
    __syncthreads();
    IL_PT(12);
    if (wave == 0) {
      // ---- PotentialField value / gradient / Hessian (potential.py:146-260) and the node's cost derivatives
      double g[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) g[k] = gcell[n * 9 + k];
      const double s00 = (((g[0] + g[1]) + g[3]) + g[4]) / 4.0;
      const double s02 = (((g[1] + g[2]) + g[4]) + g[5]) / 4.0;
      const double s20 = (((g[3] + g[4]) + g[6]) + g[7]) / 4.0;
      const double s22 = (((g[4] + g[5]) + g[7]) + g[8]) / 4.0;
      const double s01 = (g[1] + g[4]) / 2.0;
      const double s10 = (g[3] + g[4]) / 2.0;
      const double s12 = (g[4] + g[5]) / 2.0;
      const double s21 = (g[4] + g[7]) / 2.0;
      const double s11 = g[4];
      const double res = C.res;
      const double uu = (x[0] - il_gx(C, (int)xi)) / res + 0.5;
      const double vv = (x[1] - il_gy(C, (int)yi)) / res + 0.5;
      const double u1 = 1 - uu, v1 = 1 - vv;
      FieldOut fe;
      fe.val = u1 * u1 * v1 * v1 * s00 + u1 * u1 * 2.0 * v1 * vv * s10 + u1 * u1 * vv * vv * s20 +
               2.0 * u1 * uu * v1 * v1 * s01 + 2.0 * u1 * uu * 2.0 * v1 * vv * s11 + 2.0 * u1 * uu * vv * vv * s21 +
               uu * uu * v1 * v1 * s02 + uu * uu * 2.0 * v1 * vv * s12 + uu * uu * vv * vv * s22;
      const double a = -2.0 + 2.0 * uu, b = 2.0 * (1.0 - 2.0 * uu), cq = uu * 2.0;
      fe.gx = 1.0 / res * (a * v1 * v1 * s00 + a * 2.0 * v1 * vv * s10 + a * vv * vv * s20 +
                           b * v1 * v1 * s01 + b * 2.0 * v1 * vv * s11 + b * vv * vv * s21 +
                           cq * v1 * v1 * s02 + cq * 2.0 * v1 * vv * s12 + cq * vv * vv * s22);
      const double av = -2.0 + 2.0 * vv, bv = 2.0 * (1.0 - 2.0 * vv), cv = 2.0 * vv;
      fe.gy = 1.0 / res * (u1 * u1 * av * s00 + u1 * u1 * bv * s10 + u1 * u1 * cv * s20 +
                           2.0 * u1 * uu * av * s01 + 2.0 * u1 * uu * bv * s11 + 2.0 * u1 * uu * cv * s21 +
                           uu * uu * av * s02 + uu * uu * bv * s12 + uu * uu * cv * s22);
      const double r2 = 1.0 / (res * res);
      fe.hxx = r2 * (2.0 * v1 * v1 * s00 + 2.0 * v1 * 2.0 * vv * s10 + 2.0 * vv * vv * s20 +
                     -4.0 * v1 * v1 * s01 + -4.0 * v1 * 2.0 * vv * s11 + -4.0 * vv * vv * s21 +
                     2.0 * v1 * v1 * s02 + 2.0 * v1 * 2.0 * vv * s12 + 2.0 * vv * vv * s22);
      fe.hyy = r2 * (2.0 * u1 * u1 * s00 + -4.0 * u1 * u1 * s10 + 2.0 * u1 * u1 * s20 +
                     2.0 * u1 * 2.0 * uu * s01 + -4.0 * u1 * 2.0 * uu * s11 + 2.0 * u1 * 2.0 * uu * s21 +
                     2.0 * uu * uu * s02 + -4.0 * uu * uu * s12 + 2.0 * uu * uu * s22);
      fe.hxy = r2 * (a * av * s00 + a * bv * s10 + a * cv * s20 +
                     b * av * s01 + b * bv * s11 + b * cv * s21 +
                     2.0 * uu * av * s02 + 2.0 * uu * bv * s12 + 2.0 * uu * cv * s22);
      const IlNodeW<GEN> NW{C, GEN ? (T.node_w + (size_t)c * IL_NW).p : nullptr, (double)pf};
      const double Lc = il_node_cost<GEN>(NW, x, u, fe);
      if (valid) {
        const GP<double> Lxx = T.Lxx + (size_t)c * 36, Lx = T.Lx + (size_t)c * 6;
        T.L[c] = Lc;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          double h = 0.0, gk = 0.0;
          if (k == 0) { h += fe.hxx; gk += fe.gx; }
          if (k == 1) { h += fe.hyy; gk += fe.gy; }
          h += 2.0 * NW.wdes(k);
          gk += 2.0 * (NW.wdes(k) * (x[k] - NW.des(k)));
          const double w = NW.wcon(k);
          if (x[k] > NW.ub(k) || x[k] < NW.lb(k)) h += 2.0 * w;
          if (x[k] > NW.ub(k)) gk += 2.0 * w * (x[k] - NW.ub(k));
          else if (x[k] < NW.lb(k)) gk += 2.0 * w * (x[k] - NW.lb(k));
          Lxx[k * 7] = h;
          Lx[k] = gk;
        }
        { double h = 0.0; h += fe.hxy; Lxx[1] = h; Lxx[6] = h; }
      }
    }
    IL_PT(13); IL_PCNT(14);
    __syncthreads();
  }
}

// Barrier over the G workgroups that share one wide cost tree (k_ilqr<GEN, true>): bar[0] = arrivals, bar[1] = generation.  Thread 0
// of a workgroup publishes the workgroup's global writes (agent-scope release), arrives, waits for the generation to move and
// invalidates the CU's vector L1 (agent-scope acquire) before anybody in the workgroup reads what the others wrote.  The
// workgroups of a tree are co-resident by construction (the host launches at most one workgroup per CU).
// Returns true when the launch was aborted: a workgroup that waits longer than ~2 s for its peers (they can only be missing if the
// launch was not fully resident) raises the launch-wide abort word, and every workgroup leaves at its next barrier; the host
// reports the failure instead of a hung device.
#define IL_SYNC_SPINS (1u << 24)
// poll intervals (s_sleep units of 64 cycles) of the workgroups that wait for another one's word
#ifndef IL_SLEEP_FOLLOW
#define IL_SLEEP_FOLLOW 8
#endif
#ifndef IL_SLEEP_SPEC
#define IL_SLEEP_SPEC 2
#endif
#ifndef IL_SLEEP_DONE
#define IL_SLEEP_DONE 2
#endif
__device__ __forceinline__ bool il_tree_sync(unsigned *bar, int G, unsigned *abort_word) {
  __shared__ int sh_aborted;
  __syncthreads();
  if (threadIdx.x == 0) {
    int aborted = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");     // (a workgroup-scope release -- no L2 write-back, valid only while all of
                                                           // a tree's workgroups share an XCD -- measured 2-5 % faster: not worth the assumption)
    const unsigned gen = __hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned prev = __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == (unsigned)G - 1u) {
      __hip_atomic_store(&bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&bar[1], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      unsigned spins = 0;
      while (__hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
        __builtin_amdgcn_s_sleep(4);
        if ((++spins & 0xfffu) == 0u) {          // the abort word is only looked at by a workgroup that has been waiting for a while
          if (spins >= IL_SYNC_SPINS) __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { aborted = 1; break; }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    sh_aborted = aborted;
  }
  __syncthreads();
  return sh_aborted != 0;
}

// LM schedule after one rejection (solver.py:153-158)
__device__ __forceinline__ void il_reject_update(double &mu, double &delta) {
  delta = fmax(1.0, delta) * 2.0;
  mu = fmax(1e-6, mu * delta);
}

// One iLQR.fit (solver.py:80-167) of one cost tree; starts from the controls in T.us.  MULTI == false: one workgroup per tree.
// MULTI == true (wide trees): G workgroups share the tree -- every "parallel for" below (segments of a level, (node, alpha) cost
// chunks, node blocks of the derivative pass, array copies) is dealt over the G x 8 waves, every phase boundary is a barrier
// over the G workgroups (il_tree_sync), the control variables (mu, delta, J, accepted slot ...) are recomputed identically by
// every workgroup from the same global data.  Same arithmetic per item, same results.
// SLOTS (k_ilqr<GEN, 2>, this workgroup = the master of its tree, T.ctl != null): every pass is published to the tree's follower workgroups,
// which evaluate the Levenberg-Marquardt slots 1 .. n - 1 on their own CUs (il_follow) while this workgroup evaluates slot 0 as a single workgroup
// would: a run of rejections costs one pass per n iterations and the accepted iterations cost what they always did.
__device__ __forceinline__ void il_flip_sets(IlqrTreeDev &T) {
  T.relag.p += T.dset; T.Fx.p += T.dset; T.L.p += T.dset; T.Lx.p += T.dset; T.Lxx.p += T.dset; T.rel.p += T.drel;
  T.dset = -T.dset; T.drel = -T.drel;
}

// `cur` (SLOTS with a derivative speculator, T.dset != 0): which of the launch's two derivative sets T.Fx .. T.rel are at the moment (0 = as
// launched); it lives across the fits of a launch -- a late speculation of the previous fit still writes the set that is NOT the nominal one.
template <bool GEN, bool MULTI, bool SLOTS>
__device__ __forceinline__ void il_fit(IlqrTreeDev &T, int &cur, const IlqrConst &C, int ph, double *stats, double *trace, int wg, int G, unsigned *bar, unsigned *abort_word) {
  extern __shared__ double il_dsm[];
  // LDS carve: per-wave scratch [IL_WAVES][IL_SCR] | cost sums [IL_LSUM] | compact records [IL_RECS] | staging [IL_DSTG] floats
  double *scr = il_dsm + (size_t)(threadIdx.x >> 6) * IL_SCR;
  double *lsum = il_dsm + (size_t)IL_WAVES * IL_SCR;
  double *recs0 = lsum + IL_LSUM;                                            // records region: IL_RECS doubles
  double *recs = recs0 + (size_t)(threadIdx.x >> 6) * 6 * IL_RA;            // cost pass: 6 nodes' records per wave
  float *dstg = reinterpret_cast<float *>(recs0 + IL_RECS);                  // derivative pass staging
  if ((threadIdx.x & 63) < 12) {
    const int l = threadIdx.x & 63;
    scr[IL_CST + l] = l == 0 ? 1.0 : (l == 6 ? C.dt : 0.0);
  }
  __shared__ double Jnew[IL_SLOTS][IL_NA];
  __shared__ double sh_mu, sh_delta, sh_J;
  __shared__ int sh_accepted, sh_converged, sh_stop, sh_sing, sh_pick, sh_slot, sh_it, sh_nspec, sh_hint, sh_ntot;
  __shared__ int sh_spec, sh_hit, sh_nreq, sh_nhit;      // derivative speculator: asked this pass / its set is the accepted candidate's / counts
  __shared__ unsigned sh_gen;
  __shared__ unsigned sh_fsing[IL_SLOTS];      // followers' singular flags of the pass (collected side by side before the selection)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int M = T.M;
  const int gw = MULTI ? wg * IL_WAVES + wave : wave, nw = MULTI ? G * IL_WAVES : IL_WAVES;      // this wave among the tree's waves
  const int gt = MULTI ? wg * IL_THREADS + tid : tid, nt = MULTI ? G * IL_THREADS : IL_THREADS;  // this thread among the tree's
#define IL_SYNC() do { if (MULTI) { if (il_tree_sync(bar, G, abort_word)) return; } else { __threadfence_block(); __syncthreads(); } } while (0)
#ifdef IL_PROFILE
  long long prof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  if (tid == 0) {
    sh_mu = 1.0; sh_delta = 2.0; sh_accepted = 1; sh_converged = 0; sh_stop = 0; sh_J = 0.0; sh_pick = 0; sh_slot = 0; sh_it = 0; sh_hint = 1; sh_ntot = 1;
    sh_spec = 0; sh_hit = 0; sh_nreq = 0; sh_nhit = 0;
    if (SLOTS) sh_gen = __hip_atomic_load(&T.ctl->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (this workgroup is its only writer)
  }
  if (MULTI && wg == 0 && tid == 0) {       // singular-slot words of both pass parities (read behind the barriers below)
    __hip_atomic_store(&bar[2], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&bar[3], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- initial nominal rollout (solver.py:255-330) = candidate slot 0 with k = K = 0, alpha = 0
  for (int q = gt; q < M * 2; q += nt) T.k[q] = 0.0;
  for (int q = gt; q < M * 12; q += nt) T.K[q] = 0.0;
  for (int q = gt; q < M * 6; q += nt) T.xs[q] = 0.0;
  // constant entries of F_x (identity, [2][4] = dt) and l_xx (zeros); the derivative pass only rewrites the rest
  for (int q = gt; q < M * 36; q += nt) {
    const int e = q % 36;
    T.Fx[q] = (e % 7 == 0) ? 1.0 : (e == 16 ? C.dt : 0.0);
    T.Lxx[q] = 0.0;
    if (SLOTS && T.dset) { (T.Fx + T.dset)[q] = (e % 7 == 0) ? 1.0 : (e == 16 ? C.dt : 0.0); (T.Lxx + T.dset)[q] = 0.0; }
  }
  IL_SYNC();
  for (int s = 0; s < T.n_fsteps; ++s) {
    const int lo = T.fstep_start[s], ni = T.fstep_start[s + 1] - lo;
    for (int w = gw; w * IL_PACK < ni; w += nw) il_rollout_packed(C, T, lo, ni, ni, w, 1, 0 IL_PROF_PASS);
    IL_SYNC();
  }
  bool staged = false;          // il_deriv_pass: this fit's agent rows are in the staging region already
  long long t_der = 0, t_bw = 0, t_ls = 0, t_sel = 0, t_roll = 0, t_mark = clock64();
#define IL_MARK(acc) do { long long now_ = clock64(); acc += now_ - t_mark; t_mark = now_; } while (0)
  // Each pass of this loop consumes 1..IL_SPEC reference iterations: the backward pass and line search are
  // evaluated for the current mu AND for the next mu values the LM schedule would visit if the step keeps
  // being rejected (a deterministic sequence); the first slot with an improving step is taken, exactly
  // what the sequential loop of solver.py:133-158 would have reached.
  int n_pass = 0;
  while (sh_it < C.max_iter) {
    ++n_pass;
    if (sh_accepted) {
      // adopt the accepted candidate as the nominal trajectory, then derivatives for all nodes in parallel
      const size_t off = ((size_t)sh_slot * IL_NA + sh_pick) * M;
      const GP<double> xn = T.xs_new + off * 6, un = T.us_new + off * 2;
      for (int q = gt; q < M * 6; q += nt) T.xs[q] = xn[q];
      for (int q = gt; q < M * 2; q += nt) T.us[q] = un[q];
      if (SLOTS && sh_hit) {
        // the accepted candidate is the one the speculator took (slot 0, first step size) while this workgroup priced the candidates: its
        // derivatives lie in the other set -- wait for the rest of them (il_speculate publishes as a follower does: see the selection below)
        if (tid == 0) {
          IlSlotCtl *ctl = T.ctl;
          while (__hip_atomic_load(&ctl->sdone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != sh_gen) __builtin_amdgcn_s_sleep(1);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        il_flip_sets(T);
        cur ^= 1;
      } else {
        IL_SYNC();
        il_deriv_pass<GEN>(C, T, lsum, dstg, recs0, reinterpret_cast<unsigned *>(lsum + 9 * 64), MULTI ? wg : 0, MULTI ? G : 1, staged IL_PROF_PASS);
      }
      IL_SYNC();
      if (M <= IL_LSUM) {
        for (int q = tid; q < M; q += IL_THREADS) lsum[q] = T.L[q];
        __syncthreads();
        if (tid == 0) { sh_J = il_np_sum(lsum, M); sh_accepted = 0; }
      } else if (M <= 64 * (IL_LSUM / 3)) {
        // a big tree: leaf sums of numpy's pairwise recursion by all threads, folded by one (il_np_sum walked 30 k nodes through dependent
        // global round trips: milliseconds per accepted step at stress scale); a leaf holds > 64 elements, the table 3 doubles per leaf
        const double Jn = il_np_sum_wg(T.L, M, lsum);
        if (tid == 0) { sh_J = Jn; sh_accepted = 0; }
      } else if (tid == 0) { sh_J = il_np_sum(T.L.p, M); sh_accepted = 0; }
      __syncthreads();
    }
    IL_MARK(t_der);
    // number of speculative slots this pass: bounded by waves, remaining iterations and the mu >= 1e10 stop
    if (tid == 0) {
      // speculate only while the LM schedule is in a run of rejections (the previous pass saw one): when every
      // step is accepted the extra slots would only add work to the cost pass
      int ns = sh_hint;
      const int maxseg = T.max_level_segs > 0 ? T.max_level_segs : 1;   // widest segment level
      while (ns > 1 && ns * maxseg > IL_SPEC_OVERSUB * nw) --ns;
      // followers: every pass carries as many slots as there are workgroups (their CUs would idle otherwise), this one evaluates slot 0.  They
      // are used only once ALL of them run (a launch that is not fully resident keeps its slots in this workgroup: nobody waits for a
      // workgroup that has not started)
      const unsigned alive = SLOTS ? __hip_atomic_load(&T.ctl->alive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
      const bool fol = SLOTS && G > 1 && ((alive | 1u) & ((1u << G) - 1u)) == (1u << G) - 1u;
      sh_spec = SLOTS && fol && T.dset && ((alive >> G) & 1u);        // (the speculator is workgroup G of the tree)
      sh_hit = 0;
      if (fol) ns = G < IL_SLOTS ? G : IL_SLOTS;
      if (ns > C.max_iter - sh_it) ns = C.max_iter - sh_it;
      double mu = sh_mu, de = sh_delta;
      int cnt = 1;
      for (; cnt < ns; ++cnt) { il_reject_update(mu, de); if (mu >= 1e10) break; }   // slot cnt would never be reached
      sh_ntot = cnt < ns ? cnt : ns;
      sh_nspec = fol ? 1 : sh_ntot;
      sh_sing = 0;
      if (SLOTS) {
        // the command of this pass (the derivative pass above ended with a workgroup barrier: its writes are ordered before this release)
        IlSlotCtl *ctl = T.ctl;
        ctl->mu = sh_mu; ctl->delta = sh_delta;
        __hip_atomic_store(&ctl->cmd, (unsigned)(fol ? sh_ntot : 1) | ((unsigned)ph << 8) | ((unsigned)cur << 24), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sh_gen += 1u;
        __hip_atomic_store(&ctl->gen, sh_gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
      // wide trees: the singular-slot mask is a global word, one per pass parity (this pass's word was cleared during the
      // previous pass, behind several barriers)
      if (MULTI && wg == 0) __hip_atomic_store(&bar[2 + ((n_pass + 1) & 1)], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const int nspec = sh_nspec;
    // per-wave slot view: slot = item / segments
    // ---------------- backward pass (solver.py:332-373): segments, deepest segment level first ----------------
    for (int d = T.n_slevels - 1; d >= 0; --d) {
      const int lo = T.slevel_start[d], hi = T.slevel_start[d + 1];
      for (int w = gw; w < (hi - lo) * nspec; w += nw) {
        const int slot = w / (hi - lo), seg = T.slevel_segs[lo + w % (hi - lo)];
        double mu = sh_mu, de = sh_delta;
        for (int e = 0; e < slot; ++e) il_reject_update(mu, de);
        IlqrTreeDev Ts = T;
        Ts.k += (size_t)slot * M * 2; Ts.K += (size_t)slot * M * 12; Ts.Vx += (size_t)slot * M * 6; Ts.Vxx += (size_t)slot * M * 36;
        int rec[14];
        {
          const int IL_AS1 *sr = T.seg_rec.g() + (size_t)seg * 16;
#pragma unroll
          for (int e = 0; e < 14; ++e) rec[e] = __builtin_amdgcn_readfirstlane(sr[e]);
        }
        const int s0 = rec[0], s1 = rec[1];
        {
          // value function entering the segment = sum over the last node's children, in child order; the first six children's rows
          // are requested together (one round trip instead of two per child)
          const int nch = rec[5];
          double acc = 0.0;
          if (lane < 42) {
            double v[6];
#pragma unroll
            for (int e = 0; e < 6; ++e) {
              const int ch = rec[8 + e];
              v[e] = lane < 36 ? Ts.Vxx[(size_t)ch * 36 + lane] : Ts.Vx[(size_t)ch * 6 + (lane - 36)];      // (unused slots name node 0: a valid row nobody adds)
            }
#pragma unroll
            for (int e = 0; e < 6; ++e) if (e < nch) acc += v[e];
            for (int e = rec[6] + 6; e < rec[6] + nch; ++e) {
              const int ch = T.child_list[e];
              acc += lane < 36 ? Ts.Vxx[(size_t)ch * 36 + lane] : Ts.Vx[(size_t)ch * 6 + (lane - 36)];
            }
          }
          IL_WFENCE();
          if (lane < 36) scr[36 + lane] = acc;
          else if (lane < 42) scr[176 + lane - 36] = acc;
          IL_WFENCE();
        }
        const int sing = il_backward_segment<GEN>(C, T, Ts, s0, s1, rec[2], rec[4], mu, scr IL_PROF_PASS);
        if (sing) {
          if (lane == 0) {
            if (MULTI) __hip_atomic_fetch_or(&bar[2 + (n_pass & 1)], 1u << slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else atomicOr(&sh_sing, 1 << slot);
          }
        }
        else {
          const int c = rec[3];            // value function of the segment head, for the parent's gather
          if (lane < 36) Ts.Vxx[(size_t)c * 36 + lane] = scr[36 + lane];
          else if (lane < 42) Ts.Vx[(size_t)c * 6 + lane - 36] = scr[176 + lane - 36];
        }
      }
      IL_SYNC();
    }
    IL_MARK(t_bw);
    if (MULTI) {
      if (tid == 0) sh_sing = (int)__hip_atomic_load(&bar[2 + (n_pass & 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
    }
    if (sh_sing & 1) {   // LinAlgError at the current mu: retry without raising mu (Q9) -- burns one iteration
      if (tid == 0) {
        if (trace && sh_it < T.trace_cap && (!MULTI || wg == 0)) {
          double *tr = trace + (size_t)sh_it * IL_TRACE_W;
          tr[0] = sh_mu; tr[1] = sh_J; tr[2] = -2.0; tr[3] = sh_J;
        }
        sh_it += 1;
      }
      __syncthreads();
      // wide trees: every workgroup must have read this pass's singular word before workgroup 0 clears it again (it is the word of
      // the pass after next, cleared at the top of the NEXT pass -- and this path has no other barrier in between)
      IL_SYNC();
      continue;
    }
    // slots behind a singular slot cannot be used this pass
    int nuse = nspec;
    for (int e = 1; e < nspec; ++e) if ((sh_sing >> e) & 1) { nuse = e; break; }
    // ---------------- line search: 10 alphas in lane groups, `nuse` mu slots in parallel waves ----------------
    // forward steps: the state chains of step s on the first waves, the costs of the nodes step s - 1 reached on the others (and on
    // the chain waves once they are through); one more step for the costs of the last nodes
    for (int s = 0; s <= T.n_fsteps; ++s) {
      int R = 0;
      if (s < T.n_fsteps) {
        const int lo = T.fstep_start[s], ni = T.fstep_start[s + 1] - lo;
        R = (ni * nuse + IL_PACK - 1) / IL_PACK;            // waves' worth of (slot, piece) items
        for (int w = gw; w < R; w += nw) il_rollout_packed(C, T, lo, ni, ni * nuse, w, 0, 0 IL_PROF_PASS);
      } else {
        IL_MARK(t_roll);
      }
      if (s > 0) {
        const int n0 = T.fstep_nstart[s - 1], cnt = T.fstep_nstart[s] - n0;
        // order in which the waves take cost chunks while R waves roll chains: the waves of a SIMD without a chain wave first (wave w
        // sits on SIMD w & 3; two busy waves on a SIMD share its float64 pipe and would slow the chain), the chain waves' SIMD
        // partners next, the chain waves themselves last
        int rank = (gw + nw - R % nw) % nw;
        if (!MULTI && R < 4) {
          const int fr = 4 - R;             // SIMDs without a chain wave
          rank = wave < R ? 2 * fr + R + wave : (wave < 4 ? wave - R : (wave < 4 + R ? 2 * fr + wave - 4 : fr + wave - 4 - R));
        }
        il_cost_pass<GEN>(C, T, nuse, recs, rank, nw, T.fstep_nodes + (size_t)n0, cnt, 0 IL_PROF_PASS);
      }
      IL_SYNC();
      // the state chains are through: the speculator differentiates candidate (slot 0, first step size) while the last costs, the sums and the
      // selection run here (every wave's xs_new / us_new stores are complete behind the barrier above; this release publishes them)
      if (SLOTS && s == T.n_fsteps - 1 && tid == 0 && sh_spec) {
        __hip_atomic_store(&T.ctl->sreq, sh_gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        sh_spec = 2; sh_nreq += 1;
      }
    }
    IL_MARK(t_ls);
    if (M * IL_NA * nuse <= IL_LSUM) {
      for (int q = tid; q < M * IL_NA * nuse; q += IL_THREADS) lsum[q] = T.L_new[q];
      __syncthreads();
    } else {
      const double J = il_seq_sums_tiled(T.L_new, M, IL_NA * nuse, lsum);
      if (tid < IL_NA * nuse) Jnew[tid / IL_NA][tid % IL_NA] = J;
    }
    if (M * IL_NA * nuse <= IL_LSUM && tid < IL_NA * nuse) {
      double J = 0.0;   // python sum(): sequential
      const double *Ln = lsum + (size_t)tid * M;
      // eight values requested together, added in node order (one dependent LDS round trip per node otherwise)
      int c = 0;
      for (; c + 8 <= M; c += 8) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = Ln[c + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) J += v[k];
      }
      for (; c < M; ++c) J += Ln[c];
      Jnew[tid / IL_NA][tid % IL_NA] = J;
    }
    __syncthreads();
    // With followers, and every slot of this workgroup rejected: the followers' results are collected by one lane each, side by side (the
    // selection below then scans them in slot order from LDS) -- thread 0 used to wait for them one after the other, a word, a fence and ten
    // loads per slot, in the run of rejections that ends every fit.  The followers do equal work and publish at about the same time.
    bool gathered = false;
    if (SLOTS && sh_ntot > nspec) {
      bool own = false;          // (uniform: every thread reads the same LDS values)
      for (int slot = 0; slot < nuse; ++slot)
        for (int a = 0; a < IL_NA; ++a) own |= Jnew[slot][a] < sh_J;
      if (!own) {
        if (tid >= nuse && tid < sh_ntot) {
          IlSlotCtl *ctl = T.ctl;
          while (__hip_atomic_load(&ctl->done[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != sh_gen) __builtin_amdgcn_s_sleep(IL_SLEEP_DONE);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          sh_fsing[tid] = __hip_atomic_load(&ctl->sing[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          for (int a = 0; a < IL_NA; ++a) Jnew[tid][a] = ctl->Jnew[tid][a];
        }
        __syncthreads();
        gathered = true;
      }
    }
    if (tid == 0) {
      double mu = sh_mu, de = sh_delta;
      int it = sh_it;
      bool done = false;
      const int nsel = (SLOTS && sh_ntot > nspec) ? sh_ntot : nuse;      // with followers: their slots behind this workgroup's
      for (int slot = 0; slot < nsel && !done; ++slot) {
        if (SLOTS && slot >= nuse && gathered) {
          if (sh_fsing[slot]) break;          // slots behind a singular slot cannot be used this pass
        } else if (SLOTS && slot >= nuse) {
          // a follower's slot, looked at only because every earlier slot was rejected: wait for its result (it started a command latency
          // behind this workgroup's own slot and has had the selection's time to catch up)
          IlSlotCtl *ctl = T.ctl;
          while (__hip_atomic_load(&ctl->done[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != sh_gen) __builtin_amdgcn_s_sleep(IL_SLEEP_DONE);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          if (__hip_atomic_load(&ctl->sing[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;     // slots behind a singular slot cannot be used this pass
          for (int a = 0; a < IL_NA; ++a) Jnew[slot][a] = ctl->Jnew[slot][a];
        }
        // (mu, de) is the LM state the reference holds when it runs this iteration
        int pick = -1;
        for (int a = 0; a < IL_NA; ++a)
          if (Jnew[slot][a] < sh_J) { pick = a; break; }
        if (trace && it < T.trace_cap && (!MULTI || wg == 0)) {
          double *tr = trace + (size_t)it * IL_TRACE_W;
          tr[0] = mu; tr[1] = sh_J; tr[2] = (double)pick; tr[3] = pick >= 0 ? Jnew[slot][pick] : sh_J;
        }
        it += 1;
        if (pick >= 0) {
          sh_pick = pick; sh_slot = slot;
          if (fabs((sh_J - Jnew[slot][pick]) / sh_J) < 1e-6) sh_converged = 1;
          sh_accepted = 1;
          de = fmin(1.0, de) / 2.0;
          mu *= de;
          if (mu <= 1e-6) mu = 0.0;
          done = true;
        } else {
          il_reject_update(mu, de);
          if (mu >= 1e10) { sh_stop = 1; done = true; }
        }
      }
      sh_mu = mu; sh_delta = de; sh_it = it;
      sh_hint = (sh_accepted && sh_slot == 0) ? 1 : IL_SPEC;
      if (SLOTS && sh_spec == 2 && sh_accepted && sh_slot == 0 && sh_pick == 0) { sh_hit = 1; sh_nhit += 1; }
      // An accepted candidate of a follower's slot is adopted (at the top of the next pass, by every thread) from that slot's arrays.  What makes
      // the follower's writes visible to this CU's plain loads is the pattern of il_tree_sync (MI355X_MICROARCH.md, inter-workgroup visibility):
      // producer side -- every wave's stores are complete behind the follower's `__threadfence_block(); __syncthreads()`, then its thread 0
      // publishes done[slot] with an agent-scope RELEASE (writes the XCD's L2 back); consumer side -- this thread read done[slot] and issued
      // the agent-scope ACQUIRE above (invalidates THIS CU's vector L1, the only cache private to the workgroup), and the __syncthreads()
      // below orders every other wave's loads behind it.  No per-thread __threadfence() is needed on either side.
    }
    __syncthreads();
    IL_MARK(t_sel);
    if (sh_converged || sh_stop) break;
  }
  // the reference returns the last ACCEPTED xs/us (Q19: J_opt is the cost before that step)
  if (sh_accepted) {
    const size_t off = ((size_t)sh_slot * IL_NA + sh_pick) * M;
    const GP<double> xn = T.xs_new + off * 6, un = T.us_new + off * 2;
    for (int q = gt; q < M * 6; q += nt) T.xs[q] = xn[q];
    for (int q = gt; q < M * 2; q += nt) T.us[q] = un[q];
  }
  if (tid == 0 && (!MULTI || wg == 0)) {
    stats[0] = (double)sh_it;
    stats[1] = (double)sh_converged;
    stats[2] = sh_J;
    stats[3] = sh_mu;
    stats[4] = (double)t_der; stats[5] = (double)t_bw; stats[6] = (double)t_ls; stats[7] = (double)t_sel;
#ifdef IL_PROFILE
    for (int q = 0; q < 16; ++q) stats[8 + q] = (double)prof[q];
#else
    stats[8] = (double)t_roll;          // state-chain part of the line search (stats[6] = its cost pass)
    stats[9] = (double)sh_nreq; stats[10] = (double)sh_nhit;      // derivative speculator: passes it was asked in / accepted candidates it had ready
#endif
    stats[IL_NSTAT - 1] = (double)n_pass;
  }
  IL_SYNC();
#undef IL_SYNC
}

// A follower workgroup of a tree whose Levenberg-Marquardt slots are spread over workgroups (IlSlotCtl): waits for the master's command, evaluates
// slot `wg` of that pass -- the backward sweep at the mu the schedule reaches after `wg` rejections, the state chains of the ten step sizes,
// their costs, the ten sums in the reference's order -- with the code the master runs for its own slot (il_backward_segment, il_rollout_packed,
// il_cost_pass: same arithmetic per item), publishes them and waits again.  It only ever READS the nominal trajectory and its derivatives; a
// command it picks up late (the master accepted an earlier slot and moved on) yields a result nobody reads.
template <bool GEN>
__device__ __forceinline__ void il_follow(const IlqrTreeDev &T0, const IlqrConst *consts, int wg) {
  extern __shared__ double il_dsm[];
  double *scr = il_dsm + (size_t)(threadIdx.x >> 6) * IL_SCR;
  double *lsum = il_dsm + (size_t)IL_WAVES * IL_SCR;
  double *recs0 = lsum + IL_LSUM;
  double *recs = recs0 + (size_t)(threadIdx.x >> 6) * 6 * IL_RA;
  __shared__ unsigned f_gen, f_cmd;
  __shared__ double f_mu, f_de;
  __shared__ int f_sing, f_abort;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int M = T0.M;
  IlSlotCtl *ctl = T0.ctl;
  if (tid == 0) __hip_atomic_fetch_or(&ctl->alive, 1u << wg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned last = 0;
  int ph_cst = -1;
  for (;;) {
    __syncthreads();
    if (tid == 0) {
      unsigned g;
      while ((g = __hip_atomic_load(&ctl->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == last) __builtin_amdgcn_s_sleep(IL_SLEEP_FOLLOW);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      f_gen = g; f_cmd = __hip_atomic_load(&ctl->cmd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      f_mu = ctl->mu; f_de = ctl->delta; f_sing = 0; f_abort = 0;
    }
    __syncthreads();
    const unsigned g = f_gen, cmd = f_cmd;
    last = g;
    if ((cmd >> 16) & 0xffu) return;
    const int ns = (int)(cmd & 0xffu), ph = (int)((cmd >> 8) & 0xffu);
    if (wg >= ns) continue;
    const IlqrConst &C = consts[ph];
    if (ph != ph_cst) {          // the Riccati sweep's constant operands {1,0,..} / {dt,0,..} (il_fit writes them at its start)
      if (lane < 12) scr[IL_CST + lane] = lane == 0 ? 1.0 : (lane == 6 ? C.dt : 0.0);
      ph_cst = ph;
    }
    double mu = f_mu, de = f_de;
    for (int e = 0; e < wg; ++e) il_reject_update(mu, de);
    IlqrTreeDev T = T0;
    if ((cmd >> 24) & 1u) il_flip_sets(T);        // the master's nominal derivatives are in the second set (a speculated pass was accepted)
    IlqrTreeDev Ts = T;
    Ts.k += (size_t)wg * M * 2; Ts.K += (size_t)wg * M * 12; Ts.Vx += (size_t)wg * M * 6; Ts.Vxx += (size_t)wg * M * 36;
    // ---- backward pass of this slot (solver.py:332-373), as in il_fit
    for (int d = T.n_slevels - 1; d >= 0; --d) {
      const int lo = T.slevel_start[d], hi = T.slevel_start[d + 1];
      for (int w = wave; w < hi - lo; w += IL_WAVES) {
        const int seg = T.slevel_segs[lo + w];
        int rec[14];
        {
          const int IL_AS1 *sr = T.seg_rec.g() + (size_t)seg * 16;
#pragma unroll
          for (int e = 0; e < 14; ++e) rec[e] = __builtin_amdgcn_readfirstlane(sr[e]);
        }
        {
          const int nch = rec[5];
          double acc = 0.0;
          if (lane < 42) {
            double v[6];
#pragma unroll
            for (int e = 0; e < 6; ++e) {
              const int ch = rec[8 + e];
              v[e] = lane < 36 ? Ts.Vxx[(size_t)ch * 36 + lane] : Ts.Vx[(size_t)ch * 6 + (lane - 36)];
            }
#pragma unroll
            for (int e = 0; e < 6; ++e) if (e < nch) acc += v[e];
            for (int e = rec[6] + 6; e < rec[6] + nch; ++e) {
              const int ch = T.child_list[e];
              acc += lane < 36 ? Ts.Vxx[(size_t)ch * 36 + lane] : Ts.Vx[(size_t)ch * 6 + (lane - 36)];
            }
          }
          IL_WFENCE();
          if (lane < 36) scr[36 + lane] = acc;
          else if (lane < 42) scr[176 + lane - 36] = acc;
          IL_WFENCE();
        }
#ifdef IL_PROFILE
        long long prof[16];
#endif
        const int sing = il_backward_segment<GEN>(C, T, Ts, rec[0], rec[1], rec[2], rec[4], mu, scr IL_PROF_PASS, &ctl->gen, g);
        if (sing) {
          if (lane == 0) { if (sing == 2) f_abort = 1; else f_sing = 1; }
        } else {
          const int c = rec[3];
          if (lane < 36) Ts.Vxx[(size_t)c * 36 + lane] = scr[36 + lane];
          else if (lane < 42) Ts.Vx[(size_t)c * 6 + lane - 36] = scr[176 + lane - 36];
        }
      }
      __threadfence_block();
      __syncthreads();
      if (f_abort) break;
    }
    if (f_abort) continue;          // the master has published its next command: this one's result would not be read
    if (!f_sing) {
      // ---- line search of this slot (solver.py:202-240): state chains, then candidate costs, as in il_fit with one slot
      for (int s = 0; s <= T.n_fsteps; ++s) {
        int R = 0;
        if (s < T.n_fsteps) {
          const int lo = T.fstep_start[s], ni = T.fstep_start[s + 1] - lo;
          R = (ni + IL_PACK - 1) / IL_PACK;
#ifdef IL_PROFILE
          long long prof[16];
#endif
          for (int w = wave; w < R; w += IL_WAVES)
            if (il_rollout_packed(C, T, lo, ni, ni, w, 0, wg IL_PROF_PASS, &ctl->gen, g) && lane == 0) f_abort = 1;
        }
        if (s > 0) {
          const int n0 = T.fstep_nstart[s - 1], cnt = T.fstep_nstart[s] - n0;
          int rank = (wave + IL_WAVES - R % IL_WAVES) % IL_WAVES;
          if (R < 4) {
            const int fr = 4 - R;
            rank = wave < R ? 2 * fr + R + wave : (wave < 4 ? wave - R : (wave < 4 + R ? 2 * fr + wave - 4 : fr + wave - 4 - R));
          }
#ifdef IL_PROFILE
          long long prof[16];
#endif
          il_cost_pass<GEN>(C, T, 1, recs, rank, IL_WAVES, T.fstep_nodes + (size_t)n0, cnt, wg IL_PROF_PASS, &ctl->gen, g);
        }
        __threadfence_block();
        __syncthreads();
        if (f_abort) break;
      }
      if (tid == 0 && !f_abort && __hip_atomic_load(&ctl->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != g) f_abort = 1;   // (cost chunks were skipped)
      __syncthreads();
      if (f_abort) continue;
      // ---- the ten sums J(alpha) = sum over the nodes in node order (python sum(): sequential), as il_fit adds them
      const GP<double> Lg = T.L_new + (size_t)wg * IL_NA * M;
      if (M * IL_NA <= IL_LSUM) {
        for (int q = tid; q < M * IL_NA; q += IL_THREADS) lsum[q] = Lg[q];
        __syncthreads();
      } else {
        const double J = il_seq_sums_tiled(Lg, M, IL_NA, lsum);
        if (tid < IL_NA) ctl->Jnew[wg][tid] = J;
      }
      if (M * IL_NA <= IL_LSUM && tid < IL_NA) {
        double J = 0.0;
        const double *Ln = lsum + (size_t)tid * M;
        int c = 0;
        for (; c + 8 <= M; c += 8) {
          double v[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = Ln[c + k];
#pragma unroll
          for (int k = 0; k < 8; ++k) J += v[k];
        }
        for (; c < M; ++c) J += Ln[c];
        ctl->Jnew[wg][tid] = J;
      }
    }
    __threadfence_block();
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_store(&ctl->sing[wg], (unsigned)f_sing, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&ctl->done[wg], g, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// The derivative speculator of a tree whose slots are spread over workgroups (workgroup G, behind the followers).  An accepted pass of the
// master is: backward sweep, state chains, [the last costs, the ten sums, the selection], then the derivative pass at the accepted candidate --
// which is the pass's FIRST candidate (slot 0, step size alphas[0]) for three accepted iterations in four of the recorded loops
// (tests/diag/gpu_ilqr_alpha_hist.py).  This workgroup runs that derivative pass -- il_deriv_pass itself, on xs_new / us_new of (slot 0, alpha 0),
// into the derivative set the nominal trajectory does NOT use -- as soon as the master's state chains are through (ctl->sreq), i.e. beside the
// bracketed part; the master, when its selection names exactly that candidate, waits for ctl->sdone and swaps the sets (il_flip_sets)
// instead of differentiating.  Same code on the same inputs: the same bits; a miss costs the master nothing (it never waits for a result
// it does not use) and a result nobody uses is overwritten by the next request.
template <bool GEN>
__device__ __forceinline__ void il_speculate(const IlqrTreeDev &T0, const IlqrConst *consts, int wg) {
  extern __shared__ double il_dsm[];
  double *lsum = il_dsm + (size_t)IL_WAVES * IL_SCR;
  double *recs0 = lsum + IL_LSUM;
  float *dstg = reinterpret_cast<float *>(recs0 + IL_RECS);
  __shared__ unsigned s_req, s_cmd;
  const int tid = threadIdx.x;
  IlSlotCtl *ctl = T0.ctl;
  if (tid == 0) __hip_atomic_fetch_or(&ctl->alive, 1u << wg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned last = 0;
  bool staged = false;
  for (;;) {
    __syncthreads();
    if (tid == 0) {
      unsigned r, cmd;
      for (;;) {
        r = __hip_atomic_load(&ctl->sreq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cmd = __hip_atomic_load(&ctl->cmd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (r != last || (cmd >> 16) & 0xffu) break;
        __builtin_amdgcn_s_sleep(IL_SLEEP_SPEC);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      s_req = r; s_cmd = __hip_atomic_load(&ctl->cmd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const unsigned r = s_req, cmd = s_cmd;
    if ((cmd >> 16) & 0xffu) return;
    last = r;
    // (cmd is the command of the pass that asked or of a later one of the same fit -- the master changes its set only while it waits for this
    // workgroup, and a request that is overtaken by the next fit's first command yields a result nobody reads, written to the set that is not
    // the nominal one there either: `cur` lives across the fits)
    const int ph = (int)((cmd >> 8) & 0xffu);
    IlqrTreeDev T = T0;
    if (!((cmd >> 24) & 1u)) il_flip_sets(T);        // write the set the nominal trajectory does not use
    T.xs = T0.xs_new; T.us = T0.us_new;            // candidate (slot 0, alpha index 0)
#ifdef IL_PROFILE
    long long prof[16];
#endif
    il_deriv_pass<GEN>(consts[ph], T, lsum, dstg, recs0, reinterpret_cast<unsigned *>(lsum + 9 * 64), 0, 1, staged IL_PROF_PASS);
    __threadfence_block();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&ctl->sdone, r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// n_phases == 1: one fit with consts[0].  n_phases == 2: the contingency planner's sequence (planner.py:174-178) in
// one launch -- the warm-start fit (consts[0]: lane term only) and then, from its controls, the full fit (consts[1]).
// T.stats receives IL_NSTAT doubles per phase.
// MODE 1 (wide trees): G workgroups share a tree's items; MODE 2 (narrow trees): G workgroups take a tree's Levenberg-Marquardt slots (workgroup 0
// = the master, il_fit; the others il_follow).  Both: block b -> tree 8 (b / 8G) + b % 8, workgroup (b % 8G) / 8 of it: the hardware deals
// consecutive blocks round-robin over the 8 XCDs, so the workgroups of one tree share an XCD (one L2).  bars: 4 words per tree + one abort word
// for the launch, zeroed by the host.
template <bool GEN, int MODE>
__global__ __launch_bounds__(IL_THREADS) void k_ilqr(const IlqrTreeDev *__restrict__ trees, const IlqrConst *__restrict__ consts,
                                                     int n_phases, int n_trees, int G, unsigned *__restrict__ bars, int spec) {
  constexpr bool MULTI = MODE == 1, SLOTS = MODE == 2;
  int t = blockIdx.x, wg = 0;
  if (MODE != 0) {
    const int Gl = G + (SLOTS ? spec : 0);      // workgroups per tree in the launch: MODE 2 may carry a derivative speculator behind the G slots
    const int r = blockIdx.x % (8 * Gl);
    t = 8 * (blockIdx.x / (8 * Gl)) + (r & 7);
    wg = r >> 3;
    if (t >= n_trees) return;
  }
  IlqrTreeDev T = trees[t];
  if (SLOTS && wg > 0) {
    if (bars[4 * n_trees + 1]) return;      // (tests: followers withheld -- the master then keeps its slots to itself)
    if (wg >= G) il_speculate<GEN>(T, consts, wg);
    else il_follow<GEN>(T, consts, wg);
    return;
  }
  int cur = 0;
  for (int ph = 0; ph < n_phases; ++ph)
    il_fit<GEN, MULTI, SLOTS>(T, cur, consts[ph], ph, (T.stats + (size_t)ph * IL_NSTAT).p, T.trace ? (T.trace + (size_t)ph * T.trace_cap * IL_TRACE_W).p : nullptr, wg, G, MULTI ? bars + 4 * t : nullptr, MULTI ? bars + 4 * n_trees : nullptr);
  if (T.h_xs && wg == 0) {
    // (il_fit ended with a barrier of the tree's workgroups behind the last adoption and the statistics)
    const int M = T.M;
    for (int q = threadIdx.x; q < M * 6; q += IL_THREADS) T.h_xs[q] = T.xs[q];
    for (int q = threadIdx.x; q < M * 2; q += IL_THREADS) T.h_us[q] = T.us[q];
    for (int q = threadIdx.x; q < n_phases * IL_NSTAT; q += IL_THREADS) T.h_stats[q] = T.stats[q];
    if (T.h_done) {
      __threadfence_system();
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(T.h_done, T.h_gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (SLOTS && threadIdx.x == 0) {          // the followers leave
    __hip_atomic_store(&T.ctl->cmd, 1u << 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&T.ctl->gen, __hip_atomic_load(&T.ctl->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

static inline size_t il_lds_bytes(int /*amax*/) {
  return ((size_t)IL_WAVES * IL_SCR + IL_LSUM + (size_t)IL_RECS) * sizeof(double) + (size_t)IL_DSTG * sizeof(float);
}

// TreeCost.l / l_x / l_u / l_xx / l_uu (cost.py:341-446) at arbitrary (x, u) of given nodes: one wave per
// query.  out[q] = { l, l_x[6], l_u[2], l_xx[36], l_uu diag[2] } (47 doubles).
#define IL_EVAL_OUT 47
template <bool GEN>
__global__ __launch_bounds__(64) void k_cost_eval(const IlqrTreeDev *__restrict__ trees, IlqrConst C, int nq,
                                                  const int *__restrict__ node, const double *__restrict__ xq,
                                                  const double *__restrict__ uq, double *__restrict__ out) {
  const IlqrTreeDev T = trees[0];
  extern __shared__ double il_dsm[];
  double *scr = il_dsm, *ag = il_dsm + IL_SCR;
  const int q = blockIdx.x, lane = threadIdx.x;
  if (q >= nq) return;
  const int c = node[q];
  double x[6], u[2];
#pragma unroll
  for (int k = 0; k < 6; ++k) x[k] = xq[(size_t)q * 6 + k];
  u[0] = uq[(size_t)q * 2]; u[1] = uq[(size_t)q * 2 + 1];
  if (C.use_exo) il_stage_agents(C, T, c, ag);
  double *o = out + (size_t)q * IL_EVAL_OUT;
  il_node_derivs<GEN>(C, T, c, x, u, scr, ag, o + 9, nullptr, o + 1, o);
  if (lane < 2) {
    const IlNodeW<GEN> NW{C, GEN ? (T.node_w + (size_t)c * IL_NW).p : nullptr, GEN ? 0.0 : (double)T.prob[c]};
    o[7 + lane] = 2.0 * (NW.wctrl(lane) * u[lane]);
    o[45 + lane] = 2.0 * NW.wctrl(lane);
  }
}
