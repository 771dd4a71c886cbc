/*
 * libmind_hip.so -- C-ABI of the MI355X-native MIND hot path.
 *
 * The reference (HKUST-Aerial-Robotics/MIND) is pure Python and has no FFI of its own; every entry
 * point below cites the reference Python interface it replaces.  All functions return 0 on success
 * or a negative MIND_E* code; no exception crosses the boundary.  The caller owns every buffer it
 * passes in (host or device as stated); the library never frees caller memory.  One context per
 * (process, GPU); calls on one context are serialised on its HIP stream.
 */
#ifndef MIND_HIP_H
#define MIND_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIND_OK 0
#define MIND_EINVAL (-1)   /* bad argument / shape */
#define MIND_ENOMEM (-2)   /* device allocation failed */
#define MIND_EHIP (-3)     /* HIP runtime error, see mind_last_error_string */
#define MIND_ESTATE (-4)   /* weights not loaded, context destroyed ... */
#define MIND_ENOTFOUND (-5)/* tensor name missing from the state_dict table */

typedef struct mind_ctx mind_ctx;

/* replaces MINDPlanner.init_device (planners/mind/planner.py:35-39).  `stream` is a hipStream_t
 * on which all work of this context is queued (NULL = the device's default stream).  */
int mind_ctx_create(int device, void *stream, mind_ctx **out);
int mind_ctx_destroy(mind_ctx *ctx);
const char *mind_last_error_string(mind_ctx *ctx);
/* blocks until all work queued on the context's stream is done */
int mind_ctx_synchronize(mind_ctx *ctx);

/* One tensor of ckpt["state_dict"] (planners/mind/planner.py:46-47): fp32, C-contiguous, host. */
typedef struct {
  const char *name;     /* reference state_dict key, e.g. "fusion_net.proj_actor.0.weight" */
  const float *data;    /* host pointer */
  int64_t numel;
} mind_tensor_desc;

/* replaces network.load_state_dict(ckpt["state_dict"]) + .to(device) (planner.py:47-48).  The
 * library re-lays the 328 tensors out into its packed device blob (MFMA fragment order). */
int mind_weights_load(mind_ctx *ctx, const mind_tensor_desc *tensors, int n_tensors);

/* Inputs of ScenePredNet.pre_process / forward (planners/mind/networks/network.py:582-606) for a
 * batch of B scenes (= AIME scenario-tree nodes).  Scene b has a_b = actor_off[b+1]-actor_off[b]
 * agents (ego first) and l_b lane polylines; its token order is [agents, lanes, cls].
 * Device pointers unless stated. */
typedef struct {
  int n_scenes;
  const int32_t *actor_off;   /* HOST [B+1] prefix sums (ACTOR_IDCS are contiguous ranges)        */
  const int32_t *lane_off;    /* HOST [B+1]                                                       */
  const float *actors;        /* [A,14,48]  ACTORS  (mind/utils.py:114-139)                       */
  const float *lanes;         /* [L,10,16]  LANES   (mind/utils.py:103-110); may be NULL if lane_feat */
  const float *lane_feat;     /* [L,128] optional LaneNet output computed earlier (reuse across tree
                                 nodes of one plan: lane node features do not change), or NULL    */
  const float *actor_ctrs;    /* [A,2] TRAJS_CTRS, [A,2] TRAJS_VECS                               */
  const float *actor_vecs;
  const float *lane_ctrs;     /* [L,2] lane_ctrs, lane_vecs of each scene's LANE_GRAPH            */
  const float *lane_vecs;
  const float *const *rpe;    /* HOST array of B device pointers to RPE['scene'] [5,n,n], or NULL
                                 to have it computed in-kernel from ctrs/vecs (utils.py:193-212)  */
  const float *tgt_nodes;     /* [B,10,16] TGT_NODES                                               */
  const float *tgt_rpe;       /* [B,20]    TGT_RPE                                                 */
} mind_scene_batch;

/* Outputs of ScenePredNet.forward (network.py:545-556): res_cls, res_reg, res_aux[0]. */
typedef struct {
  float *cls;        /* [B,6]          softmax mode probabilities                                  */
  float *reg;        /* [A,6,60,5]     (x, y, exp sx, exp sy, exp rho)  agent-local frame          */
  float *vel;        /* [A,6,60,2]                                                                 */
  float *lane_feat;  /* optional [L,128]: LaneNet output written back for reuse (may be NULL)      */
  float *actor_emb;  /* optional debug taps [A,128] fused actor tokens, may be NULL                */
  float *cls_emb;    /* optional [B,128] fused cls token, may be NULL                              */
} mind_pred_out;

/* replaces ScenePredNet.forward over the whole AIME batch (scenario_tree.py:69-71). */
int mind_predict_batch(mind_ctx *ctx, const mind_scene_batch *in, mind_pred_out *out);

/* Per-kernel timing of the last mind_predict_batch measured with HIP events on the context stream:
 * returns the number of fusion pair-kernel launches and their summed milliseconds. */
int mind_last_fusion_stats(mind_ctx *ctx, int *n_launches, float *total_ms, double *pairs_processed);
/* Duration of the tree-iLQR kernel of the last mind_ilqr_* call on this context (HIP events on the context stream; 0 unless
 * profiling is on), its number of cost trees and the workgroups per tree it ran with. */
int mind_last_ilqr_stats(mind_ctx *ctx, float *kernel_ms, int *n_trees, int *workgroups_per_tree);
/* Shader-clock cycles of the last tree-iLQR launch's critical cost tree (the one that bounds the launch), summed over its fits:
 * out9 = { nodes M, serial depth (levels), passes of the iteration loop, cycles of the derivative pass, the backward (Riccati) sweep,
 * the line search's state chain, its cost pass, the selection, cost trees in the launch }. */
int mind_last_ilqr_profile(mind_ctx *ctx, double *out9);
/* Per-iteration trace of one fit of the last mind_ilqr_* call on this context (iLQR.fit's loop, planners/ilqr/solver.py:133-158): row k
 * = reference iteration k = { mu the backward pass ran with, J of the nominal trajectory (the reference's J_opt when the line search
 * starts), index of the accepted step size among the ten candidates (-1: every candidate rejected, mu raised; -2: Q_uu singular, the
 * iteration is retried), J of the accepted candidate (J of the nominal trajectory otherwise) }.  tree: index in the call's tree array;
 * phase: 0, or 1 for the full fit of mind_ilqr_contingency.  Writes min(rows, cap_rows) rows of 4 doubles to out and the number of
 * iterations to *n_rows.  MIND_ESTATE when the last call holds no such fit.  Parity diagnostics: the tests compare this trace with the
 * reference's own, iteration by iteration. */
int mind_last_ilqr_trace(mind_ctx *ctx, int tree, int phase, double *out, int cap_rows, int *n_rows);
/* enable/disable event timing (off by default: events add a little latency) */
int mind_set_profiling(mind_ctx *ctx, int enable);

/* Arithmetic of the RelaFusionLayer pair contractions (planners/mind/networks/network.py:165-232; the reference runs them
 * in torch fp32) -- and of every other MFMA contraction of the predictor (ActorNet, the MFMA decoder / token kernels where enabled):
 *   MIND_PAIR_BF16X6 (default) = both operands split EXACTLY into three bf16 parts (hi + mid + lo = the 24 significand bits of the fp32
 *     value), the six partial products of weight >= 2^-24 on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: the reference's fp32
 *     arithmetic class at 2.7 x the fp32-MFMA rate (k_pair_t6; ActorNet: k_actor_mfma<6>); error against the fp32 oracle = the fp32 MFMA's;
 *   MIND_PAIR_F32 = plain fp32 operands on the fp32 MFMA (v_mfma_f32_16x16x4_f32; k_pair, k_actor_f32);
 *   MIND_PAIR_BF16X3 = operands split into bf16 hi + lo (16 significand bits), hi.hi + hi.lo + lo.hi: narrower than the reference's fp32 (1e-5 m
 *     from the oracle where the fp32 classes show 3e-6; meets the 2e-4 m parity bar), 1.5-1.8 x faster than bf16x6 on large scenes; opt-in;
 *   MIND_PAIR_BF16 = plain bf16 operands (BASELINE config 5's "bf16 MFMA attention": misses the 1e-3 m parity bar; opt-in).
 * Also settable at context creation through the environment, MIND_PAIR_PREC=f32|bf16x3|bf16|bf16x6. */
#define MIND_PAIR_F32 0
#define MIND_PAIR_BF16X3 1
#define MIND_PAIR_BF16 2
#define MIND_PAIR_BF16X6 3
int mind_set_pair_precision(mind_ctx *ctx, int mode);

/* Kernel-selection knobs (A/B measurements and tests; the defaults are the measured winners, the results are the same within the
 * arithmetic's accuracy): "dec_mfma_min" (agents per call from which the decoder's actor part uses the MFMA kernel; default: never),
 * "ilqr_wgs" (workgroups per wide cost tree, default 16, halved until the launch is resident; 1 = the one-workgroup kernel), "ilqr_multi_min" (node count from which a
 * tree is "wide", default 192; a wide-tree launch that is not fully resident -- another context holds CUs -- aborts at its first
 * barrier and the call is solved again by the one-workgroup kernel: mind_last_ilqr_stats then reports 1 workgroup per tree),
 * "ilqr_wgs_big" / "ilqr_big_min" (workgroups per tree of at least ilqr_big_min nodes: 32 from 12 288 nodes),
 * "ilqr_slots" (narrow cost trees: workgroups per tree that take the fit's Levenberg-Marquardt slots -- the master evaluates slot 0 of every
 * pass, a follower the value the schedule reaches after its number of rejections; default 10, at most 12, 1 = everything in one workgroup; followers
 * that are not resident are simply not used; same bits for every value; MIND_ILQR_SLOTS),
 * "ilqr_spec_deriv" (1, default: with slots on follower workgroups one more workgroup per tree runs the derivative pass of a pass's first
 * candidate beside the master's pricing of the candidates, and the master swaps derivative sets instead of differentiating when that
 * candidate is the accepted one; same bits; mind_last_ilqr_stats counts it among the workgroups per tree; MIND_ILQR_SPEC_DERIV),
 * copies off the planning cycle's critical path (all of them leave every result bit for bit): "upload_kernel_max" (uploads of at most this many
 * bytes from the context's page-locked staging -- the root scene, the solver's tables -- run as a kernel on the consumer's queue that reads the
 * host buffer, instead of a copy the SDMA engine hands over; default 1 MB, 0 = always copies), "ilqr_host_out_max" (tree-iLQR launches of at
 * most this many nodes write xs / us / statistics to the host staging themselves; default 4 096), "dec_mirror" (mind_aime_plan, unsharded: the
 * round's last glue kernel writes the decisions where the host reads them), "tab_host_max" / "tab_small" (its small index tables are read from
 * the host staging / travel in the kernel arguments), "early_eval" (mind_loop prices a candidate tree as soon as the pending launch marks it
 * complete), "glue_fused" (mind_aime_plan's pruning decisions and branch-time bits from one launch, k_aime_select_branch); MIND_GLUE_FUSED, MIND_UPLOAD_KERNEL_MAX, MIND_ILQR_HOST_OUT_MAX, MIND_DEC_MIRROR, MIND_TAB_HOST_MAX, MIND_TAB_SMALL, MIND_EARLY_EVAL,
 * "actor_f32" (0: the fp32 VALU ActorNet instead of the fp32-MFMA one under the exact-fp32 setting), "actor_f32_min" / "actor_f32_pair_min" (actors per
 * call from which the fp32-MFMA ActorNet runs with two actors per workgroup, under every setting / under exact fp32; default never), "enc_mfma" (0: the fp32 VALU ActorNet / decoder kernels under every precision), "actor_split" (6: three-way operand split,
 * fp32-class; 3: two-way), "xcd_order" (XCD-aware job order of the pair kernel), "tok_mfma" (1: the per-token epilogue / prologue of the fusion layers on the fp32 MFMA kernel k_token_mfma instead of the fp32 VALU one; off by default: measured slower), "tok_small_max" (batches of at most this many tokens run k_token with four tokens per workgroup instead of eight; same bits), "tgt_side" (0: the context stream waits for the target embedding before the fusion layers), "dec_overlap" (0: the decoder's actor part as one kernel behind
 * k_dec_scene instead of its actor_proj half beside it on the side stream; bit-identical).  Environment: MIND_DEC_MFMA_MIN, MIND_ENC_MFMA,
 * MIND_ACTOR_SPLIT, MIND_XCD_ORDER, MIND_ILQR_WGS, MIND_DEC_OVERLAP at context creation. */
int mind_set_tuning(mind_ctx *ctx, const char *name, int value);
int mind_get_pair_precision(mind_ctx *ctx);   /* -> mode, or MIND_EINVAL */
/* host-only helper (tests): the bf16 hi / lo MFMA fragment packing of one [128][row_stride] weight matrix,
 * out[16384] dwords = [part 2][out block 8][k group 4][lane 64][4] (see pair_bf16_kernels.hip).  Needs no GPU. */
int mind_debug_pack_bfrag(const float *w, int row_stride, uint32_t *out);

/* host-only helper (tests): the pair kernels' job schedule (mind_amd/csrc/pair_jobs.h) for a batch of n_scenes scenes of scene_tokens[b]
 * tokens on a device of n_cu compute units -- exactly what mind_predict_batch builds.  last_layer != 0: the list of the last fusion layer
 * (actor + cls columns only; scene_actors[b] actors per scene).  out_jobs receives up to cap records of six ints
 * {scene, column, first tile, one past the last tile, partial slot, wave slot = wave * grid + workgroup} in list order without the empty
 * padding jobs; out_info[4] = {jobs per column of scene 0, grid, list length with padding, number of jobs}.  Returns the number of jobs or a
 * negative error.  Needs no GPU. */
int mind_debug_pair_schedule(const int *scene_tokens, const int *scene_actors, int n_scenes, int n_cu, int last_layer, int *out_jobs, int cap,
                             int *out_info);

/* host-only helper (tests): the bf16 hi / mid / lo MFMA A-operand packing of one Conv1d weight [co][ci][ksz] (torch layout) for the
 * ActorNet GEMMs of actor_mfma_kernels.hip: [co/16][k-step][part 3 = hi, mid, lo][lane 64][4] dwords, GEMM index k = tap * ci_pad + ci
 * (ci_pad = ci rounded up to a power of two >= 16).  Returns the number of dwords written (<= cap) or a negative error. */
int mind_debug_pack_conv_frag(const float *w, int co, int ci, int ksz, uint32_t *out, size_t cap);

/* Debug tap (tests): the tree-iLQR kernel's sin / cos / tan (mind_amd/csrc/mind_trig.h, the routine oracle/ilqr_ref.c shares) of n host
 * doubles, evaluated on the device one wave of 64 arguments at a time -- out[4 n] = sin, cos, tan, the cosine that comes with the
 * tangent.  The oracle's oracle_sincos / oracle_tan_cos must give the same bits (tests/test_gpu_ilqr.py). */
int mind_debug_trig(mind_ctx *ctx, const double *x, int n, double *out);

/* Debug taps used by the parity tests only: run just the first n (0..6) fusion layers on the next
 * mind_predict_batch calls, and read internal device buffers ("x", "ST", "QK", "edge", "part",
 * "actor_feat", "tokpos") back to the host.  mind_debug_read returns the number of floats copied (or
 * the buffer's size when host == NULL) or a negative error. */
int mind_debug_set_layers(mind_ctx *ctx, int n);
int64_t mind_debug_read(mind_ctx *ctx, const char *name, float *host, int64_t max_floats);

/* ------------------------------------------------------------------------------------------------
 * Tree-iLQR contingency solves: replaces TrajectoryTreeOptimizer.init_warm_start_cost_tree /
 * warm_start_solve / init_cost_tree / solve (planners/mind/trajectory_tree.py:19-147) together with
 * iLQR.fit (planners/ilqr/solver.py:80-167), TreeCost (ilqr/cost.py:326-446) and the potentials
 * (ilqr/potential.py).  One cost tree per scenario tree; all trees of a plan are solved in one call.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int n_nodes;              /* M trajectory nodes, keys 0..M-1 in creation order (LIFO DFS, Q13)  */
  const int32_t *parent;    /* HOST [M] parent key, -1 for node 0 (child of the x0 root)          */
  const float *prob;        /* HOST [M] scenario-node probability (fp32 as in the reference)      */
  int n_agents;             /* a (agent 0 = ego)                                                  */
  const float *agent_mean;  /* HOST [M, a, 2] predicted mean of every agent at the node's step    */
  const float *agent_cov;   /* HOST [M, a]    max-sigma                                           */
  /* generic mode only (mind_ilqr_solve_fields / mind_cost_eval with a grid), NULL otherwise:        */
  const double *field;      /* HOST [M, H, W] cost_field of each node's PotentialField             */
  const double *node_w;     /* HOST [M, 32]  per node: diag w_des[6], diag w_con[6], lower[6],     */
                            /*   upper[6], diag w_ctrl[2], des_state[6] (StatePotential,           */
                            /*   StateConstraint, ControlPotential of ilqr/potential.py:4-59)      */
} mind_cost_tree;

typedef struct {
  double dt, wheelbase;                 /* 0.2, 2.5 (trajectory_tree.py:15)                        */
  double w_des_state[6];                /* diag of w_des_state                                     */
  double w_state_con[6];                /* diag of w_state_con                                     */
  double state_lower[6], state_upper[6];
  double w_ctrl[2];                     /* diag of w_ctrl                                          */
  double w_tgt, w_ego, w_ego_cov_offset, w_exo, w_exo_cov_offset, w_exo_cost_offset;
  double grid_res; int grid_w, grid_h;  /* 0.4, 256, 256                                           */
  int max_iter;                         /* 100                                                     */
} mind_ilqr_cfg;

typedef struct { int iterations; int converged; double J; double mu; } mind_ilqr_stats;

/* x0[6] = (x,y,v,yaw,a,delta); target_lane HOST [P,2] float64 (gt_tgt_lane as float64);
 * us_init HOST [sum M, 2] (NULL = zeros); use_exo = 0 builds the warm-start tree (lane term only).
 * Outputs HOST xs [sum M, 6], us [sum M, 2], stats [n_trees]. */
int mind_ilqr_solve_trees(mind_ctx *ctx, const mind_ilqr_cfg *cfg, const mind_cost_tree *trees,
                          int n_trees, const double *x0, const double *target_lane, int n_lane_pts,
                          double target_vel, int use_exo, const double *us_init, double *xs,
                          double *us, mind_ilqr_stats *stats);

/* MINDPlanner.get_traj_tree (planner.py:174-178) for every scenario tree of a plan in ONE launch: the warm-start
 * fit (init_warm_start_cost_tree + warm_start_solve: lane term only, cfg_warm = w_opt_cfg, zero initial controls)
 * followed, from its controls, by the full fit (init_cost_tree + solve, cfg_full = opt_cfg).  xs / us are the
 * full fit's; both configurations must share dt, wheelbase and the grid. */
int mind_ilqr_contingency(mind_ctx *ctx, const mind_ilqr_cfg *cfg_warm, const mind_ilqr_cfg *cfg_full,
                          const mind_cost_tree *trees, int n_trees, const double *x0,
                          const double *target_lane, int n_lane_pts, double target_vel, double *xs,
                          double *us, mind_ilqr_stats *stats_warm, mind_ilqr_stats *stats_full);
/* The same call in two halves, for a caller that has host work of its own while the kernel runs (the planner builds the scenario
 * trees' Python objects meanwhile): _begin uploads, launches and queues the read-backs on the context stream and returns without
 * waiting (trees / x0 / target_lane are consumed before it returns); mind_ilqr_finish waits and writes xs / us / stats_* -- those
 * arrays must stay valid until then.  One call can be pending per context; another mind_ilqr_* call in between returns MIND_ESTATE,
 * as does mind_ilqr_finish with nothing pending. */
int mind_ilqr_contingency_begin(mind_ctx *ctx, const mind_ilqr_cfg *cfg_warm, const mind_ilqr_cfg *cfg_full,
                                const mind_cost_tree *trees, int n_trees, const double *x0,
                                const double *target_lane, int n_lane_pts, double target_vel, double *xs,
                                double *us, mind_ilqr_stats *stats_warm, mind_ilqr_stats *stats_full);
/* ... on the cost trees the last mind_aime_plan of this context flattened (mind_aime_plan_out.flat_*: n_trees trees of tree_off[t+1] -
 * tree_off[t] nodes): the tree arrays stay inside the library.  xs / us: [tree_off[n_trees], 6 / 2], stats_*: [n_trees]. */
int mind_ilqr_contingency_begin_plan(mind_ctx *ctx, const mind_ilqr_cfg *cfg_warm, const mind_ilqr_cfg *cfg_full, const double *x0,
                                     const double *target_lane, int n_lane_pts, double target_vel, double *xs, double *us,
                                     mind_ilqr_stats *stats_warm, mind_ilqr_stats *stats_full);
int mind_ilqr_finish(mind_ctx *ctx);

/* ------------------------------------------------------------------------------------------------
 * AIME glue (k7): the arithmetic of ScenarioTreeGenerator.prune_merge (planners/mind/scenario_tree.py:
 * 281-412) that touches every (agent, mode, step) -- world-frame positions / velocities / headings,
 * max-sigma covariances and the topology signatures -- for all scenes of one AIME round, reading the
 * predictor outputs where they already are (device), and optionally the pruning decisions themselves.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int n_scenes;
  const int32_t *actor_off;   /* HOST [n+1]                                                        */
  const float *reg, *vel;     /* DEVICE [A,6,60,5], [A,6,60,2]: mind_pred_out.reg / .vel            */
  const float *actor_ctrs, *actor_vecs; /* DEVICE [A,2] agent frames (as in mind_scene_batch)      */
  const float *rot, *orig;    /* HOST [n,2,2], [n,2]: scene frames ROT / ORIG                       */
  const float *cov_last;      /* HOST [A] last history covariance TRAJS_COV_HIST[:, -1, 0]          */
  const int32_t *last;        /* HOST [n] index of the last predicted step kept by seq_len (or < 0) */
  const float *target_lane;   /* HOST [n_lane_pts,2] target lane polyline (float32), or NULL            */
  int n_lane_pts;
  /* optional: the pruning decisions on the device too (needs mind_world_out.sel / sel_prob) */
  const float *cls;           /* DEVICE [n,6] mode probabilities (mind_pred_out.cls), or NULL                */
  const float *scen_prob;     /* HOST [n] SCEN_PROB of every scene                                          */
  int lane_check;             /* 1: drop modes whose ego end point is farther than dist_thres (+ sigma) from the target
                                 lane (needs target_lane and last >= 0 for every scene)                     */
  float dist_thres;           /* tar_dist_thres                                                             */
} mind_world_in;

typedef struct {
  float *world;               /* DEVICE [A,6,60,6]: x, y, vx, vy, heading, max-sigma (world frame)  */
  float *topo;                /* DEVICE [A,6] winding of (agent - ego of its scene); ego rows = 0   */
  float *ego_end;             /* DEVICE [n,6,4]: ego x, y, max-sigma at step `last` and the distance */
                              /*   of that point to the target lane (if last >= 0; inf without lane) */
  float *sel, *sel_prob;      /* DEVICE [n,6] or NULL: kept modes of every scene in visiting order (mode index as a    */
                              /*   float, -1 = none) and their path probabilities: probability floor, target-lane      */
                              /*   pruning and the greedy topology merge of prune_merge (:293-327, 361-395)            */
} mind_world_out;

/* asynchronous on the context stream (read the outputs after mind_ctx_synchronize or a stream-ordered copy) */
int mind_aime_world(mind_ctx *ctx, const mind_world_in *in, const mind_world_out *out);

/* Re-basing of the observation at a branch point for all branching nodes of a round: update_obser
 * (scenario_tree.py:467-567) = get_origin_rotation / normalisation into the AV and agent frames
 * (utils.py:180-190, scenario_tree.py:128-158), actor features (utils.py:113-134), get_new_lane_graph
 * (utils.py:171-177), get_high_level_command (scenario_tree.py:613-652) and the target RPE (utils.py:193-242).
 * The outputs are the next round's mind_scene_batch tensors, written on the device. */
typedef struct {
  int n_scenes, n_agents, n_lanes;  /* S child scenes sharing one agent set (a) and one lane graph (l)   */
  const float *pos, *ang, *vel;     /* HOST [S,a,50,2], [S,a,50], [S,a,50,2]: last 50 world-frame steps    */
  const float *types;               /* HOST [a,50,7] TRAJS_TYPE                                            */
  const float *pad;                 /* HOST [S,a,50] PAD_OBS or NULL (= ones, as update_obser sets it)     */
  const float *lane_ctrs, *lane_vecs; /* HOST [l,2] lane_graph["lane_ctrs"/"lane_vecs"]                    */
  const float *target_lane;         /* HOST [P,2]                                                          */
  const float *target_lane_info;    /* HOST [P,12]                                                         */
  int n_lane_pts;                   /* P >= 12                                                             */
  float time_ahead, min_vel;        /* tar_time_ahead (5.0), 0.5                                           */
  /* Optional: the windows assembled ON THE DEVICE instead of uploaded (pos / ang / vel may then be NULL).  A child's window is
   * the last 50 steps of [its parent's window | its own first `dur` predicted steps] (scenario_tree.py:396-412, 470-473): the
   * parent's window is taken from the arena of the PREVIOUS mind_aime_rebase call on this context (its generation must be
   * prev_gen), the child's steps from `rows` (the [R,60,6] world-frame rows mind_aime_world's caller gathered for the kept modes:
   * x, y, vx, vy, heading, max-sigma). */
  const float *rows_dev;            /* DEVICE [R,60,6] or NULL (= host windows above)                      */
  const int32_t *parent_slot;       /* HOST [S] scene index of the parent in the previous call             */
  const int32_t *row0;              /* HOST [S] first row (agent 0) of the child in rows_dev                */
  const int32_t *dur;               /* HOST [S] number of predicted steps kept (END_T - CUR_T), 0..60       */
  int prev_gen;                     /* generation of the call that re-based the parents                     */
} mind_rebase_in;

typedef struct {
  float *actors;                    /* DEVICE [S*a,14,48]                                                  */
  float *actor_ctrs, *actor_vecs;   /* DEVICE [S*a,2]                                                      */
  float *lane_ctrs, *lane_vecs;     /* DEVICE [S*l,2]                                                      */
  float *tgt_nodes, *tgt_rpe;       /* DEVICE [S,10,16], [S,20]                                            */
  float *frames;                    /* DEVICE [S,28]: ROT (4, row-major), ORIG (2), TGT_PTS (11 x 2)       */
  int32_t *gen;                     /* HOST, optional: receives this call's generation (for the children's prev_gen) */
} mind_rebase_out;

int mind_aime_rebase(mind_ctx *ctx, const mind_rebase_in *in, const mind_rebase_out *out);

/* ------------------------------------------------------------------------------------------------
 * ScenarioTreeGenerator.branch_aime (planners/mind/scenario_tree.py:38-58) in ONE call: every AIME round -- the batched scene
 * prediction (:69-71), prune_merge (:281-412), create_nodes (:73-80), decide_branch / get_branch_time (:82-100, 592-611) and
 * update_obser (:467-567) of the branching nodes -- runs on the device with the per-round bookkeeping in native code; the host
 * sees one small read-back per round (kept modes + branch-time bits).  The caller supplies what process_data (:122-206) and
 * prepare_root_data (:414-465) produce for the root and receives the internal tree's nodes plus, for the nodes of finished
 * branches, the rows get_scenario_tree (:208-272) attaches.  All pointers HOST, float32.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int n_agents, n_lanes, n_lane_pts;
  const float *actors;                       /* [a,14,48]  ACTORS of the root scene                                       */
  const float *actor_ctrs, *actor_vecs;      /* [a,2]      TRAJS_CTRS / TRAJS_VECS                                        */
  const float *lanes;                        /* [l,10,16]  LANES (instance-frame lane features)                           */
  const float *lane_ctrs, *lane_vecs;        /* [l,2]      lane_graph anchors (AV frame; update_obser re-expresses them)  */
  const float *tgt_nodes, *tgt_rpe;          /* [10,16], [20]                                                             */
  const float *rot, *orig, *tgt_pts;         /* [2,2], [2], [11,2]: ROT, ORIG, TGT_PTS of the root                        */
  const float *hist;                         /* [a,50,6]   root world-frame history: x, y, vx, vy, heading, max-sigma     */
  const float *types;                        /* [a,50,7]   TRAJS_TYPE                                                     */
  const float *target_lane;                  /* [P,2]                                                                     */
  const float *target_lane_info;             /* [P,12]                                                                    */
  float time_ahead, min_vel, dist_thres;     /* tar_time_ahead, 0.5, tar_dist_thres                                       */
  int max_depth;                             /* ScenTreeCfg.max_depth                                                     */
  int max_rounds;                            /* safety bound on the number of rounds (>= max_depth + 1)                   */
  int pred_len;                              /* planning horizon in predicted steps (the generator's pred_len, 50: END_T of  */
                                             /*   a fresh observation; seq_len = 50 + pred_len), <= 60                     */
  /* Root scene built ON THE DEVICE (process_data + prepare_root_data as kernels): when raw_pos != NULL the fields actors ... hist
   * above are ignored (may be NULL) and the library computes them from what get_agent_trajectories (utils.py:245-342) returns and
   * the map's resampled lane polylines: normalisation into the AV / agent frames and actor features (k_aime_rebase on the raw
   * windows), lane graph + LANES features (k_aime_root_lanes, float64 as update_lane_graph_from_argo), high-level command, the
   * root's world-frame histories as prepare_root_data reconstructs them (k_aime_root_hist). */
  const float *raw_pos, *raw_ang, *raw_vel;  /* [a,50,2], [a,50], [a,50,2] world-frame padded histories, AV first            */
  const float *raw_pad;                      /* [a,50] observed flags (PAD_OBS)                                               */
  const double *lane_pts;                    /* [l,11,2] world-frame points of every 15 m lane piece (10 sub-segments)        */
  const int32_t *lane_flags;                 /* [l,6]: lane type (0 vehicle, 1 bike, 2 bus), intersection, left / right mark   */
                                             /*   class (0 crossable, 1 not, 2 other), has left / right neighbour              */
  float travel0;                             /* max(ego speed, min_vel) * time_ahead (float32), scenario_tree.py:131,620-624   */
  /* Test / benchmark hook (mind_amd/synth.py ScriptedBranching, ScriptedFullTree: the reference ships no trained checkpoint and the
   * formula weights collapse every plan to a node or two, so the full-tree workloads of BASELINE configs 4 / 5 script the modes): when
   * script_cls != NULL the predictor still runs on every scene of every round, and these DEVICE arrays then take the place of its
   * outputs for every scene: cls [6], reg [a,6,60,5], vel [a,6,60,2]. */
  const float *script_cls, *script_reg, *script_vel;
  /* path-probability floor of prune_merge (scenario_tree.py:368-370).  0 = the reference's 0.001; the scripted stress workload lifts it
   * (a small positive value) so that a full 6-ary depth-5 tree can grow: 6^-4 is already below 0.001. */
  float prob_floor;
  /* Optional: begin the contingency solves of the plan (planner.py:174-178: warm-start fit + full fit of every scenario tree) straight
   * behind it, before the call returns -- everything they need besides the plan's own cost trees is known when the plan starts (ego
   * state x0 [6], target lane [solve_n_lane_pts, 2] float64, target velocity, the two configurations).  solve_cfg_full != NULL asks
   * for it; out->solves_begun tells whether it happened (not on a sharded context, not without a finished branch), and
   * mind_ilqr_finish_plan collects.  Saves the host round trip between the plan's last kernel and k_ilqr. */
  const mind_ilqr_cfg *solve_cfg_warm, *solve_cfg_full;
  const double *solve_x0, *solve_lane;
  int solve_n_lane_pts;
  double solve_target_vel;
} mind_aime_plan_in;

#define MIND_AIME_BRANCH 1
#define MIND_AIME_END 2
#define MIND_AIME_TERMINATE 4
typedef struct {
  int round, scene, mode;      /* SCEN_ID = "{round}_{scene}_{mode}" (scenario_tree.py:321)                               */
  int parent;                  /* index of the parent node in this table, -1 = child of the root                          */
  float prob;                  /* SCEN_PROB (path probability, float32)                                                   */
  int cur_t, end_t;            /* CUR_T, END_T after the branching decisions                                              */
  int flags;                   /* MIND_AIME_BRANCH | _END | _TERMINATE of the internal tree node                          */
  int dur;                     /* steps of `rows` (END_T - CUR_T), 0 when the node is not on a finished branch            */
  int64_t row_off;             /* offset in floats of its rows [a, dur, 3] = (x, y, max-sigma) in out->rows, or -1        */
  float tgt_pts[22];           /* TGT_PTS [11,2] of the scene the node was predicted from                                 */
} mind_aime_node;

typedef struct {
  const mind_aime_node *nodes; /* library-owned host table [n_nodes] in creation order (= the internal tree's insertion order     */
  int n_nodes;                 /*   without its root), valid until the next mind_aime_plan call on this context                 */
  const float *rows;           /* library-owned host buffer [n_row_floats], same lifetime                                        */
  int64_t n_row_floats;
  int n_expanded, n_rounds;    /* scenes pushed through the predictor, rounds                                             */
  int root_flags;              /* flags of the internal root node                                                         */
  int round_scenes[32];        /* scenes of every round (for the caller's throughput accounting)                          */
  float pair_ms;               /* summed duration of the pair-kernel launches (HIP events; 0 unless profiling is on)      */
  int pair_launches;
  /* The cost trees TrajectoryTreeOptimizer builds from the returned scenario trees (trajectory_tree.py:19-124: LIFO depth-first
   * order, every even step of a node's window = one trajectory node), ready for mind_ilqr_contingency: one per root child on a
   * finished branch, in the order get_scenario_tree returns them.  Library-owned, same lifetime as `nodes`. */
  int n_trees;                 /* scenario trees                                                                          */
  const int32_t *tree_top;     /* [n_trees] index in `nodes` of every tree's top node                                     */
  const int32_t *tree_off;     /* [n_trees+1] trajectory-node offsets of the trees in the arrays below                    */
  const int32_t *flat_parent;  /* [M_total] parent key inside its tree, -1 for a tree's node 0                            */
  const float *flat_prob;      /* [M_total] sibling-normalised scenario probability (float32 arithmetic)                   */
  const float *flat_mean;      /* [M_total, a, 2]                                                                         */
  const float *flat_cov;       /* [M_total, a]                                                                            */
  int solves_begun;            /* the contingency solves were begun behind the plan (in->solve_cfg_full): mind_ilqr_finish_plan   */
} mind_aime_plan_out;

/* ------------------------------------------------------------------------------------------------
 * Multi-GPU: one process (one mind_ctx) per GPU plans the SAME scene; the scenes of every AIME round of mind_aime_plan are
 * block-distributed over the ranks.  (The reference has no distributed code; the independence this rests on: the predictor loops
 * per scene, planners/mind/networks/network.py:318,497, and prune_merge works scene by scene, scenario_tree.py:281-412.)
 * Per round: every rank runs predictor + pruning + branch-time test on its block -> ONE all-gather of the decisions (96 B per
 * scene) -> every rank replays create_nodes / decide_branch on the same table (the tree is replicated, 0.2 ms at 1 555 nodes) ->
 * the rank that holds a branching node's parent scene re-bases it (update_obser, on the device) -> ONE small all-gather of the
 * children's frames (28 floats per scene: the replicated tree's node records; + LaneNet's output after the root round) and ONE
 * all-to-all that moves a re-based scene (predictor inputs + history windows, ~240 KB at 64 agents) from the rank that re-based it to
 * the rank whose block of the next round holds it -- on a full tree the two are the same rank except at the block boundaries, so only
 * the boundary scenes travel (skipped by every rank when nothing does).  At the end every rank packs the rows / cost tree entries of
 * the nodes whose predicted rows it holds and ONE all-reduce (sum over zero-filled buffers) completes them everywhere.  With
 * world == 1 the same code runs without exchanges.
 *
 * The transport is the caller's: a function that performs a collective on DEVICE buffers of this context's device,
 *   MIND_XCHG_ALLGATHER:  recv [world][bytes] <- every rank's send [bytes], in rank order;
 *   MIND_XCHG_ALLREDUCE:  recv [bytes / 4 words] <- sum over the ranks of send (send may equal recv).  Every word is non-zero on at most
 *                         one rank (its owner), so the transport should sum 32-bit INTEGERS: the owner's bits then arrive unchanged
 *                         (a float sum would turn an owner's -0.0 into +0.0).
 *   MIND_XCHG_ALLTOALLV:  the last argument is the ADDRESS of a host table int64 [2][world]: bytes this rank sends to / receives from every
 *                         rank; send / recv hold the per-rank segments packed in rank order (torch.distributed.all_to_all_single with split
 *                         sizes).
 * It is called with everything the library queued on the context's stream complete, and must return with the result complete
 * (mind_amd/parallel.py: torch.distributed -- RCCL over xGMI on a node, gloo in the tests).  world <= 1 or fn == NULL switches
 * the sharding off; force != 0 runs the exchanges of a one-rank group too (tests: RCCL on a one-GPU box). */
#define MIND_XCHG_ALLGATHER 0
#define MIND_XCHG_ALLREDUCE 1
#define MIND_XCHG_ALLTOALLV 2
typedef int (*mind_exchange_fn)(void *user, int op, void *send, void *recv, int64_t bytes);
int mind_set_exchange(mind_ctx *ctx, int rank, int world, mind_exchange_fn fn, void *user, int force);
/* collectives run and bytes received by this context's sharded plans so far */
int mind_last_exchange_stats(mind_ctx *ctx, long long *collectives, long long *bytes);

/* Returns MIND_ESTATE ("unsupported: ...") for the situations only the round-by-round host path handles -- a node re-expanded in a
 * later round than the one that created it, no finished branch (the host path raises the reference's assertion), more than
 * max_rounds rounds -- in which case the caller runs that path instead. */
int mind_aime_plan(mind_ctx *ctx, const mind_aime_plan_in *in, mind_aime_plan_out *out);
/* mind_aime_plan in two halves, for a host thread that plans several scenes (one context each): _begin copies *in (the arrays it
 * points to must stay valid until _finish) and runs the plan on a thread of the library; _poll returns 1 while it runs, 0 once
 * _finish will not block; _finish returns what mind_aime_plan would have.  No other call on the context between _begin and _finish
 * (a second _begin while the plan runs returns MIND_ESTATE; a finished plan that was never collected is dropped by the next _begin).
 * mind_ctx_busy: 1 while work queued on the context's stream has not completed (e.g. the contingency solves begun with
 * mind_ilqr_contingency_begin), 0 once collecting it will not block. */
/* collects the contingency solves a plan began itself (out->solves_begun): waits for them and copies the results of all its
 * scenario trees, concatenated in tree order: xs [n_nodes, 6], us [n_nodes, 2], the warm-start and full fits' statistics [n_trees].
 * n_nodes / n_trees = out->tree_off[out->n_trees] / out->n_trees of the plan the caller collects for: MIND_EINVAL (nothing copied) when
 * the pending solves belong to a plan of another shape, MIND_ESTATE when none is pending.  Solves that are never collected are
 * drained and dropped by the context's next mind_aime_plan. */
int mind_ilqr_finish_plan(mind_ctx *ctx, int n_nodes, int n_trees, double *xs, double *us, mind_ilqr_stats *stats_warm,
                          mind_ilqr_stats *stats_full);
int mind_aime_plan_begin(mind_ctx *ctx, const mind_aime_plan_in *in);
int mind_aime_plan_poll(mind_ctx *ctx);
int mind_aime_plan_finish(mind_ctx *ctx, mind_aime_plan_out *out);
int mind_ctx_busy(mind_ctx *ctx);

/* The array part of get_agent_trajectories (planners/mind/utils.py:245-342): raw [a,T,6] float64 rows (observed flag, x, y, heading, vx,
 * vy; T = 50) of the kept tracks, slot [a] = the type one-hot position of every track -> pos [a,T,2], ang [a,T], vel [a,T,2] float32 with the
 * positions / headings of unobserved steps taken from the nearest earlier observed step (the first observed one before it), velocities zero
 * there, typ [a,T,7] int16 one-hot on observed steps, have [a,T] int16.  Host arithmetic (copies and casts), no context needed. */
int mind_fill_tracks(const double *raw, int a, int T, const int32_t *slot, float *pos, float *ang, float *vel, int16_t *typ, int16_t *have);

/* MINDPlanner.evaluate_traj_tree (planners/mind/planner.py:180-198) for every candidate trajectory tree of a plan (host arithmetic,
 * float64, numpy's operation and summation order): states [sum counts, 6], ctrls [sum counts, 2] = the trees' nodes in key order, root
 * (x0, zero control) first; lane [P,2] float32 (lane_is_f32 != 0) or float64; out[n_trees] = mean node cost.  Needs no context. */
int mind_eval_traj_trees(const double *states, const double *ctrls, const int32_t *counts, int n_trees, const void *lane,
                         int lane_is_f32, int n_lane_pts, double target_vel, double *out);

/* ------------------------------------------------------------------------------------------------
 * The closed loop of one scene behind ONE call per simulator step (or per planning cycle): Simulator.run_sim (simulator.py:51-107),
 * CustomizedAgent.check_enable / check_trigger / step (agent.py:255-299), MINDAgent.observe / plan (agent.py:317-331), MINDPlanner.
 * update_observation / plan (planner.py:50-145), kine_propagate (common/kinematics.py:22-36).  A step = take-over check -> planner
 * trigger (every plan_step seconds) -> observation fan-out into the 50-frame windows of every track seen so far (unobserved tracks
 * repeat their last state with observed = false) -> when the agent is enabled: get_agent_trajectories (kept tracks, AV first) ->
 * mind_aime_plan with the device-built root and the plan-begun contingency solves -> mind_ilqr_finish_plan -> evaluate_traj_tree of
 * every candidate -> the reference's strict `<` scan -> first control -> ego plant.  Everything the interpreter did between the two
 * device waits of a cycle happens here; mind_amd/closed_loop.py ClosedLoopSim.step calls it once and reads the plan's objects only
 * when somebody asks for them (mind_loop_last_plan).  Exo agents are replayed from tables the caller tabulated once (one row per
 * simulator step, built with the driver's own observation code, so the windows hold the same float64 values).
 * Host arithmetic is float64 in the reference's operation order; sin / cos / tan are the C library's (the Python driver uses
 * math.sin / math.cos / math.tan for the same expressions).
 * ---------------------------------------------------------------------------------------------- */
typedef struct mind_loop mind_loop;
typedef struct {
  /* replayed scene, one row per simulator step n (sim_time = n additions of sim_step): tables are copied */
  int n_tracks;                 /* tracks of the world, track 0 = the closed-loop agent ("AV")                                  */
  int n_steps;                  /* rows of the tables below                                                                     */
  int clamp_last;               /* != 0: steps >= n_steps read row n_steps - 1 (a recording that holds its last frame)          */
  const double *ego_state;      /* [n_steps][4] world.agent_state(0, t_n) = (x, y, v, yaw): observation before the take-over and    */
                                /*   the state taken over at enable_time                                                        */
  int ego_state_is_f32;         /* world.agent_state returns float32 arrays (the reference's loader keeps tracks in float32,       */
                                /*   loader.py:166-168): numpy then evaluates the FIRST plant step after the take-over in float32   */
                                /*   (Python floats are weak operands, the state array is float64 from the second step on)          */
  const double *ego_obs;        /* [n_steps][5] ObjectState of the RECORDED ego at step n (x, y, heading, vx, vy), as the driver's     */
                                /*   to_object_state builds it: the ego's window entry before the take-over and at the take-over step  */
  const float *ego_trig32;      /* [n_steps][2] numpy's float32 cos / sin of the recorded yaw (the first plant step after a take-over  */
                                /*   from a float32 recording uses them); may be NULL when ego_state_is_f32 == 0                      */
  /* numpy's elementary functions where they are not the C library's: on AVX-512 hosts np.tan (float64) is a SIMD routine that differs
   * from libm's by an ulp in 0.5 % of the arguments, and the reference's plant (kinematics.py:22-36) calls np.tan.  NULL = the C library's
   * tan / sin / cos.  tan_fn is called once per control (the steer angle is constant between two plans), sincos_fn once per step. */
  double (*tan_fn)(double);
  void (*sincos_fn)(double, double *sin_cos);
  const double *exo_obs;        /* [n_steps][n_tracks][5] ObjectState of track i at step n: x, y, heading, vx, vy (row 0 unused)    */
  const uint8_t *exo_valid;     /* [n_steps][n_tracks] the track is reported at step n (simulator.py:60-63)                       */
  const int32_t *timestep;      /* [n_steps] ObjectState.timestep = int(round(t_n / 0.1))                                        */
  const int32_t *type_slot;     /* [n_tracks] one-hot slot of the track's object type (utils.py:245-342)                          */
  /* simulator + ego plant (simulator.py:51-107, agent.py:255-299, kinematics.py:22-36) */
  double sim_step, plan_step, enable_time;
  double wheelbase, max_speed, max_steer, max_acc, max_dec;
  /* planner constants of the scene (planner.py:147-171, scenario_tree.py:106-126): as mind_aime_plan_in with the device-built root */
  int n_lanes; const double *lane_pts; const int32_t *lane_flags;      /* [l,11,2], [l,6]                                       */
  int n_lane_pts; const float *target_lane, *target_lane_info;         /* resampled target lane [P,2], info [P,12]               */
  double time_ahead; float min_vel, dist_thres; int max_depth, max_rounds, pred_len; float prob_floor;
  /* contingency solves (planner.py:174-178) and candidate evaluation (planner.py:180-198) */
  const mind_ilqr_cfg *cfg_warm, *cfg_full;
  int solve_n_lane_pts; const double *solve_lane;                      /* gt_tgt_lane [P',2] float64                              */
  double target_vel;
  int eval_n_lane_pts, eval_lane_is_f32; const void *eval_lane;        /* lcl_smp.target_lane [P'',2] in its own dtype            */
  /* != 0: the speculative warm start of this repo's TrajectoryTreeOptimizer (trajectory_tree.py speculate_warm): the warm-start fits of
   * the previous plan's tree shapes run on a second context of the loop beside the AIME rounds; a tree whose shape recurs runs the full fit
   * only.  Same kernels on the same inputs: the results are the same bits either way; it pays when several loops share the device. */
  int speculative;
} mind_loop_desc;

/* running totals over every plan of the loop since it was created (a caller's accounting takes differences) */
typedef struct {
  long long plans, expansions, scen_trees, rounds;
  double aime_s, ilqr_s, total_s;   /* host wall time: the AIME call | collecting the solves | the whole planning cycle                */
  long long iterations, node_iterations, node_iterations_exo;      /* warm + full fits (TrajectoryTreeOptimizer.counters)              */
  long long warm_speculated, warm_hits;                             /* speculated warm-start fits begun / used                           */
  /* with profiling on (mind_set_profiling): kernel durations from HIP events on the context stream, as mind_aime_plan_out.pair_ms /
   * mind_last_ilqr_stats / mind_last_ilqr_profile report them per call */
  double pair_ms; long long pair_launches;
  double scene_n2, scene_n_a1;      /* sums over the expanded scenes of N^2 and N (a + 1), N = a + l + 1 tokens (algorithmic FLOPs / bytes) */
  double ilqr_ms; long long ilqr_launches, ilqr_trees; int ilqr_workgroups_per_tree;
  double ilqr_prof[9], ilqr_node_steps;
} mind_loop_totals;

typedef struct {
  int planned;                  /* the call computed at least one plan                                                          */
  int enabled;
  long long n_steps, n_plans;   /* since the loop was created (resets do not clear them)                                        */
  long long episode_steps;      /* simulator steps of the current episode (= table row of the next step)                          */
  double sim_time, last_trigger;    /* last_trigger < 0: none yet                                                               */
  double state[4], ctrl[2];     /* ego plant state (x, y, v, yaw) and the control in force                                      */
  /* the last plan */
  int n_agents, n_trees, best, n_expanded, n_rounds, n_traj_nodes;
  const double *costs;          /* [n_trees] library-owned, valid until the loop's next plan                                    */
  double aime_s, ilqr_s, total_s;
  mind_loop_totals tot;
} mind_loop_out;

/* the loop runs on `ctx` (weights loaded; not sharded); one loop per context at a time plans */
int mind_loop_create(mind_ctx *ctx, const mind_loop_desc *desc, mind_loop **out);
int mind_loop_destroy(mind_loop *loop);
/* ClosedLoopSim._start_episode: scene back to t = 0, windows cleared, agent disabled (n_steps / n_plans keep counting) */
int mind_loop_reset(mind_loop *loop);
/* Steps until `until_plans` more plans were computed (> 0), or sim_time >= until_time - 1e-9 (until_time >= 0: ClosedLoopSim.run_until),
 * or `max_steps` steps were taken -- whichever comes first; until_plans = 0, until_time < 0, max_steps = 1 is ClosedLoopSim.step.
 * MIND_ESTATE "unsupported: ..." = a plan only the round-by-round host path handles (see mind_aime_plan): the step's observation update has
 * happened, the plan and the rest of the step have not; the caller takes the loop over (mind_loop_export) and finishes the step on the host. */
int mind_loop_advance(mind_loop *loop, int until_plans, double until_time, long long max_steps, mind_loop_out *out);
/* the state of the loop without stepping */
int mind_loop_state(mind_loop *loop, mind_loop_out *out);
/* The last plan's tables for a caller that builds the reference's objects from them (scenario trees, trajectory trees): *plan as
 * mind_aime_plan returned it, the solves' results concatenated in tree order (xs [n_traj_nodes,6], us [n_traj_nodes,2], statistics
 * [n_trees]), the tracks of the plan's agents (index into the world's tracks, AV first) and their TRAJS_TYPE [a,50,7] float32.
 * Library-owned, valid until the next plan on the loop's context; MIND_ESTATE when another plan ran on the context since (a context may be
 * shared by several planners). */
int mind_loop_last_plan(mind_loop *loop, mind_aime_plan_out *plan, const double **xs, const double **us, const mind_ilqr_stats **stats_warm,
                        const mind_ilqr_stats **stats_full, const int32_t **agent_tracks, const float **types, double *x0);
/* Hands the loop over to a host driver (ClosedLoopSim falls back to its Python steps: a planner attribute changed under it, a plan the
 * library reports unsupported): the tracks seen so far in first-appearance order (track 0 first), their window lengths and rows
 * [n][50][7] = (observed, x, y, heading, vx, vy, timestep), oldest first, the first `count` rows of every window used.  cap = tracks the
 * arrays hold; the number of tracks comes back in *n (MIND_EINVAL with *n set when cap is too small). */
int mind_loop_export(mind_loop *loop, int cap, int *n, int32_t *track, int32_t *count, double *rows);

/* ------------------------------------------------------------------------------------------------
 * planners/ilqr call surface (iLQR.fit over a TreeCost of arbitrary PotentialField / StatePotential /
 * StateConstraint / ControlPotential objects; solver.py:80-167, cost.py:326-446, potential.py:62-264).
 * The grid is what PotentialField.__init__ receives: field_offset, resolution, xx[0,:], yy[:,0].
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int W, H;                 /* cost_field.shape = (H, W)                                          */
  double res;               /* resolution                                                         */
  double off_x, off_y;      /* field_offset                                                       */
  const double *gx, *gy;    /* HOST [W] = xx[0,:], [H] = yy[:,0]                                  */
} mind_field_grid;

/* gen_dist_field (ilqr/utils.py:5-22): grid of W x H centroids centred on ego_xy, distance of each to
 * the polyline `lane` [n_pts,2].  Outputs HOST offset[2] (field_offset), gx[W] (= xx[0,:]), gy[H]
 * (= yy[:,0]), dist[H,W]. */
int mind_lane_dist_field(mind_ctx *ctx, const double *ego_xy, const double *lane, int n_pts, int W, int H,
                         double res, double *offset, double *gx, double *gy, double *dist);

/* iLQR.fit on trees whose nodes carry materialised cost fields and their own quadratic potentials
 * (tree[i].field / .node_w set; .prob / agents ignored).  cfg supplies dt, wheelbase, max_iter only. */
int mind_ilqr_solve_fields(mind_ctx *ctx, const mind_ilqr_cfg *cfg, const mind_field_grid *grid,
                           const mind_cost_tree *trees, int n_trees, const double *x0,
                           const double *us_init, double *xs, double *us, mind_ilqr_stats *stats);

/* TreeCost.l / l_x / l_u / l_xx / l_uu (cost.py:341-446) of ONE tree at arbitrary points:
 * out HOST [n_query, 47] = { l, l_x[6], l_u[2], l_xx[36] row-major, diag l_uu[2] } for node[q] at
 * (x[q], u[q]).  grid == NULL: planner mode (fields from prob / agents / target lane as in
 * mind_ilqr_solve_trees); grid != NULL: generic mode (tree->field / node_w; lane arguments ignored). */
int mind_cost_eval(mind_ctx *ctx, const mind_ilqr_cfg *cfg, const mind_field_grid *grid,
                   const mind_cost_tree *tree, const double *x0, const double *target_lane,
                   int n_lane_pts, double target_vel, int use_exo, int n_query,
                   const int32_t *node, const double *x, const double *u, double *out);

#ifdef __cplusplus
}
#endif
#endif /* MIND_HIP_H */
