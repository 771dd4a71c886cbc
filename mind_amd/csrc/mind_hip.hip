// libmind_hip.so: C-ABI (include/mind_hip.h) over the hand-written gfx950 kernels.
// Host side: context, state_dict -> packed device blob, per-call job tables, kernel launches.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <atomic>
#include <functional>
#include <vector>

#include "../../include/mind_hip.h"
#include "encdec_kernels.hip"
#include "fusion_kernels.hip"
#include "pair_bf16_kernels.hip"
#include "pair_tile_kernels.hip"
#include "pair_tile6_kernels.hip"
#include "token_mfma_kernels.hip"
#include "actor_mfma_kernels.hip"
#include "actor_f32_kernels.hip"
#include "dec_mfma_kernels.hip"
#include "ilqr_kernels.hip"
#include "pair_jobs.h"
#include "aime_kernels.hip"

namespace {

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
};

struct WeightBlob {
  std::vector<float> host;
  std::map<std::string, size_t> off;
  size_t add(const std::string &key, const std::vector<float> &v) {
    while (host.size() % 64) host.push_back(0.f);  // 256-byte alignment
    size_t o = host.size();
    host.insert(host.end(), v.begin(), v.end());
    off[key] = o;
    return o;
  }
};

}  // namespace

// job / token tables of one batch shape (scene sizes), resident on the device
struct TableSet {
  std::vector<int> key;               // Bn, actor_off[0..Bn], lane_off[0..Bn]
  DevBuf meta, jobs, jobs5, rows;      // jobs5: the columns the last fusion layer runs (actors + cls), for k_pair_t
  std::vector<int> actor_row, cls_row, scene_n;
  long long edge_pairs = 0, edge_pairs_t = 0, stamp = 0;     // edge_pairs_t: with every column padded to whole 16-row tiles
  int ntok = 0, slot = 0, njobs = 0, njobs5 = 0;
  double pairs_full = 0, pairs_l5 = 0;
};
#define MIND_TABLE_SETS 8

struct mind_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // internal side stream: the lane encoders run beside the actor encoder; always fenced against `stream` with
  // events on both sides, so callers only ever see work ordered on `stream`
  hipStream_t side = nullptr;
  hipEvent_t ev_side = nullptr, ev_main = nullptr, ev_stage = nullptr, ev_tgt = nullptr, ev_cls = nullptr;
  std::vector<char> aime_stage;   // host staging of mind_aime_world's small tables (guarded by ev_stage)
  bool aime_stage_busy = false;
  std::string err;
  // weights
  float *wdev = nullptr;
  bool have_weights = false;
  WeightBlob blob;
  LaneW laneW;
  ActorW actorW;
  DmW decBW;     // actor part of the decoder, same packing (dec_mfma_kernels.hip)
  AmW actorBW;   // the same convolutions as bf16 hi / lo MFMA fragments (actor_mfma_kernels.hip)
  AfW actorFW;   // ... and as fp32 MFMA A fragments (actor_f32_kernels.hip)
  DecW decW;
  TokWeights tokW[7];  // [L]: epilogue of layer L-1 (L>=1) + prologue of layer L (L<=5); [0] = init
  TokWeightsM tokWM[7]; // the same matrices as fp32 MFMA A fragments (k_token_mfma<0>)
  TokWeightsM tokWB[7]; // ... and as bf16 hi / lo A fragments (k_token_mfma<1>: scenes of >= tok_bf_min_n tokens)
  // token kernel on the fp32 MFMA (k_token_mfma; MIND_TOK_MFMA=1 / mind_set_tuning("tok_mfma")).  Opt-in: measured on the MI355X it is
  // SLOWER than the VALU kernel -- 41 vs 30 us per launch at demo size (6 workgroups), 315 vs 282 us average on the full cfg4 tree
  // (profiles/r03o_*, r03p_*): 640 fp32 MFMAs of 32 cycles per wave and launch are 8.5 us by themselves and the weight stream (64 KB per
  // projection and workgroup) is the same; a bf16-split variant would cut the MFMA time, not the rest
  bool tok_mfma = false;
  bool tgt_side = true;         // the target polyline's encoder + embedding stay on the side stream through the fusion layers ("tgt_side")
  // scenes of at least this many tokens run k_token_mfma<1> (bf16 hi + lo split operands) under the bf16 arithmetics ("tok_bf_min_n";
  // 0 = never, the default).  Measured on the cfg4 full tree (N = 321, launches of up to 69 k tokens): 221 us per launch on average and
  // 0.80 ms for the largest, the same as the VALU kernel (which is LDS-bound there) -- with 97 KB of LDS the MFMA kernel keeps one
  // four-wave workgroup per CU and waits on its partial-sum and fragment loads instead (profiles/r03bb); opt-in until it is faster
  int tok_bf_min_n = 0;
  bool tok_merge = true;        // small token launches merge their independent projections (k_token_m; MIND_TOK_MERGE=0 / "tok_merge": the plain kernel)
  int tok_small_max = 2048;     // batches of at most this many tokens run k_token with 4 tokens per workgroup ("tok_small_max")
  const float *WAe[6], *WAp[6], *vtab[6], *rtab = nullptr;
  const u32 *WBe[6], *WBp[6];   // bf16 hi / lo fragments of the same matrices (pair_bf16_kernels.hip)
  const u32 *WLe[6], *WLp[6];   // the third part of the exact three-way split (hi + mid + lo; mid = the two-way split's lo): k_pair_t6
  // job / token tables of recent mind_predict_batch calls (least-recently-used of MIND_TABLE_SETS): a call whose scene sizes
  // were seen before rebuilds, uploads and synchronises nothing -- the closed loop's rounds recur every cycle, a full tree's
  // rounds (1 / 6 / 36 / 216 scenes) every plan
  TableSet tabs[MIND_TABLE_SETS];
  long long tab_clock = 0;
  long long n_table_hits = 0;
  int actor_np = 6;             // partial products per term of the MFMA ActorNet under bf16x3: 6 (three-way split, fp32-class) or 3 (MIND_ACTOR_SPLIT=3)
  // nodes per forward step of a narrow tree's line search ("ilqr_chunk"; 0: whole segments, the default).  Measured on the recorded
  // demo_1 loop: chunks of 6 / 8 / 12 nodes cost 2.10 / 2.05 / 2.04 ms per launch against 1.99 for whole segments (round 3: the cost
  // waves shared the SIMDs' float64 pipe with five state-chain waves); with the chains of a level packed into ONE wave and the cost
  // chunks kept off its SIMD (round 4) still 2.12 / 2.00 / 2.02 against 1.89: every extra forward step pays a rollout prologue (parent
  // state, first operands: two dependent round trips) and a barrier, more than the shorter cost tail saves (profiles/r04x_*)
  int ilqr_chunk = 0;
  // wide cost trees: workgroups per tree (halved until every workgroup of the launch is resident; cfg4 full tree, six trees per launch:
  // 8.33 / 7.40 / 7.37 / 7.63 ms per plan with 8 / 16 / 24 / 32, profiles/r03an), node count from which they are used (mind_set_tuning)
  int ilqr_wgs = 16, ilqr_multi_min = 192, ilqr_wgs_big = 32, ilqr_big_min = 12288;
  // narrow cost trees (below ilqr_multi_min nodes): workgroups per tree that take the fit's Levenberg-Marquardt slots (k_ilqr<GEN, 2>: a master +
  // ilqr_slots - 1 followers, one slot each; 1 = everything in one workgroup).  "ilqr_slots" / MIND_ILQR_SLOTS
  int ilqr_slots = 10;
  // ... and one more workgroup per tree that differentiates a pass's first candidate while the master prices the candidates (il_speculate; the
  // master swaps derivative sets instead of running its derivative pass when that candidate is the accepted one).  "ilqr_spec_deriv" / MIND_ILQR_SPEC_DERIV
  bool ilqr_spec_deriv = true;
  // uploads of at most this many bytes from the context's page-locked staging run as a kernel on the consumer's queue (pl_upload; 0 = always
  // hipMemcpyAsync).  "upload_kernel_max" / MIND_UPLOAD_KERNEL_MAX
  int upload_kernel_max = 1 << 20;
  // tree-iLQR launches of at most this many nodes in all write their results (xs, us, statistics) to the host staging themselves at the kernel's
  // end instead of two copies behind it (0 = always copies).  "ilqr_host_out_max" / MIND_ILQR_HOST_OUT_MAX
  long ilqr_host_out_max = 4096;
  // mind_aime_plan, unsharded: k_aime_branch writes a round's decisions to the host staging itself (no copy behind it).  "dec_mirror" / MIND_DEC_MIRROR
  bool dec_mirror = true;
  // ... and its pruning decisions + branch-time bits come from one launch (k_aime_select_branch) instead of two.  "glue_fused" / MIND_GLUE_FUSED
  bool glue_fused = true;
  // mind_loop: price a candidate tree as soon as the pending tree-iLQR launch marks it complete, beside the trees still being solved.  "early_eval" / MIND_EARLY_EVAL
  bool early_eval = true;
  // ... and its small index tables (the branch set of a round, the job tables of the two packing kernels) are read by the kernels from the
  // page-locked staging directly while they name at most this many workgroups (0 = always uploaded).  "tab_host_max" / MIND_TAB_HOST_MAX
  int tab_host_max = 4096;
  // ... and the scene tables of a round of at most AIME_SMALL scenes per chunk travel in the glue kernels' arguments (no upload between the
  // predictor and k_aime_world).  "tab_small" / MIND_TAB_SMALL
  bool tab_small = true;
  // host tables of a tree-iLQR call (ilqr_impl): kept between calls so that a planning cycle does not allocate a hundred small vectors
  struct IlScratch { std::vector<std::vector<int>> vv[14]; std::vector<int> tmp[6]; std::vector<double> hD; std::vector<float> hF; std::vector<int> hI; } il_scr;
  std::vector<std::vector<int>> pl_scr_kids; std::vector<float> pl_scr_pr; std::vector<int> pl_scr_i[3];      // mind_aime_plan's flattening scratch
  // the pending tree-iLQR launch writes its results to the host itself and marks every tree when it is complete (ilqr_impl `early`): where, and
  // the word's value that means "complete" for THIS launch.  Valid between the launch and its mind_ilqr_finish.
  struct IlEarly { const double *xs = nullptr, *us = nullptr; volatile unsigned *done = nullptr; unsigned gen = 0; int n_trees = 0; long nodes = 0; } il_early;
  unsigned il_gen = 0;
  long long il_spec_req = 0, il_spec_hit = 0;       // last launch, all trees and fits: passes the speculator was asked in / results the master took
  int dec_mfma_min = 1 << 30;   // agents per call from which the decoder's actor part runs on the MFMA kernel (MIND_DEC_MFMA_MIN; default: never)
  bool enc_mfma = true;         // MFMA ActorNet under the bf16x3 / bf16 settings (MIND_ENC_MFMA=0: the fp32 VALU kernel, for A/B)
  // fp32-MFMA ActorNet (k_actor_f32): "actor_f32" 1 (default) = the ActorNet of the exact-fp32 setting (0: the fp32 VALU kernel, for A/B);
  // "actor_f32_min" = actors per call from which every setting takes it with TWO actors per workgroup (the 256-channel layers' weight stream is
  // then shared by two actors: 7.1 vs 8.1 ms for the 13.8 k actors of a cfg4 round); "actor_f32_pair_min" = the same threshold inside the
  // exact-fp32 setting.  Both default to never: a threshold on the batch size would give the blocks of a sharded round another kernel -- other
  // last bits -- than the whole round (a workload that wants it sets 0, as with "dec_mfma_min")
  bool actor_f32 = true;
  int actor_f32_min = 1 << 30, actor_f32_pair_min = 1 << 30;
  bool xcd_order = true;        // XCD-aware job order for big batches (MIND_XCD_ORDER=0 switches it off, for A/B measurements)
  int pair_prec = 3;            // arithmetic of the pair kernel: 0 = fp32 MFMA, 1 = bf16x3 (two-way split operands), 2 = bf16, 3 = bf16x6 (exact three-way split: fp32 class, default)
  // bf16 arithmetics: k_pair_t (tile-native edge tensor, pair_tile_kernels.hip; default) or the row-major k_pair_bf of rounds 2-3
  // (mind_set_tuning("pair_tile", 0) / MIND_PAIR_TILE=0, kept for same-box A/B measurements)
  bool pair_tile = true;
  // mind_aime_plan: a round whose edge tensor would exceed this many MB goes through the predictor in chunks of scenes ("plan_chunk_mb";
  // 96 GB by default: a third of the MI355X's HBM; the scenes of a round are independent, so chunking changes nothing but the launch sizes)
  int plan_chunk_mb = 96 * 1024;
  std::vector<int> last_scene_n;   // tokens per scene of the last predictor call (mind_debug_read("edge") un-permutes with it)
  bool last_edge_tiled = false, last_edge_bf16 = false;
  // workspaces (grow only)
  DevBuf edge, x, ST, QK, part, tokpos, meta, jobs, actor_feat, lane_feat, tgt_feat, cmode, tgt_emb,
      rows, rpe_ptrs;
  // ilqr workspaces
  DevBuf ilqr_dev, aime_dev, rebase_dev2[2], dec_h2;
  // mind_aime_plan (aime_plan.hip): device arenas, page-locked staging (uploads / read-backs are true async copies: a pageable
  // source makes hipMemcpyAsync wait for the stream to drain first), host result tables
  DevBuf pl_root, pl_in[2], pl_lf, pl_lrep, pl_pred, pl_small, pl_tab[2], pl_win[2];
  std::vector<DevBuf> pl_world;
  void *pl_pin[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};     // [4], [5]: upload / read-back staging of the tree-iLQR calls; [6], [7]: sharded plan
  size_t pl_pin_cap[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  hipEvent_t ev_pl = nullptr, ev_tab = nullptr, ev_root = nullptr;
  // the lane-distance field of a plan's contingency solves (gen_dist_field: a function of the ego position, the grid and the target lane --
  // all known when the plan starts), computed on the side stream beside the plan's first round instead of between its last decisions and
  // k_ilqr (il_field_prepare; adopted by ilqr_impl when grid and lane are the ones it was made for)
  DevBuf il_field;
  void *il_field_pin = nullptr;
  size_t il_field_pin_cap = 0;
  hipEvent_t ev_field = nullptr;
  bool il_field_valid = false;
  double il_field_key[5] = {0, 0, 0, 0, 0};       // x0[0], x0[1], W, H, res
  std::vector<double> il_field_lane;
  // mind_set_exchange: the sharded mind_aime_plan (rank, world, the caller's transport, packing buffers, collectives of the last plan)
  int xr = 0, xw = 1;
  mind_exchange_fn xfn = nullptr;
  void *xuser = nullptr;
  bool xforce = false;
  DevBuf x_send, x_recv, x_seg;
  long long x_collectives = 0, x_bytes = 0;
  // "pl_tab_side" / MIND_PL_TAB_SIDE=1: the round tables travel on their own stream while the round's predictor runs instead of behind it
  // on the context stream.  Measured on the recorded demo_1 loop: no difference (AIME 2.00 vs 2.01 ms per plan, profiles/r03ag) -- off.
  hipStream_t pl_copy = nullptr;
  bool pl_tab_side = false;
  std::vector<mind_aime_node> pl_nodes;
  std::vector<float> pl_flat_prob;
  std::vector<double> pl_sol_xs, pl_sol_us;          // results of the solves a plan began itself (mind_ilqr_finish_plan)
  std::vector<mind_ilqr_stats> pl_sol_stw, pl_sol_stf;
  // mind_aime_plan_begin / _finish: the plan on a thread of the library (state 0 idle, 1 running, 2 done)
  std::thread pa_thread;
  std::atomic<int> pa_state{0};
  mind_aime_plan_in pa_in;
  mind_aime_plan_out pa_out;
  int pa_rc = 0;
  const float *pl_rows_p = nullptr, *pl_fmean_p = nullptr, *pl_fcov_p = nullptr;      // into page-locked slot 2, valid until the next plan
  std::vector<int32_t> pl_tree_top, pl_tree_off, pl_flat_parent;
  int pl_plan_agents = 0;       // agents per scene of the plan those tables belong to
  long long pl_gen = 0;         // plans begun on this context so far: whoever holds a plan's library-owned tables (mind_loop) checks they are still that plan's
  DevBuf pl_flat;
  bool dec_overlap = true;      // actor_proj of the decoder on the side stream beside k_dec_scene (mind_set_tuning("dec_overlap"))
  // k_dec_scene_mw: eight workgroups per scene on the decoder's five big stages (bit-identical to the one-workgroup kernel), possible whenever
  // every workgroup of the launch is resident, i.e. for calls of at most n_cu / 8 scenes.  Opt-in ("dec_mw" / MIND_DEC_MW=1): measured 88.6
  // against 94.9 us per demo-size launch (profiles/r06aa_*) -- only ctx_proj's second layer is really bound by one CU's L2 port (30 k -> 20 k
  // cycles); the feed-forward layers are bound by their 24 accumulators per thread and win 2-4 k cycles each, less the 36 KB exchange --
  // 12 us per plan, not worth eight spinning workgroups per scene when several scenes share the device
  bool dec_mw = false;
  // the decoder's cls head as its own launch on the side stream beside the actor part's head ("dec_cls_side" / MIND_DEC_CLS_SIDE=1).  Opt-in:
  // bit-identical, but the second launch and its two event waits cost more than the ~10 us of overlap (1 609-1 630 against 1 632-1 652
  // sim steps/s, profiles/r06am_*)
  bool dec_cls_side = false;
  DevBuf dec_xbuf, dec_bars;
  unsigned *dec_abort = nullptr;      // host-visible abort word of its barriers (page-locked, mapped)
  int rb_cur = 0, rb_gen = 0;     // re-basing arenas: which one the last call filled, its generation and geometry
  size_t rb_S = 0, rb_a = 0;
  // profiling
  bool profiling = false;
  hipEvent_t ev_il0 = nullptr, ev_il1 = nullptr;   // around the tree-iLQR launch of the last call (profiling on)
  float ilqr_ms = 0.f;
  int ilqr_multi = 0, ilqr_trees = 0;
  bool ilqr_test_starve = false;
  // per-iteration traces of the last tree-iLQR call (mind_last_ilqr_trace): device address per tree, rows per phase, iterations run
  std::function<int()> il_finish;     // the pending half of a call begun with mind_ilqr_contingency_begin
  bool il_finish_owned = false;       // ... whose outputs are library buffers (the solves a plan began itself): may be drained and dropped
  // the cost trees' agent means / sigmas of the context's last plan where k_aime_flat wrote them (device): a tree-iLQR call on the plan's own
  // trees (mind_ilqr_contingency_begin_plan) reads them there instead of taking them through the host
  const float *pl_dev_fmean = nullptr, *pl_dev_fcov = nullptr;
  bool il_use_dev_flat = false;
  hipEvent_t ev_rows = nullptr;
  bool il_begin_only = false;
  std::vector<const double *> il_trace_dev;
  std::vector<int> il_trace_its;      // [tree][phase 2]
  int il_trace_cap = 0, il_trace_phases = 0;
  double il_prof[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // critical tree of the last launch: M, depth, passes, cycles of the five phases, trees
  long long n_ilqr_fallbacks = 0;   // wide-tree launches that were not fully resident and were re-run on one workgroup per tree
  int n_pair_launch = 0;
  float pair_ms = 0.f;
  double pairs_done = 0.0;
  std::vector<hipEvent_t> ev;
  // mind_aime_plan with profiling on: the pair-kernel events of its predictor calls are taken from this pool and read once, behind the
  // plan's last synchronisation (a stream drain per predictor call to read twelve events cost ~30 us per round of the timed plan)
  bool ev_defer = false;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_pool_used = 0;
  std::vector<size_t> ev_pending;      // first pool index of every predictor call not read yet
  int n_cu = 256;
  int debug_layers = 6;
  int last_ntok = 0;
  size_t il_dbg[6] = {0, 0, 0, 0, 0, 0};   // tree 0 of the last iLQR call: byte offsets of L, Lx, Lxx, Fx, xs in ilqr_dev; M
  long long last_edge_pairs = 0;
  int last_slots = 0, last_A = 0, last_B = 0;
};

static int fail(mind_ctx *c, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (c) c->err = buf;
  return code;
}

#define HIPCHK(c, call)                                                                        \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return fail(c, MIND_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

static int ensure(mind_ctx *c, DevBuf &b, size_t bytes) {
  if (bytes <= b.cap) return MIND_OK;
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
  // growth: a buffer is re-allocated (hipFree + hipMalloc: a device synchronisation each) whenever a plan needs more than any before it --
  // small buffers (demo-size scenes: the scene and tree-node counts of the first dozens of plans of a process creep upwards) double and start
  // at 1 MB, so that a closed loop stops re-allocating after its first plans (a fresh process ran its first 20 timed cycles at 2.1-2.6 ms of
  // AIME wall instead of 1.8, profiles/r05w_*); big arenas (the deep stress trees: hundreds of GB) keep the 25 % margin
  size_t want = bytes < ((size_t)64 << 20) ? std::max<size_t>(2 * bytes, (size_t)1 << 20) : bytes + bytes / 4 + 4096;
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) {
    e = hipMalloc(&b.p, bytes);
    want = bytes;
    if (e != hipSuccess) return fail(c, MIND_ENOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  }
  b.cap = want;
  return MIND_OK;
}

extern "C" int mind_ctx_create(int device, void *stream, mind_ctx **out) {
  if (!out) return MIND_EINVAL;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return MIND_EHIP;
  if (device < 0 || device >= n) return MIND_EINVAL;
  mind_ctx *c = new mind_ctx();
  c->device = device;
  if (hipSetDevice(device) != hipSuccess) { delete c; return MIND_EHIP; }
  // NULL = the device's default (null) stream, which is what torch.cuda.current_stream() is unless the
  // caller switched streams; everything the caller can observe is ordered on this stream.
  c->stream = (hipStream_t)stream;
  if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess) c->side = nullptr;
  if (c->side && hipEventCreateWithFlags(&c->ev_side, hipEventDisableTiming) != hipSuccess) { (void)hipStreamDestroy(c->side); c->side = nullptr; }
  if (c->side && hipEventCreateWithFlags(&c->ev_main, hipEventDisableTiming) != hipSuccess) { (void)hipStreamDestroy(c->side); c->side = nullptr; }
  if (c->side && hipEventCreateWithFlags(&c->ev_tgt, hipEventDisableTiming) != hipSuccess) { (void)hipStreamDestroy(c->side); c->side = nullptr; }
  if (hipEventCreateWithFlags(&c->ev_stage, hipEventDisableTiming) != hipSuccess) { *out = nullptr; delete c; return MIND_EHIP; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->n_cu = prop.multiProcessorCount;
  (void)hipFuncSetAttribute((const void *)k_pair<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mind_pair_lds_bytes());
  (void)hipFuncSetAttribute((const void *)k_pair<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mind_pair_lds_bytes());
  (void)hipFuncSetAttribute((const void *)k_pair_bf<0, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mind_pair_bf_lds_bytes());
  (void)hipFuncSetAttribute((const void *)k_pair_bf<1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mind_pair_bf_lds_bytes());
  (void)hipFuncSetAttribute((const void *)k_pair_bf<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mind_pair_bf_lds_bytes());
  (void)hipFuncSetAttribute((const void *)k_pair_bf<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mind_pair_bf_lds_bytes());
  (void)hipFuncSetAttribute((const void *)k_pair_t<0, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mind_pair_bf_lds_bytes());
  (void)hipFuncSetAttribute((const void *)k_pair_t<1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mind_pair_bf_lds_bytes());
  (void)hipFuncSetAttribute((const void *)k_pair_t<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mind_pair_bf_lds_bytes());
  (void)hipFuncSetAttribute((const void *)k_pair_t<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mind_pair_bf_lds_bytes());
  (void)hipFuncSetAttribute((const void *)k_pair_t6<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mind_pair_bf_lds_bytes());
  (void)hipFuncSetAttribute((const void *)k_pair_t6<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mind_pair_bf_lds_bytes());
  if (const char *te = getenv("MIND_PAIR_TILE")) c->pair_tile = !(te[0] == '0');
  if (const char *xe = getenv("MIND_XCD_ORDER")) c->xcd_order = !(xe[0] == '0');

  if (const char *pe = getenv("MIND_PAIR_PREC")) {
    const std::string v = pe;
    if (v == "f32" || v == "0") c->pair_prec = 0;
    else if (v == "bf16x3" || v == "1") c->pair_prec = 1;
    else if (v == "bf16" || v == "2") c->pair_prec = 2;
    else if (v == "bf16x6" || v == "3") c->pair_prec = 3;
  }
  (void)hipFuncSetAttribute((const void *)k_actor_net, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mind_actor_lds_bytes());
  (void)hipFuncSetAttribute((const void *)k_actor_mfma<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mind_actor_mfma_lds_bytes());
  (void)hipFuncSetAttribute((const void *)k_actor_mfma<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mind_actor_mfma_lds_bytes());
  (void)hipFuncSetAttribute((const void *)k_dec_actor_mfma<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mind_dec_actor_mfma_lds_bytes());
  (void)hipFuncSetAttribute((const void *)k_dec_actor_mfma<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mind_dec_actor_mfma_lds_bytes());
  (void)hipFuncSetAttribute((const void *)k_dec_actor_mfma<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mind_dec_actor_mfma_lds_bytes());
  if (const char *se = getenv("MIND_ACTOR_SPLIT")) c->actor_np = atoi(se) == 3 ? 3 : 6;
  (void)hipFuncSetAttribute((const void *)k_actor_mfma<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mind_actor_mfma_lds_bytes());
  if (const char *me = getenv("MIND_ENC_MFMA")) c->enc_mfma = !(me[0] == '0');
  if (const char *me = getenv("MIND_ACTOR_F32")) c->actor_f32 = !(me[0] == '0');
  if (const char *me = getenv("MIND_ACTOR_F32_MIN")) c->actor_f32_min = atoi(me);
  (void)hipFuncSetAttribute((const void *)k_actor_f32<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mind_actor_f32_lds_bytes(1));
  (void)hipFuncSetAttribute((const void *)k_actor_f32<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mind_actor_f32_lds_bytes(2));
  if (const char *de = getenv("MIND_DEC_MFMA_MIN")) c->dec_mfma_min = atoi(de);
  if (const char *oe = getenv("MIND_DEC_OVERLAP")) c->dec_overlap = !(oe[0] == '0');
  (void)hipFuncSetAttribute((const void *)k_ilqr<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)il_lds_bytes(0));
  (void)hipFuncSetAttribute((const void *)k_ilqr<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)il_lds_bytes(0));
  (void)hipFuncSetAttribute((const void *)k_ilqr<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)il_lds_bytes(0));
  (void)hipFuncSetAttribute((const void *)k_ilqr<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)il_lds_bytes(0));
  if (const char *te = getenv("MIND_TOK_MFMA")) c->tok_mfma = !(te[0] == '0');
  if (const char *te = getenv("MIND_TOK_SMALL_MAX")) c->tok_small_max = atoi(te);
  if (const char *te = getenv("MIND_TOK_MERGE")) c->tok_merge = !(te[0] == '0');
  if (const char *te = getenv("MIND_DEC_MW")) c->dec_mw = !(te[0] == '0');
  if (const char *te = getenv("MIND_DEC_CLS_SIDE")) c->dec_cls_side = !(te[0] == '0');
  if (const char *te = getenv("MIND_TOK_BF_MIN_N")) c->tok_bf_min_n = atoi(te);
  if (const char *te = getenv("MIND_TGT_SIDE")) c->tgt_side = !(te[0] == '0');
  if (const char *te = getenv("MIND_PL_TAB_SIDE")) c->pl_tab_side = !(te[0] == '0');
  (void)hipFuncSetAttribute((const void *)k_token_mfma<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mind_token_mfma_lds_bytes());
  (void)hipFuncSetAttribute((const void *)k_token_mfma<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mind_token_mfma_lds_bytes());
  if (const char *we = getenv("MIND_ILQR_CHUNK")) c->ilqr_chunk = atoi(we) < 0 ? 0 : atoi(we);
  if (const char *ce = getenv("MIND_PLAN_CHUNK_MB")) { const long v = atol(ce); if (v > 0) c->plan_chunk_mb = v; }
  if (const char *we = getenv("MIND_ILQR_WGS")) { const int v = atoi(we); c->ilqr_wgs = v < 1 ? 1 : (v > 32 ? 32 : v); }
  if (const char *we = getenv("MIND_ILQR_SPEC_DERIV")) c->ilqr_spec_deriv = atoi(we) != 0;
  if (const char *we = getenv("MIND_TAB_SMALL")) c->tab_small = atoi(we) != 0;
  if (const char *we = getenv("MIND_TAB_HOST_MAX")) c->tab_host_max = std::max(0, atoi(we));
  if (const char *we = getenv("MIND_EARLY_EVAL")) c->early_eval = atoi(we) != 0;
  if (const char *we = getenv("MIND_GLUE_FUSED")) c->glue_fused = atoi(we) != 0;
  if (const char *we = getenv("MIND_DEC_MIRROR")) c->dec_mirror = atoi(we) != 0;
  if (const char *we = getenv("MIND_ILQR_HOST_OUT_MAX")) c->ilqr_host_out_max = std::max(0, atoi(we));
  if (const char *we = getenv("MIND_UPLOAD_KERNEL_MAX")) c->upload_kernel_max = std::max(0, atoi(we));
  if (const char *we = getenv("MIND_ILQR_SLOTS")) { const int v = atoi(we); c->ilqr_slots = v < 1 ? 1 : (v > IL_SLOTS ? IL_SLOTS : v); }
  (void)hipFuncSetAttribute((const void *)k_dec_scene, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mind_dec_scene_lds_bytes());
  (void)hipFuncSetAttribute((const void *)k_dec_scene_mw, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mind_dec_scene_lds_bytes());
  (void)hipFuncSetAttribute((const void *)k_dec_scene_c, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mind_dec_scene_lds_bytes());
  (void)hipFuncSetAttribute((const void *)k_dec_cls, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mind_dec_scene_lds_bytes());
  (void)hipFuncSetAttribute((const void *)k_dec_tgt, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mind_dec_scene_lds_bytes());
  (void)hipFuncSetAttribute((const void *)k_dec_actor<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mind_dec_actor_lds_bytes());
  (void)hipFuncSetAttribute((const void *)k_dec_actor<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mind_dec_actor_lds_bytes());
  (void)hipFuncSetAttribute((const void *)k_dec_actor<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mind_dec_actor_lds_bytes());
  *out = c;
  return MIND_OK;
}

extern "C" int mind_ctx_destroy(mind_ctx *c) {
  if (!c) return MIND_EINVAL;
  if (c->pa_thread.joinable()) c->pa_thread.join();
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  DevBuf *bufs[] = {&c->edge, &c->x, &c->ST, &c->QK, &c->part, &c->tokpos, &c->meta, &c->jobs, &c->actor_feat,
                    &c->lane_feat, &c->tgt_feat, &c->cmode, &c->tgt_emb, &c->rows, &c->rpe_ptrs, &c->ilqr_dev, &c->aime_dev, &c->rebase_dev2[0], &c->rebase_dev2[1], &c->dec_h2};
  for (DevBuf *b : bufs)
    if (b->p) (void)hipFree(b->p);
  for (TableSet &t : c->tabs)
    for (DevBuf *b : {&t.meta, &t.jobs, &t.jobs5, &t.rows})
      if (b->p) (void)hipFree(b->p);
  if (c->wdev) (void)hipFree(c->wdev);
  for (DevBuf *b : {&c->dec_xbuf, &c->dec_bars})
    if (b->p) (void)hipFree(b->p);
  if (c->dec_abort) (void)hipHostFree(c->dec_abort);
  for (DevBuf *b : {&c->pl_root, &c->pl_in[0], &c->pl_in[1], &c->pl_lf, &c->pl_lrep, &c->pl_pred, &c->pl_small, &c->pl_tab[0], &c->pl_tab[1], &c->pl_win[0],
                    &c->pl_win[1], &c->pl_flat, &c->x_send, &c->x_recv, &c->x_seg})
    if (b->p) (void)hipFree(b->p);
  for (DevBuf &b : c->pl_world)
    if (b.p) (void)hipFree(b.p);
  for (void *q : c->pl_pin)
    if (q) (void)hipHostFree(q);
  if (c->ev_pl) (void)hipEventDestroy(c->ev_pl);
  if (c->ev_root) (void)hipEventDestroy(c->ev_root);
  if (c->ev_field) (void)hipEventDestroy(c->ev_field);
  if (c->il_field.p) (void)hipFree(c->il_field.p);
  if (c->il_field_pin) (void)hipHostFree(c->il_field_pin);
  if (c->ev_tab) (void)hipEventDestroy(c->ev_tab);
  if (c->ev_rows) (void)hipEventDestroy(c->ev_rows);
  if (c->pl_copy) (void)hipStreamDestroy(c->pl_copy);
  for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
  if (c->ev_side) (void)hipEventDestroy(c->ev_side);
  if (c->ev_main) (void)hipEventDestroy(c->ev_main);
  if (c->ev_tgt) (void)hipEventDestroy(c->ev_tgt);
  if (c->ev_stage) (void)hipEventDestroy(c->ev_stage);
  if (c->ev_cls) (void)hipEventDestroy(c->ev_cls);
  if (c->ev_il0) (void)hipEventDestroy(c->ev_il0);
  if (c->ev_il1) (void)hipEventDestroy(c->ev_il1);
  if (c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamDestroy(c->side); }
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return MIND_OK;
}

extern "C" const char *mind_last_error_string(mind_ctx *c) { return c ? c->err.c_str() : "null context"; }

extern "C" int mind_ctx_synchronize(mind_ctx *c) {
  if (!c) return MIND_EINVAL;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MIND_OK;
}

extern "C" int mind_set_tuning(mind_ctx *c, const char *name, int value) {
  if (!c || !name) return MIND_EINVAL;
  const std::string n = name;
  if (n == "dec_mfma_min") c->dec_mfma_min = value;
  else if (n == "enc_mfma") c->enc_mfma = value != 0;
  else if (n == "actor_f32") c->actor_f32 = value != 0;
  else if (n == "actor_f32_min") c->actor_f32_min = value;
  else if (n == "actor_f32_pair_min") c->actor_f32_pair_min = value;
  else if (n == "actor_split") c->actor_np = value == 3 ? 3 : 6;
  else if (n == "xcd_order") c->xcd_order = value != 0;
  else if (n == "pair_tile") c->pair_tile = value != 0;
  else if (n == "plan_chunk_mb") c->plan_chunk_mb = value < 1 ? 1 : value;
  else if (n == "dec_overlap") c->dec_overlap = value != 0;
  else if (n == "tok_mfma") c->tok_mfma = value != 0;
  else if (n == "tok_small_max") c->tok_small_max = (int)value;
  else if (n == "tok_merge") c->tok_merge = value != 0;
  else if (n == "dec_mw") c->dec_mw = value != 0;
  else if (n == "dec_cls_side") c->dec_cls_side = value != 0;
  else if (n == "tok_bf_min_n") c->tok_bf_min_n = (int)value;
  else if (n == "tgt_side") c->tgt_side = value != 0;
  else if (n == "pl_tab_side") c->pl_tab_side = value != 0;
  else if (n == "ilqr_chunk") c->ilqr_chunk = value < 0 ? 0 : (int)value;
  else if (n == "ilqr_wgs") c->ilqr_wgs = value < 1 ? 1 : (value > 32 ? 32 : value);
  else if (n == "ilqr_multi_min") c->ilqr_multi_min = value;
  else if (n == "ilqr_wgs_big") c->ilqr_wgs_big = value < 1 ? 1 : (value > 32 ? 32 : value);
  else if (n == "ilqr_big_min") c->ilqr_big_min = value;
  else if (n == "ilqr_spec_deriv") c->ilqr_spec_deriv = value != 0;
  else if (n == "tab_small") c->tab_small = value != 0;
  else if (n == "tab_host_max") c->tab_host_max = value < 0 ? 0 : value;
  else if (n == "early_eval") c->early_eval = value != 0;
  else if (n == "glue_fused") c->glue_fused = value != 0;
  else if (n == "dec_mirror") c->dec_mirror = value != 0;
  else if (n == "ilqr_host_out_max") c->ilqr_host_out_max = value < 0 ? 0 : value;
  else if (n == "upload_kernel_max") c->upload_kernel_max = value < 0 ? 0 : value;
  else if (n == "ilqr_slots") c->ilqr_slots = value < 1 ? 1 : (value > IL_SLOTS ? IL_SLOTS : value);
  else if (n == "ilqr_test_starve") c->ilqr_test_starve = value != 0;   // tests: launch a wide tree without its last workgroups
  else return fail(c, MIND_EINVAL, "mind_set_tuning: unknown knob '%s'", name);
  return MIND_OK;
}

extern "C" int mind_set_exchange(mind_ctx *c, int rank, int world, mind_exchange_fn fn, void *user, int force) {
  if (!c) return MIND_EINVAL;
  if (fn && (world < 1 || rank < 0 || rank >= world)) return fail(c, MIND_EINVAL, "mind_set_exchange: rank %d of %d", rank, world);
  c->xfn = fn; c->xuser = user;
  c->xr = fn ? rank : 0; c->xw = fn ? world : 1; c->xforce = fn && force != 0;
  return MIND_OK;
}

extern "C" int mind_last_exchange_stats(mind_ctx *c, long long *collectives, long long *bytes) {
  if (!c) return MIND_EINVAL;
  if (collectives) *collectives = c->x_collectives;
  if (bytes) *bytes = c->x_bytes;
  return MIND_OK;
}

extern "C" int mind_set_pair_precision(mind_ctx *c, int mode) {
  if (!c || mode < 0 || mode > 3) return MIND_EINVAL;
  c->pair_prec = mode;
  return MIND_OK;
}

extern "C" int mind_get_pair_precision(mind_ctx *c) { return c ? c->pair_prec : MIND_EINVAL; }

namespace { std::vector<float> pack_bfrag(const std::vector<float> &w, int row_stride); }
extern "C" int mind_debug_pack_bfrag(const float *w, int row_stride, uint32_t *out) {
  if (!w || !out || row_stride < 128) return MIND_EINVAL;
  std::vector<float> v(w, w + (size_t)127 * row_stride + 128);
  v.resize((size_t)128 * row_stride, 0.f);
  const std::vector<float> t = pack_bfrag(v, row_stride);
  memcpy(out, t.data(), 16384 * sizeof(uint32_t));
  return MIND_OK;
}

// the pair kernels' job lists as mind_predict_batch builds them (same helpers of pair_jobs.h), for a host-side check of the schedule
extern "C" int mind_debug_pair_schedule(const int *scene_tokens, const int *scene_actors, int n_scenes, int n_cu, int last_layer, int *out_jobs, int cap,
                                        int *out_info) {
  if (!scene_tokens || !scene_actors || n_scenes <= 0 || n_cu <= 0 || !out_jobs || !out_info) return MIND_EINVAL;
  std::vector<PairJob> jl;
  int slot = 0;
  for (int b = 0; b < n_scenes; ++b) {
    const int N = scene_tokens[b], a = scene_actors[b];
    if (N <= 0 || a < 0 || a >= N) return MIND_EINVAL;
    const int tiles = (N + 15) / 16, ns = pair_column_splits(N);
    for (int j = 0; j < N; ++j)
      for (int s_ = 0; s_ < ns; ++s_) {
        PairJob J;
        memset(&J, 0, sizeof(J));
        J.N = N; J.j = j; J.scene = b; J.slot = slot++;
        pair_job_range(tiles, ns, s_, &J.t0, &J.t1);
        J.flags = (j < a || j == N - 1) ? 1 : 0;
        if (!last_layer || (J.flags & 1)) jl.push_back(J);
      }
  }
  const int njobs = (int)jl.size();
  const int grid = njobs < n_cu ? njobs : n_cu;
  pair_jobs_deal(jl, grid, PAIR_WAVES, (n_scenes >= 8 && grid % 8 == 0) ? 8 : 1);
  const int stride = grid * PAIR_WAVES;
  int n = 0;
  for (size_t i = 0; i < jl.size(); ++i) {
    const PairJob &J = jl[i];
    if (J.t1 <= J.t0) continue;
    if (n < cap) {
      int *o = out_jobs + (size_t)6 * n;
      o[0] = J.scene; o[1] = J.j; o[2] = J.t0; o[3] = J.t1; o[4] = J.slot; o[5] = (int)(i % stride);
    }
    ++n;
  }
  out_info[0] = pair_column_splits(scene_tokens[0]); out_info[1] = grid; out_info[2] = (int)jl.size(); out_info[3] = n;
  return n;
}

extern "C" int mind_set_profiling(mind_ctx *c, int enable) {
  if (!c) return MIND_EINVAL;
  c->profiling = enable != 0;
  return MIND_OK;
}

extern "C" int mind_last_ilqr_stats(mind_ctx *c, float *kernel_ms, int *n_trees, int *workgroups_per_tree) {
  if (!c) return MIND_EINVAL;
  if (kernel_ms) *kernel_ms = c->ilqr_ms;
  if (n_trees) *n_trees = c->ilqr_trees;
  if (workgroups_per_tree) *workgroups_per_tree = c->ilqr_multi;
  return MIND_OK;
}

extern "C" int mind_last_ilqr_trace(mind_ctx *c, int tree, int phase, double *out, int cap_rows, int *n_rows) {
  if (!c || !n_rows || (cap_rows > 0 && !out)) return MIND_EINVAL;
  if (tree < 0 || tree >= (int)c->il_trace_dev.size() || phase < 0 || phase >= c->il_trace_phases || !c->il_trace_dev[tree])
    return fail(c, MIND_ESTATE, "mind_last_ilqr_trace: no trace for tree %d phase %d (the last tree-iLQR call had %d trees, %d phases)", tree, phase,
                (int)c->il_trace_dev.size(), c->il_trace_phases);
  const int rows = std::min(c->il_trace_its[2 * tree + phase], c->il_trace_cap);
  *n_rows = rows;
  const int take = std::min(rows, cap_rows);
  if (take > 0) {
    HIPCHK(c, hipMemcpyAsync(out, c->il_trace_dev[tree] + (size_t)phase * c->il_trace_cap * IL_TRACE_W, (size_t)take * IL_TRACE_W * sizeof(double),
                             hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return MIND_OK;
}

extern "C" int mind_last_ilqr_profile(mind_ctx *c, double *out9) {
  if (!c || !out9) return MIND_EINVAL;
  memcpy(out9, c->il_prof, sizeof(c->il_prof));
  return MIND_OK;
}

// deferred pair-kernel events of a plan (ev_defer): summed once the stream has been drained
static int mind_pair_events_resolve(mind_ctx *c, float *total_ms) {
  float sum = 0.f;
  for (size_t base : c->ev_pending)
    for (int L = 0; L < c->debug_layers; ++L) {
      float ms = 0.f;
      HIPCHK(c, hipEventElapsedTime(&ms, c->ev_pool[base + 2 * L], c->ev_pool[base + 2 * L + 1]));
      sum += ms;
    }
  c->ev_pending.clear();
  c->ev_pool_used = 0;
  *total_ms = sum;
  return MIND_OK;
}

extern "C" int mind_last_fusion_stats(mind_ctx *c, int *n_launches, float *total_ms, double *pairs) {
  if (!c) return MIND_EINVAL;
  if (n_launches) *n_launches = c->n_pair_launch;
  if (total_ms) *total_ms = c->pair_ms;
  if (pairs) *pairs = c->pairs_done;
  return MIND_OK;
}

// -------------------------------------------------------------------------------------------------
// weight packing
// -------------------------------------------------------------------------------------------------
namespace {

struct SD {
  std::map<std::string, std::pair<const float *, int64_t>> m;
  std::string missing;
  const float *get(const std::string &k, int64_t numel) {
    auto it = m.find(k);
    if (it == m.end() || it->second.second != numel) {
      if (missing.empty()) missing = k;
      return nullptr;
    }
    return it->second.first;
  }
};

std::vector<float> vec(const float *p, size_t n) { return p ? std::vector<float>(p, p + n) : std::vector<float>(n, 0.f); }

// [out][in] -> [in][out]
std::vector<float> transpose(const float *w, int n_out, int n_in, int row_stride = -1, int col0 = 0) {
  if (row_stride < 0) row_stride = n_in;
  std::vector<float> t((size_t)n_out * n_in, 0.f);
  if (!w) return t;
  for (int o = 0; o < n_out; ++o)
    for (int k = 0; k < n_in; ++k) t[(size_t)k * n_out + o] = w[(size_t)o * row_stride + col0 + k];
  return t;
}

// MFMA 16x16x4 A-fragment order: [ob 8][s4 8][lane 64][w 4] = W[16 ob + (lane&15)][16 s4 + 4 (lane>>4) + w]
std::vector<float> pack_afrag(const float *w, int row_stride) {
  std::vector<float> t(16384, 0.f);
  if (!w) return t;
  for (int ob = 0; ob < 8; ++ob)
    for (int s4 = 0; s4 < 8; ++s4)
      for (int lane = 0; lane < 64; ++lane)
        for (int k = 0; k < 4; ++k)
          t[((size_t)(ob * 8 + s4) * 64 + lane) * 4 + k] =
              w[(size_t)(16 * ob + (lane & 15)) * row_stride + 16 * s4 + 4 * (lane >> 4) + k];
  return t;
}

// folded-K-query fragments of k_token_mfma: per head hd and output block ob the A operand [16 features] x [16 d]:
// [hd 8][ob 8][lane 64][w 4] = Wk[hd * 16 + 4 (lane >> 4) + w][16 ob + (lane & 15)]   (Wk = rows 128..255 of in_proj_weight)
std::vector<float> pack_wk_frag(const float *wk) {
  std::vector<float> t(16384, 0.f);
  if (!wk) return t;
  for (int hd = 0; hd < 8; ++hd)
    for (int ob = 0; ob < 8; ++ob)
      for (int lane = 0; lane < 64; ++lane)
        for (int k = 0; k < 4; ++k)
          t[((size_t)(hd * 8 + ob) * 64 + lane) * 4 + k] = wk[(size_t)(hd * 16 + 4 * (lane >> 4) + k) * 128 + 16 * ob + (lane & 15)];
  return t;
}

// round-to-nearest-even bf16 of a float (bits), and the split x = hi + lo
inline uint16_t bf16_rne(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
inline float bf16_to_f32(uint16_t h) {
  const uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// bf16 MFMA 16x16x32 A-fragment order, hi and lo parts: [part 2][ob 8][g 4][lane 64][dword 4]; dword d of lane
// (r = lane & 15, q = lane >> 4) packs k-slots 2d, 2d+1; k-slot (q, i) <-> input feature 16 (2g + (i >> 2)) + 4q + (i & 3)
// (the chained B-operand order of pair_bf16_kernels.hip).  Returned as float bit patterns (16384 dwords).
// the third part of the exact three-way split x = hi + mid + lo (mid = pack_bfrag's lo part), same fragment order: [ob 8][g 4][lane 64][dword 4]
std::vector<float> pack_bfrag_lo3(const std::vector<float> &w, int row_stride) {
  std::vector<uint32_t> t(8192, 0u);
  for (int ob = 0; ob < 8; ++ob)
    for (int g = 0; g < 4; ++g)
      for (int lane = 0; lane < 64; ++lane)
        for (int d = 0; d < 4; ++d) {
          uint32_t lo = 0;
          for (int e = 0; e < 2; ++e) {
            const int i = 2 * d + e, q = lane >> 4;
            const int f = 16 * (2 * g + (i >> 2)) + 4 * q + (i & 3);
            const float x = w[(size_t)(16 * ob + (lane & 15)) * row_stride + f];
            const float r1 = x - bf16_to_f32(bf16_rne(x));
            const float r2 = r1 - bf16_to_f32(bf16_rne(r1));
            lo |= (uint32_t)bf16_rne(r2) << (16 * e);
          }
          t[((size_t)(ob * 4 + g) * 64 + lane) * 4 + d] = lo;
        }
  std::vector<float> out(8192);
  memcpy(out.data(), t.data(), 8192 * sizeof(float));
  return out;
}

std::vector<float> pack_bfrag(const std::vector<float> &w, int row_stride) {
  std::vector<uint32_t> t(16384, 0u);
  for (int ob = 0; ob < 8; ++ob)
    for (int g = 0; g < 4; ++g)
      for (int lane = 0; lane < 64; ++lane)
        for (int d = 0; d < 4; ++d) {
          uint32_t hi = 0, lo = 0;
          for (int e = 0; e < 2; ++e) {
            const int i = 2 * d + e, q = lane >> 4;
            const int f = 16 * (2 * g + (i >> 2)) + 4 * q + (i & 3);
            const float x = w[(size_t)(16 * ob + (lane & 15)) * row_stride + f];
            const uint16_t h = bf16_rne(x);
            const uint16_t l = bf16_rne(x - bf16_to_f32(h));
            hi |= (uint32_t)h << (16 * e);
            lo |= (uint32_t)l << (16 * e);
          }
          const size_t o = ((size_t)(ob * 4 + g) * 64 + lane) * 4 + d;
          t[o] = hi;
          t[8192 + o] = lo;
        }
  std::vector<float> out(16384);
  memcpy(out.data(), t.data(), 16384 * sizeof(float));
  return out;
}

// k_token_mfma<1> (bf16 hi + lo split operands): A fragments of a [128 out] x [128 in] block of w (row stride row_stride) for the bf16 MFMA
// 16x16x32 in NATURAL k order -- the B operand is read from an fp32 LDS tile, 8 consecutive features per lane --:
// [ob 8][s 8 = part * 4 + g][lane 64][dword 4]; dword d of lane (r = lane & 15, q = lane >> 4) packs W[16 ob + r][32 g + 8 q + 2 d (+ 1)].
// Same addressing as pack_afrag's (tm_load reads both), returned as float bit patterns (16384 dwords).
std::vector<float> pack_tok_bfrag(const float *w, int row_stride) {
  std::vector<uint32_t> t(16384, 0u);
  if (w)
    for (int ob = 0; ob < 8; ++ob)
      for (int g = 0; g < 4; ++g)
        for (int lane = 0; lane < 64; ++lane)
          for (int d = 0; d < 4; ++d) {
            uint32_t hi = 0, lo = 0;
            for (int e = 0; e < 2; ++e) {
              const float x = w[(size_t)(16 * ob + (lane & 15)) * row_stride + 32 * g + 8 * (lane >> 4) + 2 * d + e];
              const uint16_t h = bf16_rne(x);
              const uint16_t l = bf16_rne(x - bf16_to_f32(h));
              hi |= (uint32_t)h << (16 * e);
              lo |= (uint32_t)l << (16 * e);
            }
            t[((size_t)(ob * 8 + g) * 64 + lane) * 4 + d] = hi;
            t[((size_t)(ob * 8 + 4 + g) * 64 + lane) * 4 + d] = lo;
          }
  std::vector<float> f(16384);
  memcpy(f.data(), t.data(), 16384 * sizeof(float));
  return f;
}

// LayerNorm is invariant to a shift along the features: fold the mean subtraction of the LayerNorm that follows a Linear
// into the Linear ((I - 11^T/n) W, (I - 11^T/n) b).  W is [n_out][n_in] row-major.
void center_outputs(std::vector<float> &W, std::vector<float> &b, int n_out, int n_in) {
  for (int k = 0; k < n_in; ++k) {
    double m = 0;
    for (int o = 0; o < n_out; ++o) m += W[(size_t)o * n_in + k];
    m /= n_out;
    for (int o = 0; o < n_out; ++o) W[(size_t)o * n_in + k] = (float)((double)W[(size_t)o * n_in + k] - m);
  }
  double mb = 0;
  for (int o = 0; o < n_out; ++o) mb += b[o];
  mb /= n_out;
  for (int o = 0; o < n_out; ++o) b[o] = (float)((double)b[o] - mb);
}

// conv weight [co][ci][k] -> [ci][k][co]
std::vector<float> conv_t(const float *w, int co, int ci, int k) {
  std::vector<float> t((size_t)co * ci * k, 0.f);
  if (!w) return t;
  for (int o = 0; o < co; ++o)
    for (int i = 0; i < ci; ++i)
      for (int d = 0; d < k; ++d) t[((size_t)i * k + d) * co + o] = w[((size_t)o * ci + i) * k + d];
  return t;
}

// conv weight [co][ci][k] (torch layout) -> A-operand fragments of v_mfma_f32_16x16x32_bf16 for the GEMM of actor_mfma_kernels.hip:
// [m-tile co/16][k-step][part hi, mid, lo][lane 64][4 dwords] (w = hi + mid + lo exactly, three bf16 parts); lane (r, q) holds row
// co = 16 mt + r, k-slots 8 q + (0..7) of the step, dword d = slots 2 d, 2 d + 1; GEMM index k = dk * ci_pad + ci (ci_pad = ci
// rounded up to a power of two >= 16; zeros beyond).
std::vector<float> pack_conv_frag(const float *w, int co, int ci, int ksz, int cp_force = 0, int co_pad = 0) {
  int cp = 16;
  while (cp < ci) cp *= 2;
  if (cp_force) cp = cp_force;     // Linear layers: ci itself (a multiple of 32); co_pad: output rows rounded up to 16 (zero rows)
  const int ks = (ksz * cp + 31) / 32, mts = (co_pad > co ? co_pad : co) / 16;
  std::vector<uint32_t> t((size_t)mts * ks * 768, 0u);
  if (w)
    for (int mt = 0; mt < mts; ++mt)
      for (int s = 0; s < ks; ++s)
        for (int lane = 0; lane < 64; ++lane)
          for (int d = 0; d < 4; ++d) {
            uint32_t part[3] = {0, 0, 0};
            for (int e = 0; e < 2; ++e) {
              const int k = s * 32 + 8 * (lane >> 4) + 2 * d + e;
              const int dk = k / cp, c = k % cp, o = 16 * mt + (lane & 15);
              float x = (dk < ksz && c < ci && o < co) ? w[((size_t)o * ci + c) * ksz + dk] : 0.f;
              for (int pi = 0; pi < 3; ++pi) {
                const uint16_t h = bf16_rne(x);
                part[pi] |= (uint32_t)h << (16 * e);
                x -= bf16_to_f32(h);
              }
            }
            const size_t base = ((size_t)(mt * ks + s) * 3) * 256 + (size_t)lane * 4 + d;
            for (int pi = 0; pi < 3; ++pi) t[base + 256 * pi] = part[pi];
          }
  std::vector<float> out(t.size());
  memcpy(out.data(), t.data(), t.size() * sizeof(float));
  return out;
}

// conv weight [co][ci][k] (torch layout) -> A-operand fragments of v_mfma_f32_16x16x4_f32 for the GEMM of actor_f32_kernels.hip:
// [m-tile co/16][k-group of 16][lane 64][4 floats]; lane (r, q) holds row co = 16 mt + r, GEMM indices k = 16 g + 4 q + (0..3),
// k = dk * ci_pad + ci (ci_pad = ci rounded up to a power of two >= 16; zeros beyond)
std::vector<float> pack_conv_f32(const float *w, int co, int ci, int ksz) {
  int cp = 16;
  while (cp < ci) cp *= 2;
  const int G = ksz * cp / 16, mts = co / 16;
  std::vector<float> t((size_t)mts * G * 256, 0.f);
  if (w)
    for (int mt = 0; mt < mts; ++mt)
      for (int g = 0; g < G; ++g)
        for (int lane = 0; lane < 64; ++lane)
          for (int m = 0; m < 4; ++m) {
            const int k = g * 16 + 4 * (lane >> 4) + m;
            const int dk = k / cp, cc = k % cp, o = 16 * mt + (lane & 15);
            t[((size_t)(mt * G + g) * 64 + lane) * 4 + m] = (dk < ksz && cc < ci) ? w[((size_t)o * ci + cc) * ksz + dk] : 0.f;
          }
  return t;
}

}  // namespace

extern "C" int mind_debug_pack_conv_frag(const float *w, int co, int ci, int ksz, uint32_t *out, size_t cap) {
  if (!w || !out || co <= 0 || co % 16 || ci <= 0 || ksz <= 0) return MIND_EINVAL;
  const std::vector<float> t = pack_conv_frag(w, co, ci, ksz);
  if (t.size() > cap) return MIND_EINVAL;
  memcpy(out, t.data(), t.size() * sizeof(float));
  return (int)t.size();
}

extern "C" int mind_weights_load(mind_ctx *c, const mind_tensor_desc *tensors, int n) {
  if (!c || !tensors || n <= 0) return MIND_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  SD sd;
  for (int i = 0; i < n; ++i)
    if (tensors[i].name && tensors[i].data) sd.m[tensors[i].name] = {tensors[i].data, tensors[i].numel};
  WeightBlob &B = c->blob;
  B.host.clear();
  B.off.clear();
  auto lin_ln = [&](const std::string &key, const std::string &p, int idx, int n_out, int n_in) {
    // Linear @idx (transposed) + LayerNorm @idx+1
    B.add(key + ".W", transpose(sd.get(p + "." + std::to_string(idx) + ".weight", (int64_t)n_out * n_in), n_out, n_in));
    B.add(key + ".b", vec(sd.get(p + "." + std::to_string(idx) + ".bias", n_out), n_out));
    B.add(key + ".g", vec(sd.get(p + "." + std::to_string(idx + 1) + ".weight", n_out), n_out));
    B.add(key + ".be", vec(sd.get(p + "." + std::to_string(idx + 1) + ".bias", n_out), n_out));
  };
  // ---- lane_net
  lin_ln("lane.proj", "lane_net.proj", 0, 128, 16);
  for (int b = 0; b < 2; ++b) {
    std::string p = "lane_net.aggre" + std::to_string(b + 1), k = "lane.a" + std::to_string(b);
    lin_ln(k + ".f10", p + ".fc1", 0, 128, 128);
    lin_ln(k + ".f13", p + ".fc1", 3, 128, 128);
    lin_ln(k + ".f20", p + ".fc2", 0, 128, 256);
    lin_ln(k + ".f23", p + ".fc2", 3, 128, 128);
    B.add(k + ".ng", vec(sd.get(p + ".norm.weight", 128), 128));
    B.add(k + ".nb", vec(sd.get(p + ".norm.bias", 128), 128));
  }
  // ---- actor_net
  {
    const int chans[4] = {32, 64, 128, 256};
    int n_in = 14;
    int ri = 0;
    auto res = [&](const std::string &p, int ci, int co, bool ds) {
      std::string k = "act.r" + std::to_string(ri++);
      B.add(k + ".c1", conv_t(sd.get(p + ".conv1.weight", (int64_t)co * ci * 3), co, ci, 3));
      B.add(k + ".c2", conv_t(sd.get(p + ".conv2.weight", (int64_t)co * co * 3), co, co, 3));
      B.add(k + ".c1B", pack_conv_frag(sd.get(p + ".conv1.weight", (int64_t)co * ci * 3), co, ci, 3));
      B.add(k + ".c2B", pack_conv_frag(sd.get(p + ".conv2.weight", (int64_t)co * co * 3), co, co, 3));
      B.add(k + ".c1F", pack_conv_f32(sd.get(p + ".conv1.weight", (int64_t)co * ci * 3), co, ci, 3));
      B.add(k + ".c2F", pack_conv_f32(sd.get(p + ".conv2.weight", (int64_t)co * co * 3), co, co, 3));
      B.add(k + ".g1", vec(sd.get(p + ".bn1.weight", co), co));
      B.add(k + ".b1", vec(sd.get(p + ".bn1.bias", co), co));
      B.add(k + ".g2", vec(sd.get(p + ".bn2.weight", co), co));
      B.add(k + ".b2", vec(sd.get(p + ".bn2.bias", co), co));
      if (ds) {
        B.add(k + ".ds", conv_t(sd.get(p + ".downsample.0.weight", (int64_t)co * ci), co, ci, 1));
        B.add(k + ".dsB", pack_conv_frag(sd.get(p + ".downsample.0.weight", (int64_t)co * ci), co, ci, 1));
        B.add(k + ".dsF", pack_conv_f32(sd.get(p + ".downsample.0.weight", (int64_t)co * ci), co, ci, 1));
        B.add(k + ".gd", vec(sd.get(p + ".downsample.1.weight", co), co));
        B.add(k + ".bd", vec(sd.get(p + ".downsample.1.bias", co), co));
      }
    };
    for (int g = 0; g < 4; ++g) {
      res("actor_net.groups." + std::to_string(g) + ".0", n_in, chans[g], true);
      res("actor_net.groups." + std::to_string(g) + ".1", chans[g], chans[g], false);
      n_in = chans[g];
    }
    res("actor_net.output", 128, 128, false);
    for (int g = 0; g < 4; ++g) {
      std::string p = "actor_net.lateral." + std::to_string(g), k = "act.lat" + std::to_string(g);
      B.add(k + ".W", conv_t(sd.get(p + ".conv.weight", (int64_t)128 * chans[g] * 3), 128, chans[g], 3));
      B.add(k + ".WB", pack_conv_frag(sd.get(p + ".conv.weight", (int64_t)128 * chans[g] * 3), 128, chans[g], 3));
      B.add(k + ".WF", pack_conv_f32(sd.get(p + ".conv.weight", (int64_t)128 * chans[g] * 3), 128, chans[g], 3));
      B.add(k + ".g", vec(sd.get(p + ".norm.weight", 128), 128));
      B.add(k + ".b", vec(sd.get(p + ".norm.bias", 128), 128));
    }
  }
  // ---- fusion_net
  lin_ln("fus.pa", "fusion_net.proj_actor", 0, 128, 128);
  lin_ln("fus.pl", "fusion_net.proj_lane", 0, 128, 128);
  B.add("fus.pa.WM", pack_afrag(sd.get("fusion_net.proj_actor.0.weight", 128 * 128), 128));      // the same as fp32 MFMA A fragments (k_token_mfma)
  B.add("fus.pl.WM", pack_afrag(sd.get("fusion_net.proj_lane.0.weight", 128 * 128), 128));
  B.add("fus.pa.WB", pack_tok_bfrag(sd.get("fusion_net.proj_actor.0.weight", 128 * 128), 128));     // ... and as bf16 hi / lo fragments (k_token_mfma<1>)
  B.add("fus.pl.WB", pack_tok_bfrag(sd.get("fusion_net.proj_lane.0.weight", 128 * 128), 128));
  {
    // rpe table [32 chunks][8][4]: k<5 W_r[f][k], 5 bias, 6 gamma, 7 beta, f = 4 chunk + w
    const float *W = sd.get("fusion_net.proj_rpe_scene.0.weight", 128 * 5);
    const float *b = sd.get("fusion_net.proj_rpe_scene.0.bias", 128);
    const float *g = sd.get("fusion_net.proj_rpe_scene.1.weight", 128);
    const float *be = sd.get("fusion_net.proj_rpe_scene.1.bias", 128);
    std::vector<float> t(1024, 0.f);
    if (W && b && g && be)
      for (int ch = 0; ch < 32; ++ch)
        for (int w = 0; w < 4; ++w) {
          const int f = 4 * ch + w;
          for (int k = 0; k < 5; ++k) t[ch * 32 + k * 4 + w] = W[f * 5 + k];
          t[ch * 32 + 20 + w] = b[f];
          t[ch * 32 + 24 + w] = g[f];
          t[ch * 32 + 28 + w] = be[f];
        }
    B.add("fus.rtab", t);
  }
  for (int L = 0; L < 6; ++L) {
    std::string p = "fusion_net.fuse_scene.fusion." + std::to_string(L), k = "fus.L" + std::to_string(L);
    // proj_memory / proj_edge are followed by a LayerNorm: its mean subtraction is folded into the weights (center_outputs)
    std::vector<float> Wmc = vec(sd.get(p + ".proj_memory.0.weight", 128 * 384), 128 * 384);
    std::vector<float> bmc = vec(sd.get(p + ".proj_memory.0.bias", 128), 128);
    center_outputs(Wmc, bmc, 128, 384);
    const float *Wm = Wmc.data();
    B.add(k + ".WAe", pack_afrag(Wm, 384));
    B.add(k + ".WBe", pack_bfrag(Wmc, 384));
    B.add(k + ".WLe", pack_bfrag_lo3(Wmc, 384));
    B.add(k + ".WsT", transpose(Wm, 128, 128, 384, 128));
    B.add(k + ".WtT", transpose(Wm, 128, 128, 384, 256));
    B.add(k + ".bm", bmc);
    std::vector<float> vt(VT_SIZE, 0.f);
    auto put = [&](int off, const float *src) {
      if (src) memcpy(vt.data() + off, src, 128 * sizeof(float));
    };
    put(VT_GM, sd.get(p + ".proj_memory.1.weight", 128));
    put(VT_BM, sd.get(p + ".proj_memory.1.bias", 128));
    if (L != 5) {
      std::vector<float> Wpc = vec(sd.get(p + ".proj_edge.0.weight", 128 * 128), 128 * 128);
      std::vector<float> bpc = vec(sd.get(p + ".proj_edge.0.bias", 128), 128);
      center_outputs(Wpc, bpc, 128, 128);
      B.add(k + ".WAp", pack_afrag(Wpc.data(), 128));
      B.add(k + ".WBp", pack_bfrag(Wpc, 128));
      B.add(k + ".WLp", pack_bfrag_lo3(Wpc, 128));
      put(VT_BP, bpc.data());
      put(VT_GP, sd.get(p + ".proj_edge.1.weight", 128));
      put(VT_BEP, sd.get(p + ".proj_edge.1.bias", 128));
      put(VT_GE, sd.get(p + ".norm_edge.weight", 128));
      put(VT_BE, sd.get(p + ".norm_edge.bias", 128));
    }
    B.add(k + ".vtab", vt);
    const float *Win = sd.get(p + ".multihead_attn.in_proj_weight", 384 * 128);
    const float *bin = sd.get(p + ".multihead_attn.in_proj_bias", 384);
    {
      // the token kernel's projections as fp32 MFMA A fragments (k_token_mfma)
      const float *Wo_ = sd.get(p + ".multihead_attn.out_proj.weight", 128 * 128);
      const float *W1_ = sd.get(p + ".linear1.weight", 256 * 128), *W2_ = sd.get(p + ".linear2.weight", 128 * 256);
      B.add(k + ".WsM", pack_afrag(Wm + 128, 384));
      B.add(k + ".WtM", pack_afrag(Wm + 256, 384));
      B.add(k + ".WqM", pack_afrag(Win, 128));
      B.add(k + ".WkM", pack_wk_frag(Win ? Win + 128 * 128 : nullptr));
      B.add(k + ".WvM", pack_afrag(Win ? Win + 2 * 128 * 128 : nullptr, 128));
      B.add(k + ".WoM", pack_afrag(Wo_, 128));
      B.add(k + ".W1aM", pack_afrag(W1_, 128));
      B.add(k + ".W1bM", pack_afrag(W1_ ? W1_ + 128 * 128 : nullptr, 128));
      B.add(k + ".W2aM", pack_afrag(W2_, 256));
      B.add(k + ".W2bM", pack_afrag(W2_ ? W2_ + 128 : nullptr, 256));
      // ... and as bf16 hi / lo A fragments (k_token_mfma<1>; the folded K query keeps the fp32 fragments)
      B.add(k + ".WsB", pack_tok_bfrag(Wm + 128, 384));
      B.add(k + ".WtB", pack_tok_bfrag(Wm + 256, 384));
      B.add(k + ".WqB", pack_tok_bfrag(Win, 128));
      B.add(k + ".WvB", pack_tok_bfrag(Win ? Win + 2 * 128 * 128 : nullptr, 128));
      B.add(k + ".WoB", pack_tok_bfrag(Wo_, 128));
      B.add(k + ".W1aB", pack_tok_bfrag(W1_, 128));
      B.add(k + ".W1bB", pack_tok_bfrag(W1_ ? W1_ + 128 * 128 : nullptr, 128));
      B.add(k + ".W2aB", pack_tok_bfrag(W2_, 256));
      B.add(k + ".W2bB", pack_tok_bfrag(W2_ ? W2_ + 128 : nullptr, 256));
    }
    B.add(k + ".WqT", transpose(Win, 128, 128));
    B.add(k + ".Wk", vec(Win ? Win + 128 * 128 : nullptr, 128 * 128));
    B.add(k + ".WvT", transpose(Win ? Win + 2 * 128 * 128 : nullptr, 128, 128));
    B.add(k + ".bq", vec(bin, 128));
    B.add(k + ".bv", vec(bin ? bin + 256 : nullptr, 128));
    B.add(k + ".WoT", transpose(sd.get(p + ".multihead_attn.out_proj.weight", 128 * 128), 128, 128));
    B.add(k + ".bo", vec(sd.get(p + ".multihead_attn.out_proj.bias", 128), 128));
    B.add(k + ".W1T", transpose(sd.get(p + ".linear1.weight", 256 * 128), 256, 128));
    B.add(k + ".b1", vec(sd.get(p + ".linear1.bias", 256), 256));
    B.add(k + ".W2T", transpose(sd.get(p + ".linear2.weight", 128 * 256), 128, 256));
    B.add(k + ".b2", vec(sd.get(p + ".linear2.bias", 128), 128));
    B.add(k + ".g2", vec(sd.get(p + ".norm2.weight", 128), 128));
    B.add(k + ".be2", vec(sd.get(p + ".norm2.bias", 128), 128));
    B.add(k + ".g3", vec(sd.get(p + ".norm3.weight", 128), 128));
    B.add(k + ".be3", vec(sd.get(p + ".norm3.bias", 128), 128));
  }
  // ---- pred_scene
  lin_ln("dec.rpe", "pred_scene.proj_rpe", 0, 128, 20);
  lin_ln("dec.t0", "pred_scene.proj_tgt", 0, 128, 256);
  lin_ln("dec.t3", "pred_scene.proj_tgt", 3, 128, 128);
  lin_ln("dec.c0", "pred_scene.ctx_proj", 0, 384, 128);
  lin_ln("dec.c3", "pred_scene.ctx_proj", 3, 768, 384);
  lin_ln("dec.a0", "pred_scene.actor_proj", 0, 384, 128);
  lin_ln("dec.a3", "pred_scene.actor_proj", 3, 768, 384);
  B.add("dec.a0.WB", pack_conv_frag(sd.get("pred_scene.actor_proj.0.weight", 384 * 128), 384, 128, 1, 128));
  B.add("dec.a3.WB", pack_conv_frag(sd.get("pred_scene.actor_proj.3.weight", 768 * 384), 768, 384, 1, 384));
  B.add("dec.reg0.WB", pack_conv_frag(sd.get("pred_scene.reg.0.weight", 128 * 128), 128, 128, 1, 128));
  B.add("dec.reg3.WB", pack_conv_frag(sd.get("pred_scene.reg.3.weight", 128 * 128), 128, 128, 1, 128));
  B.add("dec.reg6.WB", pack_conv_frag(sd.get("pred_scene.reg.6.weight", 40 * 128), 40, 128, 1, 128, 48));
  {
    std::vector<float> b6 = vec(sd.get("pred_scene.reg.6.bias", 40), 40);
    b6.resize(48, 0.f);
    B.add("dec.reg6.bB", b6);
  }
  for (int L = 0; L < 2; ++L) {
    std::string p = "pred_scene.ctx_sat.layers." + std::to_string(L), k = "dec.e" + std::to_string(L);
    B.add(k + ".inW", transpose(sd.get(p + ".self_attn.in_proj_weight", 384 * 128), 384, 128));
    B.add(k + ".inb", vec(sd.get(p + ".self_attn.in_proj_bias", 384), 384));
    B.add(k + ".outW", transpose(sd.get(p + ".self_attn.out_proj.weight", 128 * 128), 128, 128));
    B.add(k + ".outb", vec(sd.get(p + ".self_attn.out_proj.bias", 128), 128));
    B.add(k + ".l1W", transpose(sd.get(p + ".linear1.weight", 1536 * 128), 1536, 128));
    B.add(k + ".l1b", vec(sd.get(p + ".linear1.bias", 1536), 1536));
    B.add(k + ".l2W", transpose(sd.get(p + ".linear2.weight", 128 * 1536), 128, 1536));
    B.add(k + ".l2b", vec(sd.get(p + ".linear2.bias", 128), 128));
    B.add(k + ".n1g", vec(sd.get(p + ".norm1.weight", 128), 128));
    B.add(k + ".n1b", vec(sd.get(p + ".norm1.bias", 128), 128));
    B.add(k + ".n2g", vec(sd.get(p + ".norm2.weight", 128), 128));
    B.add(k + ".n2b", vec(sd.get(p + ".norm2.bias", 128), 128));
  }
  for (const char *hn : {"cls", "reg"}) {
    std::string p = std::string("pred_scene.") + hn, k = std::string("dec.") + hn;
    lin_ln(k + "0", p, 0, 128, 128);
    lin_ln(k + "3", p, 3, 128, 128);
    const int nl = strcmp(hn, "cls") == 0 ? 1 : 40;
    B.add(k + "6.W", transpose(sd.get(p + ".6.weight", (int64_t)nl * 128), nl, 128));
    B.add(k + "6.b", vec(sd.get(p + ".6.bias", nl), nl));
  }
  {
    // Bezier basis (network.py:449-464): float64 numpy -> fp32 (Q21)
    std::vector<float> T(60 * 8), Tp(60 * 7);
    auto comb = [](int n, int k) { double r = 1; for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i; return r; };
    for (int t = 0; t < 60; ++t) {
      double ts = (t == 59) ? 1.0 : (double)t * (1.0 / 59.0);
      // numpy linspace: start + arange * step, step = 1/59; last point exactly 1
      ts = (t == 59) ? 1.0 : 0.0 + (double)t * (1.0 / 59.0);
      for (int i = 0; i < 8; ++i) T[t * 8 + i] = (float)(comb(7, i) * std::pow(1.0 - ts, 7 - i) * std::pow(ts, i));
      for (int i = 0; i < 7; ++i) Tp[t * 7 + i] = (float)(7.0 * comb(6, i) * std::pow(1.0 - ts, 6 - i) * std::pow(ts, i));
    }
    B.add("dec.T", T);
    B.add("dec.Tp", Tp);
  }
  if (!sd.missing.empty()) return fail(c, MIND_ENOTFOUND, "state_dict tensor missing or wrong size: %s", sd.missing.c_str());

  if (c->wdev) (void)hipFree(c->wdev);
  c->wdev = nullptr;
  HIPCHK(c, hipMalloc((void **)&c->wdev, B.host.size() * sizeof(float)));
  HIPCHK(c, hipMemcpy(c->wdev, B.host.data(), B.host.size() * sizeof(float), hipMemcpyHostToDevice));
  auto P = [&](const std::string &k) -> const float * {
    auto it = B.off.find(k);
    return it == B.off.end() ? nullptr : c->wdev + it->second;
  };
  // ---- pointer tables
  LaneW &lw = c->laneW;
  lw.pW = P("lane.proj.W"); lw.pb = P("lane.proj.b"); lw.pg = P("lane.proj.g"); lw.pbe = P("lane.proj.be");
  for (int b = 0; b < 2; ++b) {
    std::string k = "lane.a" + std::to_string(b);
    lw.f10W[b] = P(k + ".f10.W"); lw.f10b[b] = P(k + ".f10.b"); lw.f10g[b] = P(k + ".f10.g"); lw.f10be[b] = P(k + ".f10.be");
    lw.f13W[b] = P(k + ".f13.W"); lw.f13b[b] = P(k + ".f13.b"); lw.f13g[b] = P(k + ".f13.g"); lw.f13be[b] = P(k + ".f13.be");
    lw.f20W[b] = P(k + ".f20.W"); lw.f20b[b] = P(k + ".f20.b"); lw.f20g[b] = P(k + ".f20.g"); lw.f20be[b] = P(k + ".f20.be");
    lw.f23W[b] = P(k + ".f23.W"); lw.f23b[b] = P(k + ".f23.b"); lw.f23g[b] = P(k + ".f23.g"); lw.f23be[b] = P(k + ".f23.be");
    lw.ng[b] = P(k + ".ng"); lw.nbe[b] = P(k + ".nb");
  }
  ActorW &aw = c->actorW;
  for (int r = 0; r < 9; ++r) {
    std::string k = "act.r" + std::to_string(r);
    aw.res[r].c1 = P(k + ".c1"); aw.res[r].g1 = P(k + ".g1"); aw.res[r].b1 = P(k + ".b1");
    aw.res[r].c2 = P(k + ".c2"); aw.res[r].g2 = P(k + ".g2"); aw.res[r].b2 = P(k + ".b2");
    aw.res[r].ds = P(k + ".ds"); aw.res[r].gd = P(k + ".gd"); aw.res[r].bd = P(k + ".bd");
  }
  for (int g = 0; g < 4; ++g) {
    std::string k = "act.lat" + std::to_string(g);
    aw.latW[g] = P(k + ".W"); aw.latG[g] = P(k + ".g"); aw.latB[g] = P(k + ".b");
  }
  AmW &bw = c->actorBW;
  for (int r = 0; r < 9; ++r) {
    std::string k = "act.r" + std::to_string(r);
    AmRes &R = bw.res[r];
    R.c1 = (const u32 *)P(k + ".c1B"); R.c2 = (const u32 *)P(k + ".c2B"); R.ds = (const u32 *)P(k + ".dsB");
    R.g1 = aw.res[r].g1; R.b1 = aw.res[r].b1; R.g2 = aw.res[r].g2; R.b2 = aw.res[r].b2;
    R.gd = aw.res[r].gd; R.bd = aw.res[r].bd;
  }
  for (int g = 0; g < 4; ++g) {
    bw.lat[g].w = (const u32 *)P("act.lat" + std::to_string(g) + ".WB");
    bw.lat[g].g = aw.latG[g]; bw.lat[g].b = aw.latB[g];
  }
  AfW &fw = c->actorFW;
  for (int r = 0; r < 9; ++r) {
    std::string k = "act.r" + std::to_string(r);
    AfRes &R = fw.res[r];
    R.c1 = P(k + ".c1F"); R.c2 = P(k + ".c2F"); R.ds = P(k + ".dsF");
    R.g1 = aw.res[r].g1; R.b1 = aw.res[r].b1; R.g2 = aw.res[r].g2; R.b2 = aw.res[r].b2;
    R.gd = aw.res[r].gd; R.bd = aw.res[r].bd;
  }
  for (int g = 0; g < 4; ++g) {
    fw.lat[g].w = P("act.lat" + std::to_string(g) + ".WF");
    fw.lat[g].g = aw.latG[g]; fw.lat[g].b = aw.latB[g];
  }
  c->rtab = P("fus.rtab");
  for (int L = 0; L < 6; ++L) {
    std::string k = "fus.L" + std::to_string(L);
    c->WAe[L] = P(k + ".WAe");
    c->WAp[L] = P(k + ".WAp");
    c->WBe[L] = (const u32 *)P(k + ".WBe");
    c->WBp[L] = (const u32 *)P(k + ".WBp");
    c->WLe[L] = (const u32 *)P(k + ".WLe");
    c->WLp[L] = (const u32 *)P(k + ".WLp");
    c->vtab[L] = P(k + ".vtab");
  }
  for (int L = 0; L <= 6; ++L) {
    TokWeights &w = c->tokW[L];
    memset(&w, 0, sizeof(w));
    if (L >= 1) {
      std::string k = "fus.L" + std::to_string(L - 1);
      w.WvT = P(k + ".WvT"); w.bv = P(k + ".bv"); w.WoT = P(k + ".WoT"); w.bo = P(k + ".bo");
      w.g2 = P(k + ".g2"); w.b2 = P(k + ".be2"); w.W1T = P(k + ".W1T"); w.b1 = P(k + ".b1");
      w.W2T = P(k + ".W2T"); w.bb2 = P(k + ".b2"); w.g3 = P(k + ".g3"); w.b3 = P(k + ".be3");
    }
    if (L <= 5) {
      std::string k = "fus.L" + std::to_string(L);
      w.WsT = P(k + ".WsT"); w.WtT = P(k + ".WtT"); w.bm = P(k + ".bm");
      w.WqT = P(k + ".WqT"); w.bq = P(k + ".bq"); w.Wk = P(k + ".Wk");
    }
    w.WpaT = P("fus.pa.W"); w.bpa = P("fus.pa.b"); w.gpa = P("fus.pa.g"); w.bepa = P("fus.pa.be");
    w.WplT = P("fus.pl.W"); w.bpl = P("fus.pl.b"); w.gpl = P("fus.pl.g"); w.bepl = P("fus.pl.be");
    TokWeightsM &m = c->tokWM[L];
    memset(&m, 0, sizeof(m));
    if (L >= 1) {
      std::string k = "fus.L" + std::to_string(L - 1);
      m.Wv = P(k + ".WvM"); m.Wo = P(k + ".WoM"); m.W1a = P(k + ".W1aM"); m.W1b = P(k + ".W1bM"); m.W2a = P(k + ".W2aM"); m.W2b = P(k + ".W2bM");
    }
    if (L <= 5) {
      std::string k = "fus.L" + std::to_string(L);
      m.Ws = P(k + ".WsM"); m.Wt = P(k + ".WtM"); m.Wq = P(k + ".WqM"); m.Wkf = P(k + ".WkM");
    }
    m.Wpa = P("fus.pa.WM"); m.Wpl = P("fus.pl.WM");
    TokWeightsM &bm = c->tokWB[L];
    memset(&bm, 0, sizeof(bm));
    if (L >= 1) {
      std::string k = "fus.L" + std::to_string(L - 1);
      bm.Wv = P(k + ".WvB"); bm.Wo = P(k + ".WoB"); bm.W1a = P(k + ".W1aB"); bm.W1b = P(k + ".W1bB"); bm.W2a = P(k + ".W2aB"); bm.W2b = P(k + ".W2bB");
    }
    if (L <= 5) {
      std::string k = "fus.L" + std::to_string(L);
      bm.Ws = P(k + ".WsB"); bm.Wt = P(k + ".WtB"); bm.Wq = P(k + ".WqB"); bm.Wkf = P(k + ".WkM");
    }
    bm.Wpa = P("fus.pa.WB"); bm.Wpl = P("fus.pl.WB");
  }
  DecW &d = c->decW;
  d.rpeW = P("dec.rpe.W"); d.rpeb = P("dec.rpe.b"); d.rpeg = P("dec.rpe.g"); d.rpebe = P("dec.rpe.be");
  d.t0W = P("dec.t0.W"); d.t0b = P("dec.t0.b"); d.t0g = P("dec.t0.g"); d.t0be = P("dec.t0.be");
  d.t3W = P("dec.t3.W"); d.t3b = P("dec.t3.b"); d.t3g = P("dec.t3.g"); d.t3be = P("dec.t3.be");
  d.c0W = P("dec.c0.W"); d.c0b = P("dec.c0.b"); d.c0g = P("dec.c0.g"); d.c0be = P("dec.c0.be");
  d.c3W = P("dec.c3.W"); d.c3b = P("dec.c3.b"); d.c3g = P("dec.c3.g"); d.c3be = P("dec.c3.be");
  d.a0W = P("dec.a0.W"); d.a0b = P("dec.a0.b"); d.a0g = P("dec.a0.g"); d.a0be = P("dec.a0.be");
  d.a3W = P("dec.a3.W"); d.a3b = P("dec.a3.b"); d.a3g = P("dec.a3.g"); d.a3be = P("dec.a3.be");
  for (int L = 0; L < 2; ++L) {
    std::string k = "dec.e" + std::to_string(L);
    d.inW[L] = P(k + ".inW"); d.inb[L] = P(k + ".inb"); d.outW[L] = P(k + ".outW"); d.outb[L] = P(k + ".outb");
    d.l1W[L] = P(k + ".l1W"); d.l1b[L] = P(k + ".l1b"); d.l2W[L] = P(k + ".l2W"); d.l2b[L] = P(k + ".l2b");
    d.n1g[L] = P(k + ".n1g"); d.n1b[L] = P(k + ".n1b"); d.n2g[L] = P(k + ".n2g"); d.n2b[L] = P(k + ".n2b");
  }
  d.k0W = P("dec.cls0.W"); d.k0b = P("dec.cls0.b"); d.k0g = P("dec.cls0.g"); d.k0be = P("dec.cls0.be");
  d.k3W = P("dec.cls3.W"); d.k3b = P("dec.cls3.b"); d.k3g = P("dec.cls3.g"); d.k3be = P("dec.cls3.be");
  d.k6W = P("dec.cls6.W"); d.k6b = P("dec.cls6.b");
  d.r0W = P("dec.reg0.W"); d.r0b = P("dec.reg0.b"); d.r0g = P("dec.reg0.g"); d.r0be = P("dec.reg0.be");
  d.r3W = P("dec.reg3.W"); d.r3b = P("dec.reg3.b"); d.r3g = P("dec.reg3.g"); d.r3be = P("dec.reg3.be");
  d.r6W = P("dec.reg6.W"); d.r6b = P("dec.reg6.b");
  d.T = P("dec.T"); d.Tp = P("dec.Tp");
  DmW &m = c->decBW;
  m.a0 = (const u32 *)P("dec.a0.WB"); m.a3 = (const u32 *)P("dec.a3.WB"); m.r0 = (const u32 *)P("dec.reg0.WB");
  m.r3 = (const u32 *)P("dec.reg3.WB"); m.r6 = (const u32 *)P("dec.reg6.WB");
  m.a0b = d.a0b; m.a0g = d.a0g; m.a0be = d.a0be; m.a3b = d.a3b; m.a3g = d.a3g; m.a3be = d.a3be;
  m.r0b = d.r0b; m.r0g = d.r0g; m.r0be = d.r0be; m.r3b = d.r3b; m.r3g = d.r3g; m.r3be = d.r3be; m.r6b = P("dec.reg6.bB");
  m.T = d.T; m.Tp = d.Tp;
  c->have_weights = true;
  return MIND_OK;
}

// -------------------------------------------------------------------------------------------------
// predictor forward
// -------------------------------------------------------------------------------------------------
__global__ void k_tokpos(const TokMeta *__restrict__ meta, int n_tok, const float *__restrict__ actr,
                         const float *__restrict__ avec, const float *__restrict__ lctr,
                         const float *__restrict__ lvec, float *__restrict__ tokpos) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_tok) return;
  const TokMeta m = meta[t];
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  if (m.type == 0 && actr) o = make_float4(actr[m.src * 2], actr[m.src * 2 + 1], avec[m.src * 2], avec[m.src * 2 + 1]);
  if (m.type == 1 && lctr) o = make_float4(lctr[m.src * 2], lctr[m.src * 2 + 1], lvec[m.src * 2], lvec[m.src * 2 + 1]);
  ((float4 *)tokpos)[t] = o;
}

extern "C" int mind_predict_batch(mind_ctx *c, const mind_scene_batch *in, mind_pred_out *out) {
  if (!c || !in || !out) return MIND_EINVAL;
  if (!c->have_weights) return fail(c, MIND_ESTATE, "weights not loaded");
  const int Bn = in->n_scenes;
  if (Bn <= 0 || !in->actor_off || !in->lane_off || !in->actors || !in->tgt_nodes || !in->tgt_rpe || !out->cls ||
      !out->reg || !out->vel)
    return fail(c, MIND_EINVAL, "null input/output pointer");
  if (!in->lanes && !in->lane_feat) return fail(c, MIND_EINVAL, "need lanes or lane_feat");
  if (!in->rpe && !(in->actor_ctrs && in->actor_vecs && in->lane_ctrs && in->lane_vecs))
    return fail(c, MIND_EINVAL, "need rpe or ctrs/vecs");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t st = c->stream;
  const int A = in->actor_off[Bn], Ltot = in->lane_off[Bn];
  if (A <= 0 || Ltot < 0) return fail(c, MIND_EINVAL, "empty batch");

  // ---- host tables: tokens, jobs (cached by the batch's scene sizes)
  std::vector<int> key;
  key.reserve(2 * Bn + 3);
  key.push_back(Bn);
  for (int b = 0; b <= Bn; ++b) key.push_back(in->actor_off[b]);
  for (int b = 0; b <= Bn; ++b) key.push_back(in->lane_off[b]);
  TableSet *ts = nullptr;
  for (TableSet &t : c->tabs)
    if (t.key == key) ts = &t;
  const bool tab_hit = ts != nullptr;
  if (!ts) {
    ts = &c->tabs[0];
    for (TableSet &t : c->tabs)
      if (t.stamp < ts->stamp) ts = &t;         // least recently used (empty sets have stamp 0)
  }
  ts->stamp = ++c->tab_clock;
  int rc;
  if (!tab_hit) {
    std::vector<TokMeta> meta;
    std::vector<PairJob> jobs;
    std::vector<int> actor_row(A), actor_scene(A), cls_row(Bn), scene_n(Bn);
    long long edge_pairs = 0, edge_pairs_t = 0;
    int ntok = 0;
    for (int b = 0; b < Bn; ++b) {
      const int a = in->actor_off[b + 1] - in->actor_off[b], l = in->lane_off[b + 1] - in->lane_off[b];
      if (a <= 0 || l < 0) return fail(c, MIND_EINVAL, "scene %d has %d agents, %d lanes", b, a, l);
    }
    int slot = 0;
    double pairs_full = 0, pairs_l5 = 0;
    for (int b = 0; b < Bn; ++b) {
      const int a = in->actor_off[b + 1] - in->actor_off[b], l = in->lane_off[b + 1] - in->lane_off[b];
      const int N = a + l + 1;
      const int tiles = (N + 15) / 16;
      const int ns = pair_column_splits(N);      // (a function of the scene's own size only: pair_jobs.h)
      for (int j = 0; j < N; ++j) {
        TokMeta m;
        memset(&m, 0, sizeof(m));
        m.type = j < a ? 0 : (j < a + l ? 1 : 2);
        m.src = j < a ? in->actor_off[b] + j : (j < a + l ? in->lane_off[b] + (j - a) : 0);
        m.slot0 = slot;
        m.nsplit = ns;
        m.flags = (j < a || j == N - 1) ? 1 : 0;
        meta.push_back(m);
        for (int s_ = 0; s_ < ns; ++s_) {
          PairJob J;
          memset(&J, 0, sizeof(J));
          J.edge_base = edge_pairs;
          J.edge_base_t = edge_pairs_t;
          J.N = N;
          J.j = j;
          pair_job_range(tiles, ns, s_, &J.t0, &J.t1);
          J.tok_base = ntok;
          J.slot = slot++;
          J.flags = m.flags;
          J.scene = b;
          jobs.push_back(J);
        }
        if (j < a) { actor_row[in->actor_off[b] + j] = ntok + j; actor_scene[in->actor_off[b] + j] = b; }
      }
      cls_row[b] = ntok + N - 1;
      scene_n[b] = N;
      ntok += N;
      edge_pairs += (long long)N * N;
      edge_pairs_t += (long long)N * tiles * 16;
      pairs_full += (double)N * N;
      pairs_l5 += (double)N * (a + 1);
    }
    // The order of a job list is its schedule (pair_jobs.h): equal work per wave slot, and for batches of eight scenes or more all column
    // jobs of a scene on one XCD.  The last fusion layer runs the consumed columns only (actors + cls): k_pair_t walks a list of its own
    // instead of skipping the other jobs after a dependent load each.
    std::vector<PairJob> jobs5;
    for (const PairJob &J : jobs)
      if (J.flags & 1) jobs5.push_back(J);
    auto deal = [&](std::vector<PairJob> &jl) {
      const int grid_ = (int)jl.size() < c->n_cu ? (int)jl.size() : c->n_cu;
      pair_jobs_deal(jl, grid_, PAIR_WAVES, (c->xcd_order && Bn >= 8 && grid_ % 8 == 0) ? 8 : 1);
    };
    deal(jobs);
    deal(jobs5);
    ts->key.clear();                    // invalid until the upload below has completed
    if ((rc = ensure(c, ts->meta, meta.size() * sizeof(TokMeta)))) return rc;
    if ((rc = ensure(c, ts->jobs, jobs.size() * sizeof(PairJob)))) return rc;
    if ((rc = ensure(c, ts->jobs5, jobs5.size() * sizeof(PairJob)))) return rc;
    if ((rc = ensure(c, ts->rows, (size_t)(2 * A + Bn) * sizeof(int)))) return rc;
    HIPCHK(c, hipMemcpyAsync(ts->meta.p, meta.data(), meta.size() * sizeof(TokMeta), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(ts->jobs.p, jobs.data(), jobs.size() * sizeof(PairJob), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(ts->jobs5.p, jobs5.data(), jobs5.size() * sizeof(PairJob), hipMemcpyHostToDevice, st));
    std::vector<int> rows(2 * A + Bn);
    memcpy(rows.data(), actor_row.data(), A * sizeof(int));
    memcpy(rows.data() + A, actor_scene.data(), A * sizeof(int));
    memcpy(rows.data() + 2 * A, cls_row.data(), Bn * sizeof(int));
    HIPCHK(c, hipMemcpyAsync(ts->rows.p, rows.data(), rows.size() * sizeof(int), hipMemcpyHostToDevice, st));
    // the host vectors above must outlive the async copies
    HIPCHK(c, hipStreamSynchronize(st));
    ts->actor_row.swap(actor_row);
    ts->cls_row.swap(cls_row);
    ts->scene_n.swap(scene_n);
    ts->edge_pairs = edge_pairs; ts->edge_pairs_t = edge_pairs_t; ts->ntok = ntok; ts->slot = slot; ts->njobs = (int)jobs.size();
    ts->njobs5 = (int)jobs5.size();
    ts->pairs_full = pairs_full; ts->pairs_l5 = pairs_l5;
    ts->key.swap(key);
  } else {
    c->n_table_hits++;
  }
  const std::vector<int> &actor_row = ts->actor_row, &cls_row = ts->cls_row;
  const long long edge_pairs = ts->edge_pairs;
  const int ntok = ts->ntok, slot = ts->slot, njobs = ts->njobs;
  const bool tiled = c->pair_prec == 3 || (c->pair_prec != 0 && c->pair_tile);      // k_pair_t / k_pair_t6: the edge tensor in its tile-native layout
  const double pairs_full = ts->pairs_full, pairs_l5 = ts->pairs_l5;

  // ---- workspaces
  // (k_pair_t<*, 1> keeps the tensor in bf16: the fp32-sized buffer is simply half used)
  if ((rc = ensure(c, c->edge, (size_t)(tiled ? ts->edge_pairs_t : edge_pairs) * 128 * sizeof(float)))) return rc;
  if ((rc = ensure(c, c->x, (size_t)ntok * 128 * sizeof(float)))) return rc;
  if ((rc = ensure(c, c->ST, (size_t)ntok * 256 * sizeof(float)))) return rc;
  const size_t qk_stride = c->pair_prec == 3 ? P6_QK_STRIDE : 1024;      // dwords of folded query per token (three parts under bf16x6)
  if ((rc = ensure(c, c->QK, (size_t)(ntok + 1) * qk_stride * sizeof(float)))) return rc;      // (+ 1: k_pair_t6's padding rows read past the last record)
  if ((rc = ensure(c, c->part, (size_t)slot * PART_STRIDE * sizeof(float)))) return rc;
  if ((rc = ensure(c, c->tokpos, (size_t)ntok * 4 * sizeof(float)))) return rc;
  if ((rc = ensure(c, c->actor_feat, (size_t)A * 128 * sizeof(float)))) return rc;
  if ((rc = ensure(c, c->lane_feat, (size_t)(Ltot > 0 ? Ltot : 1) * 128 * sizeof(float)))) return rc;
  if ((rc = ensure(c, c->tgt_feat, (size_t)Bn * 128 * sizeof(float)))) return rc;
  if ((rc = ensure(c, c->cmode, (size_t)Bn * 768 * sizeof(float)))) return rc;
  if ((rc = ensure(c, c->tgt_emb, (size_t)Bn * 128 * sizeof(float)))) return rc;
  const float *const *rpe_dev = nullptr;
  if (in->rpe) {
    if ((rc = ensure(c, c->rpe_ptrs, (size_t)Bn * sizeof(float *)))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->rpe_ptrs.p, in->rpe, (size_t)Bn * sizeof(float *), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipStreamSynchronize(st));          // in->rpe is the caller's host array
    rpe_dev = (const float *const *)c->rpe_ptrs.p;
  }

  const TokMeta *dmeta = (const TokMeta *)ts->meta.p;
  const PairJob *djobs = (const PairJob *)ts->jobs.p;
  float *x = (float *)c->x.p, *ST = (float *)c->ST.p, *QK = (float *)c->QK.p, *part = (float *)c->part.p;
  float *edge = (float *)c->edge.p, *tokpos = (float *)c->tokpos.p;
  float *actor_feat = (float *)c->actor_feat.p;
  const float *lane_feat = in->lane_feat;

  // ---- encoders: ActorNet on the context stream, the (independent) lane encoders and token positions beside it on
  //      the side stream (everything they read was complete at the synchronisation point above)
  hipStream_t ss = c->side ? c->side : st;
  if (c->side) {
    // the side stream reads inputs the caller produced on the context stream (uploads, mind_aime_rebase outputs): it starts
    // behind everything queued there so far (a table-cache hit no longer synchronises the stream on the way in)
    HIPCHK(c, hipEventRecord(c->ev_main, st));
    HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_main, 0));
  }
  // ActorNet: fp32 VALU kernel under MIND_PAIR_F32, the bf16-split / bf16 MFMA kernel otherwise (the precision setting covers
  // every MFMA contraction of the predictor)
  // ... and the fp32-MFMA kernel (plain fp32 operands on the matrix core: the reference's arithmetic class): the ActorNet of the exact-fp32
  // setting, and -- two actors per workgroup -- of every setting on full-tree rounds (thousands of actors per call)
  const bool f32_pair = c->actor_f32 && A >= (c->pair_prec == 0 ? c->actor_f32_pair_min : c->actor_f32_min);
  if (f32_pair)
    hipLaunchKernelGGL(k_actor_f32<2>, dim3((A + 1) / 2), dim3(AF_T), mind_actor_f32_lds_bytes(2), st, in->actors, A, actor_feat, c->actorFW);
  else if (c->pair_prec == 0 && c->actor_f32 && c->enc_mfma)
    hipLaunchKernelGGL(k_actor_f32<1>, dim3(A), dim3(AF_T), mind_actor_f32_lds_bytes(1), st, in->actors, A, actor_feat, c->actorFW);
  else if (c->pair_prec == 0 || !c->enc_mfma)
    hipLaunchKernelGGL(k_actor_net, dim3(A), dim3(AT), mind_actor_lds_bytes(), st, in->actors, A, actor_feat, c->actorW);
  else if (c->pair_prec == 3 || (c->pair_prec == 1 && c->actor_np == 6))
    hipLaunchKernelGGL(k_actor_mfma<6>, dim3(A), dim3(AM_T), mind_actor_mfma_lds_bytes(), st, in->actors, A, actor_feat, c->actorBW);
  else if (c->pair_prec == 1)
    hipLaunchKernelGGL(k_actor_mfma<3>, dim3(A), dim3(AM_T), mind_actor_mfma_lds_bytes(), st, in->actors, A, actor_feat, c->actorBW);
  else
    hipLaunchKernelGGL(k_actor_mfma<1>, dim3(A), dim3(AM_T), mind_actor_mfma_lds_bytes(), st, in->actors, A, actor_feat, c->actorBW);
  if (!lane_feat) {
    float *lf = out->lane_feat ? out->lane_feat : (float *)c->lane_feat.p;
    if (Ltot > 0)
      hipLaunchKernelGGL(k_lane_net, dim3((Ltot + PL - 1) / PL), dim3(DT), 0, ss, in->lanes, Ltot, lf, c->laneW);
    lane_feat = lf;
  } else if (out->lane_feat && out->lane_feat != in->lane_feat && Ltot > 0) {
    HIPCHK(c, hipMemcpyAsync(out->lane_feat, in->lane_feat, (size_t)Ltot * 128 * sizeof(float), hipMemcpyDeviceToDevice, ss));
  }
  hipLaunchKernelGGL(k_tokpos, dim3((ntok + 255) / 256), dim3(256), 0, ss, dmeta, ntok, in->actor_ctrs, in->actor_vecs,
                     in->lane_ctrs, in->lane_vecs, tokpos);
  if (c->side) {
    HIPCHK(c, hipEventRecord(c->ev_side, c->side));
    HIPCHK(c, hipStreamWaitEvent(st, c->ev_side, 0));
  }
  // the target polyline's encoder + embedding feed the decoder only: they stay on the side stream while the fusion layers run
  hipLaunchKernelGGL(k_lane_net, dim3((Bn + PL - 1) / PL), dim3(DT), 0, ss, in->tgt_nodes, Bn, (float *)c->tgt_feat.p, c->laneW);
  hipLaunchKernelGGL(k_dec_tgt, dim3(Bn), dim3(DT), mind_dec_scene_lds_bytes(), ss, (const float *)c->tgt_feat.p, in->tgt_rpe, (float *)c->tgt_emb.p,
                     c->decW);
  if (c->side) HIPCHK(c, hipEventRecord(c->ev_tgt, c->side));
  if (c->side && !c->tgt_side) HIPCHK(c, hipStreamWaitEvent(st, c->ev_tgt, 0));

  // ---- fusion: init tokens + 6 x (pair kernel, token kernel)
  // The token kernel by SCENE (a scene's result must not depend on what else is in its batch): scenes of at least tok_bf_min_n tokens
  // under the bf16 pair-kernel arithmetics run k_token_mfma<1> (bf16 hi + lo split operands on the MFMA, 16 tokens per workgroup: the
  // VALU kernel is LDS-bound at those sizes), every other scene the VALU kernel -- consecutive scenes of one class share a launch, small
  // launches take four tokens per workgroup (more workgroups, half the LDS operand traffic each; bit-identical to eight)
  const int qsplit = c->pair_prec == 3 ? 48 : c->pair_prec != 0 ? 16 : 0;      // the bf16 pair kernels read the folded query as hi / lo (bf16x6: hi / mid / lo) fragments
  const size_t tokm_lds = mind_token_mfma_lds_bytes();
  struct TokRun { int t0, n, kind; };                  // kind 0: VALU, 1: fp32 MFMA (opt-in), 2: bf16 split MFMA
  std::vector<TokRun> tok_runs;
  {
    int t0 = 0;
    for (int b = 0; b < Bn; ++b) {
      const int N = (in->actor_off[b + 1] - in->actor_off[b]) + (in->lane_off[b + 1] - in->lane_off[b]) + 1;
      // (the two-way-split token kernel is not an fp32-class arithmetic: never under bf16x6)
      const int kind = c->tok_mfma ? 1 : (c->pair_prec != 0 && c->pair_prec != 3 && c->tok_bf_min_n > 0 && N >= c->tok_bf_min_n) ? 2 : 0;
      if (!tok_runs.empty() && tok_runs.back().kind == kind) tok_runs.back().n += N;
      else tok_runs.push_back({t0, N, kind});
      t0 += N;
    }
  }
  auto launch_tokens = [&](int mode, int Lw) {
    for (const TokRun &r : tok_runs) {
      const TokMeta *m_ = dmeta + r.t0;
      float *x_ = x + (size_t)r.t0 * 128, *ST_ = ST + (size_t)r.t0 * 256, *QK_ = QK + (size_t)r.t0 * qk_stride;
      if (r.kind == 0) {
        const bool small = r.n <= c->tok_small_max;
        const int tpw = small ? TOK_TPW_SMALL : TOK_TPW_BIG;
        // (small launches: independent projections merged, k_token_m -- bit-identical; "tok_merge" 0 keeps the plain form for A/B)
        hipLaunchKernelGGL(small ? (c->tok_merge ? k_token_m : k_token<TOK_TPW_SMALL>) : k_token<TOK_TPW_BIG>, dim3((r.n + tpw - 1) / tpw), dim3(TT_THREADS), 0,
                           st, m_, r.n, mode, actor_feat, lane_feat, x_, part, ST_, QK_, c->tokW[Lw]);
      } else if (r.kind == 1) {
        hipLaunchKernelGGL(k_token_mfma<0>, dim3((r.n + TM_TOK - 1) / TM_TOK), dim3(TM_THREADS), tokm_lds, st, m_, r.n, mode, actor_feat, lane_feat, x_,
                           part, ST_, QK_, c->tokW[Lw], c->tokWM[Lw]);
      } else {
        hipLaunchKernelGGL(k_token_mfma<1>, dim3((r.n + TM_TOK - 1) / TM_TOK), dim3(TM_THREADS), tokm_lds, st, m_, r.n, mode, actor_feat, lane_feat, x_,
                           part, ST_, QK_, c->tokW[Lw], c->tokWB[Lw]);
      }
    }
  };
  launch_tokens(1 | 4 | qsplit, 0);
  int grid = njobs < c->n_cu ? njobs : c->n_cu;      // jobs are dealt wave-major over the workgroups
  const size_t lds = mind_pair_lds_bytes();
  c->n_pair_launch = 0;
  c->pairs_done = 0;
  if (c->profiling && c->ev.size() < 12) {
    while (c->ev.size() < 12) {
      hipEvent_t e;
      HIPCHK(c, hipEventCreate(&e));
      c->ev.push_back(e);
    }
  }
  hipEvent_t *evs = c->ev.data();
  if (c->profiling && c->ev_defer) {
    while (c->ev_pool.size() < c->ev_pool_used + 12) {
      hipEvent_t e;
      HIPCHK(c, hipEventCreate(&e));
      c->ev_pool.push_back(e);
    }
    evs = c->ev_pool.data() + c->ev_pool_used;
    c->ev_pending.push_back(c->ev_pool_used);
    c->ev_pool_used += 12;
  }
  c->last_ntok = ntok; c->last_edge_pairs = edge_pairs; c->last_slots = slot; c->last_A = A; c->last_B = Bn;
  c->last_scene_n = ts->scene_n; c->last_edge_tiled = tiled; c->last_edge_bf16 = tiled && c->pair_prec == 2;
  for (int L = 0; L < c->debug_layers; ++L) {
    const int um = L < 4 ? 0 : (L == 4 ? 1 : 2);
    if (c->profiling) HIPCHK(c, hipEventRecord(evs[2 * L], st));
    if (c->pair_prec == 0) {
      if (L == 0)
        hipLaunchKernelGGL(k_pair<0>, dim3(grid), dim3(PAIR_THREADS), lds, st, djobs, njobs, edge, ST, QK, part, c->WAe[L], c->WAp[L],
                           c->vtab[L], c->rtab, tokpos, rpe_dev, um);
      else
        hipLaunchKernelGGL(k_pair<1>, dim3(grid), dim3(PAIR_THREADS), lds, st, djobs, njobs, edge, ST, QK, part, c->WAe[L],
                           L == 5 ? c->WAe[L] : c->WAp[L], c->vtab[L], c->rtab, tokpos, rpe_dev, um);
    } else if (tiled) {
      const size_t ldsb = mind_pair_bf_lds_bytes();
      const u32 *we = c->WBe[L], *wp = L == 5 ? c->WBe[L] : c->WBp[L];
      const PairJob *jl = L == 5 ? (const PairJob *)ts->jobs5.p : djobs;
      const int nj = L == 5 ? ts->njobs5 : njobs;
      const int grid_t = nj < c->n_cu ? nj : c->n_cu;
#define LAUNCH_T(M, NPV)                                                                                                       \
  hipLaunchKernelGGL((k_pair_t<M, NPV>), dim3(grid_t), dim3(PAIR_THREADS), ldsb, st, jl, nj, edge, ST, QK, part, we, wp,       \
                     c->vtab[L], c->rtab, tokpos, rpe_dev, um)
      if (c->pair_prec == 3) {
        const u32 *wle = c->WLe[L], *wlp = L == 5 ? c->WLe[L] : c->WLp[L];
        if (L == 0)
          hipLaunchKernelGGL(k_pair_t6<0>, dim3(grid_t), dim3(PAIR_THREADS), ldsb, st, jl, nj, edge, ST, QK, part, we, wp, wle, wlp, c->vtab[L], c->rtab,
                             tokpos, rpe_dev, um);
        else
          hipLaunchKernelGGL(k_pair_t6<1>, dim3(grid_t), dim3(PAIR_THREADS), ldsb, st, jl, nj, edge, ST, QK, part, we, wp, wle, wlp, c->vtab[L], c->rtab,
                             tokpos, rpe_dev, um);
      } else if (c->pair_prec == 1) { if (L == 0) LAUNCH_T(0, 3); else LAUNCH_T(1, 3); }
      else { if (L == 0) LAUNCH_T(0, 1); else LAUNCH_T(1, 1); }
#undef LAUNCH_T
    } else {
      const size_t ldsb = mind_pair_bf_lds_bytes();
      const u32 *we = c->WBe[L], *wp = L == 5 ? c->WBe[L] : c->WBp[L];
#define LAUNCH_BF(M, NPV)                                                                                                       \
  hipLaunchKernelGGL((k_pair_bf<M, NPV>), dim3(grid), dim3(PAIR_THREADS), ldsb, st, djobs, njobs, edge, ST, QK, part, we, wp, \
                     c->vtab[L], c->rtab, tokpos, rpe_dev, um)
      if (c->pair_prec == 1) { if (L == 0) LAUNCH_BF(0, 3); else LAUNCH_BF(1, 3); }
      else { if (L == 0) LAUNCH_BF(0, 1); else LAUNCH_BF(1, 1); }
#undef LAUNCH_BF
    }
    if (c->profiling) HIPCHK(c, hipEventRecord(evs[2 * L + 1], st));
    c->n_pair_launch++;
    c->pairs_done += (L == 5) ? pairs_l5 : pairs_full;
    const int mode = 2 | (L < 5 ? 4 : 8) | qsplit;
    launch_tokens(mode, L + 1);
  }
  // ---- decoder
  const int *d_actor_row = (const int *)ts->rows.p;
  const int *d_actor_scene = d_actor_row + A;
  const int *d_cls_row = d_actor_row + 2 * A;
  // the actor part's first half (actor_proj: 85 % of its weights) needs only the fused actor tokens: it runs on the side stream
  // beside k_dec_scene, the head follows both on the context stream (bit-identical to the one-kernel form)
  const bool fp32_dec = c->pair_prec == 0 || !c->enc_mfma || A < c->dec_mfma_min;
  const bool split_dec = fp32_dec && c->side && c->dec_overlap;
  if (split_dec) {
    if ((rc = ensure(c, c->dec_h2, (size_t)A * 768 * sizeof(float)))) return rc;
    HIPCHK(c, hipEventRecord(c->ev_main, st));
    HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_main, 0));
    hipLaunchKernelGGL(k_dec_actor<1>, dim3((A + RA - 1) / RA), dim3(DT), mind_dec_actor_lds_bytes(), c->side, x, d_actor_row, d_actor_scene, A,
                       (const float *)nullptr, (const float *)nullptr, (float *)nullptr, (float *)nullptr, c->decW, (float *)c->dec_h2.p);
    HIPCHK(c, hipEventRecord(c->ev_side, c->side));
  }
  // the scene part: eight workgroups per scene while the whole launch is resident (one workgroup per CU: 158 KB of LDS), else one per scene --
  // the two kernels give the same bits, so a scene's result does not depend on the size of its batch
  bool cls_on_side = false;
  const int mw_blocks = ((Bn + 7) / 8) * 8 * DEC_MW_G;
  bool mw = c->dec_mw && mw_blocks <= c->n_cu;
  if (mw) {
    if (c->dec_abort && *(volatile unsigned *)c->dec_abort) return fail(c, MIND_EHIP, "k_dec_scene_mw: a barrier of an earlier launch timed out (launch not resident)");
    if (!c->dec_abort) {
      if (hipHostMalloc((void **)&c->dec_abort, 64, hipHostMallocMapped) != hipSuccess) { c->dec_abort = nullptr; mw = false; }
      else *c->dec_abort = 0u;
    }
    const size_t need_x = (size_t)(c->n_cu / DEC_MW_G) * 2 * 6 * 1536 * sizeof(float), need_b = (size_t)(c->n_cu / DEC_MW_G) * 4 * sizeof(unsigned);
    if (mw && c->dec_xbuf.cap < need_x) {
      if ((rc = ensure(c, c->dec_xbuf, need_x))) return rc;
    }
    if (mw && c->dec_bars.cap < need_b) {
      if ((rc = ensure(c, c->dec_bars, need_b))) return rc;
      HIPCHK(c, hipMemsetAsync(c->dec_bars.p, 0, c->dec_bars.cap, st));        // (once: the barrier resets its arrival count itself)
    }
  }
  if (mw)
    hipLaunchKernelGGL(k_dec_scene_mw, dim3(mw_blocks), dim3(DT), mind_dec_scene_lds_bytes(), st, x, d_cls_row, (float *)c->cmode.p, out->cls, c->decW, Bn,
                       (float *)c->dec_xbuf.p, (unsigned *)c->dec_bars.p, c->dec_abort);
  else if (split_dec && c->dec_cls_side) {
    // the mode tokens on the context stream, the mode probabilities (the cls head: ~10 us a launch) on the side stream beside the actor part's
    // head, which needs the tokens only; the caller's next work on the context stream follows both
    hipLaunchKernelGGL(k_dec_scene_c, dim3(Bn), dim3(DT), mind_dec_scene_lds_bytes(), st, x, d_cls_row, (float *)c->cmode.p, c->decW);
    HIPCHK(c, hipEventRecord(c->ev_main, st));
    HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_main, 0));
    hipLaunchKernelGGL(k_dec_cls, dim3(Bn), dim3(DT), mind_dec_scene_lds_bytes(), c->side, (float *)c->cmode.p, out->cls, c->decW);
    if (!c->ev_cls) HIPCHK(c, hipEventCreateWithFlags(&c->ev_cls, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(c->ev_cls, c->side));
    cls_on_side = true;
  } else
    hipLaunchKernelGGL(k_dec_scene, dim3(Bn), dim3(DT), mind_dec_scene_lds_bytes(), st, x, d_cls_row, (float *)c->cmode.p, out->cls, c->decW);
  if (c->side) HIPCHK(c, hipStreamWaitEvent(st, c->ev_tgt, 0));      // the decoder's actor part reads the target embedding
  // actor part of the decoder: the K-split fp32 kernel (a handful of workgroups, bound by the latency of one pass over the weights:
  // 62 us at 40 agents), or -- opt-in, mind_set_tuning("dec_mfma_min") -- the MFMA kernel (16 agents per workgroup: 104 us at 40
  // agents, 209 vs 277 us at 13.8 k).  Off by default: a plan's result must not depend on what else is in the batch.
  if (split_dec) {
    HIPCHK(c, hipStreamWaitEvent(st, c->ev_side, 0));
    hipLaunchKernelGGL(k_dec_actor<2>, dim3((A + RA - 1) / RA), dim3(DT), mind_dec_actor_lds_bytes(), st, x, d_actor_row, d_actor_scene, A,
                       (const float *)c->cmode.p, (const float *)c->tgt_emb.p, out->reg, out->vel, c->decW, (float *)c->dec_h2.p);
  } else if (fp32_dec)
    hipLaunchKernelGGL(k_dec_actor<0>, dim3((A + RA - 1) / RA), dim3(DT), mind_dec_actor_lds_bytes(), st, x, d_actor_row, d_actor_scene, A,
                       (const float *)c->cmode.p, (const float *)c->tgt_emb.p, out->reg, out->vel, c->decW, (float *)nullptr);
  else if (c->pair_prec == 3 || (c->pair_prec == 1 && c->actor_np == 6))
    hipLaunchKernelGGL(k_dec_actor_mfma<6>, dim3((A + DM_RA - 1) / DM_RA), dim3(DM_T), mind_dec_actor_mfma_lds_bytes(), st, x, d_actor_row,
                       d_actor_scene, A, (const float *)c->cmode.p, (const float *)c->tgt_emb.p, out->reg, out->vel, c->decBW);
  else if (c->pair_prec == 1)
    hipLaunchKernelGGL(k_dec_actor_mfma<3>, dim3((A + DM_RA - 1) / DM_RA), dim3(DM_T), mind_dec_actor_mfma_lds_bytes(), st, x, d_actor_row,
                       d_actor_scene, A, (const float *)c->cmode.p, (const float *)c->tgt_emb.p, out->reg, out->vel, c->decBW);
  else
    hipLaunchKernelGGL(k_dec_actor_mfma<1>, dim3((A + DM_RA - 1) / DM_RA), dim3(DM_T), mind_dec_actor_mfma_lds_bytes(), st, x, d_actor_row,
                       d_actor_scene, A, (const float *)c->cmode.p, (const float *)c->tgt_emb.p, out->reg, out->vel, c->decBW);
  if (cls_on_side) HIPCHK(c, hipStreamWaitEvent(st, c->ev_cls, 0));      // whatever follows on the context stream sees the mode probabilities too
  if (out->actor_emb || out->cls_emb) {
    // debug taps: gather fused tokens
    for (int a = 0; a < A && out->actor_emb; ++a)
      HIPCHK(c, hipMemcpyAsync(out->actor_emb + (size_t)a * 128, x + (size_t)actor_row[a] * 128, 128 * sizeof(float),
                               hipMemcpyDeviceToDevice, st));
    for (int b = 0; b < Bn && out->cls_emb; ++b)
      HIPCHK(c, hipMemcpyAsync(out->cls_emb + (size_t)b * 128, x + (size_t)cls_row[b] * 128, 128 * sizeof(float),
                               hipMemcpyDeviceToDevice, st));
  }
  HIPCHK(c, hipGetLastError());
  if (c->profiling && c->ev_defer) {
    c->pair_ms = 0.f;         // (read by mind_pair_events_resolve behind the plan's last synchronisation)
  } else if (c->profiling) {
    HIPCHK(c, hipStreamSynchronize(st));
    c->pair_ms = 0.f;
    for (int L = 0; L < c->debug_layers; ++L) {
      float ms = 0.f;
      HIPCHK(c, hipEventElapsedTime(&ms, c->ev[2 * L], c->ev[2 * L + 1]));
      c->pair_ms += ms;
    }
  }
  return MIND_OK;
}

// -------------------------------------------------------------------------------------------------
// tree-iLQR host side
// -------------------------------------------------------------------------------------------------
// grid coordinates exactly as numpy builds them (ilqr/utils.py:7-13): linspace(0, size, n) + offset
static void il_make_grid(int W, int H, double res, const double *ego_xy, double *gx, double *gy, double &offx, double &offy) {
  const double fsx = (double)(W - 1) * res, fsy = (double)(H - 1) * res;
  offx = ego_xy[0] - 0.5 * fsx; offy = ego_xy[1] - 0.5 * fsy;
  const double sx = fsx / (double)(W - 1), sy = fsy / (double)(H - 1);
  for (int i = 0; i < W; ++i) gx[i] = (double)i * sx + 0.0;
  gx[W - 1] = fsx;
  for (int i = 0; i < H; ++i) gy[i] = (double)i * sy + 0.0;
  gy[H - 1] = fsy;
  for (int i = 0; i < W; ++i) gx[i] += offx;
  for (int i = 0; i < H; ++i) gy[i] += offy;
}

// gen_dist_field (ilqr/utils.py:5-22): distance of every grid centroid to the polyline
extern "C" int mind_lane_dist_field(mind_ctx *c, const double *ego_xy, const double *lane, int n_pts, int W, int H,
                                    double res, double *offset, double *gx, double *gy, double *dist) {
  if (!c || !ego_xy || !lane || n_pts < 2 || W < 2 || H < 2 || !(res > 0) || !offset || !gx || !gy || !dist)
    return fail(c, MIND_EINVAL, "mind_lane_dist_field: bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t st = c->stream;
  il_make_grid(W, H, res, ego_xy, gx, gy, offset[0], offset[1]);
  const size_t nd = (size_t)W + H + 2 * (size_t)n_pts + (size_t)W * H;
  int rc;
  if ((rc = ensure(c, c->ilqr_dev, nd * sizeof(double)))) return rc;
  double *d = (double *)c->ilqr_dev.p;
  HIPCHK(c, hipMemcpyAsync(d, gx, W * sizeof(double), hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(d + W, gy, H * sizeof(double), hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(d + W + H, lane, 2 * (size_t)n_pts * sizeof(double), hipMemcpyHostToDevice, st));
  double *out = d + W + H + 2 * (size_t)n_pts;
  hipLaunchKernelGGL(k_lane_field, dim3((W * H + 255) / 256), dim3(256), 0, st, d, d + W, W, H, d + W + H, n_pts, out, 0);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(dist, out, (size_t)W * H * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  return MIND_OK;
}

struct IlqrEvalReq { int nq; const int32_t *node; const double *x, *u; double *out; };
namespace { int pl_pin(mind_ctx *c, int which, size_t bytes); }     // page-locked staging buffers of the context (aime_plan.hip)

// Upload of a few ten KB from the context's page-locked staging (hipHostMalloc: mapped into the device's address space) as a KERNEL on the
// consumer's own queue: the device reads the host buffer over PCIe (a few us) and the consumer follows back to back.  hipMemcpyAsync sends
// copies above its blit threshold to the SDMA engine, whose hand-over to the compute queue stood 10-15 us on either side of the copy in the
// plan's timeline (root scene: 60 KB, the solver's tables: 40 KB; profiles/r06az_timeline.txt).  16-byte words; large uploads stay copies.
typedef unsigned up_u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_upload(const up_u4 *__restrict__ src, up_u4 *__restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = __builtin_nontemporal_load(src + i);
}
static int pl_upload(mind_ctx *c, void *dst, const void *pinned_src, size_t bytes, hipStream_t s) {
  if (!bytes) return MIND_OK;
  if (c->upload_kernel_max > 0 && bytes <= (size_t)c->upload_kernel_max && bytes % 16 == 0 && (uintptr_t)dst % 16 == 0 && (uintptr_t)pinned_src % 16 == 0) {
    const size_t n16 = bytes / 16;
    hipLaunchKernelGGL(k_upload, dim3((unsigned)std::min<size_t>((n16 + 255) / 256, 256)), dim3(256), 0, s, (const up_u4 *)pinned_src, (up_u4 *)dst, n16);
    HIPCHK(c, hipGetLastError());
    return MIND_OK;
  }
  HIPCHK(c, hipMemcpyAsync(dst, pinned_src, bytes, hipMemcpyHostToDevice, s));
  return MIND_OK;
}

// Shared host side of mind_ilqr_solve_trees / mind_ilqr_solve_fields / mind_cost_eval: build the device
// arena, then either run the solver (ev == nullptr) or evaluate node costs at the requested points.
// grid != nullptr selects the generic mode (materialised per-node fields + per-node weights).
// the field of the NEXT tree-iLQR call on this context, ahead of it: grid + lane up, k_lane_field on `s`, an event behind it
static int il_field_prepare(mind_ctx *c, const mind_ilqr_cfg *cfg, const double *x0, const double *lane, int n_lane_pts, hipStream_t s) {
  c->il_field_valid = false;
  const int W = cfg->grid_w, H = cfg->grid_h;
  if (W < 3 || H < 3 || n_lane_pts < 2 || !(cfg->grid_res > 0)) return MIND_OK;        // (the call itself reports bad arguments)
  const size_t nin = (size_t)W + H + 2 * (size_t)n_lane_pts, nd = nin + (size_t)W * H;
  int rc;
  if ((rc = ensure(c, c->il_field, nd * sizeof(double)))) return rc;
  if (c->il_field_pin_cap < nin * sizeof(double)) {
    if (c->il_field_pin) (void)hipHostFree(c->il_field_pin);
    c->il_field_pin = nullptr; c->il_field_pin_cap = 0;
    if (hipHostMalloc(&c->il_field_pin, 2 * nin * sizeof(double), hipHostMallocDefault) != hipSuccess) { c->il_field_pin = nullptr; return MIND_OK; }
    c->il_field_pin_cap = 2 * nin * sizeof(double);
  }
  if (!c->ev_field) HIPCHK(c, hipEventCreateWithFlags(&c->ev_field, hipEventDisableTiming));
  double *h = (double *)c->il_field_pin, ox, oy;
  il_make_grid(W, H, cfg->grid_res, x0, h, h + W, ox, oy);
  memcpy(h + W + H, lane, 2 * (size_t)n_lane_pts * sizeof(double));
  double *d = (double *)c->il_field.p;
  HIPCHK(c, hipMemcpyAsync(d, h, nin * sizeof(double), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_lane_field, dim3((W * H + 255) / 256), dim3(256), 0, s, d, d + W, W, H, d + W + H, n_lane_pts, d + nin);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipEventRecord(c->ev_field, s));
  c->il_field_key[0] = x0[0]; c->il_field_key[1] = x0[1]; c->il_field_key[2] = W; c->il_field_key[3] = H; c->il_field_key[4] = cfg->grid_res;
  c->il_field_lane.assign(lane, lane + 2 * (size_t)n_lane_pts);
  c->il_field_valid = true;
  return MIND_OK;
}

static int ilqr_impl(mind_ctx *c, const mind_ilqr_cfg *cfg, const mind_field_grid *grid, const mind_cost_tree *trees, int n_trees,
                     const double *x0, const double *target_lane, int n_lane_pts, double target_vel,
                     int use_exo, const double *us_init, double *xs, double *us,
                     mind_ilqr_stats *stats, const IlqrEvalReq *ev,
                     const mind_ilqr_cfg *cfg2 = nullptr, mind_ilqr_stats *stats2 = nullptr) {
  // cfg2 != nullptr: two fits in one launch -- (cfg, lane term only) then, from its controls, (cfg2, full cost)
  const bool gen = grid != nullptr;
  const int n_phases = cfg2 ? 2 : 1;
  const int use_exo_first = cfg2 ? 0 : use_exo;
  if (cfg2) use_exo = 1;
  if (!c || !cfg || !trees || n_trees <= 0 || !x0) return fail(c, MIND_EINVAL, "iLQR: bad argument");
  if (c->il_finish) return fail(c, MIND_ESTATE, "a tree-iLQR call begun with mind_ilqr_contingency_begin has not been finished (mind_ilqr_finish)");
  if (!ev && (!xs || !us)) return fail(c, MIND_EINVAL, "iLQR: null output");
  if (!gen && (!target_lane || n_lane_pts < 2)) return fail(c, MIND_EINVAL, "iLQR: target lane needs >= 2 points");
  if (gen) { use_exo = 0; n_lane_pts = 0; }
  const int W = gen ? grid->W : cfg->grid_w, H = gen ? grid->H : cfg->grid_h;
  if (W < 3 || H < 3 || cfg->max_iter < 0) return fail(c, MIND_EINVAL, "bad grid / max_iter");
  if (gen && (!grid->gx || !grid->gy || !(grid->res > 0))) return fail(c, MIND_EINVAL, "bad field grid");
  for (int t = 0; t < n_trees; ++t)
    if (gen != (trees[t].field != nullptr) || gen != (trees[t].node_w != nullptr))
      return fail(c, MIND_EINVAL, "tree %d: field / node_w must be given exactly in the generic (grid) mode", t);
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t st = c->stream;
  // MIND_PLAN_TRACE=1: host time stamps of this call's sections on stderr (as in mind_aime_plan)
  static const bool il_trace = getenv("MIND_PLAN_TRACE") != nullptr;
  const auto il_t0 = std::chrono::steady_clock::now();
  auto ITR = [&](const char *what) {
    if (il_trace) fprintf(stderr, "[ilqr] %8.1f us  %s\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - il_t0).count(), what);
  };
  const double grid_res = gen ? grid->res : cfg->grid_res;
  // ---- grid coordinates exactly as numpy builds them (ilqr/utils.py:7-13)
  std::vector<double> gx(W), gy(H);
  const double fsx = (double)(W - 1) * grid_res, fsy = (double)(H - 1) * grid_res;
  const double offx = gen ? grid->off_x : x0[0] - 0.5 * fsx, offy = gen ? grid->off_y : x0[1] - 0.5 * fsy;
  if (gen) {
    memcpy(gx.data(), grid->gx, W * sizeof(double));
    memcpy(gy.data(), grid->gy, H * sizeof(double));
  } else {
    double ox, oy;
    il_make_grid(W, H, grid_res, x0, gx.data(), gy.data(), ox, oy);
  }
  // (mind_ilqr_contingency_begin_plan: the trees' agent arrays are the plan's own device buffers, tree t at node offset pl_tree_off[t])
  const bool dev_flat = c->il_use_dev_flat && !gen && !ev && c->pl_dev_fmean && c->pl_dev_fcov;
  c->il_use_dev_flat = false;
  // ---- launch mode: one workgroup per tree | wide trees: G workgroups share a tree's items | narrow trees: G workgroups take a tree's LM slots
  int maxM = 0;
  for (int t = 0; t < n_trees; ++t) maxM = trees[t].n_nodes > maxM ? trees[t].n_nodes : maxM;
  int G = c->ilqr_wgs;
  // trees of tens of thousands of nodes (the deep stress trees: 29.5 k) keep twice the workgroups busy: 55 -> 35 ms per launch at 32 per tree,
  // while the cfg4 trees (6.5 k nodes) are best at 16-24 (profiles/r03an_*)
  if (G > 1 && G < c->ilqr_wgs_big && maxM >= c->ilqr_big_min) G = c->ilqr_wgs_big;
  while (G > 1 && ((n_trees + 7) / 8) * 8 * G > c->n_cu) G >>= 1;     // every workgroup of the launch must be resident (1 per CU)
  const bool multi = !gen && !ev && G > 1 && maxM >= c->ilqr_multi_min;
  int GS = multi ? 1 : c->ilqr_slots;
  // (c->ilqr_wgs == 1 is the caller's "stay on few CUs": the speculative warm start beside the predictor)
  if (gen || ev || c->ilqr_wgs <= 1) GS = 1;
  while (GS > 1 && ((n_trees + 7) / 8) * 8 * GS > c->n_cu) --GS;
  const bool slots = GS > 1;
  // the derivative speculator: one more workgroup per tree, only where the whole launch stays resident with it
  const int spec = slots && c->ilqr_spec_deriv && GS < 31 && ((n_trees + 7) / 8) * 8 * (GS + 1) <= c->n_cu ? 1 : 0;
  // sets of per-slot arrays (gains, value functions, candidates: ~146 doubles per node and slot): what this launch can use -- the followers' slots
  // (GS, after the residency loop above) or the master's own speculation (IL_SPEC), not IL_SLOTS for every narrow-tree launch
  const size_t nslot = slots ? (size_t)(GS > IL_SPEC ? GS : IL_SPEC) : IL_SPEC;
  // ---- layout of one device arena: [doubles | floats | ints | tree structs]
  size_t nd = 0, nf = 0, ni = 0;
  auto takeD = [&](size_t n) { size_t o = nd; nd += (n + 1) & ~(size_t)1; return o; };
  auto takeF = [&](size_t n) { size_t o = nf; nf += (n + 3) & ~(size_t)3; return o; };
  auto takeI = [&](size_t n) { size_t o = ni; ni += (n + 3) & ~(size_t)3; return o; };
  // the doubles region starts with everything the host uploads (grid, lane, queries, initial controls, per-node
  // weights); the workspace behind `nd_in` is produced by the kernels and never copied from the host
  const size_t o_gx = takeD(W), o_gy = takeD(H), o_lane = takeD((size_t)n_lane_pts * 2 + 2);
  const size_t o_evx = takeD(ev ? (size_t)ev->nq * 6 : 0), o_evu = takeD(ev ? (size_t)ev->nq * 2 : 0);
  const size_t o_evn = takeI(ev ? ev->nq : 0);
  const size_t o_bars = takeI(4 * (size_t)n_trees + 4);   // barrier words of the multi-workgroup launch + its abort word (zero at upload)
  const size_t ctl_ints = (sizeof(IlSlotCtl) + 15) / 16 * 4;
  const size_t o_ctl = takeI(slots ? ctl_ints * (size_t)n_trees : 0);      // slot control blocks (zero at upload; 16-byte aligned: the ints region is)
  const int trace_cap = std::min(256, std::max(cfg->max_iter, cfg2 ? cfg2->max_iter : 0));      // rows of the per-iteration trace, per phase
  struct TL { size_t nodew, field, relag, relag2, Fx2, L2, Lx2, Lxx2, rel2; size_t xs, us, Fx, L, Lx, Lxx, k, K, Vx, Vxx, xsn, usn, Ln, stats, prob, mean, cov, parent, lstart, lnodes, cstart, clist, sstart, snodes, slstart, slsegs, segrec, rel, fsstart, fsq0, fsq1, fsnstart, fsnodes, trace; int M, a, nl, nseg, nsl, maxls, nfs; };
  std::vector<TL> tl(n_trees);
  long Mtot = 0;
  for (int t = 0; t < n_trees; ++t) tl[t].us = takeD(2 * (size_t)(trees[t].n_nodes > 0 ? trees[t].n_nodes : 0));
  for (int t = 0; t < n_trees; ++t) tl[t].nodew = takeD(gen ? (size_t)(trees[t].n_nodes > 0 ? trees[t].n_nodes : 0) * IL_NW : 0);
  const size_t nd_in = nd;
  const size_t o_quad = takeD(gen ? 2 : (size_t)W * H), o_evo = takeD(ev ? (size_t)ev->nq * IL_EVAL_OUT : 0);
  // results of all trees are contiguous (us already is: it lives in the upload region), so they come back in three copies
  for (int t = 0; t < n_trees; ++t) tl[t].xs = takeD(6 * (size_t)(trees[t].n_nodes > 0 ? trees[t].n_nodes : 0));
  for (int t = 0; t < n_trees; ++t) tl[t].stats = takeD(2 * IL_NSTAT);
  for (int t = 0; t < n_trees; ++t) {
    const mind_cost_tree &tr = trees[t];
    if (tr.n_nodes <= 0 || !tr.parent || (!gen && (!tr.prob || tr.n_agents <= 0)) || (use_exo && !dev_flat && (!tr.agent_mean || !tr.agent_cov)))
      return fail(c, MIND_EINVAL, "tree %d: bad arrays", t);
    if (tr.n_agents > IL_MAXA) return fail(c, MIND_EINVAL, "tree %d: %d agents > %d supported", t, tr.n_agents, IL_MAXA);
    const size_t M = tr.n_nodes;
    TL &L = tl[t];
    L.M = (int)M; L.a = gen ? 1 : tr.n_agents;
    L.trace = takeD((size_t)2 * trace_cap * IL_TRACE_W);
    // what the derivative pass writes, and (derivative speculator) a second set laid out alike right behind it
    L.relag = takeD(use_exo ? M * IL_RA : 0); L.Fx = takeD(36 * M); L.L = takeD(M); L.Lx = takeD(6 * M); L.Lxx = takeD(36 * M);
    L.relag2 = takeD(spec && use_exo ? M * IL_RA : 0); L.Fx2 = takeD(spec ? 36 * M : 0); L.L2 = takeD(spec ? M : 0); L.Lx2 = takeD(spec ? 6 * M : 0);
    L.Lxx2 = takeD(spec ? 36 * M : 0);
    L.k = takeD(nslot * 2 * M); L.K = takeD(nslot * 12 * M); L.Vx = takeD(nslot * 6 * M); L.Vxx = takeD(nslot * 36 * M);
    L.xsn = takeD(nslot * 60 * M); L.usn = takeD(nslot * 20 * M); L.Ln = takeD(nslot * 10 * M);
    L.prob = takeF(M); L.mean = takeF(dev_flat ? 0 : M * L.a * 2); L.cov = takeF(dev_flat ? 0 : M * L.a);
    L.parent = takeI(M); L.lnodes = takeI(M); L.cstart = takeI(M + 1); L.clist = takeI(M); L.rel = takeI(M); L.rel2 = takeI(spec ? M : 0);
    Mtot += (long)M;
  }
  // levels need the depth first
  // (the vectors live in the context: their capacity survives the call)
  for (auto &v : c->il_scr.vv) if ((int)v.size() < n_trees) v.resize(n_trees);
  auto &lvl_start = c->il_scr.vv[0], &lvl_nodes = c->il_scr.vv[1], &cst = c->il_scr.vv[2], &cls = c->il_scr.vv[3];
  auto &sg_start = c->il_scr.vv[4], &sg_nodes = c->il_scr.vv[5], &sl_start = c->il_scr.vv[6], &sl_segs = c->il_scr.vv[7];
  auto &seg_rec = c->il_scr.vv[8], &fs_start = c->il_scr.vv[9], &fs_q0 = c->il_scr.vv[10], &fs_q1 = c->il_scr.vv[11], &fs_nstart = c->il_scr.vv[12], &fs_nodes = c->il_scr.vv[13];
  auto &depth = c->il_scr.tmp[0], &fill = c->il_scr.tmp[1], &cf = c->il_scr.tmp[2], &seg_of = c->il_scr.tmp[3], &seg_depth = c->il_scr.tmp[4], &sf = c->il_scr.tmp[5];
  for (int t = 0; t < n_trees; ++t) {
    const mind_cost_tree &tr = trees[t];
    const int M = tr.n_nodes;
    depth.assign(M, 0);
    int maxd = 0;
    for (int i = 0; i < M; ++i) {
      const int p = tr.parent[i];
      if (i == 0 ? p != -1 : (p < 0 || p >= i)) return fail(c, MIND_EINVAL, "tree %d: node %d has parent %d", t, i, p);
      depth[i] = i == 0 ? 0 : depth[p] + 1;
      maxd = depth[i] > maxd ? depth[i] : maxd;
    }
    tl[t].nl = maxd + 1;
    lvl_start[t].assign(maxd + 2, 0);
    for (int i = 0; i < M; ++i) lvl_start[t][depth[i] + 1]++;
    for (int d = 0; d <= maxd; ++d) lvl_start[t][d + 1] += lvl_start[t][d];
    lvl_nodes[t].resize(M);
    fill.assign(maxd + 1, 0);
    for (int i = 0; i < M; ++i) lvl_nodes[t][lvl_start[t][depth[i]] + fill[depth[i]]++] = i;
    cst[t].assign(M + 1, 0);
    for (int i = 1; i < M; ++i) cst[t][tr.parent[i] + 1]++;
    for (int i = 0; i < M; ++i) cst[t][i + 1] += cst[t][i];
    cls[t].assign(M > 1 ? M : 1, 0);
    cf.assign(M, 0);
    for (int i = 1; i < M; ++i) cls[t][cst[t][tr.parent[i]] + cf[tr.parent[i]]++] = i;
    tl[t].lstart = takeI(maxd + 2);
    // chain segments: a node starts a segment if it is node 0 or its parent has >= 2 children
    seg_of.assign(M, -1); seg_depth.clear();
    auto nchild = [&](int i) { return cst[t][i + 1] - cst[t][i]; };
    sg_start[t].clear(); sg_nodes[t].clear();
    for (int i = 0; i < M; ++i) {
      if (!(i == 0 || nchild(tr.parent[i]) >= 2)) continue;
      const int sidx = (int)sg_start[t].size();
      sg_start[t].push_back((int)sg_nodes[t].size());
      seg_depth.push_back(i == 0 ? 0 : seg_depth[seg_of[tr.parent[i]]] + 1);
      int c = i;
      while (true) {
        seg_of[c] = sidx;
        sg_nodes[t].push_back(c);
        if (nchild(c) != 1) break;
        c = cls[t][cst[t][c]];
      }
    }
    sg_start[t].push_back((int)sg_nodes[t].size());
    const int nseg = (int)seg_depth.size();
    int maxsd = 0;
    for (int d : seg_depth) maxsd = d > maxsd ? d : maxsd;
    sl_start[t].assign(maxsd + 2, 0);
    for (int d : seg_depth) sl_start[t][d + 1]++;
    for (int d = 0; d <= maxsd; ++d) sl_start[t][d + 1] += sl_start[t][d];
    sl_segs[t].resize(nseg);
    sf.assign(maxsd + 1, 0);
    for (int sgi = 0; sgi < nseg; ++sgi) sl_segs[t][sl_start[t][seg_depth[sgi]] + sf[seg_depth[sgi]]++] = sgi;
    tl[t].nseg = nseg; tl[t].nsl = maxsd + 1;
    tl[t].maxls = 1;
    for (int d = 0; d <= maxsd; ++d) tl[t].maxls = std::max(tl[t].maxls, sl_start[t][d + 1] - sl_start[t][d]);
    tl[t].sstart = takeI(nseg + 1); tl[t].snodes = takeI(M); tl[t].slstart = takeI(maxsd + 2); tl[t].slsegs = takeI(nseg);
    // segment record of the backward sweep (16 ints): positions [s0, s1) of seg_nodes, last / first node, the node before the last, the
    // last node's child count, where its children start in child_list and the first six of them -- one round trip instead of the walk
    // segment -> positions -> node -> child range -> children
    seg_rec[t].assign((size_t)nseg * 16, 0);
    for (int sgi = 0; sgi < nseg; ++sgi) {
      int *r = seg_rec[t].data() + (size_t)sgi * 16;
      const int s0 = sg_start[t][sgi], s1 = sg_start[t][sgi + 1], cl = sg_nodes[t][s1 - 1];
      r[0] = s0; r[1] = s1; r[2] = cl; r[3] = sg_nodes[t][s0]; r[4] = sg_nodes[t][s1 - 2 >= s0 ? s1 - 2 : s1 - 1];
      r[5] = nchild(cl); r[6] = cst[t][cl];
      for (int e = 0; e < 6 && e < r[5]; ++e) r[8 + e] = cls[t][cst[t][cl] + e];
    }
    tl[t].segrec = takeI((size_t)nseg * 16);
    // forward steps of the line search: the segments of a level, cut into chunks of ilqr_chunk nodes when only a few chains run
    // side by side (the other waves then price the nodes the previous step reached); wide levels stay whole
    const int chunk = (c->ilqr_chunk > 0 && tl[t].maxls <= 6) ? c->ilqr_chunk : M;
    fs_start[t].assign(1, 0); fs_nstart[t].assign(1, 0);
    fs_q0[t].clear(); fs_q1[t].clear(); fs_nodes[t].clear();
    for (int d = 0; d <= maxsd; ++d) {
      int maxlen = 0;
      for (int e = sl_start[t][d]; e < sl_start[t][d + 1]; ++e) { const int sgi = sl_segs[t][e]; maxlen = std::max(maxlen, sg_start[t][sgi + 1] - sg_start[t][sgi]); }
      for (int k0 = 0; k0 < maxlen; k0 += chunk) {
        for (int e = sl_start[t][d]; e < sl_start[t][d + 1]; ++e) {
          const int sgi = sl_segs[t][e], q0 = sg_start[t][sgi] + k0, q1 = std::min(sg_start[t][sgi + 1], q0 + chunk);
          if (q0 >= q1) continue;
          // item record: positions [q0, q1) of seg_nodes, its first two nodes and the first node's parent (the rollout's prologue
          // would otherwise walk q0 -> node -> parent -> state through four dependent loads)
          const int c0 = sg_nodes[t][q0], c1 = sg_nodes[t][q0 + 1 < q1 ? q0 + 1 : q1 - 1];
          for (int v : {q0, q1, c0, c1, c0 == 0 ? -1 : tr.parent[c0], 0, 0, 0}) fs_q0[t].push_back(v);
          fs_q1[t].push_back(q1);
          for (int q = q0; q < q1; ++q) fs_nodes[t].push_back(sg_nodes[t][q]);
        }
        fs_start[t].push_back((int)fs_q1[t].size()); fs_nstart[t].push_back((int)fs_nodes[t].size());
      }
    }
    tl[t].nfs = (int)fs_start[t].size() - 1;
    tl[t].fsstart = takeI(fs_start[t].size()); tl[t].fsq0 = takeI(fs_q0[t].size()); tl[t].fsq1 = takeI(fs_q1[t].size());
    tl[t].fsnstart = takeI(fs_nstart[t].size()); tl[t].fsnodes = takeI(M);
  }
  ITR("tables built");
  // arena: [uploaded doubles (nd_in) | floats | ints | tree structs | constants] = ONE host->device copy, then the doubles the
  // kernels produce (workspace + results), then (generic mode) the materialised fields
  const size_t bytesIn = nd_in * sizeof(double), bytesF = nf * sizeof(float), bytesI = ni * sizeof(int);
  const size_t o_structs = bytesIn + bytesF + bytesI;
  const size_t o_consts = (o_structs + (size_t)n_trees * sizeof(IlqrTreeDev) + 15) & ~(size_t)15;
  const size_t o_work = (o_consts + 2 * sizeof(IlqrConst) + 15) & ~(size_t)15;
  size_t total = (o_work + (nd - nd_in) * sizeof(double) + 15) & ~(size_t)15;
  for (int t = 0; t < n_trees && gen; ++t) { tl[t].field = total; total += (size_t)tl[t].M * W * H * sizeof(double); }
  int rc;
  if ((rc = ensure(c, c->ilqr_dev, total))) return rc;
  char *base = (char *)c->ilqr_dev.p;
  double *dD = (double *)base;                 // uploaded doubles: offsets < nd_in
  double *dW = (double *)(base + o_work);      // produced doubles: offsets >= nd_in
  auto Dp = [=](size_t o) -> double * { return o < nd_in ? dD + o : dW + (o - nd_in); };
  float *dF = (float *)(base + bytesIn);
  int *dI = (int *)(base + bytesIn + bytesF);
  // host staging of the read-only part
  auto &hD = c->il_scr.hD; auto &hF = c->il_scr.hF; auto &hI = c->il_scr.hI;
  hD.assign(nd_in, 0.0); hF.assign(nf, 0.f); hI.assign(ni, 0);
  memcpy(hD.data() + o_gx, gx.data(), W * sizeof(double));
  memcpy(hD.data() + o_gy, gy.data(), H * sizeof(double));
  if (n_lane_pts) memcpy(hD.data() + o_lane, target_lane, (size_t)n_lane_pts * 2 * sizeof(double));
  if (ev) {
    memcpy(hD.data() + o_evx, ev->x, (size_t)ev->nq * 6 * sizeof(double));
    memcpy(hD.data() + o_evu, ev->u, (size_t)ev->nq * 2 * sizeof(double));
    memcpy(hI.data() + o_evn, ev->node, (size_t)ev->nq * sizeof(int));
    for (int q = 0; q < ev->nq; ++q)
      if (ev->node[q] < 0 || ev->node[q] >= trees[0].n_nodes) return fail(c, MIND_EINVAL, "mind_cost_eval: node %d out of range", ev->node[q]);
  }
  // where the results land on the host: xs of all trees and, right behind them, the stats of all trees; us (it lives in the uploaded region).
  // A launch of small trees writes them there ITSELF at its end (k_ilqr: the staging is page-locked and mapped), instead of two copies behind it
  const size_t n_hs = (size_t)2 * IL_NSTAT * n_trees;
  const size_t n_xs = (size_t)(tl[0].stats - tl[0].xs);
  const size_t n_hx = n_xs + n_hs, n_us = (size_t)Mtot * 2;
  if ((rc = pl_pin(c, 5, (n_hx + n_us + 2) * sizeof(double) + (size_t)n_trees * sizeof(unsigned)))) return rc;
  double *hx = (double *)c->pl_pin[5], *hus = hx + n_hx;
  unsigned *h_abort = (unsigned *)(hus + n_us);
  unsigned *h_done = h_abort + 4;               // (behind the two doubles kept for the abort word)
  const bool host_out = !ev && c->ilqr_host_out_max > 0 && Mtot <= c->ilqr_host_out_max;
  // per-tree completion words (a launch that cannot abort): the caller may look at a tree's results before the launch has ended (mind_loop)
  const bool early = host_out && !multi && !gen;
  c->il_early = mind_ctx::IlEarly();
  if (early) {
    c->il_gen += 1u;
    if (c->il_gen == 0u) c->il_gen = 1u;
    for (int t = 0; t < n_trees; ++t) h_done[t] = 0u;
    c->il_early.xs = hx; c->il_early.us = hus; c->il_early.done = h_done; c->il_early.gen = c->il_gen; c->il_early.n_trees = n_trees; c->il_early.nodes = Mtot;
  }
  std::vector<IlqrTreeDev> hT(n_trees);
  long moff = 0;
  for (int t = 0; t < n_trees; ++t) {
    const mind_cost_tree &tr = trees[t];
    const TL &L = tl[t];
    const size_t M = L.M;
    if (us_init) memcpy(hD.data() + L.us, us_init + moff * 2, 2 * M * sizeof(double));
    if (tr.prob) memcpy(hF.data() + L.prob, tr.prob, M * sizeof(float));
    if (gen) memcpy(hD.data() + L.nodew, tr.node_w, M * IL_NW * sizeof(double));
    if (tr.agent_mean && !gen && !dev_flat) memcpy(hF.data() + L.mean, tr.agent_mean, M * L.a * 2 * sizeof(float));
    if (tr.agent_cov && !gen && !dev_flat) memcpy(hF.data() + L.cov, tr.agent_cov, M * L.a * sizeof(float));
    memcpy(hI.data() + L.parent, tr.parent, M * sizeof(int));
    memcpy(hI.data() + L.lstart, lvl_start[t].data(), lvl_start[t].size() * sizeof(int));
    memcpy(hI.data() + L.lnodes, lvl_nodes[t].data(), M * sizeof(int));
    memcpy(hI.data() + L.cstart, cst[t].data(), (M + 1) * sizeof(int));
    memcpy(hI.data() + L.clist, cls[t].data(), cls[t].size() * sizeof(int));
    memcpy(hI.data() + L.sstart, sg_start[t].data(), sg_start[t].size() * sizeof(int));
    memcpy(hI.data() + L.snodes, sg_nodes[t].data(), sg_nodes[t].size() * sizeof(int));
    memcpy(hI.data() + L.slstart, sl_start[t].data(), sl_start[t].size() * sizeof(int));
    memcpy(hI.data() + L.slsegs, sl_segs[t].data(), sl_segs[t].size() * sizeof(int));
    memcpy(hI.data() + L.segrec, seg_rec[t].data(), seg_rec[t].size() * sizeof(int));
    memcpy(hI.data() + L.fsstart, fs_start[t].data(), fs_start[t].size() * sizeof(int));
    memcpy(hI.data() + L.fsq0, fs_q0[t].data(), fs_q0[t].size() * sizeof(int));
    memcpy(hI.data() + L.fsq1, fs_q1[t].data(), fs_q1[t].size() * sizeof(int));
    memcpy(hI.data() + L.fsnstart, fs_nstart[t].data(), fs_nstart[t].size() * sizeof(int));
    memcpy(hI.data() + L.fsnodes, fs_nodes[t].data(), fs_nodes[t].size() * sizeof(int));
    IlqrTreeDev &D = hT[t];
    D.M = L.M; D.n_agents = L.a; D.n_levels = L.nl; D.pad = 0;
    D.parent = dI + L.parent; D.level_start = dI + L.lstart; D.level_nodes = dI + L.lnodes;
    D.child_start = dI + L.cstart; D.child_list = dI + L.clist;
    D.rel = dI + L.rel;
    D.relag = Dp(L.relag);
    D.field = gen ? (const double *)(base + L.field) : nullptr;
    D.node_w = gen ? Dp(L.nodew) : nullptr;
    D.n_segs = L.nseg; D.n_slevels = L.nsl; D.max_level_segs = L.maxls; D.pad2 = 0;
    D.seg_start = dI + L.sstart; D.seg_nodes = dI + L.snodes; D.slevel_start = dI + L.slstart; D.slevel_segs = dI + L.slsegs; D.seg_rec = dI + L.segrec;
    D.n_fsteps = L.nfs; D.padf = 0;
    D.trace = trace_cap > 0 ? Dp(L.trace) : nullptr; D.trace_cap = trace_cap; D.padt = 0;
    D.fstep_start = dI + L.fsstart; D.fstep_q0 = dI + L.fsq0; D.fstep_q1 = dI + L.fsq1; D.fstep_nstart = dI + L.fsnstart; D.fstep_nodes = dI + L.fsnodes;
    D.prob = dF + L.prob; D.mean = dF + L.mean; D.cov = dF + L.cov;
    if (dev_flat) { D.mean = c->pl_dev_fmean + (size_t)moff * L.a * 2; D.cov = c->pl_dev_fcov + (size_t)moff * L.a; }
    D.xs = Dp(L.xs); D.us = Dp(L.us); D.Fx = Dp(L.Fx); D.L = Dp(L.L); D.Lx = Dp(L.Lx); D.Lxx = Dp(L.Lxx);
    D.k = Dp(L.k); D.K = Dp(L.K); D.Vx = Dp(L.Vx); D.Vxx = Dp(L.Vxx);
    D.xs_new = Dp(L.xsn); D.us_new = Dp(L.usn); D.L_new = Dp(L.Ln); D.stats = Dp(L.stats);
    D.ctl = slots ? (IlSlotCtl *)(dI + o_ctl + ctl_ints * (size_t)t) : nullptr;
    D.h_xs = host_out ? hx + (L.xs - tl[0].xs) : nullptr; D.h_us = host_out ? hus + (L.us - tl[0].us) : nullptr;
    D.h_stats = host_out ? hx + (L.stats - tl[0].xs) : nullptr;
    D.h_done = early ? h_done + t : nullptr; D.h_gen = c->il_gen; D.pad_h = 0;
    if (early && ((L.xs - tl[0].xs) != (size_t)moff * 6 || (L.us - tl[0].us) != (size_t)moff * 2)) return fail(c, MIND_EINVAL, "tree-iLQR arena: results are not contiguous");
    D.dset = spec ? (long long)L.Fx2 - (long long)L.Fx : 0; D.drel = spec ? (long long)L.rel2 - (long long)L.rel : 0;
    if (spec && (L.L2 - L.L != L.Fx2 - L.Fx || L.Lx2 - L.Lx != L.Fx2 - L.Fx || L.Lxx2 - L.Lxx != L.Fx2 - L.Fx || (use_exo && L.relag2 - L.relag != L.Fx2 - L.Fx)))
      return fail(c, MIND_EINVAL, "tree-iLQR arena: the two derivative sets are laid out differently");
    moff += (long)M;
  }
  for (int t = 0; t < n_trees && gen; ++t)
    HIPCHK(c, hipMemcpyAsync(base + tl[t].field, trees[t].field, (size_t)tl[t].M * W * H * sizeof(double), hipMemcpyHostToDevice, st));
  IlqrConst K;
  memset(&K, 0, sizeof(K));
  K.dt = cfg->dt; K.wb = cfg->wheelbase;
  for (int i = 0; i < 6; ++i) { K.w_des[i] = cfg->w_des_state[i]; K.w_con[i] = cfg->w_state_con[i]; K.lb[i] = cfg->state_lower[i]; K.ub[i] = cfg->state_upper[i]; K.x0[i] = x0[i]; }
  K.w_ctrl[0] = cfg->w_ctrl[0]; K.w_ctrl[1] = cfg->w_ctrl[1];
  K.w_tgt = cfg->w_tgt; K.w_ego = cfg->w_ego; K.w_ego_off = cfg->w_ego_cov_offset; K.w_exo = cfg->w_exo;
  K.w_exo_off = cfg->w_exo_cov_offset; K.w_exo_cost = cfg->w_exo_cost_offset;
  K.res = grid_res; K.off_x = offx; K.off_y = offy; K.target_vel = target_vel;
  K.W = W; K.H = H; K.max_iter = cfg->max_iter; K.use_exo = use_exo_first;
  for (int j = 0; j < IL_NA; ++j) K.alphas[j] = std::pow(1.1, -(double)(j * j));
  K.gx = Dp(o_gx); K.gy = Dp(o_gy); K.quad = Dp(o_quad);
  // a field prepared ahead for exactly this grid and lane (il_field_prepare): the kernels read it where it is
  const bool field_ahead = !gen && !ev && c->il_field_valid && c->il_field_key[0] == x0[0] && c->il_field_key[1] == x0[1] && c->il_field_key[2] == (double)W &&
                           c->il_field_key[3] == (double)H && c->il_field_key[4] == grid_res && c->il_field_lane.size() == 2 * (size_t)n_lane_pts &&
                           memcmp(c->il_field_lane.data(), target_lane, c->il_field_lane.size() * sizeof(double)) == 0;
  c->il_field_valid = false;         // (one call's worth: the next call makes its own or prepares again)
  if (field_ahead) K.quad = (double *)c->il_field.p + (size_t)W + H + 2 * (size_t)n_lane_pts;
  // cell centres are computed in the kernels when the grid is the numpy linspace (always in the planner mode)
  K.stepx = fsx / (double)(W - 1); K.stepy = fsy / (double)(H - 1); K.fsx = fsx; K.fsy = fsy;
  K.lin = 1;
  for (int i = 0; i < W && K.lin; ++i) K.lin = gx[i] == ((i == W - 1 ? K.fsx : (double)i * K.stepx) + offx);
  for (int i = 0; i < H && K.lin; ++i) K.lin = gy[i] == ((i == H - 1 ? K.fsy : (double)i * K.stepy) + offy);
  K.in_x0 = offx + 1.5 * grid_res; K.in_x1 = offx + ((double)W - 2.5) * grid_res;
  K.in_y0 = offy + 1.5 * grid_res; K.in_y1 = offy + ((double)H - 2.5) * grid_res;
  IlqrConst K2[2];
  K2[0] = K;
  K2[1] = K;
  if (cfg2) {      // the full-cost fit: same grid / state / lane field, its own weights
    IlqrConst &F = K2[1];
    for (int i = 0; i < 6; ++i) { F.w_des[i] = cfg2->w_des_state[i]; F.w_con[i] = cfg2->w_state_con[i]; F.lb[i] = cfg2->state_lower[i]; F.ub[i] = cfg2->state_upper[i]; }
    F.w_ctrl[0] = cfg2->w_ctrl[0]; F.w_ctrl[1] = cfg2->w_ctrl[1];
    F.w_tgt = cfg2->w_tgt; F.w_ego = cfg2->w_ego; F.w_ego_off = cfg2->w_ego_cov_offset; F.w_exo = cfg2->w_exo;
    F.w_exo_off = cfg2->w_exo_cov_offset; F.w_exo_cost = cfg2->w_exo_cost_offset;
    F.max_iter = cfg2->max_iter; F.use_exo = 1;
    if (cfg2->dt != cfg->dt || cfg2->wheelbase != cfg->wheelbase || cfg2->grid_res != cfg->grid_res || cfg2->grid_w != cfg->grid_w ||
        cfg2->grid_h != cfg->grid_h)
      return fail(c, MIND_EINVAL, "mind_ilqr_contingency: both configurations must share dt / wheelbase / grid");
  }
  // one staged copy of everything the host provides (`up` lives until the stream has been synchronised below)
  // (page-locked staging: a pageable source makes hipMemcpyAsync a blocking staged copy that also stalls the other contexts of the
  // process -- several planner threads on one GPU then run slower together than one alone)
  ITR("staged in vectors");
  if ((rc = pl_pin(c, 4, o_work))) return rc;
  char *up = (char *)c->pl_pin[4];
  {
    memset(up, 0, o_work);
    memcpy(up, hD.data(), bytesIn);
    if (bytesF) memcpy(up + bytesIn, hF.data(), bytesF);
    if (bytesI) memcpy(up + bytesIn + bytesF, hI.data(), bytesI);
    // (tests: "ilqr_test_starve" with slots on follower workgroups = the followers leave at once, as if they were never scheduled)
    if (slots && c->ilqr_test_starve) ((unsigned *)(up + bytesIn + bytesF))[o_bars + 4 * (size_t)n_trees + 1] = 1u;
    memcpy(up + o_structs, hT.data(), (size_t)n_trees * sizeof(IlqrTreeDev));
    memcpy(up + o_consts, K2, 2 * sizeof(IlqrConst));
    if ((rc = pl_upload(c, base, up, o_work, st))) return rc;
  }
  ITR("upload queued");
  const IlqrConst *dK = (const IlqrConst *)(base + o_consts);
  if (field_ahead) HIPCHK(c, hipStreamWaitEvent(st, c->ev_field, 0));
  else if (!gen) hipLaunchKernelGGL(k_lane_field, dim3((W * H + 255) / 256), dim3(256), 0, st, K.gx.p, K.gy.p, W, H, Dp(o_lane), n_lane_pts, Dp(o_quad));
  int amax = 1;
  for (int t = 0; t < n_trees; ++t) amax = tl[t].a > amax ? tl[t].a : amax;
  const IlqrTreeDev *dT = (const IlqrTreeDev *)(base + o_structs);
  {
    auto off = [&](size_t o) { return (size_t)((const char *)Dp(o) - base); };
    c->il_dbg[0] = off(tl[0].L); c->il_dbg[1] = off(tl[0].Lx); c->il_dbg[2] = off(tl[0].Lxx); c->il_dbg[3] = off(tl[0].Fx); c->il_dbg[4] = off(tl[0].xs);
  }
  c->il_dbg[5] = (size_t)tl[0].M;
  if (ev) {
    const size_t lds = (IL_SCR + (size_t)4 * amax) * sizeof(double);
    if (gen) hipLaunchKernelGGL(k_cost_eval<true>, dim3(ev->nq), dim3(64), lds, st, dT, K, ev->nq, dI + o_evn, Dp(o_evx), Dp(o_evu), Dp(o_evo));
    else hipLaunchKernelGGL(k_cost_eval<false>, dim3(ev->nq), dim3(64), lds, st, dT, K, ev->nq, dI + o_evn, Dp(o_evx), Dp(o_evu), Dp(o_evo));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(ev->out, Dp(o_evo), (size_t)ev->nq * IL_EVAL_OUT * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    return MIND_OK;
  }
  const size_t il_lds = il_lds_bytes(amax);
  // wide trees (hundreds of nodes, dozens of chain segments per level): several workgroups per tree (ilqr_kernels.hip il_fit<.., true>)
  unsigned *dBars = (unsigned *)(dI + o_bars);
  if (c->profiling) {
    if (!c->ev_il0) { HIPCHK(c, hipEventCreate(&c->ev_il0)); HIPCHK(c, hipEventCreate(&c->ev_il1)); }
    HIPCHK(c, hipEventRecord(c->ev_il0, st));
  }
  c->ilqr_trees = n_trees; c->ilqr_multi = multi ? G : (slots ? GS + spec : 1); c->ilqr_ms = 0.f;
  auto launch = [=](bool multi_) {
    if (gen) {
      hipLaunchKernelGGL((k_ilqr<true, 0>), dim3(n_trees), dim3(IL_THREADS), il_lds, st, dT, dK, n_phases, n_trees, 1, dBars, 0);
    } else if (slots) {
      // a master + GS - 1 followers per tree; a follower that is not resident yet is simply not used (IlSlotCtl.alive): no co-residency needed
      // (+ the derivative speculator, the last workgroup of a tree)
      hipLaunchKernelGGL((k_ilqr<false, 2>), dim3(((n_trees + 7) / 8) * 8 * (GS + spec)), dim3(IL_THREADS), il_lds, st, dT, dK, n_phases, n_trees, GS, dBars, spec);
    } else if (multi_) {
      // (ilqr_test_starve: the last eight workgroups are withheld, as if the device could not hold the whole launch: their peers wait
      // at the first barrier, raise the abort word and the call falls back to the one-workgroup kernel below)
      hipLaunchKernelGGL((k_ilqr<false, 1>), dim3(((n_trees + 7) / 8) * 8 * G - (c->ilqr_test_starve ? 8 : 0)), dim3(IL_THREADS), il_lds, st, dT, dK,
                         n_phases, n_trees, G, dBars, 0);
    } else {
      hipLaunchKernelGGL((k_ilqr<false, 0>), dim3(n_trees), dim3(IL_THREADS), il_lds, st, dT, dK, n_phases, n_trees, 1, dBars, 0);
    }
  };
  launch(multi);
  ITR("kernel launched");
  HIPCHK(c, hipGetLastError());
  if (c->profiling) HIPCHK(c, hipEventRecord(c->ev_il1, st));
  if (!host_out) {
    HIPCHK(c, hipMemcpyAsync(hx, Dp(tl[0].xs), n_hx * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(hus, Dp(tl[0].us), n_us * sizeof(double), hipMemcpyDeviceToHost, st));
  }
  if (multi) HIPCHK(c, hipMemcpyAsync(h_abort, dBars + 4 * (size_t)n_trees, sizeof(unsigned), hipMemcpyDeviceToHost, st));
  // everything behind the launch -- the wait, the fallback of a launch that was not resident, the outputs -- as one closure over values:
  // run at once, or kept in the context by mind_ilqr_contingency_begin and run by mind_ilqr_finish (the caller's thread is free meanwhile)
  const size_t hs_size = n_hs;
  std::function<int()> fin = [=]() -> int {
  std::vector<double> hs(hs_size);
  HIPCHK(c, hipStreamSynchronize(st));
  if (c->profiling) HIPCHK(c, hipEventElapsedTime(&c->ilqr_ms, c->ev_il0, c->ev_il1));
  if (multi) {
    const unsigned aborted = *h_abort;
    if (aborted) {
      // the workgroups of a wide tree did not meet at a barrier within ~2 s: the launch was not fully resident (another context or
      // stream held CUs -- several planners on one GPU).  The one-workgroup-per-tree kernel needs no co-residency: the upload (initial
      // controls, zeroed barrier words) is repeated and the call solved with it -- same arithmetic, same results, just slower.
      c->n_ilqr_fallbacks++;
      HIPCHK(c, hipMemcpyAsync(base, up, o_work, hipMemcpyHostToDevice, st));
      launch(false);
      HIPCHK(c, hipGetLastError());
      c->ilqr_multi = 1;
      if (!host_out) {
        HIPCHK(c, hipMemcpyAsync(hx, Dp(tl[0].xs), n_hx * sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipMemcpyAsync(hus, Dp(tl[0].us), n_us * sizeof(double), hipMemcpyDeviceToHost, st));
      }
      HIPCHK(c, hipStreamSynchronize(st));
    }
  }
  memcpy(xs, hx, (size_t)Mtot * 6 * sizeof(double));
  memcpy(us, hus, n_us * sizeof(double));
  memcpy(hs.data(), hx + n_xs, hs.size() * sizeof(double));
  c->il_trace_dev.assign(n_trees, nullptr);
  c->il_trace_its.assign((size_t)2 * n_trees, 0);
  c->il_trace_cap = trace_cap; c->il_trace_phases = n_phases;
  for (int t = 0; t < n_trees; ++t) {
    c->il_trace_dev[t] = trace_cap > 0 ? Dp(tl[t].trace) : nullptr;
    for (int ph = 0; ph < n_phases; ++ph) c->il_trace_its[2 * t + ph] = (int)hs[(size_t)2 * IL_NSTAT * t + (size_t)ph * IL_NSTAT];
  }
  c->il_spec_req = 0; c->il_spec_hit = 0;
#ifndef IL_PROFILE
  for (int t = 0; t < n_trees; ++t)
    for (int ph = 0; ph < n_phases; ++ph) {
      const double *h = hs.data() + (size_t)2 * IL_NSTAT * t + (size_t)ph * IL_NSTAT;
      c->il_spec_req += (long long)h[9]; c->il_spec_hit += (long long)h[10];
    }
#endif
  {
    // phase cycles of the launch's critical tree (the one with the most cycles over all its fits): what bounds the launch
    double best = -1.0;
    for (int t = 0; t < n_trees; ++t) {
      double tot = 0.0, ph_c[5] = {0, 0, 0, 0, 0}, passes = 0.0;
      for (int ph = 0; ph < n_phases; ++ph) {
        const double *h = hs.data() + (size_t)2 * IL_NSTAT * t + (size_t)ph * IL_NSTAT;
        ph_c[0] += h[4]; ph_c[1] += h[5]; ph_c[2] += h[8]; ph_c[3] += h[6]; ph_c[4] += h[7];
        passes += h[IL_NSTAT - 1];
      }
      for (double v : ph_c) tot += v;
      if (tot > best) {
        best = tot;
        double *o = c->il_prof;
        o[0] = tl[t].M; o[1] = tl[t].nl; o[2] = passes;
        for (int q = 0; q < 5; ++q) o[3 + q] = ph_c[q];       // derivatives, backward, state chain, cost pass, selection
        o[8] = (double)n_trees;
      }
    }
  }
  for (int ph = 0; ph < n_phases; ++ph) {
    mind_ilqr_stats *so = ph == 0 ? stats : stats2;
    if (!so) continue;
    for (int t = 0; t < n_trees; ++t) {
      const double *h = hs.data() + (size_t)2 * IL_NSTAT * t + (size_t)ph * IL_NSTAT;
      mind_ilqr_stats *stats = so;      // (shadows the parameter inside this loop body)
      stats[t].iterations = (int)h[0]; stats[t].converged = (int)h[1];
      stats[t].J = h[2]; stats[t].mu = h[3];
      if (getenv("MIND_ILQR_TRACE"))
        fprintf(stderr, "[k_ilqr] tree %d exo %d M %d segs %d seg-levels %d widest %d agents %d it %d passes %.0f: cycles derivatives %.0f backward %.0f state chain %.0f cost pass %.0f select %.0f\n", t,
                n_phases == 2 ? ph : use_exo, tl[t].M, tl[t].nseg, tl[t].nsl, tl[t].maxls, tl[t].a, stats[t].iterations, h[IL_NSTAT - 1], h[4], h[5], h[8], h[6], h[7]);
#ifdef IL_PROFILE
      if (getenv("MIND_ILQR_TRACE")) {
        fprintf(stderr, "[k_ilqr prof] wave0: chain node (n=%.0f): stage %.0f u+dyn+store %.0f | cost chunk (n=%.0f): stage+loads %.0f field %.0f cost+store %.0f | riccati node (n=%.0f): products %.0f Qxx %.0f solve %.0f update %.0f | deriv block (n=%.0f): setup %.0f tasks %.0f assemble %.0f\n",
                h[13], h[8] / fmax(h[13], 1), h[9] / fmax(h[13], 1),
                h[23], h[10] / fmax(h[23], 1), h[11] / fmax(h[23], 1), h[12] / fmax(h[23], 1),
                h[18], h[14] / fmax(h[18], 1), h[15] / fmax(h[18], 1), h[16] / fmax(h[18], 1), h[17] / fmax(h[18], 1),
                h[22], h[19] / fmax(h[22], 1), h[20] / fmax(h[22], 1), h[21] / fmax(h[22], 1));
      }
#endif
    }
  }
  return MIND_OK;
  };
  if (c->il_begin_only) {
    c->il_begin_only = false;
    c->il_finish = std::move(fin);
    return MIND_OK;
  }
  return fin();
}

extern "C" int mind_ilqr_contingency_begin(mind_ctx *c, const mind_ilqr_cfg *cfg_warm, const mind_ilqr_cfg *cfg_full,
                                           const mind_cost_tree *trees, int n_trees, const double *x0, const double *target_lane,
                                           int n_lane_pts, double target_vel, double *xs, double *us,
                                           mind_ilqr_stats *stats_warm, mind_ilqr_stats *stats_full) {
  if (!c || !cfg_full) return fail(c, MIND_EINVAL, "mind_ilqr_contingency_begin: null configuration");
  c->il_begin_only = true;
  const int rc = ilqr_impl(c, cfg_warm, nullptr, trees, n_trees, x0, target_lane, n_lane_pts, target_vel, 0, nullptr, xs, us, stats_warm, nullptr,
                           cfg_full, stats_full);
  c->il_begin_only = false;
  return rc;
}

// mind_ilqr_contingency_begin on the cost trees the last mind_aime_plan of this context flattened (its library-owned tables: no tree
// arrays cross the boundary again)
extern "C" int mind_ilqr_contingency_begin_plan(mind_ctx *c, const mind_ilqr_cfg *cfg_warm, const mind_ilqr_cfg *cfg_full, const double *x0,
                                                const double *target_lane, int n_lane_pts, double target_vel, double *xs, double *us,
                                                mind_ilqr_stats *stats_warm, mind_ilqr_stats *stats_full) {
  if (!c || !cfg_full) return fail(c, MIND_EINVAL, "mind_ilqr_contingency_begin_plan: null configuration");
  const int nt = (int)c->pl_tree_top.size();
  if (nt <= 0 || c->pl_plan_agents <= 0) return fail(c, MIND_ESTATE, "mind_ilqr_contingency_begin_plan: the context holds no planned cost trees");
  const int a = c->pl_plan_agents;
  std::vector<mind_cost_tree> trees(nt);
  for (int t = 0; t < nt; ++t) {
    const size_t lo = (size_t)c->pl_tree_off[t];
    mind_cost_tree &T = trees[t];
    memset(&T, 0, sizeof(T));
    T.n_nodes = c->pl_tree_off[t + 1] - c->pl_tree_off[t];
    T.parent = c->pl_flat_parent.data() + lo; T.prob = c->pl_flat_prob.data() + lo;
    T.n_agents = a;
    // (host copies when the plan has read them back already; the call itself reads the device buffers k_aime_flat filled)
    T.agent_mean = c->pl_fmean_p ? c->pl_fmean_p + lo * a * 2 : nullptr; T.agent_cov = c->pl_fcov_p ? c->pl_fcov_p + lo * a : nullptr;
  }
  c->il_begin_only = true;
  c->il_use_dev_flat = c->pl_dev_fmean != nullptr && c->pl_dev_fcov != nullptr;
  const int rc = ilqr_impl(c, cfg_warm, nullptr, trees.data(), nt, x0, target_lane, n_lane_pts, target_vel, 0, nullptr, xs, us, stats_warm, nullptr,
                           cfg_full, stats_full);
  c->il_begin_only = false;
  c->il_use_dev_flat = false;
  return rc;
}

extern "C" int mind_ilqr_finish(mind_ctx *c) {
  if (!c) return MIND_EINVAL;
  if (!c->il_finish) return fail(c, MIND_ESTATE, "mind_ilqr_finish: no tree-iLQR call was begun on this context");
  std::function<int()> fin = std::move(c->il_finish);
  c->il_finish = nullptr;
  c->il_finish_owned = false;
  c->il_early = mind_ctx::IlEarly();
  return fin();
}

extern "C" int mind_ilqr_contingency(mind_ctx *c, const mind_ilqr_cfg *cfg_warm, const mind_ilqr_cfg *cfg_full,
                                     const mind_cost_tree *trees, int n_trees, const double *x0, const double *target_lane,
                                     int n_lane_pts, double target_vel, double *xs, double *us,
                                     mind_ilqr_stats *stats_warm, mind_ilqr_stats *stats_full) {
  if (!cfg_full) return fail(c, MIND_EINVAL, "mind_ilqr_contingency: null configuration");
  return ilqr_impl(c, cfg_warm, nullptr, trees, n_trees, x0, target_lane, n_lane_pts, target_vel, 0, nullptr, xs, us, stats_warm, nullptr,
                   cfg_full, stats_full);
}

extern "C" int mind_ilqr_solve_trees(mind_ctx *c, const mind_ilqr_cfg *cfg, const mind_cost_tree *trees, int n_trees,
                                     const double *x0, const double *target_lane, int n_lane_pts, double target_vel,
                                     int use_exo, const double *us_init, double *xs, double *us,
                                     mind_ilqr_stats *stats) {
  return ilqr_impl(c, cfg, nullptr, trees, n_trees, x0, target_lane, n_lane_pts, target_vel, use_exo, us_init, xs, us, stats, nullptr);
}

extern "C" int mind_ilqr_solve_fields(mind_ctx *c, const mind_ilqr_cfg *cfg, const mind_field_grid *grid,
                                      const mind_cost_tree *trees, int n_trees, const double *x0,
                                      const double *us_init, double *xs, double *us, mind_ilqr_stats *stats) {
  if (!grid) return fail(c, MIND_EINVAL, "mind_ilqr_solve_fields: null grid");
  return ilqr_impl(c, cfg, grid, trees, n_trees, x0, nullptr, 0, 0.0, 0, us_init, xs, us, stats, nullptr);
}

extern "C" int mind_cost_eval(mind_ctx *c, const mind_ilqr_cfg *cfg, const mind_field_grid *grid, const mind_cost_tree *tree,
                              const double *x0, const double *target_lane, int n_lane_pts, double target_vel, int use_exo,
                              int n_query, const int32_t *node, const double *x, const double *u, double *out) {
  if (n_query <= 0 || !node || !x || !u || !out) return fail(c, MIND_EINVAL, "mind_cost_eval: bad argument");
  IlqrEvalReq ev{n_query, node, x, u, out};
  return ilqr_impl(c, cfg, grid, tree, 1, x0, target_lane, n_lane_pts, target_vel, use_exo, nullptr, nullptr, nullptr, nullptr, &ev);
}

// -------------------------------------------------------------------------------------------------
// AIME glue (k7): world-frame modes + topology signatures of a round's scenes
// -------------------------------------------------------------------------------------------------
extern "C" int mind_aime_world(mind_ctx *c, const mind_world_in *in, const mind_world_out *out) {
  if (!c || !in || !out || in->n_scenes <= 0 || !in->actor_off || !in->reg || !in->vel || !in->actor_ctrs || !in->actor_vecs ||
      !in->rot || !in->orig || !in->cov_last || !in->last || !out->world || !out->topo || !out->ego_end)
    return fail(c, MIND_EINVAL, "mind_aime_world: bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t st = c->stream;
  const int B = in->n_scenes, A = in->actor_off[B];
  if (A <= 0) return fail(c, MIND_EINVAL, "mind_aime_world: no agents");
  std::vector<AimeScene> hs(B);
  std::vector<int> ascene(A);
  for (int b = 0; b < B; ++b) {
    AimeScene &S = hs[b];
    S.a0 = in->actor_off[b]; S.a1 = in->actor_off[b + 1]; S.last = in->last[b]; S.cmp = 0; S.pad2 = 0.f;
    if (S.a1 <= S.a0) return fail(c, MIND_EINVAL, "mind_aime_world: scene %d has no agents", b);
    S.r00 = in->rot[4 * b]; S.r01 = in->rot[4 * b + 1]; S.r10 = in->rot[4 * b + 2]; S.r11 = in->rot[4 * b + 3];
    S.ox = in->orig[2 * b]; S.oy = in->orig[2 * b + 1];
    S.theta_g = atan2f(S.r10, S.r00);
    for (int i = S.a0; i < S.a1; ++i) ascene[i] = b;
  }
  const size_t bS = ((size_t)B * sizeof(AimeScene) + 15) & ~(size_t)15, bI = ((size_t)A * sizeof(int) + 15) & ~(size_t)15;
  const size_t bC = ((size_t)A * sizeof(float) + 15) & ~(size_t)15;
  const bool select = in->cls && in->scen_prob && out->sel && out->sel_prob;
  const size_t bP = select ? (((size_t)B * sizeof(float) + 15) & ~(size_t)15) : 0;
  const int n_lane = in->target_lane ? in->n_lane_pts : 0;
  if (in->target_lane && n_lane < 2) return fail(c, MIND_EINVAL, "mind_aime_world: target lane needs >= 2 points");
  int rc;
  if (select && in->lane_check) {
    if (!n_lane) return fail(c, MIND_EINVAL, "mind_aime_world: lane_check needs the target lane");
    for (int b = 0; b < B; ++b)
      if (in->last[b] < 0) return fail(c, MIND_EINVAL, "mind_aime_world: lane_check needs last >= 0 (scene %d)", b);
  }
  if ((rc = ensure(c, c->aime_dev, bS + bI + bC + bP + (size_t)(n_lane > 0 ? n_lane : 1) * 2 * sizeof(float)))) return rc;
  char *base = (char *)c->aime_dev.p;
  {
    // one staged host->device copy for the four small tables
    // (the staging buffer lives in the context and is guarded by an event: the copy is queued behind the predictor's kernels and
    // the host goes on to queue k_aime_world / k_aime_select without waiting for them)
    const size_t tot = bS + bI + bC + bP + (size_t)n_lane * 2 * sizeof(float);
    if (c->aime_stage_busy) { HIPCHK(c, hipEventSynchronize(c->ev_stage)); c->aime_stage_busy = false; }
    std::vector<char> &stage = c->aime_stage;
    stage.assign(tot, 0);
    if (select) memcpy(stage.data() + bS + bI + bC, in->scen_prob, (size_t)B * sizeof(float));
    memcpy(stage.data(), hs.data(), (size_t)B * sizeof(AimeScene));
    memcpy(stage.data() + bS, ascene.data(), (size_t)A * sizeof(int));
    memcpy(stage.data() + bS + bI, in->cov_last, (size_t)A * sizeof(float));
    if (n_lane) memcpy(stage.data() + bS + bI + bC + bP, in->target_lane, (size_t)n_lane * 2 * sizeof(float));
    HIPCHK(c, hipMemcpyAsync(base, stage.data(), tot, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipEventRecord(c->ev_stage, st));
    c->aime_stage_busy = true;
  }
  AimeSmall sm0;                  // (tables in memory)
  memset(&sm0, 0, sizeof(sm0));
  hipLaunchKernelGGL(k_aime_world, dim3(A * AIME_K), dim3(64), 0, st, (const AimeScene *)base, (const int *)(base + bS), in->reg, in->vel,
                     in->actor_ctrs, in->actor_vecs, (const float *)(base + bS + bI), out->world, out->topo, out->ego_end,
                     (const float *)(base + bS + bI + bC + bP), n_lane, sm0);
  if (select)
    hipLaunchKernelGGL(k_aime_select, dim3(B), dim3(64), 0, st, (const AimeScene *)base, in->cls, (const float *)(base + bS + bI + bC),
                       out->topo, out->ego_end, in->lane_check ? 1 : 0, in->dist_thres, out->sel, out->sel_prob, 0.001f, sm0);
  HIPCHK(c, hipGetLastError());
  return MIND_OK;
}

extern "C" int mind_aime_rebase(mind_ctx *c, const mind_rebase_in *in, const mind_rebase_out *out) {
  const bool dev_src = in && in->rows_dev;
  if (!c || !in || !out || in->n_scenes <= 0 || in->n_agents <= 0 || in->n_lanes < 0 ||
      (!dev_src && (!in->pos || !in->ang || !in->vel)) || (dev_src && (!in->parent_slot || !in->row0 || !in->dur)) || !in->types ||
      (in->n_lanes > 0 && (!in->lane_ctrs || !in->lane_vecs)) || !in->target_lane || !in->target_lane_info || in->n_lane_pts < 12 ||
      !out->actors || !out->actor_ctrs || !out->actor_vecs || (in->n_lanes > 0 && (!out->lane_ctrs || !out->lane_vecs)) ||
      !out->tgt_nodes || !out->tgt_rpe || !out->frames)
    return fail(c, MIND_EINVAL, "mind_aime_rebase: bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t st = c->stream;
  const size_t S = in->n_scenes, a = in->n_agents, l = in->n_lanes, P = in->n_lane_pts;
  if (dev_src) {
    if (in->prev_gen != c->rb_gen || c->rb_gen == 0 || c->rb_a != a)
      return fail(c, MIND_ESTATE, "mind_aime_rebase: the parents' windows (generation %d, %zu agents) are not the previous call's (%d, %zu)",
                  in->prev_gen, a, c->rb_gen, c->rb_a);
    for (size_t s_ = 0; s_ < S; ++s_)
      if (in->parent_slot[s_] < 0 || (size_t)in->parent_slot[s_] >= c->rb_S || in->row0[s_] < 0 || in->dur[s_] < 0 || in->dur[s_] > AIME_T)
        return fail(c, MIND_EINVAL, "mind_aime_rebase: scene %zu: parent slot %d / row %d / dur %d out of range", s_, in->parent_slot[s_],
                    in->row0[s_], in->dur[s_]);
  }
  // staging layout (floats): pos | ang | vel | types | pad | lane_ctrs | lane_vecs | tlane | tinfo | (device source: parent_slot | row0 | dur)
  const size_t n_pos = S * a * 50 * 2, n_ang = S * a * 50, n_types = a * 50 * 7, n_pad = in->pad ? S * a * 50 : 0;
  const size_t o_pos = 0, o_ang = o_pos + n_pos, o_vel = o_ang + n_ang, o_types = o_vel + n_pos, o_pad = o_types + n_types;
  const size_t o_lc = o_pad + n_pad, o_lv = o_lc + 2 * l, o_tl = o_lv + 2 * l, o_ti = o_tl + 2 * P, o_idx = o_ti + 12 * P;
  const size_t total = o_idx + (dev_src ? 3 * S : 0);
  // two arenas, used alternately: the windows of THIS call are the parents' windows of the next one
  DevBuf &cur = c->rebase_dev2[c->rb_cur ^ 1];
  const float *prev = (const float *)c->rebase_dev2[c->rb_cur].p;
  const size_t prev_o_ang = c->rb_S * c->rb_a * 50 * 2, prev_o_vel = prev_o_ang + c->rb_S * c->rb_a * 50;
  int rc;
  if ((rc = ensure(c, cur, total * sizeof(float)))) return rc;
  float *d = (float *)cur.p;
  // the arena is contiguous: small rounds (a few child scenes) go in ONE staged copy; big ones (cfg4: 14 MB of windows) copy the
  // three window arrays straight from the caller's memory and stage only the small tables (types | pad | lane anchors | target
  // lane | its info | indices); with a device source the windows are not uploaded at all
  const bool one_copy = !dev_src && total * sizeof(float) <= ((size_t)1 << 20);
  const size_t s0 = one_copy ? 0 : o_types;
  std::vector<float> small(total - s0);
  if (one_copy) {
    memcpy(small.data() + o_pos, in->pos, n_pos * sizeof(float));
    memcpy(small.data() + o_ang, in->ang, n_ang * sizeof(float));
    memcpy(small.data() + o_vel, in->vel, n_pos * sizeof(float));
  } else if (!dev_src) {
    HIPCHK(c, hipMemcpyAsync(d + o_pos, in->pos, n_pos * sizeof(float), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(d + o_ang, in->ang, n_ang * sizeof(float), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(d + o_vel, in->vel, n_pos * sizeof(float), hipMemcpyHostToDevice, st));
  }
  memcpy(small.data() + (o_types - s0), in->types, n_types * sizeof(float));
  if (in->pad) memcpy(small.data() + (o_pad - s0), in->pad, n_pad * sizeof(float));
  if (l) {
    memcpy(small.data() + (o_lc - s0), in->lane_ctrs, 2 * l * sizeof(float));
    memcpy(small.data() + (o_lv - s0), in->lane_vecs, 2 * l * sizeof(float));
  }
  memcpy(small.data() + (o_tl - s0), in->target_lane, 2 * P * sizeof(float));
  memcpy(small.data() + (o_ti - s0), in->target_lane_info, 12 * P * sizeof(float));
  if (dev_src) {
    memcpy(small.data() + (o_idx - s0), in->parent_slot, S * sizeof(int));
    memcpy(small.data() + (o_idx - s0) + S, in->row0, S * sizeof(int));
    memcpy(small.data() + (o_idx - s0) + 2 * S, in->dur, S * sizeof(int));
  }
  HIPCHK(c, hipMemcpyAsync(d + s0, small.data(), small.size() * sizeof(float), hipMemcpyHostToDevice, st));
  if (dev_src) {
    const int *di = (const int *)(d + o_idx);
    hipLaunchKernelGGL(k_aime_windows, dim3((unsigned)(S * a)), dim3(64), 0, st, prev, prev + prev_o_ang, prev + prev_o_vel, in->rows_dev,
                       di, di + S, di + 2 * S, (int)a, d + o_pos, d + o_ang, d + o_vel, 1, (float *)nullptr);
  }
  c->rb_cur ^= 1; c->rb_gen += 1; c->rb_S = S; c->rb_a = a;
  if (out->gen) *out->gen = c->rb_gen;
  RebaseArgs A;
  A.a = (int)a; A.l = (int)l; A.n_lane = (int)P; A.pad_ones = in->pad ? 0 : 1;
  A.pos = d + o_pos; A.ang = d + o_ang; A.vel = d + o_vel; A.types = d + o_types; A.pad = in->pad ? d + o_pad : nullptr;
  A.lane_ctrs0 = d + o_lc; A.lane_vecs0 = d + o_lv; A.tlane = d + o_tl; A.tinfo = d + o_ti;
  A.time_ahead = in->time_ahead; A.min_vel = in->min_vel; A.travel0 = -1.f;
  A.actors = out->actors; A.actor_ctrs = out->actor_ctrs; A.actor_vecs = out->actor_vecs; A.lane_ctrs = out->lane_ctrs;
  A.lane_vecs = out->lane_vecs; A.tgt_nodes = out->tgt_nodes; A.tgt_rpe = out->tgt_rpe; A.frames = out->frames;
  hipLaunchKernelGGL(k_aime_rebase, dim3((unsigned)S, 1 + RB_FEAT_BLOCKS(a)), dim3(RB_THREADS), 5 * a * sizeof(float), st, A);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(st));     // the caller's host arrays may be reused after return
  return MIND_OK;
}

#include "aime_plan.hip"

// ---- MINDPlanner.evaluate_traj_tree (planner.py:180-198) for all candidate trajectory trees of a plan: mean over a tree's nodes of
//      .1 jerk^2 + 5 steer_rate^2 + .01 (v_tgt - v)^2 + .01 dist(target lane).  Host only (a few 10^4 point-segment pairs); float64 with
//      numpy's operation order (per-node terms left to right, the per-tree sum as np.add.reduceat forms it: first element + pairwise
//      sum of the rest, blocks of 8 / 128), so that the strict `<` scan over the candidates sees the costs numpy computes.
namespace {
double np_pairwise(const double *a, long n) {
  if (n < 8) {
    double r = 0.0;
    for (long i = 0; i < n; ++i) r += a[i];
    return r;
  }
  if (n <= 128) {
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
  return np_pairwise(a, n2) + np_pairwise(a + n2, n - n2);
}
template <class T>
void eval_nodes(const double *st, const double *ct, long N, const T *lane, int P, double tv, double *per_node) {
  const int S = P - 1;
  std::vector<T> dx(S), dy(S), l2(S);
  for (int q = 0; q < S; ++q) {
    dx[q] = lane[2 * q + 2] - lane[2 * q];
    dy[q] = lane[2 * q + 3] - lane[2 * q + 1];
    l2[q] = dx[q] * dx[q] + dy[q] * dy[q];                      // in the lane's own dtype, as numpy computes it
  }
  // Only the MINIMUM over the segments is used, so a segment that provably cannot hold it is not priced: segments are grouped in blocks of
  // eight with a bounding circle (centre, radius inflated by 1e-9 relative + 1e-9 m: far above any rounding of the distances involved), and a
  // block -- then a segment, by its own circle -- is skipped when |p - centre| > sqrt(current minimum) + radius.  Every segment that could
  // equal or undercut the running minimum is evaluated with numpy's expression, so the result is the same bits; a trajectory's
  // nodes follow the lane, so the scan starts at the block that held the previous node's minimum.
  constexpr int BS = 8;
  const int NB = (S + BS - 1) / BS;
  std::vector<double> sx_(S), sy_(S), sr_(S), bx(NB), by(NB), br(NB);
  for (int q = 0; q < S; ++q) {
    const double ax = (double)lane[2 * q], ay = (double)lane[2 * q + 1], ex = (double)lane[2 * q + 2], ey = (double)lane[2 * q + 3];
    sx_[q] = 0.5 * (ax + ex); sy_[q] = 0.5 * (ay + ey);
    sr_[q] = 0.5 * sqrt((ex - ax) * (ex - ax) + (ey - ay) * (ey - ay)) * (1.0 + 1e-9) + 1e-9;
  }
  for (int b = 0; b < NB; ++b) {
    const int q0 = b * BS, q1 = std::min(S, q0 + BS);
    double cx = 0.0, cy = 0.0;
    for (int q = q0; q < q1; ++q) { cx += sx_[q]; cy += sy_[q]; }
    cx /= (double)(q1 - q0); cy /= (double)(q1 - q0);
    double r = 0.0;
    for (int q = q0; q < q1; ++q) r = std::max(r, sqrt((sx_[q] - cx) * (sx_[q] - cx) + (sy_[q] - cy) * (sy_[q] - cy)) + sr_[q]);
    bx[b] = cx; by[b] = cy; br[b] = r * (1.0 + 1e-9) + 1e-9;
  }
  int b_prev = 0;
  for (long i = 0; i < N; ++i) {
    const double px = st[6 * i], py = st[6 * i + 1];
    // min over the segments of sqrt(e) = sqrt of the min of e: the correctly rounded square root is monotone, so ONE square root per node
    // gives numpy's bits (np.sqrt per segment, then min) -- the division and the square root share the host's divider unit
    double emin = INFINITY, rmin = INFINITY;
    bool saw_nan = false;
    int b_best = b_prev;
    auto price_block = [&](int b) {
      const int q0 = b * BS, q1 = std::min(S, q0 + BS);
      for (int q = q0; q < q1; ++q) {
        if (emin < INFINITY) {
          const double cxq = px - sx_[q], cyq = py - sy_[q], lim = rmin + sr_[q];
          if (cxq * cxq + cyq * cyq > lim * lim) continue;
        }
        const double sx = (double)lane[2 * q], sy = (double)lane[2 * q + 1], ddx = (double)dx[q], ddy = (double)dy[q];
        double t = ((px - sx) * ddx + (py - sy) * ddy) / (double)l2[q];
        t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
        const double ex = px - (sx + t * ddx), ey = py - (sy + t * ddy);
        const double e = ex * ex + ey * ey;
        if (e < emin) { emin = e; rmin = sqrt(e) * (1.0 + 1e-9) + 1e-9; b_best = b; }
        else if (e != e) saw_nan = true;                            // numpy's min propagates a NaN
      }
    };
    price_block(b_prev);
    for (int b = 0; b < NB; ++b) {
      if (b == b_prev) continue;
      if (emin < INFINITY) {
        const double cxb = px - bx[b], cyb = py - by[b], lim = rmin + br[b];
        if (cxb * cxb + cyb * cyb > lim * lim) continue;
      }
      price_block(b);
    }
    b_prev = b_best;
    const double dmin = saw_nan ? (double)NAN : sqrt(emin);
    const double c0 = ct[2 * i], c1 = ct[2 * i + 1], dv = tv - st[6 * i + 2];
    per_node[i] = ((0.1 * (c0 * c0) + 5.0 * (c1 * c1)) + 0.01 * (dv * dv)) + 0.01 * dmin;
  }
}
}  // namespace

// get_agent_trajectories' array part (planners/mind/utils.py:245-342 of the reference: per track the observed flags, positions and headings
// of unobserved steps taken from the nearest earlier observed one -- the first observed one before it --, velocities zero there, the type
// one-hot on observed steps) for all tracks at once; host arithmetic, plain copies and float64 -> float32 casts.
extern "C" int mind_fill_tracks(const double *raw /*[a,T,6]: observed, x, y, heading, vx, vy*/, int a, int T, const int32_t *slot /*[a]*/,
                                float *pos /*[a,T,2]*/, float *ang /*[a,T]*/, float *vel /*[a,T,2]*/, int16_t *typ /*[a,T,7]*/,
                                int16_t *have /*[a,T]*/) {
  if (!raw || !slot || !pos || !ang || !vel || !typ || !have || a < 0 || T <= 0) return MIND_EINVAL;
  for (int i = 0; i < a; ++i) {
    const double *r = raw + (size_t)i * T * 6;
    int first = 0;
    for (int t = 0; t < T; ++t)
      if (r[t * 6] != 0.0) { first = t; break; }       // np.argmax(have): the first observed step (0 when none is)
    int src = -1;
    for (int t = 0; t < T; ++t) {
      const bool h = r[t * 6] != 0.0;
      if (h) src = t;
      const int f = src < 0 ? first : src;
      const bool hf = r[f * 6] != 0.0;                  // (a track without any observed step fills with zeros, as np.where(have, raw, 0) does)
      const size_t o = (size_t)i * T + t;
      pos[2 * o] = hf ? (float)r[f * 6 + 1] : 0.f;
      pos[2 * o + 1] = hf ? (float)r[f * 6 + 2] : 0.f;
      ang[o] = hf ? (float)r[f * 6 + 3] : 0.f;
      vel[2 * o] = h ? (float)r[t * 6 + 4] : 0.f;
      vel[2 * o + 1] = h ? (float)r[t * 6 + 5] : 0.f;
      have[o] = h ? 1 : 0;
      for (int k = 0; k < 7; ++k) typ[o * 7 + k] = (h && k == slot[i]) ? 1 : 0;
    }
  }
  return MIND_OK;
}

extern "C" int mind_eval_traj_trees(const double *states, const double *ctrls, const int32_t *counts, int n_trees, const void *lane,
                                    int lane_is_f32, int n_lane_pts, double target_vel, double *out) {
  if (!states || !ctrls || !counts || n_trees <= 0 || !lane || n_lane_pts < 2 || !out) return MIND_EINVAL;
  long N = 0;
  for (int t = 0; t < n_trees; ++t) { if (counts[t] <= 0) return MIND_EINVAL; N += counts[t]; }
  std::vector<double> per(N);
  // the nodes are priced independently (a node = one distance-to-polyline scan): big plans -- the deep stress trees hold 177 k trajectory
  // nodes -- are spread over a few host threads; the per-tree sums below keep numpy's order either way
  auto run = [&](long lo, long hi) {
    if (lane_is_f32) eval_nodes<float>(states + 6 * lo, ctrls + 2 * lo, hi - lo, (const float *)lane, n_lane_pts, target_vel, per.data() + lo);
    else eval_nodes<double>(states + 6 * lo, ctrls + 2 * lo, hi - lo, (const double *)lane, n_lane_pts, target_vel, per.data() + lo);
  };
  const long nth = std::min<long>(std::min<long>(8, (long)std::max(1u, std::thread::hardware_concurrency())), N / 8192);
  if (nth <= 1) run(0, N);
  else {
    std::vector<std::thread> th;
    for (long k = 0; k < nth; ++k) th.emplace_back(run, N * k / nth, N * (k + 1) / nth);
    for (std::thread &t : th) t.join();
  }
  long o = 0;
  for (int t = 0; t < n_trees; ++t) {
    const long n = counts[t];
    out[t] = (n > 1 ? per[o] + np_pairwise(per.data() + o + 1, n - 1) : per[o]) / (double)n;
    o += n;
  }
  return MIND_OK;
}

// ---- debug taps (tests only): run only the first n fusion layers; read back internal buffers
// the kernel-side sin / cos / tan on an array (tests: bitwise equality with the oracle's build of the same header)
__global__ void k_debug_trig(const double *__restrict__ x, int n, double *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const double v = i < n ? x[i] : 0.0;          // (the wave-uniform small-argument path needs every lane of the wave in the call)
  double s, c, t, c2;
  mind_sincos(v, &s, &c);
  mind_tan_cos(v, &t, &c2);
  if (i < n) { out[4 * i] = s; out[4 * i + 1] = c; out[4 * i + 2] = t; out[4 * i + 3] = c2; }
}
extern "C" int mind_debug_trig(mind_ctx *c, const double *x, int n, double *out) {
  if (!c || !x || !out || n <= 0) return MIND_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  double *d = nullptr;
  HIPCHK(c, hipMalloc((void **)&d, (size_t)n * 5 * sizeof(double)));
  hipError_t e = hipMemcpy(d, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_debug_trig, dim3((n + 63) / 64), dim3(64), 0, c->stream, d, n, d + n);
    e = hipStreamSynchronize(c->stream);
  }
  if (e == hipSuccess) e = hipMemcpy(out, d + n, (size_t)n * 4 * sizeof(double), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(c, MIND_EHIP, "mind_debug_trig: %s", hipGetErrorString(e));
  return MIND_OK;
}

extern "C" int mind_debug_set_layers(mind_ctx *c, int n) {
  if (!c || n < 0 || n > 6) return MIND_EINVAL;
  c->debug_layers = n;
  return MIND_OK;
}
extern "C" int64_t mind_debug_read(mind_ctx *c, const char *name, float *host, int64_t max_floats) {
  if (!c || !name) return MIND_EINVAL;
  const void *src = nullptr;
  int64_t n = 0;
  std::string k = name;
  if (k == "x") { src = c->x.p; n = (int64_t)c->last_ntok * 128; }
  else if (k == "ST") { src = c->ST.p; n = (int64_t)c->last_ntok * 256; }
  else if (k == "QK") { src = c->QK.p; n = (int64_t)c->last_ntok * 1024; }
  else if (k == "edge") { src = c->edge.p; n = c->last_edge_pairs * 128; }
  else if (k == "part") { src = c->part.p; n = (int64_t)c->last_slots * PART_STRIDE; }
  else if (k == "actor_feat") { src = c->actor_feat.p; n = (int64_t)c->last_A * 128; }
  else if (k == "tokpos") { src = c->tokpos.p; n = (int64_t)c->last_ntok * 4; }
  else if (k == "cmode") { src = c->cmode.p; n = (int64_t)c->last_B * 768; }
  else if (k == "tgt_emb") { src = c->tgt_emb.p; n = (int64_t)c->last_B * 128; }
  else if (k == "tgt_feat") { src = c->tgt_feat.p; n = (int64_t)c->last_B * 128; }
  else if (k == "il_spec") {        // the derivative speculator in the last tree-iLQR launch: {passes it was asked in, results the master took}
    if (!host) return 2;
    if (max_floats < 2) return MIND_EINVAL;
    host[0] = (float)c->il_spec_req; host[1] = (float)c->il_spec_hit;
    return 2;
  }
  else if (k.rfind("il_", 0) == 0 && c->il_dbg[5]) {
    // float64 arrays of tree 0 of the last tree-iLQR call, returned as raw bytes (2 floats per double)
    const size_t M = c->il_dbg[5];
    int w = -1; size_t per = 0;
    if (k == "il_L") { w = 0; per = 1; } else if (k == "il_Lx") { w = 1; per = 6; } else if (k == "il_Lxx") { w = 2; per = 36; }
    else if (k == "il_Fx") { w = 3; per = 36; } else if (k == "il_xs") { w = 4; per = 6; }
    if (w < 0) return MIND_EINVAL;
    src = (const char *)c->ilqr_dev.p + c->il_dbg[w];
    n = (int64_t)(M * per * 2);
  }
  else return MIND_EINVAL;
  if (!host) return n;
  if (n > max_floats) n = max_floats;
  if (hipStreamSynchronize(c->stream) != hipSuccess) return MIND_EHIP;
  if (k == "edge" && c->last_edge_tiled) {
    // k_pair_t keeps the tensor as [scene][column j][tile][chunk b 8][lane 64][4]: hand back the logical [scene][j][i][128]
    long long pt = 0;
    for (int N : c->last_scene_n) pt += (long long)N * (((N + 15) / 16) * 16);
    const bool eb = c->last_edge_bf16;      // plain bf16 arithmetic: [k-group 4][lane 64][4 dwords of two bf16], 64 dwords per pair
    std::vector<float> raw((size_t)pt * (eb ? 64 : 128));
    if (hipMemcpy(raw.data(), src, raw.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return MIND_EHIP;
    int64_t o = 0;
    size_t sb = 0;
    for (int N : c->last_scene_n) {
      const int rows = ((N + 15) / 16) * 16;
      for (int j = 0; j < N; ++j)
        for (int i = 0; i < N; ++i)
          for (int f = 0; f < 128 && o < n; ++f, ++o) {
            const int t = i >> 4, pp = i & 15, b = f >> 4, qq = (f >> 2) & 3, r = f & 3;
            const size_t tile0 = sb + (size_t)j * rows + (size_t)t * 16;
            if (!eb) host[o] = raw[tile0 * 128 + ((b * 64 + qq * 16 + pp) * 4 + r)];
            else {
              uint32_t w;
              memcpy(&w, &raw[tile0 * 64 + (((b >> 1) * 64 + qq * 16 + pp) * 4 + 2 * (b & 1) + (r >> 1))], 4);
              w = (r & 1) ? (w & 0xffff0000u) : (w << 16);
              memcpy(&host[o], &w, 4);
            }
          }
      sb += (size_t)N * rows;
    }
    return n;
  }
  if (hipMemcpy(host, src, (size_t)n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return MIND_EHIP;
  return n;
}

// ---- the closed loop of one scene behind one call per step (host code: the interpreter's share of a planning cycle)
#include "loop.hip"
