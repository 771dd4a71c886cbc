"""ctypes binding of libmind_hip.so (C-ABI in include/mind_hip.h).

The product path has NO CPU fallback: if the HIP library is missing this raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MIND_HIP_LIB", os.path.join(_HERE, "libmind_hip.so"))   # override: diagnostic builds only

MIND_OK = 0
MIND_EINVAL, MIND_ENOMEM, MIND_EHIP, MIND_ESTATE, MIND_ENOTFOUND = -1, -2, -3, -4, -5     # include/mind_hip.h:21-25


class TensorDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.POINTER(C.c_float)), ("numel", C.c_int64)]


class SceneBatch(C.Structure):
    _fields_ = [("n_scenes", C.c_int), ("actor_off", C.POINTER(C.c_int32)), ("lane_off", C.POINTER(C.c_int32)),
                ("actors", C.c_void_p), ("lanes", C.c_void_p), ("lane_feat", C.c_void_p),
                ("actor_ctrs", C.c_void_p), ("actor_vecs", C.c_void_p), ("lane_ctrs", C.c_void_p),
                ("lane_vecs", C.c_void_p), ("rpe", C.POINTER(C.c_void_p)), ("tgt_nodes", C.c_void_p),
                ("tgt_rpe", C.c_void_p)]


class PredOut(C.Structure):
    _fields_ = [("cls", C.c_void_p), ("reg", C.c_void_p), ("vel", C.c_void_p), ("lane_feat", C.c_void_p),
                ("actor_emb", C.c_void_p), ("cls_emb", C.c_void_p)]


class CostTree(C.Structure):
    _fields_ = [("n_nodes", C.c_int), ("parent", C.POINTER(C.c_int32)), ("prob", C.POINTER(C.c_float)),
                ("n_agents", C.c_int), ("agent_mean", C.POINTER(C.c_float)), ("agent_cov", C.POINTER(C.c_float)),
                ("field", C.POINTER(C.c_double)), ("node_w", C.POINTER(C.c_double))]


class FieldGrid(C.Structure):
    _fields_ = [("W", C.c_int), ("H", C.c_int), ("res", C.c_double), ("off_x", C.c_double), ("off_y", C.c_double),
                ("gx", C.POINTER(C.c_double)), ("gy", C.POINTER(C.c_double))]


class WorldIn(C.Structure):
    _fields_ = [("n_scenes", C.c_int), ("actor_off", C.POINTER(C.c_int32)), ("reg", C.c_void_p), ("vel", C.c_void_p),
                ("actor_ctrs", C.c_void_p), ("actor_vecs", C.c_void_p), ("rot", C.POINTER(C.c_float)),
                ("orig", C.POINTER(C.c_float)), ("cov_last", C.POINTER(C.c_float)), ("last", C.POINTER(C.c_int32)),
                ("target_lane", C.POINTER(C.c_float)), ("n_lane_pts", C.c_int), ("cls", C.c_void_p),
                ("scen_prob", C.POINTER(C.c_float)), ("lane_check", C.c_int), ("dist_thres", C.c_float)]


class WorldOut(C.Structure):
    _fields_ = [("world", C.c_void_p), ("topo", C.c_void_p), ("ego_end", C.c_void_p), ("sel", C.c_void_p), ("sel_prob", C.c_void_p)]


class RebaseIn(C.Structure):
    _fields_ = [("n_scenes", C.c_int), ("n_agents", C.c_int), ("n_lanes", C.c_int), ("pos", C.POINTER(C.c_float)),
                ("ang", C.POINTER(C.c_float)), ("vel", C.POINTER(C.c_float)), ("types", C.POINTER(C.c_float)),
                ("pad", C.POINTER(C.c_float)), ("lane_ctrs", C.POINTER(C.c_float)), ("lane_vecs", C.POINTER(C.c_float)),
                ("target_lane", C.POINTER(C.c_float)), ("target_lane_info", C.POINTER(C.c_float)), ("n_lane_pts", C.c_int),
                ("time_ahead", C.c_float), ("min_vel", C.c_float), ("rows_dev", C.c_void_p), ("parent_slot", C.POINTER(C.c_int32)),
                ("row0", C.POINTER(C.c_int32)), ("dur", C.POINTER(C.c_int32)), ("prev_gen", C.c_int)]


class RebaseOut(C.Structure):
    _fields_ = [("actors", C.c_void_p), ("actor_ctrs", C.c_void_p), ("actor_vecs", C.c_void_p), ("lane_ctrs", C.c_void_p),
                ("lane_vecs", C.c_void_p), ("tgt_nodes", C.c_void_p), ("tgt_rpe", C.c_void_p), ("frames", C.c_void_p),
                ("gen", C.POINTER(C.c_int32))]


class AimePlanIn(C.Structure):
    _fields_ = [("n_agents", C.c_int), ("n_lanes", C.c_int), ("n_lane_pts", C.c_int)] + \
               [(k, C.POINTER(C.c_float)) for k in ("actors", "actor_ctrs", "actor_vecs", "lanes", "lane_ctrs", "lane_vecs", "tgt_nodes", "tgt_rpe",
                                                    "rot", "orig", "tgt_pts", "hist", "types", "target_lane", "target_lane_info")] + \
               [("time_ahead", C.c_float), ("min_vel", C.c_float), ("dist_thres", C.c_float), ("max_depth", C.c_int), ("max_rounds", C.c_int), ("pred_len", C.c_int),
                ("raw_pos", C.POINTER(C.c_float)), ("raw_ang", C.POINTER(C.c_float)), ("raw_vel", C.POINTER(C.c_float)), ("raw_pad", C.POINTER(C.c_float)),
                ("lane_pts", C.POINTER(C.c_double)), ("lane_flags", C.POINTER(C.c_int32)), ("travel0", C.c_float),
                ("script_cls", C.c_void_p), ("script_reg", C.c_void_p), ("script_vel", C.c_void_p), ("prob_floor", C.c_float),
                ("solve_cfg_warm", C.c_void_p), ("solve_cfg_full", C.c_void_p), ("solve_x0", C.c_void_p), ("solve_lane", C.c_void_p),
                ("solve_n_lane_pts", C.c_int), ("solve_target_vel", C.c_double)]


class AimeNode(C.Structure):
    _fields_ = [("round", C.c_int), ("scene", C.c_int), ("mode", C.c_int), ("parent", C.c_int), ("prob", C.c_float),
                ("cur_t", C.c_int), ("end_t", C.c_int), ("flags", C.c_int), ("dur", C.c_int), ("row_off", C.c_int64),
                ("tgt_pts", C.c_float * 22)]


class AimePlanOut(C.Structure):
    _fields_ = [("nodes", C.POINTER(AimeNode)), ("n_nodes", C.c_int), ("rows", C.POINTER(C.c_float)), ("n_row_floats", C.c_int64),
                ("n_expanded", C.c_int), ("n_rounds", C.c_int), ("root_flags", C.c_int), ("round_scenes", C.c_int * 32),
                ("pair_ms", C.c_float), ("pair_launches", C.c_int), ("n_trees", C.c_int), ("tree_top", C.POINTER(C.c_int32)),
                ("tree_off", C.POINTER(C.c_int32)), ("flat_parent", C.POINTER(C.c_int32)), ("flat_prob", C.POINTER(C.c_float)),
                ("flat_mean", C.POINTER(C.c_float)), ("flat_cov", C.POINTER(C.c_float)), ("solves_begun", C.c_int)]


class IlqrCfg(C.Structure):
    _fields_ = [("dt", C.c_double), ("wheelbase", C.c_double), ("w_des_state", C.c_double * 6),
                ("w_state_con", C.c_double * 6), ("state_lower", C.c_double * 6), ("state_upper", C.c_double * 6),
                ("w_ctrl", C.c_double * 2), ("w_tgt", C.c_double), ("w_ego", C.c_double),
                ("w_ego_cov_offset", C.c_double), ("w_exo", C.c_double), ("w_exo_cov_offset", C.c_double),
                ("w_exo_cost_offset", C.c_double), ("grid_res", C.c_double), ("grid_w", C.c_int), ("grid_h", C.c_int),
                ("max_iter", C.c_int)]


class IlqrStats(C.Structure):
    _fields_ = [("iterations", C.c_int), ("converged", C.c_int), ("J", C.c_double), ("mu", C.c_double)]


LOOP_TAN_FN = C.CFUNCTYPE(C.c_double, C.c_double)                       # mind_loop_desc.tan_fn / .sincos_fn
LOOP_SINCOS_FN = C.CFUNCTYPE(None, C.c_double, C.POINTER(C.c_double))


class LoopDesc(C.Structure):       # mind_loop_desc (include/mind_hip.h)
    _fields_ = [("n_tracks", C.c_int), ("n_steps", C.c_int), ("clamp_last", C.c_int), ("ego_state", C.c_void_p), ("ego_state_is_f32", C.c_int), ("ego_obs", C.c_void_p), ("ego_trig32", C.c_void_p),
                ("tan_fn", C.c_void_p), ("sincos_fn", C.c_void_p), ("exo_obs", C.c_void_p),
                ("exo_valid", C.c_void_p), ("timestep", C.c_void_p), ("type_slot", C.c_void_p),
                ("sim_step", C.c_double), ("plan_step", C.c_double), ("enable_time", C.c_double),
                ("wheelbase", C.c_double), ("max_speed", C.c_double), ("max_steer", C.c_double), ("max_acc", C.c_double), ("max_dec", C.c_double),
                ("n_lanes", C.c_int), ("lane_pts", C.c_void_p), ("lane_flags", C.c_void_p),
                ("n_lane_pts", C.c_int), ("target_lane", C.c_void_p), ("target_lane_info", C.c_void_p),
                ("time_ahead", C.c_double), ("min_vel", C.c_float), ("dist_thres", C.c_float), ("max_depth", C.c_int), ("max_rounds", C.c_int),
                ("pred_len", C.c_int), ("prob_floor", C.c_float),
                ("cfg_warm", C.c_void_p), ("cfg_full", C.c_void_p), ("solve_n_lane_pts", C.c_int), ("solve_lane", C.c_void_p), ("target_vel", C.c_double),
                ("eval_n_lane_pts", C.c_int), ("eval_lane_is_f32", C.c_int), ("eval_lane", C.c_void_p), ("speculative", C.c_int)]


class LoopTotals(C.Structure):     # mind_loop_totals
    _fields_ = [("plans", C.c_longlong), ("expansions", C.c_longlong), ("scen_trees", C.c_longlong), ("rounds", C.c_longlong),
                ("aime_s", C.c_double), ("ilqr_s", C.c_double), ("total_s", C.c_double),
                ("iterations", C.c_longlong), ("node_iterations", C.c_longlong), ("node_iterations_exo", C.c_longlong),
                ("warm_speculated", C.c_longlong), ("warm_hits", C.c_longlong),
                ("pair_ms", C.c_double), ("pair_launches", C.c_longlong), ("scene_n2", C.c_double), ("scene_n_a1", C.c_double),
                ("ilqr_ms", C.c_double), ("ilqr_launches", C.c_longlong), ("ilqr_trees", C.c_longlong), ("ilqr_workgroups_per_tree", C.c_int),
                ("ilqr_prof", C.c_double * 9), ("ilqr_node_steps", C.c_double)]


class LoopOut(C.Structure):        # mind_loop_out
    _fields_ = [("planned", C.c_int), ("enabled", C.c_int), ("n_steps", C.c_longlong), ("n_plans", C.c_longlong), ("episode_steps", C.c_longlong),
                ("sim_time", C.c_double), ("last_trigger", C.c_double), ("state", C.c_double * 4), ("ctrl", C.c_double * 2),
                ("n_agents", C.c_int), ("n_trees", C.c_int), ("best", C.c_int), ("n_expanded", C.c_int), ("n_rounds", C.c_int), ("n_traj_nodes", C.c_int),
                ("costs", C.POINTER(C.c_double)), ("aime_s", C.c_double), ("ilqr_s", C.c_double), ("total_s", C.c_double),
                ("tot", LoopTotals)]


EXPORTS = ["mind_ctx_create", "mind_ctx_destroy", "mind_last_error_string", "mind_ctx_synchronize",
           "mind_weights_load", "mind_predict_batch", "mind_last_fusion_stats", "mind_set_profiling",
           "mind_ilqr_solve_trees", "mind_ilqr_contingency", "mind_ilqr_solve_fields", "mind_cost_eval", "mind_lane_dist_field", "mind_aime_world", "mind_aime_rebase", "mind_debug_set_layers",
           "mind_debug_read", "mind_set_pair_precision", "mind_get_pair_precision", "mind_debug_pack_bfrag", "mind_debug_pack_conv_frag", "mind_debug_pair_schedule", "mind_set_tuning", "mind_last_ilqr_stats", "mind_aime_plan", "mind_last_ilqr_profile", "mind_eval_traj_trees", "mind_last_ilqr_trace", "mind_ilqr_contingency_begin", "mind_ilqr_finish", "mind_fill_tracks", "mind_ilqr_contingency_begin_plan", "mind_debug_trig", "mind_aime_plan_begin", "mind_aime_plan_poll", "mind_aime_plan_finish", "mind_ctx_busy", "mind_ilqr_finish_plan",
           "mind_set_exchange", "mind_last_exchange_stats",
           "mind_loop_create", "mind_loop_destroy", "mind_loop_reset", "mind_loop_advance", "mind_loop_state", "mind_loop_last_plan", "mind_loop_export"]

# transport of the sharded mind_aime_plan (include/mind_hip.h): int fn(void *user, int op, void *send, void *recv, int64 bytes)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64)
XCHG_ALLGATHER, XCHG_ALLREDUCE, XCHG_ALLTOALLV = 0, 1, 2

_lib = None


def load():
    """Load the library (raises if it has not been built: run ``python -c 'import __graft_entry__ as g; g.build()'``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: the HIP extension must be built (hipcc --offload-arch=gfx950); "
                          "there is no CPU fallback for the product path")
    lib = C.CDLL(LIB_PATH)
    lib.mind_ctx_create.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.mind_ctx_destroy.argtypes = [C.c_void_p]
    lib.mind_last_error_string.argtypes = [C.c_void_p]
    lib.mind_last_error_string.restype = C.c_char_p
    lib.mind_ctx_synchronize.argtypes = [C.c_void_p]
    lib.mind_weights_load.argtypes = [C.c_void_p, C.POINTER(TensorDesc), C.c_int]
    lib.mind_predict_batch.argtypes = [C.c_void_p, C.POINTER(SceneBatch), C.POINTER(PredOut)]
    lib.mind_last_fusion_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_double)]
    lib.mind_set_profiling.argtypes = [C.c_void_p, C.c_int]
    lib.mind_ilqr_solve_trees.argtypes = [C.c_void_p, C.POINTER(IlqrCfg), C.POINTER(CostTree), C.c_int,
                                          C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.c_double, C.c_int,
                                          C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                          C.POINTER(IlqrStats)]
    lib.mind_ilqr_contingency.argtypes = [C.c_void_p, C.POINTER(IlqrCfg), C.POINTER(IlqrCfg), C.POINTER(CostTree), C.c_int,
                                          C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.c_double,
                                          C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(IlqrStats), C.POINTER(IlqrStats)]
    lib.mind_ilqr_contingency_begin.argtypes = [C.c_void_p, C.POINTER(IlqrCfg), C.POINTER(IlqrCfg), C.POINTER(CostTree), C.c_int,
                                          C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.c_double,
                                          C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(IlqrStats), C.POINTER(IlqrStats)]
    lib.mind_ilqr_finish.argtypes = [C.c_void_p]
    lib.mind_ilqr_contingency_begin_plan.argtypes = [C.c_void_p, C.POINTER(IlqrCfg), C.POINTER(IlqrCfg), C.c_void_p, C.c_void_p, C.c_int, C.c_double,
                                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mind_fill_tracks.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mind_ilqr_solve_fields.argtypes = [C.c_void_p, C.POINTER(IlqrCfg), C.POINTER(FieldGrid), C.POINTER(CostTree), C.c_int,
                                           C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                           C.POINTER(C.c_double), C.POINTER(IlqrStats)]
    lib.mind_cost_eval.argtypes = [C.c_void_p, C.POINTER(IlqrCfg), C.POINTER(FieldGrid), C.POINTER(CostTree),
                                   C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.c_double, C.c_int, C.c_int,
                                   C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.mind_lane_dist_field.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int,
                                         C.c_double] + [C.POINTER(C.c_double)] * 4
    lib.mind_aime_world.argtypes = [C.c_void_p, C.POINTER(WorldIn), C.POINTER(WorldOut)]
    lib.mind_aime_rebase.argtypes = [C.c_void_p, C.POINTER(RebaseIn), C.POINTER(RebaseOut)]
    lib.mind_aime_plan.argtypes = [C.c_void_p, C.POINTER(AimePlanIn), C.POINTER(AimePlanOut)]
    lib.mind_aime_plan_begin.argtypes = [C.c_void_p, C.POINTER(AimePlanIn)]
    lib.mind_aime_plan_poll.argtypes = [C.c_void_p]
    lib.mind_aime_plan_finish.argtypes = [C.c_void_p, C.POINTER(AimePlanOut)]
    lib.mind_ctx_busy.argtypes = [C.c_void_p]
    lib.mind_ilqr_finish_plan.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mind_set_pair_precision.argtypes = [C.c_void_p, C.c_int]
    lib.mind_set_tuning.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    lib.mind_last_ilqr_stats.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.mind_last_ilqr_profile.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    lib.mind_last_ilqr_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]
    lib.mind_eval_traj_trees.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_int, C.c_void_p, C.c_int, C.c_int,
                                         C.c_double, C.POINTER(C.c_double)]
    lib.mind_get_pair_precision.argtypes = [C.c_void_p]
    lib.mind_debug_pack_bfrag.argtypes = [C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_uint32)]
    lib.mind_debug_pair_schedule.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]
    lib.mind_debug_trig.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double)]
    lib.mind_debug_pack_conv_frag.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.c_size_t]
    lib.mind_debug_set_layers.argtypes = [C.c_void_p, C.c_int]
    lib.mind_debug_read.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_float), C.c_int64]
    lib.mind_set_exchange.argtypes = [C.c_void_p, C.c_int, C.c_int, EXCHANGE_FN, C.c_void_p, C.c_int]
    lib.mind_last_exchange_stats.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    lib.mind_loop_create.argtypes = [C.c_void_p, C.POINTER(LoopDesc), C.POINTER(C.c_void_p)]
    lib.mind_loop_destroy.argtypes = [C.c_void_p]
    lib.mind_loop_reset.argtypes = [C.c_void_p]
    lib.mind_loop_advance.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_longlong, C.POINTER(LoopOut)]
    lib.mind_loop_state.argtypes = [C.c_void_p, C.POINTER(LoopOut)]
    lib.mind_loop_last_plan.argtypes = [C.c_void_p, C.POINTER(AimePlanOut)] + [C.POINTER(C.c_void_p)] * 6 + [C.c_void_p]
    lib.mind_loop_export.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_void_p]
    for n in EXPORTS:
        getattr(lib, n)
        if n not in ("mind_last_error_string", "mind_debug_read"):
            getattr(lib, n).restype = C.c_int
    lib.mind_debug_read.restype = C.c_int64
    _lib = lib
    return lib


class MindError(RuntimeError):
    pass


def check(lib, ctx, rc, what):
    if rc != MIND_OK:
        msg = lib.mind_last_error_string(ctx) if ctx else b""
        raise MindError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")
