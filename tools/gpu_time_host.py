"""GPU-box diagnostic: wall-clock of the host-side sections of a planning cycle (method-level timers, no profiler)."""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import DEEP, FULL_TREE, WORKLOADS, make_closed_loop

acc = collections.OrderedDict()
def wrap(obj, name, label=None, sync=False):
    f = getattr(obj, name)
    label = label or name
    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        if sync:
            torch.cuda.synchronize()
        acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
        return r
    setattr(obj, name, g)

wl = sys.argv[1] if len(sys.argv) > 1 else "demo1"
ckpt = sys.argv[3] if len(sys.argv) > 3 else None
pl, sim, w = make_closed_loop(dict(WORKLOADS[wl]), full_tree=DEEP.get(wl, wl in FULL_TREE), ckpt=ckpt, speculative=os.environ.get("SPEC", "1") == "1")
gen, net, rt, opt = pl.scen_tree_gen, pl.scen_tree_gen.network, pl.network.rt, pl.traj_tree_opt
sim.run_plans(int(os.environ.get("WARM", "3")))
wrap(pl, "plan"); wrap(gen, "branch_aime"); wrap(gen, "process_data"); wrap(gen, "collate"); wrap(net, "pre_process")
wrap(rt, "predict", "rt.predict(launch)"); wrap(rt, "aime_world", "rt.aime_world(sync+launch)"); wrap(gen, "prune_select"); wrap(gen, "assemble_children"); wrap(gen, "predict_inputs"); wrap(rt, "aime_rebase"); wrap(rt, "ilqr_solve"); wrap(opt, "speculate_warm"); wrap(pl, "resample_target_lane")
wrap(gen, "create_nodes"); wrap(gen, "decide_branch"); wrap(gen, "update_obser_batch"); wrap(gen, "get_scenario_tree")
wrap(rt, "ilqr_contingency"); wrap(opt, "solve_batch"); wrap(pl, "evaluate_traj_trees"); wrap(pl, "update_observation")
wrap(sim, "_observation", "sim._observation"); wrap(gen, "prepare_root_data"); wrap(gen, "_select_modes")
import mind_amd.planners.mind.trajectory_tree as TT
import mind_amd.predictor as PR
wrap(TT, "flatten_scenario_tree"); wrap(TT, "to_traj_tree"); wrap(TT, "ilqr_cfg_from")
wrap(PR.IlqrCall, "__init__", "IlqrCall.__init__"); wrap(PR.IlqrCall, "run", "IlqrCall.run (C call: prep + kernel + copies)"); wrap(PR.IlqrCall, "finish", "IlqrCall.finish")
wrap(opt, "_take_speculation")
import mind_amd.planners.mind.utils as UU
for fn in ("get_agent_trajectories", "normalize_agents", "lane_graph_from_map", "lane_features", "actor_features"):
    if hasattr(UU, fn): wrap(UU, fn, "U." + fn)
wrap(gen, "_scene_inputs"); wrap(gen, "get_branch_times"); wrap(gen, "_prune_select_device"); wrap(gen, "_hdr")
class _LibTimer:
    def __init__(self, fn, label):
        self.fn, self.label = fn, label
    def __call__(self, *a):
        t0 = time.perf_counter()
        r = self.fn(*a)
        acc[self.label] = acc.get(self.label, 0.0) + time.perf_counter() - t0
        return r
rt.lib.mind_aime_rebase = _LibTimer(rt.lib.mind_aime_rebase, "C: mind_aime_rebase")
rt.lib.mind_aime_world = _LibTimer(rt.lib.mind_aime_world, "C: mind_aime_world")
rt.lib.mind_predict_batch = _LibTimer(rt.lib.mind_predict_batch, "C: mind_predict_batch")
wrap(gen, "_update_obser_device_windows"); wrap(gen, "_branch_aime_native"); wrap(rt, "aime_plan", "rt.aime_plan (C call + marshalling)")
rt.lib.mind_aime_plan = _LibTimer(rt.lib.mind_aime_plan, "C: mind_aime_plan")
wrap(opt, "solve_batch_begin", "solve_batch_begin (tables + upload + launch of the contingency solves, returns with the kernel queued)")
rt.lib.mind_ilqr_contingency_begin_plan = _LibTimer(rt.lib.mind_ilqr_contingency_begin_plan, "C: mind_ilqr_contingency_begin_plan")
rt.lib.mind_ilqr_finish = _LibTimer(rt.lib.mind_ilqr_finish, "C: mind_ilqr_finish (waits for the kernel)")
rt.lib.mind_ilqr_finish_plan = _LibTimer(rt.lib.mind_ilqr_finish_plan, "C: mind_ilqr_finish_plan (waits for the kernel)")
import mind_amd._lib as _L
_L.load().mind_eval_traj_trees = _LibTimer(_L.load().mind_eval_traj_trees, "C: mind_eval_traj_trees")
_L.load().mind_fill_tracks = _LibTimer(_L.load().mind_fill_tracks, "C: mind_fill_tracks")
wrap(gen, "_native_args"); wrap(gen, "_native_trees"); wrap(pl, "_solve_and_select"); wrap(pl, "_plan_begin"); wrap(pl, "_solve_hooks")
wrap(sim, "step_begin", "sim.step_begin"); wrap(sim, "step_end", "sim.step_end")
import mind_amd.closed_loop as CLm
wrap(CLm, "kine_propagate")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
t0 = time.perf_counter()
sim.run_plans(n)
tot = time.perf_counter() - t0
print("cycle %.2f ms" % (tot / n * 1e3))
for k, v in acc.items():
    print("  %-34s %7.3f ms" % (k, v / n * 1e3))
