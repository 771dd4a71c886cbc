"""Generate tests/golden/*.npz from the IMPORTED REFERENCE (build container only).

The reference Python cannot travel to the GPU box; these small fixtures (inputs are formula
generated, so only expected outputs are stored) can.  Re-run:  python tests/golden/gen_golden.py [section ...]
Sections: predictor rpe potential ilqr aime plan scenes demo_plans demo_branch demo_runs demo_branch_runs demo_traces demo_branch_traces
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from oracle import ref_harness as rh
from mind_amd.weights import formula_state_dict
from mind_amd.synth import predictor_batch

GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)
PRED_CASES = [(3, 4, 1, 1), (8, 20, 2, 1), (40, 55, 1, 1), (2, 2, 1, 3)]  # l=1 crashes the reference (Q7 .squeeze())


def to_t(pb):
    return {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else [torch.from_numpy(x) for x in v])
            for k, v in pb.items()}


def gen_predictor():
    m = rh.ref_modules()
    sd = formula_state_dict(as_torch=True)
    net = rh.build_ref_network(sd)
    get_rpe = m["planners.mind.utils"].get_rpe
    out = {}
    for (a, l, B, seed) in PRED_CASES:
        pb = to_t(predictor_batch(a, l, B, seed=seed))
        rpes = [{"scene": get_rpe(c, v)[0], "scene_mask": None} for c, v in zip(pb["CTRS"], pb["VECS"])]
        data = (pb["ACTORS"], pb["ACTOR_IDCS"], pb["LANES"], pb["LANE_IDCS"], rpes, pb["TGT_NODES"], pb["TGT_RPE"])
        with torch.no_grad():
            act = net.actor_net(pb["ACTORS"])
            lan = net.lane_net(pb["LANES"])
        rc, rr, ra = net(data)   # autograd on, as the reference runs it (scenario_tree.py:69-71)
        key = f"a{a}_l{l}_b{B}_s{seed}"
        out[key + "_actor_net"] = act.numpy()
        out[key + "_lane_net"] = lan.numpy().reshape(-1, 128)
        out[key + "_cls"] = np.stack([c.detach().numpy()[0] for c in rc])
        reg = np.concatenate([r.detach().numpy() for r in rr])
        vel = np.concatenate([x[0].detach().numpy() for x in ra])
        if a >= 40:   # keep the fixture small: every 5th step + the last
            idx = np.r_[0:60:5, 59]
            out[key + "_tidx"] = idx
            reg, vel = reg[:, :, idx], vel[:, :, idx]
        out[key + "_reg"] = reg
        out[key + "_vel"] = vel
        out[key + "_rpe0"] = rpes[0]["scene"].numpy() if a + l <= 30 else rpes[0]["scene"].numpy()[:, :8, :8]
    np.savez_compressed(os.path.join(GOLD, "predictor.npz"), **out)
    print("predictor.npz", sum(v.nbytes for v in out.values()) // 1024, "KiB raw")


# ----------------------------------------------------------------------------------------------
ILQR_CASES = [("straight", 3, 100), ("lead", 4, 100), ("branch3", 6, 100), ("branch3", 40, 100), ("deep", 3, 4)]


def _ref_tree(nodes):
    m = rh.ref_modules()
    Tree, Node = m["planners.basic.tree"].Tree, m["planners.basic.tree"].Node
    t = Tree()
    for k, p, d in nodes:
        t.add_node(Node(k, p, d))
    return t


def gen_ilqr():
    """G5/G6: cost-tree construction order + iLQR results of the reference TrajectoryTreeOptimizer
    (closed-form bicycle stand-in for the Theano dynamics) on scripted scenario trees."""
    from mind_amd.synth import scripted_scenario_tree
    m = rh.ref_modules()
    TTO = m["planners.mind.trajectory_tree"].TrajectoryTreeOptimizer
    cfgmod = m["planners.mind.configs.planning.demo_1"]
    out = {}
    for kind, a, max_iter in ILQR_CASES:
        sst = scripted_scenario_tree(kind, a)
        tree = _ref_tree(sst["nodes"])
        opt = TTO(cfgmod.TrajTreeCfg())
        opt.init_warm_start_cost_tree(tree, sst["state"], sst["ctrl"], sst["target_lane"], sst["target_vel"])
        M = opt.cost_tree.tree.size() - 1
        xs_w, us_w = opt.ilqr.fit(np.zeros((M, 2)), opt.cost_tree, n_iterations=max_iter)
        xs_w, us_w = xs_w.copy(), us_w.copy()
        Jw, muw = opt.ilqr.J_opt, opt.ilqr._mu
        opt.init_cost_tree(tree, sst["state"], sst["ctrl"], sst["target_lane"], sst["target_vel"])
        parents = np.array([opt.cost_tree.tree.nodes[k].parent_key for k in range(M)], np.int32)
        xs_f, us_f = opt.ilqr.fit(us_w, opt.cost_tree, n_iterations=max_iter)
        key = f"{kind}_a{a}_it{max_iter}"
        # a few per-node costs at the final iterate (field spot values, G5)
        Lf = np.array([float(np.ravel(opt.cost_tree.l(xs_f[i], us_f[i], i))[0]) for i in range(M)])
        out.update({key + "_parent": parents, key + "_xs_w": xs_w, key + "_us_w": us_w, key + "_Jw": np.array([Jw, muw]),
                    key + "_xs_f": xs_f.copy(), key + "_us_f": us_f.copy(),
                    key + "_Jf": np.array([opt.ilqr.J_opt, opt.ilqr._mu]), key + "_Lf": Lf})
        print(key, "M", M, "Jw %.8g Jf %.8g" % (Jw, opt.ilqr.J_opt))
    np.savez_compressed(os.path.join(GOLD, "ilqr.npz"), **out)


def gen_potential():
    """G4: PotentialField value / gradient / Hessian on a random field incl. all 8 border cases and
    .5 rounding ties (banker's rounding)."""
    m = rh.ref_modules()
    PF = m["planners.ilqr.potential"].PotentialField
    rng = np.random.default_rng(7)
    W = H = 32
    res = 0.4
    off = np.array([3.0, -2.0])
    x = np.linspace(0.0, (W - 1) * res, W) + off[0]
    y = np.linspace(0.0, (H - 1) * res, H) + off[1]
    xx, yy = np.meshgrid(x, y)
    F = rng.random((H, W)) * 10.0
    pf = PF(off, res, xx, yy, F)
    pts = []
    for ix in (0, 1, 15, W - 2, W - 1):
        for iy in (0, 1, 15, H - 2, H - 1):
            pts.append([x[ix] + 0.13, y[iy] - 0.11])
    pts += [[x[5] + 0.5 * res, y[6]], [x[6] + 0.5 * res, y[7] + 0.5 * res], [x[0] - 3.0, y[3]], [x[W - 1] + 2.0, y[H - 1] + 5.0]]
    pts += list(np.stack([rng.uniform(x[0], x[-1], 40), rng.uniform(y[0], y[-1], 40)], 1))
    pts = np.array(pts)
    vals = []
    for p in pts:
        st = np.array([p[0], p[1], 0, 0, 0, 0.0])
        g = pf.get_gradient(st)
        h = pf.get_hessian(st)
        vals.append([pf.get_potential(st), g[0], g[1], h[0, 0], h[1, 1], h[0, 1]])
    np.savez_compressed(os.path.join(GOLD, "potential.npz"), F=F, off=off, res=np.array(res), pts=pts, vals=np.array(vals))
    print("potential.npz", len(pts), "points")


AIME_CASES = [
    ("w6", dict(n_agents=6, n_lanes=3, n_segs=8, seed=1), dict()),
    ("w3_nogrowth", dict(n_agents=3, n_lanes=2, n_segs=6, seed=2), dict(growth=(0.02,) * 6)),
    ("w9_branch", dict(n_agents=9, n_lanes=3, n_segs=8, seed=3),
     dict(lateral=(0.0, 9.0, -9.0, 0.1, -6.0, 0.2), growth=(0.3, 0.25, 0.1, 0.02, 0.4, 0.03))),
    ("w5_deep", dict(n_agents=5, n_lanes=3, n_segs=8, seed=4),
     dict(lateral=(0.0, 9.0, -9.0, 4.0, -6.0, 0.2), growth=(0.5, 0.45, 0.4, 0.5, 0.4, 0.3),
          probs=(0.3, 0.25, 0.2, 0.15, 0.0995, 0.0005))),
    ("w1_ego_only", dict(n_agents=1, n_lanes=2, n_segs=6, seed=5), dict()),
]


def synth_world(kw):
    from mind_amd.synth import SynthWorld
    return SynthWorld(object_type_cls=rh.ObjectType, lane_type_cls=rh.LaneType, lane_mark_cls=rh.LaneMarkType, **kw)


def gen_aime():
    """G3: AIME decisions of the reference ScenarioTreeGenerator driven by the scripted FakeNet."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from fake_net import FakeNet
    m = rh.ref_modules()
    STG = m["planners.mind.scenario_tree"].ScenarioTreeGenerator
    RefPlanner = m["planners.mind.planner"].MINDPlanner
    cfgmod = m["planners.mind.configs.planning.demo_1"]
    out = {}
    for name, wkw, nkw in AIME_CASES:
        w = synth_world(wkw)
        lcl = w.local_semantic_map(4.9)
        obs = w.tracks(4.9, state_cls=rh.ObjectState, track_cls=rh.Track, drop={2: 30} if wkw["n_agents"] > 2 else None)
        lane, info = RefPlanner.resample_target_lane(RefPlanner.__new__(RefPlanner), lcl)
        net = FakeNet(**nkw)
        g = STG(torch.device("cpu"), net, 50, 50, cfgmod.ScenTreeCfg())
        g.reset()
        g.set_target_lane(lane, info)
        trees = g.branch_aime(lcl, obs)
        out[name + "_batches"] = np.array(net.calls)
        out[name + "_internal_keys"] = np.array(list(g.tree.nodes.keys()))
        out[name + "_internal_flags"] = np.array([[n.data.branch_flag, n.data.end_flag, n.data.terminate_flag]
                                                  for n in g.tree.nodes.values()])
        out[name + "_ntrees"] = np.array(len(trees))
        for ti, t in enumerate(trees):
            keys = list(t.nodes.keys())
            out[f"{name}_t{ti}_keys"] = np.array(keys)
            out[f"{name}_t{ti}_parents"] = np.array([str(t.nodes[k].parent_key) for k in keys])
            out[f"{name}_t{ti}_probs"] = np.array([float(np.ravel(t.nodes[k].data[0])[0]) for k in keys], np.float64)
            out[f"{name}_t{ti}_durs"] = np.array([t.nodes[k].data[1].shape[1] for k in keys])
            for k in keys:
                d = t.nodes[k].data
                out[f"{name}_t{ti}_{k}_pos"] = d[1][:, ::5]      # every 5th step keeps the fixture small
                out[f"{name}_t{ti}_{k}_cov"] = d[2][:, ::5]
                out[f"{name}_t{ti}_{k}_tgt"] = np.asarray(d[3])
        print(name, "batches", net.calls, "trees", [list(t.nodes.keys()) for t in trees])
    np.savez_compressed(os.path.join(GOLD, "aime.npz"), **out)


PLAN_CASES = [("p6", dict(n_agents=6, n_lanes=3, n_segs=8, seed=1)), ("p12", dict(n_agents=12, n_lanes=3, n_segs=10, seed=2))]


def gen_plan():
    """G7: one full reference MINDPlanner.plan() (torch-CPU predictor with the formula weights,
    AIME, python tree-iLQR) on synthetic worlds: chosen tree, control, trees."""
    import json
    import tempfile
    m = rh.ref_modules()
    RefPlanner = m["planners.mind.planner"].MINDPlanner
    tmp = tempfile.mkdtemp()
    ck = os.path.join(tmp, "formula.tar")
    torch.save({"state_dict": formula_state_dict(as_torch=True)}, ck)
    cfgp = os.path.join(tmp, "cfg.json")
    json.dump({"use_cuda": False, "network_config": "planners.mind.configs.networks.net_cfg", "ckpt_path": ck,
               "planning_config": "planners.mind.configs.planning.demo_1"}, open(cfgp, "w"))
    out = {}
    for name, wkw in PLAN_CASES:
        w = synth_world(wkw)
        pl = RefPlanner(cfgp)
        # 50 observation updates at 10 Hz ending at t = 4.9 s (as agent.py does before planning)
        for s in range(50):
            pl.update_observation(w.local_semantic_map(round(0.1 * s, 6)))
        lcl = w.local_semantic_map(4.9)
        pl.update_target_lane(np.asarray(w.target_lane[::2], dtype=np.float64))
        pl.update_state_ctrl(lcl.ego_agent.state, np.array([0.0, 0.0]))
        ok, ctrl, (st, tt) = pl.plan(lcl)
        st, tt = st[0], tt[0]
        out[name + "_ctrl"] = np.asarray(ctrl, np.float64)
        keys = list(st.nodes.keys())
        out[name + "_scen_keys"] = np.array(keys)
        out[name + "_scen_probs"] = np.array([float(np.ravel(st.nodes[k].data[0])[0]) for k in keys])
        for k in keys:
            out[f"{name}_scen_{k}_pos"] = st.nodes[k].data[1][:, ::5]
            out[f"{name}_scen_{k}_cov"] = st.nodes[k].data[2][:, ::5]
        tk = [k for k in tt.nodes.keys() if k != -1]
        out[name + "_traj_xs"] = np.array([tt.nodes[k].data[0] for k in tk])
        out[name + "_traj_us"] = np.array([tt.nodes[k].data[1] for k in tk])
        out[name + "_traj_parent"] = np.array([tt.nodes[k].parent_key for k in tk])
        print(name, "ctrl", ctrl, "scen tree", keys, "traj nodes", len(tk))
    np.savez_compressed(os.path.join(GOLD, "plan.npz"), **out)


def gen_scenes():
    """G8: the four recorded demo scenes.
    (1) tests/golden/scenes/demo_N.npz: compact array form of the scene (lane segments + 10 Hz tracks + the
        closed-loop agent entry of configs/demo_N.json), derived with mind_amd.av2_lite from data/<seq_id>/.
    (2) tests/golden/scene_io.npz: what the reference's own scene I/O (SemanticMap, ArgoAgentLoader,
        MINDAgent target lane) produces on those scenes, running on top of the av2_lite readers."""
    import importlib
    import json
    from mind_amd import av2_lite
    from mind_amd.scene_io import DEMO_SCENES
    rh.install()
    SemanticMap = importlib.import_module("common.semantic_map").SemanticMap
    Loader = importlib.import_module("loader").ArgoAgentLoader
    agent_mod = importlib.import_module("agent")
    geom = importlib.import_module("common.geometry")
    os.makedirs(os.path.join(GOLD, "scenes"), exist_ok=True)
    out = {}
    for name, seq in DEMO_SCENES.items():
        d = os.path.join(rh.REF_ROOT, "data", seq)
        mp, sp = os.path.join(d, f"log_map_archive_{seq}.json"), os.path.join(d, f"scenario_{seq}.parquet")
        cfg = json.load(open(os.path.join(rh.REF_ROOT, "configs", name + ".json")))
        cl = {k: cfg["cl_agents"][0][k] for k in ("id", "enable_timestep", "semantic_lane", "target_velocity")}
        av2_lite.save_scene(os.path.join(GOLD, "scenes", name + ".npz"), av2_lite.StaticMap.from_json(mp),
                            av2_lite.load_argoverse_scenario_parquet(sp), meta=dict(cl_agent=json.dumps(cl), seq_id=seq))
        from pathlib import Path
        smp = SemanticMap()
        smp.load_from_argo2(Path(mp))
        out[name + "_n_sem"] = np.array(len(smp.semantic_lanes))
        out[name + "_sem_len"] = np.array([len(v) for v in smp.semantic_lanes.values()])
        out[name + "_sem_pts"] = np.concatenate(list(smp.semantic_lanes.values()))
        for c, nm in enumerate(("intersect", "lane_type", "cross_left", "cross_right", "left", "right")):
            out[f"{name}_sem_{nm}"] = np.concatenate([v[c] for v in smp.semantic_lanes_infos.values()]).astype(np.int8)
        out[name + "_limits"] = np.array(smp.limits)
        pos, ang, vel, types, tids, cats, flags = Loader(Path(sp)).get_trajs_info(smp)
        out[name + "_tids"] = np.array(tids)
        out[name + "_cats"] = np.array(cats)
        out[name + "_types"] = np.array([t[0].name for t in types])
        out[name + "_shape"] = np.array(pos.shape)
        out[name + "_pos7"], out[name + "_ang7"], out[name + "_vel7"] = pos[:, ::7], ang[:, ::7], vel[:, ::7]
        out[name + "_flags"] = np.packbits(flags.astype(bool), axis=1)
        out[name + "_sums"] = np.array([pos.astype(np.float64).sum(), ang.astype(np.float64).sum(),
                                        vel.astype(np.float64).sum(), flags.sum()])
        k = tids.index("AV")
        lane_id = None if cl["semantic_lane"] == -1 else cl["semantic_lane"]
        tv = None if cl["target_velocity"] == -1 else cl["target_velocity"]
        ag = agent_mod.MINDAgent()
        ag.init("AV", types[k], cats[k], [pos[k], ang[k], vel[k], flags[k]], smp, None, semantic_lane_id=lane_id,
                target_velocity=tv)
        out[name + "_closest_lane"] = np.array(-1 if (c := ag.get_closest_semantic_lane(smp, pos[k], ang[k])) is None else c)
        out[name + "_target_lane"] = np.asarray(ag.lcl_smp.target_lane)
        out[name + "_target_velocity"] = np.array(float(ag.lcl_smp.target_velocity))
        gt, _ = ag.get_target_lane(smp, True, lane_id)
        out[name + "_gt_tgt_lane"] = geom.remove_close_points(gt, 4.0)
        print(name, "sem lanes", len(smp.semantic_lanes), "tracks kept", len(tids), "of",
              len(av2_lite.load_argoverse_scenario_parquet(sp).tracks), "closest lane", out[name + "_closest_lane"],
              "gt lane pts", len(out[name + "_gt_tgt_lane"]))
    np.savez_compressed(os.path.join(GOLD, "scene_io.npz"), **out)


def gen_demo_branch(n_plans=12):
    """G11: the same as demo_plans with the BRANCHING formula weights (mind_amd.weights variant "branching"): the reference's
    AIME tree then keeps several modes and runs two rounds per plan on every recorded scene (6 expansions on demo_1), so
    the branch-selection parity of the four demo scenes is exercised on multi-node trees.  Also stores, per plan, every
    node of the internal AIME tree (ids, END_T, flags) via the number of nodes, and every returned scenario tree."""
    gen_demo_plans(n_plans, variant="branching", fname="demo_branch.npz")


def gen_demo_branch_runs(n_plans=60):
    """G12: the reference's WHOLE closed loop (t = 4.0 .. 9.9 s, 60 planning cycles) on the four recorded scenes with the
    branching formula weights -- discrete results only (every scenario tree's keys, every internal AIME node with END_T and end
    flag, candidate costs, chosen tree, the ego state / control planned from), for a teacher-forced comparison of the
    branch-selection indices over entire runs."""
    gen_demo_plans(n_plans, variant="branching", fname="demo_branch_runs.npz", slim=True)


def gen_demo_plans(n_plans=4, variant=None, fname="demo_plans.npz", slim=False):
    """G9: the reference's OWN closed loop (Simulator.run_sim, headless) on its four recorded demo scenes up to the
    first `n_plans` planning cycles (t = 4.0 s, 4.1 s, ...), with the formula weights (the trained checkpoint is not in the
    reference tree): control, chosen scenario/trajectory trees and the ego state after each cycle."""
    import importlib
    import json
    import tempfile
    from mind_amd.scene_io import DEMO_SCENES
    rh.install()
    os.chdir(rh.REF_ROOT)                       # the simulator opens 'data/<seq_id>/...' relative to its cwd (read only)
    import types
    vis = types.ModuleType("common.visualization")     # rendering (matplotlib + shapely polygons) is never reached headless
    for fn in ("draw_map", "draw_agent", "draw_scen_trees", "reset_ax", "draw_traj_trees", "draw_traj"):
        setattr(vis, fn, None)
    sys.modules["common.visualization"] = vis
    Simulator = importlib.import_module("simulator").Simulator
    tmp = tempfile.mkdtemp()
    ck = os.path.join(tmp, "formula.tar")
    torch.save({"state_dict": formula_state_dict(as_torch=True, variant=variant)}, ck)
    out = {}
    # every plan's whole AIME tree (not only the best scenario tree the simulator keeps): recorded around branch_aime
    stm = importlib.import_module("planners.mind.scenario_tree")
    all_trees = []
    if not hasattr(stm.ScenarioTreeGenerator, "_orig_branch_aime"):
        stm.ScenarioTreeGenerator._orig_branch_aime = stm.ScenarioTreeGenerator.branch_aime
    def branch_aime(self, lcl_smp, agent_obs):
        trees = stm.ScenarioTreeGenerator._orig_branch_aime(self, lcl_smp, agent_obs)
        all_trees.append(([list(t.nodes.keys()) for t in trees],
                          sorted((k, int(n.data.data["END_T"]) if n.data.data is not None else -1, bool(n.data.end_flag))
                                 for k, n in self.tree.nodes.items() if k != "root")))
        return trees
    stm.ScenarioTreeGenerator.branch_aime = branch_aime
    # per plan: the ego state / control the reference planned from (for teacher forcing) and the cost of EVERY candidate
    # trajectory tree (planner.py:131-137), so that a near-tie between two candidates can be told from a real disagreement
    agent_mod = importlib.import_module("agent")
    plm = importlib.import_module("planners.mind.planner")
    plan_in, plan_costs = [], []
    if not hasattr(agent_mod.MINDAgent, "_orig_plan"):
        agent_mod.MINDAgent._orig_plan = agent_mod.MINDAgent.plan
        plm.MINDPlanner._orig_eval = plm.MINDPlanner.evaluate_traj_tree
    def rec_plan(self):
        plan_in.append((np.array(self.lcl_smp.ego_agent.state, np.float64), np.array(self.ctrl, np.float64)))
        plan_costs.append([])
        return agent_mod.MINDAgent._orig_plan(self)
    def rec_eval(self, lcl_smp, traj_tree):
        c = plm.MINDPlanner._orig_eval(self, lcl_smp, traj_tree)
        plan_costs[-1].append(float(c))
        return c
    agent_mod.MINDAgent.plan = rec_plan
    plm.MINDPlanner.evaluate_traj_tree = rec_eval
    for name in DEMO_SCENES:
        cfg = json.load(open(os.path.join(rh.REF_ROOT, "configs", name + ".json")))
        pcfg = json.load(open(os.path.join(rh.REF_ROOT, cfg["cl_agents"][0]["planner_config"])))
        pcfg.update(use_cuda=False, ckpt_path=ck)
        pp = os.path.join(tmp, name + "_planner.json")
        json.dump(pcfg, open(pp, "w"))
        cfg["cl_agents"][0]["planner_config"] = pp
        cfg.update(render=False, output_dir=tmp)
        cp = os.path.join(tmp, name + ".json")
        json.dump(cfg, open(cp, "w"))
        del all_trees[:]
        del plan_in[:]
        del plan_costs[:]
        sim = Simulator(cp)
        sim.init_sim()
        sim.sim_horizon = 201 + 5 * (n_plans - 1)
        sim.run_sim()
        ego = [a for a in sim.agents if a.id == "AV"][0]
        planned = [i for i, f in enumerate(sim.frames) if "scen_tree" in f]
        assert len(planned) == n_plans == len(all_trees), planned
        assert len(plan_in) == n_plans == len(plan_costs)
        out[name + "_state_in"] = np.array([p[0] for p in plan_in])
        out[name + "_ctrl_in"] = np.array([p[1] for p in plan_in])
        for pi, (tree_keys, nodes) in enumerate(all_trees):
            out[f"{name}_p{pi}_tree_costs"] = np.array(plan_costs[pi])
            out[f"{name}_p{pi}_all_tree_keys"] = np.array(["|".join(k) for k in tree_keys])
            out[f"{name}_p{pi}_all_node_ids"] = np.array([n[0] for n in nodes])
            out[f"{name}_p{pi}_all_node_end_t"] = np.array([n[1] for n in nodes])
            out[f"{name}_p{pi}_all_node_end_flag"] = np.array([n[2] for n in nodes])
        out[name + "_plan_steps"] = np.array(planned)
        out[name + "_final_state"] = np.array(ego.state, np.float64)
        out[name + "_final_ctrl"] = np.array(ego.ctrl, np.float64)
        out[name + "_n_tracked"] = np.array(len(ego.planner.agent_obs))
        for pi, fi in enumerate(planned):
            st, tt = sim.frames[fi]["scen_tree"][0], sim.frames[fi]["traj_tree"][0]
            keys = list(st.nodes.keys())
            out[f"{name}_p{pi}_scen_keys"] = np.array(keys)
            out[f"{name}_p{pi}_scen_probs"] = np.array([float(np.ravel(st.nodes[k].data[0])[0]) for k in keys])
            tk = [k for k in tt.nodes.keys() if k != -1]
            if not slim:
                for k in keys:
                    out[f"{name}_p{pi}_scen_{k}_pos"] = st.nodes[k].data[1][:, ::5]
                    out[f"{name}_p{pi}_scen_{k}_cov"] = st.nodes[k].data[2][:, ::5]
                out[f"{name}_p{pi}_traj_xs"] = np.array([tt.nodes[k].data[0] for k in tk])
                out[f"{name}_p{pi}_traj_us"] = np.array([tt.nodes[k].data[1] for k in tk])
                out[f"{name}_p{pi}_traj_parent"] = np.array([tt.nodes[k].parent_key for k in tk])
            else:
                out[f"{name}_p{pi}_ctrl_out"] = np.array(tt.nodes[tk[0]].data[0][-2:], np.float64) if tk else np.zeros(2)
            out[f"{name}_p{pi}_n_scen_trees"] = np.array(len(sim.frames[fi]["scen_tree"]))
            print(name, "plan", pi, "step", fi, "scen keys", keys, "traj nodes", len(tk), "agents",
                  st.nodes[keys[0]].data[1].shape[0])
        print(name, "final state", ego.state, "ctrl", ego.ctrl)
    stm.ScenarioTreeGenerator.branch_aime = stm.ScenarioTreeGenerator._orig_branch_aime
    agent_mod.MINDAgent.plan = agent_mod.MINDAgent._orig_plan
    plm.MINDPlanner.evaluate_traj_tree = plm.MINDPlanner._orig_eval
    np.savez_compressed(os.path.join(GOLD, fname), **out)


def gen_demo_runs(n_plans=60):
    """G10: the reference's whole closed loop on the four recorded scenes (every planning cycle from t = 4.0 s to
    t = 9.9 s), for TEACHER-FORCED comparison: per cycle the ego state and control the reference planned from, and
    what it planned (branch ids, probabilities, ego trajectory, control, coarse agent trajectories)."""
    import importlib
    import json
    import tempfile
    import types
    from mind_amd.scene_io import DEMO_SCENES
    rh.install()
    os.chdir(rh.REF_ROOT)
    vis = types.ModuleType("common.visualization")
    for fn in ("draw_map", "draw_agent", "draw_scen_trees", "reset_ax", "draw_traj_trees", "draw_traj"):
        setattr(vis, fn, None)
    sys.modules["common.visualization"] = vis
    Simulator = importlib.import_module("simulator").Simulator
    agent_mod = importlib.import_module("agent")
    tmp = tempfile.mkdtemp()
    ck = os.path.join(tmp, "formula.tar")
    torch.save({"state_dict": formula_state_dict(as_torch=True)}, ck)
    rec = []
    orig_plan = agent_mod.MINDAgent.plan

    def recording_plan(self):
        before = (np.array(self.lcl_smp.ego_agent.state, np.float64), np.array(self.ctrl, np.float64))
        ok, res = orig_plan(self)
        rec.append((before, np.array(self.ctrl, np.float64), res))
        return ok, res

    agent_mod.MINDAgent.plan = recording_plan
    out = {}
    for name in DEMO_SCENES:
        cfg = json.load(open(os.path.join(rh.REF_ROOT, "configs", name + ".json")))
        pcfg = json.load(open(os.path.join(rh.REF_ROOT, cfg["cl_agents"][0]["planner_config"])))
        pcfg.update(use_cuda=False, ckpt_path=ck)
        pp = os.path.join(tmp, name + "_planner.json")
        json.dump(pcfg, open(pp, "w"))
        cfg["cl_agents"][0]["planner_config"] = pp
        cfg.update(render=False, output_dir=tmp)
        cp = os.path.join(tmp, name + ".json")
        json.dump(cfg, open(cp, "w"))
        del rec[:]
        sim = Simulator(cp)
        sim.init_sim()
        sim.sim_horizon = 201 + 5 * (n_plans - 1)
        sim.run_sim()
        assert len(rec) == n_plans, len(rec)
        out[name + "_state_in"] = np.array([r[0][0] for r in rec])
        out[name + "_ctrl_in"] = np.array([r[0][1] for r in rec])
        out[name + "_ctrl_out"] = np.array([r[1] for r in rec])
        out[name + "_n_scen_trees"] = np.array([len(r[2][0]) for r in rec])
        out[name + "_scen_keys"] = np.array(["|".join(r[2][0][0].nodes.keys()) for r in rec])
        out[name + "_n_agents"] = np.array([next(iter(r[2][0][0].nodes.values())).data[1].shape[0] for r in rec])
        out[name + "_root_prob"] = np.array([float(np.ravel(next(iter(r[2][0][0].nodes.values())).data[0])[0]) for r in rec])
        xs = []
        for pi, r in enumerate(rec):
            st, tt = r[2][0][0], r[2][1][0]
            tk = [k for k in tt.nodes.keys() if k != -1]
            xs.append(np.array([tt.nodes[k].data[0] for k in tk])[:25])
            root = next(iter(st.nodes.values())).data
            if pi % 4 == 0:         # agent trajectories of every 4th cycle keep the fixture small
                out[f"{name}_p{pi}_pos"] = root[1][:, ::10].astype(np.float32)
                out[f"{name}_p{pi}_cov"] = root[2][:, ::10].astype(np.float32)
        out[name + "_traj_xs"] = np.array(xs)
        print(name, "plans", len(rec), "branch ids", sorted(set(out[name + "_scen_keys"])), "agents", out[name + "_n_agents"].min(),
              out[name + "_n_agents"].max())
    np.savez_compressed(os.path.join(GOLD, "demo_runs.npz"), **out)


def gen_demo_traces(n_plans=60, variant=None, fname="demo_traces.npz"):
    """G13: per-iteration traces of the reference's tree-iLQR over its whole closed loop on the four recorded scenes (the runs of
    demo_runs / demo_branch_runs): for every planning cycle, every candidate scenario tree and both fits (warm start, full) the
    rows {mu the backward pass ran with, J_opt when the line search starts, line-search outcome (1 accepted, -1 rejected, -2
    LinAlgError)} of iLQR.fit's loop (planners/ilqr/solver.py:133-158), recorded by wrapping _backward_pass /
    _backtrack_line_search.  Each candidate is then solved twice more from inputs moved by their rounding resolution (the ego state
    by a relative 1e-13; the predicted means by +-1 float32 ulp): `split` = the first iteration at which one of those runs leaves
    the unperturbed trace (another line-search outcome or mu, or J off by more than 1e-4 relative) = how far the reference itself is reproducible on
    that cost tree.  The GPU tests compare this planner's own traces (mind_last_ilqr_trace) with these, iteration by iteration."""
    import copy
    import importlib
    import json
    import tempfile
    import types
    from mind_amd.scene_io import DEMO_SCENES
    rh.install()
    os.chdir(rh.REF_ROOT)
    vis = types.ModuleType("common.visualization")
    for fn in ("draw_map", "draw_agent", "draw_scen_trees", "reset_ax", "draw_traj_trees", "draw_traj"):
        setattr(vis, fn, None)
    sys.modules["common.visualization"] = vis
    Simulator = importlib.import_module("simulator").Simulator
    tmp = tempfile.mkdtemp()
    ck = os.path.join(tmp, "formula.tar")
    torch.save({"state_dict": formula_state_dict(as_torch=True, variant=variant)}, ck)
    sol = importlib.import_module("planners.ilqr.solver")
    plm = importlib.import_module("planners.mind.planner")
    agent_mod = importlib.import_module("agent")
    iLQR = sol.iLQR
    o_bp, o_ls, o_fit, o_gtt, o_plan = iLQR._backward_pass, iLQR._backtrack_line_search, iLQR.fit, plm.MINDPlanner.get_traj_tree, agent_mod.MINDAgent.plan
    cur = {"fits": None}

    def bp(self):
        if cur["fits"] is not None:
            cur["fits"][-1].append([float(self._mu), float(self.J_opt), -2.0])      # stays -2 when the pass raises LinAlgError
        return o_bp(self)

    def ls(self, alphas):
        acc, conv = o_ls(self, alphas)
        if cur["fits"] is not None:
            cur["fits"][-1][-1][2] = 1.0 if acc else -1.0
        return acc, conv

    def fit(self, *a, **k):
        if cur["fits"] is not None:
            cur["fits"].append([])
        return o_fit(self, *a, **k)

    def first_split(base, other):
        for i in range(max(len(base), len(other))):
            if i >= len(base) or i >= len(other):
                return i
            b, o = base[i], other[i]
            if b[2] != o[2] or abs(b[1] - o[1]) > 1e-4 * abs(b[1]) or abs(b[0] - o[0]) > 1e-9 * abs(b[0]):
                return i
        return max(len(base), len(other))

    plans = []

    def get_traj_tree(self, scen_tree, lcl_smp):
        cur["fits"] = base = []
        res = o_gtt(self, scen_tree, lcl_smp)
        split = [len(base[0]), len(base[1])]
        state0 = self.state
        for seed in range(2):
            rng = np.random.default_rng(seed)
            st2 = scen_tree
            if seed == 0:
                self.state = np.asarray(state0, np.float64) * (1.0 + 1e-13 * rng.standard_normal(np.shape(state0)))
            else:
                st2 = copy.deepcopy(scen_tree)
                for k in st2.nodes:
                    d = st2.nodes[k].data
                    m = np.asarray(d[1])
                    up = rng.random(m.shape) < 0.5
                    d[1] = np.where(up, np.nextafter(m, np.asarray(np.inf, m.dtype)), np.nextafter(m, np.asarray(-np.inf, m.dtype))).astype(m.dtype)
            cur["fits"] = pert = []
            o_gtt(self, st2, lcl_smp)
            self.state = state0
            for ph in range(2):
                split[ph] = min(split[ph], first_split(base[ph], pert[ph]))
        cur["fits"] = None
        plans[-1].append((base, split))
        return res

    def plan(self):
        plans.append([])
        return o_plan(self)

    iLQR._backward_pass, iLQR._backtrack_line_search, iLQR.fit = bp, ls, fit
    plm.MINDPlanner.get_traj_tree = get_traj_tree
    agent_mod.MINDAgent.plan = plan
    out = {}
    try:
        for name in DEMO_SCENES:
            cfg = json.load(open(os.path.join(rh.REF_ROOT, "configs", name + ".json")))
            pcfg = json.load(open(os.path.join(rh.REF_ROOT, cfg["cl_agents"][0]["planner_config"])))
            pcfg.update(use_cuda=False, ckpt_path=ck)
            pp = os.path.join(tmp, name + "_planner.json")
            json.dump(pcfg, open(pp, "w"))
            cfg["cl_agents"][0]["planner_config"] = pp
            cfg.update(render=False, output_dir=tmp)
            cp = os.path.join(tmp, name + ".json")
            json.dump(cfg, open(cp, "w"))
            del plans[:]
            sim = Simulator(cp)
            sim.init_sim()
            sim.sim_horizon = 201 + 5 * (n_plans - 1)
            sim.run_sim()
            assert len(plans) == n_plans, len(plans)
            rows, index, split = [], [], []
            for pi, trees in enumerate(plans):
                for ti, (fits, sp) in enumerate(trees):
                    for ph in range(2):
                        index.append((pi, ti, ph, len(rows), len(fits[ph])))
                        split.append(sp[ph])
                        rows.extend(fits[ph])
            out[name + "_trace_rows"] = np.array(rows, np.float64).reshape(-1, 3)
            out[name + "_trace_index"] = np.array(index, np.int32)
            out[name + "_trace_split"] = np.array(split, np.int32)
            nsp = sum(1 for (i, s_) in zip(index, split) if s_ < i[4])
            print(name, "fits", len(index), "iterations", len(rows), "fits that split under rounding noise", nsp)
    finally:
        iLQR._backward_pass, iLQR._backtrack_line_search, iLQR.fit = o_bp, o_ls, o_fit
        plm.MINDPlanner.get_traj_tree = o_gtt
        agent_mod.MINDAgent.plan = o_plan
    np.savez_compressed(os.path.join(GOLD, fname), **out)


def gen_demo_branch_traces(n_plans=60):
    gen_demo_traces(n_plans, variant="branching", fname="demo_branch_traces.npz")


SECTIONS = {"demo_traces": gen_demo_traces, "demo_branch_traces": gen_demo_branch_traces, "demo_runs": gen_demo_runs, "demo_plans": gen_demo_plans, "demo_branch": gen_demo_branch, "demo_branch_runs": gen_demo_branch_runs, "scenes": gen_scenes, "predictor": gen_predictor, "ilqr": gen_ilqr, "potential": gen_potential, "aime": gen_aime, "plan": gen_plan}

if __name__ == "__main__":
    torch.manual_seed(0)
    which = sys.argv[1:] or list(SECTIONS)
    for s in which:
        SECTIONS[s]()
