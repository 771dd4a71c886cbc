"""Build-container diagnostic (needs /root/reference): is the REFERENCE's own contingency planner ill-conditioned at the
cycles where the teacher-forced comparison disagrees?  Runs the reference's closed loop on a recorded scene up to a given
planning cycle, then re-solves that cycle's trajectory tree with the ego state perturbed by a relative 1e-13 and with the
C oracle on the same inputs.
usage: python tests/diag/ref_conditioning_probe.py demo_2 43"""
import importlib
import json
import os
import sys
import tempfile
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from mind_amd.weights import formula_state_dict
from oracle import ilqr as oi
from oracle import ref_harness as rh

name, cycle = sys.argv[1], int(sys.argv[2])
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
cfg = json.load(open(os.path.join(rh.REF_ROOT, "configs", name + ".json")))
pcfg = json.load(open(os.path.join(rh.REF_ROOT, cfg["cl_agents"][0]["planner_config"])))
pcfg.update(use_cuda=False, ckpt_path=ck)
pp = os.path.join(tmp, "p.json"); json.dump(pcfg, open(pp, "w"))
cfg["cl_agents"][0]["planner_config"] = pp
cfg.update(render=False, output_dir=tmp)
cp = os.path.join(tmp, "c.json"); json.dump(cfg, open(cp, "w"))
sim = Simulator(cp); sim.init_sim()
cap = {}
planner_mod = importlib.import_module("planners.mind.planner")
orig = planner_mod.MINDPlanner.get_traj_tree
count = [0]

def spy(self, scen_tree, lcl_smp):
    r = orig(self, scen_tree, lcl_smp)
    if count[0] == cycle and "tree" not in cap:
        cap.update(tree=scen_tree, lcl=lcl_smp, planner=self, state=np.array(self.state), ctrl=np.array(self.ctrl), out=r)
    return r

planner_mod.MINDPlanner.get_traj_tree = spy
orig_plan = agent_mod.MINDAgent.plan

def plan(self):
    r = orig_plan(self)
    count[0] += 1
    return r

agent_mod.MINDAgent.plan = plan
sim.sim_horizon = 201 + 5 * cycle
sim.run_sim()
pl, st, lcl = cap["planner"], cap["tree"], cap["lcl"]
xs_of = lambda tt: np.array([tt.nodes[k].data[0] for k in tt.nodes if k != -1])
base = xs_of(cap["out"][0])
print(f"{name} cycle {cycle}: scenario tree {list(st.nodes.keys())}, reference ego plan spans x {base[:, 0].min():.1f}..{base[:, 0].max():.1f}")
pl.state, pl.ctrl = cap["state"].copy(), cap["ctrl"].copy()
again = xs_of(orig(pl, st, lcl)[0])
print("reference re-solved with identical inputs:      max |dx| = %.3e" % np.abs(again[:, :2] - base[:, :2]).max())
rng = np.random.default_rng(0)
for rel in (1e-13, 1e-12):
    for trial in range(2):
        pl.state = cap["state"] * (1.0 + rel * rng.standard_normal(4))
        pl.ctrl = cap["ctrl"].copy()
        pert = xs_of(orig(pl, st, lcl)[0])
        print("reference re-solved, ego state moved by %.0e rel: max |dx| = %.3e" % (rel, np.abs(pert[:, :2] - base[:, :2]).max()))
pl.state, pl.ctrl = cap["state"].copy(), cap["ctrl"].copy()
nodes = [(k, n.parent_key, n.data) for k, n in st.nodes.items()]
flat, w, f = oi.contingency(oi.default_cfg(w_vel=pl.traj_tree_opt.config.w_opt_cfg["w_des_state"][2, 2]), nodes, cap["state"], cap["ctrl"],
                            pl.gt_tgt_lane, lcl.target_velocity)
print("C oracle on the reference's inputs:             max |dx| = %.3e (iterations warm %d full %d)" % (
    np.abs(f["xs"][:len(base), :2] - base[:, :2]).max(), w["iterations"], f["iterations"]))
