"""Worker for tests/test_dropin.py (build container only): the reference's Simulator, with its own agent.py / loader.py
unchanged, run on a recorded demo scene to its first planning cycles -- once on the reference's planners package, once on
this package's mirror installed by mind_amd.dropin.  Both get the same CPU test doubles where a GPU / a checkpoint would
be needed (scripted FakeNet predictor; this package's planner additionally takes the C oracle in place of the HIP
tree-iLQR), so the comparison is about the CALL SURFACE: what agent.py hands over and gets back.
usage: python dropin_sim_worker.py ref|mine demo_3 out.pkl"""
import importlib
import json
import os
import pickle
import sys
import tempfile
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

which, name, out_path = sys.argv[1], sys.argv[2], sys.argv[3]
from fake_net import FakeNet
from oracle import ref_harness as rh

rh.install()
os.chdir(rh.REF_ROOT)
vis = types.ModuleType("common.visualization")
for fn in ("draw_map", "draw_agent", "draw_scen_trees", "reset_ax", "draw_traj_trees", "draw_traj"):
    setattr(vis, fn, None)
sys.modules["common.visualization"] = vis

if which == "mine":
    import mind_amd.dropin
    mind_amd.dropin.install()
    from dist_worker import oracle_solver
    import planners.mind.planner as P
    from planners.mind.trajectory_tree import TrajectoryTreeOptimizer
    assert P.__name__.startswith("mind_amd.")

    def init_device(self):
        self.device = torch.device("cpu")

    def init_network(self):
        self.network = FakeNet()

    def init_traj_tree_opt(self):
        cfg = importlib.import_module(self.planner_cfg["planning_config"]).TrajTreeCfg()
        self.traj_tree_opt = TrajectoryTreeOptimizer(cfg, None)
        self.traj_tree_opt.solver = oracle_solver
    P.MINDPlanner.init_device, P.MINDPlanner.init_network, P.MINDPlanner.init_traj_tree_opt = init_device, init_network, init_traj_tree_opt
else:
    import planners.mind.planner as P
    assert not P.__name__.startswith("mind_amd.") and P.__file__.startswith(rh.REF_ROOT)

    def init_device(self):
        self.device = torch.device("cpu")

    def init_network(self):
        self.network = FakeNet()
    P.MINDPlanner.init_device, P.MINDPlanner.init_network = init_device, init_network

for k in [k for k in sys.modules if k in ("agent", "loader", "simulator")]:
    del sys.modules[k]
Simulator = importlib.import_module("simulator").Simulator          # the reference's, with its agent.py and loader.py
tmp = tempfile.mkdtemp()
cfg = json.load(open(os.path.join(rh.REF_ROOT, "configs", name + ".json")))
pcfg = json.load(open(os.path.join(rh.REF_ROOT, cfg["cl_agents"][0]["planner_config"])))
pcfg.update(use_cuda=False)
pp = os.path.join(tmp, "p.json"); json.dump(pcfg, open(pp, "w"))
cfg["cl_agents"][0]["planner_config"] = pp
cfg.update(render=False, output_dir=tmp)
cp = os.path.join(tmp, "c.json"); json.dump(cfg, open(cp, "w"))
sim = Simulator(cp)
sim.init_sim()
sim.sim_horizon = 201 + 5 * 2
sim.run_sim()
ego = [a for a in sim.agents if a.id == "AV"][0]
assert type(ego.planner).__module__ == P.__name__
res = []
for f in sim.frames:
    if "scen_tree" in f:
        st, tt = f["scen_tree"][0], f["traj_tree"][0]
        res.append(dict(keys=list(st.nodes.keys()), probs=[float(np.ravel(n.data[0])[0]) for n in st.nodes.values()],
                        pos=[np.asarray(n.data[1]) for n in st.nodes.values()], cov=[np.asarray(n.data[2]) for n in st.nodes.values()],
                        tgt=[np.asarray(n.data[3]) for n in st.nodes.values()],
                        traj_keys=list(tt.nodes.keys()), xs=np.array([np.asarray(n.data[0]) for k, n in tt.nodes.items()]),
                        us=np.array([np.asarray(n.data[1]) for k, n in tt.nodes.items()])))
pickle.dump(dict(res=res, ctrl=np.array(ego.ctrl), state=np.array(ego.state), planner=type(ego.planner).__module__,
                 n_tracked=len(ego.planner.agent_obs)), open(out_path, "wb"))
print("ok", which, len(res), "plans, scenario keys", [r["keys"] for r in res])
