"""The module-level drop-in of INTEGRATION.md A: after mind_amd.dropin.install(), `planners...` IS this package's mirror,
and (build container) the reference's own agent.py reaches this MINDPlanner."""
import importlib
import subprocess
import sys
import os

import pytest

from oracle import ref_harness as rh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code):
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp", timeout=300,
                         env=dict(os.environ, PYTHONPATH=ROOT))
    assert out.returncode == 0, out.stderr[-3000:]
    return out.stdout


def test_install_aliases_every_module():
    out = _run("""
import mind_amd.dropin as d
d.install()
import planners.mind.planner, planners.mind.scenario_tree, planners.mind.trajectory_tree, planners.mind.utils
import planners.mind.networks.network, planners.ilqr.solver, planners.ilqr.cost, planners.ilqr.potential, planners.ilqr.utils
import planners.basic.tree, planners.mind.configs.planning.demo_3, planners.mind.configs.networks.net_cfg
import mind_amd.planners.mind.planner as mine
assert planners.mind.planner is mine and planners.mind.planner.MINDPlanner is mine.MINDPlanner
from importlib import import_module
assert import_module("planners.mind.configs.planning.demo_1").TrajTreeCfg().dt == 0.2
print("ok")
""")
    assert out.strip().endswith("ok")


@pytest.mark.reference
@pytest.mark.skipif(not rh.available(), reason="reference tree not present")
def test_reference_agent_reaches_this_planner():
    """Build container only: the reference's agent.py, imported unchanged after install(), constructs THIS MINDPlanner --
    which refuses to run without a GPU (there is no CPU path)."""
    out = _run("""
import sys, json, tempfile, os
from oracle import ref_harness as rh
rh.install()                                    # av2 / shapely / theano stand-ins + the reference root on sys.path
import mind_amd.dropin as d
d.install()
for k in [k for k in sys.modules if k in ("agent", "loader", "simulator")]:
    del sys.modules[k]
import agent                                    # the reference's agent.py
import mind_amd.planners.mind.planner as mine
assert agent.MINDPlanner is mine.MINDPlanner
cfg = os.path.join(tempfile.mkdtemp(), "p.json")
json.dump({"use_cuda": True, "network_config": "planners.mind.configs.networks.net_cfg", "ckpt_path": "formula:20240121",
           "planning_config": "planners.mind.configs.planning.demo_1"}, open(cfg, "w"))
a = agent.MINDAgent()
try:
    a.init_planner(cfg)
    import torch
    assert torch.cuda.is_available()
    print("constructed", type(a.planner).__module__)
except RuntimeError as e:
    assert "GPU" in str(e), e
    print("refused without a GPU:", e)
print("ok")
""")
    assert out.strip().endswith("ok")


@pytest.mark.reference
@pytest.mark.skipif(not rh.available(), reason="reference tree not present")
def test_reference_simulator_drives_this_planner_like_its_own(tmp_path):
    """Build container only.  The reference's Simulator with its agent.py and loader.py UNCHANGED runs a recorded demo scene
    to its first three planning cycles twice: on the reference's `planners` package and on this package's mirror installed
    by mind_amd.dropin.  Both take the same CPU test doubles where a GPU or a checkpoint would be needed (tests/
    dropin_sim_worker.py), so what is compared is the call surface -- what agent.py hands to the planner and reads back:
    same tracked agents, branch ids, probabilities; agent trajectories within one float32 ulp of the map coordinates; ego
    trajectory, control and the simulator's final ego state to rounding."""
    import pickle
    import numpy as np
    worker = os.path.join(ROOT, "tests", "dropin_sim_worker.py")
    res = {}
    for which in ("ref", "mine"):
        out = os.path.join(tmp_path, which + ".pkl")
        p = subprocess.run([sys.executable, worker, which, "demo_3", out], capture_output=True, text=True, timeout=900, cwd="/tmp")
        assert p.returncode == 0, p.stderr[-3000:]
        res[which] = pickle.load(open(out, "rb"))
    a, b = res["ref"], res["mine"]
    assert a["planner"] == "planners.mind.planner" and b["planner"] == "mind_amd.planners.mind.planner"
    assert a["n_tracked"] == b["n_tracked"] and len(a["res"]) == len(b["res"]) == 3
    for ra, rb in zip(a["res"], b["res"]):
        assert ra["keys"] == rb["keys"] and ra["traj_keys"] == rb["traj_keys"]
        assert np.array_equal(ra["probs"], rb["probs"])
        for x, y in zip(ra["pos"], rb["pos"]):
            assert x.dtype == y.dtype and x.shape == y.shape and np.abs(x - y).max() <= 2 * np.spacing(np.float32(np.abs(x).max()))
        for k in ("cov", "tgt"):
            assert all(np.array_equal(x, y) for x, y in zip(ra[k], rb[k]))
        assert np.abs(ra["xs"] - rb["xs"]).max() < 1e-9 and np.abs(ra["us"] - rb["us"]).max() < 1e-9
    assert np.abs(a["ctrl"] - b["ctrl"]).max() < 1e-9 and np.abs(a["state"] - b["state"]).max() < 1e-9
