"""GPU-box diagnostic: full plan timing breakdown on synthetic worlds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mind_amd.synth import SynthWorld
from mind_amd.planners.mind.planner import MINDPlanner
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "mind_amd", "planners", "mind", "configs", "synthetic.json")
G = dict(np.load(os.path.join(ROOT, "tests", "golden", "plan.npz")))
for name, wkw in [("p6", dict(n_agents=6, n_lanes=3, n_segs=8, seed=1)), ("p12", dict(n_agents=12, n_lanes=3, n_segs=10, seed=2)),
                  ("demo-like", dict(n_agents=40, n_lanes=5, n_segs=11, seed=3))]:
    w = SynthWorld(**wkw)
    pl = MINDPlanner(CFG)
    for s in range(50):
        pl.update_observation(w.local_semantic_map(round(0.1 * s, 6)))
    lcl = w.local_semantic_map(4.9)
    pl.update_target_lane(np.asarray(w.target_lane[::2], dtype=np.float64))
    pl.update_state_ctrl(lcl.ego_agent.state, np.array([0.0, 0.0]))
    ok, ctrl, (st, tt) = pl.plan(lcl)
    print(name, "ctrl", ctrl, "scen keys", list(st[0].nodes.keys()), "timing", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in pl.timing.items()})
    if name + "_ctrl" in G:
        tk = [k for k in tt[0].nodes.keys() if k != -1]
        xs = np.array([tt[0].nodes[k].data[0] for k in tk])
        print("   vs reference: ctrl diff %.3e xs diff %.3e keys equal %s" % (np.abs(np.asarray(ctrl) - G[name + "_ctrl"]).max(), np.abs(xs - G[name + "_traj_xs"]).max(), list(st[0].nodes.keys()) == list(G[name + "_scen_keys"])))
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); pl.plan(lcl); ts.append(time.perf_counter() - t0)
    print("   plan() wall: min %.2f ms median %.2f ms ; last breakdown %s" % (min(ts) * 1e3, sorted(ts)[2] * 1e3, {k: (round(v * 1e3, 2) if isinstance(v, float) else v) for k, v in pl.timing.items()}))
