"""GPU-box diagnostic: teacher-forced whole run on a recorded scene, per-cycle differences against tests/golden/demo_runs.npz."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from bench import WORKLOADS, make_closed_loop
scene = sys.argv[1] if len(sys.argv) > 1 else "demo_2"
D = np.load(os.path.join(ROOT, "tests", "golden", "demo_runs.npz"))
pl, sim, w = make_closed_loop(dict(WORKLOADS[scene]), scripted=False)
from oracle import ilqr as oi
rt = pl.network.rt
pl.traj_tree_opt.speculative = False      # this diagnostic captures the arguments of the one-launch contingency call
cap = {}
orig_cont = rt.ilqr_contingency
def capture(*a, **k):
    r = orig_cont(*a, **k)
    cap["args"], cap["out"] = a, r
    return r
rt.ilqr_contingency = capture
state_in, ctrl_in, ctrl_out, xs_all = D[scene + "_state_in"], D[scene + "_ctrl_in"], D[scene + "_ctrl_out"], D[scene + "_traj_xs"]
if os.environ.get("TRACE_FROM"):
    lo = int(os.environ["TRACE_FROM"])
for pi in range(len(state_in)):
    while True:
        will = sim.sim_time >= sim.enable_time and (sim.last_trigger is None or sim.sim_time - sim.last_trigger >= sim.PLAN_STEP)
        if will and pi > 0:
            sim.state, sim.ctrl = state_in[pi].copy(), ctrl_in[pi].copy()
        if will and os.environ.get("TRACE_FROM") and pi >= lo:
            os.environ["MIND_ILQR_TRACE"] = "1"
        if sim.step():
            break
    st, tt = sim.last_result[0][0], sim.last_result[1][0]
    tk = [k for k in tt.nodes.keys() if k != -1]
    xs = np.array([tt.nodes[k].data[0] for k in tk])[:25]
    print("cycle %2d keys %-12s ref %-12s n_trees %d/%d  ctrl %s ref %s  |dctrl| %.2e  |dego| %.2e  ref |xs| max %.1f  state_in v %.3f" % (
        pi, "|".join(st.nodes.keys()), str(D[scene + "_scen_keys"][pi]), len(sim.last_result[0]), int(D[scene + "_n_scen_trees"][pi]),
        np.round(sim.ctrl, 4), np.round(ctrl_out[pi], 4), np.abs(np.asarray(sim.ctrl) - ctrl_out[pi]).max(),
        np.abs(xs[:, :2] - xs_all[pi][:, :2]).max(), np.abs(xs_all[pi][:, 2:]).max(), state_in[pi][2]))
    if np.abs(np.asarray(sim.ctrl) - ctrl_out[pi]).max() > 1e-4:
        cw, cf, flats, x0, lane, tv = cap["args"]
        hx, hu, sw, sf = cap["out"]
        ow = oi.solve(cw, flats[0], x0, lane, tv, 0)
        of = oi.solve(cf, flats[0], x0, lane, tv, 1, us_init=ow["us"])
        moved = []
        for seed in range(4):
            rng = np.random.default_rng(seed)
            f2 = dict(flats[0])
            m = flats[0]["mean"]
            f2["mean"] = np.where(rng.random(m.shape) < 0.5, np.nextafter(m, np.float32(np.inf)), np.nextafter(m, np.float32(-np.inf))).astype(np.float32)
            px, pu, _, _ = orig_cont(cw, cf, [f2], x0, lane, tv)
            moved.append(float(np.abs(px[0][:, :2] - hx[0][:, :2]).max()))
        print("      HIP re-solved with the agent means moved by +-1 float32 ulp (4 draws): ego moves by", np.round(moved, 4))
        print("      oracle on the SAME inputs: |xs_hip - xs_oracle| %.2e  us0 oracle %s hip %s  iterations warm %d/%d full %d/%d  J %.6g/%.6g" % (
            np.abs(hx[0] - of["xs"]).max(), np.round(of["us"][0], 4), np.round(hu[0][0], 4), sw[0]["iterations"], ow["iterations"],
            sf[0]["iterations"], of["iterations"], sf[0]["J"], of["J"]))
