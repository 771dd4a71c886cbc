"""GPU parity: HIP tree-iLQR (through the C-ABI) against the C oracle and the reference goldens."""
import os

import numpy as np
import pytest

from mind_amd.synth import scripted_scenario_tree
from oracle import ilqr as oi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = dict(np.load(os.path.join(ROOT, "tests", "golden", "ilqr.npz")))


@pytest.mark.parametrize("kind,a,max_iter", [("straight", 3, 100), ("lead", 4, 100), ("branch3", 6, 100),
                                              ("branch3", 40, 100), ("deep", 3, 4)])
def test_hip_ilqr_matches_reference_golden(kind, a, max_iter, hip_predictor):
    key = f"{kind}_a{a}_it{max_iter}"
    sst = scripted_scenario_tree(kind, a)
    cfg = oi.default_cfg(max_iter=max_iter)
    flat = oi.flatten(sst["nodes"])
    x0 = oi.init_state(sst["state"], sst["ctrl"])
    xs_w, us_w, st_w = hip_predictor.ilqr_solve(cfg, [flat], x0, sst["target_lane"], sst["target_vel"], 0)
    xs_f, us_f, st_f = hip_predictor.ilqr_solve(cfg, [flat], x0, sst["target_lane"], sst["target_vel"], 1, us_init=us_w)
    # tolerance: float64; well below the 1e-3 m the north star asks for
    assert np.abs(xs_w[0] - G[key + "_xs_w"]).max() < 1e-8
    assert np.abs(xs_f[0] - G[key + "_xs_f"]).max() < 1e-7
    assert np.abs(us_f[0] - G[key + "_us_f"]).max() < 1e-7
    assert st_f[0]["mu"] == G[key + "_Jf"][1]                      # identical accept/reject history
    assert abs(st_f[0]["J"] - G[key + "_Jf"][0]) < 1e-8 * max(1.0, abs(st_f[0]["J"]))


@pytest.mark.parametrize("kind,a", [("lead", 5), ("branch3", 12), ("deep", 4), ("branch3", 128)])
def test_hip_ilqr_matches_oracle_and_batches(kind, a, hip_predictor):
    """several trees in one launch give the same result as one by one; compared with the C oracle."""
    sst = scripted_scenario_tree(kind, a, seed=2)
    cfg = oi.default_cfg(max_iter=6)
    flat = oi.flatten(sst["nodes"])
    x0 = oi.init_state(sst["state"], sst["ctrl"])
    ref = oi.solve(cfg, flat, x0, sst["target_lane"], sst["target_vel"], 1)
    sst2 = scripted_scenario_tree("straight", a, seed=2)
    flat2 = oi.flatten(sst2["nodes"])
    xs, us, st = hip_predictor.ilqr_solve(cfg, [flat, flat2, flat], x0, sst["target_lane"], sst["target_vel"], 1)
    assert np.abs(xs[0] - ref["xs"]).max() < 1e-9 and np.array_equal(xs[0], xs[2])
    ref2 = oi.solve(cfg, flat2, x0, sst["target_lane"], sst["target_vel"], 1)
    assert np.abs(xs[1] - ref2["xs"]).max() < 1e-9
    assert st[0]["iterations"] == ref["iterations"] and st[0]["mu"] == ref["mu"]


def test_hip_ilqr_grid_border_and_outside(hip_predictor):
    """start near / outside the 102 m field: exercises the misplaced border windows (Q4) and clamping."""
    sst = scripted_scenario_tree("straight", 3)
    cfg = oi.default_cfg(max_iter=3)
    flat = oi.flatten(sst["nodes"])
    lane = sst["target_lane"] + np.array([49.0, 50.5])      # lane far from the ego: trajectories pulled to the border
    x0 = oi.init_state(sst["state"], sst["ctrl"])
    ref = oi.solve(cfg, flat, x0, lane, sst["target_vel"], 1)
    xs, us, st = hip_predictor.ilqr_solve(cfg, [flat], x0, lane, sst["target_vel"], 1)
    assert np.abs(xs[0] - ref["xs"]).max() < 1e-7


@pytest.mark.parametrize("kind,a", [("lead", 4), ("branch3", 40)])
def test_contingency_in_one_launch_equals_two_solves(kind, a, hip_predictor):
    """mind_ilqr_contingency (warm-start fit + full fit in one persistent launch) == the two separate calls, bit for bit."""
    sst = scripted_scenario_tree(kind, a)
    cfg_w, cfg_f = oi.default_cfg(max_iter=100), oi.default_cfg(max_iter=100)
    cfg_w.w_ego = cfg_w.w_exo = 0.0                          # w_opt_cfg carries no exo weights (unused by the lane-only fit)
    flat = oi.flatten(sst["nodes"])
    flat2 = oi.flatten(scripted_scenario_tree("straight", a, seed=2)["nodes"])
    x0 = oi.init_state(sst["state"], sst["ctrl"])
    xs_w, us_w, st_w = hip_predictor.ilqr_solve(cfg_w, [flat, flat2], x0, sst["target_lane"], sst["target_vel"], 0)
    xs_f, us_f, st_f = hip_predictor.ilqr_solve(cfg_f, [flat, flat2], x0, sst["target_lane"], sst["target_vel"], 1, us_init=us_w)
    xs, us, sw, sf = hip_predictor.ilqr_contingency(cfg_w, cfg_f, [flat, flat2], x0, sst["target_lane"], sst["target_vel"])
    for t in range(2):
        assert np.array_equal(xs[t], xs_f[t]) and np.array_equal(us[t], us_f[t])
        assert sw[t] == st_w[t] and sf[t] == st_f[t]


def test_kernel_and_oracle_trigonometry_are_the_same_bits(hip_predictor):
    """mind_trig.h compiled for the device (k_ilqr) and for the host (oracle/ilqr_ref.c): sin, cos, tan and the cosine that comes with
    the tangent of the same arguments agree to the bit -- arguments inside pi/4 (the kernel skips the reduction when a whole wave is
    there), outside, mixed within one wave, huge, and the special values."""
    import ctypes as C
    rt, olib = hip_predictor, oi.lib()
    for f in (olib.oracle_sincos, olib.oracle_tan_cos):
        f.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        f.restype = None
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-0.78, 0.78, 640),                   # ten waves that take the short path
                        rng.uniform(-3.2, 3.2, 640), rng.uniform(-100, 100, 640), rng.uniform(-1e5, 1e5, 640),
                        np.where(rng.random(640) < 0.9, rng.uniform(-0.5, 0.5, 640), rng.uniform(-50, 50, 640)),      # mixed waves
                        [0.0, -0.0, 1e-300, np.pi / 4, -np.pi / 4, np.pi / 2, np.pi, 1e15, np.inf, np.nan]])
    out = np.zeros((len(x), 4))
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    rc = rt.lib.mind_debug_trig(rt.ctx, dp(x), len(x), dp(out))
    assert rc == 0
    ref = np.zeros_like(out)
    a, b = C.c_double(), C.c_double()
    for i, v in enumerate(x):
        olib.oracle_sincos(float(v), C.byref(a), C.byref(b)); ref[i, 0], ref[i, 1] = a.value, b.value
        olib.oracle_tan_cos(float(v), C.byref(a), C.byref(b)); ref[i, 2], ref[i, 3] = a.value, b.value
    assert np.array_equal(out.view(np.uint64), ref.view(np.uint64)) or np.array_equal(out[np.isfinite(ref)], ref[np.isfinite(ref)]) \
        and np.array_equal(np.isnan(out), np.isnan(ref))
    assert np.array_equal(out[:, 1], out[:, 3]) or np.array_equal(np.isnan(out[:, 1]), np.isnan(out[:, 3]))


def test_wide_cost_tree_matches_oracle(hip_predictor):
    """The biggest scenario tree of the scripted 6-ary depth-4 AIME tree (16 agents): hundreds of trajectory nodes,
    dozens of chain segments per level (several rounds of waves per level).  Bit-identical to the C oracle after three
    iterations with and without the exo-agent terms: kernel and oracle run the same float64 operations in the same order,
    sin / cos / tan included (mind_amd/csrc/mind_trig.h; with the device library's functions in the kernel and glibc's in
    the oracle this tree differed by <= 6e-14)."""
    from test_aime_host import _full_tree_run
    g, trees = _full_tree_run(True)
    st = max(trees, key=lambda t: len(t.nodes))
    nodes = [(k, n.parent_key, n.data) for k, n in st.nodes.items()]
    flat = oi.flatten(nodes)
    assert len(flat["parent"]) > 300
    lane = np.asarray(g.target_lane[::2], np.float64)
    d0 = nodes[0][2][1][0, 0]                                   # ego position at the first step
    state = np.array([float(d0[0]), float(d0[1]), 4.0, 0.0])
    x0 = oi.init_state(state, np.array([0.0, 0.0]))
    cfg = oi.default_cfg(max_iter=3)
    for use_exo in (0, 1):
        ref = oi.solve(cfg, flat, x0, lane, 4.0, use_exo)
        xs, us, stt = hip_predictor.ilqr_solve(cfg, [flat], x0, lane, 4.0, use_exo)
        assert np.array_equal(xs[0], ref["xs"]) and np.array_equal(us[0], ref["us"])
        assert stt[0]["iterations"] == ref["iterations"] and stt[0]["mu"] == ref["mu"] and stt[0]["J"] == ref["J"]
    cfg1 = oi.default_cfg(max_iter=1)
    ref = oi.solve(cfg1, flat, x0, lane, 4.0, 0)
    xs, us, _ = hip_predictor.ilqr_solve(cfg1, [flat], x0, lane, 4.0, 0)
    assert np.array_equal(xs[0], ref["xs"]) and np.array_equal(us[0], ref["us"])


def test_wide_trees_on_several_workgroups_are_bit_identical(hip_predictor):
    """k_ilqr<.., true>: from 192 nodes on a cost tree is shared by several workgroups (segments, cost chunks and node blocks
    dealt over all their waves, phase boundaries = barriers over the workgroups).  Same arithmetic per item: the result --
    states, controls, iteration count, LM state, J -- must be bit-identical for 1, 2, 8 and 16 workgroups per tree, for the
    contingency sequence (warm-start fit + full fit in one launch), and with small and wide trees mixed in one launch."""
    from test_aime_host import _full_tree_run
    g, trees = _full_tree_run(True)
    by_size = sorted(trees, key=lambda t: len(t.nodes))
    flats = [oi.flatten([(k, n.parent_key, n.data) for k, n in t.nodes.items()]) for t in (by_size[-1], by_size[0], by_size[-2])]
    assert len(flats[0]["parent"]) > 300
    lane = np.asarray(g.target_lane[::2], np.float64)
    d0 = next(iter(by_size[-1].nodes.values())).data[1][0, 0]
    x0 = oi.init_state(np.array([float(d0[0]), float(d0[1]), 4.0, 0.0]), np.array([0.0, 0.0]))
    cfg_w, cfg_f = oi.default_cfg(max_iter=6), oi.default_cfg(max_iter=6)
    res = {}
    try:
        for G in (1, 2, 8, 16):
            hip_predictor.set_tuning("ilqr_wgs", G)
            res[G] = hip_predictor.ilqr_contingency(cfg_w, cfg_f, flats, x0, lane, 4.0)
    finally:
        hip_predictor.set_tuning("ilqr_wgs", 16)
    xs1, us1, sw1, sf1 = res[1]
    for G in (2, 8, 16):
        xs, us, sw, sf = res[G]
        for t in range(len(flats)):
            assert np.array_equal(xs[t], xs1[t]) and np.array_equal(us[t], us1[t]), (G, t)
        assert sw == sw1 and sf == sf1, G
    assert sf1[0]["iterations"] >= 2


def test_singular_q_uu_path_on_one_and_on_several_workgroups(hip_predictor):
    """Quirk Q9 (solver.py:147-151): a LinAlgError of the 2x2 solve burns an iteration without raising mu.  With control weights
    -dt^2 / 2 the leaf's Q_uu = 2 w p + dt^2 (0 + mu) is EXACTLY zero at mu = 1 (node probability 1), so every pass of the fit
    takes the singular-retry path until max_iter.  The kernel must follow the oracle through it -- and the several-workgroup
    variant (whose singular-slot word lives in global memory and is cleared by workgroup 0 at the top of the pass after next: the
    retry path needs its own barrier for that) must equal the one-workgroup kernel."""
    sst = scripted_scenario_tree("straight", 4)
    flat = oi.flatten(sst["nodes"])
    assert set(flat["prob"].tolist()) == {1.0}
    x0 = oi.init_state(sst["state"], sst["ctrl"])
    cfg = oi.default_cfg(max_iter=9)
    w = -(0.2 * 0.2) / 2
    cfg.w_ctrl[:] = [w, w]
    ref = oi.solve(cfg, flat, x0, sst["target_lane"], sst["target_vel"], 1)
    assert ref["iterations"] == 9 and ref["mu"] == 1.0                      # the oracle never leaves the retry path
    one = hip_predictor.ilqr_solve(cfg, [flat], x0, sst["target_lane"], sst["target_vel"], 1)
    assert one[2][0]["iterations"] == 9 and one[2][0]["mu"] == 1.0
    assert np.array_equal(one[0][0], ref["xs"]) and np.array_equal(one[1][0], ref["us"])
    try:
        hip_predictor.set_tuning("ilqr_multi_min", 8)
        for G in (2, 4, 8):
            hip_predictor.set_tuning("ilqr_wgs", G)
            for rep in range(3):
                got = hip_predictor.ilqr_solve(cfg, [flat, flat], x0, sst["target_lane"], sst["target_vel"], 1)
                assert hip_predictor.ilqr_stats()[2] == G
                for t in range(2):
                    assert np.array_equal(got[0][t], one[0][0]) and np.array_equal(got[1][t], one[1][0]) and got[2][t] == one[2][0], G
    finally:
        hip_predictor.set_tuning("ilqr_multi_min", 192)
        hip_predictor.set_tuning("ilqr_wgs", 16)


def test_wide_tree_launch_that_is_not_resident_falls_back_to_one_workgroup(hip_predictor):
    """Several workgroups per tree need every workgroup of the launch on the device at once; when another context holds CUs
    (several planners on one GPU) the barrier would never complete.  The waiting workgroups raise the abort word (~1 s) and the
    call is solved again by the one-workgroup-per-tree kernel: same results, no error.  Reproduced by withholding the last
    workgroups of the launch (mind_set_tuning("ilqr_test_starve"))."""
    sst = scripted_scenario_tree("branch3", 6)
    flat = oi.flatten(sst["nodes"])
    x0 = oi.init_state(sst["state"], sst["ctrl"])
    cfg = oi.default_cfg(max_iter=5)
    ref = hip_predictor.ilqr_solve(cfg, [flat], x0, sst["target_lane"], sst["target_vel"], 1)
    try:
        hip_predictor.set_tuning("ilqr_multi_min", 8)
        hip_predictor.set_tuning("ilqr_wgs", 4)
        ok = hip_predictor.ilqr_solve(cfg, [flat], x0, sst["target_lane"], sst["target_vel"], 1)
        assert hip_predictor.ilqr_stats()[2] == 4                       # really on four workgroups
        hip_predictor.set_tuning("ilqr_test_starve", 1)
        got = hip_predictor.ilqr_solve(cfg, [flat], x0, sst["target_lane"], sst["target_vel"], 1)
        assert hip_predictor.ilqr_stats()[2] == 1                       # aborted, solved again on one workgroup
    finally:
        hip_predictor.set_tuning("ilqr_test_starve", 0)
        hip_predictor.set_tuning("ilqr_multi_min", 192)
        hip_predictor.set_tuning("ilqr_wgs", 16)
    for r in (ok, got):
        assert np.array_equal(r[0][0], ref[0][0]) and np.array_equal(r[1][0], ref[1][0]) and r[2] == ref[2]


@pytest.mark.parametrize("n_agents", [1, 2])
def test_tiny_trees_and_ego_only(n_agents, hip_predictor):
    """one- and two-node cost trees, a scene with the ego alone (no exo term at all)."""
    sst = scripted_scenario_tree("straight", n_agents)
    cfg = oi.default_cfg(max_iter=8)
    x0 = oi.init_state(sst["state"], sst["ctrl"])
    for keep in (1, 3):
        key, parent, data = sst["nodes"][0]
        short = [(key, parent, [data[0], data[1][:, :keep], data[2][:, :keep], data[3]])]
        flat = oi.flatten(short)
        assert len(flat["parent"]) == (keep + 1) // 2
        ref = oi.solve(cfg, flat, x0, sst["target_lane"], sst["target_vel"], 1)
        xs, us, stt = hip_predictor.ilqr_solve(cfg, [flat], x0, sst["target_lane"], sst["target_vel"], 1)
        assert np.array_equal(xs[0], ref["xs"]) and np.array_equal(us[0], ref["us"]) and stt[0]["iterations"] == ref["iterations"]


def test_full_size_tree_properties(hip_predictor):
    """Size-independent properties on the biggest cost tree of the full 6-ary AIME tree (hundreds of trajectory nodes), run
    to the solver's own stop: (1) every accepted step lowers the cost, so the result is never worse than the plain
    rollout of the initial controls; (2) determinism: the same call twice is bit-identical; (3) fixed point: restarting
    from the returned controls cannot be improved upon at the first regularisation level, so it returns those controls'
    own rollout -- the same trajectory; (4) the tree structure of the solution: node states obey the bicycle model from
    their parent's state under their own control (trajectory_tree.py:168-175), to rounding."""
    from test_aime_host import _full_tree_run
    g, trees = _full_tree_run(True)
    st = max(trees, key=lambda t: len(t.nodes))
    nodes = [(k, n.parent_key, n.data) for k, n in st.nodes.items()]
    flat = oi.flatten(nodes)
    M = len(flat["parent"])
    assert M > 300
    lane = np.asarray(g.target_lane[::2], np.float64)
    d0 = nodes[0][2][1][0, 0]
    x0 = oi.init_state(np.array([float(d0[0]), float(d0[1]), 4.0, 0.0]), np.array([0.0, 0.0]))
    cfg = oi.default_cfg(max_iter=100)
    cfg0 = oi.default_cfg(max_iter=0)
    _, _, st0 = hip_predictor.ilqr_solve(oi.default_cfg(max_iter=1), [flat], x0, lane, 4.0, 1)   # J = cost of the zero-control rollout
    xs, us, st1 = hip_predictor.ilqr_solve(cfg, [flat], x0, lane, 4.0, 1)
    xs2, us2, st2 = hip_predictor.ilqr_solve(cfg, [flat], x0, lane, 4.0, 1)
    assert np.array_equal(xs[0], xs2[0]) and np.array_equal(us[0], us2[0]) and st1[0]["J"] == st2[0]["J"]
    assert st1[0]["iterations"] >= 1 and np.all(np.isfinite(xs[0])) and np.all(np.isfinite(us[0]))
    assert st1[0]["J"] <= st0[0]["J"]                  # J = cost before the last accepted step (Q19): already below the start
    xs3, us3, st3 = hip_predictor.ilqr_solve(cfg0, [flat], x0, lane, 4.0, 1, us_init=[us[0]])
    assert np.array_equal(us3[0], us[0])                                                  # max_iter 0: the rollout of us
    assert np.abs(xs3[0] - xs[0]).max() < 1e-9
    # bicycle model along every tree edge (node 0 hangs off x0)
    dt, wb = cfg.dt, cfg.wheelbase
    par = flat["parent"]
    prev = np.where(par[:, None] >= 0, xs[0][np.maximum(par, 0)], x0[None, :])
    u = us[0]
    nxt = np.stack([prev[:, 0] + prev[:, 2] * np.cos(prev[:, 3]) * dt, prev[:, 1] + prev[:, 2] * np.sin(prev[:, 3]) * dt,
                    prev[:, 2] + prev[:, 4] * dt, prev[:, 3] + prev[:, 2] / wb * np.tan(prev[:, 5]) * dt,
                    prev[:, 4] + u[:, 0] * dt, prev[:, 5] + u[:, 1] * dt], axis=1)
    assert np.abs(nxt - xs[0]).max() < 1e-9


@pytest.mark.parametrize("kind,a,max_iter", [("lead", 4, 100), ("branch3", 40, 100), ("deep", 3, 4), ("branch3", 128, 30)])
def test_iteration_trace_equals_the_oracles(kind, a, max_iter, hip_predictor):
    """mind_last_ilqr_trace: per reference iteration the Levenberg-Marquardt value, the cost of the nominal trajectory, the accepted
    step index (-1 rejected, -2 singular) and the accepted candidate's cost -- the rows of the C oracle's loop
    (oracle/ilqr_ref.c, after planners/ilqr/solver.py:133-158), for both fits of mind_ilqr_contingency and for a plain solve, on
    one and on several workgroups."""
    sst = scripted_scenario_tree(kind, a)
    cfg_w, cfg_f = oi.default_cfg(max_iter=max_iter), oi.default_cfg(max_iter=max_iter)
    cfg_w.w_ego = cfg_w.w_exo = 0.0
    flat = oi.flatten(sst["nodes"])
    flat2 = oi.flatten(scripted_scenario_tree("straight", a, seed=2)["nodes"])
    x0 = oi.init_state(sst["state"], sst["ctrl"])
    refs = []
    for f in (flat, flat2):
        w = oi.solve(cfg_w, f, x0, sst["target_lane"], sst["target_vel"], 0, trace=True)
        full = oi.solve(cfg_f, f, x0, sst["target_lane"], sst["target_vel"], 1, us_init=w["us"], trace=True)
        refs.append((w, full))
        assert len(w["trace"]) == w["iterations"] and np.array_equal(w["trace"][:, 1], w["J_trace"][:w["iterations"]])

    def same(got, ref, what):
        # the decisions exactly (Levenberg-Marquardt value, accepted step index), the costs to the last bits (the kernel's node costs
        # may differ from the oracle's in the last ulp: another FMA contraction, never another decision)
        assert got.shape == ref.shape, (what, got.shape, ref.shape)
        assert np.array_equal(got[:, [0, 2]], ref[:, [0, 2]]), (what, got[:, [0, 2]], ref[:, [0, 2]])
        assert np.allclose(got[:, [1, 3]], ref[:, [1, 3]], rtol=1e-13, atol=0.0), (what, np.abs(got[:, [1, 3]] / ref[:, [1, 3]] - 1).max())

    def check():
        hip_predictor.ilqr_contingency(cfg_w, cfg_f, [flat, flat2], x0, sst["target_lane"], sst["target_vel"])
        for t in range(2):
            for ph in range(2):
                got, ref = hip_predictor.ilqr_trace(t, ph), refs[t][ph]["trace"]
                same(got, ref, (t, ph))
        hip_predictor.ilqr_solve(cfg_f, [flat2], x0, sst["target_lane"], sst["target_vel"], 1, us_init=[refs[1][0]["us"]])
        same(hip_predictor.ilqr_trace(0, 0), refs[1][1]["trace"], "plain solve")
        with pytest.raises(Exception):
            hip_predictor.ilqr_trace(0, 1)                  # a plain solve has one fit
        with pytest.raises(Exception):
            hip_predictor.ilqr_trace(1, 0)

    check()
    try:
        hip_predictor.set_tuning("ilqr_multi_min", 8)
        hip_predictor.set_tuning("ilqr_wgs", 4)
        check()
    finally:
        hip_predictor.set_tuning("ilqr_multi_min", 192)
        hip_predictor.set_tuning("ilqr_wgs", 16)


def test_contingency_in_two_halves_equals_the_one_call(hip_predictor):
    """mind_ilqr_contingency_begin / mind_ilqr_finish (the planner builds its scenario trees' Python objects between the two): the same
    results as the blocking call, bit for bit; one call pending per context, and nothing to finish without a begin."""
    from mind_amd import _lib
    from mind_amd.predictor import IlqrCall
    sst = scripted_scenario_tree("branch3", 40)
    cfg_w, cfg_f = oi.default_cfg(max_iter=100), oi.default_cfg(max_iter=100)
    cfg_w.w_ego = cfg_w.w_exo = 0.0
    flats = [oi.flatten(sst["nodes"]), oi.flatten(scripted_scenario_tree("straight", 40, seed=2)["nodes"])]
    x0 = oi.init_state(sst["state"], sst["ctrl"])
    ref = hip_predictor.ilqr_contingency(cfg_w, cfg_f, flats, x0, sst["target_lane"], sst["target_vel"])
    call = IlqrCall(hip_predictor.lib, cfg_w, flats, x0, sst["target_lane"], sst["target_vel"], cfg_full=cfg_f)
    call.begin(hip_predictor)
    with pytest.raises(_lib.MindError):                       # the context holds a begun call: no other tree-iLQR call in between
        hip_predictor.ilqr_contingency(cfg_w, cfg_f, flats, x0, sst["target_lane"], sst["target_vel"])
    xs, us, sw, sf = call.wait().finish()
    for t in range(2):
        assert np.array_equal(xs[t], ref[0][t]) and np.array_equal(us[t], ref[1][t]) and sw[t] == ref[2][t] and sf[t] == ref[3][t]
    assert hip_predictor.lib.mind_ilqr_finish(hip_predictor.ctx) == _lib.MIND_ESTATE
    again = hip_predictor.ilqr_contingency(cfg_w, cfg_f, flats, x0, sst["target_lane"], sst["target_vel"])      # the context is free again
    assert np.array_equal(again[0][0], ref[0][0])


def _contingency_cases():
    cases = []
    for name, depth in (("lead", 4), ("branch3", 6), ("deep", 5)):
        sst = scripted_scenario_tree(name, depth)
        cases.append((name, sst, oi.flatten(sst["nodes"]), oi.init_state(sst["state"], sst["ctrl"])))
    return cases


def test_levenberg_marquardt_slots_on_follower_workgroups_equal_one_workgroup(hip_predictor):
    """k_ilqr<GEN, 2>: the master workgroup of a tree evaluates slot 0 of every pass, follower workgroups the slots the LM schedule reaches after
    1 .. n - 1 rejections (solver.py:133-158 is sequential: the first slot with an improving step is what it would have reached).  Every
    slot count must return the single workgroup's bits -- trajectories, statistics and the per-iteration traces of both fits -- and so
    must a launch whose followers never start (the master then keeps its slots), with and without the derivative speculator (il_speculate:
    one more workgroup per tree differentiates a pass's first candidate beside the master's pricing of the candidates; "il_spec" counts the
    passes it was asked in and the results the master adopted by swapping derivative sets)."""
    cw, cf = oi.default_cfg(), oi.default_cfg()
    for name, sst, flat, x0 in _contingency_cases():
        try:
            hip_predictor.set_tuning("ilqr_slots", 1)
            ref = hip_predictor.ilqr_contingency(cw, cf, [flat, flat, flat], x0, sst["target_lane"], sst["target_vel"])
            assert hip_predictor.ilqr_stats()[2] == 1
            ref_tr = [(hip_predictor.ilqr_trace(t, 0), hip_predictor.ilqr_trace(t, 1)) for t in range(3)]
            assert sum(int((tr[:, 2] < 0).sum()) for tr, _ in ref_tr) + sum(int((tr[:, 2] < 0).sum()) for _, tr in ref_tr) > 0, name   # rejections happen
            taken = 0
            for slots, starve, spec in ((2, 0, 0), (3, 0, 1), (4, 0, 0), (4, 0, 1), (8, 0, 1), (12, 0, 0), (12, 0, 1), (10, 1, 1), (10, 0, 1)):
                hip_predictor.set_tuning("ilqr_slots", slots)
                hip_predictor.set_tuning("ilqr_test_starve", starve)
                hip_predictor.set_tuning("ilqr_spec_deriv", spec)
                for rep in range(2):
                    got = hip_predictor.ilqr_contingency(cw, cf, [flat, flat, flat], x0, sst["target_lane"], sst["target_vel"])
                    assert hip_predictor.ilqr_stats()[2] == slots + spec
                    asked, hits = (int(v) for v in hip_predictor.debug_read("il_spec"))
                    assert hits <= asked and (asked > 0) == bool(spec and not starve), (name, slots, starve, spec, asked, hits)
                    taken += hits
                    for t in range(3):
                        assert np.array_equal(got[0][t], ref[0][t]) and np.array_equal(got[1][t], ref[1][t]), (name, slots, starve, spec, t)
                        assert got[2][t] == ref[2][t] and got[3][t] == ref[3][t], (name, slots, starve, spec, t)
                        for ph in (0, 1):
                            assert np.array_equal(hip_predictor.ilqr_trace(t, ph), ref_tr[t][ph]), (name, slots, starve, spec, t, ph)
            # the derivative speculator's set really became the nominal one: accepted first candidates of slot 0 exist in every case
            assert taken > 0, name
        finally:
            hip_predictor.set_tuning("ilqr_test_starve", 0)
            hip_predictor.set_tuning("ilqr_slots", 10)
            hip_predictor.set_tuning("ilqr_spec_deriv", 1)
    # the oracle's result, for good measure (the other tests of this file run with the default slot count)
    name, sst, flat, x0 = _contingency_cases()[0]
    w = oi.solve(cw, flat, x0, sst["target_lane"], sst["target_vel"], 0)
    f = oi.solve(cf, flat, x0, sst["target_lane"], sst["target_vel"], 1, us_init=w["us"])
    got = hip_predictor.ilqr_contingency(cw, cf, [flat], x0, sst["target_lane"], sst["target_vel"])
    assert np.array_equal(got[0][0], f["xs"]) and got[3][0]["iterations"] == f["iterations"]


def test_singular_q_uu_in_a_followers_slot(hip_predictor):
    """Quirk Q9 where a FOLLOWER meets it: with control weights -dt^2 / 2 the leaf's Q_uu = 2 w p + dt^2 mu is exactly zero at mu = 1.  On this
    tree the lane-only fit accepts two steps, then rejects seven in a row while the LM schedule climbs from 2^-6 back to mu = 1 -- the singular
    value, met right behind a rejection (found by tests/diag/gpu_find_follower_singular.py): with 2 / 4 / 8 slots per pass it falls into a
    follower's slot.  Slots behind a singular slot are not used; the fit then sits at mu = 1 in the singular-retry path.  Must equal the oracle
    and the single workgroup, trace row by trace row."""
    sst = scripted_scenario_tree("branch3", 6)
    flat = oi.flatten(sst["nodes"])
    x0 = oi.init_state(sst["state"], sst["ctrl"])
    cfg = oi.default_cfg(max_iter=30)
    cfg.w_ctrl[:] = [-0.2 * 0.2 / 2, -0.2 * 0.2 / 2]
    ref = oi.solve(cfg, flat, x0, sst["target_lane"], sst["target_vel"], 0)
    res = {}
    try:
        for slots in (1, 2, 4, 8, 12):
            hip_predictor.set_tuning("ilqr_slots", slots)
            out = hip_predictor.ilqr_solve(cfg, [flat, flat], x0, sst["target_lane"], sst["target_vel"], 0)
            res[slots] = (out, hip_predictor.ilqr_trace(0, 0), hip_predictor.ilqr_trace(1, 0))
    finally:
        hip_predictor.set_tuning("ilqr_slots", 10)
    one, tr1, _ = res[1]
    picks = tr1[:, 2]
    first = int(np.argmax(picks == -2))
    assert (picks == -2).any() and picks[first - 1] == -1 and (picks[first:] == -2).all(), picks       # singular right behind a rejection, then stuck
    for slots in (2, 4, 8, 12):
        many, tra, trb = res[slots]
        assert np.array_equal(tr1, tra) and np.array_equal(tr1, trb), slots
        for t in range(2):
            assert np.array_equal(one[0][0], many[0][t]) and np.array_equal(one[1][0], many[1][t]) and one[2][0] == many[2][t], (slots, t)
    assert np.array_equal(one[0][0], ref["xs"]) and np.array_equal(one[1][0], ref["us"]) and one[2][0]["iterations"] == ref["iterations"] and one[2][0]["mu"] == ref["mu"]


def test_cost_tree_beyond_the_lds_sums_matches_oracle(hip_predictor):
    """A cost tree of ~2 400 trajectory nodes (a short trunk, 300 branches): more nodes than the LDS staging of the cost sums holds, so the
    nominal cost (numpy's pairwise L.sum(): leaf sums by all threads, folded along the same recursion) and the ten candidate costs (python
    sum(): sequential, walked through LDS tiles) take their big-tree paths.  Bit-identical to the C oracle on one workgroup and on several."""
    rng = np.random.default_rng(11)
    trunk, nb, blen, a = 10, 300, 8, 3
    parent = list(range(-1, trunk - 1))
    prob = [1.0] * trunk
    for b in range(nb):
        for k in range(blen):
            parent.append(trunk - 1 if k == 0 else len(parent) - 1)
            prob.append(1.0 / nb)
    M = len(parent)
    assert M > 2048
    depth = np.zeros(M, int)
    for i in range(1, M):
        depth[i] = depth[parent[i]] + 1
    mean = np.zeros((M, a, 2), np.float32)
    mean[:, 0, 0] = 0.8 * (depth + 1)                                                 # the ego's own prediction: straight ahead at 4 m/s
    mean[:, 1] = np.stack([0.8 * (depth + 1) + 6.0, 0.2 + 0.3 * rng.standard_normal(M)], 1)       # a lead vehicle
    mean[:, 2] = np.stack([0.8 * (depth + 1) - 1.0, 3.5 + 0.2 * rng.standard_normal(M)], 1)       # a neighbour
    cov = (0.4 + 0.05 * depth[:, None] + 0.02 * rng.random((M, a))).astype(np.float32)
    flat = dict(parent=np.asarray(parent, np.int32), prob=np.asarray(prob, np.float32), mean=mean, cov=cov)
    lane = np.stack([np.linspace(-5.0, 60.0, 40), np.zeros(40)], 1)
    x0 = oi.init_state(np.array([0.0, 0.1, 4.0, 0.0]), np.array([0.0, 0.0]))
    cfg = oi.default_cfg(max_iter=4)
    ref = oi.solve(cfg, flat, x0, lane, 4.0, 1, trace=True)
    assert ref["iterations"] >= 2
    try:
        for wgs in (1, 4):
            hip_predictor.set_tuning("ilqr_wgs", wgs)
            xs, us, stt = hip_predictor.ilqr_solve(cfg, [flat], x0, lane, 4.0, 1)
            assert hip_predictor.ilqr_stats()[2] == wgs
            assert np.array_equal(xs[0], ref["xs"]) and np.array_equal(us[0], ref["us"]), wgs
            assert stt[0]["iterations"] == ref["iterations"] and stt[0]["mu"] == ref["mu"] and stt[0]["J"] == ref["J"], wgs
    finally:
        hip_predictor.set_tuning("ilqr_wgs", 16)
