"""GPU parity of the planners/ilqr call surface (iLQR.fit over a TreeCost of PotentialField / StatePotential /
StateConstraint / ControlPotential objects, gen_dist_field) -- SURVEY 8(b) secondary surfaces, goldens G4-G6."""
import os

import numpy as np
import pytest

from mind_amd.planners.basic.tree import Node, Tree
from mind_amd.planners.ilqr.cost import TreeCost
from mind_amd.planners.ilqr.dynamics import BicycleDynamics
from mind_amd.planners.ilqr.potential import ControlPotential, PotentialField, StateConstraint, StatePotential
from mind_amd.planners.ilqr.solver import iLQR
from mind_amd.planners.ilqr.utils import gen_dist_field
from mind_amd.synth import scripted_scenario_tree
from oracle import ilqr as oi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_potential_field_matches_reference_golden(hip_predictor):
    """G4: value / gradient / Hessian on a random 32x32 field incl. the 8 border cases and .5 rounding ties
    (values produced by the reference's PotentialField, tests/golden/gen_golden.py potential)."""
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "potential.npz")))
    F, off, res = g["F"], g["off"], float(g["res"])
    H, W = F.shape
    gx = np.linspace(0.0, (W - 1) * res, W) + off[0]
    gy = np.linspace(0.0, (H - 1) * res, H) + off[1]
    xx, yy = np.meshgrid(gx, gy)
    pf = PotentialField(off, res, xx, yy, F)
    for (px, py), want in zip(g["pts"], g["vals"]):
        s = np.array([px, py, 0.0, 0.0, 0.0, 0.0])
        val, grad, hess = pf.get_potential(s), pf.get_gradient(s), pf.get_hessian(s)
        got = np.array([val, grad[0], grad[1], hess[0, 0], hess[1, 1], hess[0, 1]])
        assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max()), (px, py, got, want)
        assert hess[0, 1] == hess[1, 0] and not grad[2:].any() and not hess[2:, 2:].any()


def test_quadratic_potentials(hip_predictor):
    """potential.py:4-59 on the device against the closed forms (diagonal weights)."""
    rng = np.random.default_rng(5)
    Wd = np.diag(rng.uniform(0, 3, 6))
    des = rng.normal(size=6)
    x = rng.normal(size=6) * 3
    sp = StatePotential(Wd, des)
    assert abs(sp.get_potential(x) - (x - des) @ Wd @ (x - des)) < 1e-12
    assert np.abs(sp.get_gradient(x) - 2 * Wd @ (x - des)).max() < 1e-12
    assert np.abs(sp.get_hessian(x) - 2 * Wd).max() == 0
    lb, ub = -np.abs(rng.normal(size=6)), np.abs(rng.normal(size=6))
    sc = StateConstraint(Wd, lb, ub)
    d = np.maximum(x - ub, 0) + np.maximum(lb - x, 0)
    assert abs(sc.get_potential(x) - d @ Wd @ d) < 1e-12
    viol = (x > ub) | (x < lb)
    want_g = np.where(x > ub, 2 * np.diag(Wd) * (x - ub), np.where(x < lb, 2 * np.diag(Wd) * (x - lb), 0.0))
    assert np.abs(sc.get_gradient(x) - want_g).max() < 1e-12
    assert np.abs(np.diag(sc.get_hessian(x)) - np.where(viol, 2 * np.diag(Wd), 0.0)).max() == 0
    Wc = np.diag([5.0, 2.0])
    u = rng.normal(size=2)
    cp = ControlPotential(Wc)
    assert abs(cp.get_potential(u) - u @ Wc @ u) < 1e-12
    assert np.abs(cp.get_gradient(u) - 2 * Wc @ u).max() < 1e-12 and np.abs(cp.get_hessian(u) - 2 * Wc).max() == 0
    with pytest.raises(NotImplementedError):
        StatePotential(np.ones((6, 6)), des).get_potential(x)


def _reference_style_cost_tree(cfg, sst, use_exo):
    """Build the cost tree the way trajectory_tree.py:19-124 does -- one PotentialField + three quadratic
    potentials per trajectory node, root key -1 holding x0 -- from the oracle's materialised fields."""
    flat = oi.flatten(sst["nodes"])
    x0 = oi.init_state(sst["state"], sst["ctrl"])
    fields, gx, gy, off = oi.node_fields(cfg, flat, x0, sst["target_lane"], use_exo)
    xx, yy = np.meshgrid(gx, gy)
    t = Tree()
    t.add_node(Node(-1, None, x0))
    w_des, w_con, w_ctrl = np.diag(list(cfg.w_des_state)), np.diag(list(cfg.w_state_con)), np.diag(list(cfg.w_ctrl))
    for k in range(len(flat["parent"])):
        p = flat["prob"][k]                      # np.float32, as in the reference
        pots = [[PotentialField(off, cfg.grid_res, xx, yy, fields[k]),
                 StatePotential(w_des * p, np.array([0, 0, sst["target_vel"], 0.0, 0.0, 0.0])),
                 StateConstraint(w_con * p, np.array(list(cfg.state_lower)), np.array(list(cfg.state_upper)))],
                [ControlPotential(w_ctrl * p)]]
        t.add_node(Node(k, int(flat["parent"][k]), pots))
    return flat, x0, TreeCost(t, 6, 2)


@pytest.mark.parametrize("kind,a,max_iter", [("lead", 4, 100), ("branch3", 6, 100)])
def test_ilqr_fit_on_tree_cost_matches_reference_golden(kind, a, max_iter, hip_predictor):
    """G6 through the reference's own call sequence: iLQR(dynamics).fit(us_init, TreeCost) warm start then full
    solve.  Bit-identical to the C oracle and to the planner-mode kernel (analytic fields); golden within 1e-7."""
    G = dict(np.load(os.path.join(ROOT, "tests", "golden", "ilqr.npz")))
    key = f"{kind}_a{a}_it{max_iter}"
    sst = scripted_scenario_tree(kind, a)
    cfg = oi.default_cfg(max_iter=max_iter)
    solver = iLQR(BicycleDynamics(cfg.dt, cfg.wheelbase))
    flat, x0, cost_w = _reference_style_cost_tree(cfg, sst, 0)
    xs_w, us_w = solver.fit(np.zeros((len(flat["parent"]), 2)), cost_w, n_iterations=max_iter)
    _, _, cost_f = _reference_style_cost_tree(cfg, sst, 1)
    xs_f, us_f = solver.fit(us_w, cost_f, n_iterations=max_iter)
    assert np.abs(xs_w - G[key + "_xs_w"]).max() < 1e-8
    assert np.abs(xs_f - G[key + "_xs_f"]).max() < 1e-7 and np.abs(us_f - G[key + "_us_f"]).max() < 1e-7
    assert solver._mu == G[key + "_Jf"][1]
    ref = oi.solve(cfg, flat, x0, sst["target_lane"], sst["target_vel"], 1, us_init=us_w)
    assert np.array_equal(xs_f, ref["xs"]) and np.array_equal(us_f, ref["us"]) and solver.iterations == ref["iterations"]
    pxs, pus, _ = hip_predictor.ilqr_solve(cfg, [flat], x0, sst["target_lane"], sst["target_vel"], 1, us_init=[us_w])
    assert np.array_equal(xs_f, pxs[0]) and np.array_equal(us_f, pus[0])


def test_tree_cost_derivatives_match_oracle(hip_predictor):
    """TreeCost.l / l_x / l_u / l_xx / l_uu (cost.py:341-446) at perturbed states, generic and planner mode."""
    sst = scripted_scenario_tree("branch3", 6)
    cfg = oi.default_cfg(max_iter=5)
    flat, x0, cost = _reference_style_cost_tree(cfg, sst, 1)
    M = len(flat["parent"])
    sol = oi.solve(cfg, flat, x0, sst["target_lane"], sst["target_vel"], 1)
    rng = np.random.default_rng(1)
    xs = sol["xs"] + rng.normal(size=(M, 6)) * np.array([0.7, 0.7, 2.0, 0.1, 3.0, 0.3])
    us = sol["us"] + rng.normal(size=(M, 2))
    want = oi.node_derivs(cfg, flat, x0, sst["target_lane"], sst["target_vel"], 1, xs, us)
    got = hip_predictor.cost_eval(cfg, np.arange(M), xs, us, flat, x0=x0, lane=sst["target_lane"], target_vel=sst["target_vel"], use_exo=1)
    for k in ("l", "l_x", "l_u", "l_xx", "l_uu"):
        assert np.array_equal(got[k], want[k]), k
    for i in (0, M // 2, M - 1):
        assert cost.l(xs[i], us[i], i) == want["l"][i]
        assert np.array_equal(cost.l_x(xs[i], us[i], i), want["l_x"][i])
        assert np.array_equal(cost.l_u(xs[i], us[i], i), want["l_u"][i])
        assert np.array_equal(cost.l_xx(xs[i], us[i], i), want["l_xx"][i])
        assert np.array_equal(cost.l_uu(xs[i], us[i], i), want["l_uu"][i])
        assert not cost.l_ux(xs[i], us[i], i).any()


def test_gen_dist_field_matches_oracle(hip_predictor):
    """G5 spot values: the lane distance field and its grid (ilqr/utils.py:5-22), bit-exact."""
    sst = scripted_scenario_tree("lead", 4)
    cfg = oi.default_cfg()
    x0 = oi.init_state(sst["state"], sst["ctrl"])
    flat = oi.flatten(sst["nodes"])
    fields, gx, gy, off = oi.node_fields(cfg, flat, x0, sst["target_lane"], 0)
    o2, xx, yy, dist = gen_dist_field(x0, sst["target_lane"], (cfg.grid_w, cfg.grid_h), cfg.grid_res)
    assert np.array_equal(o2, off) and np.array_equal(xx[0], gx) and np.array_equal(yy[:, 0], gy)
    wp = np.float64(np.float32(cfg.w_tgt) * flat["prob"][0])
    assert np.array_equal(wp * dist ** 2, fields[0])


def test_solver_rejects_what_the_kernel_does_not_implement(hip_predictor):
    with pytest.raises(NotImplementedError):
        iLQR(object())
    with pytest.raises(NotImplementedError):
        iLQR(BicycleDynamics(), hessians=True)
