"""ORACLE wrapper (test infrastructure): ctypes binding of oracle/ilqr_ref.c plus the scenario-tree
-> cost-tree flattening restated from planners/mind/trajectory_tree.py:27-54,67-122 (LIFO DFS,
every even sub-step becomes a trajectory node, keys in creation order, Q13)."""
import ctypes as C
import os

import numpy as np

from mind_amd._lib import CostTree, IlqrCfg, IlqrStats

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _load(name):
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", _HERE])
    L = C.CDLL(path)
    L.oracle_ilqr_solve.restype = C.c_int
    L.oracle_field_eval.restype = C.c_int
    L.oracle_lane_field.restype = C.c_int
    return L


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _load("libilqr_oracle.so")
    return _LIB


class libm_trig:
    """``with oracle.ilqr.libm_trig(): ...`` -- the calls inside run the oracle built with the C library's sin / cos / tan
    (libilqr_oracle_libm.so) instead of mind_trig.h: the independent witness for the header kernel and oracle share."""

    def __enter__(self):
        global _LIB
        self.before = lib()
        _LIB = _load("libilqr_oracle_libm.so")
        return self

    def __exit__(self, *exc):
        global _LIB
        _LIB = self.before


def default_cfg(w_vel=0.1, max_iter=100):
    """TrajTreeCfg of planners/mind/configs/planning/demo_1.py:13-81 (demo_3: w_vel=.5)."""
    c = IlqrCfg()
    c.dt, c.wheelbase = 0.2, 2.5
    c.w_des_state[:] = [0, 0, w_vel, 0, 1.0, 10.0]
    c.w_state_con[:] = [0, 0, 50.0, 0, 50.0, 500.0]
    c.state_upper[:] = [100000.0, 100000.0, 8.0, 10.0, 4.0, 0.2]
    c.state_lower[:] = [-100000.0, -100000.0, 0.0, -10.0, -6.0, -0.2]
    c.w_ctrl[:] = [5.0, 5.0]
    c.w_tgt, c.w_ego, c.w_ego_cov_offset = 1.0, 1.0, 1.0
    c.w_exo, c.w_exo_cov_offset, c.w_exo_cost_offset = 10.0, 2.5, 10.0
    c.grid_res, c.grid_w, c.grid_h = 0.4, 256, 256
    c.max_iter = max_iter
    return c


def flatten(nodes):
    """nodes: [(key, parent_key, [prob, trajs, covs, tgt])] insertion order -> flat cost tree arrays.
    Mirrors the ``queue.pop()`` (LIFO) traversal: the LAST child is expanded first."""
    by_key = {k: (p, d) for k, p, d in nodes}
    children = {k: [] for k, _, _ in nodes}
    root = None
    for k, p, _ in nodes:
        if p is None:
            root = k
        else:
            children[p].append(k)
    parent, prob, mean, cov = [], [], [], []
    last = {}
    stack = [root]
    while stack:
        k = stack.pop()
        p, (pr, trajs, covs, _) = by_key[k]
        li = last[p] if p is not None else -1
        for i in range(trajs.shape[1]):
            if i % 2 == 1:
                continue
            cur = len(parent)
            parent.append(li)
            prob.append(np.float32(pr))
            mean.append(trajs[:, i, :])
            cov.append(covs[:, i, 0])
            li = cur
        last[k] = len(parent) - 1
        stack.extend(children[k])
    return dict(parent=np.asarray(parent, np.int32), prob=np.asarray(prob, np.float32),
                mean=np.ascontiguousarray(np.stack(mean), np.float32),
                cov=np.ascontiguousarray(np.stack(cov), np.float32))


def _ct(flat):
    t = CostTree()
    t.n_nodes = len(flat["parent"])
    t.parent = flat["parent"].ctypes.data_as(C.POINTER(C.c_int32))
    t.prob = flat["prob"].ctypes.data_as(C.POINTER(C.c_float))
    t.n_agents = flat["mean"].shape[1]
    t.agent_mean = flat["mean"].ctypes.data_as(C.POINTER(C.c_float))
    t.agent_cov = flat["cov"].ctypes.data_as(C.POINTER(C.c_float))
    return t


def solve(cfg, flat, x0, lane, target_vel, use_exo, us_init=None, trace=False):
    M = len(flat["parent"])
    x0 = np.ascontiguousarray(x0, np.float64)
    lane = np.ascontiguousarray(lane, np.float64)
    xs = np.zeros((M, 6))
    us = np.zeros((M, 2))
    st = IlqrStats()
    jt = np.zeros(cfg.max_iter) if trace else None
    ui = None if us_init is None else np.ascontiguousarray(us_init, np.float64)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None
    t = _ct(flat)
    rc = lib().oracle_ilqr_solve(C.byref(cfg), C.byref(t), dp(x0), dp(lane), C.c_int(len(lane)),
                                 C.c_double(target_vel), C.c_int(use_exo), dp(ui), dp(xs), dp(us), C.byref(st), dp(jt))
    assert rc == 0
    out = dict(xs=xs, us=us, iterations=st.iterations, converged=st.converged, J=st.J, mu=st.mu)
    if trace:
        out["J_trace"] = jt
        # per-iteration rows {mu, J_opt, accepted alpha index (-1 rejected, -2 LinAlgError), J of the accepted candidate}
        tb = np.zeros((256, 4))
        n = lib().oracle_ilqr_last_trace(dp(tb), C.c_int(256))
        out["trace"] = tb[:min(n, 256)].copy()
    return out


def init_state(state, ctrl):
    """trajectory_tree.py:149-151"""
    return np.array([state[0], state[1], state[2], state[3], ctrl[0], ctrl[1]], np.float64)


def contingency(cfg, nodes, state, ctrl, lane, target_vel):
    """get_traj_tree (planner.py:174-178): warm-start solve then full solve."""
    flat = flatten(nodes)
    x0 = init_state(state, ctrl)
    w = solve(cfg, flat, x0, lane, target_vel, 0)
    f = solve(cfg, flat, x0, lane, target_vel, 1, us_init=w["us"])
    return flat, w, f


def field_eval(F, res, origin, px, py):
    F = np.ascontiguousarray(F, np.float64)
    out = np.zeros(6)
    o = np.ascontiguousarray(origin, np.float64)
    lib().oracle_field_eval(F.ctypes.data_as(C.POINTER(C.c_double)), C.c_int(F.shape[1]), C.c_int(F.shape[0]),
                            C.c_double(res), o.ctypes.data_as(C.POINTER(C.c_double)), C.c_double(px), C.c_double(py),
                            out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def node_fields(cfg, flat, x0, lane, use_exo):
    """Materialised cost field of every trajectory node + the grid (field_offset, xx[0,:], yy[:,0]) exactly as
    trajectory_tree.py builds them for PotentialField.  -> (fields [M,H,W], gx [W], gy [H], off [2])."""
    M, W, H = len(flat["parent"]), cfg.grid_w, cfg.grid_h
    x0 = np.ascontiguousarray(x0, np.float64)
    lane = np.ascontiguousarray(lane, np.float64)
    fields, gx, gy, off = np.zeros((M, H, W)), np.zeros(W), np.zeros(H), np.zeros(2)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    t = _ct(flat)
    rc = lib().oracle_node_fields(C.byref(cfg), C.byref(t), dp(x0), dp(lane), C.c_int(len(lane)), C.c_int(use_exo),
                                  dp(fields), dp(gx), dp(gy), dp(off))
    assert rc == 0
    return fields, gx, gy, off


def node_derivs(cfg, flat, x0, lane, target_vel, use_exo, xs, us):
    """l, l_x, l_u, l_xx, l_uu of every node i at (xs[i], us[i]) -> dict of arrays."""
    M = len(flat["parent"])
    x0 = np.ascontiguousarray(x0, np.float64)
    lane = np.ascontiguousarray(lane, np.float64)
    xs = np.ascontiguousarray(xs, np.float64)
    us = np.ascontiguousarray(us, np.float64)
    out = np.zeros((M, 47))
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    t = _ct(flat)
    rc = lib().oracle_node_derivs(C.byref(cfg), C.byref(t), dp(x0), dp(lane), C.c_int(len(lane)), C.c_double(target_vel),
                                  C.c_int(use_exo), dp(xs), dp(us), dp(out))
    assert rc == 0
    luu = np.zeros((M, 2, 2))
    luu[:, 0, 0], luu[:, 1, 1] = out[:, 45], out[:, 46]
    return dict(l=out[:, 0], l_x=out[:, 1:7], l_u=out[:, 7:9], l_xx=out[:, 9:45].reshape(M, 6, 6), l_uu=luu)
