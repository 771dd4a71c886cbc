"""Potentials of the tree cost (reference planners/ilqr/potential.py:4-264), same constructors.

The objects are parameter holders: ``get_potential / get_gradient / get_hessian`` are evaluated by the
device kernel ``k_cost_eval`` (libmind_hip.so: mind_cost_eval) through a one-node cost tree, the solver
consumes them packed (``mind_amd.planners.ilqr.cost.TreeCost.pack``).  The device cost model keeps the
reference's structure -- quadratic forms with DIAGONAL weights (all the reference's configs use
``np.diag``); a non-diagonal weight raises.
"""
import numpy as np

from ... import _lib
from ...runtime import get_runtime

STATE_SIZE, ACTION_SIZE = 6, 2


def _diag(weight, n, what):
    w = np.asarray(weight, np.float64)
    if w.shape != (n, n) or np.count_nonzero(w - np.diag(np.diagonal(w))):
        raise NotImplementedError(f"{what}: the device cost model takes a diagonal [{n},{n}] weight")
    return np.diagonal(w).copy()


def _dummy_grid():
    return dict(offset=np.zeros(2), res=1.0, gx=np.arange(3.0), gy=np.arange(3.0)), np.zeros((1, 3, 3))


def pack_node_w(state_pots, ctrl_pots):
    """One node's quadratic potentials -> the 32 doubles of mind_cost_tree.node_w
    (w_des[6], w_con[6], lower[6], upper[6], w_ctrl[2], des_state[6]); at most one potential per kind."""
    w = np.zeros(32)
    w[12:18], w[18:24] = -np.inf, np.inf
    seen = set()
    for p in list(state_pots) + list(ctrl_pots):
        kind = type(p)
        if isinstance(p, PotentialField):
            continue
        if kind in seen:
            raise NotImplementedError(f"more than one {kind.__name__} in one cost-tree node")
        seen.add(kind)
        if isinstance(p, StatePotential):
            w[0:6] = _diag(p.weight, STATE_SIZE, "StatePotential")
            w[26:32] = np.asarray(p.des_state, np.float64)
        elif isinstance(p, StateConstraint):
            w[6:12] = _diag(p.weight, STATE_SIZE, "StateConstraint")
            w[12:18] = np.asarray(p.lower_bound, np.float64)
            w[18:24] = np.asarray(p.upper_bound, np.float64)
        elif isinstance(p, ControlPotential):
            w[24:26] = _diag(p.weight, ACTION_SIZE, "ControlPotential")
        else:
            raise NotImplementedError(f"unsupported potential {kind.__name__}")
    return w


def _eval(state_pots, ctrl_pots, x, u):
    """Device evaluation of the summed potentials of one node at (x, u)."""
    fields = [p for p in state_pots if isinstance(p, PotentialField)]
    if len(fields) > 1:
        raise NotImplementedError("more than one PotentialField in one cost-tree node")
    if fields:
        grid, fld = fields[0].grid(), np.asarray(fields[0].cost_field, np.float64)[None]
    else:
        grid, fld = _dummy_grid()
    tree = dict(parent=np.array([-1], np.int32), field=fld, node_w=pack_node_w(state_pots, ctrl_pots)[None])
    cfg = _lib.IlqrCfg()
    cfg.dt, cfg.wheelbase, cfg.max_iter = 0.2, 2.5, 0
    x = np.zeros(STATE_SIZE) if x is None else np.asarray(x, np.float64)
    u = np.zeros(ACTION_SIZE) if u is None else np.asarray(u, np.float64)
    return get_runtime().cost_eval(cfg, [0], x[None], u[None], tree, grid=grid)


class ControlPotential:
    def __init__(self, weight):
        self.weight = weight

    def get_potential(self, control):
        return _eval([], [self], None, control)["l"][0]

    def get_gradient(self, control):
        return _eval([], [self], None, control)["l_u"][0]

    def get_hessian(self, control):
        return _eval([], [self], None, control)["l_uu"][0]


class StateConstraint:
    def __init__(self, weight, lower_bound, upper_bound):
        self.weight = weight
        self.lower_bound = lower_bound
        self.upper_bound = upper_bound

    def get_potential(self, state):
        return _eval([self], [], state, None)["l"][0]

    def get_gradient(self, state):
        return _eval([self], [], state, None)["l_x"][0]

    def get_hessian(self, state):
        return _eval([self], [], state, None)["l_xx"][0]


class StatePotential:
    def __init__(self, weight, des_state):
        self.des_state = des_state
        self.weight = weight

    def get_potential(self, state):
        return _eval([self], [], state, None)["l"][0]

    def get_gradient(self, state):
        return _eval([self], [], state, None)["l_x"][0]

    def get_hessian(self, state):
        return _eval([self], [], state, None)["l_xx"][0]


class PotentialField:
    """Smoothed biquadratic interpolation of a cost grid (potential.py:62-264): half-to-even cell lookup,
    the reference's border windows, 3x3 smoothing, Bernstein-2 value / gradient / Hessian -- all inside
    the kernel (ilqr_kernels.hip: il_field)."""

    def __init__(self, field_offset, resolution, xx, yy, cost_field):
        self.offset = field_offset
        self.res = resolution
        self.xx = xx
        self.yy = yy
        self.limits = (np.min(xx), np.max(xx), np.min(yy), np.max(yy))
        self.cost_field = cost_field

    def grid(self):
        xx, yy = np.asarray(self.xx, np.float64), np.asarray(self.yy, np.float64)
        return dict(offset=np.asarray(self.offset, np.float64), res=float(self.res), gx=xx[0, :].copy(), gy=yy[:, 0].copy())

    def get_limits(self):
        return self.limits

    def _state(self, state):
        s = np.zeros(STATE_SIZE)
        s[:len(state)] = np.asarray(state, np.float64)[:STATE_SIZE]
        return s

    def get_potential(self, state):
        return _eval([self], [], self._state(state), None)["l"][0]

    def get_gradient(self, state):
        g = np.zeros(len(state))
        g[:2] = _eval([self], [], self._state(state), None)["l_x"][0][:2]
        return g

    def get_hessian(self, state):
        h = np.zeros((len(state), len(state)))
        h[:2, :2] = _eval([self], [], self._state(state), None)["l_xx"][0][:2, :2]
        return h
