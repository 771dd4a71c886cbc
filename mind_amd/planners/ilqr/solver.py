"""iLQR of the reference (planners/ilqr/solver.py:21-421) on the MI355X: ``fit`` packs the TreeCost and
runs the whole loop -- rollout, tree Riccati sweep, 10-step backtracking line search, Levenberg-Marquardt
schedule, convergence test -- in ONE persistent kernel launch (k_ilqr, generic mode)."""
import numpy as np

from ... import _lib
from ...runtime import get_runtime
from .cost import TreeCost
from .dynamics import BicycleDynamics


class iLQR:
    def __init__(self, dynamics, max_reg=1e10, hessians=False):
        if not isinstance(dynamics, BicycleDynamics):
            raise NotImplementedError("the kernel integrates the kinematic bicycle of trajectory_tree.py:153-177; "
                                      "pass mind_amd.planners.ilqr.dynamics.BicycleDynamics(dt, wheelbase)")
        if hessians or max_reg != 1e10:
            raise NotImplementedError("k_ilqr implements the reference's configuration: max_reg=1e10, hessians=False")
        self.dynamics = dynamics
        self.cost = None
        self.N = None
        self.xs = None
        self.us = None
        self.J_opt = None
        self.iterations = None
        self.converged = None
        self._mu = 1.0

    def fit(self, us_init, cost: TreeCost = None, n_iterations=100):
        """-> (xs [N,6], us [N,2]); node key k of the cost tree <-> row k (solver.py:80-167)."""
        self.cost = cost
        us_init = np.asarray(us_init, np.float64)
        self.N = len(us_init)
        p = cost.pack()
        if len(p["parent"]) != self.N:
            raise ValueError(f"us_init has {self.N} rows, the cost tree {len(p['parent'])} nodes")
        cfg = _lib.IlqrCfg()
        cfg.dt, cfg.wheelbase, cfg.max_iter = float(self.dynamics.dt), float(self.dynamics.wheelbase), int(n_iterations)
        xs, us, st = get_runtime().ilqr_solve_fields(cfg, p["grid"], p, p["x0"], us_init)
        self.xs, self.us = xs, us
        self.J_opt, self._mu = st["J"], st["mu"]
        self.iterations, self.converged = st["iterations"], bool(st["converged"])
        return xs, us
