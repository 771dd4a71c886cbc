"""Cost / TreeCost of the reference (planners/ilqr/cost.py:25-121, 326-446): instantaneous cost of a
trajectory-tree node = sum of its potentials.  Node ``data`` = ``[[state potentials], [control potentials]]``,
root (key -1) ``data`` = x0, exactly as trajectory_tree.py:26-54 builds it.  Values come from the device
(mind_cost_eval); ``pack()`` hands the whole tree to the solver kernel."""
import numpy as np

from ... import _lib
from ...runtime import get_runtime
from .potential import PotentialField, pack_node_w


class Cost:
    """Instantaneous cost interface (cost.py:25-121)."""

    def l(self, x, u, i, terminal=False):
        raise NotImplementedError

    def l_x(self, x, u, i, terminal=False):
        raise NotImplementedError

    def l_u(self, x, u, i, terminal=False):
        raise NotImplementedError

    def l_xx(self, x, u, i, terminal=False):
        raise NotImplementedError

    def l_ux(self, x, u, i, terminal=False):
        raise NotImplementedError

    def l_uu(self, x, u, i, terminal=False):
        raise NotImplementedError


class TreeCost(Cost):
    def __init__(self, tree, state_size, action_size):
        if (state_size, action_size) != (6, 2):
            raise NotImplementedError("the device model is the 6-state / 2-control bicycle of trajectory_tree.py:153-177")
        self.tree = tree
        self.state_size = state_size
        self.action_size = action_size
        self._packed = None

    # ---- packing for the kernels --------------------------------------------------------------------
    def pack(self):
        """-> dict(x0 [6], parent int32 [M], field [M,H,W], node_w [M,32], grid) with node keys 0..M-1."""
        if self._packed is not None:
            return self._packed
        nodes = self.tree.nodes
        M = len(nodes) - 1
        root = self.tree.get_root()
        parent = np.zeros(M, np.int32)
        node_w = np.zeros((M, 32))
        fields, grid = [], None
        for k in range(M):
            n = nodes[k]                      # keys are 0..M-1 in creation order (trajectory_tree.py:33)
            pk = n.parent_key
            if not (pk == root.key or (isinstance(pk, (int, np.integer)) and 0 <= pk < k)):
                raise ValueError(f"cost tree node {k}: parent {pk} must precede it")
            parent[k] = -1 if pk == root.key else pk
            state_pots, ctrl_pots = n.data
            node_w[k] = pack_node_w(state_pots, ctrl_pots)
            pf = [p for p in state_pots if isinstance(p, PotentialField)]
            if len(pf) != 1:
                raise NotImplementedError("every cost-tree node needs exactly one PotentialField")
            g = pf[0].grid()
            if grid is None:
                grid = g
            elif not (np.array_equal(g["gx"], grid["gx"]) and np.array_equal(g["gy"], grid["gy"]) and g["res"] == grid["res"]
                      and np.array_equal(g["offset"], grid["offset"])):
                raise NotImplementedError("all PotentialFields of a cost tree must share one grid")
            fields.append(np.asarray(pf[0].cost_field, np.float64))
        self._packed = dict(x0=np.asarray(root.data, np.float64), parent=parent, field=np.stack(fields), node_w=node_w, grid=grid)
        return self._packed

    def _eval(self, x, u, i):
        p = self.pack()
        cfg = _lib.IlqrCfg()
        cfg.dt, cfg.wheelbase, cfg.max_iter = 0.2, 2.5, 0
        u = np.zeros(self.action_size) if u is None else u
        return get_runtime().cost_eval(cfg, [int(i)], np.asarray(x, np.float64)[None], np.asarray(u, np.float64)[None], p, grid=p["grid"])

    # ---- cost.py:341-446 ------------------------------------------------------------------------------
    def l(self, x, u, i, terminal=False):
        return self._eval(x, u, i)["l"][0]

    def l_x(self, x, u, i, terminal=False):
        return self._eval(x, u, i)["l_x"][0]

    def l_u(self, x, u, i, terminal=False):
        return self._eval(x, u, i)["l_u"][0]

    def l_xx(self, x, u, i, terminal=False):
        return self._eval(x, u, i)["l_xx"][0]

    def l_ux(self, x, u, i, terminal=False):
        return np.zeros((self.action_size, self.state_size))

    def l_uu(self, x, u, i, terminal=False):
        return self._eval(x, u, i)["l_uu"][0]
