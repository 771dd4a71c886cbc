"""AIME (Adaptive Interaction Modality Exploration) scenario-tree generator.

Call surface of the reference ``planners/mind/scenario_tree.py`` (``ScenarioTreeGenerator(device,
network, obs_len, pred_len, config)``, ``reset``, ``set_target_lane``, ``branch_aime``); the batched
scene prediction of every round runs on the MI355X through ``network`` (``ScenePredNet`` ->
libmind_hip.so), the per-round bookkeeping is vectorised numpy on the host (float32, the dtype the
reference computes these steps in).

Reference map (file:line in planners/mind/scenario_tree.py):
  branch_aime :38-58, init_scenario_tree :60-67, create_nodes :73-80, decide_branch :82-100,
  get_branch_set :102-108, set_target_lane :110-120, process_data :122-206, get_scenario_tree :208-272,
  prune_merge :281-412, prepare_root_data :414-465, update_obser :467-567, get_branch_time :592-611,
  get_high_level_command :613-652.  Quirks Q5, Q8, Q12, Q20 are reproduced.
"""
import os

import numpy as np
import torch

from ..basic.tree import LazyTree, Node, Tree
from . import utils as U

F32 = np.float32


class ScenarioData:
    def __init__(self, data, obs_data, branch_flag=False, end_flag=False, terminate_flag=False):
        self.data = data
        self.obs_data = obs_data
        self.branch_flag = branch_flag
        self.end_flag = end_flag
        self.terminate_flag = terminate_flag


class DevScene(dict):
    """Observation of a branch node whose predictor inputs were built on the device (mind_aime_rebase): holds the host
    fields the AIME bookkeeping reads (frames, world-frame history windows, ids) plus a handle on the device batch.
    The host-side predictor inputs (ACTORS, LANE_CTRS, TGT_NODES, ...) are only materialised if somebody asks."""
    _HOST_KEYS = ("TRAJS_POS_OBS", "TRAJS_ANG_OBS", "TRAJS_VEL_OBS", "PAD_OBS", "TRAJS_CTRS", "TRAJS_VECS", "ACTORS",
                  "LANES", "LANE_CTRS", "LANE_VECS", "TGT_NODES", "TGT_RPE")

    def __init__(self, gen, dev, g, fields, lazy=None, cov_last=None):
        super().__init__(fields)
        self.gen, self.dev, self.g = gen, dev, g
        # windows assembled on the device (mind_aime_rebase with a device source): the host copies of the four 50-step history
        # windows are cut out of (child, w0, keep) only if somebody asks for them by name
        self.lazy = lazy
        if lazy is not None:
            self.hist_n, self.cov_last = gen.obs_len, cov_last

    def __contains__(self, key):
        return (self.lazy is not None and key in _HIST_COLS) or dict.__contains__(self, key)

    def __missing__(self, key):
        if self.lazy is not None and key in _HIST_COLS:
            c, w0, keep = self.lazy
            w6 = c.window6(w0, keep)
            self._win6 = w6
            for k, col in _HIST_COLS.items():
                dict.__setitem__(self, k, w6[:, :, col])
            return dict.__getitem__(self, key)
        if key not in self._HOST_KEYS:
            raise KeyError(key)
        full = self.gen._host_obs(self["TRAJS_POS_HIST"], self["TRAJS_ANG_HIST"], self["TRAJS_VEL_HIST"], self["TRAJS_TYPE"])
        for k in self._HOST_KEYS:
            dict.__setitem__(self, k, full[k])
        return dict.__getitem__(self, key)


_HIST_COLS = {"TRAJS_POS_HIST": slice(0, 2), "TRAJS_VEL_HIST": slice(2, 4), "TRAJS_ANG_HIST": 4, "TRAJS_COV_HIST": slice(5, 6)}


class ChildScene(dict):
    """A kept mode of prune_merge (scenario_tree.py:396-412).  The reference stores the parent's world-frame history extended
    by the mode's 60 predicted steps, truncated to seq_len; here the child keeps a reference to the parent's arrays and a
    view of its own rows [a,60,6] (x, y, vx, vy, heading, max-sigma) and cuts what a consumer needs out of the two pieces
    (``rows``): the branch-time test reads the predicted sigmas, re-basing the last 50 steps, the returned tree the first
    `dur` predicted steps -- the four concatenated arrays (169 KB per child at 64 agents) are only built if somebody asks
    for them by name."""

    def __init__(self, fields, parent, new, seq_len, shared=None):
        super().__init__(fields)
        self.parent, self.new, self.n_hist = parent, new, hist_len(parent)
        self.shared = shared if shared is not None else {}          # siblings share the parent's packed window
        self.length = min(self.n_hist + new.shape[1], seq_len)          # Q8: truncated to seq_len

    def trim(self, keep):
        """update_obser's in-place truncation of the stored history (scenario_tree.py:470-473)"""
        self.length = min(self.length, keep)
        for k in _HIST_COLS:
            if dict.__contains__(self, k):
                dict.__setitem__(self, k, dict.__getitem__(self, k)[:, :keep])

    def rows(self, key, lo, hi):
        """[a, hi-lo, ...] = rows lo..hi of the (virtual) concatenated history array `key`"""
        hi = min(hi, self.length)
        col, h = _HIST_COLS[key], self.n_hist
        if lo >= h:
            return self.new[:, lo - h:hi - h, col]
        par = self.parent[key]
        if hi <= h:
            return par[:, lo:hi]
        return np.concatenate([par[:, lo:h], self.new[:, :hi - h, col]], axis=1)

    def window6(self, lo, hi):
        """rows lo..hi of the packed history [a, hi-lo, 6] = (x, y, vx, vy, heading, max-sigma): ONE concatenation (the parent's
        packed window is built once per parent and shared by its children) instead of one per array"""
        h = self.n_hist
        if lo >= h:
            return self.new[:, lo - h:hi - h]
        par = self.parent
        p6 = self.shared.get("p6")
        if p6 is None:
            p6 = getattr(par, "_win6", None)
            if p6 is None:
                p6 = np.concatenate([par["TRAJS_POS_HIST"], par["TRAJS_VEL_HIST"], par["TRAJS_ANG_HIST"][..., None], par["TRAJS_COV_HIST"]], axis=2)
            self.shared["p6"] = p6
        return np.concatenate([p6[:, lo:h], self.new[:, :hi - h]], axis=1)

    def __missing__(self, key):
        if key not in _HIST_COLS:
            raise KeyError(key)
        v = self.rows(key, 0, self.length)
        dict.__setitem__(self, key, v)
        return v

    def __contains__(self, key):
        return key in _HIST_COLS or dict.__contains__(self, key)


def hist_rows(d, key, lo, hi):
    """rows lo..hi of a node's history array, without materialising a ChildScene's concatenation"""
    return d.rows(key, lo, hi) if isinstance(d, (ChildScene, NativeScene)) else d[key][:, lo:hi]


def hist_trim(d, keep):
    if isinstance(d, ChildScene):
        d.trim(keep)
    else:
        for k in _HIST_COLS:
            d[k] = d[k][:, :keep]


def hist_len(d):
    if isinstance(d, ChildScene):
        return d.length
    n = getattr(d, "hist_n", None)           # DevScene whose windows were assembled on the device
    return n if n is not None else d["TRAJS_POS_HIST"].shape[1]


def cov_last_of(d):
    """TRAJS_COV_HIST[:, -1, 0] of a scene, without materialising a lazily held history"""
    v = getattr(d, "cov_last", None)
    return v if v is not None else d["TRAJS_COV_HIST"][:, -1, 0]


class NativeScene(dict):
    """Data of a tree node grown by mind_aime_plan (the whole AIME loop in native code): the fields the returned scenario trees and
    the planner read, and the node's first `dur` predicted steps [a,dur,3] = (x, y, max-sigma) -- the only rows get_scenario_tree
    (scenario_tree.py:208-272) attaches."""

    def __init__(self, fields, packed, obs_len):
        super().__init__(fields)
        self.packed, self.obs_len = packed, obs_len

    def rows(self, key, lo, hi):
        if lo != self.obs_len or self.packed is None or key not in ("TRAJS_POS_HIST", "TRAJS_COV_HIST"):
            raise KeyError(f"{key}[{lo}:{hi}] of a natively grown node is not held on the host")
        n = min(hi - lo, self.packed.shape[1])
        return self.packed[:, :n, 0:2] if key == "TRAJS_POS_HIST" else self.packed[:, :n, 2:3]


class RemoteScene(dict):
    """Observation of a branch node that ANOTHER rank re-bases and expands (sharded rounds): only the world-frame fields the
    tree bookkeeping and the child assembly read; no predictor inputs exist on this rank."""


def _np(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)


class ScenarioTreeGenerator:
    def __init__(self, device, network, obs_len=50, pred_len=60, config=None):
        self.device = device
        self.network = network
        self.obs_len = obs_len
        self.pred_len = pred_len
        self.seq_len = obs_len + pred_len
        self.config = config
        self.tree = Tree()
        self.lane_graph = None
        self.lane_feat_in = None      # [l,10,16] instance-frame lane features (constant within a plan)
        self.lane_feat_cache = None   # LaneNet output on device, reused by every tree node of the plan
        self.target_lane = None
        self.target_lane_info = None
        self.ego_idx = 0
        self.device_glue = True     # prune_merge arithmetic on the device when the network leaves its outputs there
        # re-basing windows of the kept modes cut out of the device-resident rows (mind_aime_rebase device source) instead of stacked
        # and uploaded: bit-identical; - 5 ms per cfg4 plan when the planner's context stream is a non-blocking stream, but on the
        # legacy default stream the call stalls on entry for longer than the host-side window building took.  None =
        # decide by the runtime's stream; MIND_DEVICE_WINDOWS=0/1 forces it
        env = os.environ.get("MIND_DEVICE_WINDOWS")
        self.device_windows = None if env is None else env == "1"
        self.device_select = True   # ... and its pruning decisions (k_aime_select); False: decided on the host from the device's signatures
        # the whole AIME loop in ONE native call (mind_aime_plan: per-round bookkeeping in C++, one small read-back per round) when the
        # network is the HIP predictor itself; MIND_NATIVE_AIME=0 / the attribute keep the round-by-round path (same kernels, same trees)
        self.native_aime = os.environ.get("MIND_NATIVE_AIME", "1") != "0"
        # ... with the root scene featurised on the device as well (process_data / prepare_root_data as kernels: k_aime_rebase on the
        # raw windows, k_aime_root_lanes, k_aime_root_hist); MIND_DEVICE_ROOT=0 / the attribute: the host featuriser feeds the native plan
        self.device_root = os.environ.get("MIND_DEVICE_ROOT", "1") != "0"
        self.n_lanes = None
        self.n_native_plans = 0
        self.branch_depth = 0
        self.n_expanded = 0           # scenes pushed through the predictor (metric: nodes expanded)
        self.shard = None             # mind_amd.parallel.Shard: block-distribute each round's scenes over ranks
        self._plan_round = 0

    # ------------------------------------------------------------------------------------------
    def reset(self):
        self.branch_depth = 0
        self.tree = Tree()
        self.last_trees = None
        self.lane_feat_cache = None
        self._plan_round = 0
        self._root_todo = None

    def set_target_lane(self, target_lane, target_lane_info):
        self.target_lane = np.asarray(target_lane).astype(F32) if np.asarray(target_lane).dtype != np.float64 \
            else np.asarray(target_lane)
        # the reference keeps the lane in whatever dtype numpy gave it and moves it to a torch tensor;
        # every consumer computes in float32 after mixing with float32 operands
        self.target_lane = np.asarray(target_lane, dtype=F32)
        cols = [np.asarray(target_lane_info[0])[:, None], np.asarray(target_lane_info[1]),
                np.asarray(target_lane_info[2]), np.asarray(target_lane_info[3]),
                np.asarray(target_lane_info[4])[:, None], np.asarray(target_lane_info[5])[:, None]]
        self.target_lane_info = np.concatenate(cols, axis=-1).astype(F32)

    # ------------------------------------------------------------------------------------------
    def _native_ok(self):
        net = self.network
        if type(net).__name__ in ("ScriptedBranching", "ScriptedFullTree", "ScriptedDeepTree", "ScriptedDeeperTree"):      # scripted modes on top of the real forward (mind_amd/synth.py)
            net = net.net
        return (self.native_aime and self.device_glue and self.device_select and type(net).__name__ == "ScenePredNet"
                and getattr(net, "rt", None) is not None and getattr(net, "_loaded", False) and hasattr(net.rt, "aime_plan")
                and (self.shard is None or not self.shard.sharded or getattr(self.shard, "native", False)) and self.ego_idx == 0 and self.target_lane is not None
                and len(self.target_lane) >= 12 and self.config is not None)

    def _sync_exchange(self, rt):
        """the context's exchange (mind_set_exchange) is this planner's group or none: contexts are shared by the planners of a thread"""
        want = self.shard if (self.shard is not None and getattr(self.shard, "native", False)) else None
        if getattr(rt, "_exchange_owner", None) is not want:
            if want is None:
                from ...parallel import Shard
                Shard.detach(rt)
            else:
                want.attach(rt)

    def _native_args(self, lcl_smp, agent_obs, solve=None):
        """(root dict, positional args, keyword args) of the runtime's aime_plan for this cycle, or None when the native plan does not
        apply (wrong horizons, no lanes)"""
        cfg = self.config
        if self.obs_len != 50 or not (2 <= self.pred_len <= 60):
            return None
        self._sync_exchange(self.network.rt)
        scripted = type(self.network).__name__ in ("ScriptedBranching", "ScriptedFullTree", "ScriptedDeepTree", "ScriptedDeeperTree")
        floor = getattr(self.network, "prob_floor", None) if scripted else None
        modes = (lambda n_agents: self.network._modes(n_agents, self.network.rt.device)) if scripted else (lambda n_agents: None)
        if self.device_root:
            # process_data's host part is reduced to get_agent_trajectories (Track lists -> padded arrays); frames, actor features,
            # lane graph, high-level command and the root's world-frame histories are computed by the library on the device
            pos, ang, vel, types, flags, tids, cats = U.get_agent_trajectories(agent_obs)
            st = U._static_lane_pieces(lcl_smp.map_data, 15.0, 10)
            if st["num_lanes"] == 0 or pos.shape[1] != self.obs_len:
                return None
            cur_vel = lcl_smp.ego_agent.state[2]
            raw = dict(pos=pos, ang=ang, vel=vel, pad=flags, types=types, lane_pts=st["pts"], lane_flags=st["flags"],
                       travel0=F32(max(cur_vel, 0.5) * cfg.tar_time_ahead))
            root = {"TRAJS_TYPE": types, "TRAJS_TID": tids, "TRAJS_CAT": cats}
            self.n_lanes = int(st["num_lanes"])
            return root, (None, None, None, None, self.target_lane, self.target_lane_info, cfg.tar_time_ahead, cfg.tar_dist_thres, cfg.max_depth), \
                dict(pred_len=self.pred_len, raw=raw, script=modes(pos.shape[0]), prob_floor=floor, solve=solve)
        else:
            root = self.process_data(lcl_smp, agent_obs)
            self.prepare_root_data(root)
            hist = np.concatenate([root["TRAJS_POS_HIST"], root["TRAJS_VEL_HIST"], root["TRAJS_ANG_HIST"][..., None], root["TRAJS_COV_HIST"]], axis=2)
            if hist.shape[1] != self.obs_len or root["LANES"].shape[0] == 0:
                return None
            self.n_lanes = int(root["LANES"].shape[0])
            return root, (root, hist, self.lane_graph["lane_ctrs"], self.lane_graph["lane_vecs"], self.target_lane, self.target_lane_info,
                          cfg.tar_time_ahead, cfg.tar_dist_thres, cfg.max_depth), \
                dict(pred_len=self.pred_len, script=modes(root["ACTORS"].shape[0]), prob_floor=floor, solve=solve)

    def _branch_aime_native(self, lcl_smp, agent_obs, on_flats=None, solve=None):
        """branch_aime through mind_aime_plan; None = the library left this plan to the round-by-round path.  ``on_flats``: called
        with the plan's flattened cost trees (in get_scenario_tree's order) as soon as the native call returns, BEFORE the tree's
        Python objects are built -- the planner starts the contingency solves there (TrajectoryTreeOptimizer.solve_batch_begin)."""
        call = self._native_args(lcl_smp, agent_obs, solve)
        if call is None:
            return None
        root, args, kw = call
        return self._native_trees(self.network.rt.aime_plan(*args, **kw), root, on_flats)

    def branch_aime_begin(self, lcl_smp, agent_obs, solve=None):
        """First half of branch_aime for a driver that plans several scenes from one thread (mind_amd/pipelined.py): starts the native
        plan on a thread of the library (mind_aime_plan_begin) and returns a token, or None when only the blocking paths apply (the caller
        then calls branch_aime).  branch_aime_ready(token) tells whether branch_aime_finish(token, on_flats) would block."""
        if not self._native_ok() or not hasattr(self.network.rt, "aime_plan_begin"):
            return None
        call = self._native_args(lcl_smp, agent_obs, solve)
        if call is None:
            return None
        root, args, kw = call
        self.network.rt.aime_plan_begin(*args, **kw)
        return (root, lcl_smp, agent_obs)

    def branch_aime_ready(self, token):
        return self.network.rt.aime_plan_ready()

    def branch_aime_finish(self, token, on_flats=None, host_context=None):
        root, lcl_smp, agent_obs = token
        trees = self._native_trees(self.network.rt.aime_plan_finish(), root, on_flats)
        if trees is not None:
            return trees
        self.reset()                # the library left this plan to the round-by-round path
        import contextlib
        with (host_context() if host_context is not None else contextlib.nullcontext()):
            return self._branch_aime_host(lcl_smp, agent_obs)

    def _native_trees(self, res, root, on_flats, count=True):
        if res is None:
            return None
        nodes, rows, info = res
        if on_flats is not None:      # (solves_begun: the library began the plan's contingency solves itself, behind its last kernel)
            on_flats([flat for _, flat in info["flats"]], info.get("solves_begun", False)) if getattr(on_flats, "takes_begun", False) \
                else on_flats([flat for _, flat in info["flats"]])
        rf = info["root_flags"]
        a = info["a"]
        col = {f: nodes[f] for f in ("round", "scene", "mode")}

        def node_key(i):
            return "{}_{}_{}".format(int(col["round"][i]), int(col["scene"][i]), int(col["mode"][i]))

        def build_aime_tree(tree):
            """the AIME tree as Python objects: built when somebody looks (tests, the visualiser, get_scenario_tree's reference walk)"""
            tree.add_node(Node("root", None, ScenarioData(None, root, branch_flag=bool(rf & 1), end_flag=bool(rf & 2), terminate_flag=bool(rf & 4))))
            keys = []
            types, tids, cats = root["TRAJS_TYPE"], root["TRAJS_TID"], root["TRAJS_CAT"]
            # the node table column by column (one conversion per field instead of one numpy scalar per field and node)
            cl = {f: nodes[f].tolist() for f in ("round", "scene", "mode", "parent", "dur", "row_off", "flags", "cur_t", "end_t")}
            probs, tgt = nodes["prob"], nodes["tgt_pts"].reshape(-1, 11, 2)
            for i in range(len(nodes)):
                key = "{}_{}_{}".format(cl["round"][i], cl["scene"][i], cl["mode"][i])
                pkey = "root" if cl["parent"][i] < 0 else keys[cl["parent"][i]]
                dur, off, fl = cl["dur"][i], cl["row_off"][i], cl["flags"][i]
                packed = rows[off:off + a * dur * 3].reshape(a, dur, 3) if off >= 0 else None
                d = NativeScene({"SCEN_PROB": probs[i], "CUR_T": cl["cur_t"][i], "END_T": cl["end_t"][i], "PARENT_ID": pkey, "SCEN_ID": key,
                                 "TRAJS_TYPE": types, "TRAJS_TID": tids, "TRAJS_CAT": cats, "TGT_PTS": tgt[i]}, packed, self.obs_len)
                tree.add_node(Node(key, pkey, ScenarioData(d, None, branch_flag=bool(fl & 1), end_flag=bool(fl & 2), terminate_flag=bool(fl & 4))))
                keys.append(key)

        self.tree = LazyTree(build_aime_tree, root="root")
        if count:           # (count=False: the native loop -- mind_amd/native_loop.py -- counted the plan when it ran and builds its objects later)
            self.n_expanded += info["n_expanded"]
            self.branch_depth = info["n_rounds"]
            self.n_native_plans += 1
        # The scenario trees handed to the contingency planner: the library flattened them already (TrajectoryTreeOptimizer.solve_batch
        # takes `_flat` instead of walking the nodes), in get_scenario_tree's order; their Python nodes -- tens of thousands on the deep
        # stress trees, a third of such a plan's wall time -- are built by the reference walk below only when somebody reads them.
        shared = {}
        plan_tree = self.tree         # THIS plan's AIME tree: a scenario tree first read after the next reset() / plan must not see another one's

        def build_scenario_tree(idx):
            def build(tree):
                if "full" not in shared:
                    shared["full"] = self._scenario_trees_from_tree(plan_tree)
                src = shared["full"][idx]
                tree._nodes, tree._lv, tree.root = src.nodes, src._leaves, src.root
            return build

        trees = []
        for idx, (top, flat) in enumerate(info["flats"]):
            t = LazyTree(build_scenario_tree(idx), root=node_key(top))
            t._flat = flat
            trees.append(t)
        self.last_trees = trees
        return trees

    def branch_aime(self, lcl_smp, agent_obs, on_flats=None, solve=None):
        if self._native_ok():
            trees = self._branch_aime_native(lcl_smp, agent_obs, on_flats, solve)
            if trees is not None:
                return trees
            self.reset()            # (keeps lane graph / target lane: only the per-plan bookkeeping)
        return self._branch_aime_host(lcl_smp, agent_obs)

    def _branch_aime_host(self, lcl_smp, agent_obs):
        root = self.process_data(lcl_smp, agent_obs)
        self.init_scenario_tree(root)
        branch_nodes = self.get_branch_set()
        while branch_nodes:
            batch = [n.data.obs_data for n in branch_nodes]
            self.create_nodes(self.expand(batch))
            self.decide_branch()
            branch_nodes = self.get_branch_set()
        assert len(self.get_end_set()) > 0, "No end node found in the scenario tree."
        return self.get_scenario_tree()

    def init_scenario_tree(self, root):
        # prepare_root_data (the root's world-frame histories: read by prune_merge, not by the predictor) runs while the first
        # predictor call is in flight (expand -> _finish_root)
        self._root_todo = root
        self.tree.add_node(Node("root", None, ScenarioData(None, root, branch_flag=True)))
        self.create_nodes(self.expand([root]))
        self.decide_branch()

    HDR = 25      # per kept child: [scene index in the round, mode, path probability, TGT_PTS (11 x 2)]

    def _use_device_windows(self, rt, n_scenes):
        """device-side window assembly for this round?  Forced by MIND_DEVICE_WINDOWS / the attribute; otherwise only on a
        non-default context stream and for rounds of >= 16 kept modes (a handful of windows upload faster than they pay)"""
        if self.device_windows is not None:
            return self.device_windows
        return n_scenes >= 16 and not getattr(rt, "on_default_stream", True)

    def _finish_root(self):
        root, self._root_todo = getattr(self, "_root_todo", None), None
        if root is not None:
            self.prepare_root_data(root)

    def expand(self, batch):
        """One AIME round: predict + prune/merge every scene of the branch set, then build the child dicts.  With a
        Shard the scenes are block-distributed over the ranks; the round's only exchange step is one packed all-gather
        of the kept children (header [P,25] + world-frame rows [sum a,60,6] of the surviving modes, device tensors under
        RCCL), after which every rank assembles the same list in batch order.  The first round of a plan (the root
        scene, on rank 0) is followed by a broadcast of LaneNet's output, which every later round reuses."""
        sh = self.shard
        if sh is None or not sh.sharded:
            pred = self.predict_scenes(batch) if batch else None
            self._finish_root()
            hdr, rows = self.prune_select(batch, pred, 0)
            return self.assemble_children(batch, hdr, _np(rows), rows if getattr(self, "_rows_persistent", False) else None)
        lo, hi = sh.block(len(batch))
        mine = batch[lo:hi]
        assert not any(isinstance(s, RemoteScene) for s in mine), "a rank was dealt a scene it did not re-base"
        first_round = self._plan_round == 0
        self._plan_round += 1
        pred = self.predict_scenes(mine) if mine else None
        self._finish_root()
        if mine:
            hdr, rows = self.prune_select(mine, pred, lo)
        else:
            hdr, rows = np.zeros((0, self.HDR), F32), torch.zeros(0, 60, 6)
        if first_round:
            self._broadcast_lane_features(sh)
        hdr_all, rows_all = sh.all_gather_rows(torch.from_numpy(np.ascontiguousarray(hdr, F32)), rows.reshape(-1, 360))
        return self.assemble_children(batch, hdr_all.cpu().numpy(), rows_all.cpu().numpy().reshape(-1, 60, 6))

    def _broadcast_lane_features(self, sh):
        """LaneNet's output [l,128] (instance-frame lane features: constant within a plan, utils.py:171-177) from the rank
        that ran the root scene to every rank: RCCL broadcast of the device tensor (a host tensor over gloo).  The last
        element says whether the sender's network left a cache at all (host networks of the CPU tests do not)."""
        l = int(self.lane_feat_in.shape[0])
        buf = torch.zeros(l * 128 + 1, dtype=torch.float32, device=sh.device)
        cache = self.lane_feat_cache
        if sh.rank == 0 and cache is not None and cache.shape[0] == l:
            buf[:l * 128] = cache.reshape(-1).to(sh.device)
            buf[-1] = 1.0
        sh.broadcast(buf, 0)
        if sh.rank != 0 and float(buf[-1]) == 1.0:
            dev = getattr(getattr(self.network, "rt", None), "device", None)
            self.lane_feat_cache = buf[:l * 128].view(l, 128).to(dev if dev is not None else buf.device).contiguous()

    # ------------------------------------------------------------------------------------------
    def collate(self, scenes):
        """Batch dict in the layout ScenePredNet.pre_process reads (mind/utils.py:142-168)."""
        a_counts = [len(s["ACTORS"]) for s in scenes]
        l_counts = [s["LANES"].shape[0] for s in scenes]
        a_off = np.concatenate([[0], np.cumsum(a_counts)])
        l_off = np.concatenate([[0], np.cumsum(l_counts)])
        data = {
            "BATCH_SIZE": len(scenes),
            "ACTORS": torch.from_numpy(np.concatenate([s["ACTORS"] for s in scenes])),
            "ACTOR_IDCS": [torch.arange(a_off[i], a_off[i + 1]) for i in range(len(scenes))],
            "LANES": torch.from_numpy(np.concatenate([s["LANES"] for s in scenes])),
            "LANE_IDCS": [torch.arange(l_off[i], l_off[i + 1]) for i in range(len(scenes))],
            "TGT_NODES": torch.from_numpy(np.stack([s["TGT_NODES"] for s in scenes])),
            "TGT_RPE": torch.from_numpy(np.stack([s["TGT_RPE"] for s in scenes]).reshape(len(scenes), -1)),
            "ACTOR_CTRS": torch.from_numpy(np.concatenate([s["TRAJS_CTRS"] for s in scenes])),
            "ACTOR_VECS": torch.from_numpy(np.concatenate([s["TRAJS_VECS"] for s in scenes])),
            "LANE_CTRS": torch.from_numpy(np.concatenate([s["LANE_CTRS"] for s in scenes])),
            "LANE_VECS": torch.from_numpy(np.concatenate([s["LANE_VECS"] for s in scenes])),
            "LANE_SHARED": all(s["LANES"] is scenes[0]["LANES"] for s in scenes),
            "LANE_FEAT_CACHE": self.lane_feat_cache,
        }
        if not getattr(self.network, "computes_rpe_in_kernel", False):
            data["RPE"] = [{"scene": torch.from_numpy(U.get_rpe(np.concatenate([s["TRAJS_CTRS"], s["LANE_CTRS"]]),
                                                                np.concatenate([s["TRAJS_VECS"], s["LANE_VECS"]]))),
                            "scene_mask": None} for s in scenes]
        return data

    def _device_batch(self, scenes):
        """If the scenes are a contiguous run of one mind_aime_rebase batch, the predictor input dict built from
        its device tensors (what ScenePredNet.pre_process would return); else None."""
        s0 = scenes[0]
        if not isinstance(s0, DevScene) or self.lane_feat_cache is None:
            return None
        dev, g0 = s0.dev, s0.g
        if any((not isinstance(sc, DevScene)) or sc.dev is not dev or sc.g != g0 + i for i, sc in enumerate(scenes)):
            return None
        B, a, l = len(scenes), dev["a"], dev["l"]
        if self.lane_feat_cache.shape[0] != l:
            return None
        cache = self.lane_feat_cache
        return {"actors": dev["actors"][g0 * a:(g0 + B) * a], "a_off": [a * i for i in range(B + 1)],
                "l_off": [l * i for i in range(B + 1)], "tgt_nodes": dev["tgt_nodes"][g0:g0 + B], "tgt_rpe": dev["tgt_rpe"][g0:g0 + B],
                "lanes": None, "lane_feat": cache.repeat(B, 1) if B > 1 else cache, "rpe": None, "lane_shared": True,
                "actor_ctrs": dev["actor_ctrs"][g0 * a:(g0 + B) * a], "actor_vecs": dev["actor_vecs"][g0 * a:(g0 + B) * a],
                "lane_ctrs": dev["lane_ctrs"][g0 * l:(g0 + B) * l], "lane_vecs": dev["lane_vecs"][g0 * l:(g0 + B) * l]}

    def predict_inputs(self, scenes):
        """The predictor input dict of a round's scenes (device tensors + offsets, what ScenePredNet.__call__ takes) and
        the collated host batch it came from (None when the inputs were built on the device by mind_aime_rebase)."""
        d = self._device_batch(scenes) if self.device_glue else None
        if d is not None:          # inputs already on the device: no collate, no host->device copies
            return d, None
        data = self.collate(scenes)
        return self.network.pre_process(data), data

    def predict_done(self, scenes, data, lane_feat):
        """bookkeeping after a predictor call: expansion count, LaneNet output of a first round kept for the plan"""
        self.n_expanded += len(scenes)
        if data is not None and lane_feat is not None and data["LANE_SHARED"]:
            self.lane_feat_cache = lane_feat

    def predict_scenes(self, scenes):
        d, data = self.predict_inputs(scenes)
        out = self.network(d)
        self.predict_done(scenes, data, getattr(self.network, "last_lane_feat", None) if data is not None else None)
        return out

    def branch_aime_rounds(self, lcl_smp, agent_obs):
        """branch_aime as a generator over its rounds: yields (scenes, d) -- the branch set and its predictor inputs -- and
        is sent (out, packed, lane_feat) = the network outputs for exactly these scenes; returns the scenario trees.  Lets a
        caller merge the rounds of SEVERAL scenes' trees into one predictor batch (mind_amd.fused, BASELINE config 3);
        driven alone it computes what branch_aime computes."""
        root = self.process_data(lcl_smp, agent_obs)
        self.prepare_root_data(root)
        self.tree.add_node(Node("root", None, ScenarioData(None, root, branch_flag=True)))
        batch = [root]
        while batch:
            d, data = self.predict_inputs(batch)
            out, packed, lane_feat = yield batch, d
            self.predict_done(batch, data, lane_feat)
            hdr, rows = self.prune_select(batch, out, 0, packed=packed)
            self.create_nodes(self.assemble_children(batch, hdr, _np(rows), rows if getattr(self, "_rows_persistent", False) else None))
            self.decide_branch()
            batch = [n.data.obs_data for n in self.get_branch_set()]
        assert len(self.get_end_set()) > 0, "No end node found in the scenario tree."
        return self.get_scenario_tree()

    def create_nodes(self, pred_bar):
        for pred in pred_bar:
            self.tree.add_node(Node(pred["SCEN_ID"], pred["PARENT_ID"], ScenarioData(pred, None)))

    def decide_branch(self):
        """scenario_tree.py:82-100; the per-leaf work (branch time, re-basing of the observation) is done for all
        candidate leaves of the round at once."""
        cand = []
        for l in self.tree.get_leaf_nodes():
            d = l.data
            if d.branch_flag:
                d.branch_flag = False
                d.terminate_flag = True
            elif not d.end_flag:
                if l.depth >= self.config.max_depth:
                    d.terminate_flag = True
                else:
                    cand.append(l)
        if not cand:
            return
        tbs = self.get_branch_times([l.data.data for l in cand])
        todo = []
        for l, t_b in zip(cand, tbs):
            if t_b < self.pred_len:
                todo.append(l)
            else:
                l.data.end_flag = True
        if todo:
            own = None
            if self.shard is not None and self.shard.sharded:
                own = self.shard.block(len(todo))       # the rank that expands a scene next round is the one that re-bases it
            for l, (obs, cur) in zip(todo, self.update_obser_batch([l.data.data for l in todo], own)):
                l.data.obs_data, l.data.data = obs, cur
                l.data.branch_flag = True

    def get_branch_set(self):
        out = [l for l in self.tree.get_leaf_nodes() if l.data.branch_flag]
        self.branch_depth += 1
        return out

    def get_end_set(self):
        return [n for n in self.tree.get_leaf_nodes() if n.data.end_flag]

    # ------------------------------------------------------------------------------------------
    def process_data(self, lcl_smp, agent_obs):
        pos, ang, vel, types, flags, tids, cats = U.get_agent_trajectories(agent_obs)
        cur_vel = lcl_smp.ego_agent.state[2]
        orig, rot, theta, pos_n, ang_n, vel_n, ctrs, vecs = U.normalize_agents(pos, ang, vel)
        lane_graph = U.lane_graph_from_map(lcl_smp.map_data, orig, rot)
        self.lane_graph = lane_graph
        self.lane_feat_in = U.lane_features(lane_graph)
        self.n_lanes = int(self.lane_feat_in.shape[0])
        s = self._scene_inputs(orig, rot, pos_n, ang_n, vel_n, types, flags, ctrs, vecs, cur_vel,
                               lane_graph["lane_ctrs"], lane_graph["lane_vecs"])
        s["TRAJS_TID"], s["TRAJS_CAT"] = tids, cats
        return s

    def _scene_inputs(self, orig, rot, pos_n, ang_n, vel_n, types, pad, ctrs, vecs, cur_vel, lane_ctrs, lane_vecs):
        tgt_pts, tgt_nodes, (tgt_ctr, tgt_vec) = self.get_high_level_command(orig, rot, cur_vel)
        tgt_rpe = U.get_rpe(np.stack([tgt_ctr, ctrs[0]]), np.stack([tgt_vec, vecs[0]]))
        return {"ORIG": orig, "ROT": rot, "TRAJS_POS_OBS": pos_n, "TRAJS_ANG_OBS": ang_n, "TRAJS_VEL_OBS": vel_n,
                "TRAJS_TYPE": types, "PAD_OBS": pad, "TRAJS_CTRS": ctrs, "TRAJS_VECS": vecs,
                "ACTORS": U.actor_features(pos_n, ang_n, vel_n, types, pad), "LANES": self.lane_feat_in,
                "LANE_CTRS": np.ascontiguousarray(lane_ctrs, F32), "LANE_VECS": np.ascontiguousarray(lane_vecs, F32),
                "TGT_PTS": tgt_pts, "TGT_NODES": tgt_nodes, "TGT_RPE": tgt_rpe.reshape(-1)}

    def get_high_level_command(self, orig, rot, cur_vel, min_vel=0.5):
        lane = self.target_lane
        d = lane - orig
        closest = int(np.argmin(np.sqrt((d * d).sum(-1))))
        # python float - float32 tensor -> float32 arithmetic from the first subtraction on
        travel = max(cur_vel, min_vel) * self.config.tar_time_ahead
        idx = closest
        while idx < len(lane) - 1 and travel > 0:
            idx += 1
            step = lane[idx] - lane[idx - 1]
            travel = F32(travel) - np.sqrt((step * step).sum(), dtype=F32)
        if idx == len(lane) - 1:
            idx -= 1
        idx = max(5, min(idx, len(lane) - 6))
        sel = np.arange(idx - 5, idx + 6)
        pts = lane[sel]
        info = self.target_lane_info[sel][1:]
        assert len(pts) == 11
        ctrln = np.matmul(pts - orig, rot)
        anch_pos = ctrln.mean(axis=0, dtype=F32)
        dv = ctrln[-1] - ctrln[0]
        anch_vec = dv / np.sqrt((dv * dv).sum(), dtype=F32)
        anch_rot = np.array([[anch_vec[0], -anch_vec[1]], [anch_vec[1], anch_vec[0]]], dtype=F32)
        ctrln = np.matmul(ctrln - anch_pos, anch_rot)
        ctrs = (ctrln[:-1] + ctrln[1:]) / F32(2.0)
        vecs = ctrln[1:] - ctrln[:-1]
        tgt_nodes = np.concatenate([ctrs, vecs, info], axis=-1).astype(F32)
        return pts.copy(), tgt_nodes, (anch_pos.astype(F32), anch_vec.astype(F32))

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def _to_world(local, ctrs, vecs, rot, orig, translate=True):
        """((x R_i^T) + c_i) R^T + o  with R_i = rot(atan2(vec_i)) (scenario_tree.py:328-339), batched."""
        th = np.arctan2(vecs[:, 1], vecs[:, 0])
        c, s = np.cos(th), np.sin(th)
        Ri = np.stack([np.stack([c, -s], -1), np.stack([s, c], -1)], -2).astype(F32)       # [a,2,2]
        out = np.matmul(local, np.transpose(Ri, (0, 2, 1)))
        if translate:
            out = out + ctrs[:, None, :]
        out = np.matmul(out, rot.T)
        if translate:
            out = out + orig
        return out.astype(F32), th

    def prepare_root_data(self, root):
        rot, orig = root["ROT"], root["ORIG"]
        theta_g = np.arctan2(rot[1, 0], rot[0, 0])
        pos_w, th = self._to_world(root["TRAJS_POS_OBS"], root["TRAJS_CTRS"], root["TRAJS_VECS"], rot, orig)
        vel_w, _ = self._to_world(root["TRAJS_VEL_OBS"], root["TRAJS_CTRS"], root["TRAJS_VECS"], rot, orig, False)
        root["TRAJS_POS_HIST"] = pos_w
        root["TRAJS_VEL_HIST"] = vel_w
        root["TRAJS_ANG_HIST"] = (root["TRAJS_ANG_OBS"] + th[:, None] + theta_g).astype(F32)
        a, T = pos_w.shape[:2]
        root["TRAJS_COV_HIST"] = np.full((a, T, 1), 1e-5, F32)
        root["SCEN_PROB"] = 1.0
        root["SCEN_ID"] = "root"
        root["PARENT_ID"] = None
        root["CUR_T"] = 0
        root["END_T"] = self.pred_len
        return root

    def _select_modes(self, sc, cls_row, topo_all, ego_end, dis_all=None, ego_cov=None):
        """Pruning decisions of one scene (scenario_tree.py:293-327, 361-395): probability floor, distance of the
        ego end point to the target lane, greedy merge of modes with the same topology.  cls_row [K] f32,
        topo_all [a-1,K], ego_end(k) -> (mean [2], cov [1]) (or ego_cov [K] with dis_all [K]: every mode's end-point sigma
        and lane distance at once).  -> [(k, prob)] in visiting order.  Same float32 arithmetic per element as the
        reference's per-mode loop; the per-pair signature comparisons of the greedy merge are taken in one array op."""
        K = len(cls_row)
        order = np.argsort(-cls_row, kind="stable")
        probs = np.asarray(cls_row * sc["SCEN_PROB"]).astype(F32, copy=False)
        keep = ~(probs < F32(0.001))
        if self.target_lane is not None and self.ego_idx is not None:
            # ego end point of every mode against the target lane, all modes at once
            if ego_cov is None or dis_all is None:
                ends = [ego_end(int(k)) for k in range(K)]
                if dis_all is None:
                    dis_all = U.get_distances_to_polyline(self.target_lane, np.stack([e[0] for e in ends]).astype(F32, copy=False))
                ego_cov = np.stack([np.asarray(e[1], F32).reshape(-1) for e in ends])          # [K, 1]
            else:
                ego_cov = np.asarray(ego_cov, F32).reshape(K, -1)
            keep &= ~((np.asarray(dis_all)[:, None] - ego_cov) > self.config.tar_dist_thres).any(axis=1)
        ks = [int(k) for k in order if keep[k]]
        if not ks:
            return []
        if len(ks) == 1:
            return [(ks[0], probs[ks[0]])]
        # greedy merge of modes whose signatures differ by <= pi/6 for every exo agent: differ[i][j] for candidate i before j
        sig = topo_all[:, ks] if len(topo_all) else np.zeros((0, len(ks)), F32)
        diff = sig[:, :, None] - sig[:, None, :]
        diff = np.arctan2(np.sin(diff), np.cos(diff))
        differ = (((np.abs(diff) - F32(np.pi / 6)) > 0).sum(axis=0) > 0).tolist()
        selected, alive = [], list(range(len(ks)))
        while alive:
            i = alive[0]
            selected.append((ks[i], probs[ks[i]]))
            row = differ[i]
            alive = [j for j in alive[1:] if row[j]]
        return selected

    def _hdr(self, picks, scenes, idx_offset):
        hdr = np.zeros((len(picks), self.HDR), F32)
        for r, (lidx, k, prob) in enumerate(picks):
            hdr[r, 0], hdr[r, 1], hdr[r, 2] = lidx + idx_offset, k, prob
            hdr[r, 3:] = np.asarray(scenes[lidx]["TGT_PTS"], F32).reshape(-1)
        return hdr

    def _select_round_host(self, scenes, w, a_off, lasts, B, A, L):
        """the pruning decisions of a round on the host, from the device's signatures / end points (one small copy)"""
        small = w["small"].cpu().numpy()            # [cls | topology signatures | ego end points], one copy
        cls_all = small[:B * 6].reshape(B, 6)
        topo = small[B * 6:B * 6 + A * 6].reshape(A, 6)
        ego_all = small[B * 6 + A * 6:].reshape(B, 6, 4)
        picks = []
        dis_round = None
        if self.target_lane is not None and self.ego_idx is not None and all(l >= 0 for l in lasts):
            # ego end points of every (scene, mode) of the round against the target lane in one call
            dis_round = ego_all[:, :, 3]        # distance of every ego end point to the target lane, from the kernel
        for lidx, sc in enumerate(scenes):
            def ego_end(k, lidx=lidx, sc=sc):
                if lasts[lidx] >= 0:
                    return ego_all[lidx, k, :2], ego_all[lidx, k, 2:3]
                return sc["TRAJS_POS_HIST"][self.ego_idx][L - 1], sc["TRAJS_COV_HIST"][self.ego_idx][L - 1]
            for k, prob in self._select_modes(sc, cls_all[lidx], topo[a_off[lidx] + 1:a_off[lidx + 1]], ego_end,
                                              None if dis_round is None else dis_round[lidx],
                                              None if dis_round is None else ego_all[lidx, :, 2]):
                picks.append((lidx, k, prob))
        return picks

    def _prune_select_device(self, scenes, packed, idx_offset):
        """Pruning decisions with the per-(agent, mode, step) arithmetic on the MI355X (mind_aime_world): the predictor
        outputs never leave the device; the host reads cls / topology signatures / ego end points (one small copy) and
        decides; the world-frame rows of the SURVIVING modes are gathered on the device (``rows`` stays there: the caller
        copies it to the host once, after the multi-GPU exchange if there is one)."""
        rt, a_off = packed["rt"], packed["a_off"]
        B, A, L = len(scenes), int(packed["a_off"][-1]), self.seq_len
        lasts = [L - 1 - hist_len(sc) for sc in scenes]
        lane_check = self.target_lane is not None and self.ego_idx is not None
        # the decisions themselves on the device too, unless an ego end point lies in the history (then the host decides)
        dev_select = self.device_select and (not lane_check or all(l >= 0 for l in lasts))
        w = rt.aime_world(packed["reg"], packed["vel"], packed["actor_ctrs"], packed["actor_vecs"], a_off,
                          np.stack([sc["ROT"] for sc in scenes]), np.stack([sc["ORIG"] for sc in scenes]),
                          np.concatenate([cov_last_of(sc) for sc in scenes]), [max(l, -1) for l in lasts],
                          target_lane=self.target_lane, cls=packed["cls"],
                          scen_prob=[sc["SCEN_PROB"] for sc in scenes] if dev_select else None,
                          dist_thres=self.config.tar_dist_thres if (dev_select and lane_check) else None)
        picks = []
        if dev_select:
            sel = w["sel"].cpu().numpy()                # [2, B, 6]: kept modes in visiting order (-1 = none), path probabilities
            li, ji = np.nonzero(sel[0] >= 0)            # row-major = scene by scene, visiting order within a scene
            picks = list(zip(li.tolist(), sel[0][li, ji].astype(np.int64).tolist(), sel[1][li, ji]))
        else:
            picks = self._select_round_host(scenes, w, a_off, lasts, B, A, L)
        dev = w["world"].device
        if not picks:
            return np.zeros((0, self.HDR), F32), torch.zeros(0, 60, 6, device=dev)
        flat = np.concatenate([np.arange(a_off[l], a_off[l + 1]) * 6 + k for l, k, _ in picks])       # (agent row, mode) -> row of [A*6]
        idx = torch.from_numpy(flat).to(dev)
        # the children only keep a handle on the device rows when those sit in the persistent ping-pong buffer (the next re-basing then
        # cuts their windows out of it): a fresh index_select block held by the whole tree until reset() pinned tens of MB per round
        self._rows_persistent = self._use_device_windows(rt, len(picks))
        if self._rows_persistent:
            # the kept rows stay on the device until the next re-basing cuts the children's windows out of them: they go into one
            # of two persistent buffers of the runtime (alternating by round) instead of a fresh allocator block -- holding
            # allocator blocks across rounds pushed the caching allocator into its slow path (hipMalloc) every plan
            bufs = rt.__dict__.setdefault("_rows_bufs", [None, None])
            k = rt.__dict__["_rows_turn"] = 1 - rt.__dict__.get("_rows_turn", 0)
            if bufs[k] is None or bufs[k].shape[0] < len(flat):
                bufs[k] = torch.empty(max(len(flat), 2 * (0 if bufs[k] is None else bufs[k].shape[0])), 60, 6, device=dev)
            rows = torch.index_select(w["world"].view(A * 6, 60, 6), 0, idx, out=bufs[k][:len(flat)])
        else:
            rows = w["world"].view(A * 6, 60, 6).index_select(0, idx)           # [R,60,6] (x,y,vx,vy,heading,max-sigma)
        return self._hdr(picks, scenes, idx_offset), rows

    def prune_select(self, scenes, out, idx_offset=0, packed=None):
        """prune_merge, first half (scenario_tree.py:281-395): world-frame modes, probability / target-lane pruning, greedy
        topology merge.  -> (hdr [P,25] float32: scene index in the round's full batch, mode, path probability, the
        scene's target window; rows [sum a,60,6]: (x, y, vx, vy, heading, max-sigma) of every agent of every kept mode,
        a device tensor when the predictor outputs live on the device)."""
        self._rows_persistent = False
        if not scenes:
            return np.zeros((0, self.HDR), F32), torch.zeros(0, 60, 6)
        res_cls_b, res_reg_b, res_aux_b = out
        if packed is None:
            packed = getattr(self.network, "last_packed", None)
        if (self.device_glue and packed is not None and packed["n"] == len(scenes) and packed.get("rt") is not None
                and packed.get("actor_ctrs") is not None and self.ego_idx == 0):
            return self._prune_select_device(scenes, packed, idx_offset)
        if packed is not None and packed["n"] == len(scenes):
            # one device->host copy per tensor for the whole round instead of three per scene
            cls_all, reg_all, vel_all = (_np(packed[k]) for k in ("cls", "reg", "vel"))
            off = packed["a_off"]
            res_cls_b = [cls_all[i:i + 1] for i in range(len(scenes))]
            res_reg_b = [reg_all[off[i]:off[i + 1]] for i in range(len(scenes))]
            res_aux_b = [(vel_all[off[i]:off[i + 1]], None, None) for i in range(len(scenes))]
        picks, rows = [], []
        for lidx, sc in enumerate(scenes):
            rot, orig = sc["ROT"], sc["ORIG"]
            ctrs, vecs = sc["TRAJS_CTRS"], sc["TRAJS_VECS"]
            theta_g = np.arctan2(rot[1, 0], rot[0, 0])
            reg = _np(res_reg_b[lidx]).astype(F32, copy=False)      # [a,6,60,5]
            cls = _np(res_cls_b[lidx]).astype(F32, copy=False)      # [1,6]
            vel = _np(res_aux_b[lidx][0]).astype(F32, copy=False)   # [a,6,60,2]
            ang_loc = U.get_angle(vel)                               # from the un-rotated velocity
            a_, K_, T_ = reg.shape[:3]
            # all K modes at once: [a, K*T, 2] through the same per-agent / global rotations
            pos_all, th = self._to_world(reg[..., :2].reshape(a_, K_ * T_, 2), ctrs, vecs, rot, orig)
            vel_all, _ = self._to_world(vel.reshape(a_, K_ * T_, 2), ctrs, vecs, rot, orig, False)
            pos_all, vel_all = pos_all.reshape(a_, K_, T_, 2), vel_all.reshape(a_, K_, T_, 2)
            ang_all = (ang_loc + th[:, None, None] + theta_g).astype(F32)
            cov_all = U.get_max_covariance(reg[..., 2:]) + sc["TRAJS_COV_HIST"][:, -1][:, None, None]
            # topology signatures of all modes at once: cumulative winding of (exo - ego)
            rel = pos_all[1:] - pos_all[0:1]
            rel = rel / np.sqrt((rel * rel).sum(-1, keepdims=True))
            phi = np.arctan2(rel[..., 1], rel[..., 0])                       # [a-1, K, T]
            dphi = phi[..., 1:] - phi[..., :-1]
            dphi = np.arctan2(np.sin(dphi), np.cos(dphi))
            topo_all = dphi.sum(axis=-1, dtype=F32)                          # [a-1, K]
            L = self.seq_len
            n_hist = sc["TRAJS_POS_HIST"].shape[1]
            last = L - 1 - n_hist                                            # index of the last kept predicted step
            def ego_end(k):
                if last >= 0:
                    return pos_all[self.ego_idx, k, last], cov_all[self.ego_idx, k, last]
                return sc["TRAJS_POS_HIST"][self.ego_idx][L - 1], sc["TRAJS_COV_HIST"][self.ego_idx][L - 1]
            for k, prob in self._select_modes(sc, cls[0], topo_all, ego_end):
                picks.append((lidx, k, prob))
                rows.append(np.concatenate([pos_all[:, k], vel_all[:, k], ang_all[:, k][..., None],
                                            cov_all[:, k].astype(F32).reshape(a_, T_, 1)], axis=-1).astype(F32, copy=False))
        if not picks:
            return np.zeros((0, self.HDR), F32), torch.zeros(0, 60, 6)
        return self._hdr(picks, scenes, idx_offset), torch.from_numpy(np.ascontiguousarray(np.concatenate(rows)))

    def assemble_children(self, batch, hdr, rows, rows_dev=None):
        """prune_merge, second half (scenario_tree.py:396-412): the child dict of every kept mode -- the parent's world-frame
        history extended by the mode's 60 steps and truncated to seq_len (Q8).  ``batch`` is the round's full branch set
        (every rank holds its world-frame fields), ``hdr`` / ``rows`` as returned by prune_select, concatenated in batch order
        over all ranks when the round was sharded."""
        kept, r0, L = [], 0, self.seq_len
        shared = {}
        for h in hdr:
            gidx, k = int(h[0]), int(h[1])
            sc = batch[gidx]
            a = sc["TRAJS_TYPE"].shape[0]
            m = rows[r0:r0 + a]
            row_first = r0
            r0 += a
            kept.append(ChildScene({
                "SCEN_PROB": F32(h[2]), "CUR_T": sc["CUR_T"], "END_T": sc["END_T"],
                "PARENT_ID": sc["SCEN_ID"], "SCEN_ID": "{}_{}_{}".format(self.branch_depth, gidx, k),
                "TRAJS_TYPE": sc["TRAJS_TYPE"], "TRAJS_TID": sc["TRAJS_TID"], "TRAJS_CAT": sc["TRAJS_CAT"],
                "TGT_PTS": np.array(h[3:], F32).reshape(11, 2),
            }, sc, m, L, shared.setdefault(gidx, {})))
            # where the child's rows live on the device (the re-basing of the next round can cut its window out of them there)
            kept[-1].dev_rows, kept[-1].row0 = rows_dev, row_first
        assert r0 == len(rows), (r0, len(rows))
        return kept

    def prune_merge(self, scenes, out):
        """scenario_tree.py:281-412 in one call (single process)."""
        hdr, rows = self.prune_select(scenes, out, 0)
        return self.assemble_children(scenes, hdr, _np(rows))

    def get_branch_time(self, d):
        cur_t, end_t = d["CUR_T"], d["END_T"]
        compare_t = self.obs_len + cur_t + (1 if cur_t == 0 else 0)
        ts = np.arange(cur_t + 1 + (cur_t + 1) % 2, end_t, 2)          # even t in (cur_t, end_t)
        if len(ts):
            if isinstance(d, ChildScene) and d.n_hist == self.obs_len and self.obs_len + int(ts[-1]) < d.length and compare_t < d.length:
                sig = d.new[:, :, 5]                                   # the predicted steps' max-sigma: rows obs_len + t
                hit = (sig[:, ts] / sig[:, compare_t - self.obs_len][:, None] > 9).any(axis=0)
            else:
                cov = d["TRAJS_COV_HIST"]
                hit = (cov[:, self.obs_len + ts, 0] / cov[:, compare_t, 0][:, None] > 9).any(axis=0)
            if hit.any():
                t = int(ts[int(np.argmax(hit))])
                d["END_T"] = t
                return t
        return end_t

    def get_branch_times(self, datas):
        """get_branch_time for several nodes; nodes with equally shaped covariance histories share one array pass."""
        out = [None] * len(datas)
        groups = {}
        for i, d in enumerate(datas):
            lazy = isinstance(d, ChildScene) and d.n_hist == self.obs_len
            groups.setdefault((d.new.shape[0], d.length, True) if lazy else d["TRAJS_COV_HIST"].shape, []).append(i)
        for shape, idx in groups.items():
            T = shape[1]
            if len(idx) == 1 or T <= self.obs_len:
                for i in idx:
                    out[i] = self.get_branch_time(datas[i])
                continue
            if shape[2] is True:     # only the predicted steps are read (rows >= obs_len): no concatenated history needed
                cov = np.empty((len(idx), shape[0], T), F32)
                cov[:, :, :self.obs_len] = 1.0
                cov[:, :, self.obs_len:] = np.stack([datas[i].new[:, :T - self.obs_len, 5] for i in idx])
            else:
                cov = np.stack([datas[i]["TRAJS_COV_HIST"][:, :, 0] for i in idx])          # [G,a,T]
            cur = np.array([datas[i]["CUR_T"] for i in idx])
            end = np.array([datas[i]["END_T"] for i in idx])
            cmp_t = self.obs_len + cur + (cur == 0)
            if (end > T - self.obs_len).any() or (cmp_t >= T).any():
                for i in idx:                                   # would index past the history: per-node semantics
                    out[i] = self.get_branch_time(datas[i])
                continue
            ts = np.arange(0, T - self.obs_len)
            ratio = cov[:, :, self.obs_len:] / np.take_along_axis(cov, cmp_t[:, None, None], axis=2)
            hit = (ratio > 9).any(axis=1)                                                   # [G, T-obs]
            hit &= (ts[None, :] > cur[:, None]) & (ts[None, :] < end[:, None]) & (ts[None, :] % 2 == 0)
            anyhit = hit.any(axis=1)
            first = np.argmax(hit, axis=1)
            for g, i in enumerate(idx):
                if anyhit[g]:
                    datas[i]["END_T"] = int(first[g])
                    out[i] = int(first[g])
                else:
                    out[i] = datas[i]["END_T"]
        return out

    def high_level_command_batch(self, origs, rots, cur_vels, min_vel=0.5):
        """get_high_level_command for G frames at once (same float32 arithmetic per frame): the running ``travel``
        subtraction of the reference's while-loop is a float32 cumulative sum of [travel0, -seg, -seg, ...]."""
        lane = self.target_lane
        n, G = len(lane), len(origs)
        d = lane[None] - origs[:, None, :]
        closest = np.argmin(np.sqrt((d * d).sum(-1)), axis=1)
        seg = self._lane_seg_f32()                                                          # [n-1] |lane[i+1]-lane[i]|
        travel0 = np.array([max(v, min_vel) * self.config.tar_time_ahead for v in cur_vels]).astype(F32)
        steps = np.zeros((G, n), F32)
        steps[:, 0] = travel0
        j = np.arange(1, n)
        src = closest[:, None] + j[None, :] - 1                                             # seg index used at step j
        ok = src < n - 1
        steps[:, 1:] = np.where(ok, -seg[np.minimum(src, n - 2)], F32(0))
        trav = np.cumsum(steps, axis=1, dtype=F32)                                          # trav[:, j] after j steps
        done = (trav[:, 1:] <= 0) & ok                                                      # loop stops after this step
        nstep = np.where(done.any(axis=1), np.argmax(done, axis=1) + 1, (n - 1) - closest)
        idx = closest + nstep
        idx = np.where(idx == n - 1, idx - 1, idx)
        idx = np.maximum(5, np.minimum(idx, n - 6))
        sel = idx[:, None] + np.arange(-5, 6)[None, :]                                      # [G,11]
        pts = lane[sel]                                                                     # [G,11,2]
        info = self.target_lane_info[sel][:, 1:]                                            # [G,10,12]
        ctrln = np.matmul(pts - origs[:, None, :], rots)
        anch_pos = ctrln.mean(axis=1, dtype=F32)
        dv = ctrln[:, -1] - ctrln[:, 0]
        anch_vec = dv / np.sqrt((dv * dv).sum(-1), dtype=F32)[:, None]
        anch_rot = np.stack([np.stack([anch_vec[:, 0], -anch_vec[:, 1]], -1), np.stack([anch_vec[:, 1], anch_vec[:, 0]], -1)], -2).astype(F32)
        ctrln = np.matmul(ctrln - anch_pos[:, None, :], anch_rot)
        ctrs = (ctrln[:, :-1] + ctrln[:, 1:]) / F32(2.0)
        vecs = ctrln[:, 1:] - ctrln[:, :-1]
        tgt_nodes = np.concatenate([ctrs, vecs, info], axis=-1).astype(F32)
        return pts.copy(), tgt_nodes, anch_pos.astype(F32), anch_vec.astype(F32)

    def _lane_seg_f32(self):
        lane = self.target_lane
        hit = getattr(self, "_seg_cache", None)
        if hit is None or hit[0] is not lane:
            dl = lane[1:] - lane[:-1]
            self._seg_cache = (lane, np.sqrt((dl * dl).sum(-1), dtype=F32))
        return self._seg_cache[1]

    def update_obser_batch(self, curs, own=None):
        """update_obser (scenario_tree.py:467-567) for all branching nodes of a round at once; nodes must share the
        agent set (they are children of one plan).  Returns [(obs_data, cur)] in order.  ``own`` = [lo, hi): only these
        nodes are re-based here (sharded rounds); the others get a RemoteScene with the world-frame fields."""
        if own is not None:
            lo, hi = own
            o = self.obs_len
            out = [None] * len(curs)
            for g, c in enumerate(curs):
                if lo <= g < hi:
                    continue
                keep = o + (c["END_T"] - c["CUR_T"])
                hist_trim(c, keep)
                w0 = max(min(keep, hist_len(c)) - o, 0)
                out[g] = (RemoteScene({"TRAJS_TYPE": c["TRAJS_TYPE"], "SCEN_PROB": c["SCEN_PROB"], "SCEN_ID": c["SCEN_ID"],
                                       "PARENT_ID": c["PARENT_ID"], "CUR_T": c["END_T"], "END_T": self.pred_len,
                                       "TRAJS_TID": c["TRAJS_TID"], "TRAJS_CAT": c["TRAJS_CAT"],
                                       **{k: np.ascontiguousarray(hist_rows(c, k, w0, keep)) for k in _HIST_COLS}}), c)
            if hi > lo:
                out[lo:hi] = self.update_obser_batch(curs[lo:hi])
            return out
        a_counts = {(c.new.shape[0] if isinstance(c, ChildScene) else c["TRAJS_POS_HIST"].shape[0]) for c in curs}
        rt = getattr(self.network, "rt", None) if self.device_glue else None
        on_dev = (rt is not None and len(a_counts) == 1 and self.target_lane is not None and len(self.target_lane) >= 12
                  and all(c["TRAJS_TYPE"] is curs[0]["TRAJS_TYPE"] or np.array_equal(c["TRAJS_TYPE"], curs[0]["TRAJS_TYPE"]) for c in curs))
        if not on_dev and (len(curs) < 2 or len(a_counts) != 1):
            return [self.update_obser(c) for c in curs]
        o = self.obs_len
        G = len(curs)
        if on_dev and self._use_device_windows(rt, len(curs)):
            dev_out = self._update_obser_device_windows(curs, rt)
            if dev_out is not None:
                return dev_out
        wins = {}
        for c in curs:
            keep = o + (c["END_T"] - c["CUR_T"])
            hist_trim(c, keep)
            w0 = max(min(keep, hist_len(c)) - o, 0)
            if isinstance(c, ChildScene) and keep <= c.length and keep > c.n_hist:
                w6 = c.window6(w0, keep)
                for k, col in _HIST_COLS.items():
                    wins.setdefault(k, []).append(w6[:, :, col])
                wins.setdefault("_WIN6", []).append(w6)
            else:
                for k in _HIST_COLS:
                    wins.setdefault(k, []).append(hist_rows(c, k, w0, keep))
                wins.setdefault("_WIN6", []).append(None)
        if on_dev:
            # the whole re-basing arithmetic on the device; the next round's predictor reads its outputs in place.  The
            # per-scene windows are stacked straight into the runtime's page-locked staging buffers (no stacked host copy)
            wp, wc, wa, wv = (wins[k] for k in ("TRAJS_POS_HIST", "TRAJS_COV_HIST", "TRAJS_ANG_HIST", "TRAJS_VEL_HIST"))
            dev = rt.aime_rebase(wp, wa, wv, curs[0]["TRAJS_TYPE"], self.lane_graph["lane_ctrs"], self.lane_graph["lane_vecs"],
                                 self.target_lane, self.target_lane_info, time_ahead=self.config.tar_time_ahead)
            dev["a"], dev["l"] = wp[0].shape[0], self.lane_graph["lane_ctrs"].shape[0]
            fr = dev["frames"].cpu().numpy()
            out = []
            for g, c in enumerate(curs):
                out.append((DevScene(self, dev, g, {
                    "ORIG": fr[g, 4:6].copy(), "ROT": fr[g, :4].reshape(2, 2).copy(), "TGT_PTS": fr[g, 6:].reshape(11, 2).copy(),
                    "TRAJS_TYPE": c["TRAJS_TYPE"], "SCEN_PROB": c["SCEN_PROB"], "SCEN_ID": c["SCEN_ID"], "PARENT_ID": c["PARENT_ID"],
                    "CUR_T": c["END_T"], "END_T": self.pred_len, "TRAJS_TID": c["TRAJS_TID"], "TRAJS_CAT": c["TRAJS_CAT"],
                    "TRAJS_POS_HIST": wp[g], "TRAJS_COV_HIST": wc[g], "TRAJS_ANG_HIST": wa[g], "TRAJS_VEL_HIST": wv[g]}), c))
                out[-1][0]._win6 = wins["_WIN6"][g]
            return out
        pos, cov = np.stack(wins["TRAJS_POS_HIST"]), np.stack(wins["TRAJS_COV_HIST"])
        ang, vel = np.stack(wins["TRAJS_ANG_HIST"]), np.stack(wins["TRAJS_VEL_HIST"])
        orig, rot, theta, pos_n, ang_n, vel_n, ctrs, vecs = U.normalize_agents_batch(pos, ang, vel)
        lane_ctrs = np.ascontiguousarray(np.matmul(self.lane_graph["lane_ctrs"][None] - orig[:, None, :], rot), F32)
        lane_vecs = np.ascontiguousarray(np.matmul(self.lane_graph["lane_vecs"][None], rot), F32)
        cur_vel = np.sqrt((vel_n[:, 0, -1] * vel_n[:, 0, -1]).sum(-1), dtype=F32)
        pad = np.ones(ang.shape, F32)
        types = np.stack([c["TRAJS_TYPE"] for c in curs])
        tgt_pts, tgt_nodes, tgt_ctr, tgt_vec = self.high_level_command_batch(orig, rot, cur_vel)
        tgt_rpe = U.get_rpe_batch(np.stack([tgt_ctr, ctrs[:, 0]], 1), np.stack([tgt_vec, vecs[:, 0]], 1)).reshape(G, -1)
        actors = U.actor_features_batch(pos_n, ang_n, vel_n, types, pad)
        out = []
        for g, c in enumerate(curs):
            s = {"ORIG": orig[g], "ROT": rot[g], "TRAJS_POS_OBS": pos_n[g], "TRAJS_ANG_OBS": ang_n[g], "TRAJS_VEL_OBS": vel_n[g],
                 "TRAJS_TYPE": c["TRAJS_TYPE"], "PAD_OBS": pad[g], "TRAJS_CTRS": ctrs[g], "TRAJS_VECS": vecs[g],
                 "ACTORS": actors[g], "LANES": self.lane_feat_in, "LANE_CTRS": lane_ctrs[g], "LANE_VECS": lane_vecs[g],
                 "TGT_PTS": tgt_pts[g], "TGT_NODES": tgt_nodes[g], "TGT_RPE": tgt_rpe[g],
                 "SCEN_PROB": c["SCEN_PROB"], "SCEN_ID": c["SCEN_ID"], "PARENT_ID": c["PARENT_ID"],
                 "CUR_T": c["END_T"], "END_T": self.pred_len, "TRAJS_TID": c["TRAJS_TID"], "TRAJS_CAT": c["TRAJS_CAT"],
                 "TRAJS_POS_HIST": pos[g], "TRAJS_COV_HIST": cov[g], "TRAJS_ANG_HIST": ang[g], "TRAJS_VEL_HIST": vel[g]}
            out.append((s, c))
        return out

    def _update_obser_device_windows(self, curs, rt):
        """update_obser_batch with the 50-step windows assembled ON THE DEVICE (mind_aime_rebase, device source): possible when
        every node is a kept mode whose parent was re-based by the previous mind_aime_rebase call of this runtime (its window is
        still in that call's arena) and whose own rows are still on the device (the tensor prune_select gathered).  Nothing is
        stacked or uploaded; the host copies of the windows are only cut out if somebody asks (DevScene.lazy).  None = not
        applicable (the caller then takes the host-window path)."""
        o = self.obs_len
        c0 = curs[0]
        if not all(isinstance(c, ChildScene) and isinstance(c.parent, DevScene) and c.n_hist == o for c in curs):
            return None
        pd, rows_dev = c0.parent.dev, getattr(c0, "dev_rows", None)
        if rows_dev is None or pd.get("gen") is None or getattr(rt, "_rebase_gen", None) != pd["gen"]:
            return None
        durs = []
        for c in curs:
            dur = c["END_T"] - c["CUR_T"]
            if c.parent.dev is not pd or getattr(c, "dev_rows", None) is not rows_dev or not (1 <= dur <= 60) or o + dur > c.length:
                return None
            durs.append(int(dur))
        a = c0.new.shape[0]
        dev = rt.aime_rebase(None, None, None, c0["TRAJS_TYPE"], self.lane_graph["lane_ctrs"], self.lane_graph["lane_vecs"],
                             self.target_lane, self.target_lane_info, time_ahead=self.config.tar_time_ahead,
                             dev_src=dict(rows=rows_dev, parent_slot=[c.parent.g for c in curs], row0=[c.row0 for c in curs],
                                          dur=durs, gen=pd["gen"], a=a))
        dev["a"], dev["l"] = a, self.lane_graph["lane_ctrs"].shape[0]
        fr = dev["frames"].cpu().numpy()
        out = []
        for g, (c, dur) in enumerate(zip(curs, durs)):
            keep = o + dur
            hist_trim(c, keep)
            out.append((DevScene(self, dev, g, {
                "ORIG": fr[g, 4:6].copy(), "ROT": fr[g, :4].reshape(2, 2).copy(), "TGT_PTS": fr[g, 6:].reshape(11, 2).copy(),
                "TRAJS_TYPE": c["TRAJS_TYPE"], "SCEN_PROB": c["SCEN_PROB"], "SCEN_ID": c["SCEN_ID"], "PARENT_ID": c["PARENT_ID"],
                "CUR_T": c["END_T"], "END_T": self.pred_len, "TRAJS_TID": c["TRAJS_TID"], "TRAJS_CAT": c["TRAJS_CAT"]},
                lazy=(c, dur, keep), cov_last=np.ascontiguousarray(c.new[:, dur - 1, 5])), c))
        return out

    def _host_obs(self, pos, ang, vel, types):
        """Host restatement of the predictor inputs of one re-based scene (what update_obser computes), for consumers
        of a DevScene that want the arrays on the host."""
        orig, rot, theta, pos_n, ang_n, vel_n, ctrs, vecs = U.normalize_agents(pos, ang, vel)
        lane_ctrs = np.matmul(self.lane_graph["lane_ctrs"] - orig, rot)
        lane_vecs = np.matmul(self.lane_graph["lane_vecs"], rot)
        cur_vel = np.sqrt((vel_n[0, -1] * vel_n[0, -1]).sum(), dtype=F32)
        pad = np.ones(ang.shape, F32)[:, :self.obs_len]
        return self._scene_inputs(orig, rot, pos_n, ang_n, vel_n, types, pad, ctrs, vecs, cur_vel, lane_ctrs, lane_vecs)

    def update_obser(self, cur):
        end_t, cur_t = cur["END_T"], cur["CUR_T"]
        keep = self.obs_len + (end_t - cur_t)
        hist_trim(cur, keep)
        o = self.obs_len
        w0 = max(min(keep, hist_len(cur)) - o, 0)
        pos, cov = hist_rows(cur, "TRAJS_POS_HIST", w0, keep), hist_rows(cur, "TRAJS_COV_HIST", w0, keep)
        ang, vel = hist_rows(cur, "TRAJS_ANG_HIST", w0, keep), hist_rows(cur, "TRAJS_VEL_HIST", w0, keep)
        orig, rot, theta, pos_n, ang_n, vel_n, ctrs, vecs = U.normalize_agents(pos, ang, vel)
        # lane anchors re-expressed in the new AV frame (get_new_lane_graph, utils.py:171-177); the
        # instance-frame node features (and hence LaneNet's output) are unchanged
        lane_ctrs = np.matmul(self.lane_graph["lane_ctrs"] - orig, rot)
        lane_vecs = np.matmul(self.lane_graph["lane_vecs"], rot)
        cur_vel = np.sqrt((vel_n[0, -1] * vel_n[0, -1]).sum(), dtype=F32)
        pad = np.ones(ang.shape, F32)[:, :o]
        s = self._scene_inputs(orig, rot, pos_n, ang_n, vel_n, cur["TRAJS_TYPE"], pad, ctrs, vecs, cur_vel,
                               lane_ctrs, lane_vecs)
        s.update({"SCEN_PROB": cur["SCEN_PROB"], "SCEN_ID": cur["SCEN_ID"], "PARENT_ID": cur["PARENT_ID"],
                  "CUR_T": end_t, "END_T": self.pred_len, "TRAJS_TID": cur["TRAJS_TID"], "TRAJS_CAT": cur["TRAJS_CAT"],
                  "TRAJS_POS_HIST": pos.copy(), "TRAJS_COV_HIST": cov.copy(), "TRAJS_ANG_HIST": ang.copy(),
                  "TRAJS_VEL_HIST": vel.copy()})
        return s, cur

    # ------------------------------------------------------------------------------------------
    def get_scenario_tree(self):
        if getattr(self, "last_trees", None) is not None:      # the trees of the native plan in flight (reset() drops them)
            return self.last_trees
        return self._scenario_trees_from_tree()

    def _scenario_trees_from_tree(self, tree=None):
        """tree: the AIME tree to walk (default: the generator's current one)"""
        src = self.tree if tree is None else tree
        root = src.get_root()
        for node in [n for n in src.get_leaf_nodes() if n.data.end_flag]:              # (get_end_set) label every node on a finished branch
            while node.parent_key is not None:
                node.data.end_flag = True
                node = src.get_node(node.parent_key)
        probs = {}
        order = {}
        for key in root.children_keys:
            top = src.get_node(key)
            if not top.data.end_flag:
                continue
            probs[key] = 1.0
            order[key] = [key]
            queue = [top]
            while queue:
                cur = queue.pop(0)
                ch = [src.get_node(k) for k in cur.children_keys]
                ch = [c for c in ch if c.data.end_flag]
                total = 0.0
                for c in ch:
                    total = total + c.data.data["SCEN_PROB"]
                for c in ch:
                    probs[c.key] = c.data.data["SCEN_PROB"] / total * probs[cur.key]
                    order[key].append(c.key)
                    queue.append(c)
        trees = []
        o = self.obs_len
        for key, keys in order.items():
            t = Tree()
            for k in keys:
                n = src.get_node(k)
                d = n.data.data
                dur = d["END_T"] - d["CUR_T"]
                t.add_node(Node(k, None if k == key else n.parent_key,
                                [probs[k], hist_rows(d, "TRAJS_POS_HIST", o, o + dur), hist_rows(d, "TRAJS_COV_HIST", o, o + dur),
                                 d["TGT_PTS"]]))
            trees.append(t)
        return trees
