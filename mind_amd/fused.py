"""BASELINE config 3 as written: several scenes planned by ONE process with the AIME rounds of all scenes merged into one
predictor batch.

The reference batches the branch set of ONE scene's tree per round (planners/mind/scenario_tree.py:69-71 through
collate_fn, planners/mind/utils.py:142-168).  Scenes never interact inside the predictor (fusion and decoder loop per
scene, networks/network.py:318,497) and the HIP predictor's result for a scene does not depend on what else is in the
batch (tests/test_gpu_predictor.py::test_deterministic_and_batch_invariant), so round r of every scene's tree can share a
`mind_predict_batch` call: the small-scene launches (a few dozen workgroups each) then fill more of the GPU.  Each
planner runs as a generator over its rounds (`MINDPlanner.plan_rounds`); `FusedScenes.plan_all` answers all pending rounds
with one predictor call and hands every planner its slice.  Results are bit-identical to planning the scenes one by one.
"""
import numpy as np
import torch


class FusedScenes:
    def __init__(self, planners):
        self.planners = list(planners)
        nets = [pl.network for pl in self.planners]
        self.rt = nets[0].rt
        for n in nets:
            if not hasattr(n, "rt") or n.rt is not self.rt or type(n).__name__ != "ScenePredNet":
                raise ValueError("FusedScenes needs plain ScenePredNet planners that share one HIP runtime (same thread / stream)")
        self.n_calls = 0
        self.n_scenes = 0

    # ---- one predictor call for the pending rounds of several scenes --------------------------------------------------
    def _predict_merged(self, reqs):
        """reqs: [(scenes, d)] -> [(out, packed, lane_feat)] in the same order"""
        with_lanes = [r[1]["lane_feat"] is None for r in reqs]
        if any(with_lanes) and not all(with_lanes):       # a first round next to later rounds: one call per kind
            res = [None] * len(reqs)
            for kind in (True, False):
                idx = [i for i, w in enumerate(with_lanes) if w == kind]
                for i, o in zip(idx, self._predict_merged([reqs[i] for i in idx])):
                    res[i] = o
            return res
        ds = [r[1] for r in reqs]
        cat = lambda k: None if ds[0][k] is None else (ds[0][k] if len(ds) == 1 else torch.cat([d[k] for d in ds]))
        a_off, l_off = [0], [0]
        for d in ds:
            a_off += [a_off[-1] + (v - d["a_off"][0]) for v in d["a_off"][1:]]
            l_off += [l_off[-1] + (v - d["l_off"][0]) for v in d["l_off"][1:]]
        rpe = None
        if ds[0]["rpe"] is not None:
            rpe = [r for d in ds for r in d["rpe"]]
        want = with_lanes[0]
        o = self.rt.predict(cat("actors"), a_off, cat("lanes"), l_off, cat("actor_ctrs"), cat("actor_vecs"), cat("lane_ctrs"),
                            cat("lane_vecs"), cat("tgt_nodes"), cat("tgt_rpe"), rpe=rpe, lane_feat=cat("lane_feat"), want_lane_feat=want)
        self.n_calls += 1
        res, b0, a0, l0 = [], 0, 0, 0
        for d in ds:
            B = len(d["a_off"]) - 1
            A, L = d["a_off"][-1] - d["a_off"][0], d["l_off"][-1] - d["l_off"][0]
            self.n_scenes += B
            cls, reg, vel = o["cls"][b0:b0 + B], o["reg"][a0:a0 + A], o["vel"][a0:a0 + A]
            ao = [v - d["a_off"][0] for v in d["a_off"]]
            packed = {"n": B, "cls": cls, "reg": reg, "vel": vel, "a_off": ao, "actor_ctrs": d["actor_ctrs"],
                      "actor_vecs": d["actor_vecs"], "rt": self.rt}
            out = ([cls[b:b + 1] for b in range(B)], [reg[ao[b]:ao[b + 1]] for b in range(B)],
                   [(vel[ao[b]:ao[b + 1]], None, None) for b in range(B)])
            lane_feat = None
            if want and "lane_feat" in o and d.get("lane_shared"):
                lane_feat = o["lane_feat"][l0:l0 + (d["l_off"][1] - d["l_off"][0])]
            elif not want:
                lane_feat = d["lane_feat"][:d["l_off"][1] - d["l_off"][0]]
            res.append((out, packed, lane_feat))
            b0, a0, l0 = b0 + B, a0 + A, l0 + L
        return res

    # ---- lock-step planning ---------------------------------------------------------------------------------------------
    def plan_all(self, lcl_smps, which=None):
        """plan() of planners[i] on lcl_smps[i] for i in `which` (default all): -> list of plan() results (None elsewhere)."""
        which = list(range(len(self.planners))) if which is None else list(which)
        results = [None] * len(self.planners)
        gens, pending = {}, {}
        for i in which:
            gens[i] = self.planners[i].plan_rounds(lcl_smps[i])
            try:
                pending[i] = next(gens[i])
            except StopIteration as e:
                results[i] = e.value
        while pending:
            idx = sorted(pending)
            outs = self._predict_merged([pending[i] for i in idx])
            nxt = {}
            for i, o in zip(idx, outs):
                try:
                    nxt[i] = gens[i].send(o)
                except StopIteration as e:
                    results[i] = e.value
            pending = nxt
        return results


class FusedClosedLoops:
    """P headless closed loops (mind_amd.closed_loop.ClosedLoopSim) advanced in lock-step, every planning trigger of the
    step answered through FusedScenes.plan_all."""

    def __init__(self, sims):
        self.sims = list(sims)
        self.fused = FusedScenes([s.planner for s in self.sims])

    def step(self):
        lcls = [s.step_begin() for s in self.sims]
        due = [i for i, l in enumerate(lcls) if l is not None]
        res = self.fused.plan_all(lcls, due) if due else [None] * len(self.sims)
        return sum(s.step_end(r) for s, r in zip(self.sims, res))

    def run_plans(self, n):
        """advance until every loop computed n more plans; returns the simulator steps taken (summed over the loops)."""
        p0 = [s.n_plans for s in self.sims]
        s0 = sum(s.n_steps for s in self.sims)
        replay = 0
        while min(s.n_plans - p for s, p in zip(self.sims, p0)) < n:
            for s in self.sims:
                b = s.n_steps
                if s.maybe_restart_episode():
                    replay += s.n_steps - b                  # reset() keeps n_steps: nothing to subtract, kept for clarity
            self.step()
        return sum(s.n_steps for s in self.sims) - s0 - replay
