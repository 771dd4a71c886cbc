"""Several closed loops planned from ONE host thread with their device work overlapped (BASELINE config 3 in one process).

Every scene's planner owns a HIP context on its own stream (planner config "own_context", runtime.new_runtime).  The driver calls scene
i + 1's ``plan_begin`` -- AIME rounds, start of the contingency solves -- before scene i's ``plan_end``: scene i's tree-iLQR kernel
(a handful of workgroups, ~2 ms) runs on the device beside scene i + 1's predictor launches, and the host thread, which waits inside
scene i + 1's native AIME call anyway, collects scene i's result afterwards.  The scenes stay independent closed loops: every plan is
bit for bit the plan the scene computes alone (tests/test_gpu_plan.py)."""


class PipelinedClosedLoops:
    def __init__(self, sims):
        self.sims = list(sims)
        for s in self.sims:
            if not hasattr(s.planner, "plan_begin"):
                raise TypeError("the planner of a pipelined closed loop needs plan_begin / plan_end (MINDPlanner)")

    @staticmethod
    def _advance_to_plan(sim):
        """simulator steps up to (and including the first half of) the next planning step; returns its local semantic map"""
        while True:
            sim.maybe_restart_episode()
            lcl = sim.step_begin()
            if lcl is not None:
                return lcl
            sim.step_end(None)

    def run_plans(self, n):
        """n more planning cycles per scene, scene by scene in round-robin order, one plan in flight behind the one being started.
        Returns the number of simulator steps taken (all scenes)."""
        s0 = sum(s.n_steps for s in self.sims)
        target = [s.n_plans + n for s in self.sims]
        started = [s.n_plans for s in self.sims]
        pending = None
        while True:
            progressed = False
            for i, s in enumerate(self.sims):
                if started[i] >= target[i]:
                    continue
                if pending is not None and pending[0] is s:          # (a single scene: nothing to overlap with)
                    ps, pb = pending
                    ps.step_end(ps.planner.plan_end(pb))
                    pending = None
                begun = s.planner.plan_begin(self._advance_to_plan(s))
                started[i] += 1
                progressed = True
                if pending is not None:
                    ps, pb = pending
                    ps.step_end(ps.planner.plan_end(pb))
                pending = (s, begun)
            if not progressed:
                break
        if pending is not None:
            ps, pb = pending
            ps.step_end(ps.planner.plan_end(pb))
        return sum(s.n_steps for s in self.sims) - s0
