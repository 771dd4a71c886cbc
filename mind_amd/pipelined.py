"""Several closed loops planned from ONE host thread with their device work overlapped (BASELINE config 3 in one process).

Every scene's planner owns a HIP context on its own stream (planner config "own_context", runtime.new_runtime).  A planning cycle has
three host pieces -- plan_start (observation, the start of the native AIME plan on a thread of the library), plan_begin_finish (collects
it, starts the contingency solves on the scene's context) and plan_end (collects them, evaluates, returns the control) -- with device
work between them.  The driver is an event loop over the scenes: it runs the next piece of whichever scene's device work has finished
(`plan_started_ready`, `plan_end_ready`: polls, no waits) and only blocks, on the oldest piece in flight, when no scene is ready.  The
host thread therefore does one scene's Python while the other scenes' predictor rounds and tree-iLQR kernels run.  Planners without
the three-piece surface (plan_begin / plan_end only) are driven as before: scene i + 1's plan_begin before scene i's plan_end.  The
scenes stay independent closed loops: every plan is bit for bit the plan the scene computes alone (tests/test_gpu_plan.py)."""


class PipelinedClosedLoops:
    def __init__(self, sims):
        self.sims = list(sims)
        for s in self.sims:
            if not hasattr(s.planner, "plan_begin"):
                raise TypeError("the planner of a pipelined closed loop needs plan_begin / plan_end (MINDPlanner)")
        self._three = all(hasattr(s.planner, "plan_start") for s in self.sims)

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
        """n more planning cycles per scene.  Returns the number of simulator steps taken (all scenes)."""
        return self._run_event_loop(n) if self._three and len(self.sims) > 1 else self._run_two_halves(n)

    # ---- three pieces per cycle, whichever scene is ready next
    def _run_event_loop(self, n):
        s0 = sum(s.n_steps for s in self.sims)
        target = [s.n_plans + n for s in self.sims]
        started = [s.n_plans for s in self.sims]
        state = [None] * len(self.sims)          # None: idle | ("started", token) | ("begun", tuple)
        order = []                               # scenes with a piece in flight, oldest first
        while True:
            progressed = False
            # 1. collect what is ready (oldest first), keeping every scene's next device work queued as early as possible
            for i in list(order):
                kind, tok = state[i]
                pl = self.sims[i].planner
                if kind == "started" and pl.plan_started_ready(tok):
                    state[i] = ("begun", pl.plan_begin_finish(tok))
                    progressed = True
                elif kind == "begun" and pl.plan_end_ready(tok):
                    self.sims[i].step_end(pl.plan_end_piece(tok))
                    state[i] = None
                    order.remove(i)
                    progressed = True
            # 2. start the next cycle of an idle scene (its simulator steps and observation are host work that hides device time)
            for i, s in enumerate(self.sims):
                if state[i] is None and started[i] < target[i]:
                    state[i] = ("started", s.planner.plan_start(self._advance_to_plan(s)))
                    started[i] += 1
                    order.append(i)
                    progressed = True
                    break                        # one at a time: look at the ready list again before more host work
            if progressed:
                continue
            if not order:
                break
            # 3. nothing is ready and nothing can start: poll the pieces in flight until one is (whichever scene's device work ends first),
            #    then go round again; after a while without any, wait for the oldest one (a blocking collect reports a failed call)
            import time
            t_spin = time.perf_counter()
            ready = False
            while not ready and time.perf_counter() - t_spin < 2.0:
                for i in order:
                    kind, tok = state[i]
                    pl = self.sims[i].planner
                    if (kind == "started" and pl.plan_started_ready(tok)) or (kind == "begun" and pl.plan_end_ready(tok)):
                        ready = True
                        break
            if ready:
                continue
            i = order[0]
            kind, tok = state[i]
            pl = self.sims[i].planner
            if kind == "started":
                state[i] = ("begun", pl.plan_begin_finish(tok))
            else:
                self.sims[i].step_end(pl.plan_end_piece(tok))
                state[i] = None
                order.remove(i)
        return sum(s.n_steps for s in self.sims) - s0

    # ---- two halves per cycle, scene by scene in round-robin order, one plan in flight behind the one being started
    def _run_two_halves(self, n):
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
