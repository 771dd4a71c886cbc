"""Headless closed-loop driver: the reference's simulator loop and MINDAgent trigger logic
without rendering, so that "sim steps/s" can be measured on the GPU box where the reference harness
cannot travel.

Mirrors: Simulator.run_sim (simulator.py:51-107; 0.02 s steps), CustomizedAgent.check_trigger /
check_enable (agent.py:255-286; planner at 10 Hz with plan_step = 0.1 - 1e-4, enabled from t = 4.0 s),
MINDAgent.plan / update_observation (agent.py:324-331) and the ego plant kine_propagate
(common/kinematics.py:22-36 with the simulator's wheelbase 3.0, max speed 15 m/s, max steer 45 deg;
agent.py:298-299 -- the planner itself uses wheelbase 2.5, Q16).
Exo agents replay the world's trajectories: a SynthWorld's formula tracks or a scene_io.ReplayWorld's recorded
AV2 tracks (then only agents observed at the current step are reported, simulator.py:60-63, and the planner
gets the 4 m-spaced target lane of MINDAgent.update_target_lane, agent.py:320-322).
"""
from types import SimpleNamespace

import numpy as np


def kine_propagate(state, ctrl, dt, wb=2.5, max_spd=20.0, max_steer=np.deg2rad(45.0), max_acc=6.0, max_dec=-6.0):
    x, y, v, yaw = state
    a, delta = ctrl
    a = np.clip(a, max_dec, max_acc)
    delta = np.clip(delta, -max_steer, max_steer)
    out = np.array([x + v * np.cos(yaw) * dt, y + v * np.sin(yaw) * dt, v + a * dt, yaw + v / wb * np.tan(delta) * dt])
    out[2] = np.clip(out[2], -max_spd, max_spd)
    return out


class ClosedLoopSim:
    SIM_STEP = 0.02
    PLAN_STEP = 1.0 / 10 - 1e-4
    WB, MAX_SPD, MAX_STR = 3.0, 15.0, np.deg2rad(45.0)

    def __init__(self, world, planner, enable_time=None, episode_plans=None, native=None):
        """native: None = use the native loop (mind_amd/native_loop.py: one C call per step / planning cycle) when it applies to this
        planner and world, False = the Python steps below, True = raise when it does not apply"""
        self.world = world
        self._native, self._last_result = None, None
        self.planner = planner
        self.enable_time = enable_time if enable_time is not None else getattr(world, "enable_time", 4.0)
        self.n_steps = 0
        self.n_plans = 0
        # episode_plans = E: after E planning cycles the scene starts over from t = 0 (the reference's runs are
        # episodes too: 500 steps = 60 cycles).  Long throughput runs need it: a recording ends after 11 s, and the
        # synthetic worlds' non-reactive scripted agents eventually push the ego out of every mode's reach.
        self.episode_plans = episode_plans
        self.n_episodes = 0
        gt_lane = getattr(world, "gt_tgt_lane", None)
        planner.update_target_lane(np.asarray(world.target_lane[::2], dtype=np.float64) if gt_lane is None else gt_lane)
        self._valid = getattr(world, "is_valid", None)
        self._exo_ahead = None
        import os
        if os.environ.get("MIND_PREFETCH_OBS", "1") != "0" and hasattr(planner, "plan"):
            try:
                # (MINDPlanner calls it while the device computes; other planners ignore it.)  Always THIS simulator's method: a second
                # simulator around the same planner must not leave the first one's hook -- and its world -- behind
                planner.idle_hook = self._prefetch_observation
            except AttributeError:
                pass
        self._start_episode()
        if native is not False:
            from .native_loop import NativeLoop
            why = NativeLoop.why_not(self)
            if why is None:
                try:
                    self._native = NativeLoop(self)
                except ValueError as e:          # (a scene the native plan does not take: no lanes, a short target lane)
                    why = str(e)
            if why is not None and native:
                raise RuntimeError("ClosedLoopSim(native=True): " + why)
        self.native_reason = None if self._native is not None else "native=False" if native is False else why

    @property
    def last_result(self):
        """[[scenario tree], [trajectory tree]] of the last plan; under the native loop the objects are built when this is read"""
        nl = self._native
        return nl.last_result() if nl is not None else self._last_result

    @last_result.setter
    def last_result(self, v):
        self._last_result = v

    def _native_now(self):
        """the native loop if it still applies (a planner attribute may have changed under it: then the loop is handed back to the steps below)"""
        nl = self._native
        if nl is not None and not nl.ok():
            nl.hand_back()
            nl = None
        return nl

    def _start_episode(self):
        self.sim_time = 0.0
        self.enabled = False
        self.last_trigger = None
        self.state = self.world.agent_state(0, 0.0)     # (x, y, v, yaw)
        self.ctrl = np.array([0.0, 0.0])
        self.timestep = 0.0
        self.last_result = None
        self._exo_ahead = None
        self._episode_plan0 = self.n_plans
        if hasattr(self.planner, "agent_obs"):
            self.planner.agent_obs.clear()
        nl = getattr(self, "_native", None)
        if nl is not None:
            nl.lib.mind_loop_reset(nl.h)
            nl._result = None

    def reset(self):
        """Start the next episode: scene back to t = 0, observation history rebuilt up to the enable time (these
        replay steps are not counted in n_steps)."""
        n = self.n_steps
        self._start_episode()
        self.n_episodes += 1
        self.run_until(self.enable_time)
        self.n_steps = n

    def _exo_observation(self, t):
        w = self.world
        return [SimpleNamespace(state=w.agent_state(i, t), type=w.object_type(i), id=w.agent_ids[i],
                                timestep=int(round(t / 0.1))) for i in range(1, w.n_agents)
                if self._valid is None or self._valid(i, t)]

    def _prefetch_observation(self):
        """planner idle hook (MINDPlanner.idle_hook: called while the device computes the contingency solves): the replayed agents of the
        NEXT planning step do not depend on this plan -- their observation list is built now and picked up by _observation() if the
        simulator time it was built for is exactly the one that comes (the same float additions are replayed here)."""
        t = self.sim_time
        for _ in range(int(round(self.PLAN_STEP / self.SIM_STEP))):
            t += self.SIM_STEP
        exo = self._exo_observation(t)
        to_state = getattr(self.planner, "to_object_state", None)
        if to_state is not None:
            for a in exo:
                a.obj_state = to_state(a)
        self._exo_ahead = (t, exo)

    def _observation(self):
        t = self.sim_time
        w = self.world
        ego_state = self.state if self.enabled else w.agent_state(0, t)
        ego = SimpleNamespace(state=ego_state, type=w.object_type(0), id="AV", timestep=int(round(t / 0.1)))
        ahead, self._exo_ahead = getattr(self, "_exo_ahead", None), None
        exo = ahead[1] if ahead is not None and ahead[0] == t else self._exo_observation(t)
        return SimpleNamespace(ego_agent=ego, exo_agents=exo, map_data=w, target_lane=w.target_lane,
                               target_lane_info=w.target_lane_info, target_velocity=w.target_velocity)

    def step_begin(self):
        """First half of a simulator step: take-over check, observation fan-out, planner trigger.  Returns the local
        semantic map if a plan is due in this step (the caller plans and passes the result to step_end), else None."""
        if self._native is not None:       # a driver that steps in halves (pipelined.py, fused.py) runs the Python steps: the loop is handed back
            self._native.hand_back()
        if self.sim_time >= self.enable_time and not self.enabled:
            self.enabled = True                                  # check_enable: take over from the recording
            self.state = self.world.agent_state(0, self.sim_time)
            self.ctrl = np.array([0.0, 0.0])
        if self.last_trigger is None or (self.sim_time - self.last_trigger) >= self.PLAN_STEP:
            self.last_trigger = self.sim_time
            lcl = self._observation()
            self.planner.update_observation(lcl)
            if self.enabled:
                self.planner.update_state_ctrl(lcl.ego_agent.state, self.ctrl)
                return lcl
        return None

    def step_end(self, plan_result=None):
        """Second half: apply the plan computed for this step (if one was due), propagate the ego plant, advance time."""
        planned = plan_result is not None
        if planned:
            ok, self.ctrl, self.last_result = plan_result
            if not ok:
                raise RuntimeError("plan failed")
            self.n_plans += 1
        if self.enabled:
            self.state = kine_propagate(self.state, self.ctrl, self.SIM_STEP, self.WB, self.MAX_SPD, self.MAX_STR)
        self.sim_time += self.SIM_STEP
        self.n_steps += 1
        return planned

    def step(self):
        """One simulator step (0.02 s).  Returns True if a plan was computed in this step."""
        nl = self._native_now()
        if nl is not None:
            return nl.advance(0, -1.0, 1) > 0
        lcl = self.step_begin()
        return self.step_end(self.planner.plan(lcl) if lcl is not None else None)

    def run_until(self, t_end):
        nl = self._native_now()
        if nl is not None:
            nl.advance(0, float(t_end), 1 << 40)
            if self._native is nl:
                return
        while self.sim_time < t_end - 1e-9:
            self.step()

    def maybe_restart_episode(self):
        """start the next episode if this one is over (used by run_plans and by lock-step multi-scene drivers)"""
        if self.episode_plans is not None and self.n_plans - self._episode_plan0 >= self.episode_plans:
            self.reset()
            return True
        return False

    def run_plans(self, n):
        """advance until n more plans were computed; returns the number of simulator steps taken."""
        s0, p0 = self.n_steps, self.n_plans
        while self.n_plans - p0 < n:
            if self.episode_plans is not None and self.n_plans - self._episode_plan0 >= self.episode_plans:
                s_keep = self.n_steps - s0
                self.reset()
                s0 = self.n_steps - s_keep
            nl = self._native_now()
            if nl is not None:         # up to the end of the episode in one call
                want = n - (self.n_plans - p0)
                if self.episode_plans is not None:
                    want = min(want, self.episode_plans - (self.n_plans - self._episode_plan0))
                nl.advance(want, -1.0, 1 << 40)
            else:
                self.step()
        return self.n_steps - s0
