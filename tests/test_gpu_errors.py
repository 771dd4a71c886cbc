"""GPU: error behaviour of the C-ABI through the Python mirror -- every entry point returns a negative MIND_E* code and
a message instead of crashing; the mirror raises MindError (errors surface as exceptions, as in the reference).
The context stays usable after an error."""
import numpy as np
import pytest

from mind_amd import _lib
from mind_amd.synth import predictor_batch, scripted_scenario_tree
from oracle import ilqr as oi

pytestmark = pytest.mark.gpu


def _tree():
    sst = scripted_scenario_tree("lead", 4)
    return sst, oi.flatten(sst["nodes"]), oi.init_state(sst["state"], sst["ctrl"])


def test_ilqr_argument_errors_leave_the_context_usable(hip_predictor):
    sst, flat, x0 = _tree()
    cfg = oi.default_cfg(max_iter=3)
    good = hip_predictor.ilqr_solve(cfg, [flat], x0, sst["target_lane"], sst["target_vel"], 1)
    # more agents than the kernel stages
    big = dict(flat, mean=np.zeros((len(flat["parent"]), 129, 2), np.float32), cov=np.ones((len(flat["parent"]), 129), np.float32))
    with pytest.raises(_lib.MindError, match="129 agents"):
        hip_predictor.ilqr_solve(cfg, [big], x0, sst["target_lane"], sst["target_vel"], 1)
    # a node whose parent comes after it / a second root
    for bad_parent in (np.r_[-1, 2, flat["parent"][2:]], np.r_[-1, -1, flat["parent"][2:]]):
        with pytest.raises(_lib.MindError, match="has parent"):
            hip_predictor.ilqr_solve(cfg, [dict(flat, parent=bad_parent.astype(np.int32))], x0, sst["target_lane"], sst["target_vel"], 1)
    # target lane of one point, no trees, mismatching configurations of a contingency call
    with pytest.raises(_lib.MindError, match="target lane"):
        hip_predictor.ilqr_solve(cfg, [flat], x0, sst["target_lane"][:1], sst["target_vel"], 1)
    with pytest.raises(_lib.MindError):
        hip_predictor.ilqr_solve(cfg, [], x0, sst["target_lane"], sst["target_vel"], 1)
    other = oi.default_cfg(max_iter=3)
    other.grid_res = 0.5
    with pytest.raises(_lib.MindError, match="share"):
        hip_predictor.ilqr_contingency(cfg, other, [flat], x0, sst["target_lane"], sst["target_vel"])
    again = hip_predictor.ilqr_solve(cfg, [flat], x0, sst["target_lane"], sst["target_vel"], 1)
    assert np.array_equal(good[0][0], again[0][0])


def test_predictor_argument_errors(hip_predictor, formula_sd):
    from mind_amd.predictor import HipPredictor
    pb = predictor_batch(3, 4, 1, seed=1)
    fresh = HipPredictor(0)
    with pytest.raises(_lib.MindError, match="weights not loaded"):
        fresh.predict_numpy_batch(pb)
    bad = {k: v for k, v in formula_sd.items() if k != "actor_net.groups.0.0.conv1.weight"}
    assert len(bad) == len(formula_sd) - 1
    with pytest.raises(_lib.MindError, match="missing"):
        fresh.load_state_dict(bad)
    fresh.close()
    out = hip_predictor.predict_numpy_batch(pb)                 # the shared context is unaffected
    assert np.all(np.isfinite(out["reg"].cpu().numpy()))


def test_context_creation_rejects_bad_device():
    import ctypes as C
    lib = _lib.load()
    ctx = C.c_void_p()
    assert lib.mind_ctx_create(999, None, C.byref(ctx)) == _lib.MIND_EINVAL
    assert lib.mind_ctx_create(0, None, None) == _lib.MIND_EINVAL


def test_plan_begun_solves_that_are_never_collected_do_not_outlive_their_plan():
    """The native plan begins its contingency solves itself (mind_aime_plan_in.solve_*) and keeps their outputs in library vectors until
    mind_ilqr_finish_plan.  A caller that gives such a plan up (an exception between plan_start and plan_end) must not leave the next plan a
    dangling closure: the next mind_aime_plan drains and drops it, mind_ilqr_finish_plan refuses to copy a plan of another shape into the
    caller's arrays, and a finished-but-uncollected mind_aime_plan_begin does not block the next one."""
    import ctypes as C
    import os
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    from bench import BRANCHING_WEIGHTS, WORKLOADS, make_closed_loop
    ref_pl, ref_sim, _ = make_closed_loop(dict(WORKLOADS["demo_1"]), ckpt=BRANCHING_WEIGHTS, speculative=False)
    pl, sim, _ = make_closed_loop(dict(WORKLOADS["demo_1"]), ckpt=BRANCHING_WEIGHTS, speculative=False, own_context=True)
    rt = pl.network.rt
    for cycle in range(3):
        lcl_ref, lcl = None, None
        while lcl is None:
            lcl_ref, lcl = ref_sim.step_begin(), sim.step_begin()
            if lcl is None:
                ref_sim.step_end(None); sim.step_end(None)
        want = ref_pl.plan(lcl_ref)
        # 1. a plan begun in three pieces and abandoned after its solves were begun by the library
        begun = pl.plan_begin_finish(pl.plan_start(lcl))
        assert pl.timing is not None and len(begun[1]) > 0
        if cycle == 0:
            # 2. collecting with the wrong shape copies nothing and says so; the pending half is gone afterwards
            xs, us = np.zeros((3, 6)), np.zeros((3, 2))
            st = (_lib.IlqrStats * 1)()
            rc = rt.lib.mind_ilqr_finish_plan(rt.ctx, 3, 1, xs.ctypes.data, us.ctypes.data, None, C.cast(st, C.c_void_p))
            assert rc == _lib.MIND_EINVAL and not xs.any()
            rc = rt.lib.mind_ilqr_finish_plan(rt.ctx, 3, 1, xs.ctypes.data, us.ctypes.data, None, C.cast(st, C.c_void_p))
            assert rc == _lib.MIND_ESTATE
        if cycle == 1:
            # 3. a native plan that ran to its end on the library's thread and is never collected
            started = pl.plan_start(lcl)
            while not pl.plan_started_ready(started):
                pass
        pl.traj_tree_opt._pending = None                # (what an exception handler around the pieces would leave behind)
        got = pl.plan(lcl)
        assert got[0] and want[0] and np.array_equal(np.asarray(got[1]), np.asarray(want[1])), cycle
        assert pl.timing["best_traj_idx"] == ref_pl.timing["best_traj_idx"]
        ref_sim.step_end(want); sim.step_end(got)


def test_native_loop_errors_are_codes_and_messages():
    """mind_loop_*: a bad descriptor, scene tables that end, a plan read before there is one, a sharded context -- negative codes and a
    message, never a crash; the context (shared with the other tests of the process) stays usable."""
    import ctypes as C
    import os
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    from bench import BRANCHING_WEIGHTS, WORKLOADS, make_closed_loop
    pl, sim, w = make_closed_loop(dict(WORKLOADS["demo_1"]), ckpt=BRANCHING_WEIGHTS, native=None)
    nl = sim._native
    assert nl is not None
    lib, ctx = nl.lib, nl.rt.ctx
    # a descriptor without tables
    bad = _lib.LoopDesc()
    h = C.c_void_p()
    assert lib.mind_loop_create(ctx, C.byref(bad), C.byref(h)) == _lib.MIND_EINVAL
    assert b"bad argument" in lib.mind_last_error_string(ctx)
    assert lib.mind_loop_create(ctx, None, C.byref(h)) == _lib.MIND_EINVAL
    # no plan yet: nothing to hand out
    po = _lib.AimePlanOut()
    ptr = [C.c_void_p() for _ in range(6)]
    x0 = np.zeros(6)
    assert lib.mind_loop_last_plan(nl.h, C.byref(po), *[C.byref(p) for p in ptr], x0.ctypes.data) == _lib.MIND_ESTATE
    assert sim.last_result is None
    # the scene tables cover one episode (+ a cycle): stepping past them is refused, and the loop can be reset
    sim.episode_plans = None
    out = _lib.LoopOut()
    rc = lib.mind_loop_advance(nl.h, 0, -1.0, 10 ** 6, C.byref(out))
    assert rc == _lib.MIND_EINVAL and b"scene tables end" in lib.mind_last_error_string(ctx)
    assert out.n_plans >= 60 and np.all(np.isfinite(np.array(out.state)))
    assert lib.mind_loop_reset(nl.h) == 0
    assert lib.mind_loop_advance(nl.h, 1, -1.0, 10 ** 6, C.byref(out)) == 0 and out.planned == 1
    # a sharded context plans through its exchange: the loop refuses it
    cb = _lib.EXCHANGE_FN(lambda user, op, send, recv, nbytes: 0)
    assert lib.mind_set_exchange(ctx, 0, 2, cb, None, 0) == 0
    try:
        assert lib.mind_loop_advance(nl.h, 1, -1.0, 10 ** 6, C.byref(out)) == _lib.MIND_ESTATE
        assert b"sharded" in lib.mind_last_error_string(ctx)
    finally:
        assert lib.mind_set_exchange(ctx, 0, 1, C.cast(None, _lib.EXCHANGE_FN), None, 0) == 0
    assert lib.mind_loop_advance(nl.h, 1, -1.0, 10 ** 6, C.byref(out)) == 0 and out.planned == 1
