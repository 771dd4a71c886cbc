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
