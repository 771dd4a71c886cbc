"""GPU parity sweep on seeded random inputs: scenario trees of random topology / agent layout through the tree-iLQR, and
ragged multi-scene batches through the predictor, each against the CPU oracle on the same inputs."""
import numpy as np
import pytest
import torch

from mind_amd.synth import predictor_batch, scripted_scenario_tree
from oracle import ilqr as oi
from oracle import predictor as op

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(1, 9))
def test_random_scenario_trees_match_oracle(seed, hip_predictor):
    """Random topologies (1..21 scenario nodes, 25..253 trajectory nodes), 9..44 agents of which a third crowd the ego
    path (relevant-agent lists up to and beyond their capacity), warm-start fit then full fit.  The first iterations are
    compared tightly (the reference's solver amplifies rounding differences afterwards)."""
    sst = scripted_scenario_tree("random", 4 + 5 * seed, seed=seed)
    flat = oi.flatten(sst["nodes"])
    x0 = oi.init_state(sst["state"], sst["ctrl"])
    cfg = oi.default_cfg(max_iter=4)
    w = oi.solve(cfg, flat, x0, sst["target_lane"], sst["target_vel"], 0)
    f = oi.solve(cfg, flat, x0, sst["target_lane"], sst["target_vel"], 1, us_init=w["us"])
    xs, us, st_w, st_f = hip_predictor.ilqr_contingency(cfg, cfg, [flat], x0, sst["target_lane"], sst["target_vel"])
    assert st_w[0]["iterations"] == w["iterations"] and st_f[0]["iterations"] == f["iterations"]
    assert st_f[0]["mu"] == f["mu"]                                               # same accept / reject history
    scale = max(1.0, float(np.abs(f["xs"]).max()))
    assert np.abs(xs[0] - f["xs"]).max() < 1e-9 * scale
    assert np.abs(us[0] - f["us"]).max() < 1e-9 * max(1.0, float(np.abs(f["us"]).max()))
    assert abs(st_f[0]["J"] - f["J"]) < 1e-9 * max(1.0, abs(f["J"]))


def test_random_trees_batched_in_one_launch(hip_predictor):
    """All eight random trees (different sizes, different agent counts) in ONE launch equal the single-tree solves."""
    ssts = [scripted_scenario_tree("random", 4 + 5 * s, seed=s) for s in range(1, 9)]
    flats = [oi.flatten(s["nodes"]) for s in ssts]
    x0 = oi.init_state(ssts[0]["state"], ssts[0]["ctrl"])
    cfg = oi.default_cfg(max_iter=3)
    lane, tv = ssts[0]["target_lane"], ssts[0]["target_vel"]
    xs_all, us_all, _, st_all = hip_predictor.ilqr_contingency(cfg, cfg, flats, x0, lane, tv)
    for i in (0, 3, 7):
        xs1, us1, _, st1 = hip_predictor.ilqr_contingency(cfg, cfg, [flats[i]], x0, lane, tv)
        assert np.array_equal(xs_all[i], xs1[0]) and np.array_equal(us_all[i], us1[0]) and st_all[i]["J"] == st1[0]["J"]


def _ragged_batch(rng):
    B = int(rng.integers(1, 5))
    parts = [predictor_batch(int(rng.integers(1, 30)), int(rng.integers(1, 50)), 1, seed=int(rng.integers(1, 10_000))) for _ in range(B)]
    a_off = np.cumsum([0] + [p["ACTORS"].shape[0] for p in parts])
    l_off = np.cumsum([0] + [p["LANES"].shape[0] for p in parts])
    pb = {"ACTORS": np.concatenate([p["ACTORS"] for p in parts]), "LANES": np.concatenate([p["LANES"] for p in parts]),
          "ACTOR_IDCS": [np.arange(a_off[i], a_off[i + 1]) for i in range(B)],
          "LANE_IDCS": [np.arange(l_off[i], l_off[i + 1]) for i in range(B)],
          "CTRS": sum((p["CTRS"] for p in parts), []), "VECS": sum((p["VECS"] for p in parts), []),
          "TGT_NODES": np.concatenate([p["TGT_NODES"] for p in parts]), "TGT_RPE": np.concatenate([p["TGT_RPE"] for p in parts])}
    return pb, parts, a_off


def _to_t(pb):
    return {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else [torch.from_numpy(x) for x in v]) for k, v in pb.items()}


@pytest.mark.parametrize("seed", range(6))
def test_random_ragged_predictor_batches_match_oracle(seed, hip_predictor, formula_sd):
    """1..4 scenes of independent random sizes (1..29 agents, 1..49 lane polylines) collated into one call."""
    rng = np.random.default_rng(100 + seed)
    pb, parts, a_off = _ragged_batch(rng)
    out = hip_predictor.predict_numpy_batch(pb)
    cls, reg, vel = out["cls"].cpu().numpy(), out["reg"].cpu().numpy(), out["vel"].cpu().numpy()
    for b, p in enumerate(parts):
        oc, orr, ov = op.forward(formula_sd, _to_t(p))
        assert np.abs(cls[b] - oc[0].numpy()[0]).max() < 1e-5
        assert np.abs(reg[a_off[b]:a_off[b + 1]] - orr[0].numpy()).max() < 2e-4
        assert np.abs(vel[a_off[b]:a_off[b + 1]] - ov[0].numpy()).max() < 2e-4
