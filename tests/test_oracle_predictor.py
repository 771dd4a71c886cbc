"""CPU: the oracle predictor restatement against the committed golden vectors (captured from the
imported reference by tests/golden/gen_golden.py) and, in the build container, against the reference itself."""
import numpy as np
import pytest
import torch

from mind_amd.synth import predictor_batch
from mind_amd.weights import formula_state_dict, state_dict_spec
from oracle import predictor as op
from oracle import ref_harness as rh

CASES = [(3, 4, 1, 1), (8, 20, 2, 1), (40, 55, 1, 1), (2, 2, 1, 3)]
TOL = 2e-5   # fp32 forward; reference-vs-fp64 itself differs by ~2.5e-6 on O(1..10) outputs


def to_t(pb):
    return {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else [torch.from_numpy(x) for x in v])
            for k, v in pb.items()}


def test_state_dict_spec_counts():
    spec = state_dict_spec()
    assert len(spec) == 328
    assert sum(int(np.prod(s)) for _, s in spec) == 4370921


def test_formula_weights_are_deterministic():
    a = formula_state_dict()
    b = formula_state_dict()
    assert all(np.array_equal(a[k], b[k]) for k in a)
    w = a["fusion_net.proj_actor.0.weight"]
    assert w.dtype == np.float32 and abs(float(w.mean())) < 0.01 and 0.07 < float(w.std()) < 0.1
    assert abs(float(a["lane_net.proj.1.weight"].mean()) - 1.0) < 0.05


@pytest.mark.parametrize("a,l,B,seed", CASES)
def test_oracle_matches_golden(a, l, B, seed, formula_sd, golden_predictor):
    g = golden_predictor
    key = f"a{a}_l{l}_b{B}_s{seed}"
    tb = to_t(predictor_batch(a, l, B, seed=seed))
    taps = {}
    cls, reg, vel = op.forward(formula_sd, tb, taps=taps)
    assert np.abs(taps["actor_net"].numpy() - g[key + "_actor_net"]).max() < TOL
    assert np.abs(taps["lane_net"].numpy() - g[key + "_lane_net"]).max() < TOL
    assert np.abs(np.stack([c.numpy()[0] for c in cls]) - g[key + "_cls"]).max() < 1e-6
    regc = np.concatenate([r.numpy() for r in reg])
    velc = np.concatenate([v.numpy() for v in vel])
    if key + "_tidx" in g:
        regc, velc = regc[:, :, g[key + "_tidx"]], velc[:, :, g[key + "_tidx"]]
    assert np.abs(regc - g[key + "_reg"]).max() < TOL
    assert np.abs(velc - g[key + "_vel"]).max() < TOL
    r0 = op.rpe(tb["CTRS"][0], tb["VECS"][0]).numpy()
    gr = g[key + "_rpe0"]
    assert np.array_equal(r0[:, :gr.shape[1], :gr.shape[2]], gr)   # RPE is bit exact (incl. coincident points)


def test_rpe_diagonal_and_coincident():
    tb = to_t(predictor_batch(8, 20, 1, seed=1))
    r = op.rpe(tb["CTRS"][0], tb["VECS"][0]).numpy()
    n = r.shape[1]
    d = np.arange(n)
    assert np.allclose(r[0, d, d], 1.0, atol=1e-6) and np.all(r[1:, d, d] == 0)
    assert np.all(r[2:, 1, 2] == 0) and np.all(r[2:, 2, 1] == 0)     # coincident centres -> 0/eps


@pytest.mark.reference
@pytest.mark.skipif(not rh.available(), reason="reference tree not present")
def test_oracle_matches_imported_reference(formula_sd):
    m = rh.ref_modules()
    net = rh.build_ref_network(formula_sd)
    sd_ref = net.state_dict()
    assert [k for k, _ in state_dict_spec()] == list(sd_ref.keys())
    get_rpe = m["planners.mind.utils"].get_rpe
    tb = to_t(predictor_batch(6, 9, 2, seed=7))
    rpes = [{"scene": get_rpe(c, v)[0], "scene_mask": None} for c, v in zip(tb["CTRS"], tb["VECS"])]
    rc, rr, ra = net((tb["ACTORS"], tb["ACTOR_IDCS"], tb["LANES"], tb["LANE_IDCS"], rpes, tb["TGT_NODES"], tb["TGT_RPE"]))
    oc, orr, ov = op.forward(formula_sd, tb)
    for b in range(2):
        assert (rc[b] - oc[b]).abs().max() < 1e-6
        assert (rr[b] - orr[b]).abs().max() < TOL
        assert (ra[b][0] - ov[b]).abs().max() < TOL
