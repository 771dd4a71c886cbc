"""GPU-box diagnostic: stage-by-stage comparison of the HIP predictor with the CPU oracle, for every arithmetic of the pair
kernel (f32 / bf16x3 / bf16), and pair-kernel timings.
Usage (on the GPU box): python tests/diag/gpu_diag_predictor.py [--big] [--prec f32,bf16x3,bf16] [--tile 0,1] [--timing-only]
--tile: the bf16 arithmetics with the row-major pair kernel of rounds 2-3 (0) and / or the tile-native one (1), same box.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from mind_amd.predictor import HipPredictor
from mind_amd.synth import predictor_batch
from mind_amd.weights import formula_state_dict
from oracle import predictor as op


def tt(pb):
    return {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else [torch.from_numpy(x) for x in v])
            for k, v in pb.items()}


def edge_ij(hp, B, n):
    """the device edge tensor is stored [scene][j][i][128] (query column major): back to the reference's [i][j]"""
    return hp.debug_read("edge").reshape(B, n, n, 128).transpose(0, 2, 1, 3)


def main():
    sd = formula_state_dict(as_torch=True)
    hp = HipPredictor(0)
    hp.load_state_dict(sd)
    precs = ["f32", "bf16x3", "bf16"]
    if "--prec" in sys.argv:
        precs = sys.argv[sys.argv.index("--prec") + 1].split(",")
    ok = True
    tiles = [1]
    if "--tile" in sys.argv:
        tiles = [int(v) for v in sys.argv[sys.argv.index("--tile") + 1].split(",")]
    for prec in precs:
        hp.set_pair_precision(prec)
        for tile in (tiles if prec != "f32" else [1]):
            hp.set_tuning("pair_tile", tile)
            print(f"######## pair kernel arithmetic: {prec}" + ("" if prec == "f32" else f", pair_tile = {tile}"))
            if "--timing-only" not in sys.argv:
                ok = run_parity(hp, sd, prec) and ok
            run_timing(hp)
    print("DIAG_OK" if ok else "DIAG_FAIL")


def run_parity(hp, sd, prec):
    ok = True
    bar = 3e-2 if prec == "bf16" else 1e-3
    for (a, l, B) in [(3, 4, 1), (8, 20, 2), (40, 55, 1), (17, 30, 3)]:
        pb = predictor_batch(a, l, B, seed=1)
        tb = tt(pb)
        taps = {}
        oc, orr, ov = op.forward(sd, tb, taps=taps)
        n = a + l + 1
        print(f"== config a={a} l={l} B={B} N={n}")
        # encoders
        hp.debug_set_layers(0)
        out = hp.predict_numpy_batch(pb, want_lane_feat=True)
        af = hp.debug_read("actor_feat").reshape(-1, 128)
        print("  actor_net  max|d| = %.3e (ref max %.3f)" % (np.abs(af - taps["actor_net"].numpy()).max(), taps["actor_net"].abs().max()))
        lf = out["lane_feat"].cpu().numpy()
        print("  lane_net   max|d| = %.3e" % np.abs(lf - taps["lane_net"].numpy()).max())
        for k in range(0, 7):
            hp.debug_set_layers(k)
            out = hp.predict_numpy_batch(pb)
            x = hp.debug_read("x").reshape(B, n, 128)
            if k == 0:
                continue
            errs, eerrs = [], []
            for b in range(B):
                xr = taps["fusion"][b][k - 1][0].numpy()
                if k == 6:
                    d = np.abs(x[b] - xr)
                    sel = list(range(a)) + [n - 1]
                    errs.append(d[sel].max())
                else:
                    errs.append(np.abs(x[b] - xr).max())
            if k <= 5:
                e = edge_ij(hp, B, n)
                for b in range(B):
                    er = taps["fusion"][b][k - 1][1].numpy()
                    d = np.abs(e[b] - er)
                    if k == 5:
                        sel = list(range(a)) + [n - 1]
                        d = d[:, sel]
                    eerrs.append(d.max())
            print(f"  layer {k}: x max|d| = {max(errs):.3e}" + (f"  edge max|d| = {max(eerrs):.3e}" if eerrs else ""))
        hp.debug_set_layers(6)
        out = hp.predict_numpy_batch(pb)
        cm = hp.debug_read("cmode").reshape(B, 6, 128)
        te = hp.debug_read("tgt_emb").reshape(B, 128)
        tf = hp.debug_read("tgt_feat").reshape(B, 128)
        print("  dec: tgt_feat %.3e tgt_emb %.3e cmode %.3e" % (
            np.abs(tf - taps["tgt_feat"].numpy()).max(),
            max(np.abs(te[b] - taps["dec"][b]["tgt_emb"].numpy()[0]).max() for b in range(B)),
            max(np.abs(cm[b] - taps["dec"][b]["cmode"].numpy()[:, 0]).max() for b in range(B))))
        pbr = dict(pb)
        pbr["RPE"] = [op.rpe(c, v).numpy() for c, v in zip(tb["CTRS"], tb["VECS"])]
        hp.debug_set_layers(1)
        hp.predict_numpy_batch(pbr, use_rpe=True)
        x = hp.debug_read("x").reshape(B, n, 128)
        e = edge_ij(hp, B, n)
        print("  rpe-in layer1: x %.3e edge %.3e" % (
            max(np.abs(x[b] - taps["fusion"][b][0][0].numpy()).max() for b in range(B)),
            max(np.abs(e[b] - taps["fusion"][b][0][1].numpy()).max() for b in range(B))))
        hp.debug_set_layers(6)
        for use_rpe in (False, True):
            if use_rpe:
                pb["RPE"] = [op.rpe(c, v).numpy() for c, v in zip(tb["CTRS"], tb["VECS"])]
            out = hp.predict_numpy_batch(pb, use_rpe=use_rpe)
            cls = out["cls"].cpu().numpy()
            reg = out["reg"].cpu().numpy()
            vel = out["vel"].cpu().numpy()
            ec = max(np.abs(cls[b] - oc[b].numpy()[0]).max() for b in range(B))
            er = max(np.abs(reg[b * a:(b + 1) * a] - orr[b].numpy()).max() for b in range(B))
            ev = max(np.abs(vel[b * a:(b + 1) * a] - ov[b].numpy()).max() for b in range(B))
            print(f"  final (rpe_in={use_rpe}): cls {ec:.3e} reg {er:.3e} vel {ev:.3e}")
            if not (ec < bar and er < bar and ev < bar):
                ok = False
    return ok


def run_timing(hp):
    hp.set_profiling(True)
    for (a, l, B) in [(40, 55, 1), (40, 55, 6), (64, 256, 4)] + ([(64, 256, 24)] if "--big" in sys.argv else []):
        pb = predictor_batch(a, l, B, seed=2)
        out = hp.predict_numpy_batch(pb)
        torch.cuda.synchronize()
        t0 = time.time()
        reps = 5
        for _ in range(reps):
            out = hp.predict_numpy_batch(pb)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / reps
        nl, ms, pairs = hp.fusion_stats()
        n = a + l + 1
        fmin = 754944.0 * n * n * B
        print(f"timing a={a} l={l} B={B}: {dt*1e3:.2f} ms/forward ({B/dt:.1f} scenes/s); pair kernels {ms:.3f} ms over {nl} launches;"
              f" F_min-rate {fmin/ (ms*1e-3)/1e12:.1f} TFLOP/s, {4096.0*n*n*B/(ms*1e-3)/1e12:.2f} TB/s edge traffic (pair kernels only)")
    hp.set_profiling(False)


if __name__ == "__main__":
    main()
