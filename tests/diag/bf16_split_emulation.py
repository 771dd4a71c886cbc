"""CPU emulation of the pair kernel's reduced-precision MFMA modes (diagnostic, build container or anywhere).

The pair GEMMs of RelaFusionLayer (W_e.e of proj_memory, proj_edge; network.py:197-202) and the folded attention scores
are evaluated with operands rounded to bf16 (``bf16``) or split into bf16 hi + lo parts with the three significant
products kept (``bf16x3``: hi.hi + hi.lo + lo.hi, fp32 accumulate) -- the arithmetic of mind_amd's k_pair<PREC> -- on
top of the fp32 oracle; prints max |d reg| / |d cls| against the fp32 and fp64 oracle runs.

    python tests/diag/bf16_split_emulation.py [a l [seed]]
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import predictor as op          # noqa: E402
from mind_amd.synth import predictor_batch  # noqa: E402
from mind_amd.weights import formula_state_dict  # noqa: E402


def split(t):
    hi = t.to(torch.bfloat16).to(torch.float32)
    lo = (t - hi).to(torch.bfloat16).to(torch.float32)
    return hi, lo


def mm(a, w, mode):
    """a [..., K] x w [O, K]^T in the emulated precision."""
    if mode == "f32":
        return F.linear(a, w)
    ah, al = split(a)
    wh, wl = split(w)
    if mode == "bf16":
        return F.linear(ah, wh)
    if mode == "bf16x3":
        return F.linear(ah, wh) + (F.linear(ah, wl) + F.linear(al, wh))
    raise ValueError(mode)


def fusion_layer(sd, p, x, edge, update_edge, mode, dtype=torch.float32):
    n = x.shape[0]
    D = 128
    W = op._w(sd, p + ".proj_memory.0.weight", dtype)
    b = op._w(sd, p + ".proj_memory.0.bias", dtype)
    S = F.linear(x, W[:, D:2 * D])                 # src = x[j]
    T = F.linear(x, W[:, 2 * D:]) + b              # tar = x[i]
    pre = mm(edge, W[:, :D], mode) + S.unsqueeze(0) + T.unsqueeze(1)
    mem = torch.relu(op._ln(pre, sd, p + ".proj_memory.1", dtype))
    if update_edge:
        up = mm(mem, op._w(sd, p + ".proj_edge.0.weight", dtype), mode) + op._w(sd, p + ".proj_edge.0.bias", dtype)
        up = torch.relu(op._ln(up, sd, p + ".proj_edge.1", dtype))
        edge = op._ln(edge + up, sd, p + ".norm_edge", dtype)
    w_in = op._w(sd, p + ".multihead_attn.in_proj_weight", dtype)
    b_in = op._w(sd, p + ".multihead_attn.in_proj_bias", dtype)
    q = F.linear(x, w_in[:D], b_in[:D]).view(n, 8, 16)
    # folded K: qk[j,h,:] = W_k[h]^T q[j,h] / 4 ; s[j,h,i] = qk[j,h,:] . mem[i,j,:]
    Wk = w_in[D:2 * D].view(8, 16, D)
    qk = torch.einsum("jhd,hdf->jhf", q, Wk) / 4.0
    if mode == "f32":
        s = torch.einsum("jhf,ijf->jhi", qk, mem)
    else:
        qh, ql = split(qk)
        mh, ml = split(mem)
        s = torch.einsum("jhf,ijf->jhi", qh, mh)
        if mode == "bf16x3":
            s = s + torch.einsum("jhf,ijf->jhi", qh, ml) + torch.einsum("jhf,ijf->jhi", ql, mh)
    pr = torch.softmax(s, dim=-1)
    # folded V: o = W_v,h (sum_i p mem) + b_v   (sum p.mem kept in fp32 here: the kernel runs it on the fp32 MFMA)
    mbar = torch.einsum("jhi,ijf->jhf", pr, mem)
    Wv = w_in[2 * D:].view(8, 16, D)
    o = (torch.einsum("jhf,hdf->jhd", mbar, Wv) + b_in[2 * D:].view(8, 16)).reshape(n, D)
    att = op._lin(o, sd, p + ".multihead_attn.out_proj", dtype)
    x1 = op._ln(x + att, sd, p + ".norm2", dtype)
    ff = op._lin(torch.relu(op._lin(x1, sd, p + ".linear1", dtype)), sd, p + ".linear2", dtype)
    return op._ln(x1 + ff, sd, p + ".norm3", dtype), edge


def run(a, l, seed, modes=("f32", "bf16x3", "bf16")):
    sd = formula_state_dict(as_torch=True)
    pb = predictor_batch(a, l, 1, seed=seed)
    tb = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else [torch.from_numpy(x) for x in v]) for k, v in pb.items()}
    c64, r64, v64 = op.forward(sd, tb, dtype=torch.float64)
    c32, r32, v32 = op.forward(sd, tb)
    print(f"a={a} l={l} N={a+l+1}: oracle fp32 vs fp64: reg {float((r32[0]-r64[0]).abs().max()):.2e} cls {float((c32[0]-c64[0]).abs().max()):.2e}")
    orig = op.fusion_layer
    for mode in modes:
        op.fusion_layer = lambda sd_, p, x, e, ue, dt, m=mode: fusion_layer(sd_, p, x, e, ue, m, dt)
        try:
            c, r, v = op.forward(sd, tb)
        finally:
            op.fusion_layer = orig
        print(f"  {mode:7s} vs fp32 oracle: reg[xy] {float((r[0][..., :2]-r32[0][..., :2]).abs().max()):.2e}  reg[all] {float((r[0]-r32[0]).abs().max()):.2e}"
              f"  vel {float((v[0]-v32[0]).abs().max()):.2e}  cls {float((c[0]-c32[0]).abs().max()):.2e}"
              f"   | vs fp64: reg {float((r[0]-r64[0]).abs().max()):.2e}")


if __name__ == "__main__":
    a = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    l = int(sys.argv[2]) if len(sys.argv) > 2 else 55
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 9
    run(a, l, seed)
