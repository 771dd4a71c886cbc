"""ORACLE (test infrastructure, never shipped as the product path).

CPU restatement of the reference scene predictor forward, written from the
formulas, as plain torch tensor ops in fp32 (or fp64 with ``dtype``).  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

Pinned against the imported reference (tests/test_oracle_vs_reference.py, build
container only) and against tests/golden/predictor_*.npz (everywhere).

Follows, function by function:
  rpe()            planners/mind/utils.py:193-242          (get_rpe/get_cos/get_sin)
  actor_net()      planners/mind/networks/network.py:47-61 + layers.py:55-60,175-188
  lane_net()       network.py:90-99,117-121
  fusion_layer()   network.py:165-232   (materialises memory[N,N,128] like the reference)
  fusion_net()     network.py:306-340
  scene_decoder()  network.py:483-556 (bezier branch), basis :449-464
  forward()        network.py:582-595
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

D = 128
K_MODES = 6
PRED_LEN = 60
N_ORDER = 7


def _w(sd, name, dtype):
    v = sd[name]
    if isinstance(v, np.ndarray):
        v = torch.from_numpy(v)
    return v.to(dtype)


def _ln(x, sd, prefix, dtype):
    return F.layer_norm(x, (x.shape[-1],), _w(sd, prefix + ".weight", dtype), _w(sd, prefix + ".bias", dtype), 1e-5)


def _lin(x, sd, prefix, dtype):
    return F.linear(x, _w(sd, prefix + ".weight", dtype), _w(sd, prefix + ".bias", dtype))


def _lin_ln_relu(x, sd, prefix, idx, dtype):
    return torch.relu(_ln(_lin(x, sd, f"{prefix}.{idx}", dtype), sd, f"{prefix}.{idx + 1}", dtype))


def rpe(ctrs, vecs):
    """[5, n, n] relative pose encoding (utils.py:193-212).  Row i / col j:
    v1 = vecs[j], v2 = vecs[i] for a1; v1 = vecs[j], v2 = ctrs[j]-ctrs[i] for a2."""
    d = ctrs.unsqueeze(0) - ctrs.unsqueeze(1)          # [i, j] = c_j - c_i
    dist = d.norm(dim=-1)
    v1 = vecs.unsqueeze(0).expand_as(d)                # v_j
    v2 = vecs.unsqueeze(1).expand_as(d)                # v_i

    def cs(a, b):
        den = a.norm(dim=-1) * b.norm(dim=-1) + 1e-10
        return ((a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) / den,
                (a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]) / den)

    c1, s1 = cs(v1, v2)
    c2, s2 = cs(v1, d)
    return torch.stack([c1, s1, c2, s2, dist * 2 / 100.0])


def _gn(x, sd, prefix, dtype):
    return F.group_norm(x, 1, _w(sd, prefix + ".weight", dtype), _w(sd, prefix + ".bias", dtype), 1e-5)


def _res1d(x, sd, p, stride, dtype):
    out = F.conv1d(x, _w(sd, p + ".conv1.weight", dtype), stride=stride, padding=1)
    out = torch.relu(_gn(out, sd, p + ".bn1", dtype))
    out = _gn(F.conv1d(out, _w(sd, p + ".conv2.weight", dtype), padding=1), sd, p + ".bn2", dtype)
    if (p + ".downsample.0.weight") in sd:
        x = _gn(F.conv1d(x, _w(sd, p + ".downsample.0.weight", dtype), stride=stride), sd, p + ".downsample.1", dtype)
    return torch.relu(out + x)


def actor_net(sd, actors, dtype=torch.float32):
    """[A,14,48] -> [A,128]."""
    out = actors.to(dtype)
    outs = []
    for g in range(4):
        out = _res1d(out, sd, f"actor_net.groups.{g}.0", 1 if g == 0 else 2, dtype)
        out = _res1d(out, sd, f"actor_net.groups.{g}.1", 1, dtype)
        outs.append(out)

    def lateral(g, x):
        return _gn(F.conv1d(x, _w(sd, f"actor_net.lateral.{g}.conv.weight", dtype), padding=1),
                   sd, f"actor_net.lateral.{g}.norm", dtype)

    out = lateral(3, outs[3])
    for g in (2, 1, 0):
        out = F.interpolate(out, scale_factor=2, mode="linear", align_corners=False)
        out = out + lateral(g, outs[g])
    return _res1d(out, sd, "actor_net.output", 1, dtype)[:, :, -1]


def _point_aggregate(x_inp, sd, p, dtype):
    x = _lin_ln_relu(_lin_ln_relu(x_inp, sd, p + ".fc1", 0, dtype), sd, p + ".fc1", 3, dtype)
    m = x.max(dim=1, keepdim=True).values
    cat = torch.cat([x, m.expand_as(x)], dim=-1)
    y = _lin_ln_relu(_lin_ln_relu(cat, sd, p + ".fc2", 0, dtype), sd, p + ".fc2", 3, dtype)
    return _ln(x_inp + y, sd, p + ".norm", dtype)


def lane_net(sd, feats, dtype=torch.float32):
    """[L,10,16] -> [L,128]."""
    x = _lin_ln_relu(feats.to(dtype), sd, "lane_net.proj", 0, dtype)
    x = _point_aggregate(x, sd, "lane_net.aggre1", dtype)
    x = _point_aggregate(x, sd, "lane_net.aggre2", dtype)
    return x.max(dim=1).values


def fusion_layer(sd, p, x, edge, update_edge, dtype):
    """One RelaFusionLayer (network.py:165-232). x [N,128], edge [N,N,128]."""
    n = x.shape[0]
    src = x.unsqueeze(0).expand(n, n, D)   # [i,j] = x[j]
    tar = x.unsqueeze(1).expand(n, n, D)   # [i,j] = x[i]
    mem = _lin_ln_relu(torch.cat([edge, src, tar], dim=-1), sd, p + ".proj_memory", 0, dtype)
    if update_edge:
        edge = _ln(edge + _lin_ln_relu(mem, sd, p + ".proj_edge", 0, dtype), sd, p + ".norm_edge", dtype)
    w_in = _w(sd, p + ".multihead_attn.in_proj_weight", dtype)
    b_in = _w(sd, p + ".multihead_attn.in_proj_bias", dtype)
    q = F.linear(x, w_in[:D], b_in[:D]).view(n, 8, 16)                      # [j,h,d]
    k = F.linear(mem, w_in[D:2 * D], b_in[D:2 * D]).view(n, n, 8, 16)       # [i,j,h,d]
    v = F.linear(mem, w_in[2 * D:], b_in[2 * D:]).view(n, n, 8, 16)
    s = torch.einsum("jhd,ijhd->jhi", q, k) / 4.0
    pr = torch.softmax(s, dim=-1)
    o = torch.einsum("jhi,ijhd->jhd", pr, v).reshape(n, D)
    att = _lin(o, sd, p + ".multihead_attn.out_proj", dtype)
    x1 = _ln(x + att, sd, p + ".norm2", dtype)
    ff = _lin(torch.relu(_lin(x1, sd, p + ".linear1", dtype)), sd, p + ".linear2", dtype)
    x2 = _ln(x1 + ff, sd, p + ".norm3", dtype)
    return x2, edge


def fusion_net(sd, actor_feat, lane_feat, scene_rpe, dtype=torch.float32, taps=None):
    """One scene. actor_feat [a,128] (ActorNet out), lane_feat [l,128], scene_rpe [5,n,n].
    Returns (actors [a,128], lanes [l,128], cls [128])."""
    a = actor_feat.shape[0]
    act = _lin_ln_relu(actor_feat, sd, "fusion_net.proj_actor", 0, dtype)
    lan = _lin_ln_relu(lane_feat, sd, "fusion_net.proj_lane", 0, dtype)
    tokens = torch.cat([act, lan, torch.zeros(1, D, dtype=dtype)], dim=0)
    n = tokens.shape[0]
    e = _lin_ln_relu(scene_rpe.to(dtype).permute(1, 2, 0), sd, "fusion_net.proj_rpe_scene", 0, dtype)
    edge = torch.zeros(n, n, D, dtype=dtype)
    edge[:n - 1, :n - 1] = e
    x = tokens
    for i in range(6):
        x, edge = fusion_layer(sd, f"fusion_net.fuse_scene.fusion.{i}", x, edge, i != 5, dtype)
        if taps is not None:
            taps.append((x.clone(), edge.clone() if i != 5 else None))
    return x[:a], x[a:-1], x[-1]


def bezier_T(dtype=torch.float32):
    ts = np.linspace(0.0, 1.0, PRED_LEN, endpoint=True)
    T = np.array([math.comb(N_ORDER, i) * (1.0 - ts) ** (N_ORDER - i) * ts ** i for i in range(N_ORDER + 1)]).T
    Tp = np.array([N_ORDER * math.comb(N_ORDER - 1, i) * (1.0 - ts) ** (N_ORDER - 1 - i) * ts ** i
                   for i in range(N_ORDER)]).T
    # reference casts the float64 basis to fp32 (torch.Tensor(...)), Q21
    return torch.from_numpy(T).float().to(dtype), torch.from_numpy(Tp).float().to(dtype)


def _mha_self(sd, p, x, nhead, dtype):
    """x [S, B, 128] self attention (post-norm encoder layer's attention)."""
    S, B, _ = x.shape
    w_in = _w(sd, p + ".in_proj_weight", dtype)
    b_in = _w(sd, p + ".in_proj_bias", dtype)
    hd = D // nhead
    q = F.linear(x, w_in[:D], b_in[:D]).view(S, B, nhead, hd)
    k = F.linear(x, w_in[D:2 * D], b_in[D:2 * D]).view(S, B, nhead, hd)
    v = F.linear(x, w_in[2 * D:], b_in[2 * D:]).view(S, B, nhead, hd)
    s = torch.einsum("sbhd,tbhd->bhst", q, k) / math.sqrt(hd)
    pr = torch.softmax(s, dim=-1)
    o = torch.einsum("bhst,tbhd->sbhd", pr, v).reshape(S, B, D)
    return _lin(o, sd, p + ".out_proj", dtype)


def scene_decoder(sd, cls_tok, actors, tgt_feat, tgt_rpe, dtype=torch.float32, taps=None):
    """One scene: cls_tok [128], actors [a,128], tgt_feat [128], tgt_rpe [20].
    Returns cls [1,6], reg [a,6,60,5], vel [a,6,60,2]."""
    a = actors.shape[0]
    tr = _lin_ln_relu(tgt_rpe.to(dtype).view(1, 20), sd, "pred_scene.proj_rpe", 0, dtype)
    tgt = torch.cat([tgt_feat.view(1, D), tr], dim=-1)
    tgt = _lin_ln_relu(_lin_ln_relu(tgt, sd, "pred_scene.proj_tgt", 0, dtype), sd, "pred_scene.proj_tgt", 3, dtype)

    def mm_proj(x, nm):
        y = _lin_ln_relu(_lin_ln_relu(x, sd, f"pred_scene.{nm}", 0, dtype), sd, f"pred_scene.{nm}", 3, dtype)
        return y.view(-1, K_MODES, D).permute(1, 0, 2)   # [6, rows, 128]

    C = mm_proj(cls_tok.view(1, D), "ctx_proj")            # [6,1,128]
    for i in range(2):
        p = f"pred_scene.ctx_sat.layers.{i}"
        C = _ln(C + _mha_self(sd, p + ".self_attn", C, 4, dtype), sd, p + ".norm1", dtype)
        ff = _lin(torch.relu(_lin(C, sd, p + ".linear1", dtype)), sd, p + ".linear2", dtype)
        C = _ln(C + ff, sd, p + ".norm2", dtype)
    if taps is not None:
        taps["tgt_emb"] = tgt.clone()
        taps["cmode"] = C.clone()
    A = mm_proj(actors, "actor_proj")                      # [6,a,128]
    E = C + A
    E = torch.cat([E[:1] + tgt.view(1, 1, D), E[1:]], dim=0)   # target embedding on mode 0 only (Q6)

    def head(x, nm):
        h = _lin_ln_relu(_lin_ln_relu(x, sd, f"pred_scene.{nm}", 0, dtype), sd, f"pred_scene.{nm}", 3, dtype)
        return _lin(h, sd, f"pred_scene.{nm}.6", dtype)

    logits = head(C, "cls").view(K_MODES, -1).permute(1, 0)    # [1,6]
    cls = torch.softmax(logits, dim=1)
    param = head(E, "reg").view(K_MODES, a, N_ORDER + 1, 5).permute(1, 0, 2, 3)   # [a,6,8,5]
    T, Tp = bezier_T(dtype)
    pos = torch.matmul(T, param[..., :2])
    vel = torch.matmul(Tp, torch.diff(param[..., :2], dim=2)) / (PRED_LEN * 0.1)
    cov = torch.exp(torch.matmul(T, param[..., 2:]))
    return cls, torch.cat([pos, cov], dim=-1), vel


def forward(sd, batch, dtype=torch.float32, taps=None):
    """batch: dict with ACTORS [A,14,48], ACTOR_IDCS list, LANES [L,10,16], LANE_IDCS list,
    RPE list of [5,n,n] (or CTRS/VECS lists), TGT_NODES [B,10,16], TGT_RPE [B,20].
    Returns (cls list[B x [1,6]], reg list[B x [a,6,60,5]], vel list[B x [a,6,60,2]])."""
    with torch.no_grad():
        act = actor_net(sd, batch["ACTORS"], dtype)
        lan = lane_net(sd, batch["LANES"], dtype)
        tgt = lane_net(sd, batch["TGT_NODES"], dtype)
        if taps is not None:
            taps["actor_net"] = act.clone()
            taps["lane_net"] = lan.clone()
            taps["fusion"] = []
        res_cls, res_reg, res_vel = [], [], []
        for b, (ai, li) in enumerate(zip(batch["ACTOR_IDCS"], batch["LANE_IDCS"])):
            if "RPE" in batch:
                r = batch["RPE"][b]
                r = r["scene"] if isinstance(r, dict) else r
            else:
                r = rpe(batch["CTRS"][b].to(dtype), batch["VECS"][b].to(dtype))
            ft = [] if taps is not None else None
            a_new, _, c_new = fusion_net(sd, act[ai], lan[li], r, dtype, ft)
            if taps is not None:
                taps["fusion"].append(ft)
            dt = {} if taps is not None else None
            c, rg, v = scene_decoder(sd, c_new, a_new, tgt[b], batch["TGT_RPE"][b], dtype, dt)
            if taps is not None:
                taps.setdefault("dec", []).append(dt)
                taps["tgt_feat"] = tgt.clone()
            res_cls.append(c)
            res_reg.append(rg)
            res_vel.append(v)
        return res_cls, res_reg, res_vel
