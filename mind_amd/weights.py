"""Predictor weight tables: the state_dict layout and formula-generated weights.

The reference checkpoint (planners/mind/check_points/*.tar) is not available,
so tests/bench use *formula weights*: a counter-based PRNG keyed by the tensor
name, bit-identical wherever numpy runs.  A real
checkpoint's ``ckpt["state_dict"]`` (reference planners/mind/planner.py:46-47)
goes through exactly the same ``state_dict -> pack`` path.

``state_dict_spec()`` enumerates the 328 tensors of the reference
``ScenePredNet`` (planners/mind/networks/network.py:559-580) in its
``state_dict()`` order; tests pin names/shapes against the imported reference.
"""
from collections import OrderedDict

import numpy as np

D = 128          # hidden size (net_cfg.py: d_actor = d_lane = d_embed = d_rpe)
N_MODES = 6      # g_num_modes
N_FUSION = 6     # n_scene_layer
N_HEAD = 8       # n_scene_head
PRED_LEN = 60    # g_pred_len
N_ORDER = 7      # bezier order


def _seq_linear_ln(prefix, idx, n_out, n_in):
    """nn.Sequential(... Linear @idx, LayerNorm @idx+1 ...)."""
    return [(f"{prefix}.{idx}.weight", (n_out, n_in)), (f"{prefix}.{idx}.bias", (n_out,)),
            (f"{prefix}.{idx + 1}.weight", (n_out,)), (f"{prefix}.{idx + 1}.bias", (n_out,))]


def _res1d(prefix, n_in, n_out, stride):
    out = [(f"{prefix}.conv1.weight", (n_out, n_in, 3)), (f"{prefix}.conv2.weight", (n_out, n_out, 3)),
           (f"{prefix}.bn1.weight", (n_out,)), (f"{prefix}.bn1.bias", (n_out,)),
           (f"{prefix}.bn2.weight", (n_out,)), (f"{prefix}.bn2.bias", (n_out,))]
    if stride != 1 or n_in != n_out:
        out += [(f"{prefix}.downsample.0.weight", (n_out, n_in, 1)),
                (f"{prefix}.downsample.1.weight", (n_out,)), (f"{prefix}.downsample.1.bias", (n_out,))]
    return out


def state_dict_spec():
    """[(name, shape)] in reference state_dict order."""
    s = []
    # actor_net (network.py:12-45)
    n_in = 14
    chans = [32, 64, 128, 256]
    for g, c in enumerate(chans):
        s += _res1d(f"actor_net.groups.{g}.0", n_in, c, 1 if g == 0 else 2)
        s += _res1d(f"actor_net.groups.{g}.1", c, c, 1)
        n_in = c
    for g, c in enumerate(chans):
        s += [(f"actor_net.lateral.{g}.conv.weight", (D, c, 3)),
              (f"actor_net.lateral.{g}.norm.weight", (D,)), (f"actor_net.lateral.{g}.norm.bias", (D,))]
    s += _res1d("actor_net.output", D, D, 1)
    # lane_net (network.py:102-114)
    s += _seq_linear_ln("lane_net.proj", 0, D, 16)
    for blk in ("aggre1", "aggre2"):
        p = f"lane_net.{blk}"
        s += _seq_linear_ln(f"{p}.fc1", 0, D, D) + _seq_linear_ln(f"{p}.fc1", 3, D, D)
        s += _seq_linear_ln(f"{p}.fc2", 0, D, 2 * D) + _seq_linear_ln(f"{p}.fc2", 3, D, D)
        s += [(f"{p}.norm.weight", (D,)), (f"{p}.norm.bias", (D,))]
    # fusion_net (network.py:271-304, 124-163)
    s += _seq_linear_ln("fusion_net.proj_actor", 0, D, D)
    s += _seq_linear_ln("fusion_net.proj_lane", 0, D, D)
    s += _seq_linear_ln("fusion_net.proj_rpe_scene", 0, D, 5)
    for i in range(N_FUSION):
        p = f"fusion_net.fuse_scene.fusion.{i}"
        s += _seq_linear_ln(f"{p}.proj_memory", 0, D, 3 * D)
        if i != N_FUSION - 1:
            s += _seq_linear_ln(f"{p}.proj_edge", 0, D, D)
            s += [(f"{p}.norm_edge.weight", (D,)), (f"{p}.norm_edge.bias", (D,))]
        s += [(f"{p}.multihead_attn.in_proj_weight", (3 * D, D)), (f"{p}.multihead_attn.in_proj_bias", (3 * D,)),
              (f"{p}.multihead_attn.out_proj.weight", (D, D)), (f"{p}.multihead_attn.out_proj.bias", (D,)),
              (f"{p}.linear1.weight", (2 * D, D)), (f"{p}.linear1.bias", (2 * D,)),
              (f"{p}.linear2.weight", (D, 2 * D)), (f"{p}.linear2.bias", (D,)),
              (f"{p}.norm2.weight", (D,)), (f"{p}.norm2.bias", (D,)),
              (f"{p}.norm3.weight", (D,)), (f"{p}.norm3.bias", (D,))]
    # pred_scene (network.py:343-421)
    dim_mm, dim_inter = D * N_MODES, D * N_MODES // 2
    for nm in ("actor_proj", "ctx_proj"):
        s += _seq_linear_ln(f"pred_scene.{nm}", 0, dim_inter, D)
        s += _seq_linear_ln(f"pred_scene.{nm}", 3, dim_mm, dim_inter)
    for i in range(2):
        p = f"pred_scene.ctx_sat.layers.{i}"
        s += [(f"{p}.self_attn.in_proj_weight", (3 * D, D)), (f"{p}.self_attn.in_proj_bias", (3 * D,)),
              (f"{p}.self_attn.out_proj.weight", (D, D)), (f"{p}.self_attn.out_proj.bias", (D,)),
              (f"{p}.linear1.weight", (12 * D, D)), (f"{p}.linear1.bias", (12 * D,)),
              (f"{p}.linear2.weight", (D, 12 * D)), (f"{p}.linear2.bias", (D,)),
              (f"{p}.norm1.weight", (D,)), (f"{p}.norm1.bias", (D,)),
              (f"{p}.norm2.weight", (D,)), (f"{p}.norm2.bias", (D,))]
    s += _seq_linear_ln("pred_scene.proj_rpe", 0, D, 20)
    s += _seq_linear_ln("pred_scene.proj_tgt", 0, D, 2 * D) + _seq_linear_ln("pred_scene.proj_tgt", 3, D, D)
    for nm, n_last in (("cls", 1), ("reg", (N_ORDER + 1) * 5)):
        s += _seq_linear_ln(f"pred_scene.{nm}", 0, D, D) + _seq_linear_ln(f"pred_scene.{nm}", 3, D, D)
        s += [(f"pred_scene.{nm}.6.weight", (n_last, D)), (f"pred_scene.{nm}.6.bias", (n_last,))]
    return s


# --------------------------------------------------------------------------
# counter-based PRNG (splitmix64 over fnv1a64(name) + index)
# --------------------------------------------------------------------------
_M64 = (1 << 64) - 1


def fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & _M64
    return h


def splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser on uint64 (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform_pm1(key: str, n: int, seed: int = 0) -> np.ndarray:
    """n doubles in [-1, 1), element idx = f(seed, key, idx) only."""
    with np.errstate(over="ignore"):
        base = np.uint64((fnv1a64(key) + (seed * 0x9E3779B97F4A7C15)) & _M64)
        ctr = base + np.arange(n, dtype=np.uint64)
    z = splitmix64(ctr)
    return 2.0 * ((z >> np.uint64(11)).astype(np.float64) / float(1 << 53)) - 1.0


def formula_state_dict(seed: int = 20240121, as_torch: bool = False, variant: str = None):
    """Formula weights for every tensor of ``state_dict_spec()`` (fp32).

    1-D ``*.weight`` (all norm gains) <- 1 + 0.1u; every ``*bias`` <- 0.05u;
    matrices / conv kernels <- u * sqrt(3 / fan_in).

    ``variant="branching"`` re-scales the decoder heads so that the AIME tree BRANCHES on the reference's recorded demo
    scenes the way a trained checkpoint makes it branch (the plain formula weights predict six near-identical, static
    modes that merge into one node): the Bezier xy control points get 6 x the hidden-state gain (modes separate by metres,
    distinct topology signatures) on top of a 2 m/s forward motion, the sigma control points grow from 0.2 to 8 over the
    horizon (the branch-time ratio test fires), the classification logits get 3 x the gain (a clear mode ranking).  With
    it the reference itself expands 6 scenes in two AIME rounds per plan on demo_1 and keeps 3-5 modes per round on all
    four scenes (tests/golden/gen_golden.py demo_branch).
    """
    sd = OrderedDict()
    for name, shape in state_dict_spec():
        n = int(np.prod(shape))
        u = uniform_pm1(name, n, seed)
        if name.endswith("bias"):
            v = 0.05 * u
        elif len(shape) == 1:
            v = 1.0 + 0.1 * u
        else:
            fan_in = int(np.prod(shape[1:]))
            v = u * np.sqrt(3.0 / fan_in)
        sd[name] = v.astype(np.float32).reshape(shape)
    if variant == "branching":
        W = sd["pred_scene.reg.6.weight"].reshape(8, 5, D).copy()
        b = sd["pred_scene.reg.6.bias"].reshape(8, 5).copy()
        W[:, 0:2] *= np.float32(6.0)
        for k in range(8):
            b[k, 0] = 2.0 * 6.0 * k / 7.0
            b[k, 2] = np.log(0.2) + (np.log(8.0) - np.log(0.2)) * k / 7.0
            b[k, 3] = b[k, 2] + np.log(0.8)
        sd["pred_scene.reg.6.weight"] = W.reshape(40, D).astype(np.float32)
        sd["pred_scene.reg.6.bias"] = b.reshape(40).astype(np.float32)
        sd["pred_scene.cls.6.weight"] = (sd["pred_scene.cls.6.weight"] * np.float32(3.0)).astype(np.float32)
    elif variant is not None:
        raise ValueError(f"unknown formula weight variant {variant!r}")
    if as_torch:
        import torch
        return OrderedDict((k, torch.from_numpy(v.copy())) for k, v in sd.items())
    return sd
