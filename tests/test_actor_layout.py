"""CPU check of the MFMA ActorNet's index algebra (mind_amd/csrc/actor_mfma_kernels.hip) -- no GPU needed.

The lane-level numpy model of v_mfma_f32_16x16x32_bf16 from test_pair_layout.py is driven the way am_conv drives the hardware:
A operand = the library's own host-side fragment packing of a Conv1d weight (mind_debug_pack_conv_frag, the function
mind_weights_load uses), B operand = 8 consecutive input channels of one tap read from the time-major image of three bf16
planes the producing layer wrote (zeros for the conv padding, for time columns past Tout and for k-slots past ksz * Cin_pad),
tiles enumerated time-fastest, the C/D lanes written back as 4 consecutive channels of one time column.  The result must equal torch's conv1d of the
same bf16-split values for every layer geometry of the network (network.py:20-61: 14->32 ... 256->256, strides 1 / 2, taps 3 / 1)."""
import ctypes as C

import numpy as np
import pytest
import torch

from mind_amd import _lib
from tests.test_pair_layout import bits16, from16, mfma_16x16x32, split


@pytest.fixture(scope="module")
def lib():
    return _lib.load()


def pack_conv(lib, W):
    co, ci, ksz = W.shape
    out = np.zeros(6 * co * 1024, np.uint32)
    Wc = np.ascontiguousarray(W, np.float32)
    n = lib.mind_debug_pack_conv_frag(Wc.ctypes.data_as(C.POINTER(C.c_float)), co, ci, ksz,
                                      out.ctypes.data_as(C.POINTER(C.c_uint32)), out.size)
    assert n > 0
    return out[:n]


def frag(packed, ks_total, mt, ks, part):
    d = packed.reshape(-1, ks_total, 3, 64, 4)[mt, ks, part]
    out = np.zeros((64, 8), np.float32)
    out[:, 0::2] = from16((d & 0xffff).astype(np.uint16))
    out[:, 1::2] = from16((d >> 16).astype(np.uint16))
    return out


def split3(x):
    h, r = split(x)
    h2, r2 = split(x - h)
    return h, h2, r2


def am_conv_model(lib, X, W, stride):
    """X [Cin][Tin], W [Cout][Cin][ksz] -> [Cout][Tout] the way the kernel computes it (three-way split, six products)"""
    co, ci, ksz = W.shape
    Tin = X.shape[1]
    Tout = Tin // stride
    cp = 16
    while cp < ci:
        cp *= 2
    lgc = cp.bit_length() - 1
    RSD = 3 * cp // 2 + 4                               # AM_RSD: three bf16 planes of cp channels + 16 bytes pad, in dwords
    plane = cp // 2
    pad = (ksz - 1) // 2
    KS = (ksz * cp + 31) // 32
    # the split image the producing layer writes (am_store_split): dword = channels (2 j, 2 j + 1) of one plane
    Xp = np.zeros((cp, Tin), np.float32)
    Xp[:ci] = X
    img = np.full((Tin, RSD), 0xFFFFFFFF, np.uint32)     # pad dwords (NaN patterns) are never read
    for pi, part in enumerate(split3(Xp)):
        b = bits16(part.T)                              # [Tin][cp] bf16 bit patterns
        img[:, pi * plane:(pi + 1) * plane] = b[:, 0::2].astype(np.uint32) | (b[:, 1::2].astype(np.uint32) << 16)
    img = img.reshape(-1)
    packed = pack_conv(lib, W)
    assert packed.size == (co // 16) * KS * 768
    ntt = (Tout + 15) >> 4
    tiles = (co >> 4) * ntt
    out = np.full((Tout, co + 4), np.nan)
    assert tiles <= 32                                  # AM_MAXT * AM_WAVES
    for ti in range(tiles):
        mt, nt = divmod(ti, ntt)
        acc = np.zeros((64, 4))
        for ks in range(KS):
            Bp = np.zeros((3, 64, 8), np.float32)
            for lane in range(64):
                r, q = lane & 15, lane >> 4
                t = nt * 16 + r
                k0 = ks * 32 + q * 8
                dk, c0 = k0 >> lgc, k0 & (cp - 1)
                row = t * stride + dk - pad
                if dk < ksz and t < Tout and 0 <= row < Tin:
                    for pi in range(3):
                        d = img[row * RSD + pi * plane + (c0 >> 1):row * RSD + pi * plane + (c0 >> 1) + 4]
                        Bp[pi, lane, 0::2] = from16((d & 0xffff).astype(np.uint16))
                        Bp[pi, lane, 1::2] = from16((d >> 16).astype(np.uint16))
            bh, bm, bl = Bp
            ah, am, al = (frag(packed, KS, mt, ks, p) for p in range(3))
            for x, y in ((ah, bh), (al, bh), (ah, bl), (am, bm), (am, bh), (ah, bm)):
                acc = mfma_16x16x32(x, y, acc)
        for lane in range(64):
            r, q = lane & 15, lane >> 4
            t, c = nt * 16 + r, mt * 16 + q * 4
            if t < Tout:
                out[t, c:c + 4] = acc[lane]
    assert not np.isnan(out[:, :co]).any()
    return out[:, :co].T


GEOMS = [(14, 32, 3, 1, 48), (14, 32, 1, 1, 48), (32, 32, 3, 1, 48), (32, 64, 3, 2, 48), (32, 64, 1, 2, 48), (64, 64, 3, 1, 24),
         (64, 128, 3, 2, 24), (128, 128, 3, 1, 12), (128, 256, 3, 2, 12), (128, 256, 1, 2, 12), (256, 256, 3, 1, 6),
         (256, 128, 3, 1, 6), (128, 128, 3, 1, 48)]


@pytest.mark.parametrize("ci,co,ksz,stride,Tin", GEOMS)
def test_conv_as_mfma_gemm(lib, ci, co, ksz, stride, Tin):
    rng = np.random.default_rng(ci * 1000 + co + ksz + stride)
    X = rng.standard_normal((ci, Tin)).astype(np.float32)
    W = (rng.standard_normal((co, ci, ksz)) / np.sqrt(ci * ksz)).astype(np.float32)
    got = am_conv_model(lib, X, W, stride)
    (Wh, Wm, Wl), (Xh, Xm, Xl) = split3(W), split3(X)
    assert np.array_equal((Wh.astype(np.float64) + Wm + Wl).astype(np.float32), W)       # the three parts are exact

    def conv(w, x):
        return torch.nn.functional.conv1d(torch.from_numpy(x.astype(np.float64))[None], torch.from_numpy(w.astype(np.float64)),
                                          stride=stride, padding=(ksz - 1) // 2)[0].numpy()
    want = conv(Wh, Xh) + conv(Wl, Xh) + conv(Wh, Xl) + conv(Wm, Xm) + conv(Wm, Xh) + conv(Wh, Xm)
    assert got.shape == want.shape
    assert np.abs(got - want).max() < 1e-9
    exact = conv(W, X)
    assert np.abs(got - exact).max() < 2e-6 * max(1.0, np.abs(exact).max())      # the dropped products are ~2^-24
