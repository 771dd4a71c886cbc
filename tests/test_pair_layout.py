"""CPU check of the bf16 pair kernel's layout algebra (mind_amd/csrc/pair_bf16_kernels.hip) -- no GPU needed.

A lane-level numpy model of v_mfma_f32_16x16x32_bf16 (A: lane l holds row l & 15, k-slots 8 (l >> 4) .. +7; B: column l & 15,
same k-slots; C/D: column l & 15, rows 4 (l >> 4) + reg) is driven exactly the way the kernel drives the hardware: weight
fragments from the library's own host-side packer (mind_debug_pack_bfrag, the function mind_weights_load uses), activations
split into bf16 hi / lo B operands in the chained slot order, the memory tile transposed through the XOR-swizzled staging
image, the probabilities duplicated over the (hi, lo) k-slots.  The results must equal the plain matrix products of the same
bf16-split values -- i.e. every index map in the kernel and in the packer is consistent."""
import ctypes as C

import numpy as np
import pytest
import torch

from mind_amd import _lib


def bf16_rne(x):
    return torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


def split(x):
    hi = bf16_rne(x)
    return hi, bf16_rne(x - hi)


def bits16(x):
    """bf16-representable float32 array -> uint16 bit patterns"""
    return (np.ascontiguousarray(x, np.float32).view(np.uint32) >> 16).astype(np.uint16)


def b16s(x):
    """scalar version of bits16"""
    return int(bits16(np.asarray([x]))[0])


def from16(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


def mfma_16x16x32(A, B, Cacc):
    """A, B: [64 lanes][8] float (bf16 values), Cacc: [64][4].  D = A.B + C in the hardware's lane layouts."""
    Am = np.zeros((16, 32), np.float64)
    Bm = np.zeros((32, 16), np.float64)
    for l in range(64):
        for i in range(8):
            Am[l & 15, 8 * (l >> 4) + i] = A[l, i]
            Bm[8 * (l >> 4) + i, l & 15] = B[l, i]
    D = Am @ Bm
    out = Cacc.astype(np.float64).copy()
    for l in range(64):
        for r in range(4):
            out[l, r] += D[4 * (l >> 4) + r, l & 15]
    return out


def frag_from_packed(packed, part, ob, g):
    """[64][8] float values of the (part, ob, g) fragment of mind_debug_pack_bfrag's output"""
    d = packed.reshape(2, 8, 4, 64, 4)[part, ob, g]               # [lane][dword]
    lo16 = (d & 0xffff).astype(np.uint16)
    hi16 = (d >> 16).astype(np.uint16)
    out = np.zeros((64, 8), np.float32)
    out[:, 0::2] = from16(lo16)
    out[:, 1::2] = from16(hi16)
    return out


def lane_chunks(X):
    """X [16 pairs][128 features] -> the kernel's register view [64 lanes][8 chunks][4]: chunk blk of lane (p, q) =
    features 16 blk + 4 q + (0..3) of pair p"""
    out = np.zeros((64, 8, 4), np.float32)
    for l in range(64):
        p, q = l & 15, l >> 4
        for b in range(8):
            out[l, b] = X[p, 16 * b + 4 * q:16 * b + 4 * q + 4]
    return out


def b_operands(chunks):
    """split_frag: [64][8][4] -> hi, lo [4 k-groups][64][8]"""
    hi = np.zeros((4, 64, 8), np.float32)
    lo = np.zeros((4, 64, 8), np.float32)
    for g in range(4):
        for c in range(2):
            h, l_ = split(chunks[:, 2 * g + c, :])
            hi[g][:, 4 * c:4 * c + 4] = h
            lo[g][:, 4 * c:4 * c + 4] = l_
    return hi, lo


@pytest.fixture(scope="module")
def lib():
    return _lib.load()


def pack(lib, W, stride):
    out = np.zeros(16384, np.uint32)
    Wc = np.ascontiguousarray(W, np.float32)
    rc = lib.mind_debug_pack_bfrag(Wc.ctypes.data_as(C.POINTER(C.c_float)), stride, out.ctypes.data_as(C.POINTER(C.c_uint32)))
    assert rc == 0
    return out


@pytest.mark.parametrize("stride,col0", [(128, 0), (384, 0)])
def test_chained_gemm_layout(lib, stride, col0):
    """gemm_bf: acc[ob] += W_lo.B_hi + W_hi.B_hi + W_hi.B_lo over the four k-groups reproduces (W x)[feature, pair] in the
    C/D register layout -- which is again the chunk layout the next GEMM splits into its B operand."""
    rng = np.random.default_rng(7)
    Wfull = rng.standard_normal((128, stride)).astype(np.float32)
    W = Wfull[:, col0:col0 + 128]
    X = rng.standard_normal((16, 128)).astype(np.float32)           # [pair][feature]
    packed = pack(lib, Wfull, stride)
    bhi, blo = b_operands(lane_chunks(X))
    acc = np.zeros((8, 64, 4))
    for g in range(4):
        for ob in range(8):
            a_hi, a_lo = frag_from_packed(packed, 0, ob, g), frag_from_packed(packed, 1, ob, g)
            acc[ob] = mfma_16x16x32(a_lo, bhi[g], acc[ob])
            acc[ob] = mfma_16x16x32(a_hi, bhi[g], acc[ob])
            acc[ob] = mfma_16x16x32(a_hi, blo[g], acc[ob])
    Wh, Wl = split(W)
    Xh, Xl = split(X)
    want = (Wl.astype(np.float64) @ Xh.T + Wh.astype(np.float64) @ Xh.T + Wh.astype(np.float64) @ Xl.T)    # [out feature][pair]
    got = np.zeros((128, 16))
    for l in range(64):
        p, q = l & 15, l >> 4
        for ob in range(8):
            got[16 * ob + 4 * q:16 * ob + 4 * q + 4, p] = acc[ob][l]
    assert np.abs(got - want).max() < 1e-9
    # and the split error itself is of the advertised size: ~2^-16 of the fp32 product sums
    exact = W.astype(np.float64) @ X.T.astype(np.float64)
    assert np.abs(got - exact).max() < 2e-4 * np.abs(exact).max()
    assert np.abs(got - exact).max() > 0


def test_query_fragment_order_and_scores(lib):
    """k_token's bf16 hi / lo layout of the folded query ([part][k-group][row = head * 4 + q][slot]) read as the A operand
    (rows = heads, 8 of 16 used) against the memory tile's B operand gives scores[head, pair]."""
    rng = np.random.default_rng(3)
    QK = rng.standard_normal((8, 128)).astype(np.float32)           # [head][feature]
    M = rng.standard_normal((16, 128)).astype(np.float32)           # memory tile [pair][feature]
    # k_token: value of (head hd, feature col) -> u16 index ((g * 32) + hd * 4 + qq) * 8 + i (+ 1024 for the lo part)
    qs = np.zeros(2048, np.uint16)
    qh, ql = split(QK)
    for hd in range(8):
        for col in range(128):
            g, qq, i = col >> 5, (col >> 2) & 3, 4 * ((col >> 4) & 1) + (col & 3)
            idx = (g * 32 + hd * 4 + qq) * 8 + i
            qs[idx] = b16s(qh[hd, col])
            qs[1024 + idx] = b16s(ql[hd, col])
    qs32 = qs.view(np.uint32)                                        # dwords, as the kernel reads them
    mhi, mlo = b_operands(lane_chunks(M))
    acc = np.zeros((64, 4))
    for g in range(4):
        A = [np.zeros((64, 8), np.float32), np.zeros((64, 8), np.float32)]
        for l in range(64):
            if (l & 15) < 8:
                for part in range(2):
                    base = (((l & 7) * 4 + (l >> 4)) * 4) + (part * 4 + g) * 128          # qs + g * 128 (+ 4 * 128 for lo)
                    d = qs32[base:base + 4]
                    A[part][l, 0::2] = from16((d & 0xffff).astype(np.uint16))
                    A[part][l, 1::2] = from16((d >> 16).astype(np.uint16))
        acc = mfma_16x16x32(A[0], mhi[g], acc)
        acc = mfma_16x16x32(A[0], mlo[g], acc)
        acc = mfma_16x16x32(A[1], mhi[g], acc)
    Mh, Ml = split(M)
    want = qh.astype(np.float64) @ Mh.T + qh.astype(np.float64) @ Ml.T + ql.astype(np.float64) @ Mh.T     # [head][pair]
    for l in range(64):
        p, q = l & 15, l >> 4
        for r in range(4):
            head = 4 * q + r
            assert abs(acc[l, r] - (want[head, p] if head < 8 else 0.0)) < 1e-9


def test_transposed_staging_and_probability_operand():
    """sum_pairs p[head][pair] mem[pair][f] on the MFMA: the memory tile goes through the XOR-swizzled [feature][pair]
    staging image as (hi | lo << 16) dwords, one 32-feature quarter (= k-group) at a time; the A operand duplicates the
    probabilities over the (hi, lo) k-slots.  Checks the addresses (every dword written once, read where expected) and the
    result in the accumulator layout [blk][r] = head 4 q + r, feature 16 blk + (lane & 15)."""
    rng = np.random.default_rng(11)
    M = rng.standard_normal((16, 128)).astype(np.float32)
    P = rng.random((8, 16)).astype(np.float32)                       # [head][pair]
    mhi, mlo = b_operands(lane_chunks(M))
    cq = [0, 2, 3, 1]
    mbar = np.zeros((8, 64, 4))
    # A operand from ptab[head][pair]
    pah = np.zeros((64, 8), np.float32)
    pal = np.zeros((64, 8), np.float32)
    for l in range(64):
        lp, lq = l & 15, l >> 4
        pv = P[lp & 7, 4 * lq:4 * lq + 4] if lp < 8 else np.zeros(4, np.float32)
        h, lo = split(pv)
        for d in range(4):
            pah[l, 2 * d] = pah[l, 2 * d + 1] = h[d]
            pal[l, 2 * d] = pal[l, 2 * d + 1] = lo[d]
    for g in range(4):
        stage = np.full(512, 0xdeadbeef, np.uint32)
        written = np.zeros(512, int)
        for l in range(64):
            p, q = l & 15, l >> 4
            tw_base = (4 * q) * 16 + (((p >> 2) ^ cq[q]) * 4) + (p & 3)
            for b2 in range(2):
                for rr in range(2):
                    X = (b16s(mhi[g][l, 4 * b2 + 2 * rr]) | (b16s(mhi[g][l, 4 * b2 + 2 * rr + 1]) << 16))
                    Y = (b16s(mlo[g][l, 4 * b2 + 2 * rr]) | (b16s(mlo[g][l, 4 * b2 + 2 * rr + 1]) << 16))
                    a0 = tw_base + (16 * b2 + 2 * rr) * 16
                    a1 = tw_base + (16 * b2 + 2 * rr + 1) * 16
                    stage[a0] = (X & 0xffff) | ((Y << 16) & 0xffffffff)
                    stage[a1] = (X >> 16) | (Y & 0xffff0000)
                    written[a0] += 1
                    written[a1] += 1
        assert (written == 1).all()                                  # a permutation of the 512 dwords
        for b2 in range(2):
            Bf = np.zeros((64, 8), np.float32)
            for l in range(64):
                p, q = l & 15, l >> 4
                tr_base = p * 16 + ((q ^ cq[(p >> 2) & 3]) * 4)
                d = stage[tr_base + 256 * b2:tr_base + 256 * b2 + 4]
                Bf[l, 0::2] = from16((d & 0xffff).astype(np.uint16))
                Bf[l, 1::2] = from16((d >> 16).astype(np.uint16))
            mbar[2 * g + b2] = mfma_16x16x32(pah, Bf, mbar[2 * g + b2])
            mbar[2 * g + b2] = mfma_16x16x32(pal, Bf, mbar[2 * g + b2])
    Mh, Ml = split(M)
    Ph, Pl = split(P)
    want = (Ph.astype(np.float64) + Pl) @ (Mh.astype(np.float64) + Ml)         # [head][feature]
    for l in range(64):
        n, q = l & 15, l >> 4
        for blk in range(8):
            for r in range(4):
                head = 4 * q + r
                assert abs(mbar[blk][l, r] - (want[head, 16 * blk + n] if head < 8 else 0.0)) < 1e-9


def test_staging_swizzle_is_bank_conflict_free():
    """the ds_read_b128 pass (four 16-lane groups per MI355X_MICROARCH.md) touches 16 distinct 16-byte bank groups; the
    ds_write_b32 pass is at most 2-way (free for 4-byte stores)."""
    cq = [0, 2, 3, 1]
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[l + 32 for l in g] for g in groups]
    for grp in groups:
        banks = set()
        for l in grp:
            p, q = l & 15, l >> 4
            dw = p * 16 + ((q ^ cq[(p >> 2) & 3]) * 4)
            banks.add((dw % 64) // 4)
        assert len(banks) == 16
    for half in (range(0, 32), range(32, 64)):
        for r in range(4):
            cnt = {}
            for l in half:
                p, q = l & 15, l >> 4
                dw = (4 * q) * 16 + (((p >> 2) ^ cq[q]) * 4) + (p & 3) + r * 16
                cnt[dw % 32] = cnt.get(dw % 32, 0) + 1
            assert max(cnt.values()) <= 2
