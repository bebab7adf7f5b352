"""The error-free transformations under the matrix-core kernels, bit for bit against a float64 emulation.

The library is compiled with -ffp-contract=fast; a compiler-fused residual has silently broken one of these twice
(the f16 split in round 2, split_cell in round 3).  They are inline asm now, and this test pins
what they compute: split_cell (common.h; the reference's shift / fractional offset, csrc/cuda/spatial_window_operations.cu:
38-61, 85), split_pair and split_product_f16x4 (mfma_split.h), on 10^6 random inputs each plus adversarial ones (rounding
ties of the f16 conversion, cell boundaries, grids that are not a power of two up to M = 2^21).
"""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eft():
    from torch_nfft_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    vp = ctypes.c_void_p
    lib.nfft_dbg_eft.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int, vp, vp, vp, vp, vp, vp, vp]
    lib.nfft_dbg_eft.restype = ctypes.c_int

    def run(kind, M, *arrays):
        n = arrays[0].size
        dev = [torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).cuda() for a in arrays]
        while len(dev) < 4:
            dev.append(dev[0])
        o0 = torch.empty(n, dtype=torch.int32, device="cuda")
        o1 = torch.empty(n, dtype=torch.int32, device="cuda")
        rc = lib.nfft_dbg_eft(kind, n, M, *[vp(t.data_ptr()) for t in dev], vp(o0.data_ptr()), vp(o1.data_ptr()), None)
        assert rc == 0
        torch.cuda.synchronize()
        return o0.cpu().numpy().view(np.uint32), o1.cpu().numpy().view(np.uint32)
    return run


def f16_bits(v64):
    return np.asarray(v64, dtype=np.float64).astype(np.float16).view(np.uint16).astype(np.uint32)


def pack(lo16, hi16):
    return (lo16 | (hi16 << 16)).astype(np.uint32)


def halves(u32):
    u32 = np.asarray(u32, dtype=np.uint32)
    return (u32 & 0xffff).astype(np.uint16).view(np.float16).astype(np.float64), (u32 >> 16).astype(np.uint16).view(np.float16).astype(np.float64)


@pytest.mark.parametrize("M", [512, 1000, 48, 2 ** 21, 2 ** 21 - 2, 1999998])
def test_split_cell_is_exact(eft, M):
    rng = np.random.default_rng(M)
    pos = (rng.random(1_000_000) - 0.5).astype(np.float32)
    k = rng.integers(-M // 2, M // 2, 50_000)
    edge = np.concatenate([k / M, np.nextafter((k / M).astype(np.float32), np.float32(1)), np.nextafter((k / M).astype(np.float32), np.float32(-1)),
                           (k + 0.5) / M, [0.0, -0.0, 0.5, -0.5, 1e-30, -1e-30, 0.49999997, -0.49999997]]).astype(np.float32)
    pos = np.concatenate([pos, edge])
    cell, fr = eft(0, M, pos)
    frac = fr.view(np.float32).astype(np.float64)
    p = pos.astype(np.float64) * M                      # exact in float64 (24-bit x 21-bit)
    fl = np.floor(p)
    want_frac = (p - fl).astype(np.float32).astype(np.float64)   # the exactly rounded offset
    want_cell = np.mod(fl, M).astype(np.int64)
    # the offset rounds to 1 only for a negative product within half an ulp of an integer: then the next cell, offset 0
    carry = want_frac >= 1.0
    want_cell = np.where(carry, np.mod(want_cell + 1, M), want_cell)
    want_frac = np.where(carry, 0.0, want_frac)
    assert frac.min() >= 0.0 and frac.max() < 1.0
    assert cell.min() >= 0 and cell.max() < M
    if M & (M - 1) == 0:
        # power-of-two grids (every benchmark configuration): the product is exact in fp32 and so is the split, bit for bit
        assert np.array_equal(cell.astype(np.int64), want_cell)
        assert np.array_equal(frac, want_frac)
    else:
        # other grids: fp32 product + FMA residual.  cell + frac reproduces pos * M to an ulp of the OFFSET (6e-8 of a cell;
        # the contracted multiply of round 3 was off by half an ulp of pos * M, up to 0.06 cells at M = 2^21) -- where the
        # rounded product lands on the other side of an integer the offset is formed by two roundings instead of one
        dist = np.mod(cell.astype(np.float64) + frac - p + M / 2, M) - M / 2
        assert np.abs(dist).max() <= 2.0 ** -23, np.abs(dist).max()
        exact = (cell.astype(np.int64) == want_cell) & (frac == want_frac)
        assert exact.mean() > 0.999


def test_split_pair_is_exact(eft):
    rng = np.random.default_rng(1)
    v = np.concatenate([rng.standard_normal(1_000_000) * 700.0, rng.random(200_000) * 2048.0]).astype(np.float32)
    # rounding ties of the conversion: an f16 number plus exactly half an ulp (and its float32 neighbours)
    h = (rng.random(100_000) * 2000.0 + 1.0).astype(np.float16)
    ulp = np.spacing(h).astype(np.float64)
    tie = (h.astype(np.float64) + 0.5 * ulp).astype(np.float32)
    v = np.concatenate([v, tie, np.nextafter(tie, np.float32(0)), np.nextafter(tie, np.float32(4096)), -tie])
    if v.size % 2:
        v = v[:-1]
    v0, v1 = v[0::2].copy(), v[1::2].copy()
    hi, lo = eft(1, 0, v0, v1)
    for k, val in enumerate((v0, v1)):
        want_hi = val.astype(np.float16)
        resid = val.astype(np.float64) - want_hi.astype(np.float64)          # exact, and exact in float32 too
        assert np.array_equal(resid, resid.astype(np.float32).astype(np.float64))
        want_lo = resid.astype(np.float16)
        got_hi = ((hi >> (16 * k)) & 0xffff).astype(np.uint16)
        got_lo = ((lo >> (16 * k)) & 0xffff).astype(np.uint16)
        assert np.array_equal(got_hi, want_hi.view(np.uint16))
        assert np.array_equal(got_lo, want_lo.view(np.uint16))
        # hi + lo carries the value to 2^-22 of it (f16 lo parts are normal numbers at these magnitudes)
        big = np.abs(val) > 1.0
        err = np.abs(want_hi.astype(np.float64) + want_lo.astype(np.float64) - val.astype(np.float64))
        assert (err[big] <= np.abs(val.astype(np.float64))[big] * 2.0 ** -21).all()


def test_split_product_f16_is_exact_and_accurate(eft):
    rng = np.random.default_rng(2)
    n = 1_000_000
    m, c = 4, (3 * np.pi / 4) / 4
    # operands as the spreading kernel forms them: 16 psi1 and 2^11 x' psi0, both f16 split
    d1 = rng.random(n) + m - rng.integers(0, 10, n)
    d0 = rng.random(n) + m - rng.integers(0, 10, n)
    p = (np.exp(-c * d1 * d1) * 16.0).astype(np.float32)
    a = (rng.random(n) * np.exp(-c * d0 * d0) * 2048.0 * np.where(rng.random(n) < 0.5, -1.0, 1.0)).astype(np.float32)
    # second element of every packed pair: another draw
    p2, a2 = np.roll(p, 1), np.roll(a, 7)

    def split(v):
        hi = v.astype(np.float16)
        lo = (v.astype(np.float64) - hi.astype(np.float64)).astype(np.float16)
        return hi, lo
    ph, pl = split(p); ah, al = split(a)
    ph2, pl2 = split(p2); ah2, al2 = split(a2)
    u = lambda x, y: pack(x.view(np.uint16).astype(np.uint32), y.view(np.uint16).astype(np.uint32))
    hi, lo = eft(2, 0, u(ph, ph2), u(pl, pl2), u(ah, ah2), u(al, al2))
    assert not (hi == 0xffffffff).any()
    for k, (PH, PL, AH, AL, P, A) in enumerate(((ph, pl, ah, al, p, a), (ph2, pl2, ah2, al2, p2, a2))):
        PH, PL, AH, AL = (t.astype(np.float64) for t in (PH, PL, AH, AL))
        prod = PH * AH                                   # exact in float64
        want_hi = prod.astype(np.float16)
        e = prod - want_hi.astype(np.float64)            # the rounding error of an f16 product is an f16 number ...
        e16 = e.astype(np.float16).astype(np.float64)    # ... unless it underflows (products below ~2^-13: far taps)
        normal = np.abs(prod) >= 2.0 ** -3
        assert np.array_equal(e[normal], e16[normal])
        t = (PH * AL + e16).astype(np.float16)
        want_lo = (PL * AH + t.astype(np.float64)).astype(np.float16)
        got_hi = ((hi >> (16 * k)) & 0xffff).astype(np.uint16)
        got_lo = ((lo >> (16 * k)) & 0xffff).astype(np.uint16)
        assert np.array_equal(got_hi, want_hi.view(np.uint16))
        assert np.array_equal(got_lo, want_lo.view(np.uint16))
        # accuracy of the split product against the fp32 factors' exact product
        exact = P.astype(np.float64) * A.astype(np.float64)
        got = want_hi.astype(np.float64) + want_lo.astype(np.float64)
        rel = np.linalg.norm(got - exact) / np.linalg.norm(exact)
        assert rel < 3e-7, rel
        assert np.abs(got - exact).max() <= np.abs(exact).max() * 2.0 ** -20
