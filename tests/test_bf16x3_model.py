"""CPU model of the bf16x3 product mode of the GEMMs (deep_recommenders_amd/csrc/dense.hip: put4_bf3 + the six-product MFMA
sequence), in numpy: the properties DESIGN.md section 6 relies on, checked without a GPU.  The GPU parity test proper is
tests/test_gpu_kernels.py::test_gemm_bf16x3_matches_fp64_as_well_as_native."""
import numpy as np


def bf16_rn(x):
    """float32 -> nearest bfloat16 (ties to even), returned as float32 (v_cvt_pk_bf16_f32)."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    x0 = bf16_rn(x)
    r1 = (x - x0).astype(np.float32)          # exact in fp32 (checked below)
    x1 = bf16_rn(r1)
    r2 = (r1 - x1).astype(np.float32)
    x2 = bf16_rn(r2)
    return x0, x1, x2, r1, r2


def test_three_way_split_is_exact_and_ordered():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200_000) * np.exp(rng.uniform(-20, 20, 200_000))).astype(np.float32)
    x = np.concatenate([x, np.float32([0.0, 1.0, -1.0, 3.0, 1 + 2.0 ** -23, 2.0 ** -100, -2.0 ** 100, 65504.0])])
    x0, x1, x2, r1, r2 = split3(x)
    xd = x.astype(np.float64)
    # the fp32 subtractions are exact, and the three terms reproduce x to well below one fp32 ulp
    assert np.array_equal(r1.astype(np.float64), xd - x0.astype(np.float64))
    assert np.array_equal(r2.astype(np.float64), r1.astype(np.float64) - x1.astype(np.float64))
    rec = x0.astype(np.float64) + x1.astype(np.float64) + x2.astype(np.float64)
    assert np.all(np.abs(rec - xd) <= 2.0 ** -25 * np.abs(xd))
    # magnitudes: |x1| <= 2^-8 |x|, |x2| <= 2^-16 |x|  (round to nearest: half an ulp of an 8-bit significand)
    ax = np.abs(xd)
    assert np.all(np.abs(x1) <= 2.0 ** -8 * ax * (1 + 2.0 ** -7))
    assert np.all(np.abs(x2) <= 2.0 ** -16 * ax * (1 + 2.0 ** -6))


def test_six_products_match_the_fp32_product_to_2_pow_minus_23():
    rng = np.random.default_rng(1)
    a = (rng.standard_normal(100_000) * np.exp(rng.uniform(-8, 8, 100_000))).astype(np.float32)
    b = (rng.standard_normal(100_000) * np.exp(rng.uniform(-8, 8, 100_000))).astype(np.float32)
    a0, a1, a2, _, _ = split3(a)
    b0, b1, b2, _, _ = split3(b)
    d = lambda v: v.astype(np.float64)
    six = d(a0) * d(b2) + d(a1) * d(b1) + d(a2) * d(b0) + d(a0) * d(b1) + d(a1) * d(b0) + d(a0) * d(b0)
    exact = d(a) * d(b)
    # dropped: a1 b2 + a2 b1 + a2 b2 <= (2 * 2^-24 + 2^-32) |ab| plus the split's own residue
    assert np.max(np.abs(six - exact) / np.abs(exact)) <= 2.0 ** -22.5
    # every bf16 x bf16 product has at most 16 significant bits: exact in an fp32 accumulator
    for u, v in ((a0, b0), (a0, b1), (a1, b0), (a0, b2), (a1, b1), (a2, b0)):
        p = d(u) * d(v)
        assert np.array_equal(p, p.astype(np.float32).astype(np.float64))


def test_dot_products_as_accurate_as_fp32_accumulation():
    """K = 1677 dot products (the first tower layer): six-product emulation with fp32 accumulation against plain fp32
    multiply-accumulate, both measured against float64."""
    rng = np.random.default_rng(2)
    K, N = 1677, 512
    a = rng.standard_normal((N, K)).astype(np.float32)
    b = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    ref = np.sum(a.astype(np.float64) * b.astype(np.float64), axis=1)
    a0, a1, a2, _, _ = split3(a)
    b0, b1, b2, _, _ = split3(b)
    acc = np.zeros(N, dtype=np.float32)
    nat = np.zeros(N, dtype=np.float32)
    d = lambda v: v.astype(np.float64)
    # v_mfma_f32_32x32x16_bf16 adds the 16 (exact) products of a k-block to the fp32 accumulator: one rounding per MFMA
    for k0 in range(0, K, 16):
        sl = slice(k0, min(k0 + 16, K))
        for u, v in ((a0, b2), (a1, b1), (a2, b0), (a0, b1), (a1, b0), (a0, b0)):
            acc = (d(acc) + np.sum(d(u[:, sl]) * d(v[:, sl]), axis=1)).astype(np.float32)
    for k in range(K):                         # plain fp32 fused multiply-add chain
        nat = (d(nat) + d(a[:, k]) * d(b[:, k])).astype(np.float32)
    scale = np.abs(ref).max()
    e_emul = np.abs(d(acc) - ref).max() / scale
    e_nat = np.abs(d(nat) - ref).max() / scale
    assert e_emul <= 2e-6 and e_emul <= 2.0 * e_nat + 1e-7, (e_emul, e_nat)
