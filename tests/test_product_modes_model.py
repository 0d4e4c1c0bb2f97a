"""CPU model of the device GEMMs' product modes (oracle/product_modes.py) -- pins, without a GPU, the facts the f16x2 mode rests on:
the scale derived from an amax record keeps every scaled value below 2^14 and is an exact power of two with an exact inverse; the
two-term split carries 22 bits; three products reach the six-product mode's error against fp64; a record that is merely an upper
bound costs bits, not correctness; a record that is too small saturates instead of producing inf / NaN.
(The GPU kernels themselves are checked against fp64 in tests/test_gpu_h2_gemm.py; reference semantics: plain fp32 matmuls,
keras/models/ranking/deepfm.py:30-34, dcn.py:81-88.)"""
import numpy as np
import pytest

from oracle import product_modes as P


def _rel(got, ref):
    return float(np.abs(got.astype(np.float64) - ref).max() / max(np.abs(ref).max(), 1e-300))


def test_scale_of_a_record_is_a_power_of_two_with_headroom():
    rng = np.random.default_rng(0)
    mags = np.concatenate([10.0 ** rng.uniform(-30, 30, 2000), [1.0, 2.0, 0.5, 65504.0, 3e38, 1.2e-38, 1.5, 1.9999999]]).astype(np.float32)
    for m in mags:
        s, inv = P.h2_scale_of(np.float32(m).view(np.uint32))
        assert float(s) * float(inv) == 1.0
        assert np.frexp(s)[0] == 0.5 and np.frexp(inv)[0] == 0.5             # exact powers of two
        e = (int(np.float32(m).view(np.uint32)) >> 23) & 0xFF
        if 20 <= e <= 250:                                                    # the unclamped range: 2^13 <= |m s| < 2^14
            assert 2.0 ** 13 <= float(m) * float(s) < 2.0 ** 14
    # all-zero / denormal tensors and absurd magnitudes: still finite, still inverse of each other
    for bits in (0, 1, 0x007FFFFF, 0x7F7FFFFF, 0x7F800000, 0x7FC00000):
        s, inv = P.h2_scale_of(bits)
        assert np.isfinite(s) and np.isfinite(inv) and float(s) * float(inv) == 1.0


def test_two_fp16_terms_carry_22_bits_of_the_scaled_value():
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(200000) * np.exp(rng.uniform(-6, 0, 200000))).astype(np.float32)
    s, _ = P.h2_scale_of(P.amax_bits(x))
    h, l = P.h2_split(x, s)
    v = x.astype(np.float64) * float(s)
    big = np.abs(v) >= 2.0 ** -3                       # l is a normal fp16 number there: 11 + 11 bits
    assert np.all(np.abs(h + l.astype(np.float64) - v)[big] <= np.abs(v)[big] * 2.0 ** -22)
    # below that the second term is subnormal: the absolute error is at most half its spacing, 2^-25 -- 2^-39 of the largest value
    assert np.all(np.abs(h + l.astype(np.float64) - v) <= 2.0 ** -25 + np.abs(v) * 2.0 ** -22)
    assert np.all(np.abs(h) <= 65504) and np.abs(x * s).max() < 2.0 ** 14


CASES = ["randn", "rows spanning 1e-6..1", "all tiny (1e-20)", "all huge (1e15)", "one element 3e4 x the rest", "relu-sparse 1e-6 gradients"]


@pytest.mark.parametrize("case", CASES)
def test_three_fp16_products_match_six_bf16_products_against_fp64(case):
    rng = np.random.default_rng(len(case))
    M, K, N = 192, 1677, 96
    a = (rng.standard_normal((M, K)) * 0.1).astype(np.float32)
    b = (rng.standard_normal((K, N)) * 0.05).astype(np.float32)
    if case == CASES[1]:
        a = (a * 10.0 ** (-6 * rng.random((M, 1)))).astype(np.float32)
    elif case == CASES[2]:
        a, b = (a * 1e-20).astype(np.float32), (b * 1e-10).astype(np.float32)
    elif case == CASES[3]:
        a, b = (a * 1e15).astype(np.float32), (b * 1e12).astype(np.float32)
    elif case == CASES[4]:
        a[0, 0] = 3000.0
    elif case == CASES[5]:
        a = (rng.standard_normal((M, K)) * 1e-6 * (rng.random((M, K)) < 0.5)).astype(np.float32)
    ref = a.astype(np.float64) @ b.astype(np.float64)
    e2, e3, e1 = _rel(P.gemm_f16x2(a, b), ref), _rel(P.gemm_bf16x3(a, b), ref), _rel(P.gemm_f32(a, b), ref)
    assert e2 <= 4e-6, (e2, e3, e1)                    # the bound tests/test_gpu_h2_gemm.py holds the kernels to
    assert e2 <= 1.5 * e3 + 1e-7, (e2, e3, e1)
    assert np.isfinite(P.gemm_f16x2(a, b)).all()


def test_a_loose_record_costs_bits_and_a_stale_one_saturates():
    rng = np.random.default_rng(5)
    a = (rng.standard_normal((64, 512)) * 0.1).astype(np.float32)
    b = (rng.standard_normal((512, 48)) * 0.05).astype(np.float32)
    ref = a.astype(np.float64) @ b.astype(np.float64)
    exact = _rel(P.gemm_f16x2(a, b), ref)
    loose = _rel(P.gemm_f16x2(a, b, a_bits=np.float32(np.abs(a).max() * 8).view(np.uint32)), ref)      # a running maximum 8 x too large
    assert loose <= 3e-5 and loose >= exact * 0.5
    # a record 3 x too SMALL is still inside the headroom (|x s| < 2^14 * 3 < 65504): no saturation, same accuracy class
    small3 = _rel(P.gemm_f16x2(a, b, a_bits=np.float32(np.abs(a).max() / 3).view(np.uint32)), ref)
    assert small3 <= 4e-6
    # 64 x too small: values saturate at fp16's largest finite number -- wrong, but finite (no inf - inf = NaN in the second term)
    bad = P.gemm_f16x2(a, b, a_bits=np.float32(np.abs(a).max() / 64).view(np.uint32))
    assert np.isfinite(bad).all() and _rel(bad, ref) > 1e-3
