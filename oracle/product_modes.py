"""ORACLE — test infrastructure only (see oracle/README.md).  Never imported by deep_recommenders_amd/.

numpy model of the ARITHMETIC of the device GEMMs' product modes -- not of anything in the reference: the reference's Dense / Cross
matmuls are plain fp32 `tf.matmul`s (keras/models/ranking/deepfm.py:30-34, dcn.py:81-88), and the parity criterion for them is the
fp64 product within a tolerance.  This file restates how the kernels FORM an fp32 product on the 16-bit matrix pipe, so that the
tolerances the GPU tests use are pinned by a CPU test that runs everywhere (tests/test_product_modes_model.py):

  bf16x3  x = x0 + x1 + x2 exactly (three bf16 terms), a b ~ a0b2 + a1b1 + a2b0 + a0b1 + a1b0 + a0b0      (csrc/bf3_split.h)
  f16x2   x s = h + l (two fp16 terms, 22 bits), s = 2^(140 - e) from the tensor's amax record,             (csrc/bf3_gemm.hip:
          a b ~ (h_a l_b + l_a h_b + h_a h_b) / (s_a s_b)                                                     h2_scale_of / h2_split8)

Accumulation is modelled as fp32 adds of 16-deep blocks of exact products (one MFMA's reduction depth), terms in the kernels' order.
"""
import numpy as np


def amax_bits(x):
    """The amax record of a tensor: max |x| as float bits (uint32); 0 for an empty / all-zero tensor."""
    x = np.asarray(x, dtype=np.float32)
    if x.size == 0:
        return np.uint32(0)
    return np.abs(x).max().astype(np.float32).view(np.uint32)


def h2_scale_of(bits):
    """(s, 1/s) exactly as the kernels derive them from a record (bf3_gemm.hip: h2_scale_of)."""
    e = (int(bits) >> 23) & 0xFF                      # max |x| < 2^(e - 126)
    e = min(max(e, 20), 250)
    s = np.array((267 - e) << 23, dtype=np.uint32).view(np.float32)
    inv = np.array((e - 13) << 23, dtype=np.uint32).view(np.float32)
    return np.float32(s), np.float32(inv)


FP16_MAX = np.float32(65504.0)


def h2_split(x, s):
    """x s = h + l: h = f16_rn(x s) (clamped, MODE.FP16_OVFL), l = f16_rn(x s - h); returned as float32 arrays."""
    v = (np.asarray(x, dtype=np.float32) * np.float32(s)).astype(np.float32)
    with np.errstate(over="ignore"):
        h = np.clip(v, -FP16_MAX, FP16_MAX).astype(np.float16).astype(np.float32)
        r = (v - h).astype(np.float32)                   # exact: the residual of a rounding to fewer bits is representable
        l = np.clip(r, -FP16_MAX, FP16_MAX).astype(np.float16).astype(np.float32)
    return h, l


def bf16_rn(x):
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def bf3_split(x):
    x = np.asarray(x, dtype=np.float32)
    x0 = bf16_rn(x)
    x1 = bf16_rn(x - x0)
    x2 = bf16_rn(x - x0 - x1)
    return x0, x1, x2


def _acc32(terms_a, terms_b):
    out = np.zeros((terms_a[0].shape[0], terms_b[0].shape[1]), dtype=np.float32)
    K = terms_a[0].shape[1]
    for k0 in range(0, K, 16):
        for a, b in zip(terms_a, terms_b):               # one MFMA per term: exact products, one fp32 rounding of the sum
            blk = a[:, k0:k0 + 16].astype(np.float64) @ b[k0:k0 + 16].astype(np.float64)
            out = (out.astype(np.float64) + blk).astype(np.float32)
    return out


def gemm_f16x2(a, b, a_bits=None, b_bits=None):
    """a [M, K] @ b [K, N] in the f16x2 mode; records default to the tensors' exact amax."""
    sa, ia = h2_scale_of(amax_bits(a) if a_bits is None else a_bits)
    sb, ib = h2_scale_of(amax_bits(b) if b_bits is None else b_bits)
    ha, la = h2_split(a, sa)
    hb, lb = h2_split(b, sb)
    acc = _acc32([ha, la, ha], [lb, hb, hb])             # smallest terms first
    return (acc * np.float32(ia * ib)).astype(np.float32)


def gemm_bf16x3(a, b):
    a0, a1, a2 = bf3_split(a)
    b0, b1, b2 = bf3_split(b)
    return _acc32([a0, a1, a2, a0, a1, a0], [b2, b1, b0, b1, b0, b0])


def gemm_f32(a, b):
    return _acc32([np.asarray(a, dtype=np.float32)], [np.asarray(b, dtype=np.float32)])
