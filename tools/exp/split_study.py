"""EXPERIMENT (CPU, numpy): error of three ways to form an fp32 GEMM on the bf16 / fp16 matrix pipe, against fp64 --
native fp32 accumulation, the shipped bf16 three-way split with six products, and an fp16 TWO-way split with power-of-two row / column
scales and three (or four) products.  Accumulation is modelled as fp32 adds of 16-deep blocks of exact products (an MFMA chain).
Output: profiles/r04_split_study.log; discussion: DESIGN.md section 8."""
import numpy as np
rng = np.random.default_rng(0)
M, K, N = 512, 1677, 256

def bf16(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16        # round to nearest even
    return r.astype(np.uint32).view(np.float32)

def acc32(terms_a, terms_b):
    # sum of products accumulated in fp32, k-blocked like an MFMA chain (16 per instruction), products exact (float64 then rounded per block)
    out = np.zeros((terms_a[0].shape[0], terms_b[0].shape[1]), dtype=np.float32)
    Kk = terms_a[0].shape[1]
    for k0 in range(0, Kk, 16):
        blk = np.zeros(out.shape, dtype=np.float64)
        for a, b in zip(terms_a, terms_b):
            blk += a[:, k0:k0 + 16].astype(np.float64) @ b[k0:k0 + 16].astype(np.float64)
        out = (out.astype(np.float64) + blk).astype(np.float32)
    return out

def study(name, A, B):
    ref = A.astype(np.float64) @ B.astype(np.float64)
    scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)      # condition-aware denominator
    def err(C):
        return float(np.max(np.abs(C - ref) / scale)), float(np.sqrt(np.mean(((C - ref) / scale) ** 2)))
    f32 = acc32([A], [B])
    a0 = bf16(A); a1 = bf16(A - a0); a2 = bf16(A - a0 - a1)
    b0 = bf16(B); b1 = bf16(B - b0); b2 = bf16(B - b0 - b1)
    x3 = acc32([a0, a1, a2, a0, a1, a0], [b2, b1, b0, b1, b0, b0])
    # fp16 two-way split with power-of-two per-row (A) / per-column (B) scales
    sa = 2.0 ** np.ceil(np.log2(np.abs(A).max(1, keepdims=True) + 1e-300) - 14)      # row max -> < 2^14
    sb = 2.0 ** np.ceil(np.log2(np.abs(B).max(0, keepdims=True) + 1e-300) - 14)
    As, Bs = (A / sa).astype(np.float32), (B / sb).astype(np.float32)
    h0 = As.astype(np.float16).astype(np.float32); h1 = (As - h0).astype(np.float16).astype(np.float32)
    g0 = Bs.astype(np.float16).astype(np.float32); g1 = (Bs - g0).astype(np.float16).astype(np.float32)
    x2 = acc32([h0, h0, h1], [g0, g1, g0]) * sa.astype(np.float32) * sb.astype(np.float32)
    x2b = acc32([h0, h0, h1, h1], [g0, g1, g0, g1]) * sa.astype(np.float32) * sb.astype(np.float32)
    print("%-34s fp32 %.2e/%.2e  bf16x3(6) %.2e/%.2e  fp16x2(3) %.2e/%.2e  fp16x2(4) %.2e/%.2e   (max / rms of |err| / sum|a||b|)"
          % ((name,) + err(f32) + err(x3) + err(x2) + err(x2b)))

A = rng.normal(0, 0.1, (M, K)).astype(np.float32); B = rng.uniform(-0.05, 0.05, (K, N)).astype(np.float32)
study("forward-like (N(0,.1) x U(.05))", A, B)
A = (rng.normal(0, 1, (M, K)) * np.exp(rng.uniform(-9, 9, (M, 1)))).astype(np.float32)
study("rows spread over e^+-9", A, B)
A = (rng.normal(0, 1, (M, K)) * np.exp(rng.uniform(-9, 9, (M, K)))).astype(np.float32)
study("ELEMENTS spread over e^+-9", A, B)
A = (rng.normal(0, 1e-6, (M, K)) * (rng.random((M, K)) < 0.5)).astype(np.float32)          # relu-sparse tiny gradients
study("gradient-like (1e-6, half zeros)", A, B)
