"""Thin, shape-checked Python wrappers over the C-ABI kernels (one function per entry point), plus the
torch.autograd.Function glue that lets the reference-shaped model classes train with loss.backward().

Every function launches hand-written HIP kernels on torch's current stream; nothing here computes in
PyTorch.  Reference citations (relative to the reference root) are on the C-ABI declarations in
include/dr_hotpath.h.
"""
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr


def _c(t: torch.Tensor, dtype) -> torch.Tensor:
    if t.dtype != dtype:
        raise TypeError("expected %s, got %s" % (dtype, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


# ----------------------------------------------------------------------------------------------
# K1 / K2   integer id path
# ----------------------------------------------------------------------------------------------
def hash_bucket_i64(keys: torch.Tensor, col_buckets: torch.Tensor, out: Optional[torch.Tensor] = None):
    """keys [B, C] int64 -> ids [B, C]; col_buckets [C] uint64-as-int64 (0 = pass-through column)."""
    keys = _c(keys, torch.int64)
    B, C = keys.shape
    col_buckets = _c(col_buckets, torch.int64)
    assert col_buckets.numel() == C
    if out is None:
        out = torch.empty_like(keys)
    check(lib().dr_hash_bucket_i64(ptr(keys), B, C, ptr(col_buckets), ptr(out), stream_ptr()), "dr_hash_bucket_i64")
    return out


def _csr_bytes(values: Sequence, device):
    import numpy as np
    flat = [v if isinstance(v, (bytes, bytearray)) else str(v).encode("utf-8") for v in values]
    offs = np.zeros(len(flat) + 1, dtype=np.int64)
    if flat:
        offs[1:] = np.cumsum([len(b) for b in flat])
    blob = np.frombuffer(b"".join(flat) + b"\0", dtype=np.uint8).copy()
    return torch.from_numpy(blob).to(device), torch.from_numpy(offs).to(device), len(flat)


def hash_bucket_strings(values: Sequence, num_buckets: int, device="cuda"):
    """list of str/bytes -> ids [n]; "" -> -1.  (host strings are packed to CSR bytes, hashed on device)"""
    blob, offs, n = _csr_bytes(values, device)
    out = torch.empty(n, dtype=torch.int64, device=device)
    check(lib().dr_hash_bucket_bytes(ptr(blob), ptr(offs), n, int(num_buckets), ptr(out), stream_ptr()),
          "dr_hash_bucket_bytes")
    return out


def vocab_lookup_i64(keys: torch.Tensor, vocab: torch.Tensor):
    keys = _c(keys, torch.int64)
    vocab = _c(vocab, torch.int64)
    out = torch.empty_like(keys)
    check(lib().dr_vocab_lookup_i64(ptr(keys), keys.numel(), ptr(vocab), vocab.numel(), ptr(out), stream_ptr()),
          "dr_vocab_lookup_i64")
    return out


def vocab_lookup_strings(values: Sequence, vocab: Sequence, device="cuda"):
    blob, offs, n = _csr_bytes(values, device)
    vblob, voffs, m = _csr_bytes(vocab, device)
    out = torch.empty(n, dtype=torch.int64, device=device)
    check(lib().dr_vocab_lookup_bytes(ptr(blob), ptr(offs), n, ptr(vblob), ptr(voffs), m, ptr(out), stream_ptr()),
          "dr_vocab_lookup_bytes")
    return out


# ----------------------------------------------------------------------------------------------
# K3 / K4   fused embedding gather + pool (+ first-order + FM)
# ----------------------------------------------------------------------------------------------
def emb_pool_fwd(ids, F, col_start, row_base, table, lin_w=None, lin_bias=None, ld_concat=None,
                 want_sum_x=True, want_fm=True, concat=None, sum_x=None, fm_logit=None, second_order=True):
    ids = _c(ids, torch.int64)
    B, C = ids.shape
    D = table.shape[1]
    ld = F * D if ld_concat is None else ld_concat
    dev = ids.device
    if concat is None:
        concat = torch.empty((B, ld), dtype=torch.float32, device=dev)
        if ld > F * D:
            concat[:, F * D:].zero_()
    if want_sum_x and sum_x is None:
        sum_x = torch.empty((B, D), dtype=torch.float32, device=dev)
    if want_fm and fm_logit is None:
        fm_logit = torch.empty((B,), dtype=torch.float32, device=dev)
    check(lib().dr_emb_pool_fwd_ex(ptr(ids), B, F, C, ptr(col_start), ptr(row_base), ptr(table), D, ptr(lin_w),
                                   ptr(lin_bias), ptr(concat), ld, ptr(sum_x) if want_sum_x else None,
                                   ptr(fm_logit) if want_fm else None, 0 if second_order else 1, stream_ptr()),
          "dr_emb_pool_fwd_ex")
    return concat, sum_x, fm_logit


def lin_fields_fwd(ids, F, col_start, row_base, lin_w):
    """out[b, f] = sum over field f's bag of lin_w[row_base[f] + id] (FNN's per-field first-order inputs)."""
    ids = _c(ids, torch.int64)
    B, C = ids.shape
    out = torch.empty((B, _pad4(F)), dtype=torch.float32, device=ids.device)[:, :F]
    check(lib().dr_lin_fields_fwd(ptr(ids), B, F, C, ptr(col_start), ptr(row_base), ptr(lin_w), ptr(out), out.stride(0),
                                  stream_ptr()), "dr_lin_fields_fwd")
    return out


def lin_fields_bwd(ids, F, col_start, row_base, d_out, scale, dst_lin):
    ids = _c(ids, torch.int64)
    B, C = ids.shape
    assert d_out.stride(1) == 1
    check(lib().dr_lin_fields_bwd(ptr(ids), B, F, C, ptr(col_start), ptr(row_base), ptr(d_out), d_out.stride(0), float(scale),
                                  ptr(dst_lin), stream_ptr()), "dr_lin_fields_bwd")


def emb_pool_bwd(ids, F, col_start, row_base, D, d_concat, concat, sum_x, d_fm_logit, scale, dst_table, dst_lin,
                 dst_bias=None):
    ids = _c(ids, torch.int64)
    B, C = ids.shape
    check(lib().dr_emb_pool_bwd(ptr(ids), B, F, C, ptr(col_start), ptr(row_base), D,
                                ptr(d_concat), d_concat.stride(0) if d_concat is not None else 0,
                                ptr(concat), concat.stride(0) if concat is not None else 0,
                                ptr(sum_x), ptr(d_fm_logit), float(scale), ptr(dst_table), ptr(dst_lin),
                                ptr(dst_bias), stream_ptr()), "dr_emb_pool_bwd")


# ----------------------------------------------------------------------------------------------
# K6   stand-alone FM second order
# ----------------------------------------------------------------------------------------------
def fm2_fwd(x):
    x = _c(x, torch.float32)
    B, F, D = x.shape
    out = torch.empty((B, 1), dtype=torch.float32, device=x.device)
    check(lib().dr_fm2_fwd(ptr(x), B, F, D, ptr(out), stream_ptr()), "dr_fm2_fwd")
    return out


def fm2_bwd(x, d_out):
    x = _c(x, torch.float32)
    d_out = _c(d_out, torch.float32)
    B, F, D = x.shape
    dx = torch.empty_like(x)
    check(lib().dr_fm2_bwd(ptr(x), ptr(d_out), B, F, D, ptr(dx), stream_ptr()), "dr_fm2_bwd")
    return dx


# ----------------------------------------------------------------------------------------------
# K7 / K8   dense layers on fp32 MFMA
# ----------------------------------------------------------------------------------------------
def _pad4(n):
    return (n + 3) // 4 * 4


def _rowmajor_ld4(t):
    """Returns a [M, K] fp32 view whose row stride is a multiple of 4 floats and base 16-B aligned."""
    assert t.dim() == 2 and t.dtype == torch.float32
    if t.stride(1) == 1 and t.stride(0) >= t.shape[1]:
        return t      # the kernel takes float4 loads only when base and pitch are 16-B aligned
    if t.shape[1] == 1 and t.stride(0) >= 1:
        return t
    M, K = t.shape
    buf = torch.zeros((M, _pad4(K)), dtype=torch.float32, device=t.device)
    buf[:, :K].copy_(t)
    return buf[:, :K]


def linear_fwd(x, W, b=None, act=0, out=None):
    """y = act(x @ W + b); x [M,K] (row stride % 4 == 0), W [K,N] row-major."""
    x = _rowmajor_ld4(x)
    W = _rowmajor_ld4(W)
    M, K = x.shape
    N = W.shape[1]
    if out is None:
        out = torch.empty((M, _pad4(N)), dtype=torch.float32, device=x.device)[:, :N]
    check(lib().dr_linear_fwd(ptr(x), x.stride(0), ptr(W), W.stride(0), ptr(b), M, K, N, int(act), ptr(out),
                              out.stride(0), stream_ptr()), "dr_linear_fwd")
    return out


def linear_bwd_dx(dy, W, relu_src=None, accumulate=False, out=None):
    dy = _rowmajor_ld4(dy)
    W = _rowmajor_ld4(W)
    M, N = dy.shape
    K = W.shape[0]
    if out is None:
        assert not accumulate
        out = torch.empty((M, _pad4(K)), dtype=torch.float32, device=dy.device)[:, :K]
    check(lib().dr_linear_bwd_dx(ptr(dy), dy.stride(0), ptr(W), W.stride(0), M, K, N, ptr(relu_src),
                                 relu_src.stride(0) if relu_src is not None else 0, int(bool(accumulate)), ptr(out),
                                 out.stride(0), stream_ptr()), "dr_linear_bwd_dx")
    return out


def linear_fwd_splitk_workspace(M, K, N, device):
    return torch.empty(max(1, lib().dr_linear_fwd_splitk_workspace_bytes(int(M), int(K), int(N)) // 4), dtype=torch.float32,
                       device=device)


def linear_fwd_splitk(x, W, out, workspace=None):
    """out += x @ W with the reduction split over the grid (few output tiles, long K); slices summed in a fixed order."""
    x = _rowmajor_ld4(x)
    W = _rowmajor_ld4(W)
    M, K = x.shape
    N = W.shape[1]
    assert W.shape[0] == K and out.shape == (M, N) and out.stride(1) == 1
    if workspace is None or workspace.numel() * 4 < lib().dr_linear_fwd_splitk_workspace_bytes(M, K, N):
        workspace = linear_fwd_splitk_workspace(M, K, N, x.device)
    check(lib().dr_linear_fwd_splitk(ptr(x), x.stride(0), ptr(W), W.stride(0), M, K, N, ptr(out), out.stride(0), ptr(workspace),
                                     workspace.numel() * 4, stream_ptr()), "dr_linear_fwd_splitk")
    return out


def linear_bwd_dw_workspace(M, K, N, device):
    return torch.empty(max(1, lib().dr_linear_bwd_dw_workspace_bytes(int(M), int(K), int(N)) // 4), dtype=torch.float32,
                       device=device)


def linear_bwd_dw(x, dy, scale, dstW, dstb=None, workspace=None):
    """dstW += scale * x^T @ dy ; dstb += scale * colsum(dy).  With a workspace the split-K partials are combined
    deterministically by a reduce kernel; without, by fp32 atomics."""
    x = _rowmajor_ld4(x)
    dy = _rowmajor_ld4(dy)
    M, K = x.shape
    N = dy.shape[1]
    assert dstW.shape == (K, N) and dstW.stride(1) == 1
    check(lib().dr_linear_bwd_dw(ptr(x), x.stride(0), ptr(dy), dy.stride(0), M, K, N, float(scale), ptr(dstW),
                                 dstW.stride(0), ptr(dstb), ptr(workspace),
                                 workspace.numel() * 4 if workspace is not None else 0, stream_ptr()), "dr_linear_bwd_dw")


def linear_bwd_narrow_supported(M, K, N):
    """Shape domain of the fused narrow-layer backward (dr_linear_bwd_narrow)."""
    return N <= 32 and K in (128, 256, 512) and M > 0 and M % 32 == 0


def linear_bwd_narrow_workspace(M, K, N, device):
    return torch.empty(max(64, lib().dr_linear_bwd_narrow_workspace_bytes(int(M), int(K), int(N)) // 4),
                       dtype=torch.float32, device=device)


def linear_bwd_narrow(x, dy, W, scale, dstW, dstb, dx, relu_mask=True, workspace=None, parts=3, dx_amax=None):
    """One pass over x: dx = (dy @ W^T) * (x > 0 if relu_mask); dstW += scale * x^T dy; dstb += scale * colsum(dy).
    dx uses the pre-update W even when dstW is W.  Raises RuntimeError(DR_ESHAPE) outside the kernel's domain."""
    M, K = x.shape
    N = dy.shape[1]
    assert x.stride(1) == 1 and dy.stride(1) == 1 and W.stride(1) == 1 and dstW.stride(1) == 1 and dx.stride(1) == 1
    assert W.shape == (K, N) and dstW.shape == (K, N) and dx.shape == (M, K)
    if workspace is None:
        workspace = linear_bwd_narrow_workspace(M, K, N, x.device)
    if dx_amax is not None:  # also leave max |dx| in the record (the f16x2 GEMMs that take dx as an operand want it)
        check(lib().dr_linear_bwd_narrow_amax(ptr(x), x.stride(0), ptr(dy), dy.stride(0), ptr(W), W.stride(0), M, K, N,
                                              1 if relu_mask else 0, float(scale), ptr(dstW), dstW.stride(0), ptr(dstb), ptr(dx),
                                              dx.stride(0), ptr(workspace), workspace.numel() * 4, int(parts), ptr(dx_amax), stream_ptr()),
              "dr_linear_bwd_narrow_amax")
        return dx
    if parts != 3:          # 1: the one-pass kernel, 2: the reduce that applies the partials (may run on another stream)
        check(lib().dr_linear_bwd_narrow_parts(ptr(x), x.stride(0), ptr(dy), dy.stride(0), ptr(W), W.stride(0), M, K, N,
                                               1 if relu_mask else 0, float(scale), ptr(dstW), dstW.stride(0), ptr(dstb), ptr(dx),
                                               dx.stride(0), ptr(workspace), workspace.numel() * 4, int(parts), stream_ptr()),
              "dr_linear_bwd_narrow_parts")
        return dx
    check(lib().dr_linear_bwd_narrow(ptr(x), x.stride(0), ptr(dy), dy.stride(0), ptr(W), W.stride(0), M, K, N,
                                     1 if relu_mask else 0, float(scale), ptr(dstW), dstW.stride(0), ptr(dstb), ptr(dx),
                                     dx.stride(0), ptr(workspace), workspace.numel() * 4, stream_ptr()),
          "dr_linear_bwd_narrow")
    return dx


def tower_tail_supported(M, K, H):
    """Shape domain of the one-pass tower tail (dr_tower_tail_fused)."""
    return H <= 32 and K in (128, 256) and M > 0 and M % 32 == 0


def tower_tail_workspace(M, K, device):
    return torch.empty(max(128, lib().dr_tower_tail_workspace_bytes(int(M), int(K)) // 4), dtype=torch.float32, device=device)


def tower_tail_fused(x, W1, b1, W2, b2, extra_logit, labels, loss_mode, scale, dx, dst_W1="inplace", dst_b1="inplace", dst_W2="inplace",
                     dst_b2="inplace", prob=None, d_logit=None, d_h=None, loss=None, workspace=None, n_total=0, parts=3, dx_amax=None):
    """tower_head_fwd_bwd (relu) + linear_bwd_narrow (relu mask) of the same layer W1 in ONE pass over x (dr_tower_tail_fused).
    dst_* += scale * gradient: the parameters themselves by default (fused SGD, scale = -lr) or gradient buffers (scale = 1).
    Returns (loss, prob, d_logit, dx)."""
    M, K = x.shape
    H = W1.shape[1]
    dev = x.device
    dst_W1 = W1 if isinstance(dst_W1, str) else dst_W1
    dst_b1 = b1 if isinstance(dst_b1, str) else dst_b1
    dst_W2 = W2 if isinstance(dst_W2, str) else dst_W2
    dst_b2 = b2 if isinstance(dst_b2, str) else dst_b2
    assert x.stride(1) == 1 and W1.stride(1) == 1 and dst_W1.stride(1) == 1 and dx.stride(1) == 1 and W2.shape == (H, 1) and dx.shape == (M, K)
    prob = prob if prob is not None else torch.empty(M, dtype=torch.float32, device=dev)
    d_logit = d_logit if d_logit is not None else torch.empty(M, dtype=torch.float32, device=dev)
    loss = loss if loss is not None else torch.empty(1, dtype=torch.float32, device=dev)
    if workspace is None:
        workspace = tower_tail_workspace(M, K, dev)
    check(lib().dr_tower_tail_fused(ptr(x), x.stride(0), ptr(W1), W1.stride(0), ptr(b1), M, int(n_total), K, H, ptr(W2), W2.stride(0), ptr(b2),
                                    ptr(extra_logit), ptr(labels), int(loss_mode), float(scale), ptr(dst_W1), dst_W1.stride(0), ptr(dst_b1),
                                    ptr(dst_W2), dst_W2.stride(0) if dst_W2 is not None else 0, ptr(dst_b2), ptr(prob), ptr(d_logit), ptr(d_h),
                                    d_h.stride(0) if d_h is not None else 0, ptr(dx), dx.stride(0), ptr(loss), ptr(workspace),
                                    workspace.numel() * 4, int(parts), ptr(dx_amax), stream_ptr()), "dr_tower_tail_fused")
    return loss, prob, d_logit, dx


def tower_head_workspace(M, device):
    return torch.empty(max(64, lib().dr_tower_head_workspace_bytes(int(M)) // 4), dtype=torch.float32, device=device)


def tower_head_fwd_bwd(x, W1, b1, W2, b2, extra_logit, labels, loss_mode, scale, act=1, h_out=None, prob=None, d_logit=None,
                       d_h=None, loss=None, workspace=None, dst_W2="inplace", dst_b2="inplace", n_total=0, parts=3):
    """Last hidden layer (H <= 32) + Dense(1) + extra logit + BCE loss + the Dense(1) backward, fused.
    W2: [H, 1] (any row stride).  dst_W2 / dst_b2 += scale * gradient: by default the parameters themselves (fused SGD,
    scale = -lr); pass gradient buffers with scale = 1 for data-parallel training, or None to skip.  n_total: size of
    the batch the loss is a mean over when x is one slice of it (0 = M).
    Returns (loss, prob, d_logit, d_h)."""
    if isinstance(dst_W2, str):
        dst_W2 = W2
    if isinstance(dst_b2, str):
        dst_b2 = b2
    M, K = x.shape
    H = W1.shape[1]
    dev = x.device
    assert x.stride(1) == 1 and W1.stride(1) == 1 and W2.shape == (H, 1)
    prob = prob if prob is not None else torch.empty(M, dtype=torch.float32, device=dev)
    d_logit = d_logit if d_logit is not None else torch.empty(M, dtype=torch.float32, device=dev)
    d_h = d_h if d_h is not None else torch.empty((M, _pad4(H)), dtype=torch.float32, device=dev)[:, :H]
    loss = loss if loss is not None else torch.empty(1, dtype=torch.float32, device=dev)
    if workspace is None:
        workspace = tower_head_workspace(M, dev)
    if parts != 3:          # 1: GEMM + head kernel, 2: the finish kernel (partials -> dst_W2 / dst_b2 / loss; may run on another stream)
        check(lib().dr_tower_head_fwd_bwd_parts(ptr(x), x.stride(0), ptr(W1), W1.stride(0), ptr(b1), M, int(n_total), K, H, int(act), ptr(W2),
                                                W2.stride(0), ptr(b2), ptr(extra_logit), ptr(labels), int(loss_mode), float(scale),
                                                ptr(dst_W2), dst_W2.stride(0) if dst_W2 is not None else 0, ptr(dst_b2),
                                                ptr(h_out), h_out.stride(0) if h_out is not None else 0, ptr(prob), ptr(d_logit),
                                                ptr(d_h), d_h.stride(0), ptr(loss), ptr(workspace), workspace.numel() * 4, int(parts),
                                                stream_ptr()), "dr_tower_head_fwd_bwd_parts")
        return loss, prob, d_logit, d_h
    check(lib().dr_tower_head_fwd_bwd(ptr(x), x.stride(0), ptr(W1), W1.stride(0), ptr(b1), M, int(n_total), K, H, int(act), ptr(W2),
                                      W2.stride(0), ptr(b2), ptr(extra_logit), ptr(labels), int(loss_mode), float(scale),
                                      ptr(dst_W2), dst_W2.stride(0) if dst_W2 is not None else 0, ptr(dst_b2),
                                      ptr(h_out), h_out.stride(0) if h_out is not None else 0, ptr(prob), ptr(d_logit),
                                      ptr(d_h), d_h.stride(0), ptr(loss), ptr(workspace), workspace.numel() * 4,
                                      stream_ptr()), "dr_tower_head_fwd_bwd")
    return loss, prob, d_logit, d_h


def cross_fwd(x0, x, W, b, diag_scale=0.0, want_prod=False, prod=None):
    """out = x0 * (x @ W + b + diag*x) + x.  W None: `prod` holds x @ W (low-rank path) and is finished in place."""
    x0 = _rowmajor_ld4(x0)
    x = _rowmajor_ld4(x)
    if x0.stride(0) != x.stride(0):
        x = _rowmajor_ld4(x.contiguous())
        x0 = _rowmajor_ld4(x0.contiguous())
    M, Dm = x.shape
    ld = x.stride(0)
    out = torch.empty((M, ld), dtype=torch.float32, device=x.device)[:, :Dm]
    if W is not None:
        W = _rowmajor_ld4(W)
        if want_prod:
            prod = torch.empty((M, ld), dtype=torch.float32, device=x.device)[:, :Dm]
        ldw = W.stride(0)
    else:
        assert prod is not None and prod.stride(0) == ld
        ldw = 0
    check(lib().dr_cross_fwd(ptr(x0), ptr(x), ld, ptr(W), ldw, ptr(b), float(diag_scale), M, Dm, ptr(out),
                             ptr(prod) if (want_prod or W is None) else None, stream_ptr()), "dr_cross_fwd")
    return out, prod


def cross_combine_bwd(x0, prod, d_out, diag_scale, d_x0_accum, d_x_accum, d_prod_amax=None):
    M, Dm = d_out.shape
    ld = d_out.stride(0)
    for t in (x0, prod, d_x0_accum, d_x_accum):
        assert t is None or t.stride(0) == ld, "cross tensors must share one leading dimension"
    d_prod = torch.empty((M, ld), dtype=torch.float32, device=d_out.device)[:, :Dm]
    if d_prod_amax is not None:     # also the amax record of d_prod (operand of the cross layer's f16x2 dgrad / wgrad)
        check(lib().dr_cross_combine_bwd_amax(ptr(x0), ptr(prod), ptr(d_out), M, Dm, ld, float(diag_scale), ptr(d_prod),
                                              ptr(d_x0_accum), ptr(d_x_accum), ptr(d_prod_amax), stream_ptr()), "dr_cross_combine_bwd_amax")
        return d_prod
    check(lib().dr_cross_combine_bwd(ptr(x0), ptr(prod), ptr(d_out), M, Dm, ld, float(diag_scale), ptr(d_prod),
                                     ptr(d_x0_accum), ptr(d_x_accum), stream_ptr()), "dr_cross_combine_bwd")
    return d_prod


# ----------------------------------------------------------------------------------------------
# K11   fused sigmoid + BCE
# ----------------------------------------------------------------------------------------------
LOSS_SIGMOID_CE, LOSS_LOG_LOSS, LOSS_KERAS_BCE = 0, 1, 2


def bce_fwd_bwd(logits, labels, mode=LOSS_SIGMOID_CE, want_prob=True, want_grad=True, workspace=None, logits_b=None,
                out=None):
    logits = _c(logits.reshape(-1), torch.float32)
    labels = _c(labels.reshape(-1), torch.float32)
    n = logits.numel()
    dev = logits.device
    if out is not None:
        prob, d_logit, loss = out
    else:
        prob = torch.empty(n, dtype=torch.float32, device=dev) if want_prob else None
        d_logit = torch.empty(n, dtype=torch.float32, device=dev) if want_grad else None
        loss = torch.empty(1, dtype=torch.float32, device=dev)
    if workspace is None:
        workspace = torch.empty(1024, dtype=torch.float32, device=dev)
    ldb = 0
    if logits_b is not None:
        assert logits_b.dim() == 2 and logits_b.shape[0] == n and logits_b.dtype == torch.float32
        ldb = logits_b.stride(0)
    check(lib().dr_bce_fwd_bwd(ptr(logits), ptr(logits_b), ldb, ptr(labels), n, int(mode), ptr(prob), ptr(d_logit),
                               ptr(loss), ptr(workspace), stream_ptr()), "dr_bce_fwd_bwd")
    return loss, prob, d_logit


def sigmoid_fwd(x):
    x = _c(x, torch.float32)
    y = torch.empty_like(x)
    check(lib().dr_sigmoid_fwd(ptr(x), x.numel(), ptr(y), stream_ptr()), "dr_sigmoid_fwd")
    return y


def sigmoid_bwd(y, dy):
    y = _c(y, torch.float32)
    dy = _c(dy, torch.float32)
    dx = torch.empty_like(y)
    check(lib().dr_sigmoid_bwd(ptr(y), ptr(dy), y.numel(), ptr(dx), stream_ptr()), "dr_sigmoid_bwd")
    return dx


def bce_prob_fwd_bwd(prob, labels, mode, want_grad=True, workspace=None):
    prob = _c(prob.reshape(-1), torch.float32)
    labels = _c(labels.reshape(-1), torch.float32)
    n = prob.numel()
    d_prob = torch.empty_like(prob) if want_grad else None
    loss = torch.empty(1, dtype=torch.float32, device=prob.device)
    if workspace is None:
        workspace = torch.empty(1024, dtype=torch.float32, device=prob.device)
    check(lib().dr_bce_prob_fwd_bwd(ptr(prob), ptr(labels), n, int(mode), ptr(d_prob), ptr(loss), ptr(workspace),
                                    stream_ptr()), "dr_bce_prob_fwd_bwd")
    return loss, d_prob


# ----------------------------------------------------------------------------------------------
# row-sharded tables: bucketing + owner-side gather / scatter, dense axpy
# ----------------------------------------------------------------------------------------------
def shard_dedup_slots(ids, row_base, num_rows, plan, out=None):
    """(rep, unique_flags) of a micro-batch: rep[p] int64 = the lowest slot that looks up the same row as slot p (dr_shard_dedup_slots
    over `plan` = emb_sort_slots(ids, row_base, num_rows)); unique_flags = the plan's per-slot flags (row shared by no other slot)."""
    B, F = ids.shape
    n = B * F
    if out is None or out.numel() < n:
        out = torch.empty(n, dtype=torch.int64, device=ids.device)
    check(lib().dr_shard_dedup_slots(ptr(plan.rows), ptr(plan.slots), ptr(plan.dup_count), n, int(num_rows), ptr(out), stream_ptr()),
          "dr_shard_dedup_slots")
    return out, plan.flags


def shard_bucket_ids(ids, rows_per_shard, world, counts=None, send_rows=None, pos=None, workspace=None, rep=None):
    ids = _c(ids, torch.int64)
    B, C = ids.shape
    n = B * C
    dev = ids.device
    if counts is None:
        counts = torch.empty(world, dtype=torch.int64, device=dev)
    if send_rows is None:
        send_rows = torch.empty(n, dtype=torch.int64, device=dev)
    if pos is None:
        pos = torch.empty((B, C), dtype=torch.int64, device=dev)
    if workspace is None:
        nbytes = lib().dr_shard_bucket_workspace_bytes(n, world)
        workspace = torch.empty(nbytes // 8, dtype=torch.int64, device=dev)
    if rep is not None:
        check(lib().dr_shard_bucket_ids_dedup(ptr(ids), ptr(rep), n, C, int(rows_per_shard), int(world), ptr(counts), ptr(send_rows),
                                              ptr(pos), ptr(workspace), stream_ptr()), "dr_shard_bucket_ids_dedup")
        return counts, send_rows, pos
    check(lib().dr_shard_bucket_ids(ptr(ids), n, C, int(rows_per_shard), int(world), ptr(counts), ptr(send_rows),
                                    ptr(pos), ptr(workspace), stream_ptr()), "dr_shard_bucket_ids")
    return counts, send_rows, pos


def rows_gather(rows, table, lin_w=None, out_rows=None, out_lin=None):
    rows = _c(rows, torch.int64)
    n = rows.numel()
    D = table.shape[1]
    if out_rows is None:
        out_rows = torch.empty((n, D), dtype=torch.float32, device=rows.device)
    if out_lin is None and lin_w is not None:
        out_lin = torch.empty(n, dtype=torch.float32, device=rows.device)
    check(lib().dr_rows_gather(ptr(rows), n, ptr(table), D, ptr(lin_w), ptr(out_rows), ptr(out_lin), stream_ptr()),
          "dr_rows_gather")
    return out_rows, out_lin


def rows_scatter_add(rows, grads, lin_grads, scale, table, lin_w):
    rows = _c(rows, torch.int64)
    n = rows.numel()
    D = table.shape[1]
    check(lib().dr_rows_scatter_add(ptr(rows), n, ptr(grads), D, ptr(lin_grads), float(scale), ptr(table), ptr(lin_w),
                                    stream_ptr()), "dr_rows_scatter_add")


def axpy(alpha, x, y):
    assert x.is_contiguous() and y.is_contiguous() and x.numel() == y.numel()
    check(lib().dr_axpy(x.numel(), float(alpha), ptr(x), ptr(y), stream_ptr()), "dr_axpy")


# ----------------------------------------------------------------------------------------------
# K4 deterministic form: sort slots by row, segment-sum, plain read-modify-write
# ----------------------------------------------------------------------------------------------
def emb_sort_workspace(n, device):
    return torch.empty(lib().dr_emb_sort_workspace_bytes(int(n)), dtype=torch.uint8, device=device)


class SortPlan:
    """Per-batch outputs of dr_emb_sort_slots (all preallocated, reusable across steps)."""

    def __init__(self, n, device):
        self.n = int(n)                      # capacity in slots: a plan serves any batch of <= n slots
        self.rows = torch.empty(n, dtype=torch.int64, device=device)
        self.slots = torch.empty(n, dtype=torch.int32, device=device)
        self.flags = torch.empty(n, dtype=torch.uint8, device=device)
        self.dup_heads = torch.empty(n, dtype=torch.int32, device=device)
        self.dup_count = torch.zeros(4, dtype=torch.int32, device=device)     # [work-list entries, length of the sorted arrays]
        self.workspace = emb_sort_workspace(n, device)

    def sorted_len(self):
        """Length of the sorted (rows, slots) arrays of the last plan built here: only the slots of shared rows (claim path) or
        all slots (radix path).  Host read -- tests / debugging."""
        return int(self.dup_count[1].item())


def emb_sort_slots(ids, row_base, num_rows, plan=None):
    ids = _c(ids, torch.int64)
    B, F = ids.shape
    if plan is None or plan.n < B * F:       # a reused plan sized for a smaller batch would be rejected (DR_EINVAL)
        plan = SortPlan(B * F, ids.device)
    check(lib().dr_emb_sort_slots(ptr(ids), B, F, ptr(row_base), int(num_rows), ptr(plan.rows), ptr(plan.slots),
                                  ptr(plan.flags), ptr(plan.dup_heads), ptr(plan.dup_count), ptr(plan.workspace),
                                  plan.workspace.numel(), stream_ptr()), "dr_emb_sort_slots")
    return plan


def hash_sort_slots(keys, col_buckets, row_base, num_rows, ids_out, ids_t_out=None, plan=None):
    """hash_bucket_i64(keys -> ids_out) + ids_transpose_i32(ids_out -> ids_t_out, optional) + emb_sort_slots(ids_out) as one chain whose
    first kernel does the work of the first four (dr_hash_sort_slots): same outputs, bit for bit."""
    keys = _c(keys, torch.int64)
    B, F = keys.shape
    assert ids_out.shape == (B, F) and ids_out.dtype == torch.int64 and ids_out.is_contiguous()
    if ids_t_out is not None:
        assert ids_t_out.shape == (F, B) and ids_t_out.dtype == torch.int32 and ids_t_out.is_contiguous()
    if plan is None or plan.n < B * F:
        plan = SortPlan(B * F, keys.device)
    check(lib().dr_hash_sort_slots(ptr(keys), B, F, ptr(col_buckets), ptr(ids_out), ptr(ids_t_out), ptr(row_base), int(num_rows),
                                   ptr(plan.rows), ptr(plan.slots), ptr(plan.flags), ptr(plan.dup_heads), ptr(plan.dup_count),
                                   ptr(plan.workspace), plan.workspace.numel(), stream_ptr()), "dr_hash_sort_slots")
    return plan


def emb_plan_set_small_limit(limit):
    """largest shared-row list the plan's one-block LDS sort takes (0: radix path whenever a row is shared); returns the previous"""
    return int(lib().dr_emb_plan_set_small_limit(int(limit)))


def emb_snapshot_sorted_rows(plan, table, num_rows, out):
    """out[i, :] = table[plan.rows[i], :] for every work-list head i of the plan (dr_emb_snapshot_sorted_rows): the x of the slots
    that share rows, taken before K4 updates the table -- needed when the forward did not store `concat`."""
    assert out.is_contiguous() and out.shape[0] >= plan.n and out.shape[1] == table.shape[1]
    check(lib().dr_emb_snapshot_sorted_rows(ptr(plan.rows), ptr(plan.dup_heads), ptr(plan.dup_count), ptr(table), table.shape[1],
                                            int(num_rows), ptr(out), stream_ptr()), "dr_emb_snapshot_sorted_rows")
    return out


def ids_transpose_i32(ids, out=None):
    """ids [B, F] int64 -> [F, B] int32 (field-major), for bf3_wgrad_emb"""
    ids = _c(ids, torch.int64)
    B, F = ids.shape
    if out is None:
        out = torch.empty((F, B), dtype=torch.int32, device=ids.device)
    check(lib().dr_ids_transpose_i32(ptr(ids), B, F, ptr(out), stream_ptr()), "dr_ids_transpose_i32")
    return out


def emb_pool_bwd_sorted(ids, row_base, plan, D, num_rows, grad, d_fm_logit, scale, dst_table, dst_lin=None, dst_bias=None,
                        concat=None, sum_x=None, slot_lin_grad=None, x_sorted=None, parts=3, lin_old_t=None, table_amax=None):
    """K4 on the slot plan.  x_sorted ([B * F, D] scratch; with the FM term and no `concat` it must hold the snapshot of
    emb_snapshot_sorted_rows) also makes the update of rows hit more than 32 times deterministic: it is clobbered.  parts: 1 = the
    update kernel, 2 = the ordered combination of hot rows' parked pieces, 3 = both (dr_emb_pool_bwd_sorted[_parts])."""
    ids = _c(ids, torch.int64)
    B, F = ids.shape
    assert grad.stride(1) == 1
    args = (ptr(ids), ptr(row_base), ptr(plan.rows), ptr(plan.slots), ptr(plan.flags),
            ptr(plan.dup_heads), ptr(plan.dup_count), B, F, D, int(num_rows), ptr(grad),
            grad.stride(0), ptr(concat), concat.stride(0) if concat is not None else 0,
            ptr(sum_x), ptr(d_fm_logit), ptr(slot_lin_grad), float(scale), ptr(dst_table), ptr(dst_lin),
            ptr(dst_bias), ptr(x_sorted))
    if lin_old_t is not None or table_amax is not None:
        # lin_old_t: first-order weights as this step's forward read them, [F, B]: unique rows get one write, no RMW
        # table_amax: running amax record of dst_table (ops.h2_record), raised to the largest magnitude written
        assert lin_old_t is None or (lin_old_t.shape == (F, B) and lin_old_t.is_contiguous() and lin_old_t.dtype == torch.float32)
        check(lib().dr_emb_pool_bwd_sorted_ex(*args, ptr(lin_old_t), int(parts), ptr(table_amax), stream_ptr()), "dr_emb_pool_bwd_sorted_ex")
    elif parts == 3:
        check(lib().dr_emb_pool_bwd_sorted(*args, stream_ptr()), "dr_emb_pool_bwd_sorted")
    else:
        check(lib().dr_emb_pool_bwd_sorted_parts(*args, int(parts), stream_ptr()), "dr_emb_pool_bwd_sorted_parts")


def h2_dgrad_emb_sgd(dy, dy_amax, w: "H2Planes", ids_t, plan, row_base, table, lin_w, lin_old_t, sum_x, d_fm_logit, scale, d_concat,
                     table_amax=None):
    """First-layer dgrad + K4's unique-row pass in one launch (dr_h2_dgrad_emb_sgd): the rows of `table` (and `lin_w`) that exactly one
    slot of the batch looked up receive their SGD update; the gradient rows of all other slots are stored to `d_concat` [M, >= 64 F]
    for emb_pool_bwd_sorted(..., parts=... | 8).  w: the layer's kernel as H2Planes (rows = input columns, >= 64 F of them)."""
    F, M = ids_t.shape
    K = dy.shape[1]
    assert dy.stride(1) == 1 and d_concat.stride(1) == 1 and table.is_contiguous() and table.shape[1] == 64 and sum_x.is_contiguous()
    assert ids_t.dtype == torch.int32 and ids_t.is_contiguous() and dy.shape[0] == M and w.rows >= 64 * F
    assert (lin_w is None) == (lin_old_t is None)
    assert lin_old_t is None or (lin_old_t.shape == (F, M) and lin_old_t.is_contiguous())
    check(lib().dr_h2_dgrad_emb_sgd(ptr(dy), dy.stride(0), ptr(dy_amax), ptr(w.buf), w.plane_stride, w.ld, ptr(w.amax), M, F, K,
                                    ptr(ids_t), ptr(plan.flags), ptr(row_base), ptr(table), ptr(lin_w), ptr(lin_old_t), ptr(sum_x),
                                    ptr(d_fm_logit), float(scale), ptr(d_concat), d_concat.stride(0), ptr(table_amax), stream_ptr()),
          "dr_h2_dgrad_emb_sgd")


def emb_lin_update_unique(ids, row_base, plan, d_fm_logit, scale, dst_lin, slot_lin_grad=None):
    """dst_lin[row] += scale * gradient for every slot whose row is unique in the batch (plan.flags): the part of K4's first-order
    update that emb_pool_bwd_sorted(parts=... | 4) leaves out (dr_emb_lin_update_unique)."""
    ids = _c(ids, torch.int64)
    B, F = ids.shape
    check(lib().dr_emb_lin_update_unique(ptr(ids), ptr(plan.flags), B, F, ptr(row_base), ptr(d_fm_logit), ptr(slot_lin_grad),
                                         float(scale), ptr(dst_lin), stream_ptr()), "dr_emb_lin_update_unique")


def adam_lr_t(lr, beta1, beta2, step):
    """[TF] B15: lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t), t = step counted from 1."""
    return lr * (1.0 - beta2 ** step) ** 0.5 / (1.0 - beta1 ** step)


def emb_pool_bwd_sorted_adam(ids, row_base, plan, D, num_rows, grad, d_fm_logit, lr_t, beta1, beta2, eps, table, m_table,
                             v_table, lin_w=None, m_lin=None, v_lin=None, concat=None, sum_x=None, slot_lin_grad=None, x_sorted=None,
                             lin_old_t=None, table_amax=None):
    """Sorted K4 with a fused row-wise Adam update (see dr_emb_pool_bwd_sorted_adam in include/dr_hotpath.h)."""
    ids = _c(ids, torch.int64)
    B, F = ids.shape
    assert grad.stride(1) == 1
    if lin_old_t is not None or table_amax is not None:
        assert lin_old_t is None or (lin_old_t.shape == (F, B) and lin_old_t.is_contiguous())
        check(lib().dr_emb_pool_bwd_sorted_adam_ex(ptr(ids), ptr(row_base), ptr(plan.rows), ptr(plan.slots), ptr(plan.flags),
                                            ptr(plan.dup_heads), ptr(plan.dup_count), B, F, D, int(num_rows), ptr(grad),
                                            grad.stride(0), ptr(concat), concat.stride(0) if concat is not None else 0,
                                            ptr(sum_x), ptr(d_fm_logit), ptr(slot_lin_grad), float(lr_t), float(beta1),
                                            float(beta2), float(eps), ptr(table), ptr(m_table), ptr(v_table), ptr(lin_w),
                                            ptr(m_lin), ptr(v_lin), ptr(x_sorted), ptr(lin_old_t), ptr(table_amax), stream_ptr()), "dr_emb_pool_bwd_sorted_adam_ex")
        return
    check(lib().dr_emb_pool_bwd_sorted_adam(ptr(ids), ptr(row_base), ptr(plan.rows), ptr(plan.slots), ptr(plan.flags),
                                            ptr(plan.dup_heads), ptr(plan.dup_count), B, F, D, int(num_rows), ptr(grad),
                                            grad.stride(0), ptr(concat), concat.stride(0) if concat is not None else 0,
                                            ptr(sum_x), ptr(d_fm_logit), ptr(slot_lin_grad), float(lr_t), float(beta1),
                                            float(beta2), float(eps), ptr(table), ptr(m_table), ptr(v_table), ptr(lin_w),
                                            ptr(m_lin), ptr(v_lin), ptr(x_sorted), stream_ptr()), "dr_emb_pool_bwd_sorted_adam")


def adam_catchup_rows(ids, row_base, table, m_table, v_table, lin_w, m_lin, v_lin, row_step, upto, stamp, lr, beta1=0.9,
                      beta2=0.999, eps=1e-8):
    """TF's non-lazy sparse Adam evaluated lazily (dr_adam_catchup_rows): replays the decay-only steps the rows named by
    ids [B, F] have missed, up to step `upto`, and stamps them `stamp` in row_step [R] int32."""
    ids = _c(ids, torch.int64)
    B, F = ids.shape
    assert row_step.dtype == torch.int32 and row_step.is_contiguous()
    check(lib().dr_adam_catchup_rows(ptr(ids), B * F, F, ptr(row_base), table.shape[1], ptr(table), ptr(m_table), ptr(v_table),
                                     ptr(lin_w), ptr(m_lin), ptr(v_lin), ptr(row_step), int(upto), int(stamp), float(lr),
                                     float(beta1), float(beta2), float(eps), stream_ptr()), "dr_adam_catchup_rows")


def adam_step(param, grad, m, v, lr_t, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    """Dense Adam step over flat fp32 buffers (in place)."""
    assert param.is_contiguous() and grad.is_contiguous() and m.is_contiguous() and v.is_contiguous()
    n = param.numel()
    assert grad.numel() == n and m.numel() == n and v.numel() == n
    check(lib().dr_adam_step(ptr(param), ptr(grad), ptr(m), ptr(v), n, float(lr_t), float(beta1), float(beta2), float(eps),
                             float(grad_scale), stream_ptr()), "dr_adam_step")


def ftrl_step(param, grad, accum, linear, lr, lr_power=-0.5, l1=0.0, l2=0.0, grad_scale=1.0):
    """Dense FTRL-Proximal step, TensorFlow formulation, over flat fp32 buffers (in place)."""
    assert param.is_contiguous() and grad.is_contiguous() and accum.is_contiguous() and linear.is_contiguous()
    n = param.numel()
    assert grad.numel() == n and accum.numel() == n and linear.numel() == n
    check(lib().dr_ftrl_step(ptr(param), ptr(grad), ptr(accum), ptr(linear), n, float(lr), float(lr_power), float(l1), float(l2),
                             float(grad_scale), stream_ptr()), "dr_ftrl_step")


def linear_bwd_dx_fm(dy, W, d_fm_logit, sum_x, concat, D, FD, out):
    """First-layer dgrad with the FM second-order gradient folded into the epilogue."""
    dy = _rowmajor_ld4(dy)
    W = _rowmajor_ld4(W)
    M, N = dy.shape
    K = W.shape[0]
    check(lib().dr_linear_bwd_dx_fm(ptr(dy), dy.stride(0), ptr(W), W.stride(0), M, K, N, ptr(d_fm_logit), ptr(sum_x),
                                    ptr(concat), concat.stride(0), int(D), int(FD), ptr(out), out.stride(0),
                                    stream_ptr()), "dr_linear_bwd_dx_fm")
    return out


# ----------------------------------------------------------------------------------------------
# K9 / K10   two-tower retrieval
# ----------------------------------------------------------------------------------------------
def _check_inbatch_shapes(q, c, cand_prob, cand_ids, sample_weight):
    """The fused K9 kernels assume a square in-batch problem (one candidate per query): they take B from q and index every
    per-candidate / per-query vector with it.  Anything else must go through the explicit score matrix (sbcnm.Retrieval does)."""
    B, D = q.shape
    if c.dim() != 2 or c.shape[0] != B or c.shape[1] != D:
        raise ValueError("in-batch softmax kernels need candidates of shape %s, got %s" % ((B, D), tuple(c.shape)))
    for name, t in (("candidate_sampling_probability", cand_prob), ("candidate_ids", cand_ids), ("sample_weight", sample_weight)):
        if t is not None and t.numel() != B:
            raise ValueError("%s must have %d elements, got %d" % (name, B, t.numel()))


def inbatch_softmax_fwd(q, c, cand_prob=None, cand_ids=None, sample_weight=None, inv_temperature=1.0):
    q = _c(q, torch.float32)
    c = _c(c, torch.float32)
    _check_inbatch_shapes(q, c, cand_prob, cand_ids, sample_weight)
    B, D = q.shape
    dev = q.device
    row_lse = torch.empty(B, dtype=torch.float32, device=dev)
    pos = torch.empty(B, dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    nbytes = lib().dr_inbatch_softmax_workspace_bytes(B)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    check(lib().dr_inbatch_softmax_fwd(ptr(q), ptr(c), B, D, ptr(cand_prob), ptr(cand_ids), ptr(sample_weight),
                                       float(inv_temperature), ptr(row_lse), ptr(pos), ptr(loss), ptr(ws), nbytes,
                                       stream_ptr()), "dr_inbatch_softmax_fwd")
    return loss, row_lse, pos


def inbatch_softmax_grad_scores(q, c, row_lse, d_loss, cand_prob=None, cand_ids=None, sample_weight=None,
                                inv_temperature=1.0):
    _check_inbatch_shapes(q, c, cand_prob, cand_ids, sample_weight)
    B, D = q.shape
    q = _c(q, torch.float32)
    c = _c(c, torch.float32)
    G = torch.empty((B, _pad4(B)), dtype=torch.float32, device=q.device)[:, :B]
    # with a workspace the pass may run on the f16x2 register-split kernel (it needs the candidates as fp16 planes)
    nbytes = lib().dr_inbatch_softmax_workspace_bytes(B)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=q.device)
    check(lib().dr_inbatch_softmax_grad_scores_ws(ptr(q), ptr(c), B, D, ptr(cand_prob), ptr(cand_ids), ptr(sample_weight),
                                                  float(inv_temperature), ptr(row_lse), float(d_loss), ptr(G), G.stride(0),
                                                  ptr(ws), nbytes, stream_ptr()), "dr_inbatch_softmax_grad_scores_ws")
    return G


def scores_nt(a, b, out=None):
    a = _rowmajor_ld4(a)
    b = _rowmajor_ld4(b)
    M, D = a.shape
    N = b.shape[0]
    if out is None:
        out = torch.empty((M, _pad4(N)), dtype=torch.float32, device=a.device)[:, :N]
    check(lib().dr_scores_nt(ptr(a), a.stride(0), ptr(b), b.stride(0), M, N, D, ptr(out), out.stride(0), stream_ptr()),
          "dr_scores_nt")
    return out


def topk_state(Bq, k, device):
    return (torch.empty((Bq, k), dtype=torch.float32, device=device), torch.empty((Bq, k), dtype=torch.int64, device=device))


def topk_select(scores, k, index_base=0, init=True, state=None):
    """row-wise top-k of an explicit [Bq, n] score matrix, folded into `state` (scores, index)."""
    assert scores.stride(1) == 1
    Bq, n = scores.shape
    if state is None:
        state = topk_state(Bq, k, scores.device)
    check(lib().dr_topk_select(ptr(scores), scores.stride(0), Bq, n, int(k), int(index_base), int(bool(init)),
                               ptr(state[0]), ptr(state[1]), stream_ptr()), "dr_topk_select")
    return state


class TopKIndex:
    """The corpus side of an exact top-K index (dr_topk_index_build; BruteForce.index / Streaming's candidates of the reference:
    candidates are handed over once, queries arrive many times): the corpus' amax record and fp16 planes, which dr_topk_mips would
    otherwise derive from the corpus on every call.  `topk_mips(q, index, k)` takes it in place of the corpus; the result is the same
    bit for bit.  The planes follow the corpus tensor: written in place since the build (`_version`), they are rebuilt on the next use.
    Corpora the planes do not cover (D not a multiple of 4, D > 512, empty) keep `buf` None and take the plain path."""

    def __init__(self, cand):
        # `cand` must be the caller's own fp32 tensor: staleness is detected through ITS version counter (a copy made here would
        # never see the caller's later writes; writes through raw pointers -- another library -- are not seen either: call rebuild())
        self.cand = _c(cand, torch.float32)
        self.follows_caller = self.cand is cand
        N, D = self.cand.shape
        self.buf = None
        self._able = N > 0 and D % 4 == 0 and 4 <= D <= 512 and self.cand.is_cuda
        self._ver = None
        self.ensure_fresh()

    def _wanted(self):
        # the planes are only read by the f16x2 scan on the matrix-pipe GEMM mode (dr_topk_mips_indexed ignores them otherwise): a second
        # copy of the corpus is not allocated for a split that will not use it -- it is built on the first use that does
        return self._able and get_gemm_split() == "f16x2" and get_gemm_mode() == "bf16x3"

    def ensure_fresh(self):
        if not self._wanted():
            return
        if self.buf is None:
            N, D = self.cand.shape
            self.buf = torch.empty(lib().dr_topk_index_bytes(N, D) // 4, dtype=torch.float32, device=self.cand.device)
            assert self.buf.data_ptr() % 256 == 0
            self._ver = None
        if self._ver != self.cand._version:
            self.rebuild()

    def rebuild(self):
        """Recompute the record and the planes from the corpus as it stands (also after writes torch's version counter cannot see)."""
        if self.buf is not None:
            N, D = self.cand.shape
            check(lib().dr_topk_index_build(ptr(self.cand), N, D, ptr(self.buf), self.buf.numel() * 4, stream_ptr()), "dr_topk_index_build")
            self._ver = self.cand._version

    @property
    def shape(self):
        return self.cand.shape


def topk_mips(q, cand, k, index_base=0, init=True, state=None, workspace=None):
    """cand: the corpus [N, D] fp32, or a TopKIndex built from it."""
    q = _c(q, torch.float32)
    index = cand if isinstance(cand, TopKIndex) else None
    cand = index.cand if index is not None else _c(cand, torch.float32)
    Bq, D = q.shape
    N = cand.shape[0]
    if state is None:
        state = topk_state(Bq, k, q.device)
    if workspace is None:
        workspace = torch.empty(max(1, lib().dr_topk_workspace_bytes(Bq, N, int(k)) // 4), dtype=torch.float32, device=q.device)
    if index is not None:
        index.ensure_fresh()
    if index is not None and index.buf is not None and index._wanted():
        rc = lib().dr_topk_mips_indexed(ptr(q), Bq, ptr(cand), ptr(index.buf), N, D, int(k), int(index_base), int(bool(init)),
                                        ptr(state[0]), ptr(state[1]), ptr(workspace), workspace.numel() * 4, stream_ptr())
    else:
        rc = lib().dr_topk_mips(ptr(q), Bq, ptr(cand), N, D, int(k), int(index_base), int(bool(init)), ptr(state[0]),
                                ptr(state[1]), ptr(workspace), workspace.numel() * 4, stream_ptr())
    if rc == _lib.DR_ESHAPE:
        raise ValueError("Tried to retrieve k={k} top items, but candidate batch too small."
                         "To resolve this, 1. increase batch-size, 2. set `drop_remainder`=True, "
                         "3. set `handle_incomplete_batches`=True in constructor.".format(k=k))
    check(rc, "dr_topk_mips")
    return state


IVF_BUILD_MAX_LISTS = 8192          # IVB_MAXLIST of csrc/ivf.hip


def ivf_build_lists(assign, nlist):
    """(order [N] int64, list_start [nlist + 1] int64): the vectors grouped by coarse list, input order kept inside a list
    (dr_ivf_build_lists: stable counting sort on the device)."""
    assign = _c(assign, torch.int64)
    N = assign.numel()
    if int(nlist) > IVF_BUILD_MAX_LISTS:
        # more coarse lists than the kernel's LDS histogram holds (8192; the reference's own examples use 100): a stable sort by list
        # through torch -- index build, not the search path
        valid = (assign >= 0) & (assign < int(nlist))
        key = torch.where(valid, assign, torch.full_like(assign, int(nlist)))
        order = torch.sort(key, stable=True).indices[:int(valid.sum())]
        list_start = torch.zeros(int(nlist) + 1, dtype=torch.int64, device=assign.device)
        list_start[1:] = torch.cumsum(torch.bincount(assign[valid], minlength=int(nlist)), 0)
        return order, list_start
    order = torch.empty(max(N, 1), dtype=torch.int64, device=assign.device)[:N]
    list_start = torch.empty(int(nlist) + 1, dtype=torch.int64, device=assign.device)
    nb = lib().dr_ivf_build_workspace_bytes(N, int(nlist))
    ws = torch.empty(max(1, nb // 8), dtype=torch.int64, device=assign.device)
    check(lib().dr_ivf_build_lists(ptr(assign), N, int(nlist), ptr(order), ptr(list_start), ptr(ws), ws.numel() * 8, stream_ptr()),
          "dr_ivf_build_lists")
    return order, list_start


def ivf_pack(cand, order, list_start, ids=None):
    """IVF-Flat storage: lists padded to 64-vector blocks, each block dimension-major.  -> (packed, packed_ids, blk_off)"""
    cand = _c(cand, torch.float32)
    N, D = cand.shape
    nlist = list_start.numel() - 1
    counts = (list_start[1:] - list_start[:-1])
    nblk = (counts + 63) // 64
    blk_off = torch.zeros(nlist + 1, dtype=torch.int64, device=cand.device)
    blk_off[1:] = torch.cumsum(nblk, 0)
    total = int(blk_off[-1].item())
    packed = torch.empty(max(total, 1) * D * 64, dtype=torch.float32, device=cand.device)
    packed_ids = torch.empty(max(total, 1) * 64, dtype=torch.int64, device=cand.device)
    check(lib().dr_ivf_pack(ptr(cand), N, D, ptr(_c(order, torch.int64)), ptr(_c(list_start, torch.int64)), ptr(blk_off), nlist,
                            total, ptr(ids), ptr(packed), ptr(packed_ids), stream_ptr()), "dr_ivf_pack")
    return packed, packed_ids, blk_off


def ivf_scan(q, probes, blk_off, packed, packed_ids, k):
    q = _c(q, torch.float32)
    probes = _c(probes, torch.int64)
    Bq, D = q.shape
    s, i = topk_state(Bq, k, q.device)
    check(lib().dr_ivf_scan(ptr(q), Bq, D, ptr(probes), probes.shape[1], ptr(blk_off), ptr(packed), ptr(packed_ids), int(k), ptr(s),
                            ptr(i), stream_ptr()), "dr_ivf_scan")
    return s, i


def topk_merge(sa, ia, sb, ib, k):
    Bq = sa.shape[0]
    out = topk_state(Bq, k, sa.device)
    check(lib().dr_topk_merge(ptr(sa), ptr(ia), sa.shape[1], ptr(sb), ptr(ib), sb.shape[1], Bq, int(k), ptr(out[0]),
                              ptr(out[1]), stream_ptr()), "dr_topk_merge")
    return out


def rows_scale(x, s, mode=0, fallback=None, out=None):
    """out[r, :] = x[r, :] * f(s[r]) (dr_rows_scale): mode 0 f = s; 1 f = 1 / sqrt(s), rows with s == 0 unchanged; 2 f = 1 / s, rows with
    s == 0 taken from `fallback`."""
    x = _c(x, torch.float32)
    s = _c(s, torch.float32).reshape(-1)
    M, D = x.shape
    assert s.numel() == M and (fallback is None or (fallback.shape == x.shape and fallback.is_contiguous()))
    if out is None:
        out = torch.empty_like(x)
    check(lib().dr_rows_scale(ptr(x), ptr(s), int(mode), ptr(fallback), M, D, ptr(out), stream_ptr()), "dr_rows_scale")
    return out


def rowdot(a, b):
    a = _c(a, torch.float32)
    b = _c(b, torch.float32)
    out = torch.empty((a.shape[0], 1), dtype=torch.float32, device=a.device)
    check(lib().dr_rowdot(ptr(a), ptr(b), a.shape[0], a.shape[1], ptr(out), stream_ptr()), "dr_rowdot")
    return out


def gather_i64(src, idx):
    src = _c(src, torch.int64)
    idx = _c(idx, torch.int64)
    out = torch.empty_like(idx)
    check(lib().dr_gather_i64(ptr(src), src.numel(), ptr(idx), idx.numel(), ptr(out), stream_ptr()), "dr_gather_i64")
    return out


def take_along_rows(arr, idx):
    idx = _c(idx, torch.int64)
    arr = arr if arr.stride(1) == 1 else arr.contiguous()
    B, C = arr.shape
    K = idx.shape[1]
    out = torch.empty((B, K), dtype=arr.dtype, device=arr.device)
    if arr.dtype == torch.float32:
        fn, nm = lib().dr_take_along_rows_f32, "dr_take_along_rows_f32"
    elif arr.dtype == torch.int64:
        fn, nm = lib().dr_take_along_rows_i64, "dr_take_along_rows_i64"
    else:
        raise TypeError("take_along_rows supports float32 / int64, got %s" % arr.dtype)
    check(fn(ptr(arr), arr.stride(0), B, C, ptr(idx), K, ptr(out), stream_ptr()), nm)
    return out


def topk_hits(pos, topk, ks, hits):
    """hits[t] += number of rows whose positive is within the top ks[t] ([TF] in_top_k semantics)."""
    pos = _c(pos.reshape(-1), torch.float32)
    topk = _c(topk, torch.float32)
    check(lib().dr_topk_hits(ptr(pos), ptr(topk), pos.numel(), topk.shape[1], ptr(ks), ks.numel(), ptr(hits), stream_ptr()),
          "dr_topk_hits")


def exclude_adjust(scores, ids, exclude):
    scores = _c(scores, torch.float32)
    ids = _c(ids, torch.int64)
    exclude = _c(exclude, torch.int64)
    out = torch.empty_like(scores)
    check(lib().dr_exclude_adjust(ptr(scores), ptr(ids), scores.shape[0], scores.shape[1], ptr(exclude), exclude.shape[1],
                                  ptr(out), stream_ptr()), "dr_exclude_adjust")
    return out


def logits_adjust(logits, labels=None, cand_prob=None, cand_ids=None, add_label_scale=0.0):
    logits = _c(logits, torch.float32)
    labels = _c(labels, torch.float32) if labels is not None else None
    out = torch.empty_like(logits)
    check(lib().dr_logits_adjust(ptr(logits), ptr(labels), logits.shape[0], logits.shape[1], ptr(cand_prob), ptr(cand_ids),
                                 float(add_label_scale), ptr(out), stream_ptr()), "dr_logits_adjust")
    return out


def topk_init(Bq, k, device):
    """empty running top-k lists (scores -inf, index -1)"""
    state = topk_state(Bq, k, device)
    check(lib().dr_topk_select(None, 0, Bq, 0, int(k), 0, 1, ptr(state[0]), ptr(state[1]), stream_ptr()), "dr_topk_select")
    return state


def softmax_ce_rows(logits, labels, inv_temperature=1.0, sample_weight=None):
    logits = _c(logits, torch.float32)
    labels = _c(labels, torch.float32)
    B, C = logits.shape
    row = torch.empty(B, dtype=torch.float32, device=logits.device)
    loss = torch.empty(1, dtype=torch.float32, device=logits.device)
    check(lib().dr_softmax_ce_rows(ptr(logits), ptr(labels), B, C, float(inv_temperature), ptr(sample_weight), ptr(row),
                                   ptr(loss), stream_ptr()), "dr_softmax_ce_rows")
    return loss.reshape(())


def softmax_ce_rows_bwd(logits, labels, inv_temperature, sample_weight, d_loss, cols=None, out=None):
    """gradient of softmax_ce_rows w.r.t. logits; with `cols` scattered into the pre-zeroed matrix `out`."""
    logits = _c(logits, torch.float32)
    labels = _c(labels, torch.float32)
    B, C = logits.shape
    if out is None:
        out = torch.empty((B, C), dtype=torch.float32, device=logits.device)
    check(lib().dr_softmax_ce_rows_bwd(ptr(logits), ptr(labels), B, C, float(inv_temperature), ptr(sample_weight), float(d_loss),
                                       ptr(_c(cols, torch.int64)) if cols is not None else None, ptr(out), out.stride(0),
                                       stream_ptr()), "dr_softmax_ce_rows_bwd")
    return out


def emb_pack_grads(pos, D, d_concat, concat, sum_x, d_fm_logit, out_rows, out_lin=None, bias_sum=None, unique_flags=None):
    pos = _c(pos, torch.int64)
    B, F = pos.shape
    if unique_flags is not None:          # de-duplicated exchange: several slots per destination, shared rows accumulate (zero-filled buffers)
        check(lib().dr_emb_pack_grads_dedup(ptr(pos), ptr(unique_flags), B, F, D, ptr(d_concat), d_concat.stride(0), ptr(concat),
                                            concat.stride(0) if concat is not None else 0, ptr(sum_x), ptr(d_fm_logit), ptr(out_rows),
                                            ptr(out_lin), ptr(bias_sum), stream_ptr()), "dr_emb_pack_grads_dedup")
        return
    check(lib().dr_emb_pack_grads(ptr(pos), B, F, D, ptr(d_concat), d_concat.stride(0), ptr(concat),
                                  concat.stride(0) if concat is not None else 0, ptr(sum_x), ptr(d_fm_logit), ptr(out_rows),
                                  ptr(out_lin), ptr(bias_sum), stream_ptr()), "dr_emb_pack_grads")


# ---- GEMM product mode (include/dr_hotpath.h: DR_GEMM_BF16X3 / DR_GEMM_NATIVE_F32) ------------------------------
GEMM_BF16X3, GEMM_NATIVE_F32 = 0, 1
_GEMM_MODES = {"bf16x3": GEMM_BF16X3, "native": GEMM_NATIVE_F32}


def set_gemm_mode(mode):
    """'bf16x3' (default: fp32 products as six bf16 MFMA products of exact three-way splits) or 'native'
    (v_mfma_f32_32x32x2_f32).  Returns the previous mode's name.  Process-wide."""
    m = _GEMM_MODES[mode] if isinstance(mode, str) else int(mode)
    prev = _lib.lib().dr_set_gemm_mode(m)
    if prev < 0:
        raise ValueError("unknown GEMM mode %r" % (mode,))
    return "native" if prev == GEMM_NATIVE_F32 else "bf16x3"


def get_gemm_mode():
    return "native" if _lib.lib().dr_get_gemm_mode() == GEMM_NATIVE_F32 else "bf16x3"


GEMM_SPLIT_BF16X3, GEMM_SPLIT_F16X2 = 0, 1
_GEMM_SPLITS = {"bf16x3": GEMM_SPLIT_BF16X3, "f16x2": GEMM_SPLIT_F16X2}


def set_gemm_split(split):
    """'f16x2' (default) or 'bf16x3': the operand split the engines (at construction) and the exact top-K scan (at every call) use
    for their register-split GEMMs -- the library's dr_set_gemm_split, the one switch (and the one parser of DR_GEMM_SPLIT).
    Returns the previous split's name.  Process-wide."""
    v = _GEMM_SPLITS[split] if isinstance(split, str) else int(split)
    prev = _lib.lib().dr_set_gemm_split(v)
    if prev < 0:
        raise ValueError("unknown GEMM split %r" % (split,))
    return "f16x2" if prev == GEMM_SPLIT_F16X2 else "bf16x3"


def get_gemm_split():
    return "f16x2" if _lib.lib().dr_get_gemm_split() == GEMM_SPLIT_F16X2 else "bf16x3"


# ---- K7p: first tower layer on pre-split operands ("planes": three bf16 terms per fp32 value, include/dr_hotpath.h) -------
class Planes:
    """planes[3][rows_alloc][ld] bf16 in HBM holding an fp32 matrix [rows, cols] exactly (v = (v2 + v1) + v0).
    rows_alloc = roundup(rows, 32) and ld = roundup(cols, 32), zero-filled: both paddings are read by the GEMMs."""

    def __init__(self, rows, cols, device):
        self.rows, self.cols = int(rows), int(cols)
        self.rows_alloc = (self.rows + 31) // 32 * 32
        self.ld = (self.cols + 31) // 32 * 32
        self.buf = torch.zeros((3, self.rows_alloc, self.ld), dtype=torch.bfloat16, device=device)

    @property
    def plane_stride(self):
        return self.rows_alloc * self.ld


def bf3_split(src, planes: Planes, row_offset=0, col_offset=0, transpose=False):
    """fp32 [R, C] -> planes[p][row_offset + r][col_offset + c] (or [row_offset + c][col_offset + r] when transpose)."""
    assert src.dim() == 2 and src.stride(1) == 1 and src.dtype == torch.float32
    R, C = src.shape
    check(lib().dr_bf3_split(ptr(src), src.stride(0), R, C, ptr(planes.buf), planes.plane_stride, planes.ld, row_offset,
                             col_offset, 1 if transpose else 0, stream_ptr()), "dr_bf3_split")
    return planes


def bf3_join(planes: Planes, out=None):
    if out is None:
        out = torch.empty((planes.rows, planes.cols), dtype=torch.float32, device=planes.buf.device)
    check(lib().dr_bf3_join(ptr(planes.buf), planes.plane_stride, planes.ld, planes.rows, planes.cols, ptr(out),
                            out.stride(0), stream_ptr()), "dr_bf3_join")
    return out


def bf3_gemm_nt(a: Planes, b: Planes, bias=None, act=0, mask=None, out=None):
    """out[m, n] = act(sum_k A[m, k] B[n, k] + bias[n]), zeroed where mask <= 0.  A = a (rows M), B = b (rows N)."""
    M, N = a.rows, b.rows
    assert a.ld == b.ld, "operands must share the padded reduction length"
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.buf.device)
    assert out.shape == (M, N) and out.stride(1) == 1
    check(lib().dr_bf3_gemm_nt(ptr(a.buf), a.plane_stride, a.ld, ptr(b.buf), b.plane_stride, b.ld, M, N, a.ld, ptr(bias),
                               int(act), ptr(mask), mask.stride(0) if mask is not None else 0, ptr(out), out.stride(0),
                               stream_ptr()), "dr_bf3_gemm_nt")
    return out


def bf3_linear_nt(a, b: Planes, bias=None, act=0, mask=None, accumulate=False, out=None):
    """out[m, n] (+)= act(sum_k a[m, k] B[n, k] + bias[n]) (zeroed where mask <= 0): fp32 activations x pre-split weights.
    b holds the weights as planes with rows = output features and cols = the reduction (W^T for the forward, W for the dgrad)."""
    a = _rowmajor_ld4(a)
    M, K = a.shape
    if a.stride(0) % 4 or a.data_ptr() % 16:          # the kernel reads 16-byte vectors: rows must start 16-byte aligned
        buf = torch.zeros((M, _pad4(K)), dtype=torch.float32, device=a.device)
        buf[:, :K].copy_(a)
        a = buf[:, :K]
    N = b.rows
    assert b.cols == K, "weights' planes must have the reduction length as columns"
    if out is None:
        assert not accumulate
        out = torch.empty((M, _pad4(N)), dtype=torch.float32, device=a.device)[:, :N]
    assert out.shape == (M, N) and out.stride(1) == 1
    check(lib().dr_bf3_linear_nt(ptr(a), a.stride(0), ptr(b.buf), b.plane_stride, b.ld, M, N, K, ptr(bias), int(act), ptr(mask),
                                 mask.stride(0) if mask is not None else 0, int(bool(accumulate)), ptr(out), out.stride(0),
                                 stream_ptr()), "dr_bf3_linear_nt")
    return out


def bf3_linear_nt_pack(dy, w: Planes, pos, d_fm_logit, out_rows, out_lin=None, bias_sum=None, sum_x=None, x=None):
    """First-layer dgrad + emb_pack_grads in one launch (dr_bf3_linear_nt_pack): out_rows[pos[m, f], :] = (dy @ W^T)[m, 64 f : 64 f + 64]
    + d_fm_logit[m] * (sum_x[m] - x[m, 64 f : 64 f + 64]); out_lin[pos[m, f]] = d_fm_logit[m]; bias_sum += sum(d_fm_logit).
    w: the layer's W as planes (rows = input features).  D is 64."""
    dy = _rowmajor_ld4(dy)
    M, K = dy.shape
    pos = _c(pos, torch.int64)
    F = pos.shape[1]
    assert w.cols == K and w.rows >= 64 * F and out_rows.is_contiguous() and out_rows.shape[1] == 64
    check(lib().dr_bf3_linear_nt_pack(ptr(dy), dy.stride(0), ptr(w.buf), w.plane_stride, w.ld, M, w.rows, K, ptr(pos), F,
                                      ptr(d_fm_logit), ptr(sum_x), ptr(x), x.stride(0) if x is not None else 0, ptr(out_rows),
                                      ptr(out_lin), ptr(bias_sum), stream_ptr()), "dr_bf3_linear_nt_pack")
    return out_rows


def h2_linear_nt_pack(dy, dy_amax, w: "H2Planes", pos, d_fm_logit, out_rows, out_lin=None, bias_sum=None, sum_x=None, x=None):
    """bf3_linear_nt_pack in the f16x2 operand mode (dr_h2_linear_nt_pack): dy with its amax record, w the layer's W as H2Planes."""
    dy = _rowmajor_ld4(dy)
    M, K = dy.shape
    pos = _c(pos, torch.int64)
    F = pos.shape[1]
    assert w.cols == K and w.rows >= 64 * F and out_rows.is_contiguous() and out_rows.shape[1] == 64
    check(lib().dr_h2_linear_nt_pack(ptr(dy), dy.stride(0), ptr(dy_amax), ptr(w.buf), w.plane_stride, w.ld, ptr(w.amax), M, w.rows, K,
                                     ptr(pos), F, ptr(d_fm_logit), ptr(sum_x), ptr(x), x.stride(0) if x is not None else 0,
                                     ptr(out_rows), ptr(out_lin), ptr(bias_sum), stream_ptr()), "dr_h2_linear_nt_pack")
    return out_rows


def bf3_cross_fwd(x0, x, wt: Planes, b=None, diag_scale=0.0, want_prod=False, prod=None):
    """DCN cross layer on pre-split weights (wt = W^T planes): out = x0 * (x @ W + b + diag * x) + x; (out, prod) as cross_fwd."""
    M, Dm = x.shape
    assert x0.shape == x.shape and x0.stride(0) == x.stride(0) and wt.rows == Dm and wt.cols == Dm
    ld = x.stride(0)
    out = torch.empty((M, ld), dtype=torch.float32, device=x.device)[:, :Dm]
    if want_prod and prod is None:
        prod = torch.empty((M, ld), dtype=torch.float32, device=x.device)[:, :Dm]
    if prod is not None:
        assert prod.stride(0) == ld
    check(lib().dr_bf3_cross_fwd(ptr(x0), ptr(x), ld, ptr(wt.buf), wt.plane_stride, wt.ld, ptr(b), float(diag_scale), M, Dm, ptr(out),
                                 ptr(prod), stream_ptr()), "dr_bf3_cross_fwd")
    return out, prod


def bf3_emb_linear_fwd(ids, row_base, field_rows_max, table, lin_w, lin_bias, dense_pad, concat, K, wt: Planes, bias, act, sum_x, fm_logit, out,
                       lin_vals_t=None):
    """K3 + first Dense in one launch (dr_bf3_emb_linear_fwd): gathers the field embeddings of ids [M, F] from table [R, 64] as the
    GEMM's activation operand, writes them to concat[:, :64 F], the FM terms to sum_x / fm_logit, and
    out = act([embeddings, dense_pad[:, :K - 64 F]] @ W + bias).  dense_pad: [M, 32] zero-padded dense features (None iff
    K == 64 F); field_rows_max: the longest field's row count.  Raises RuntimeError(DR_ESHAPE) for D != 64, more than 32 dense
    features or a field of more than 2^24 rows."""
    ids = _c(ids, torch.int64)
    M, F = ids.shape
    assert (concat is None or concat.stride(1) == 1) and out.stride(1) == 1 and wt.cols == K and out.shape == (M, wt.rows)
    assert dense_pad is None or (dense_pad.shape == (M, 32) and dense_pad.is_contiguous())
    if lin_vals_t is not None:      # also save every slot's first-order weight, field-major [F, M] (emb_pool_bwd_sorted's lin_old_t)
        assert lin_vals_t.shape == (F, M) and lin_vals_t.is_contiguous() and lin_vals_t.dtype == torch.float32
        check(lib().dr_bf3_emb_linear_fwd_lv(ptr(ids), M, F, ptr(row_base), int(field_rows_max), ptr(table), table.shape[1], ptr(lin_w), ptr(lin_bias),
                                             ptr(dense_pad), ptr(concat), concat.stride(0) if concat is not None else 0, int(K), ptr(wt.buf),
                                             wt.plane_stride, wt.ld, wt.rows, ptr(bias), int(act), ptr(sum_x), ptr(fm_logit), ptr(out), out.stride(0),
                                             ptr(lin_vals_t), stream_ptr()), "dr_bf3_emb_linear_fwd_lv")
        return out
    check(lib().dr_bf3_emb_linear_fwd(ptr(ids), M, F, ptr(row_base), int(field_rows_max), ptr(table), table.shape[1], ptr(lin_w), ptr(lin_bias),
                                      ptr(dense_pad), ptr(concat), concat.stride(0) if concat is not None else 0, int(K), ptr(wt.buf), wt.plane_stride, wt.ld,
                                      wt.rows, ptr(bias), int(act), ptr(sum_x), ptr(fm_logit), ptr(out), out.stride(0), stream_ptr()),
          "dr_bf3_emb_linear_fwd")
    return out


def bf3_wgrad_workspace(R, F, N, device):
    return torch.empty(max(64, lib().dr_bf3_wgrad_workspace_bytes(int(R), int(F), int(N)) // 4), dtype=torch.float32, device=device)


def bf3_wgrad_emb(ids_t, row_base, table, dense_pad, dy, scale, dstW, dstb=None, workspace=None, parts=3):
    """bf3_wgrad of the first tower layer with x gathered from the tables (dr_bf3_wgrad_emb): ids_t [nf, R] int32 field-major,
    dense_pad [R, 32] or None; dstW [64 nf + Nd, N]."""
    nf, R = ids_t.shape
    F, N = dstW.shape
    assert ids_t.dtype == torch.int32 and ids_t.is_contiguous() and dy.stride(1) == 1 and dstW.stride(1) == 1 and dy.shape == (R, N)
    assert dense_pad is None or (dense_pad.shape == (R, 32) and dense_pad.is_contiguous())
    if workspace is None:
        workspace = bf3_wgrad_workspace(R, F, N, dy.device)
    if parts != 3:          # 1: the split-K GEMM into the workspace, 2: the reduce that applies it (may run on another stream)
        check(lib().dr_bf3_wgrad_emb_parts(ptr(ids_t), R, nf, ptr(row_base), ptr(table), table.shape[1], ptr(dense_pad), ptr(dy), dy.stride(0),
                                           F, N, float(scale), ptr(dstW), dstW.stride(0), ptr(dstb), ptr(workspace), workspace.numel() * 4,
                                           int(parts), stream_ptr()), "dr_bf3_wgrad_emb_parts")
        return dstW
    check(lib().dr_bf3_wgrad_emb(ptr(ids_t), R, nf, ptr(row_base), ptr(table), table.shape[1], ptr(dense_pad), ptr(dy), dy.stride(0),
                                 F, N, float(scale), ptr(dstW), dstW.stride(0), ptr(dstb), ptr(workspace), workspace.numel() * 4,
                                 stream_ptr()), "dr_bf3_wgrad_emb")
    return dstW


def bf3_wgrad(x, dy, scale, dstW, dstb=None, workspace=None):
    """dstW += scale * x^T @ dy ; dstb += scale * colsum(dy): register-split wgrad on fp32 activations (deterministic)."""
    R, F = x.shape
    N = dy.shape[1]
    assert x.stride(1) == 1 and dy.stride(1) == 1 and dstW.shape == (F, N) and dstW.stride(1) == 1 and dy.shape[0] == R
    if workspace is None:
        workspace = bf3_wgrad_workspace(R, F, N, x.device)
    check(lib().dr_bf3_wgrad(ptr(x), x.stride(0), ptr(dy), dy.stride(0), R, F, N, float(scale), ptr(dstW), dstW.stride(0),
                             ptr(dstb), ptr(workspace), workspace.numel() * 4, stream_ptr()), "dr_bf3_wgrad")
    return dstW


class WeightPlanes:
    """Both pre-split forms of a Dense kernel W [K, N]: `wt` = W^T planes (rows N: the forward's B operand) and `w` = W planes
    (rows K: the dgrad's B operand).  refresh() after every update of W (two dr_bf3_split launches)."""

    def __init__(self, W):
        K, N = W.shape
        self.W = W
        self.wt = Planes(N, K, W.device)
        self.w = Planes(K, N, W.device)
        self.refresh()

    def refresh(self):
        bf3_split(self.W, self.wt, transpose=True)
        bf3_split(self.W, self.w)
        self._ver = self.W._version

    def ensure_fresh(self):
        """Planes are SHADOW copies: forward / dgrad read them, not W.  The engines refresh them after their own (kernel-side)
        updates; a write to W from outside -- checkpoint restore, W.copy_ / fill_ in a test, a broadcast -- bumps torch's version
        counter of the parameter buffer, which is what this checks at the top of every step (host compare, no launch)."""
        if self.W._version != self._ver:
            self.refresh()


# ---- "f16x2" operand mode of the first tower layer's GEMMs (dr_h2_*, include/dr_hotpath.h) -------------------------------------------
def h2_record(device):
    """An amax record: one uint32 (held as int32) with max |x| of a tensor as float bits; 0 = nothing seen yet."""
    return torch.zeros(1, dtype=torch.int32, device=device)


def h2_amax(src, amax=None, reset=True):
    """amax = max(0 if reset else amax, max |src|) as float bits (dr_h2_amax); src fp32 [R, C] with unit inner stride (or 1-D)."""
    if src.dim() == 1:
        src = src.view(1, -1)
    assert src.dim() == 2 and src.stride(1) == 1 and src.dtype == torch.float32
    if amax is None:
        amax = h2_record(src.device)
    R, C = src.shape
    check(lib().dr_h2_amax(ptr(src), src.stride(0) if R > 1 else max(C, 1), R, C, ptr(amax), 1 if reset else 0, stream_ptr()), "dr_h2_amax")
    return amax


def h2_amax_value(amax):
    """The record as a float (host sync; tests and diagnostics)."""
    return float(amax.view(torch.float32).item())


class H2Planes:
    """planes[2][rows_alloc][ld] fp16 in HBM: the two terms of W * s, s from the record `amax` (shared by the W and W^T planes of one
    weight).  Same padding rules as Planes."""

    def __init__(self, rows, cols, device, amax=None):
        self.rows, self.cols = int(rows), int(cols)
        self.rows_alloc = (self.rows + 31) // 32 * 32
        self.ld = (self.cols + 31) // 32 * 32
        self.buf = torch.zeros((2, self.rows_alloc, self.ld), dtype=torch.float16, device=device)
        self.amax = amax if amax is not None else h2_record(device)

    @property
    def plane_stride(self):
        return self.rows_alloc * self.ld


def h2_split(src, planes: H2Planes, row_offset=0, col_offset=0, transpose=False):
    """fp32 [R, C] * s(planes.amax) -> two fp16 planes (layout of bf3_split).  planes.amax must already cover src (h2_amax)."""
    assert src.dim() == 2 and src.stride(1) == 1 and src.dtype == torch.float32
    R, C = src.shape
    check(lib().dr_h2_split(ptr(src), src.stride(0), R, C, ptr(planes.buf), planes.plane_stride, planes.ld, row_offset, col_offset,
                            1 if transpose else 0, ptr(planes.amax), stream_ptr()), "dr_h2_split")
    return planes


def h2_linear_nt(a, a_amax, b: H2Planes, bias=None, act=0, mask=None, accumulate=False, out=None, out_amax=None):
    """bf3_linear_nt in the f16x2 mode: a fp32 [M, K] with its amax record, b the weights as H2Planes.  out_amax (a record, optional)
    receives max |out|: what the next GEMM needs to take `out` as ITS operand."""
    a = _rowmajor_ld4(a)
    M, K = a.shape
    if a.stride(0) % 4 or a.data_ptr() % 16:
        buf = torch.zeros((M, _pad4(K)), dtype=torch.float32, device=a.device)
        buf[:, :K].copy_(a)
        a = buf[:, :K]
    N = b.rows
    assert b.cols == K, "weights' planes must have the reduction length as columns"
    if out is None:
        assert not accumulate
        out = torch.empty((M, _pad4(N)), dtype=torch.float32, device=a.device)[:, :N]
    assert out.shape == (M, N) and out.stride(1) == 1
    check(lib().dr_h2_linear_nt(ptr(a), a.stride(0), ptr(a_amax), ptr(b.buf), b.plane_stride, b.ld, ptr(b.amax), M, N, K, ptr(bias), int(act),
                                ptr(mask), mask.stride(0) if mask is not None else 0, int(bool(accumulate)), ptr(out), out.stride(0),
                                ptr(out_amax), stream_ptr()), "dr_h2_linear_nt")
    return out


def h2_cross_fwd(x0, x, x_amax, wt: H2Planes, b=None, diag_scale=0.0, want_prod=False, prod=None, out_amax=None):
    """bf3_cross_fwd in the f16x2 mode (x_amax: the record of x; out_amax: receives the record of out)."""
    M, Dm = x.shape
    assert x0.shape == x.shape and x0.stride(0) == x.stride(0) and wt.rows == Dm and wt.cols == Dm
    ld = x.stride(0)
    out = torch.empty((M, ld), dtype=torch.float32, device=x.device)[:, :Dm]
    if want_prod and prod is None:
        prod = torch.empty((M, ld), dtype=torch.float32, device=x.device)[:, :Dm]
    if prod is not None:
        assert prod.stride(0) == ld
    check(lib().dr_h2_cross_fwd(ptr(x0), ptr(x), ld, ptr(x_amax), ptr(wt.buf), wt.plane_stride, wt.ld, ptr(wt.amax), ptr(b), float(diag_scale),
                                M, Dm, ptr(out), ptr(prod), ptr(out_amax), stream_ptr()), "dr_h2_cross_fwd")
    return out, prod


def h2_emb_linear_fwd(ids, row_base, field_rows_max, table, table_amax, lin_w, lin_bias, dense_pad, dense_amax, concat, K, wt: H2Planes, bias, act,
                      sum_x, fm_logit, out, lin_vals_t=None):
    """bf3_emb_linear_fwd in the f16x2 mode (dr_h2_emb_linear_fwd): table_amax >= max |table|, dense_amax the record of dense_pad."""
    ids = _c(ids, torch.int64)
    M, F = ids.shape
    assert (concat is None or concat.stride(1) == 1) and out.stride(1) == 1 and wt.cols == K and out.shape == (M, wt.rows)
    assert dense_pad is None or (dense_pad.shape == (M, 32) and dense_pad.is_contiguous())
    assert lin_vals_t is None or (lin_vals_t.shape == (F, M) and lin_vals_t.is_contiguous() and lin_vals_t.dtype == torch.float32)
    check(lib().dr_h2_emb_linear_fwd(ptr(ids), M, F, ptr(row_base), int(field_rows_max), ptr(table), table.shape[1], ptr(table_amax), ptr(lin_w),
                                     ptr(lin_bias), ptr(dense_pad), ptr(dense_amax), ptr(concat), concat.stride(0) if concat is not None else 0,
                                     int(K), ptr(wt.buf), wt.plane_stride, wt.ld, ptr(wt.amax), wt.rows, ptr(bias), int(act), ptr(sum_x),
                                     ptr(fm_logit), ptr(out), out.stride(0), ptr(lin_vals_t), stream_ptr()), "dr_h2_emb_linear_fwd")
    return out


def h2_wgrad(x, x_amax, dy, dy_amax, scale, dstW, dstb=None, workspace=None):
    """bf3_wgrad in the f16x2 mode."""
    R, F = x.shape
    N = dy.shape[1]
    assert x.stride(1) == 1 and dy.stride(1) == 1 and dstW.shape == (F, N) and dstW.stride(1) == 1 and dy.shape[0] == R
    if workspace is None:
        workspace = bf3_wgrad_workspace(R, F, N, x.device)
    check(lib().dr_h2_wgrad(ptr(x), x.stride(0), ptr(x_amax), ptr(dy), dy.stride(0), ptr(dy_amax), R, F, N, float(scale), ptr(dstW),
                            dstW.stride(0), ptr(dstb), ptr(workspace), workspace.numel() * 4, stream_ptr()), "dr_h2_wgrad")
    return dstW


def h2_wgrad_emb(ids_t, row_base, table, table_amax, dense_pad, dense_amax, dy, dy_amax, scale, dstW, dstb=None, workspace=None, parts=3):
    """bf3_wgrad_emb in the f16x2 mode."""
    nf, R = ids_t.shape
    F, N = dstW.shape
    assert ids_t.dtype == torch.int32 and ids_t.is_contiguous() and dy.stride(1) == 1 and dstW.stride(1) == 1 and dy.shape == (R, N)
    assert dense_pad is None or (dense_pad.shape == (R, 32) and dense_pad.is_contiguous())
    if workspace is None:
        workspace = bf3_wgrad_workspace(R, F, N, dy.device)
    check(lib().dr_h2_wgrad_emb(ptr(ids_t), R, nf, ptr(row_base), ptr(table), table.shape[1], ptr(table_amax), ptr(dense_pad), ptr(dense_amax),
                                ptr(dy), dy.stride(0), ptr(dy_amax), F, N, float(scale), ptr(dstW), dstW.stride(0), ptr(dstb), ptr(workspace),
                                workspace.numel() * 4, int(parts), stream_ptr()), "dr_h2_wgrad_emb")
    return dstW


class H2WeightPlanes:
    """WeightPlanes in the f16x2 mode: W^T and W as two fp16 planes each, one amax record; refresh() = dr_h2_amax + two dr_h2_split."""

    def __init__(self, W, double_buffer=False):
        """double_buffer: refresh() writes the W image (`w`) and the record into a SECOND buffer pair and then makes them current, so a
        kernel still reading the previous `w` (with the previous record) may run beside the refresh -- the caller keeps the H2Planes
        object it took before calling refresh().  `wt` is single: nothing may read it while a refresh runs."""
        K, N = W.shape
        self.W = W
        self.amax = h2_record(W.device)
        self.wt = H2Planes(N, K, W.device, self.amax)
        self.w = H2Planes(K, N, W.device, self.amax)
        self._w_alt = H2Planes(K, N, W.device, h2_record(W.device)) if double_buffer else None
        self._parts = None
        self.refresh()

    def refresh(self):
        # record + both images in two launches (dr_h2_refresh_weight: no memset, no atomic pass; same record, same planes as
        # dr_h2_amax + two dr_h2_split)
        if self._parts is None:
            self._parts = torch.zeros(256, dtype=torch.int32, device=self.W.device)
        K, N = self.W.shape
        if self._w_alt is not None:
            self.w, self._w_alt = self._w_alt, self.w
            self.amax = self.w.amax
            self.wt.amax = self.amax
        check(lib().dr_h2_refresh_weight(ptr(self.W), self.W.stride(0), K, N, ptr(self.w.buf), self.w.plane_stride, self.w.ld,
                                         ptr(self.wt.buf), self.wt.plane_stride, self.wt.ld, ptr(self.amax), ptr(self._parts),
                                         stream_ptr()), "dr_h2_refresh_weight")
        self._ver = self.W._version

    def ensure_fresh(self):
        if self.W._version != self._ver:
            self.refresh()


def planes_worthwhile(M, K, N):
    """Shapes the register-split GEMM (256 x 256 tiles) is built for; smaller layers stay on dr_linear_*."""
    return get_gemm_mode() == "bf16x3" and M >= 2048 and N >= 128 and K >= 64


def bf3_gemm_tn_workspace(R, F, N, device):
    return torch.empty(max(64, lib().dr_bf3_gemm_tn_workspace_bytes(int(R), int(F), int(N)) // 4), dtype=torch.float32,
                       device=device)


def bf3_gemm_tn(x: Planes, y: Planes, scale, dstW, y_colsum=None, dstb=None, workspace=None):
    """dstW[f, n] += scale * sum_r X[r, f] Y[r, n];  dstb += scale * y_colsum."""
    R, F, N = x.rows, x.cols, y.cols
    assert y.rows == R and dstW.shape == (F, N) and dstW.stride(1) == 1
    if workspace is None:
        workspace = bf3_gemm_tn_workspace(R, F, N, x.buf.device)
    check(lib().dr_bf3_gemm_tn(ptr(x.buf), x.plane_stride, x.ld, ptr(y.buf), y.plane_stride, y.ld, R, F, N, float(scale),
                               ptr(dstW), dstW.stride(0), ptr(y_colsum), ptr(dstb), ptr(workspace), workspace.numel() * 4,
                               stream_ptr()), "dr_bf3_gemm_tn")
    return dstW


# ---- CIN (xDeepFM) and the DIN ActivationUnit input ---------------------------------------------------------------------------
ACT_CODES = {None: 0, "linear": 0, "relu": 1, "sigmoid": 2, "tanh": 3}


def cin_fwd(x0, x, W, bias=None, act=2):
    """out[b, f, d] = act(sum_ij W[i * Hk + j, f] x0[b, i, d] x[b, j, d] + bias[f]);  x0 [B, H0, D], x [B, Hk, D], W [H0 * Hk, Fm]."""
    x0, x, W = _c(x0, torch.float32), _c(x, torch.float32), _c(W, torch.float32)
    B, H0, D = x0.shape
    Hk = x.shape[1]
    Fm = W.shape[1]
    assert x.shape == (B, Hk, D) and W.shape[0] == H0 * Hk
    out = torch.empty((B, Fm, D), dtype=torch.float32, device=x0.device)
    check(lib().dr_cin_fwd(ptr(x0), ptr(x), B, H0, Hk, D, ptr(W), Fm, ptr(bias), int(act), ptr(out), stream_ptr()), "dr_cin_fwd")
    return out


def cin_bwd(x0, x, W, act, out, d_out, want_bias=False):
    x0, x, W, d_out = _c(x0, torch.float32), _c(x, torch.float32), _c(W, torch.float32), _c(d_out, torch.float32)
    B, H0, D = x0.shape
    Hk, Fm = x.shape[1], W.shape[1]
    d_x0, d_x, dW = torch.empty_like(x0), torch.empty_like(x), torch.empty_like(W)
    dbias = torch.empty(Fm, dtype=torch.float32, device=x0.device) if want_bias else None
    check(lib().dr_cin_bwd(ptr(x0), ptr(x), B, H0, Hk, D, ptr(W), Fm, int(act), ptr(out), ptr(d_out), ptr(d_x0), ptr(d_x), ptr(dW),
                           ptr(dbias), stream_ptr()), "dr_cin_bwd")
    return d_x0, d_x, dW, dbias


def din_concat_fwd(x, y, mode):
    x, y = _c(x, torch.float32), _c(y, torch.float32)
    B, D = x.shape
    assert y.shape == (B, D)
    n = (3 if mode else 2) * D
    out = torch.zeros((B, _pad4(n)), dtype=torch.float32, device=x.device)[:, :n]
    check(lib().dr_din_concat_fwd(ptr(x), ptr(y), B, D, int(mode), ptr(out), out.stride(0), stream_ptr()), "dr_din_concat_fwd")
    return out


def din_concat_bwd(x, y, mode, d_out):
    B, D = x.shape
    assert d_out.stride(1) == 1
    d_x, d_y = torch.empty_like(x), torch.empty_like(y)
    check(lib().dr_din_concat_bwd(ptr(x), ptr(y), B, D, int(mode), ptr(d_out), d_out.stride(0), ptr(d_x), ptr(d_y), stream_ptr()),
          "dr_din_concat_bwd")
    return d_x, d_y


# ---- element-wise pieces around the GEMM path (csrc/elementwise.hip) ------------------------------------------------------------
def act_fwd_(x, act):
    """in place: x = act(x) for act in {2: sigmoid, 3: tanh} (1 = relu also accepted); x [M, N] with unit column stride"""
    assert x.dim() == 2 and x.stride(1) == 1
    check(lib().dr_act_fwd(ptr(x), x.shape[0], x.shape[1], x.stride(0), int(act), stream_ptr()), "dr_act_fwd")
    return x


def act_bwd_(y, dy, act):
    """in place: dy *= act'(.) expressed through the saved output y"""
    assert y.shape == dy.shape and y.stride(1) == 1 and dy.stride(1) == 1
    check(lib().dr_act_bwd(ptr(y), y.stride(0), ptr(dy), dy.stride(0), y.shape[0], y.shape[1], int(act), stream_ptr()), "dr_act_bwd")
    return dy


def dropout_fwd(x, rate, seed):
    assert x.dim() == 2 and x.stride(1) == 1
    M, N = x.shape
    y = torch.empty((M, _pad4(N)), dtype=torch.float32, device=x.device)[:, :N]
    mask = torch.empty(M * N, dtype=torch.uint8, device=x.device)
    check(lib().dr_dropout_fwd(ptr(x), x.stride(0), M, N, float(rate), int(seed) & 0xFFFFFFFFFFFFFFFF, ptr(y), y.stride(0), ptr(mask),
                               stream_ptr()), "dr_dropout_fwd")
    return y, mask


def dropout_bwd(dy, mask, rate):
    assert dy.dim() == 2 and dy.stride(1) == 1
    M, N = dy.shape
    dx = torch.empty((M, _pad4(N)), dtype=torch.float32, device=dy.device)[:, :N]
    check(lib().dr_dropout_bwd(ptr(dy), dy.stride(0), ptr(mask), M, N, float(rate), ptr(dx), dx.stride(0), stream_ptr()),
          "dr_dropout_bwd")
    return dx


def reduce_sum(x, squared=False, alpha=1.0, out=None, accumulate=False):
    """out[0] (+)= alpha * sum(x) or alpha * sum(x^2) over a contiguous fp32 tensor (deterministic)."""
    x = x if x.is_contiguous() else x.contiguous()
    if out is None:
        out = torch.zeros(1, dtype=torch.float32, device=x.device)
    ws = torch.empty(1024, dtype=torch.float32, device=x.device)
    check(lib().dr_reduce_sum(ptr(x), x.numel(), int(bool(squared)), float(alpha), int(bool(accumulate)), ptr(out), ptr(ws),
                              stream_ptr()), "dr_reduce_sum")
    return out


def copy_nt(src, dst):
    """dst = src as a streaming 16-byte nontemporal copy (dr_copy_nt): the measured HBM ceiling bench.py reports."""
    assert src.is_contiguous() and dst.is_contiguous() and src.numel() * src.element_size() == dst.numel() * dst.element_size()
    check(lib().dr_copy_nt(ptr(src), ptr(dst), src.numel() * src.element_size(), stream_ptr()), "dr_copy_nt")
    return dst


def clock_stamp(buf, index):
    """buf[index] (int64 device tensor) = the device wall clock (100 MHz ticks) when the current stream reaches this point
    (dr_clock_stamp: a one-thread kernel; measurement plumbing for bench.py's exposed-wait report)."""
    assert buf.dtype == torch.int64 and buf.is_contiguous() and 0 <= index < buf.numel()
    check(lib().dr_clock_stamp(buf.data_ptr() + 8 * int(index), stream_ptr()), "dr_clock_stamp")
