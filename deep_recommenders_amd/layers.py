"""Host-side building blocks shared by the reference-shaped model classes: the embedding slab (all
categorical tables of a model in one HBM allocation), and the autograd glue around the C-ABI kernels.

Nothing here computes in PyTorch: forward and backward are launches of the hand-written HIP kernels
(ops.py).  PyTorch supplies device memory, the stream, nn.Parameter bookkeeping and the autograd tape.
"""
import math
from typing import Dict, List, Optional, Sequence

import torch
from torch import nn

from . import ops
from . import feature_column as fc


def _pad4(n):
    return (n + 3) // 4 * 4


# --------------------------------------------------------------------------------------------------
# autograd glue
# --------------------------------------------------------------------------------------------------
class _EmbPoolFn(torch.autograd.Function):
    """K3 forward / K4 backward.  `sparse_lr` None: dense gradient buffers are produced (small tables,
    any torch optimizer).  `sparse_lr` set: the backward applies the fused SGD update in place to the
    slab (no gradient materialised — the only feasible mode for 10 M-row tables)."""

    @staticmethod
    def forward(ctx, table, lin_w, lin_bias, ids, F, col_start, row_base, ld_concat, sparse_lr, second_order=True):
        concat, sum_x, fm = ops.emb_pool_fwd(ids, F, col_start, row_base, table, lin_w, lin_bias, ld_concat=ld_concat,
                                             second_order=second_order)
        ctx.F, ctx.sparse_lr, ctx.second_order = F, sparse_lr, second_order
        ctx.has_bias = lin_bias is not None
        ctx.bias_data = lin_bias.data if lin_bias is not None else None
        ctx.save_for_backward(table, lin_w, ids, col_start, row_base, concat, sum_x)
        ctx.mark_non_differentiable(sum_x)
        return concat, fm, sum_x

    @staticmethod
    def backward(ctx, d_concat, d_fm, _d_sum_x):
        table, lin_w, ids, col_start, row_base, concat, sum_x = ctx.saved_tensors
        F, D = ctx.F, table.shape[1]
        if col_start is None:
            col_start = torch.arange(F + 1, dtype=torch.int32, device=ids.device)
        if d_concat is not None:
            d_concat = d_concat if d_concat.stride(1) == 1 else d_concat.contiguous()
        if d_fm is not None:
            d_fm = d_fm.contiguous()
        if d_concat is None and d_fm is None:
            return (None,) * 10
        has_bias = ctx.has_bias and d_fm is not None
        if not ctx.second_order:          # first-order-only logit: its gradient reaches lin_w / bias, not the rows
            concat, sum_x = None, None
        if ctx.sparse_lr is None:
            g_table = torch.zeros_like(table)
            g_lin = torch.zeros_like(lin_w) if lin_w is not None else None
            g_bias = torch.zeros(1, dtype=torch.float32, device=table.device) if has_bias else None
            ops.emb_pool_bwd(ids, F, col_start, row_base, D, d_concat, concat, sum_x, d_fm, 1.0, g_table, g_lin, g_bias)
            return g_table, g_lin, g_bias, None, None, None, None, None, None, None
        ops.emb_pool_bwd(ids, F, col_start, row_base, D, d_concat, concat, sum_x, d_fm, -float(ctx.sparse_lr),
                         table.data, lin_w.data if lin_w is not None else None, ctx.bias_data if has_bias else None)
        return None, None, None, None, None, None, None, None, None, None


class _LinFieldsFn(torch.autograd.Function):
    """Per-field first-order outputs [B, F] (FNN's bias-free Dense(1) over each indicator column, fnn.py:53-64)."""

    @staticmethod
    def forward(ctx, lin_w, ids, F, col_start, row_base):
        ctx.F = F
        ctx.save_for_backward(lin_w, ids, col_start, row_base)
        return ops.lin_fields_fwd(ids, F, col_start, row_base, lin_w)

    @staticmethod
    def backward(ctx, d_out):
        lin_w, ids, col_start, row_base = ctx.saved_tensors
        g = torch.zeros_like(lin_w)
        d_out = d_out if d_out.stride(1) == 1 else d_out.contiguous()
        ops.lin_fields_bwd(ids, ctx.F, col_start, row_base, d_out, 1.0, g)
        return g, None, None, None, None


class _Fm2Fn(torch.autograd.Function):
    """K6 stand-alone FM second-order term."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.fm2_fwd(x)

    @staticmethod
    def backward(ctx, d_out):
        (x,) = ctx.saved_tensors
        return ops.fm2_bwd(x, d_out.reshape(-1).contiguous())


class _MlpFn(torch.autograd.Function):
    """K7: a whole Dense tower.  acts[i] in {0: linear, 1: relu (fused into the GEMM epilogue; relu' is folded into the dx epilogue
    of the layer above), 2: sigmoid, 3: tanh (dr_act_fwd on the linear output, dr_act_bwd through the saved output)}.
    args: x, acts tuple, then W0, b0, W1, b1, ..."""

    @staticmethod
    def forward(ctx, x, acts, *params):
        n = len(params) // 2
        hs = [x]
        h = x
        for i in range(n):
            h = ops.linear_fwd(h, params[2 * i], params[2 * i + 1], 1 if acts[i] == 1 else 0)
            if acts[i] in (2, 3):
                ops.act_fwd_(h, acts[i])
            hs.append(h)
        ctx.acts = acts
        ctx.save_for_backward(*hs, *params)
        return h

    @staticmethod
    def backward(ctx, dy):
        acts = ctx.acts
        n = len(acts)
        saved = ctx.saved_tensors
        hs, params = saved[:n + 1], saved[n + 1:]
        grads = [None] * (2 * n)
        dy = dy.contiguous()
        for i in range(n - 1, -1, -1):
            W, b = params[2 * i], params[2 * i + 1]
            if acts[i] in (2, 3) or (acts[i] == 1 and i == n - 1):
                # the activation's derivative through the saved output (a relu BELOW the top layer is folded into the dx GEMM)
                dy = ops.act_bwd_(hs[i + 1], dy.clone() if i == n - 1 else dy, acts[i])
            gW = torch.zeros_like(W)
            gb = torch.zeros_like(b) if b is not None else None
            ops.linear_bwd_dw(hs[i], dy, 1.0, gW, gb)
            grads[2 * i], grads[2 * i + 1] = gW, gb
            need_dx = i > 0 or ctx.needs_input_grad[0]
            if need_dx:
                relu_src = hs[i] if (i > 0 and acts[i - 1] == 1) else None
                dy = ops.linear_bwd_dx(dy, W, relu_src)
        return (dy if ctx.needs_input_grad[0] else None, None, *grads)


class _DropoutFn(torch.autograd.Function):
    """tf.nn.dropout(x, rate) (estimator/models/feature_interaction/dnn.py:26-27 of the reference)."""

    @staticmethod
    def forward(ctx, x, rate, seed):
        y, mask = ops.dropout_fwd(x if x.stride(1) == 1 else x.contiguous(), rate, seed)
        ctx.rate = rate
        ctx.save_for_backward(mask)
        return y

    @staticmethod
    def backward(ctx, dy):
        mask, = ctx.saved_tensors
        return ops.dropout_bwd(dy if dy.stride(1) == 1 else dy.contiguous(), mask, ctx.rate), None, None


class _L2Fn(torch.autograd.Function):
    """coeff * sum(w^2): the term a Keras `l2(coeff)` kernel / bias regularizer adds to the model's losses."""

    @staticmethod
    def forward(ctx, w, coeff):
        ctx.coeff = coeff
        ctx.save_for_backward(w)
        return ops.reduce_sum(w, squared=True, alpha=coeff).reshape(())

    @staticmethod
    def backward(ctx, d):
        w, = ctx.saved_tensors
        g = torch.zeros_like(w, memory_format=torch.contiguous_format)
        ops.axpy(2.0 * ctx.coeff * float(d), w.contiguous(), g)
        return g, None


class _CrossLowRankFn(torch.autograd.Function):
    """Low-rank DCN cross layer (keras/models/ranking/dcn.py:83-88): prod = (x U) V + b + diag x ; out = x0 * prod + x, with the
    two GEMMs on dr_linear_* and the combine on dr_cross_fwd(W = NULL) / dr_cross_combine_bwd."""

    @staticmethod
    def forward(ctx, x0, x, U, V, b, diag):
        u = ops.linear_fwd(x, U)                                            # [B, p]
        prod = torch.empty((x.shape[0], x.stride(0)), dtype=torch.float32, device=x.device)[:, :x.shape[1]]
        ops.linear_fwd(u, V, None, 0, out=prod)                             # x U V  (bias and diag are added by the combine)
        out, prod = ops.cross_fwd(x0, x, None, b, diag, prod=prod)
        ctx.diag = diag
        ctx.has_b = b is not None
        ctx.save_for_backward(x0, x, U, V, u, prod)
        return out

    @staticmethod
    def backward(ctx, d_out):
        x0, x, U, V, u, prod = ctx.saved_tensors
        ld = x.stride(0)
        d_out = _ld_like(d_out, ld)
        d_x0 = torch.zeros((x.shape[0], ld), dtype=torch.float32, device=x.device)[:, :x.shape[1]]
        d_x = torch.zeros((x.shape[0], ld), dtype=torch.float32, device=x.device)[:, :x.shape[1]]
        d_prod = ops.cross_combine_bwd(x0, prod, d_out, ctx.diag, d_x0, d_x)
        gV, gU = torch.zeros_like(V), torch.zeros_like(U)
        gb = torch.zeros(x.shape[1], dtype=torch.float32, device=x.device) if ctx.has_b else None
        ops.linear_bwd_dw(u, d_prod, 1.0, gV, gb)                           # dV = u^T d_prod ; db = colsum(d_prod)
        d_u = ops.linear_bwd_dx(d_prod, V)
        ops.linear_bwd_dw(x, d_u, 1.0, gU)                                  # dU = x^T d_u
        ops.linear_bwd_dx(d_u, U, None, accumulate=True, out=d_x)           # d_x += d_u U^T
        return d_x0, d_x, gU, gV, gb, None


def _ld_like(t, ld):
    """a [M, N] tensor with leading dimension `ld` (the cross kernels want x0 / x / prod / gradients on one pitch)"""
    if t.stride(1) == 1 and t.stride(0) == ld:
        return t
    buf = torch.zeros((t.shape[0], ld), dtype=torch.float32, device=t.device)
    buf[:, :t.shape[1]].copy_(t)
    return buf[:, :t.shape[1]]


class _CrossFn(torch.autograd.Function):
    """K8: out = x0 * (x @ W + b + diag*x) + x   (keras/models/ranking/dcn.py:81-88 of the reference)."""

    @staticmethod
    def forward(ctx, x0, x, W, b, diag_scale):
        out, prod = ops.cross_fwd(x0, x, W, b, diag_scale, want_prod=True)
        ctx.diag = diag_scale
        ctx.same = x0.data_ptr() == x.data_ptr()
        ctx.save_for_backward(x0, x, W, b, prod)
        return out

    @staticmethod
    def backward(ctx, d_out):
        x0, x, W, b, prod = ctx.saved_tensors
        M, Dm = x.shape
        ld = prod.stride(0)

        def same_ld(t):
            if t.stride(0) == ld and t.stride(1) == 1:
                return t
            buf = torch.zeros((M, ld), dtype=torch.float32, device=t.device)[:, :Dm]
            buf.copy_(t)
            return buf

        d_out_, x0_ = same_ld(d_out), same_ld(x0)
        d_x0 = torch.zeros((M, ld), dtype=torch.float32, device=x.device)[:, :Dm]
        d_x = torch.zeros((M, ld), dtype=torch.float32, device=x.device)[:, :Dm]
        d_prod = ops.cross_combine_bwd(x0_, prod, d_out_, ctx.diag, d_x0, d_x)
        ops.linear_bwd_dx(d_prod, W, None, accumulate=True, out=d_x)
        gW = torch.zeros_like(W)
        gb = torch.zeros_like(b) if b is not None else None
        ops.linear_bwd_dw(x, d_prod, 1.0, gW, gb)
        return d_x0, d_x, gW, gb, None


def fm_second_order(x: torch.Tensor) -> torch.Tensor:
    return _Fm2Fn.apply(x.contiguous())


def mlp(x: torch.Tensor, weights: Sequence[torch.Tensor], biases: Sequence[Optional[torch.Tensor]],
        acts: Sequence[int]) -> torch.Tensor:
    params = []
    for W, b in zip(weights, biases):
        params += [W, b]
    return _MlpFn.apply(x, tuple(int(a) for a in acts), *params)


def cross(x0, x, W, b, diag_scale=0.0):
    return _CrossFn.apply(x0, x, W, b, float(diag_scale))


def cross_low_rank(x0, x, U, V, b, diag_scale=0.0):
    x0 = ops._rowmajor_ld4(x0)
    x = _ld_like(ops._rowmajor_ld4(x), x0.stride(0))
    return _CrossLowRankFn.apply(x0, x, U, V, b, float(diag_scale))


def dropout(x, rate, seed):
    return _DropoutFn.apply(x, float(rate), int(seed))


def l2_penalty(w, coeff):
    return _L2Fn.apply(w, float(coeff))


# --------------------------------------------------------------------------------------------------
# initialisers ([TF] defaults)
# --------------------------------------------------------------------------------------------------
def truncated_normal_(t: torch.Tensor, std: float, mean: float = 0.0):
    return nn.init.trunc_normal_(t, mean=mean, std=std, a=mean - 2 * std, b=mean + 2 * std)


def glorot_uniform_(t: torch.Tensor):
    fan_in, fan_out = t.shape[0], t.shape[1]
    limit = math.sqrt(6.0 / (fan_in + fan_out))
    return nn.init.uniform_(t, -limit, limit)


# --------------------------------------------------------------------------------------------------
# the embedding slab
# --------------------------------------------------------------------------------------------------
class EmbeddingSlab(nn.Module):
    """All categorical tables of one model in a single fp32 HBM slab [R, D] (+ first-order weights
    [R] and bias [1] when indicator columns are given), addressed by per-field base rows.

    Mirrors what F separate `DenseFeatures(embedding_column)` layers + `DenseFeatures(indicator
    columns) -> Dense(1)` hold in the reference (keras/models/ranking/fm.py:47-52, deepfm.py:24-29),
    but gathers every field in one fused kernel launch."""

    def __init__(self, embedding_columns: Sequence[fc.EmbeddingColumn],
                 indicator_columns: Optional[Sequence[fc.IndicatorColumn]] = None, device="cuda"):
        super().__init__()
        if len(embedding_columns) == 0:
            raise ValueError("at least one embedding column is required")
        dims = {c.dimension for c in embedding_columns}
        if len(dims) != 1:
            raise ValueError("FM-family models stack the field embeddings: all dimensions must be equal, got {}".format(
                sorted(dims)))
        self.D = dims.pop()
        if self.D % 4 != 0 or not (4 <= self.D <= 256):
            raise ValueError("embedding dimension must be a multiple of 4 in [4, 256] for the fused kernel")
        self.keys: List[str] = [c.categorical_column.key for c in embedding_columns]
        if len(set(self.keys)) != len(self.keys):
            raise ValueError("duplicate embedding column keys")
        self.columns: Dict[str, fc.CategoricalColumn] = {c.categorical_column.key: c.categorical_column
                                                         for c in embedding_columns}
        self.has_linear = indicator_columns is not None and len(indicator_columns) > 0
        if self.has_linear:
            ind_keys = [c.categorical_column.key for c in indicator_columns]
            if sorted(ind_keys) != sorted(self.keys):
                raise ValueError("indicator and embedding columns must wrap the same categorical columns "
                                 "(as every reference model builds them); got {} vs {}".format(ind_keys, self.keys))
        self.base: Dict[str, int] = {}
        r = 0
        for k in self.keys:
            self.base[k] = r
            r += self.columns[k].num_buckets
        self.R = r
        self.table = nn.Parameter(torch.empty((self.R, self.D), dtype=torch.float32, device=device))
        for c in embedding_columns:   # [TF] B4: truncated normal, sigma = 1/sqrt(D)
            k = c.categorical_column.key
            rows = self.table.data[self.base[k]:self.base[k] + self.columns[k].num_buckets]
            if c.initializer is not None:
                c.initializer(rows)
            else:
                truncated_normal_(rows, 1.0 / math.sqrt(self.D))
        if self.has_linear:           # Dense(1, kernel_initializer="zeros") / linear_model zeros
            self.lin_w = nn.Parameter(torch.zeros(self.R, dtype=torch.float32, device=device))
            self.lin_bias = nn.Parameter(torch.zeros(1, dtype=torch.float32, device=device))
        else:
            self.lin_w, self.lin_bias = None, None
        self.sparse_lr: Optional[float] = None   # set -> fused in-kernel SGD on the slab
        self._rb_cache = {}

    # per-key views (weight import/export; TF names: <scope>/<key>_embedding/embedding_weights)
    def embedding_weights(self, key: str) -> torch.Tensor:
        return self.table.data[self.base[key]:self.base[key] + self.columns[key].num_buckets]

    def linear_weights(self, key: str) -> torch.Tensor:
        return self.lin_w.data[self.base[key]:self.base[key] + self.columns[key].num_buckets]

    def transform(self, inputs: Dict[str, object], field_keys: Sequence[str]):
        """raw features -> (ids [B, C] int64, col_start int32 [F+1] or None, row_base int64 [F])"""
        dev = self.table.device
        mats = [self.columns[k].ids(inputs[k], dev) for k in field_keys]
        widths = [m.shape[1] for m in mats]
        ids = mats[0] if len(mats) == 1 else torch.cat(mats, dim=1)
        ck = (tuple(field_keys), tuple(widths))
        cached = self._rb_cache.get(ck)
        if cached is None:
            row_base = torch.tensor([self.base[k] for k in field_keys], dtype=torch.int64, device=dev)
            if all(w == 1 for w in widths) and len(widths) <= 64:
                col_start = None
            else:
                cs = [0]
                for w in widths:
                    cs.append(cs[-1] + w)
                col_start = torch.tensor(cs, dtype=torch.int32, device=dev)
            cached = (col_start, row_base)
            self._rb_cache[ck] = cached
        return ids.contiguous(), cached[0], cached[1]

    def forward(self, inputs: Dict[str, object], field_keys: Sequence[str], ld_concat: Optional[int] = None,
                second_order: bool = True):
        """-> concat [B, ld] (first F*D columns valid), fm_logit [B] (first-order + bias + second-order; with
        second_order=False the first-order + bias only: WDL's "wide" logit), sum_x"""
        ids, col_start, row_base = self.transform(inputs, field_keys)
        F = len(field_keys)
        concat, fm, sum_x = _EmbPoolFn.apply(self.table, self.lin_w, self.lin_bias, ids, F, col_start, row_base,
                                             ld_concat, self.sparse_lr, second_order)
        return concat, fm, sum_x

    def first_order_fields(self, inputs: Dict[str, object], field_keys: Sequence[str]):
        """[B, F]: every field's own first-order output sum_bag w[id] (no bias) -- FNN's `concat_weights`."""
        ids, col_start, row_base = self.transform(inputs, field_keys)
        return _LinFieldsFn.apply(self.lin_w, ids, len(field_keys), col_start, row_base)
