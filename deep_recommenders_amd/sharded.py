"""Row-sharded DeepFM over N ranks (one process per GPU, torch.distributed; backend "nccl" == RCCL over xGMI).

New design — the reference is single-process CPU, so nothing here mirrors reference
code.  Tables are split row-wise: row `id` of every field lives on rank id % N (round-robin spreads hot rows),
the batch is data-parallel (B examples per rank, weak scaling), the dense tower is replicated.

One step on every rank:
  1. hash keys -> ids (K1)                                   local
  2. bucket the B*F slots by owner rank (dr_shard_bucket_ids) local, integer
  3. all-to-all: split sizes, then owner-local row ids        RCCL  (C1: 8 B per slot)
  4. owners gather the requested rows (dr_rows_gather)        local HBM
  5. all-to-all: rows [*, D] (+ first-order weights) back     RCCL  (C2: 4*D B per slot)
  6. fused pool + first-order + FM over the received rows (K3 with `pos` as ids) -> concat, fm_logit
  7. dense tower forward / loss / backward (K7, K11)          local MFMA
  8. K4 into a packed per-slot gradient buffer -> all-to-all gradients to the owners (C3) ->
     owners scatter-add with the SGD step fused (dr_rows_scatter_add)
  9. one flat all-reduce of the dense-tower gradients (C4) -> dr_axpy into the replicated weights

All compute goes through a `prims` object: `HipPrims` (the HIP kernels, the only implementation shipped
here) — the world_size-2 gloo tests substitute an oracle-backed implementation to check this exchange logic
on CPU, where the kernels cannot run.
"""
import math
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

from . import ops


def _pad4(n):
    return (n + 3) // 4 * 4


class HipPrims:
    """The product implementation: every primitive is a launch of a hand-written gfx950 kernel."""
    hash_bucket_i64 = staticmethod(ops.hash_bucket_i64)
    shard_bucket_ids = staticmethod(ops.shard_bucket_ids)
    rows_gather = staticmethod(ops.rows_gather)
    rows_scatter_add = staticmethod(ops.rows_scatter_add)
    emb_pool_fwd = staticmethod(ops.emb_pool_fwd)
    emb_pool_bwd = staticmethod(ops.emb_pool_bwd)
    linear_fwd = staticmethod(ops.linear_fwd)
    linear_bwd_dx = staticmethod(ops.linear_bwd_dx)
    linear_bwd_dw = staticmethod(ops.linear_bwd_dw)
    bce_fwd_bwd = staticmethod(ops.bce_fwd_bwd)
    axpy = staticmethod(ops.axpy)


class ShardedEmbeddingExchange:
    """Steps 2-6 and 8 above for one (ids [B, F]) batch."""

    def __init__(self, num_fields, vocab_per_field, dim, world, rank, device, prims=None, group=None):
        self.F, self.V, self.D = num_fields, vocab_per_field, dim
        self.world, self.rank, self.dev = world, rank, device
        self.rows_per_shard = (vocab_per_field + world - 1) // world
        self.local_rows = num_fields * self.rows_per_shard
        self.p = prims if prims is not None else HipPrims
        self.group = group
        self._zero_base = torch.zeros(num_fields, dtype=torch.int64, device=device)
        self._col_start = torch.arange(num_fields + 1, dtype=torch.int32, device=device)
        self._st = None

    def _a2a(self, out, inp, out_splits, in_splits):
        dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group)
        return out

    def forward(self, ids, table_local, lin_local, lin_bias, ld_concat, concat=None, sum_x=None, fm_logit=None):
        B, F = ids.shape
        n, D, W = B * F, self.D, self.world
        counts, send_rows, pos = self.p.shard_bucket_ids(ids, self.rows_per_shard, W)
        recv_counts = torch.empty_like(counts)
        dist.all_to_all_single(recv_counts, counts, group=self.group)                 # split sizes
        send_splits = [int(v) for v in counts.tolist()]                               # host sync: exact sizes
        recv_splits = [int(v) for v in recv_counts.tolist()]
        n_recv = sum(recv_splits)
        recv_rows = torch.empty(n_recv, dtype=torch.int64, device=ids.device)
        self._a2a(recv_rows, send_rows, recv_splits, send_splits)                     # C1
        rows_buf, lin_buf = self.p.rows_gather(recv_rows, table_local, lin_local)     # owner-side gather
        got_rows = torch.empty((n, D), dtype=torch.float32, device=ids.device)
        self._a2a(got_rows, rows_buf, send_splits, recv_splits)                       # C2
        got_lin = None
        if lin_local is not None:
            got_lin = torch.empty(n, dtype=torch.float32, device=ids.device)
            self._a2a(got_lin, lin_buf, send_splits, recv_splits)
        # K3 over the received rows: ids := position in the receive buffer, table := receive buffer
        concat, sum_x, fm_logit = self.p.emb_pool_fwd(pos, F, None if F <= 64 else self._col_start, self._zero_base,
                                                      got_rows, got_lin, lin_bias, ld_concat=ld_concat, concat=concat,
                                                      sum_x=sum_x, fm_logit=fm_logit)
        self._st = (pos, send_splits, recv_splits, recv_rows, n, n_recv, lin_local is not None)
        return concat, sum_x, fm_logit

    def backward(self, d_concat, d_fm_logit, concat, sum_x, scale, table_local, lin_local, g_bias=None):
        pos, send_splits, recv_splits, recv_rows, n, n_recv, has_lin = self._st
        D, F = self.D, self.F
        g_rows = torch.zeros((n, D), dtype=torch.float32, device=pos.device)          # packed per-slot gradients
        g_lin = torch.zeros(n, dtype=torch.float32, device=pos.device) if has_lin else None
        self.p.emb_pool_bwd(pos, F, self._col_start, self._zero_base, D, d_concat, concat, sum_x, d_fm_logit, 1.0,
                            g_rows, g_lin, g_bias)
        recv_g = torch.empty((n_recv, D), dtype=torch.float32, device=pos.device)
        self._a2a(recv_g, g_rows, recv_splits, send_splits)                           # C3
        recv_gl = None
        if has_lin:
            recv_gl = torch.empty(n_recv, dtype=torch.float32, device=pos.device)
            self._a2a(recv_gl, g_lin, recv_splits, send_splits)
        self.p.rows_scatter_add(recv_rows, recv_g, recv_gl, scale, table_local, lin_local)


class ShardedDeepFMEngine:
    """DeepFM training step with row-sharded tables; same model / loss / fused SGD as engine.DeepFMEngine.
    `batch` is the per-rank batch; the loss is the mean over the global batch (world * batch)."""

    def __init__(self, num_fields, vocab_per_field, dim, dnn_units: Sequence[int], batch, num_dense=0, lr=0.01,
                 device="cuda", world=None, rank=None, seed=42, prims=None, group=None, lin_init_std=0.0,
                 init_tables=None):
        self.world = world if world is not None else dist.get_world_size(group)
        self.rank = rank if rank is not None else dist.get_rank(group)
        self.F, self.V, self.D, self.B = num_fields, vocab_per_field, dim, batch
        self.Nd, self.lr, self.dev, self.group = num_dense, lr, device, group
        self.p = prims if prims is not None else HipPrims
        F, V, D, B, W = self.F, self.V, self.D, self.B, self.world
        self.ex = ShardedEmbeddingExchange(F, V, D, W, self.rank, device, self.p, group)
        rps = self.ex.rows_per_shard
        g = torch.Generator(device=device)
        g.manual_seed(seed)               # dense tower: same seed on every rank -> identical replicas
        f32 = dict(dtype=torch.float32, device=device)
        # ---- replicated dense parameters in ONE flat buffer (single all-reduce bucket) ----------------
        self.in_dim = F * D + num_dense
        self.ld = _pad4(self.in_dim)
        units = list(dnn_units) + [1]
        shapes, d = [], self.in_dim
        for u in units:
            shapes.append((d, u))
            d = u
        total = sum(k * _pad4(u) + _pad4(u) for k, u in shapes) + 4
        self.flat_params = torch.zeros(total, **f32)
        self.flat_grads = torch.zeros(total, **f32)
        self.Ws, self.bs, self.gWs, self.gbs = [], [], [], []
        off = 0
        for k, u in shapes:
            pu = _pad4(u)
            limit = math.sqrt(6.0 / (k + u))
            Wfull = self.flat_params[off:off + k * pu].view(k, pu)
            Wfull[:, :u].copy_((torch.rand((k, u), device=device, generator=g) * 2 - 1) * limit)
            self.Ws.append(Wfull[:, :u])
            self.gWs.append(self.flat_grads[off:off + k * pu].view(k, pu)[:, :u])
            off += k * pu
            self.bs.append(self.flat_params[off:off + u])
            self.gbs.append(self.flat_grads[off:off + u])
            off += pu
        self.lin_bias = self.flat_params[off:off + 1]
        self.g_lin_bias = self.flat_grads[off:off + 1]
        self.acts = [1] * len(dnn_units) + [0]
        # ---- this rank's table shard -------------------------------------------------------------------
        gt = torch.Generator(device=device)
        gt.manual_seed(seed * 1000 + 17 + self.rank)
        self.table = torch.empty((F * rps, D), **f32)
        if init_tables is not None:
            full_table, full_lin = init_tables           # tests: shard a given global table
            self.lin_w = torch.zeros(F * rps, **f32)
            for f in range(F):
                ids = torch.arange(self.rank, V, W)
                self.table[f * rps:f * rps + len(ids)] = full_table[f * V + ids].to(device)
                self.lin_w[f * rps:f * rps + len(ids)] = full_lin[f * V + ids].to(device)
        else:
            std = 1.0 / math.sqrt(D)
            chunk = 1 << 24
            for r0 in range(0, F * rps, chunk):
                self.table[r0:r0 + chunk].normal_(0.0, std, generator=gt).clamp_(-2 * std, 2 * std)
            self.lin_w = torch.zeros(F * rps, **f32)
            if lin_init_std > 0:
                self.lin_w.normal_(0.0, lin_init_std, generator=gt)
        self.col_buckets = torch.full((F,), V, dtype=torch.int64, device=device)
        # ---- activations ----------------------------------------------------------------------------------
        self.ids = torch.empty((B, F), dtype=torch.int64, device=device)
        self.concat = torch.zeros((B, self.ld), **f32)
        self.sum_x = torch.empty((B, D), **f32)
        self.fm_logit = torch.empty((B,), **f32)
        self.hs = [torch.empty((B, _pad4(u)), **f32)[:, :u] for u in units]
        self.dhs = [torch.empty((B, _pad4(u)), **f32)[:, :u] for u in units[:-1]]
        self.d_concat = torch.empty((B, self.ld), **f32)
        self.prob = torch.empty((B,), **f32)
        self.d_logit = torch.empty((B,), **f32)
        self.loss = torch.zeros(1, **f32)
        self.ws = torch.empty(1024, **f32)

    def train_step(self, keys, dense, labels):
        p, F, D, W = self.p, self.F, self.D, self.world
        p.hash_bucket_i64(keys, self.col_buckets, out=self.ids)
        self.ex.forward(self.ids, self.table, self.lin_w, self.lin_bias, self.ld, concat=self.concat, sum_x=self.sum_x,
                        fm_logit=self.fm_logit)
        if self.Nd:
            self.concat[:, F * D:F * D + self.Nd].copy_(dense)
        x = self.concat[:, :self.in_dim]
        for i, (Wt, b) in enumerate(zip(self.Ws, self.bs)):
            p.linear_fwd(x, Wt, b, self.acts[i], out=self.hs[i])
            x = self.hs[i]
        p.bce_fwd_bwd(self.fm_logit, labels, ops.LOSS_SIGMOID_CE, workspace=self.ws, logits_b=self.hs[-1],
                      out=(self.prob, self.d_logit, self.loss))
        # ---- backward: dense tower gradients into the flat bucket ----------------------------------------
        self.flat_grads.zero_()
        dy = self.d_logit.reshape(-1, 1)
        for i in range(len(self.Ws) - 1, -1, -1):
            xin = self.concat[:, :self.in_dim] if i == 0 else self.hs[i - 1]
            if i > 0:
                dx = self.dhs[i - 1]
                rs = self.hs[i - 1] if self.acts[i - 1] else None
            else:
                dx = self.d_concat[:, :self.in_dim]
                rs = None
            p.linear_bwd_dx(dy, self.Ws[i], relu_src=rs, out=dx)
            p.linear_bwd_dw(xin, dy, 1.0, self.gWs[i], self.gbs[i])
            dy = dx
        # ---- embedding backward: packed gradients -> owners, SGD fused in the owner-side scatter ------------
        self.ex.backward(self.d_concat, self.d_logit, self.concat, self.sum_x, -self.lr / W, self.table, self.lin_w,
                         g_bias=self.g_lin_bias)
        # ---- dense tower: one all-reduce, then w += -(lr / W) * sum_r g_r  (mean over the global batch) ------
        dist.all_reduce(self.flat_grads, group=self.group)
        p.axpy(-self.lr / W, self.flat_grads, self.flat_params)
        return self.loss
