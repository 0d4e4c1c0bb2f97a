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
  8. per-slot gradients packed into the send layout (dr_emb_pack_grads) -> all-to-all to the owners (C3) ->
     owners apply them with the sorted plain read-modify-write K4 (SGD step fused)
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
    emb_pack_grads = staticmethod(ops.emb_pack_grads)
    emb_sort_slots = staticmethod(ops.emb_sort_slots)
    emb_pool_bwd_sorted = staticmethod(ops.emb_pool_bwd_sorted)
    linear_fwd = staticmethod(ops.linear_fwd)
    linear_bwd_dx = staticmethod(ops.linear_bwd_dx)
    linear_bwd_dw = staticmethod(ops.linear_bwd_dw)
    bce_fwd_bwd = staticmethod(ops.bce_fwd_bwd)
    axpy = staticmethod(ops.axpy)


class ShardedEmbeddingExchange:
    """Steps 2-6 and 8 above for one (ids [B, F]) batch.

    Owner side: the received row list is viewed as [n/32, 32] "examples x fields" (padded with -1) so that the
    single-GPU sorted K4 (sort on a side stream during the forward, one plain read-modify-write per unique row)
    applies the returned gradients; the requesting side packs its per-slot gradients straight into the all-to-all
    send layout (dr_emb_pack_grads: a permutation, no atomics)."""

    GROUP = 32      # pseudo-fields per pseudo-example of the owner-side slot list

    def __init__(self, num_fields, vocab_per_field, dim, world, rank, device, prims=None, group=None):
        self.F, self.V, self.D = num_fields, vocab_per_field, dim
        self.world, self.rank, self.dev = world, rank, device
        self.rows_per_shard = (vocab_per_field + world - 1) // world
        self.local_rows = num_fields * self.rows_per_shard
        self.p = prims if prims is not None else HipPrims
        self.group = group
        self._zero_base = torch.zeros(num_fields, dtype=torch.int64, device=device)
        self._zero_base_g = torch.zeros(self.GROUP, dtype=torch.int64, device=device)
        self._col_start = torch.arange(num_fields + 1, dtype=torch.int32, device=device)
        self._st = None
        self._cap = 0
        self._rows_pad = self._g_pad = self._gl_pad = self._plan = None
        self._cuda = torch.device(device).type == "cuda"
        self._side = torch.cuda.Stream(device=device) if self._cuda else None
        self._ev_rows = torch.cuda.Event() if self._cuda else None
        self._ev_sorted = torch.cuda.Event() if self._cuda else None

    def _a2a(self, out, inp, out_splits, in_splits):
        dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group)
        return out

    def _ensure_capacity(self, n_recv, device):
        G = self.GROUP
        need = (n_recv + G - 1) // G * G
        if need > self._cap:
            cap = max(need, int(self._cap * 1.25) // G * G)
            self._rows_pad = torch.full((cap,), -1, dtype=torch.int64, device=device)
            self._g_pad = torch.zeros((cap, self.D), dtype=torch.float32, device=device)
            self._gl_pad = torch.zeros(cap, dtype=torch.float32, device=device)
            self._plan = None
            self._cap = cap
        return need

    def forward(self, ids, table_local, lin_local, lin_bias, ld_concat, concat=None, sum_x=None, fm_logit=None):
        B, F = ids.shape
        n, D, W = B * F, self.D, self.world
        counts, send_rows, pos = self.p.shard_bucket_ids(ids, self.rows_per_shard, W)
        recv_counts = torch.empty_like(counts)
        dist.all_to_all_single(recv_counts, counts, group=self.group)                 # split sizes
        send_splits = [int(v) for v in counts.tolist()]                               # host sync: exact sizes
        recv_splits = [int(v) for v in recv_counts.tolist()]
        n_recv = sum(recv_splits)
        n_pad = self._ensure_capacity(n_recv, ids.device)
        rows_pad = self._rows_pad[:n_pad]
        rows_pad[n_recv:].fill_(-1)
        recv_rows = rows_pad[:n_recv]
        self._a2a(recv_rows, send_rows, recv_splits, send_splits)                     # C1
        # the owner-side sort only needs the received row list: side stream, hidden under the exchange + tower
        G = self.GROUP
        if self._cuda:
            self._ev_rows.record()
            with torch.cuda.stream(self._side):
                self._side.wait_event(self._ev_rows)
                self._plan = self.p.emb_sort_slots(rows_pad.view(-1, G), self._zero_base_g, self.local_rows, self._plan)
                self._ev_sorted.record(self._side)
        else:
            self._plan = self.p.emb_sort_slots(rows_pad.view(-1, G), self._zero_base_g, self.local_rows, self._plan)
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
        self._st = (pos, send_splits, recv_splits, n, n_recv, n_pad, lin_local is not None)
        return concat, sum_x, fm_logit

    def pack(self, d_concat, d_fm_logit, concat, sum_x, g_bias=None):
        """requesting side: per-slot gradients into the send layout (local, HBM-bound)"""
        pos, send_splits, recv_splits, n, n_recv, n_pad, has_lin = self._st
        g_rows = torch.empty((n, self.D), dtype=torch.float32, device=pos.device)
        g_lin = torch.empty(n, dtype=torch.float32, device=pos.device) if has_lin else None
        self.p.emb_pack_grads(pos, self.D, d_concat, concat, sum_x, d_fm_logit, g_rows, g_lin, g_bias)
        return g_rows, g_lin

    def exchange_and_apply(self, g_rows, g_lin, scale, table_local, lin_local):
        """C3 + owner-side update (may run on a communication stream, overlapped with the tower's wgrad)"""
        pos, send_splits, recv_splits, n, n_recv, n_pad, has_lin = self._st
        D, G = self.D, self.GROUP
        g_pad, gl_pad, rows_pad = self._g_pad[:n_pad], self._gl_pad[:n_pad], self._rows_pad[:n_pad]
        self._a2a(g_pad[:n_recv], g_rows, recv_splits, send_splits)                   # C3
        if has_lin:
            self._a2a(gl_pad[:n_recv], g_lin, recv_splits, send_splits)
        if self._cuda:
            torch.cuda.current_stream().wait_event(self._ev_sorted)
        self.p.emb_pool_bwd_sorted(rows_pad.view(-1, G), self._zero_base_g, self._plan, D, self.local_rows,
                                   g_pad.view(-1, G * D), None, scale, table_local, lin_local if has_lin else None, None,
                                   slot_lin_grad=gl_pad if has_lin else None)

    def backward(self, d_concat, d_fm_logit, concat, sum_x, scale, table_local, lin_local, g_bias=None):
        g_rows, g_lin = self.pack(d_concat, d_fm_logit, concat, sum_x, g_bias)
        self.exchange_and_apply(g_rows, g_lin, scale, table_local, lin_local)


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
        self._cuda = torch.device(device).type == "cuda"
        self.comm = torch.cuda.Stream(device=device) if self._cuda else None   # backward exchange + owner update
        self.ev_packed = torch.cuda.Event() if self._cuda else None
        self.ev_applied = torch.cuda.Event() if self._cuda else None
        self._events = None

    # ---- per-phase HIP events (same contract as engine.DeepFMEngine) -------------------------------------------
    def enable_kernel_events(self, on: bool):
        self._events = {} if (on and self._cuda) else None

    def _k(self, name, bound, work, fn):
        if self._events is None:
            return fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = fn()
        e.record()
        self._events.setdefault(name, [bound, work, []])[2].append((s, e))
        return r

    def kernel_event_summary(self):
        if self._events is None:
            return {}
        torch.cuda.synchronize()
        out = {}
        for name, (bound, work, evs) in self._events.items():
            ms = sum(s.elapsed_time(e) for s, e in evs) / len(evs)
            key = "alg_bytes" if bound in ("hbm", "xgmi") else "alg_flops"
            out[name] = {"bound": bound, "ms": ms, "n": len(evs), key: work}
        return out

    def train_step(self, keys, dense, labels):
        p, F, D, W, B = self.p, self.F, self.D, self.world, self.B
        self._k("hash_bucket_i64", "hbm", B * F * 16, lambda: p.hash_bucket_i64(keys, self.col_buckets, out=self.ids))
        # steps 2-6: bucket, exchange ids, owner gather, exchange rows, fused pool+FM over the received rows
        self._k("emb_exchange_fwd(bucket+a2a+gather+a2a+K3)", "xgmi", B * F * (8 + 4 * D + 4),
                lambda: self.ex.forward(self.ids, self.table, self.lin_w, self.lin_bias, self.ld, concat=self.concat,
                                        sum_x=self.sum_x, fm_logit=self.fm_logit))
        if self.Nd:
            self.concat[:, F * D:F * D + self.Nd].copy_(dense)
        x = self.concat[:, :self.in_dim]
        for i, (Wt, b) in enumerate(zip(self.Ws, self.bs)):
            self._k("linear_fwd_L%d" % i, "mfma", 2.0 * B * Wt.shape[0] * Wt.shape[1],
                    lambda x=x, Wt=Wt, b=b, i=i: p.linear_fwd(x, Wt, b, self.acts[i], out=self.hs[i]))
            x = self.hs[i]
        p.bce_fwd_bwd(self.fm_logit, labels, ops.LOSS_SIGMOID_CE, workspace=self.ws, logits_b=self.hs[-1],
                      out=(self.prob, self.d_logit, self.loss))
        # ---- backward: dense tower gradients into the flat bucket ----------------------------------------
        self.flat_grads.zero_()
        # dgrad chain first (it produces d_concat, the input of the embedding exchange); wgrads afterwards so that they
        # overlap the backward exchange running on the communication stream
        n_layers = len(self.Ws)
        dys = [None] * n_layers
        dy = self.d_logit.reshape(-1, 1)
        for i in range(n_layers - 1, -1, -1):
            dys[i] = dy
            if i > 0:
                dx = self.dhs[i - 1]
                rs = self.hs[i - 1] if self.acts[i - 1] else None
            else:
                dx = self.d_concat[:, :self.in_dim]
                rs = None
            self._k("linear_bwd_dx_L%d" % i, "mfma", 2.0 * B * self.Ws[i].shape[0] * self.Ws[i].shape[1],
                    lambda dy=dy, i=i, rs=rs, dx=dx: p.linear_bwd_dx(dy, self.Ws[i], relu_src=rs, out=dx))
            dy = dx
        # ---- embedding backward: pack (local), then C3 + owner-side sorted update on the communication stream -------
        g_rows, g_lin = self._k("emb_pack_grads", "hbm", B * F * (12 * D + 4),
                                lambda: self.ex.pack(self.d_concat, self.d_logit, self.concat, self.sum_x, self.g_lin_bias))
        if self._cuda:
            self.ev_packed.record()
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(self.ev_packed)
                self.ex.exchange_and_apply(g_rows, g_lin, -self.lr / W, self.table, self.lin_w)
                self.ev_applied.record(self.comm)
        else:
            self.ex.exchange_and_apply(g_rows, g_lin, -self.lr / W, self.table, self.lin_w)
        for i in range(n_layers - 1, -1, -1):
            xin = self.concat[:, :self.in_dim] if i == 0 else self.hs[i - 1]
            self._k("linear_bwd_dw_L%d" % i, "mfma", 2.0 * B * self.Ws[i].shape[0] * self.Ws[i].shape[1],
                    lambda xin=xin, i=i: p.linear_bwd_dw(xin, dys[i], 1.0, self.gWs[i], self.gbs[i]))
        # ---- dense tower: one all-reduce, then w += -(lr / W) * sum_r g_r  (mean over the global batch) ------
        self._k("allreduce_dense_grads", "xgmi", self.flat_grads.numel() * 4,
                lambda: dist.all_reduce(self.flat_grads, group=self.group))
        p.axpy(-self.lr / W, self.flat_grads, self.flat_params)
        if self._cuda:
            torch.cuda.current_stream().wait_event(self.ev_applied)     # next step's gather must see the update
        return self.loss
