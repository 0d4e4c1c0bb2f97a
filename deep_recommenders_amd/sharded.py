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
from typing import Sequence

import torch
import torch.distributed as dist

from . import ops


def _pad4(n):
    return (n + 3) // 4 * 4


class HipPrims:
    """The product implementation: every primitive is a launch of a hand-written gfx950 kernel."""
    hash_bucket_i64 = staticmethod(ops.hash_bucket_i64)
    shard_bucket_ids = staticmethod(ops.shard_bucket_ids)
    shard_dedup_slots = staticmethod(ops.shard_dedup_slots)
    rows_gather = staticmethod(ops.rows_gather)
    rows_scatter_add = staticmethod(ops.rows_scatter_add)
    emb_pool_fwd = staticmethod(ops.emb_pool_fwd)
    emb_pool_bwd = staticmethod(ops.emb_pool_bwd)
    emb_pack_grads = staticmethod(ops.emb_pack_grads)
    emb_sort_slots = staticmethod(ops.emb_sort_slots)
    new_sort_plan = staticmethod(ops.SortPlan)
    emb_pool_bwd_sorted = staticmethod(ops.emb_pool_bwd_sorted)
    emb_pool_bwd_sorted_adam = staticmethod(ops.emb_pool_bwd_sorted_adam)
    adam_step = staticmethod(ops.adam_step)
    linear_fwd = staticmethod(ops.linear_fwd)
    linear_bwd_dx = staticmethod(ops.linear_bwd_dx)
    linear_bwd_dw = staticmethod(ops.linear_bwd_dw)
    bce_fwd_bwd = staticmethod(ops.bce_fwd_bwd)
    axpy = staticmethod(ops.axpy)
    tower_head_fwd_bwd = staticmethod(ops.tower_head_fwd_bwd)
    tower_tail_fused = staticmethod(ops.tower_tail_fused)
    tower_tail_supported = staticmethod(ops.tower_tail_supported)
    linear_bwd_narrow = staticmethod(ops.linear_bwd_narrow)
    linear_bwd_narrow_supported = staticmethod(ops.linear_bwd_narrow_supported)
    cross_fwd = staticmethod(ops.cross_fwd)
    cross_combine_bwd = staticmethod(ops.cross_combine_bwd)


class TorchDistTransport:
    """The exchange steps over torch.distributed: backend "nccl" == RCCL over xGMI on the GPUs, "gloo" in the CPU tests."""

    def __init__(self, group=None):
        self.group = group

    def alltoall(self, out, inp, out_splits=None, in_splits=None):
        dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group)
        return out

    def allreduce(self, t, async_op=False, op=None):
        return dist.all_reduce(t, op=op if op is not None else dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def allgather(self, out, inp):
        dist.all_gather_into_tensor(out, inp, group=self.group)
        return out


class _DoneWork:
    def wait(self):
        return True


class HostStagedTransport(TorchDistTransport):
    """Device buffers through host memory over a CPU (gloo) group.  RCCL refuses two ranks on one device; this lets TWO ranks share
    ONE GPU, so the N > 1 exchange logic runs with the real HIP kernels on a single-GPU box (tests/test_gpu_sharded_two_rank.py).
    Blocking and slow by construction -- verification only, never a bench path."""

    def _to_host(self, t):
        return t.detach().to("cpu")            # enqueued on the current stream, the host waits for it

    def alltoall(self, out, inp, out_splits=None, in_splits=None):
        h_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(h_out, self._to_host(inp).contiguous(), out_splits, in_splits, group=self.group)
        out.copy_(h_out)
        return out

    def allreduce(self, t, async_op=False, op=None):
        h = self._to_host(t).contiguous()
        dist.all_reduce(h, op=op if op is not None else dist.ReduceOp.SUM, group=self.group)
        t.copy_(h)
        return _DoneWork() if async_op else None

    def allgather(self, out, inp):
        h_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(h_out, self._to_host(inp).contiguous(), group=self.group)
        out.copy_(h_out)
        return out


class CApiTransport:
    """The exchange steps through the C-ABI library (include/dr_collectives.h -> lib/libdr_collectives.so: RCCL underneath, raw
    device pointers + the current hipStream_t) -- what a host that is not PyTorch would call.  torch.distributed is used once, to
    hand rank 0's RCCL rendezvous id to the other ranks (any side channel would do); every exchange of the training step then
    goes through dr_coll_*.  DR_TRANSPORT=capi selects it in bench.py / the engines' default."""

    def __init__(self, world, rank, group=None, id_bytes=None):
        import ctypes
        from . import _coll_lib
        self._C, self._ct = _coll_lib, ctypes
        L = _coll_lib.lib()
        if id_bytes is None:
            buf = ctypes.create_string_buffer(_coll_lib.ID_BYTES)
            if rank == 0:
                _coll_lib.check(L.dr_coll_unique_id(buf), "dr_coll_unique_id")
            box = [bytes(buf.raw)]
            if world > 1:
                dist.broadcast_object_list(box, src=0, group=group)
            id_bytes = box[0]
        self.world, self.rank = world, rank
        self._comm = ctypes.c_void_p()
        _coll_lib.check(L.dr_coll_init(ctypes.byref(self._comm), world, rank, ctypes.create_string_buffer(id_bytes, _coll_lib.ID_BYTES)),
                        "dr_coll_init")

    def close(self):
        if self._comm:
            self._C.lib().dr_coll_destroy(self._comm)
            self._comm = self._ct.c_void_p()

    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    def alltoall(self, out, inp, out_splits=None, in_splits=None):
        assert out.is_contiguous() and inp.is_contiguous() and out.dtype == inp.dtype
        L, W = self._C.lib(), self.world
        row = inp.element_size() * math.prod(inp.shape[1:])          # bytes per leading-dimension element (no indexing: a rank may send nothing)
        if out_splits is None:
            per = inp.shape[0] // W
            in_splits = out_splits = [per] * W
        if not any(in_splits) and not any(out_splits):
            return out
        arr = self._ct.c_int64 * W
        self._C.check(L.dr_coll_alltoallv(self._comm, inp.data_ptr(), arr(*in_splits), out.data_ptr(), arr(*out_splits), row,
                                          self._stream()), "dr_coll_alltoallv")
        return out

    def allreduce(self, t, async_op=False, op=None):
        assert t.is_contiguous()
        if (op is not None and op != dist.ReduceOp.SUM) or t.dtype != torch.float32:
            # the library reduces fp32 sums (the dense-gradient bucket); the odd small reduction (a shard-size MAX) is an all-gather
            # followed by a local reduce
            allv = self.allgather(torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device), t)
            red = {dist.ReduceOp.MAX: allv.amax(0), dist.ReduceOp.MIN: allv.amin(0)}.get(op)
            t.copy_(red if red is not None else allv.sum(0))
            return _DoneWork() if async_op else None
        self._C.check(self._C.lib().dr_coll_allreduce_f32(self._comm, t.data_ptr(), t.numel(), self._stream()), "dr_coll_allreduce_f32")
        return _DoneWork() if async_op else None          # stream-ordered: nothing for the host to wait for

    def allgather(self, out, inp):
        assert out.is_contiguous() and inp.is_contiguous()
        self._C.check(self._C.lib().dr_coll_allgather(self._comm, inp.data_ptr(), out.data_ptr(), inp.numel() * inp.element_size(),
                                                      self._stream()), "dr_coll_allgather")
        return out


def default_transport(world, rank, group=None):
    """torch.distributed unless DR_TRANSPORT=capi asks for the C-ABI exchange library."""
    import os as _os
    if _os.environ.get("DR_TRANSPORT", "torch") == "capi":
        return CApiTransport(world, rank, group)
    return TorchDistTransport(group)


class Route:
    """Everything about one batch that depends only on its keys: where each slot's row lives, how many rows travel to /
    from each rank, the owner-side list of requested rows and its sort plan.  Built by ShardedEmbeddingExchange.route()
    on a dedicated stream, possibly one step ahead of its use."""
    __slots__ = ("pos", "send_splits", "recv_splits", "n", "n_send", "n_recv", "n_pad", "slot", "ready", "sorted", "key", "has_lin",
                 "uniq", "_send_rows", "_counts_host", "_counts_ev", "_finished")


class ShardedEmbeddingExchange:
    """Steps 1-6 and 8 above for one (keys [B, F]) batch.

    route()   steps 1-3 (hash, bucket, split sizes + row ids to the owners) and the owner-side sort plan.  This is the
              only part that needs the host (exact all-to-all split sizes): it runs on its own stream so that its host
              sync covers a handful of small kernels instead of draining the training stream, and it can be issued for the
              NEXT batch while the current step computes.
    forward() steps 4-6: owners gather, rows come back, fused pool + first-order + FM over the received rows.
    pack() / exchange_and_apply()  step 8.

    Owner side: the received row list is viewed as [n/32, 32] "examples x fields" (padded with -1) so that the
    single-GPU sorted K4 (one plain read-modify-write per unique row) applies the returned gradients; the requesting
    side packs its per-slot gradients straight into the all-to-all send layout (dr_emb_pack_grads: a permutation)."""

    GROUP = 32      # pseudo-fields per pseudo-example of the owner-side slot list
    SLOTS = 4       # routes in flight: (micro-batches per step = 2) x (the step being trained + the prefetched one)

    def __init__(self, num_fields, vocab_per_field, dim, world, rank, device, prims=None, group=None, transport=None,
                 alias_world1=True, dedup=None):
        self.F, self.V, self.D = num_fields, vocab_per_field, dim
        # Requester-side de-duplication (DR_SH_DEDUP=1 / dedup=True; round 4): a row that several slots of a micro-batch look up
        # travels ONCE each way -- what [TF] safe_embedding_lookup_sparse's `unique` does inside the lookup (reached from
        # keras/models/ranking/fm.py:57-61 of the reference).  Uniform ids over 10 M rows: nothing to save (99.8 % of a micro-batch's
        # slots are distinct rows); Zipf(1.05) keys: 1.6 x fewer bytes on the wire, for a slot plan + a representative map per
        # micro-batch on the routing stream and fp32 atomics in the pack for the shared rows (their summation order is then not fixed).
        import os as _os
        self.dedup = (_os.environ.get("DR_SH_DEDUP", "0") == "1") if dedup is None else bool(dedup)
        self.world, self.rank, self.dev = world, rank, device
        self.rows_per_shard = (vocab_per_field + world - 1) // world
        self.local_rows = num_fields * self.rows_per_shard
        self.p = prims if prims is not None else HipPrims
        self.group = group
        self.tr = transport if transport is not None else default_transport(world, rank, group)
        # One rank owns everything: every "exchange" is the identity.  The buffers then ALIAS (bucketing writes the owner-side row
        # list, the gather output is the receive buffer, the packed gradients are written where the owner-side K4 reads them) and
        # no collective is issued -- RCCL used to copy 2 x 443 MB to itself per step.  alias_world1=False keeps the
        # collectives (a group of one), which is how the tests push the N > 1 code through RCCL on a single GPU.
        self.local = world == 1 and alias_world1 and (prims is None or prims is HipPrims) and not self.dedup
        self._field_base = torch.arange(num_fields, dtype=torch.int64, device=device) * vocab_per_field
        self._zero_base = torch.zeros(num_fields, dtype=torch.int64, device=device)
        self._zero_base_g = torch.zeros(self.GROUP, dtype=torch.int64, device=device)
        self._col_start = torch.arange(num_fields + 1, dtype=torch.int32, device=device)
        self._col_buckets = torch.full((num_fields,), vocab_per_field, dtype=torch.int64, device=device)
        self._cuda = torch.device(device).type == "cuda"
        self._rs = torch.cuda.Stream(device=device) if self._cuda else None      # routing stream
        # per-slot owner-side buffers (the route of step t is still read by step t's backward while step t+1 is routed)
        self.set_slots(self.SLOTS)

    def set_slots(self, n):
        """Number of routes that may be in flight (2 x micro-batches per step)."""
        self.SLOTS = int(n)
        self._slots = [dict(cap=0, rows_pad=None, g_pad=None, gl_pad=None, plan=None, ids=None, req_plan=None, rep=None)
                       for _ in range(self.SLOTS)]
        self._next_slot = 0

    # bench.py's exchange report: when set, called as phase_timer(tag, bytes_sent_by_this_rank, fn) around every data-path collective
    # (tags "a2a_ids" = C1, "a2a_rows" = C2, "a2a_grads" = C3) -- the engines bracket fn with HIP events
    phase_timer = None

    def _a2a(self, out, inp, out_splits, in_splits, tag=None):
        if self.phase_timer is not None and tag is not None:
            nbytes = inp.numel() * inp.element_size()
            return self.phase_timer(tag, nbytes, lambda: self.tr.alltoall(out, inp, out_splits, in_splits))
        return self.tr.alltoall(out, inp, out_splits, in_splits)

    def _ensure_capacity(self, sl, n_recv, device):
        G = self.GROUP
        need = (n_recv + G - 1) // G * G
        if need > sl["cap"]:
            cap = max(need, int(sl["cap"] * 1.25) // G * G)
            sl["rows_pad"] = torch.full((cap,), -1, dtype=torch.int64, device=device)
            sl["g_pad"] = torch.zeros((cap, self.D), dtype=torch.float32, device=device)
            sl["gl_pad"] = torch.zeros(cap, dtype=torch.float32, device=device)
            # the sort plan is sized for the slot CAPACITY, not for the batch that triggered the growth: n_recv varies from
            # batch to batch, and a later batch with old n_pad < need <= cap must find a large enough workspace
            sl["plan"] = self.p.new_sort_plan(cap, device) if hasattr(self.p, "new_sort_plan") else None
            sl["cap"] = cap
        return need

    # ---- steps 1-3 -------------------------------------------------------------------------------------------
    def route_begin(self, keys, hashed=True, wait_current=True):
        """First half of routing, never blocks the host: hash, bucket by owner, exchange the split sizes, start their copy to
        pinned host memory.  keys [B, F] int64 (raw keys if `hashed`, else ids).  wait_current=False: the caller guarantees
        `keys` is already complete (a prefetched batch), so routing does not queue behind the training stream."""
        sl_i = self._next_slot
        self._next_slot = (sl_i + 1) % self.SLOTS
        sl = self._slots[sl_i]
        W = self.world
        r = Route()
        r.key = (keys.data_ptr(), tuple(keys.shape))
        r.slot = sl_i
        r._finished = False
        ctx = torch.cuda.stream(self._rs) if self._cuda else _NullCtx()
        if self._cuda and wait_current:
            self._rs.wait_stream(torch.cuda.current_stream())        # keys may have been produced on the caller's stream
        if self._cuda and sl.get("free_ev") is not None:
            self._rs.wait_event(sl["free_ev"])       # the step that last used this slot's buffers has applied its gradients
        with ctx:
            if hashed:
                if sl["ids"] is None or sl["ids"].shape != keys.shape:
                    sl["ids"] = torch.empty(keys.shape, dtype=torch.int64, device=keys.device)
                ids = self.p.hash_bucket_i64(keys, self._col_buckets, out=sl["ids"])            # K1
            else:
                ids = keys
            B, F = ids.shape
            r.n = B * F
            if self.local:
                # no peers: the bucketed row list IS the owner-side list (written straight into the padded slot buffer), the split
                # sizes are known without asking the device
                r.n_recv = r.n
                r.n_pad = self._ensure_capacity(sl, r.n, ids.device)
                rows_pad = sl["rows_pad"][:r.n_pad]
                rows_pad[r.n:].fill_(-1)
                _, send_rows, pos = self.p.shard_bucket_ids(ids, self.rows_per_shard, W, send_rows=rows_pad[:r.n])
                r.send_splits = r.recv_splits = [r.n]
                r.n_send, r.uniq = r.n, None
                r._counts_host = r._counts_ev = None
                r.pos, r._send_rows = pos, None
                return r
            r.uniq = None
            if self.dedup:
                # slot plan of THIS micro-batch's ids over the global rows f * V + id (the kernels K4's plan uses), then every
                # slot's representative = the lowest slot looking up the same row; only representatives get a send slot
                if sl["req_plan"] is None and hasattr(self.p, "new_sort_plan"):
                    sl["req_plan"] = self.p.new_sort_plan(r.n, ids.device)
                sl["req_plan"] = self.p.emb_sort_slots(ids, self._field_base, self.F * self.V, sl["req_plan"])
                sl["rep"], r.uniq = self.p.shard_dedup_slots(ids, self._field_base, self.F * self.V, sl["req_plan"], out=sl["rep"])
                counts, send_rows, pos = self.p.shard_bucket_ids(ids, self.rows_per_shard, W, rep=sl["rep"])
            else:
                counts, send_rows, pos = self.p.shard_bucket_ids(ids, self.rows_per_shard, W)
            recv_counts = torch.empty_like(counts)
            self.tr.alltoall(recv_counts, counts)                                         # split sizes
            both = torch.stack([counts, recv_counts])
            if self._cuda:
                if sl.get("counts_host") is None:
                    sl["counts_host"] = torch.empty((2, W), dtype=torch.int64).pin_memory()
                sl["counts_host"].copy_(both, non_blocking=True)
                r._counts_host = sl["counts_host"]
                r._counts_ev = torch.cuda.Event()
                r._counts_ev.record(self._rs)
            else:
                r._counts_host, r._counts_ev = both, None
            r.pos, r._send_rows = pos, send_rows
        return r

    def route_finish(self, r):
        """Second half: read the split sizes (a host wait only for route_begin's small kernels), send the row ids to their
        owners, sort the received list for the owner-side update."""
        if r._finished:
            return r
        sl = self._slots[r.slot]
        if not self.local:
            if r._counts_ev is not None:
                r._counts_ev.synchronize()
            r.send_splits = [int(v) for v in r._counts_host[0].tolist()]
            r.recv_splits = [int(v) for v in r._counts_host[1].tolist()]
            r.n_recv = sum(r.recv_splits)
            r.n_send = sum(r.send_splits)                   # == r.n unless de-duplicated
        ctx = torch.cuda.stream(self._rs) if self._cuda else _NullCtx()
        with ctx:
            if not self.local:
                r.n_pad = self._ensure_capacity(sl, r.n_recv, r.pos.device)
                rows_pad = sl["rows_pad"][:r.n_pad]
                rows_pad[r.n_recv:].fill_(-1)
                self._a2a(rows_pad[:r.n_recv], r._send_rows[:r.n_send], r.recv_splits, r.send_splits, tag="a2a_ids")      # C1
            else:
                rows_pad = sl["rows_pad"][:r.n_pad]
            if self._cuda:
                r.ready = torch.cuda.Event()
                r.ready.record(self._rs)
            # owner-side sort: needs only the received row list
            sl["plan"] = self.p.emb_sort_slots(rows_pad.view(-1, self.GROUP), self._zero_base_g, self.local_rows, sl["plan"])
            if self._cuda:
                r.sorted = torch.cuda.Event()
                r.sorted.record(self._rs)
        r._send_rows = None
        r._finished = True
        return r

    def route(self, keys, hashed=True, wait_current=True):
        """Both halves back to back (a step whose batch was not prefetched)."""
        return self.route_finish(self.route_begin(keys, hashed=hashed, wait_current=wait_current))

    # ---- steps 4-5: owner gather + rows back (communication stream) ---------------------------------------------
    def fetch(self, route, table_local, lin_local):
        r = route
        sl = self._slots[r.slot]
        if self._cuda:
            torch.cuda.current_stream().wait_event(r.ready)
        recv_rows = sl["rows_pad"][:r.n_recv]
        rows_buf, lin_buf = self.p.rows_gather(recv_rows, table_local, lin_local)     # owner-side gather
        r.has_lin = lin_local is not None
        if self.local:
            return rows_buf, lin_buf                                                   # the gather output IS the receive buffer
        got_rows = torch.empty((r.n_send, self.D), dtype=torch.float32, device=recv_rows.device)
        self._a2a(got_rows, rows_buf, r.send_splits, r.recv_splits, tag="a2a_rows")   # C2
        got_lin = None
        if lin_local is not None:
            got_lin = torch.empty(r.n_send, dtype=torch.float32, device=recv_rows.device)
            self._a2a(got_lin, lin_buf, r.send_splits, r.recv_splits)
        r.has_lin = lin_local is not None
        return got_rows, got_lin

    # ---- step 6: K3 over the received rows (ids := position in the receive buffer, table := receive buffer) --------
    def pool(self, route, got_rows, got_lin, lin_bias, ld_concat, concat=None, sum_x=None, fm_logit=None):
        r, F = route, self.F
        if self._cuda:
            cur = torch.cuda.current_stream()
            for t in (r.pos, got_rows, got_lin):
                if t is not None:
                    t.record_stream(cur)
        return self.p.emb_pool_fwd(r.pos, F, None if F <= 64 else self._col_start, self._zero_base, got_rows, got_lin,
                                   lin_bias, ld_concat=ld_concat, concat=concat, sum_x=sum_x, fm_logit=fm_logit)

    def forward(self, route, table_local, lin_local, lin_bias, ld_concat, concat=None, sum_x=None, fm_logit=None):
        got_rows, got_lin = self.fetch(route, table_local, lin_local)
        return self.pool(route, got_rows, got_lin, lin_bias, ld_concat, concat=concat, sum_x=sum_x, fm_logit=fm_logit)

    # ---- step 8 -------------------------------------------------------------------------------------------------
    def pack_buffers(self, route):
        """the send-layout buffers the per-slot gradients of `route` go to (rows [n, D], first-order [n] or None)"""
        r = route
        if self.local:                       # straight into the buffers the owner-side K4 reads (nothing travels)
            sl = self._slots[r.slot]
            g_rows = sl["g_pad"][:r.n]
            g_lin = sl["gl_pad"][:r.n] if r.has_lin else None
        elif r.uniq is not None:             # de-duplicated: shared rows are ACCUMULATED by the pack, so they start from zero
            g_rows = torch.zeros((r.n_send, self.D), dtype=torch.float32, device=r.pos.device)
            g_lin = torch.zeros(r.n_send, dtype=torch.float32, device=r.pos.device) if r.has_lin else None
        else:
            g_rows = torch.empty((r.n, self.D), dtype=torch.float32, device=r.pos.device)
            g_lin = torch.empty(r.n, dtype=torch.float32, device=r.pos.device) if r.has_lin else None
        return g_rows, g_lin

    def pack(self, route, d_concat, d_fm_logit, concat, sum_x, g_bias=None):
        """requesting side: per-slot gradients into the send layout (local, HBM-bound)"""
        g_rows, g_lin = self.pack_buffers(route)
        if route.uniq is not None:
            self.p.emb_pack_grads(route.pos, self.D, d_concat, concat, sum_x, d_fm_logit, g_rows, g_lin, g_bias, unique_flags=route.uniq)
        else:
            self.p.emb_pack_grads(route.pos, self.D, d_concat, concat, sum_x, d_fm_logit, g_rows, g_lin, g_bias)
        return g_rows, g_lin

    def exchange_and_apply(self, route, g_rows, g_lin, scale, table_local, lin_local, adam=None, table_amax=None):
        """C3 + owner-side update (runs on the communication stream, overlapped with the tower).  adam = (lr_t, beta1, beta2,
        eps, m_table, v_table, m_lin, v_lin): one row-wise Adam update per touched row from the SUM of the gradients every
        rank sent for it (the gradients must already be those of the global-mean loss); otherwise dst += scale * sum."""
        r = route
        sl = self._slots[r.slot]
        D, G = self.D, self.GROUP
        g_pad, gl_pad, rows_pad = sl["g_pad"][:r.n_pad], sl["gl_pad"][:r.n_pad], sl["rows_pad"][:r.n_pad]
        if self._cuda:
            cur = torch.cuda.current_stream()
            g_rows.record_stream(cur)
            if g_lin is not None:
                g_lin.record_stream(cur)
        if not self.local:
            self._a2a(g_pad[:r.n_recv], g_rows, r.recv_splits, r.send_splits, tag="a2a_grads")         # C3
            if r.has_lin:
                self._a2a(gl_pad[:r.n_recv], g_lin, r.recv_splits, r.send_splits)
        if self._cuda:
            torch.cuda.current_stream().wait_event(r.sorted)
        if adam is not None:
            lr_t, b1, b2, eps, m_t, v_t, m_l, v_l = adam
            self.p.emb_pool_bwd_sorted_adam(rows_pad.view(-1, G), self._zero_base_g, sl["plan"], D, self.local_rows,
                                            g_pad.view(-1, G * D), None, lr_t, b1, b2, eps, table_local, m_t, v_t,
                                            lin_local if r.has_lin else None, m_l, v_l,
                                            slot_lin_grad=gl_pad if r.has_lin else None, **({"table_amax": table_amax} if table_amax is not None else {}))
        else:
            self.p.emb_pool_bwd_sorted(rows_pad.view(-1, G), self._zero_base_g, sl["plan"], D, self.local_rows,
                                       g_pad.view(-1, G * D), None, scale, table_local, lin_local if r.has_lin else None, None,
                                       slot_lin_grad=gl_pad if r.has_lin else None, **({"table_amax": table_amax} if table_amax is not None else {}))
        if self._cuda:
            sl["free_ev"] = torch.cuda.Event()
            sl["free_ev"].record()


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


# DR_FUSE_PACK's default: on since round 5 IN THE f16x2 SPLIT (its epilogue turns the accumulators through the LDS: world-1 step 1.62 ->
# 1.55 ms, profiles/r05_sharded_fuse_pack.log); in the bf16x3 split the fused form is the old 4-bytes-per-lane epilogue, measured 27 %
# slower than dgrad + pack (400 us against 173 + 143): off there unless DR_FUSE_PACK=1 asks for it
def _fuse_pack_default(h2):
    return "1" if h2 else "0"


class ShardedDeepFMEngine:
    """DeepFM training step with row-sharded tables; same model / loss / fused SGD as engine.DeepFMEngine.
    `batch` is the per-rank batch; the loss is the mean over the global batch (world * batch)."""

    def __init__(self, num_fields, vocab_per_field, dim, dnn_units: Sequence[int], batch, num_dense=0, lr=0.01,
                 device="cuda", world=None, rank=None, seed=42, prims=None, group=None, lin_init_std=0.0,
                 init_tables=None, micro_batches=None, optimizer="sgd", beta1=0.9, beta2=0.999, eps=1e-8, transport=None,
                 alias_world1=True, dedup=None):
        assert optimizer in ("sgd", "adam")
        self.optimizer, self.beta1, self.beta2, self.eps, self.t = optimizer, beta1, beta2, eps, 0
        self.world = world if world is not None else dist.get_world_size(group)
        self.rank = rank if rank is not None else dist.get_rank(group)
        self.F, self.V, self.D, self.B = num_fields, vocab_per_field, dim, batch
        self.Nd, self.lr, self.dev, self.group = num_dense, lr, device, group
        self.p = prims if prims is not None else HipPrims
        F, V, D, B, W = self.F, self.V, self.D, self.B, self.world
        self.ex = ShardedEmbeddingExchange(F, V, D, W, self.rank, device, self.p, group, transport=transport,
                                           alias_world1=alias_world1, dedup=dedup)
        self.tr = self.ex.tr
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
        # fused tower tail (same kernels as engine.DeepFMEngine, writing into the gradient bucket instead of the weights)
        nl = len(self.Ws)
        self.fuse_head = (hasattr(self.p, "tower_head_fwd_bwd") and nl >= 2 and self.Ws[-1].shape[1] == 1
                          and self.Ws[-2].shape[1] <= 32 and self.acts[-2] == 1)
        self.narrow = [hasattr(self.p, "linear_bwd_narrow") and i > 0 and self.acts[i - 1] == 1
                       and self.p.linear_bwd_narrow_supported(B, Wt.shape[0], Wt.shape[1]) for i, Wt in enumerate(self.Ws)]
        if self._cuda:
            self.head_ws = ops.tower_head_workspace(B, device) if self.fuse_head else None
            self.narrow_ws = [ops.linear_bwd_narrow_workspace(B, Wt.shape[0], Wt.shape[1], device) if self.narrow[i] else None
                              for i, Wt in enumerate(self.Ws)]
            self.dw_ws = [ops.linear_bwd_dw_workspace(B, Wt.shape[0], Wt.shape[1], device) for Wt in self.Ws]
        else:
            self.head_ws = None
            self.narrow_ws = [None] * nl
            self.dw_ws = [None] * nl
        self.comm = torch.cuda.Stream(device=device) if self._cuda else None   # every exchange + the owner-side updates
        # micro-batches per step (see train_step); needs the fused head (per-slice loss normalisation) and slices the
        # narrow kernel accepts
        mb = int(micro_batches) if micro_batches else 2
        if not self.fuse_head or B % mb:
            mb = 1
        if optimizer == "adam":
            # Adam is not linear in the gradient: a row hit by both micro-batches must get ONE update from the summed gradient,
            # like the single-GPU engine (and TF) -- so the whole rank batch travels as one exchange
            mb = 1
            self.m_table, self.v_table = torch.zeros_like(self.table), torch.zeros_like(self.table)
            self.m_lin, self.v_lin = torch.zeros_like(self.lin_w), torch.zeros_like(self.lin_w)
            self.flat_m, self.flat_v = torch.zeros_like(self.flat_params), torch.zeros_like(self.flat_params)
        self.mb = mb
        self.ex.set_slots(max(4, 2 * mb))
        Bm = B // mb
        # the tower tail of a micro-batch in ONE pass over its h (round 5: dr_tower_tail_fused, as engine.DeepFMEngine; DR_FUSE_TAIL=0: the
        # head and the narrow backward as two launches).  HipPrims only: the oracle-backed primitives of the gloo tests have no such call.
        import os as _os0
        self.fuse_tail = (self._cuda and hasattr(self.p, "tower_tail_fused") and _os0.environ.get("DR_FUSE_TAIL", "1") == "1"
                          and self.fuse_head and nl >= 3 and self.acts[nl - 3] == 1
                          and self.p.tower_tail_supported(Bm, self.Ws[-2].shape[0], self.Ws[-2].shape[1]))
        self.tail_ws = ops.tower_tail_workspace(Bm, self.Ws[-2].shape[0], device) if self.fuse_tail else None
        self.narrow = [hasattr(self.p, "linear_bwd_narrow") and i > 0 and self.acts[i - 1] == 1
                       and self.p.linear_bwd_narrow_supported(Bm, Wt.shape[0], Wt.shape[1]) for i, Wt in enumerate(self.Ws)]
        self.loss_parts = torch.zeros(mb, **f32)
        # Wide layers on the register-split GEMMs of engine.DeepFMEngine (weights kept as bf16 planes, refreshed after the update):
        # the dgrad of a micro-batch and the wgrad over the whole rank batch.  The forward stays on the in-kernel-split GEMM below
        # 65 536 rows: its 256-row tiles number 128 at a 32 768-row micro-batch, half a machine (267 us either way, measured).
        import os as _os
        use_planes = self._cuda and self.p is HipPrims and _os.environ.get("DR_PLANES", "1") == "1"
        self.wplanes = [ops.WeightPlanes(Wt) if (use_planes and ops.planes_worthwhile(Bm, Wt.shape[0], Wt.shape[1])) else None
                        for Wt in self.Ws]
        self.wg_ws = [ops.bf3_wgrad_workspace(B, Wt.shape[0], Wt.shape[1], device) if self.wplanes[i] is not None else None
                      for i, Wt in enumerate(self.Ws)]
        self.planes_fwd_rows = 65536
        # K3 inside the first layer's GEMM, over the RECEIVED rows: the register-split kernel gathers its activation operand by
        # LDS-DMA and does not care where the rows live -- table := the receive buffer, ids := each slot's position in it, one
        # "field" of n rows (engine.DeepFMEngine's fused first layer; DR_FUSE_K3=0: K3 + a separate GEMM as in round 2).
        self.fuse_k3 = (use_planes and _os.environ.get("DR_FUSE_K3", "1") == "1" and D == 64 and self.Nd <= 32
                        and self.wplanes[0] is not None and B * F <= (1 << 24) and self.acts[0] in (0, 1))
        self.dense_pad = torch.zeros((B, 32), **f32) if (self.fuse_k3 and self.Nd) else None
        # first-layer dgrad and the gradient pack in one launch (dr_h2_linear_nt_pack / dr_bf3_linear_nt_pack): d_concat is never
        # written (- 0.44 GB written, - 0.44 GB read per 65 536 examples).  Needs the layer's weight planes, D == 64 and single-valued
        # fields (one slot per (example, field)).  Round 4's epilogue (bf16x3: 4-byte scattered stores, a lane owning one column of
        # 16 rows) lost to dgrad + pack (400 us against 173 + 143 per half batch); round 5's, in the f16x2 split, turns each
        # accumulator block through the LDS and moves float4s with 8 lanes on a row.  DR_FUSE_PACK=0 / 1 (default: see _fuse_pack_default).
        _will_h2 = (self.fuse_k3 and ops.get_gemm_split() == "f16x2" and not (self.fuse_head and len(self.Ws) - 2 == 0))     # (= self.h2 below)
        self.fuse_pack = (use_planes and _os.environ.get("DR_FUSE_PACK", _fuse_pack_default(_will_h2)) == "1" and D == 64
                          and num_fields <= 64 and not self.ex.dedup)
        # streams for the later micro-batches' fused first layers (see train_step); DR_FWD_STREAMS=0: all on the training stream
        self.fwd_streams = ([torch.cuda.Stream(device=device) for _ in range(max(0, min(mb - 1, 3)))]
                            if (self.fuse_k3 and _os.environ.get("DR_FWD_STREAMS", "1") == "1") else [])
        # The first layer's three GEMMs in the "f16x2" operand mode (round 4; see engine.DeepFMEngine / include/dr_hotpath.h dr_h2_*).  The
        # activation scale of the forward and the wgrad comes from the amax record of the TABLES: every rank keeps a running record of
        # its own shard (K4 raises it) and, with real peers, one 4-byte all-reduce(MAX) behind the step's last owner-side update makes
        # it the bound for the rows any rank may receive in the next step (on the communication stream, long before those rows
        # arrive).  d h0's record comes out of the narrow backward per micro-batch.  DR_GEMM_SPLIT=bf16x3: the six-product mode.
        self.h2 = (self.fuse_k3 and ops.get_gemm_split() == "f16x2" and not (self.fuse_head and nl - 2 == 0))
        if self.h2:
            self.wplanes[0] = ops.H2WeightPlanes(self.Ws[0])
            self.tab_amax_local = ops.h2_amax(self.table)
            # (two buffers for the global bound: the all-reduce behind step t's last update fills the one step t + 1 will read, while
            # step t's wgrad -- whose rows were fetched before that update -- still reads the other)
            self._tab_bufs = None if self.ex.local else [self.tab_amax_local.clone(), self.tab_amax_local.clone()]
            self._tab_i, self._tab_swap = 0, False
            self.tab_amax = self.tab_amax_local if self.ex.local else self._tab_bufs[0]
            if not self.ex.local:
                self.tr.allreduce(self.tab_amax, op=dist.ReduceOp.MAX)
            self._tab_ver = (self.table.data_ptr(), self.table._version)
            self.dense_amax = ops.h2_record(device) if self.dense_pad is not None else None
            self.dh0_amax = [ops.h2_record(device) for _ in range(mb)]
            self.dh0_amax_all = ops.h2_record(device)
            self.x_amax_all = ops.h2_record(device)
        # DR_SH_TRACK_AMAX=1 (tests): the bookkeeping of the table bound -- local running record, the all-reduce(MAX) behind the step's last
        # owner-side update, the two-buffer swap -- WITHOUT the f16x2 kernels, so that the N > 1 control flow of the mode runs wherever
        # the engine runs (world-size-2 gloo on CPU with the oracle-backed primitives: the record is recomputed from the shard with
        # torch after every owner-side update instead of being raised by K4).
        self.track_amax = self.h2 or _os.environ.get("DR_SH_TRACK_AMAX", "0") == "1"
        if self.track_amax and not self.h2:
            self.tab_amax_local = self._amax_of(self.table)
            self._tab_bufs = None if self.ex.local else [self.tab_amax_local.clone(), self.tab_amax_local.clone()]
            self._tab_i, self._tab_swap = 0, False
            self.tab_amax = self.tab_amax_local if self.ex.local else self._tab_bufs[0]
            if not self.ex.local:
                self.tr.allreduce(self.tab_amax, op=dist.ReduceOp.MAX)
        self._ev_every, self._ev_step, self._ev_live = 1, 0, False
        self.k4_first = _os.environ.get("DR_SH_K4_FIRST", "0") == "1"
        self.wgrad_split = _os.environ.get("DR_SH_WGRAD_SPLIT", "0") == "1"
        self._events = None
        self._stamps = None
        self._route = None
        self._done = []
        self._comm_synced = False

    # ---- per-phase HIP events (same contract as engine.DeepFMEngine) -------------------------------------------
    def enable_kernel_events(self, on: bool, every: int = 1):
        """Per-phase HIP events / clock stamps.  every = n: only every n-th train_step is bracketed (the bracketing costs ~6 % of the
        world-1 step: 2.05 vs 1.94 ms); kernel_event_summary / exchange_report average over the bracketed steps."""
        self._events = {} if (on and self._cuda) else None
        self._ev_every, self._ev_step, self._ev_live = max(1, int(every)), 0, bool(on)
        self._stamps = None
        # the collectives themselves (C1 / C2 / C3), apart from the local kernels of their phases: bench.py's per-link rates
        self.ex.phase_timer = (lambda tag, nbytes, fn: self._k(tag, "xgmi", nbytes, fn)) if self._events is not None else None

    def _stall(self, name, stream, event):
        """stream.wait_event(event); when events are on, bracketed by two device clock stamps on `stream` (ops.clock_stamp: one-thread
        kernels writing the 100 MHz wall clock): their distance is the time the stream sat idle waiting for the exchange -- the
        EXPOSED, non-overlapped part of the communication (bench.py sums them).  Not HIP timing events: two event records around a
        cross-stream wait, four waits per step, stretched the step from 1.93 to 4.1 ms (round 4)."""
        if self._events is None or not self._cuda or not self._ev_live:
            stream.wait_event(event)
            return
        if self._stamps is None:
            self._stamps = torch.zeros(1 << 16, dtype=torch.int64, device=self.dev)
            self._stamp_n, self._stamp_names = 0, []
        i = self._stamp_n
        if i + 2 > self._stamps.numel():
            stream.wait_event(event)
            return
        with torch.cuda.stream(stream):
            ops.clock_stamp(self._stamps, i)
            stream.wait_event(event)
            ops.clock_stamp(self._stamps, i + 1)
        self._stamp_n = i + 2
        self._stamp_names.append(name)

    def stall_summary(self):
        """name -> (total microseconds the stream waited there, number of waits) since events were enabled"""
        if getattr(self, "_stamps", None) is None or not self._stamp_n:
            return {}
        torch.cuda.synchronize()
        t = self._stamps[:self._stamp_n].cpu().view(-1, 2)
        out = {}
        for nm, (a, b) in zip(self._stamp_names, t.tolist()):
            tot, n = out.get(nm, (0.0, 0))
            out[nm] = (tot + max(0, b - a) * 0.01, n + 1)          # 100 MHz ticks -> microseconds
        return out

    # Phases that get HIP events in bench.py.  The sharded step issues ~3x the launches of the single-GPU step (two
    # micro-batches, exchanges, routing) and its host thread is the scarcer resource: two event records around every small
    # kernel were enough to leave ~30 us bubbles between kernels, so only the coarse phases are bracketed.
    _TIMED = ("emb_fetch", "emb_pool_fwd", "emb_linear_fwd_L0", "linear_fwd_L0", "linear_bwd_dx_L0", "linear_bwd_dw_L0", "emb_pack_grads",
              "emb_grads", "allreduce_dense_grads", "emb_route", "a2a_")

    def _k(self, name, bound, work, fn):
        if self._events is None or not self._ev_live or not name.startswith(self._TIMED):
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

    def exchange_report(self, steps):
        """What bench.py prints for an N-rank run: per collective the bytes this rank sent, its mean duration and the rate per xGMI link
        (a rank's all-to-all payload leaves over world - 1 links, (world - 1) / world of it); `exposed_us_per_step` = time the training
        stream sat idle waiting for rows / for the last owner-side update (the part of the exchange that no compute hid)."""
        ks = self.kernel_event_summary()
        W = self.world
        rep = {"world": W, "micro_batches": self.mb, "transport": type(self.tr).__name__, "dedup_before_exchange": bool(self.ex.dedup),
               "collectives": {}, "exposed_us_per_step": None}
        for tag, what in (("a2a_ids", "C1 row ids to their owners"), ("a2a_rows", "C2 embedding rows back"), ("a2a_grads", "C3 row gradients to their owners"),
                          ("allreduce_dense_grads", "C4 dense-tower gradients")):
            k = ks.get(tag)
            if k is None:
                continue
            sec = k["ms"] * 1e-3
            sent = float(k["alg_bytes"])
            on_wire = sent * (W - 1) / W if tag != "allreduce_dense_grads" else 2.0 * sent * (W - 1) / W      # ring all-reduce: 2 (W-1)/W
            rep["collectives"][tag] = {"what": what, "bytes_sent_per_rank": int(sent), "avg_us": round(k["ms"] * 1e3, 2), "calls": k["n"],
                                       "GBps_per_rank": round(on_wire / sec / 1e9, 2) if sec > 0 else None,
                                       "GBps_per_link": round(on_wire / max(W - 1, 1) / sec / 1e9, 2) if (sec > 0 and W > 1) else None}
        stalls = self.stall_summary()
        if self._events is not None and self._ev_every > 1:
            steps = (steps + self._ev_every - 1) // self._ev_every          # bracketed steps only
        if stalls and steps:
            rep["exposed_us_per_step"] = round(sum(t for t, _ in stalls.values()) / steps, 2)
            rep["exposed_parts_us_per_step"] = {n: round(t / steps, 2) for n, (t, _) in stalls.items()}
            rep["exposed_note"] = ("time the training stream sat idle at its waits for the exchange (device clock stamps around each wait; "
                                   "the stamp kernels' own launch boundaries, ~2 us per wait, are included)")
        return rep

    def _mb_slices(self):
        Bm = self.B // self.mb
        return [slice(m * Bm, (m + 1) * Bm) for m in range(self.mb)]

    def _route_batch(self, keys, wait_current):
        return [self.ex.route(keys[sl], hashed=True, wait_current=wait_current) for sl in self._mb_slices()]

    @staticmethod
    def _keys_tag(keys):
        # identity of a key batch: buffer, shape AND torch's in-place version counter -- a loader that refills the same device
        # buffer between steps bumps the version, so a route prefetched for the old contents is not reused
        return (keys.data_ptr(), tuple(keys.shape), keys._version)

    def prefetch_route_begin(self, next_keys, ready_event=None):
        """Start routing the NEXT batch (hash, bucket, split-size exchange) on the routing stream; nothing blocks.
        `next_keys` must already be complete, or `ready_event` (recorded by the producer after writing it) must be given:
        the routing stream is deliberately not ordered against the training stream."""
        if ready_event is not None and self._cuda:
            self.ex._rs.wait_event(ready_event)
        self._route = ([self.ex.route_begin(next_keys[sl], hashed=True, wait_current=False) for sl in self._mb_slices()],
                       self._keys_tag(next_keys))

    def prefetch_route_finish(self):
        """Finish it (id exchange, owner-side sort).  The split sizes were exchanged long ago: the host wait is nominal."""
        if self._route is not None:
            for r in self._route[0]:
                self.ex.route_finish(r)

    def refresh_planes(self):
        """Re-split the wide layers' weights into their bf16 planes (see ops.WeightPlanes.ensure_fresh)."""
        for wp in self.wplanes:
            if wp is not None:
                wp.refresh()

    @staticmethod
    def _amax_of(t):
        """max |t| as float bits in an int32 [1] tensor (what dr_h2_amax leaves; torch ops: the DR_SH_TRACK_AMAX path and tests)"""
        if t.numel() == 0:
            return torch.zeros(1, dtype=torch.int32, device=t.device)
        return t.detach().abs().max().to(torch.float32).reshape(1).view(torch.int32).clone()

    def _publish_table_bound(self):
        """Behind the step's last owner-side update: the bound for the rows ANY rank may be sent next step = max over the shards' records
        (one 4-byte all-reduce, in order behind that update and in front of the next step's first row fetch), into the buffer the next
        step will read."""
        if not self.track_amax or self.ex.local:
            return
        nxt = self._tab_bufs[self._tab_i ^ 1]
        nxt.copy_(self.tab_amax_local)
        self.tr.allreduce(nxt, op=dist.ReduceOp.MAX)
        self._tab_swap = True

    def _x_record(self):
        """f16x2 mode: the record of concat = [rows this rank received, dense features] (float bits of non-negative values order like
        integers, so the larger record is the integer maximum)"""
        if self.dense_amax is not None:
            torch.maximum(self.tab_amax, self.dense_amax, out=self.x_amax_all)
        else:
            self.x_amax_all.copy_(self.tab_amax)

    def train_step(self, keys, dense, labels, next_keys=None, next_keys_ready=None):
        """One step.  The rank's batch is processed as `self.mb` micro-batches that share one set of weights (all
        forwards read pre-update tables; the dense gradients are summed; the loss is the mean over world * B examples):
        micro-batch m+1's row exchange is in flight while micro-batch m runs its tower, and micro-batch m's gradient
        exchange + owner-side update run under micro-batch m+1's tower and the first-layer wgrad."""
        p, F, D, W, B, M = self.p, self.F, self.D, self.world, self.B, self.mb
        Bm = B // M
        cuda = self._cuda
        if self._events is not None:
            self._ev_live = (self._ev_step % self._ev_every) == 0
            self._ev_step += 1
        for wp in self.wplanes:          # weights written from outside since the last refresh (load / copy_ / broadcast): re-split
            if wp is not None:
                wp.ensure_fresh()
        h2 = self.h2
        if self.track_amax and self._tab_swap:
            self._tab_i ^= 1
            self.tab_amax = self._tab_bufs[self._tab_i]      # filled behind the previous step's last owner-side update
            self._tab_swap = False
        if h2 and (self.table.data_ptr(), self.table._version) != self._tab_ver:
            # the shard was written from outside (checkpoint restore, a test's copy_): rebuild the record (and the global bound).  The
            # rebuild of the global bound is a collective: an outside write must happen on every rank before the same step (a restore
            # does; a write to ONE rank's shard alone would leave the others waiting here)
            ops.h2_amax(self.table, self.tab_amax_local)
            if not self.ex.local:
                if cuda:
                    self.comm.wait_stream(torch.cuda.current_stream())
                with (torch.cuda.stream(self.comm) if cuda else _NullCtx()):
                    self.tab_amax.copy_(self.tab_amax_local)
                    self.tr.allreduce(self.tab_amax, op=dist.ReduceOp.MAX)
            self._tab_ver = (self.table.data_ptr(), self.table._version)
        adam = self.optimizer == "adam"
        # SGD: gradients of the RANK-mean loss travel, the 1 / W of the global mean is folded into the step (-lr / W).
        # Adam: the update is not linear in the gradient, so the head normalises by the global batch and every gradient that
        # travels is already that of the global-mean loss.
        n_total = B * W if adam else B
        adam_args = None
        if adam:
            self.t += 1
            lr_t = ops.adam_lr_t(self.lr, self.beta1, self.beta2, self.t)
            adam_args = (lr_t, self.beta1, self.beta2, self.eps, self.m_table, self.v_table, self.m_lin, self.v_lin)
        if cuda and len(self._done) >= 2:                 # bound the host's run-ahead to two steps
            self._done.pop(0).synchronize()
        # steps 1-3: use the prefetched routes if they were built for exactly these keys, else route now
        routes = None
        if self._route is not None and self._route[1] == self._keys_tag(keys):
            self.prefetch_route_finish()          # no-op when the previous step already finished it
            routes = self._route[0]
        self._route = None
        if routes is None:
            routes = self._k("emb_route(hash+bucket+a2a ids+sort)", "xgmi", B * F * 24, lambda: self._route_batch(keys, True))
        main = torch.cuda.current_stream() if cuda else None
        comm_ctx = (lambda: torch.cuda.stream(self.comm)) if cuda else _NullCtx
        # ---- steps 4-5 for every micro-batch, back to back on the communication stream ---------------------------
        got, ev_rows = [], []
        if cuda and not self._comm_synced:
            # once: the tables were initialised on the construction stream.  Afterwards the gathers depend only on the
            # owner-side updates, which run on the communication stream itself (in order) -- NOT on the training stream, so
            # the next step's first row fetch may start while this step's wgrad / all-reduce are still running.
            self.comm.wait_stream(main)
            self._comm_synced = True
        with comm_ctx():
            for m in range(M):
                got.append(self._k("emb_fetch(gather+a2a rows)", "xgmi", Bm * F * (4 * D + 4),
                                   lambda m=m: self.ex.fetch(routes[m], self.table, self.lin_w)))
                if cuda:
                    ev = torch.cuda.Event()
                    ev.record()
                    ev_rows.append(ev)
        if next_keys is not None:
            # split sizes of the NEXT batch: queued right behind this step's row fetches, long before they are needed
            self.prefetch_route_begin(next_keys, next_keys_ready)
        self.flat_grads.zero_()
        if self.Nd:
            self.concat[:, F * D:F * D + self.Nd].copy_(dense)
            if self.dense_pad is not None:
                self.dense_pad[:, :self.Nd].copy_(dense)               # the fused kernel's own (k-tile wide) copy
                if h2:
                    ops.h2_amax(self.dense_pad, self.dense_amax)
        n_layers = len(self.Ws)
        x_in = self.concat[:, :self.in_dim]
        dys = [None] * n_layers
        dw_todo = []
        slices = self._mb_slices()
        prefetched = next_keys is None
        ev_last_apply = None
        fused_l0 = self.fuse_k3 and not (self.fuse_head and n_layers - 2 == 0)

        def fused_first_layer(m, sl, stream):
            rows_m, lin_m = got[m]
            for t in (routes[m].pos, rows_m, lin_m):
                if t is not None:
                    t.record_stream(stream)
            Wt, b = self.Ws[0], self.bs[0]
            if h2:
                self._k("emb_linear_fwd_L0", "mfma", 2.0 * Bm * Wt.shape[0] * Wt.shape[1],
                        lambda: ops.h2_emb_linear_fwd(
                            routes[m].pos, self.ex._zero_base, rows_m.shape[0], rows_m, self.tab_amax, lin_m, self.lin_bias,
                            self.dense_pad[sl] if self.dense_pad is not None else None, self.dense_amax, self.concat[sl], self.in_dim,
                            self.wplanes[0].wt, b, self.acts[0], self.sum_x[sl], self.fm_logit[sl], self.hs[0][sl]))
                return
            self._k("emb_linear_fwd_L0", "mfma", 2.0 * Bm * Wt.shape[0] * Wt.shape[1],
                    lambda: ops.bf3_emb_linear_fwd(
                        routes[m].pos, self.ex._zero_base, rows_m.shape[0], rows_m, lin_m, self.lin_bias,
                        self.dense_pad[sl] if self.dense_pad is not None else None, self.concat[sl], self.in_dim,
                        self.wplanes[0].wt, b, self.acts[0], self.sum_x[sl], self.fm_logit[sl], self.hs[0][sl]))

        # The fused first layer of a micro-batch is 256-row tiles x ONE column tile: B / M / 256 blocks, i.e. half of the 256 CUs at
        # M = 2 -- and a persistent block owns its CU (160 KB of LDS), so each launch takes as long as the full batch's.  The later
        # micro-batches' first layers therefore run on their own streams, each as soon as ITS rows have arrived: they fill the CUs
        # the first one leaves idle (world-1, M = 2: 2 x 380 us back to back -> both done ~120 us after the first would be alone),
        # and with real peers the second forward overlaps the first micro-batch's tower tail instead of queueing behind it.
        ev_fwd = [None] * M
        if fused_l0 and cuda and M > 1 and self.fwd_streams:
            ev_in = torch.cuda.Event()
            ev_in.record()                       # dense features placed, last step's weight / plane updates behind us
            for m in range(1, M):
                st = self.fwd_streams[(m - 1) % len(self.fwd_streams)]
                with torch.cuda.stream(st):
                    st.wait_event(ev_in)
                    st.wait_event(ev_rows[m])
                    fused_first_layer(m, slices[m], st)
                    ev_fwd[m] = torch.cuda.Event()
                    ev_fwd[m].record()
        for m, sl in enumerate(slices):
            # ---- step 6 + tower forward + loss ----------------------------------------------------------------------
            if cuda:
                self._stall("stall_rows(training stream waits for C2)", main, ev_rows[m])
                if ev_fwd[m] is not None:
                    main.wait_event(ev_fwd[m])
            if not fused_l0:
                self._k("emb_pool_fwd", "hbm", Bm * (8 * F * D + 12 * F + 8),
                        lambda m=m, sl=sl: self.ex.pool(routes[m], got[m][0], got[m][1], self.lin_bias, self.ld, concat=self.concat[sl],
                                                       sum_x=self.sum_x[sl], fm_logit=self.fm_logit[sl]))
            x = x_in[sl]
            for i, (Wt, b) in enumerate(zip(self.Ws, self.bs)):
                if self.fuse_head and i == n_layers - 2:
                    break
                if i == 0 and fused_l0:
                    if ev_fwd[m] is None:
                        fused_first_layer(m, sl, main)
                elif self.wplanes[i] is not None and Bm >= self.planes_fwd_rows:
                    self._k("linear_fwd_L%d" % i, "mfma", 2.0 * Bm * Wt.shape[0] * Wt.shape[1],
                            lambda x=x, b=b, i=i, sl=sl: ops.bf3_linear_nt(x, self.wplanes[i].wt, bias=b, act=self.acts[i],
                                                                           out=self.hs[i][sl]))
                else:
                    self._k("linear_fwd_L%d" % i, "mfma", 2.0 * Bm * Wt.shape[0] * Wt.shape[1],
                            lambda x=x, Wt=Wt, b=b, i=i, sl=sl: p.linear_fwd(x, Wt, b, self.acts[i], out=self.hs[i][sl]))
                x = self.hs[i][sl]
            tail_done = False
            if self.fuse_tail:
                # head + the backward of the last hidden layer in one pass: prob, loss part, d_logit, d_h, the gradient for the layer
                # below (with its amax record in the f16x2 mode) and both layers' gradients into the bucket
                nlt = n_layers
                self._k("tower_tail_fused", "hbm", 4.0 * Bm * (2 * self.Ws[-2].shape[0] + self.Ws[-2].shape[1] + 4),
                        lambda x=x, sl=sl, m=m: p.tower_tail_fused(
                            x, self.Ws[-2], self.bs[-2], self.Ws[-1], self.bs[-1], self.fm_logit[sl], labels[sl], ops.LOSS_SIGMOID_CE, 1.0,
                            self.dhs[nlt - 3][sl], dst_W1=self.gWs[-2], dst_b1=self.gbs[-2], dst_W2=self.gWs[-1], dst_b2=self.gbs[-1],
                            prob=self.prob[sl], d_logit=self.d_logit[sl], d_h=self.dhs[-1][sl], loss=self.loss_parts[m:m + 1],
                            workspace=self.tail_ws, n_total=n_total, dx_amax=self.dh0_amax[m] if (h2 and nlt - 2 == 1) else None))
                tail_done = True
            elif self.fuse_head:
                self._k("tower_head_fwd_bwd", "hbm", 4.0 * Bm * (self.Ws[-2].shape[0] + self.Ws[-2].shape[1] + 4),
                        lambda x=x, sl=sl, m=m: p.tower_head_fwd_bwd(
                            x, self.Ws[-2], self.bs[-2], self.Ws[-1], self.bs[-1], self.fm_logit[sl], labels[sl],
                            ops.LOSS_SIGMOID_CE, 1.0, act=1, prob=self.prob[sl], d_logit=self.d_logit[sl], d_h=self.dhs[-1][sl],
                            loss=self.loss_parts[m:m + 1], workspace=self.head_ws, dst_W2=self.gWs[-1], dst_b2=self.gbs[-1],
                            n_total=n_total))
            else:                                          # M == 1 here (see __init__)
                p.bce_fwd_bwd(self.fm_logit, labels, ops.LOSS_SIGMOID_CE, workspace=self.ws, logits_b=self.hs[-1],
                              out=(self.prob, self.d_logit, self.loss_parts[0:1]))
                if adam and W > 1:
                    # this loss kernel normalises by the RANK batch; Adam needs every travelling gradient to be that of the
                    # global-mean loss (what the fused head does through n_total): d_logit *= 1 / W.  loss_parts stays the
                    # rank mean, which is what train_step reports.
                    p.axpy(1.0 / W - 1.0, self.d_logit, self.d_logit)
            # ---- dgrad chain (produces d_concat, the input of the embedding exchange) ------------------------------
            dy = self.d_logit[sl].reshape(-1, 1)
            top = n_layers - 1
            if self.fuse_head:
                top = n_layers - 2
                dy = self.dhs[-1][sl]
            if tail_done:
                top = n_layers - 3
                dy = self.dhs[n_layers - 3][sl]
            fuse_pack = (self.fuse_pack and self.wplanes[0] is not None and not self.narrow[0] and top >= 0
                         and (not h2 or (Bm % 256 == 0 and Bm * F < (1 << 25))))
            for i in range(top, -1, -1):
                if i > 0:
                    dx = self.dhs[i - 1][sl]
                    rs = self.hs[i - 1][sl] if self.acts[i - 1] else None
                else:
                    dx = self.d_concat[sl, :self.in_dim]
                    rs = None
                if i == 0 and fuse_pack:
                    # the gradient of every slot straight into the send layout (dgrad epilogue = dr_emb_pack_grads)
                    g_rows, g_lin = self.ex.pack_buffers(routes[m])
                    if h2:
                        if not (n_layers > 1 and self.narrow[1]):
                            ops.h2_amax(dy, self.dh0_amax[m])        # (layer 1's backward was not the narrow kernel that leaves the record)
                        self._k("linear_bwd_dx_L0", "mfma", 2.0 * Bm * self.Ws[0].shape[0] * self.Ws[0].shape[1],
                                lambda dy=dy, sl=sl, m=m, g_rows=g_rows, g_lin=g_lin: ops.h2_linear_nt_pack(
                                    dy, self.dh0_amax[m], self.wplanes[0].w, routes[m].pos, self.d_logit[sl], g_rows, g_lin, self.g_lin_bias,
                                    sum_x=self.sum_x[sl], x=self.concat[sl]))
                    else:
                        self._k("linear_bwd_dx_L0", "mfma", 2.0 * Bm * self.Ws[0].shape[0] * self.Ws[0].shape[1],
                                lambda dy=dy, sl=sl, m=m, g_rows=g_rows, g_lin=g_lin: ops.bf3_linear_nt_pack(
                                    dy, self.wplanes[0].w, routes[m].pos, self.d_logit[sl], g_rows, g_lin, self.g_lin_bias,
                                    sum_x=self.sum_x[sl], x=self.concat[sl]))
                    if m == 0:
                        dw_todo.append(i)
                    continue
                if self.narrow[i]:
                    self._k("linear_bwd_narrow_L%d" % i, "hbm", 4.0 * Bm * (2 * self.Ws[i].shape[0] + self.Ws[i].shape[1]),
                            lambda dy=dy, i=i, dx=dx, sl=sl, m=m: p.linear_bwd_narrow(self.hs[i - 1][sl], dy, self.Ws[i], 1.0, self.gWs[i],
                                                                                     self.gbs[i], dx, relu_mask=True,
                                                                                     workspace=self.narrow_ws[i],
                                                                                     **({"dx_amax": self.dh0_amax[m]} if (h2 and i == 1) else {})))
                else:
                    if i == 0 and h2:
                        if not (n_layers > 1 and self.narrow[1]):
                            ops.h2_amax(dy, self.dh0_amax[m])        # (layer 1's backward was not the narrow kernel that leaves the record)
                        self._k("linear_bwd_dx_L%d" % i, "mfma", 2.0 * Bm * self.Ws[i].shape[0] * self.Ws[i].shape[1],
                                lambda dy=dy, i=i, rs=rs, dx=dx, m=m: ops.h2_linear_nt(dy, self.dh0_amax[m], self.wplanes[i].w, mask=rs, out=dx))
                    elif self.wplanes[i] is not None:
                        self._k("linear_bwd_dx_L%d" % i, "mfma", 2.0 * Bm * self.Ws[i].shape[0] * self.Ws[i].shape[1],
                                lambda dy=dy, i=i, rs=rs, dx=dx: ops.bf3_linear_nt(dy, self.wplanes[i].w, mask=rs, out=dx))
                    else:
                        self._k("linear_bwd_dx_L%d" % i, "mfma", 2.0 * Bm * self.Ws[i].shape[0] * self.Ws[i].shape[1],
                                lambda dy=dy, i=i, rs=rs, dx=dx: p.linear_bwd_dx(dy, self.Ws[i], relu_src=rs, out=dx))
                    if m == 0:
                        dw_todo.append(i)
                dy = dx
            # ---- step 8: pack (local), then C3 + owner-side sorted update on the communication stream ------------------
            if not fuse_pack:
                g_rows, g_lin = self._k("emb_pack_grads", "hbm", Bm * F * (12 * D + 4),
                                        lambda sl=sl, m=m: self.ex.pack(routes[m], self.d_concat[sl], self.d_logit[sl], self.concat[sl],
                                                                        self.sum_x[sl], self.g_lin_bias))
            if cuda:
                ev_p = torch.cuda.Event()
                ev_p.record()
                if self.wgrad_split:
                    # this micro-batch's share of the wide layers' wgrads right here (dW accumulates over the micro-batches): the
                    # exchange + owner-side K4 of micro-batch m then run beside a HALF-length wgrad, and the last K4 -- which the
                    # next step's first row fetch waits for -- beside wgrad(M - 1) only, not beside the whole batch's
                    for i in (dw_todo if m > 0 else list(dw_todo)):
                        xin = x_in[sl] if i == 0 else self.hs[i - 1][sl]
                        dyi = self.dhs[i][sl] if i < n_layers - 1 else self.d_logit[sl].reshape(-1, 1)
                        if i == 0 and h2:
                            self._x_record()
                            self._k("linear_bwd_dw_L%d" % i, "mfma", 2.0 * Bm * self.Ws[i].shape[0] * self.Ws[i].shape[1],
                                    lambda xin=xin, dyi=dyi, i=i, m=m: ops.h2_wgrad(xin, self.x_amax_all, dyi, self.dh0_amax[m], 1.0, self.gWs[i],
                                                                                    self.gbs[i], workspace=self.wg_ws[i]))
                        elif self.wg_ws[i] is not None:
                            self._k("linear_bwd_dw_L%d" % i, "mfma", 2.0 * Bm * self.Ws[i].shape[0] * self.Ws[i].shape[1],
                                    lambda xin=xin, dyi=dyi, i=i: ops.bf3_wgrad(xin, dyi, 1.0, self.gWs[i], self.gbs[i], workspace=self.wg_ws[i]))
                        else:
                            self._k("linear_bwd_dw_L%d" % i, "mfma", 2.0 * Bm * self.Ws[i].shape[0] * self.Ws[i].shape[1],
                                    lambda xin=xin, dyi=dyi, i=i: p.linear_bwd_dw(xin, dyi, 1.0, self.gWs[i], self.gbs[i], workspace=self.dw_ws[i]))
                with torch.cuda.stream(self.comm):
                    self.comm.wait_event(ev_p)
                    self._k("emb_grads(a2a+sorted K4)", "xgmi", Bm * F * (4 * D + 4),
                            lambda m=m, g_rows=g_rows, g_lin=g_lin: self.ex.exchange_and_apply(routes[m], g_rows, g_lin, -self.lr / W,
                                                                                               self.table, self.lin_w, adam=adam_args,
                                                                                               table_amax=self.tab_amax_local if h2 else None))
                    if self.track_amax and not h2:
                        torch.maximum(self.tab_amax_local, self._amax_of(self.table), out=self.tab_amax_local)
                    if m == M - 1:
                        self._publish_table_bound()
                        ev_last_apply = torch.cuda.Event()
                        ev_last_apply.record()
            else:
                self.ex.exchange_and_apply(routes[m], g_rows, g_lin, -self.lr / W, self.table, self.lin_w, adam=adam_args)
                if self.track_amax:
                    torch.maximum(self.tab_amax_local, self._amax_of(self.table), out=self.tab_amax_local)
                    if m == M - 1:
                        self._publish_table_bound()
        # ---- wgrads over the WHOLE rank batch (activations of all micro-batches are contiguous) --------------------------------
        # DR_SH_K4_FIRST=1 (measured, not adopted): the wgrad WAITS for the last owner-side update.  A persistent GEMM block owns its
        # CU, so the last K4 crawls beside the wgrad (469 us instead of 170, rocprofv3 timeline of the world-1 step) and the next
        # step's row fetches -- which must see that update -- queue behind it: K4 -> gather -> forward is a serial 180 us hole on
        # the training stream.  With K4 first the NEXT step's fetches run beside the wgrad and both forwards find their rows
        # waiting (322 us for the pair instead of 436) -- but the wgrad stretches by as much beside the gathers (309 -> 413 us) and
        # the last K4 is fully exposed: 1.97 ms against 1.94 (same call, alternating, three rounds).
        if cuda and self.k4_first and ev_last_apply is not None:
            self._stall("stall_apply(training stream waits for C3 + owner-side update)", main, ev_last_apply)
        dys[0] = self.dhs[0] if n_layers > 1 else self.d_logit.reshape(-1, 1)
        for i in ([] if (cuda and self.wgrad_split) else dw_todo):
            xin = x_in if i == 0 else self.hs[i - 1]
            dyi = self.dhs[i] if i < n_layers - 1 else self.d_logit.reshape(-1, 1)
            if i == 0 and h2:
                # records of the whole rank batch: x = [received rows, dense features], d h0 = the micro-batches' pieces
                self._x_record()
                # (float bits of non-negative values order like integers: the batch's record is the integer maximum of the micro-batches';
                # round 6: M - 1 one-element launches instead of cat + a reduce kernel)
                if len(self.dh0_amax) == 1:
                    self.dh0_amax_all.copy_(self.dh0_amax[0])
                else:
                    torch.maximum(self.dh0_amax[0], self.dh0_amax[1], out=self.dh0_amax_all)
                    for r in self.dh0_amax[2:]:
                        torch.maximum(self.dh0_amax_all, r, out=self.dh0_amax_all)
                self._k("linear_bwd_dw_L%d" % i, "mfma", 2.0 * B * self.Ws[i].shape[0] * self.Ws[i].shape[1],
                        lambda xin=xin, dyi=dyi, i=i: ops.h2_wgrad(xin, self.x_amax_all, dyi, self.dh0_amax_all, 1.0, self.gWs[i], self.gbs[i],
                                                                   workspace=self.wg_ws[i]))
            elif self.wg_ws[i] is not None:
                self._k("linear_bwd_dw_L%d" % i, "mfma", 2.0 * B * self.Ws[i].shape[0] * self.Ws[i].shape[1],
                        lambda xin=xin, dyi=dyi, i=i: ops.bf3_wgrad(xin, dyi, 1.0, self.gWs[i], self.gbs[i], workspace=self.wg_ws[i]))
            else:
                self._k("linear_bwd_dw_L%d" % i, "mfma", 2.0 * B * self.Ws[i].shape[0] * self.Ws[i].shape[1],
                        lambda xin=xin, dyi=dyi, i=i: p.linear_bwd_dw(xin, dyi, 1.0, self.gWs[i], self.gbs[i], workspace=self.dw_ws[i]))
        if not prefetched:
            # Second half of the next batch's routing (id exchange + owner-side plan: ~50 launches for two micro-batches).  Issued
            # HERE, after the last gradient exchange / K4 and the wgrads have been handed to the GPU: in front of them (round 2)
            # the host spent ~250 us on these launches while the communication stream sat idle (rocprofv3 on the single-GPU
            # engine showed the same pattern: a 233 us gap in front of K4).  The split sizes it needs arrived long ago.
            self.prefetch_route_finish()
            prefetched = True
        # ---- dense tower: one all-reduce, then w += -(lr / W) * sum_r g_r  (mean over the global batch) ------
        if not self.ex.local:
            self._k("allreduce_dense_grads", "xgmi", self.flat_grads.numel() * 4, lambda: self.tr.allreduce(self.flat_grads))
        if adam:
            p.adam_step(self.flat_params, self.flat_grads, self.flat_m, self.flat_v, adam_args[0], self.beta1, self.beta2, self.eps)
        else:
            p.axpy(-self.lr / W, self.flat_grads, self.flat_params)
        for wp in self.wplanes:
            if wp is not None:
                wp.refresh()                               # the weights just moved: their planes follow
        torch.sum(self.loss_parts, dim=0, keepdim=True, out=self.loss)
        if adam and self.fuse_head:
            self.loss.mul_(W)          # loss_parts were normalised by the global batch: report the rank's mean like the SGD mode
        if cuda:
            if not self.k4_first:
                self._stall("stall_apply(training stream waits for C3 + owner-side update)", main, ev_last_apply)   # the step ends when every owner has applied its updates
            ev = torch.cuda.Event()
            ev.record()
            self._done.append(ev)
        return self.loss


class ShardedDCNEngine:
    """DCN (BASELINE config 4: 3 full-rank cross layers + MLP + Dense(1)) with row-sharded tables and data-parallel batch; same
    model / loss / SGD as dcn_engine.DCNEngine.  The cross / MLP weights are replicated: their gradients live in ONE flat bucket
    (3 x (1677^2 + 1677) + 2.37 M floats = 43 MB at config 4) that is all-reduced asynchronously while the
    embedding gradients travel back to their owners.  `batch` is the per-rank batch; the loss is the mean over world * batch."""

    def __init__(self, num_fields, vocab_per_field, dim, num_cross, dnn_units: Sequence[int], batch, num_dense=0, lr=0.01,
                 diag_scale=0.0, device="cuda", world=None, rank=None, seed=42, prims=None, group=None, init_tables=None,
                 transport=None, alias_world1=True):
        self.world = world if world is not None else dist.get_world_size(group)
        self.rank = rank if rank is not None else dist.get_rank(group)
        self.F, self.V, self.D, self.B, self.Nd, self.lr, self.diag = num_fields, vocab_per_field, dim, batch, num_dense, lr, diag_scale
        self.group, self.dev = group, device
        self.p = prims if prims is not None else HipPrims
        F, V, D, B, W = num_fields, vocab_per_field, dim, batch, self.world
        self.ex = ShardedEmbeddingExchange(F, V, D, W, self.rank, device, self.p, group, transport=transport,
                                           alias_world1=alias_world1)
        self.tr = self.ex.tr
        rps = self.ex.rows_per_shard
        g = torch.Generator(device=device)
        g.manual_seed(seed)                       # replicated weights: same seed on every rank
        f32 = dict(dtype=torch.float32, device=device)
        self.in_dim = F * D + num_dense
        self.ld = _pad4(self.in_dim)
        units = list(dnn_units) + [1]
        n_in, ld = self.in_dim, self.ld
        total = num_cross * (n_in * ld + ld)
        d = n_in
        for u in units:
            total += d * _pad4(u) + _pad4(u)
            d = u
        self.flat_params = torch.zeros(total + 4, **f32)
        self.flat_grads = torch.zeros(total + 4, **f32)
        off = 0
        self.cross_W, self.cross_b, self.g_cross_W, self.g_cross_b = [], [], [], []
        for _ in range(num_cross):
            Wc = self.flat_params[off:off + n_in * ld].view(n_in, ld)[:, :n_in]
            Wc.copy_(torch.empty((n_in, n_in), **f32).normal_(0.0, 0.01, generator=g).clamp_(-0.02, 0.02))
            self.cross_W.append(Wc)
            self.g_cross_W.append(self.flat_grads[off:off + n_in * ld].view(n_in, ld)[:, :n_in])
            off += n_in * ld
            self.cross_b.append(self.flat_params[off:off + n_in])
            self.g_cross_b.append(self.flat_grads[off:off + n_in])
            off += ld
        self.Ws, self.bs, self.gWs, self.gbs = [], [], [], []
        d = n_in
        for u in units:
            pu = _pad4(u)
            Wm = self.flat_params[off:off + d * pu].view(d, pu)[:, :u]
            Wm.copy_((torch.rand((d, u), device=device, generator=g) * 2 - 1) * math.sqrt(6.0 / (d + u)))
            self.Ws.append(Wm)
            self.gWs.append(self.flat_grads[off:off + d * pu].view(d, pu)[:, :u])
            off += d * pu
            self.bs.append(self.flat_params[off:off + u])
            self.gbs.append(self.flat_grads[off:off + u])
            off += pu
            d = u
        self.acts = [1] * len(dnn_units) + [0]
        self.table = torch.empty((F * rps, D), **f32)
        if init_tables is not None:
            full_table = init_tables
            for f in range(F):
                ids = torch.arange(self.rank, V, W)
                self.table[f * rps:f * rps + len(ids)] = full_table[f * V + ids].to(device)
        else:
            gt = torch.Generator(device=device)
            gt.manual_seed(seed * 1000 + 17 + self.rank)
            std = 1.0 / math.sqrt(D)
            for r0 in range(0, F * rps, 1 << 24):
                self.table[r0:r0 + (1 << 24)].normal_(0.0, std, generator=gt).clamp_(-2 * std, 2 * std)
        self.x0 = torch.zeros((B, ld), **f32)
        self.zero_logit = torch.zeros(B, **f32)
        self.prob, self.d_logit, self.loss = torch.empty(B, **f32), torch.empty(B, **f32), torch.zeros(1, **f32)
        self.ws = torch.empty(1024, **f32)
        self._cuda = torch.device(device).type == "cuda"
        self.mb = 1
        # The dense half -- cross stack, MLP, loss and their backward -- is dcn_engine.DCNDense, the SAME object the single-GPU DCNEngine
        # runs (round 5): wide GEMMs on pre-split weights in the operand split the library reports ("f16x2" by default; x0's amax record
        # comes from a pass over the rows this rank received, so no record has to travel with the exchange), gradients accumulated
        # into the all-reduce bucket instead of applied in place.  The generic in-kernel-split path below remains for primitives other
        # than HipPrims (the oracle-backed ones of the gloo tests) and DR_SH_DCN_CORE=0.
        import os as _os
        self.core = None
        if self._cuda and self.p is HipPrims and _os.environ.get("DR_SH_DCN_CORE", "1") == "1":
            from .dcn_engine import DCNDense
            self.core = DCNDense(self.cross_W, self.cross_b, self.Ws, self.bs, B, self.in_dim, ld, diag_scale, device,
                                 grads=(self.g_cross_W, self.g_cross_b, self.gWs, self.gbs))
            self.loss = self.core.loss

    h2 = property(lambda self: bool(self.core is not None and self.core.h2))
    h2_all_wide = property(lambda self: bool(self.core is not None and self.core.h2_all_wide))

    def enable_kernel_events(self, on):           # bench.py contract; the sharded DCN step reports no per-kernel rows
        pass

    def kernel_event_summary(self):
        return {}

    def train_step(self, keys, dense, labels, next_keys=None):
        p, F, D, B, W, lr, n_in, ld = self.p, self.F, self.D, self.B, self.world, self.lr, self.in_dim, self.ld
        f32 = dict(dtype=torch.float32, device=self.x0.device)
        # ---- embeddings: route, owner-side gather, rows back, K3 over the received rows -------------------------------------------
        route = self.ex.route(keys, hashed=True)
        got_rows, _ = self.ex.fetch(route, self.table, None)
        p.emb_pool_fwd(route.pos, F, None if F <= 64 else self.ex._col_start, self.ex._zero_base, got_rows, None, None,
                       ld_concat=ld, concat=self.x0, want_sum_x=False, want_fm=False)
        if self.Nd:
            self.x0[:, F * D:F * D + self.Nd].copy_(dense)
        x0 = self.x0[:, :n_in]
        if self.core is not None:
            # cross stack + MLP + loss + backward on the shared dense core; every weight gradient goes into the bucket
            self.core.ensure_fresh()
            self.flat_grads.zero_()
            d_x0 = self.core.step(self.x0, labels, lr)
            return self._finish_step(route, d_x0)
        # ---- forward: cross stack, MLP, loss (mean over the rank's batch; the 1 / W of the global mean is folded into the step) --
        xs, prods = [x0], []
        for Wc, bc in zip(self.cross_W, self.cross_b):
            out, prod = p.cross_fwd(x0, xs[-1], Wc, bc, self.diag, want_prod=True)
            xs.append(out)
            prods.append(prod)
        hs, x = [], xs[-1]
        for i, (Wm, bm) in enumerate(zip(self.Ws, self.bs)):
            h = torch.empty((B, _pad4(Wm.shape[1])), **f32)[:, :Wm.shape[1]]
            x = p.linear_fwd(x, Wm, bm, self.acts[i], out=h)
            hs.append(x)
        p.bce_fwd_bwd(self.zero_logit, labels, ops.LOSS_SIGMOID_CE, workspace=self.ws, logits_b=hs[-1],
                      out=(self.prob, self.d_logit, self.loss))
        # ---- backward: MLP, then the cross stack; every weight gradient goes into the bucket --------------------------------------
        self.flat_grads.zero_()
        dy = self.d_logit.reshape(-1, 1)
        d_top = torch.zeros((B, ld), **f32)[:, :n_in]
        for i in range(len(self.Ws) - 1, -1, -1):
            xin = xs[-1] if i == 0 else hs[i - 1]
            if i > 0:
                dx = torch.empty((B, _pad4(xin.shape[1])), **f32)[:, :xin.shape[1]]
                rs = hs[i - 1] if self.acts[i - 1] else None
            else:
                dx, rs = d_top, None
            p.linear_bwd_dx(dy, self.Ws[i], relu_src=rs, out=dx)
            p.linear_bwd_dw(xin, dy, 1.0, self.gWs[i], self.gbs[i])
            dy = dx
        d_out = d_top
        d_x0 = torch.zeros((B, ld), **f32)[:, :n_in]
        for l in range(len(self.cross_W) - 1, -1, -1):
            if self.diag == 0.0:         # the dgrad accumulates into the d_out buffer itself (see dcn_engine.DCNEngine)
                d_x = d_out
                d_prod = p.cross_combine_bwd(x0, prods[l], d_out, 0.0, d_x0, None)
            else:
                d_x = torch.zeros((B, ld), **f32)[:, :n_in]
                d_prod = p.cross_combine_bwd(x0, prods[l], d_out, self.diag, d_x0, d_x)
            p.linear_bwd_dx(d_prod, self.cross_W[l], None, accumulate=True, out=d_x)
            p.linear_bwd_dw(xs[l], d_prod, 1.0, self.g_cross_W[l], self.g_cross_b[l])
            d_out = d_x
        d_x0.add_(d_out)                                  # the first layer's x IS x0
        return self._finish_step(route, d_x0)

    def _finish_step(self, route, d_x0):
        p, W, lr = self.p, self.world, self.lr
        # ---- C4: the replicated weights' gradients (asynchronous: the embedding gradients travel meanwhile) ------------------------
        work = self.tr.allreduce(self.flat_grads, async_op=True) if not self.ex.local else None
        # ---- C3: embedding-row gradients to their owners + sorted scatter with the SGD step --------------------------------------------
        g_rows, _ = self.ex.pack(route, d_x0, None, None, None)
        self.ex.exchange_and_apply(route, g_rows, None, -lr / W, self.table, None)
        if work is not None:
            work.wait()
        p.axpy(-lr / W, self.flat_grads, self.flat_params)
        if self.core is not None:
            self.core.refresh_planes()                   # the weights just moved: their planes follow
        return self.loss
