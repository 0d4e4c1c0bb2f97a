"""Two-tower retrieval over N ranks (BASELINE.json config 5: "item-tower sharded 8 x MI355X"): one process per GPU,
torch.distributed (backend "nccl" == RCCL over xGMI).  New design -- the reference is single-process.

Partitioning: the batch is data-parallel (B query / positive-item pairs per rank); the user and the item table are row-sharded
`id % N` like the DeepFM tables (ShardedEmbeddingExchange with one field each); the two towers are replicated (one flat
gradient bucket, one all-reduce); the corpus of item-tower outputs used by the FactorizedTopK metric is sharded with the item
table (rank r holds the items r, r + N, ...).

Training step on every rank (`Retrieval.call`, keras/models/retrieval/sbcnm.py:120-163 of the reference, over the GLOBAL batch):
  1. route + fetch the rank's user rows and item rows (id all-to-all, row all-to-all)            C1, C2
  2. towers -> q [B, Do], c [B, Do]                                                              K7
  3. all-gather c and the item ids -> every query sees the N * B in-batch candidates of the step  C5 (SURVEY: 4 MB at c5)
  4. scores [B, N B] = q c_all^T; labels = one-hot at the query's own column; accidental hits masked by item id; / temperature;
     CCE(from_logits, SUM)  -- the loss of the global batch is the sum of the ranks' losses        K9 (explicit-matrix kernels)
  5. G = dLoss/dScores; dq = G c_all (local), dc_all = G^T q summed over ranks (all-reduce; each rank keeps its B rows)
  6. tower backward; one all-reduce of the tower gradients; w += -lr * grad (the loss is a SUM: no 1 / N)
  7. embedding-row gradients back to their owners (all-to-all) + sorted scatter with the SGD step   C3, K4

Metric pass (`FactorizedTopK.update_state`, factorized_top_k.py:489-512): all-gather the queries, every rank runs the exact
top-k over ITS corpus shard (dr_topk_mips), the per-query lists go back to the query's rank (all-to-all) and are merged there
with dr_topk_merge -- the reduce of the reference's `Streaming.top_k` (factorized_top_k.py:215-233) with ranks as the batches.

All compute goes through a `prims` object (RetrievalPrims = the HIP kernels); the world-2 gloo test substitutes an
oracle-backed one to check the exchange / reduction plan on CPU.
"""
import math
from typing import Optional, Sequence

import torch
import torch.distributed as dist

from . import ops
from .sharded import HipPrims, ShardedEmbeddingExchange, TorchDistTransport, default_transport, _pad4


class RetrievalPrims(HipPrims):
    scores_nt = staticmethod(ops.scores_nt)
    logits_adjust = staticmethod(ops.logits_adjust)
    softmax_ce_rows = staticmethod(ops.softmax_ce_rows)
    softmax_ce_rows_bwd = staticmethod(ops.softmax_ce_rows_bwd)
    topk_mips = staticmethod(ops.topk_mips)
    topk_merge = staticmethod(ops.topk_merge)
    rowdot = staticmethod(ops.rowdot)
    topk_hits = staticmethod(ops.topk_hits)
    gather_i64 = staticmethod(ops.gather_i64)
    linear_fwd_wide = staticmethod(lambda x, W, out: ops.linear_fwd_splitk(x, W, out))   # out += x W, reduction split over the grid


def _all_gather(t, world, tr):
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    return tr.allgather(out, t.contiguous())


def sharded_topk(q_local, corpus_shard, shard_item_ids, k, world, rank, prims=RetrievalPrims, group=None, transport=None):
    """Exact top-k of every local query against a corpus sharded over the ranks; shard_item_ids[j] = item id of local row j.
    Returns (scores [B, k], item ids [B, k]) for the LOCAL queries; identical on any world size up to the order of equal scores."""
    B = q_local.shape[0]
    tr = transport if transport is not None else TorchDistTransport(group)
    q_all = _all_gather(q_local, world, tr)                                      # [world * B, Do]
    n_loc = corpus_shard.shape[0]
    # ONE list length on every rank: the equal-split all-to-all below needs identically shaped tensors, and shard sizes differ by
    # one row when the corpus does not divide by the world size.  The length is min(k, LARGEST shard); a smaller shard
    # pads its lists with (-inf, -1), which every merge ranks last.
    n_max = n_loc
    if world > 1:
        nm = torch.tensor([n_loc], dtype=torch.int64, device=q_local.device)
        tr.allreduce(nm, op=dist.ReduceOp.MAX)
        n_max = int(nm.item())
    k_loc = min(k, n_max)
    k_own = min(k_loc, n_loc)
    s_loc, i_loc = prims.topk_mips(q_all, corpus_shard, k_own)                   # local row numbers
    gid = prims.gather_i64(shard_item_ids, i_loc.reshape(-1)).reshape(i_loc.shape)   # row number -> item id (k_own <= n_loc: no empty slots)
    if k_own < k_loc:
        pad_s = torch.full((s_loc.shape[0], k_loc), float("-inf"), dtype=s_loc.dtype, device=s_loc.device)
        pad_i = torch.full((s_loc.shape[0], k_loc), -1, dtype=gid.dtype, device=gid.device)
        pad_s[:, :k_own].copy_(s_loc)
        pad_i[:, :k_own].copy_(gid)
        s_loc, gid = pad_s, pad_i
    recv_s = torch.empty_like(s_loc)
    recv_i = torch.empty_like(gid)
    tr.alltoall(recv_s, s_loc.contiguous())                                      # chunk r (the lists of rank r's queries) -> rank r
    tr.alltoall(recv_i, gid.contiguous())
    recv_s, recv_i = recv_s.view(world, B, k_loc), recv_i.view(world, B, k_loc)
    s, i = recv_s[0].contiguous(), recv_i[0].contiguous()
    for w in range(1, world):                                                    # Streaming.top_k's reduce, ranks as batches
        s, i = prims.topk_merge(s, i, recv_s[w].contiguous(), recv_i[w].contiguous(), min(k, s.shape[1] + k_loc))
    return s, i


class _ReplicatedTower:
    """Dense(u, relu) x (n - 1), Dense(u_last) with parameters / gradients as views of the engine's flat buckets."""

    def __init__(self, in_dim, units, params, grads, off, gen, device):
        self.Ws, self.bs, self.gWs, self.gbs = [], [], [], []
        d = in_dim
        for u in units:
            pu = _pad4(u)
            W = params[off:off + d * pu].view(d, pu)[:, :u]
            W.copy_((torch.rand((d, u), device=device, generator=gen) * 2 - 1) * math.sqrt(6.0 / (d + u)))
            self.Ws.append(W)
            self.gWs.append(grads[off:off + d * pu].view(d, pu)[:, :u])
            off += d * pu
            self.bs.append(params[off:off + u])
            self.gbs.append(grads[off:off + u])
            off += pu
            d = u
        self.end, self.out_dim = off, d
        self.acts = [1] * (len(units) - 1) + [0] if units else []

    @staticmethod
    def size(in_dim, units):
        n, d = 0, in_dim
        for u in units:
            n += d * _pad4(u) + _pad4(u)
            d = u
        return n

    def forward(self, p, x):
        self.xs = [x]
        for i, (W, b) in enumerate(zip(self.Ws, self.bs)):
            out = torch.empty((x.shape[0], _pad4(W.shape[1])), dtype=torch.float32, device=x.device)[:, :W.shape[1]]
            x = p.linear_fwd(x, W, b, self.acts[i], out=out)
            self.xs.append(x)
        return x

    def backward(self, p, dy):
        """gradients into the bucket (scale 1), returns d_input"""
        for i in range(len(self.Ws) - 1, -1, -1):
            xin = self.xs[i]
            dx = torch.empty((dy.shape[0], _pad4(xin.shape[1])), dtype=torch.float32, device=dy.device)[:, :xin.shape[1]]
            p.linear_bwd_dx(dy, self.Ws[i], relu_src=self.xs[i] if i > 0 else None, out=dx)
            p.linear_bwd_dw(xin, dy, 1.0, self.gWs[i], self.gbs[i])
            dy = dx
        return dy


class ShardedTwoTowerEngine:
    def __init__(self, num_users: int, num_items: int, dim: int = 128, tower_units: Sequence[int] = (256, 128), batch: int = 1024,
                 lr: float = 0.01, temperature: Optional[float] = None, remove_accidental_hits: bool = True, k: int = 100,
                 device="cuda", world=None, rank=None, seed: int = 42, prims=None, group=None, init_tables=None, transport=None):
        self.world = world if world is not None else dist.get_world_size(group)
        self.rank = rank if rank is not None else dist.get_rank(group)
        self.Vu, self.Ni, self.D, self.B, self.lr, self.k, self.group = num_users, num_items, dim, batch, lr, k, group
        self.inv_t = 1.0 / temperature if temperature is not None else 1.0
        self.remove_accidental_hits = remove_accidental_hits
        self.p = prims if prims is not None else RetrievalPrims
        W = self.world
        self.tr = transport if transport is not None else default_transport(self.world, self.rank, group)
        # (alias_world1 off: the two-tower exchanges are 4 MB, and this engine's world-1 GPU test is the one that pushes the
        # exchange code through an RCCL group of one)
        self.ex_u = ShardedEmbeddingExchange(1, num_users, dim, W, self.rank, device, self.p, group, transport=self.tr, alias_world1=False)
        self.ex_i = ShardedEmbeddingExchange(1, num_items, dim, W, self.rank, device, self.p, group, transport=self.tr, alias_world1=False)
        f32 = dict(dtype=torch.float32, device=device)
        g = torch.Generator(device=device)
        g.manual_seed(seed)                       # towers: same seed on every rank -> identical replicas
        units = list(tower_units)
        total = 2 * _ReplicatedTower.size(dim, units) + 4
        self.flat_params = torch.zeros(total, **f32)
        self.flat_grads = torch.zeros(total, **f32)
        self.q_tower = _ReplicatedTower(dim, units, self.flat_params, self.flat_grads, 0, g, device)
        self.c_tower = _ReplicatedTower(dim, units, self.flat_params, self.flat_grads, self.q_tower.end, g, device)
        self.out_dim = self.q_tower.out_dim if units else dim
        std = 1.0 / math.sqrt(dim)
        if init_tables is not None:               # tests: shard given global tables
            full_u, full_i = init_tables
            self.user_table = torch.zeros((self.ex_u.rows_per_shard, dim), **f32)
            self.item_table = torch.zeros((self.ex_i.rows_per_shard, dim), **f32)
            iu, ii = torch.arange(self.rank, num_users, W), torch.arange(self.rank, num_items, W)
            self.user_table[:len(iu)] = full_u[iu].to(device)
            self.item_table[:len(ii)] = full_i[ii].to(device)
        else:
            gt = torch.Generator(device=device)
            gt.manual_seed(seed * 1000 + 17 + self.rank)
            self.user_table = torch.empty((self.ex_u.rows_per_shard, dim), **f32).normal_(0.0, std, generator=gt)
            self.item_table = torch.empty((self.ex_i.rows_per_shard, dim), **f32).normal_(0.0, std, generator=gt)
        self.n_items_local = len(range(self.rank, num_items, W))
        self.shard_item_ids = torch.arange(self.rank, num_items, W, dtype=torch.int64, device=device)   # item id of local row j
        self.zero_base = torch.zeros(1, dtype=torch.int64, device=device)
        self.loss = torch.zeros(1, **f32)
        self.corpus = None
        self._cuda = torch.device(device).type == "cuda"
        # The two towers are independent until the score matrix, and every one of their kernels is small (64 - 128 blocks on 256
        # CUs): the candidate side (row exchange + tower forward; dc all-reduce, tower backward, gradient exchange + owner-side
        # update) CAN run on a second stream beside the query side, as in two_tower_engine.TwoTowerEngine (DR_TT_STREAMS=1; round 4).
        # Built, parity-tested on two ranks, measured at world 1 and left OFF: 2.30 ms per step against 2.25 on one stream (same
        # call, alternating) -- this step is bound by the host (two routes with a blocking split-size read each, ~120 small
        # launches), not by the device, so a second stream only adds fork / join packets.  Collectives are ISSUED in one fixed order
        # on every rank (candidate side first), whatever stream carries them.
        import os as _os
        self.two_streams = self._cuda and _os.environ.get("DR_TT_STREAMS", "0") == "1"
        if self.two_streams:
            self.side = torch.cuda.Stream(device=device)
            self.ev_fork, self.ev_c = torch.cuda.Event(), torch.cuda.Event()

    def enable_kernel_events(self, on):       # bench.py contract; the sharded step reports no per-kernel rows
        pass

    def kernel_event_summary(self):
        return {}

    # ---- embeddings of the rank's batch --------------------------------------------------------------------------------------
    def _embed(self, ex, keys, table, hashed):
        B, D = keys.shape[0], self.D
        r = ex.route(keys.reshape(B, 1), hashed=hashed)
        rows, _ = ex.fetch(r, table, None)
        emb = torch.empty((B, D), dtype=torch.float32, device=rows.device)
        self.p.emb_pool_fwd(r.pos, 1, None, self.zero_base, rows, None, None, ld_concat=D, concat=emb, want_sum_x=False,
                            want_fm=False)
        return r, emb

    def embeddings(self, user_keys, item_ids):
        if self.two_streams:
            main = torch.cuda.current_stream()
            self.ev_fork.record(main)
            item_ids.record_stream(self.side)
            with torch.cuda.stream(self.side):
                self.side.wait_event(self.ev_fork)
                ri, i = self._embed(self.ex_i, item_ids, self.item_table, False)
                c = self.c_tower.forward(self.p, i) if self.c_tower.Ws else i
                self.ev_c.record(self.side)
            ru, u = self._embed(self.ex_u, user_keys, self.user_table, True)
            q = self.q_tower.forward(self.p, u) if self.q_tower.Ws else u
            main.wait_event(self.ev_c)
            for t in [i, c] + list(getattr(self.c_tower, "xs", [])):
                t.record_stream(main)
            return ru, ri, q, c
        ru, u = self._embed(self.ex_u, user_keys, self.user_table, True)
        ri, i = self._embed(self.ex_i, item_ids, self.item_table, False)
        q = self.q_tower.forward(self.p, u) if self.q_tower.Ws else u
        c = self.c_tower.forward(self.p, i) if self.c_tower.Ws else i
        return ru, ri, q, c

    # ---- training step ------------------------------------------------------------------------------------------------------
    def train_step(self, user_keys, item_ids):
        """One SGD step on the Retrieval loss of the GLOBAL batch; returns this rank's part of the loss (their sum over ranks is
        the reference's loss value)."""
        p, W, B, lr = self.p, self.world, user_keys.shape[0], self.lr
        ru, ri, q, c = self.embeddings(user_keys, item_ids)
        c_all = _all_gather(c, W, self.tr)                                             # C5
        ids_all = _all_gather(item_ids.reshape(B), W, self.tr)
        scores = p.scores_nt(q, c_all).contiguous()                                    # sbcnm.py:129 (this rank's rows)
        labels = torch.zeros((B, W * B), dtype=torch.float32, device=q.device)
        ar = torch.arange(B, device=q.device)
        labels[ar, self.rank * B + ar] = 1.0                                           # :134 tf.eye, shifted to the rank's columns
        if self.remove_accidental_hits:
            scores = p.logits_adjust(scores, labels, cand_ids=ids_all)                 # :66-75
        self.loss = p.softmax_ce_rows(scores, labels, self.inv_t)                      # :148-151
        G = p.softmax_ce_rows_bwd(scores, labels, self.inv_t, None, 1.0)
        wide = getattr(p, "linear_fwd_wide", None)       # [B, W B] x [W B, out]: few output tiles, long reduction -> split over the grid
        if wide is not None and B >= 2048:
            dq = torch.zeros((B, _pad4(self.out_dim)), dtype=torch.float32, device=q.device)[:, :self.out_dim]
            wide(G, c_all, dq)                                                         # dq = G c_all
        else:
            dq = torch.empty((B, _pad4(self.out_dim)), dtype=torch.float32, device=q.device)[:, :self.out_dim]
            p.linear_fwd(G, c_all, None, 0, out=dq)                                    # dq = G c_all
        def cand_side():
            dc_all = torch.zeros((W * B, _pad4(self.out_dim)), dtype=torch.float32, device=q.device)[:, :self.out_dim]
            p.linear_bwd_dw(G, q, 1.0, dc_all)                                         # this rank's part of G^T q
            dc_buf = dc_all if dc_all.is_contiguous() else dc_all.contiguous()
            self.tr.allreduce(dc_buf)                                                  # candidates' gradients from every rank's queries
            dc = dc_buf[self.rank * B:(self.rank + 1) * B]
            d_i = self.c_tower.backward(p, dc) if self.c_tower.Ws else dc
            d_i = d_i if d_i.stride(1) == 1 else d_i.contiguous()
            g_rows, _ = self.ex_i.pack(ri, d_i, None, None, None)
            self.ex_i.exchange_and_apply(ri, g_rows, None, -lr, self.item_table, None)  # C3 + K4 (item rows)
            return dc_buf

        self.flat_grads.zero_()
        if self.two_streams:
            # candidate side of the backward on the second stream; the two towers' gradients live in disjoint parts of the bucket
            main = torch.cuda.current_stream()
            self.ev_fork.record(main)
            for t in (G, q):
                t.record_stream(self.side)
            with torch.cuda.stream(self.side):
                self.side.wait_event(self.ev_fork)
                keep = cand_side()
                self.ev_c.record(self.side)
            d_u = self.q_tower.backward(p, dq) if self.q_tower.Ws else dq
            main.wait_event(self.ev_c)
            keep.record_stream(main)
        else:
            d_u = self.q_tower.backward(p, dq) if self.q_tower.Ws else dq
            cand_side()
        self.tr.allreduce(self.flat_grads)                                             # C4
        p.axpy(-lr, self.flat_grads, self.flat_params)
        d_u = d_u if d_u.stride(1) == 1 else d_u.contiguous()
        g_rows, _ = self.ex_u.pack(ru, d_u, None, None, None)
        self.ex_u.exchange_and_apply(ru, g_rows, None, -lr, self.user_table, None)      # C3 + K4 (user rows)
        return self.loss

    # ---- corpus index + metric pass -------------------------------------------------------------------------------------------
    def index_corpus(self, chunk: int = 1 << 16):
        self._corpus_index = None
        rows = self.item_table[:self.n_items_local]
        if not self.c_tower.Ws:
            self.corpus = rows
            return self.corpus
        self.corpus = torch.empty((self.n_items_local, self.out_dim), dtype=torch.float32, device=rows.device)
        for r0 in range(0, self.n_items_local, chunk):
            r1 = min(self.n_items_local, r0 + chunk)
            self.corpus[r0:r1].copy_(self.c_tower.forward(self.p, rows[r0:r1]))
        return self.corpus

    def topk(self, q, k=None):
        corpus = self.corpus
        if self.p is RetrievalPrims and self.c_tower.Ws:
            # the shard's side of the scan (amax record + fp16 planes), once per index_corpus instead of once per call (ops.TopKIndex);
            # never for the live table (no item tower): the training kernels write it without moving a version counter
            ix = getattr(self, "_corpus_index", None)
            if ix is None or ix.cand.data_ptr() != corpus.data_ptr():
                ix = self._corpus_index = ops.TopKIndex(corpus)
            corpus = ix
        return sharded_topk(q, corpus, self.shard_item_ids, k or self.k, self.world, self.rank, self.p, self.group, self.tr)

    def metric_step(self, user_keys, item_ids, ks=(1, 5, 10, 50, 100)):
        assert self.corpus is not None, "The `index_corpus` method must be called first"
        _, _, q, c = self.embeddings(user_keys, item_ids)
        pos = self.p.rowdot(q, c)
        scores, _ = self.topk(q)
        ks_t = torch.tensor(list(ks), dtype=torch.int32, device=q.device)
        hits = torch.zeros(len(ks), dtype=torch.int64, device=q.device)
        self.p.topk_hits(pos, scores, ks_t, hits)
        return hits
