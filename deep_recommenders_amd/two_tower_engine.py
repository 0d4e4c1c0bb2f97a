"""Lean two-tower (DSSM-style) retrieval engine — BASELINE.json config 5: 1 M-item corpus, embedding dim 128, in-batch
softmax over the batch's 8192 candidates, FactorizedTopK metric against the whole corpus.

    q = query_tower(embed(hash(user keys)))          K1 hash -> K3 gather -> K7 MLP
    c = item_tower(embed(item ids))                  K3 gather -> K7 MLP
    loss = Retrieval()(q, c, candidate_ids=item ids) K9  keras/models/retrieval/sbcnm.py:120-163 of the reference:
           scores = q c^T (:129), labels = eye (:134), accidental hits masked when candidate_ids are given (:66-75),
           / temperature (:148-149), CategoricalCrossentropy(from_logits, SUM) (:100-102,151)
    metric = FactorizedTopK(corpus)(q, c)            K10 factorized_top_k.py:489-512: top-{1,5,10,50,100} accuracy of the
           positive against the exact top-k of q . corpus^T

The reference ships the task layer (`Retrieval`) and the index / metric layers, not a two-tower model class (its README
lists DSSM as a model family built from them): this engine composes the same kernels the `Retrieval`, `BruteForce` and
`FactorizedTopK` classes call, with no autograd tape and every update fused into the producing kernel (SGD), to put an
end-to-end number on config 5.  The B x B score matrix is never written in the forward (K9's epilogue keeps the
log-sum-exp); the backward materialises the score gradient once and feeds two GEMMs.
"""
import math
from typing import List, Optional, Sequence

import torch

from . import ops


def _pad4(n):
    return (n + 3) // 4 * 4


class _Tower:
    """Dense(u, relu) x (n - 1), Dense(u_last) — glorot-uniform kernels, zero biases ([TF] B8); forward keeps the activations,
    backward returns d_input and applies dst += scale * grad inside the wgrad kernels."""

    def __init__(self, in_dim: int, units: Sequence[int], batch: int, device, gen):
        self.Ws: List[torch.Tensor] = []
        self.bs: List[torch.Tensor] = []
        d = in_dim
        for u in units:
            limit = math.sqrt(6.0 / (d + u))
            W = (torch.rand((d, _pad4(u)), device=device, generator=gen) * 2 - 1) * limit
            W[:, u:].zero_()
            self.Ws.append(W[:, :u])
            self.bs.append(torch.zeros(u, dtype=torch.float32, device=device))
            d = u
        self.acts = [1] * (len(units) - 1) + [0] if units else []
        f32 = dict(dtype=torch.float32, device=device)
        self.hs = [torch.empty((batch, _pad4(u)), **f32)[:, :u] for u in units]
        self.dhs = [torch.empty((batch, _pad4(u)), **f32)[:, :u] for u in units[:-1]]
        self.dw_ws = [ops.linear_bwd_dw_workspace(batch, W.shape[0], W.shape[1], device) for W in self.Ws]
        self.out_dim = d

    def forward(self, x, k, tag):
        self.x_in = x
        for i, (W, b) in enumerate(zip(self.Ws, self.bs)):
            x = k("%s_fwd_L%d" % (tag, i), "mfma", 2.0 * x.shape[0] * W.shape[0] * W.shape[1],
                  lambda x=x, W=W, b=b, i=i: ops.linear_fwd(x, W, b, self.acts[i], out=self.hs[i][:x.shape[0]]))
        return x

    def forward_nograd(self, x):
        for i, (W, b) in enumerate(zip(self.Ws, self.bs)):
            x = ops.linear_fwd(x, W, b, self.acts[i])
        return x

    def backward(self, dy, d_in, scale, k, tag):
        for i in range(len(self.Ws) - 1, -1, -1):
            xin = self.x_in if i == 0 else self.hs[i - 1]
            W = self.Ws[i]
            fl = 2.0 * dy.shape[0] * W.shape[0] * W.shape[1]
            dx = d_in if i == 0 else self.dhs[i - 1]
            rs = self.hs[i - 1] if i > 0 else None                       # ReLU' of the layer below
            k("%s_bwd_dx_L%d" % (tag, i), "mfma", fl, lambda dy=dy, W=W, rs=rs, dx=dx: ops.linear_bwd_dx(dy, W, relu_src=rs, out=dx))
            k("%s_bwd_dw_L%d" % (tag, i), "mfma", fl,
              lambda xin=xin, dy=dy, W=W, i=i: ops.linear_bwd_dw(xin, dy, scale, W, self.bs[i], workspace=self.dw_ws[i]))
            dy = dx
        return dy


class TwoTowerEngine:
    def __init__(self, num_users: int, num_items: int, dim: int = 128, tower_units: Sequence[int] = (256, 128),
                 batch: int = 8192, lr: float = 0.01, temperature: Optional[float] = None,
                 remove_accidental_hits: bool = True, k: int = 100, device="cuda", seed: int = 42):
        self.Vu, self.Ni, self.D, self.B, self.lr, self.k = num_users, num_items, dim, batch, lr, k
        self.inv_t = 1.0 / temperature if temperature is not None else 1.0
        self.remove_accidental_hits = remove_accidental_hits
        g = torch.Generator(device=device)
        g.manual_seed(seed)
        std = 1.0 / math.sqrt(dim)                                       # N(0,1)/sqrt(D)
        self.user_table = torch.empty((num_users, dim), dtype=torch.float32, device=device).normal_(0.0, std, generator=g)
        self.item_table = torch.empty((num_items, dim), dtype=torch.float32, device=device).normal_(0.0, std, generator=g)
        units = list(tower_units)
        self.q_tower = _Tower(dim, units, batch, device, g)
        self.c_tower = _Tower(dim, units, batch, device, g)
        self.out_dim = self.q_tower.out_dim if units else dim
        f32 = dict(dtype=torch.float32, device=device)
        i64 = dict(dtype=torch.int64, device=device)
        self.zero_base = torch.zeros(1, **i64)
        self.user_buckets = torch.full((1,), num_users, **i64)
        self.uid = torch.empty((batch, 1), **i64)
        self.u_emb = torch.empty((batch, dim), **f32)
        self.i_emb = torch.empty((batch, dim), **f32)
        self.d_u_emb = torch.empty((batch, dim), **f32)
        self.d_i_emb = torch.empty((batch, dim), **f32)
        self.dq = torch.empty((batch, _pad4(self.out_dim)), **f32)[:, :self.out_dim]
        self.dc = torch.empty((batch, _pad4(self.out_dim)), **f32)[:, :self.out_dim]
        self.side = torch.cuda.Stream(device=device)
        self.ev_fork, self.ev_c = torch.cuda.Event(), torch.cuda.Event()
        tiles = ((batch + 127) // 128) * ((self.out_dim + 127) // 128)
        self.dq_ws = ops.linear_fwd_splitk_workspace(batch, batch, self.out_dim, device) if (tiles < 256 and batch >= 2048) else None
        self.loss = torch.zeros(1, **f32)
        self.u_plan = ops.SortPlan(batch, device)
        self.i_plan = ops.SortPlan(batch, device)
        self.corpus = None                    # [num_items, out_dim] item-tower outputs (index for the metric pass)
        self._topk_ws = None
        self._events = None

    # ---- per-kernel HIP events (bench.py) -----------------------------------------------------------------------------
    def enable_kernel_events(self, on: bool):
        self._events = {} if on else None

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
        torch.cuda.synchronize()
        out = {}
        for name, (bound, work, evs) in (self._events or {}).items():
            ms = sum(s.elapsed_time(e) for s, e in evs) / len(evs)
            out[name] = {"bound": bound, "ms": ms, "n": len(evs), "alg_bytes" if bound == "hbm" else "alg_flops": work}
        return out

    def flops_step(self):
        B, D = self.B, self.out_dim
        fl = 0.0
        for t in (self.q_tower, self.c_tower):
            fl += 3 * sum(2.0 * B * W.shape[0] * W.shape[1] for W in t.Ws)
        return fl + 4 * 2.0 * B * B * D                    # scores in the forward, again in the gradient, dq, dc

    # ---- embeddings ---------------------------------------------------------------------------------------------------
    def embeddings(self, user_keys, item_ids):
        """(query embeddings, candidate embeddings) of a batch, no gradient state kept beyond the engine buffers.
        The two towers are independent until the score matrix and each of their GEMMs is small (64 - 128 blocks at B = 8192 for
        256 CUs): the candidate side (gather + tower) runs on a second stream beside the query side."""
        B, D = user_keys.shape[0], self.D
        main = torch.cuda.current_stream()
        self.ev_fork.record(main)
        item_ids.record_stream(self.side)
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.ev_fork)
            self._k("emb_gather_items", "hbm", B * (8 * D + 8),
                    lambda: ops.emb_pool_fwd(item_ids.reshape(B, 1), 1, None, self.zero_base, self.item_table, None, None,
                                             ld_concat=D, concat=self.i_emb[:B], want_sum_x=False, want_fm=False))
            c = self.c_tower.forward(self.i_emb[:B], self._k, "c_tower") if self.c_tower.Ws else self.i_emb[:B]
            self.ev_c.record(self.side)
        self._k("hash_bucket_i64", "hbm", B * 16,
                lambda: ops.hash_bucket_i64(user_keys.reshape(B, 1), self.user_buckets, out=self.uid[:B]))
        self._k("emb_gather_users", "hbm", B * (8 * D + 8),
                lambda: ops.emb_pool_fwd(self.uid[:B], 1, None, self.zero_base, self.user_table, None, None, ld_concat=D,
                                         concat=self.u_emb[:B], want_sum_x=False, want_fm=False))
        q = self.q_tower.forward(self.u_emb[:B], self._k, "q_tower") if self.q_tower.Ws else self.u_emb[:B]
        main.wait_event(self.ev_c)
        return q, c

    # ---- training step ------------------------------------------------------------------------------------------------
    def train_step(self, user_keys, item_ids, sample_weight=None, candidate_sampling_probability=None):
        """One SGD step on `Retrieval`'s loss for a batch of (user key, positive item id) pairs; returns the loss (device)."""
        B, D, lr = user_keys.shape[0], self.D, self.lr
        item_ids = item_ids.reshape(B, 1)
        q, c = self.embeddings(user_keys, item_ids)
        cand_ids = item_ids.reshape(B) if self.remove_accidental_hits else None
        fl_s = 2.0 * B * B * self.out_dim
        loss, row_lse, _ = self._k("inbatch_softmax_fwd", "mfma", fl_s,
                                   lambda: ops.inbatch_softmax_fwd(q, c, candidate_sampling_probability, cand_ids, sample_weight,
                                                                   self.inv_t))
        self.loss = loss
        G = self._k("inbatch_softmax_grad_scores", "mfma", fl_s,
                    lambda: ops.inbatch_softmax_grad_scores(q, c, row_lse, 1.0, candidate_sampling_probability, cand_ids,
                                                            sample_weight, self.inv_t))
        dq, dc = self.dq[:B], self.dc[:B]
        # the candidate side of the backward (dc = G^T q, tower, sort + scatter into the item table) on the second stream, the query
        # side here; the next step's candidate gather follows on that same stream, the caller's stream waits for both
        main = torch.cuda.current_stream()
        self.ev_fork.record(main)
        for t in (G, q, c, item_ids):
            t.record_stream(self.side)
        # dq = G c: [B, B] x [B, out] has only B / 128 output tiles (64 at B = 8192) for 256 CUs -> reduction split over the grid,
        # slices summed in a fixed order (393 -> ~120 us at B = 8192; the plain GEMM when the tiles alone fill the machine)
        if self.dq_ws is not None:
            dq.zero_()
            self._k("retrieval_dq", "mfma", fl_s, lambda: ops.linear_fwd_splitk(G, c, dq, workspace=self.dq_ws))
        else:
            self._k("retrieval_dq", "mfma", fl_s, lambda: ops.linear_fwd(G, c, out=dq))
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.ev_fork)
            dc.zero_()
            self._k("retrieval_dc", "mfma", fl_s, lambda: ops.linear_bwd_dw(G, q, 1.0, dc))             # dc = G^T q
            d_i = self.c_tower.backward(dc, self.d_i_emb[:B], -lr, self._k, "c_tower") if self.c_tower.Ws else dc
            ops.emb_sort_slots(item_ids, self.zero_base, self.Ni, self.i_plan)
            self._k("emb_scatter_items", "hbm", B * (12 * D + 8),
                    lambda: ops.emb_pool_bwd_sorted(item_ids, self.zero_base, self.i_plan, D, self.Ni, d_i, None, -lr,
                                                    self.item_table, None, None))
            self.ev_c.record(self.side)
        d_u = self.q_tower.backward(dq, self.d_u_emb[:B], -lr, self._k, "q_tower") if self.q_tower.Ws else dq
        # K4: rows sorted so that each touched row gets one plain read-modify-write (fused SGD)
        ops.emb_sort_slots(self.uid[:B], self.zero_base, self.Vu, self.u_plan)
        self._k("emb_scatter_users", "hbm", B * (12 * D + 8),
                lambda: ops.emb_pool_bwd_sorted(self.uid[:B], self.zero_base, self.u_plan, D, self.Vu, d_u, None, -lr,
                                                self.user_table, None, None))
        main.wait_event(self.ev_c)
        return self.loss

    # ---- corpus index + FactorizedTopK metric pass (factorized_top_k.py:275-334, 489-512) -------------------------------
    def index_corpus(self, chunk: int = 1 << 16):
        """Item-tower outputs of the whole corpus (BruteForce.index over `candidates.map(item_model)`)."""
        if not self.c_tower.Ws:
            self.corpus = self.item_table
            self._corpus_index = None
            return self.corpus
        if self.corpus is None or self.corpus.data_ptr() == self.item_table.data_ptr():
            self.corpus = torch.empty((self.Ni, self.out_dim), dtype=torch.float32, device=self.item_table.device)
        for r0 in range(0, self.Ni, chunk):
            r1 = min(self.Ni, r0 + chunk)
            self.corpus[r0:r1].copy_(self.c_tower.forward_nograd(self.item_table[r0:r1]))
        self._corpus_index = None                # (rebuilt by the next topk: the corpus' amax record and fp16 planes, once per index)
        return self.corpus

    def topk(self, q, k=None):
        """Exact top-k of q . corpus^T: (scores [B, k], item ids [B, k]); ties -> lower id ([TF] B13)."""
        k = k or self.k
        if self._topk_ws is None or self._topk_ws[0] != (q.shape[0], k):
            nb = ops.lib().dr_topk_workspace_bytes(q.shape[0], self.Ni, int(k))
            self._topk_ws = ((q.shape[0], k), torch.empty(max(1, nb // 4), dtype=torch.float32, device=q.device))
        if self.corpus.data_ptr() == self.item_table.data_ptr():
            # no item tower: the corpus IS the live table, which the training kernels write through raw pointers (no version
            # counter moves) -- nothing derived from it may be kept across calls
            return ops.topk_mips(q, self.corpus, k, workspace=self._topk_ws[1])
        if getattr(self, "_corpus_index", None) is None or self._corpus_index.cand.data_ptr() != self.corpus.data_ptr():
            self._corpus_index = ops.TopKIndex(self.corpus)
        return ops.topk_mips(q, self._corpus_index, k, workspace=self._topk_ws[1])

    def metric_step(self, user_keys, item_ids, ks=(1, 5, 10, 50, 100)):
        """FactorizedTopK.update_state for one batch: hit counts of the positive among [positive ∪ top-k] for each k in ks
        (in_top_k counting, [TF] B14).  Returns a device int64 tensor of len(ks) hit counts."""
        assert self.corpus is not None, "The `index_corpus` method must be called first"
        q, c = self.embeddings(user_keys, item_ids)
        pos = ops.rowdot(q, c)
        scores, _ = self._k("topk_mips", "mfma", 2.0 * q.shape[0] * self.Ni * self.out_dim, lambda: self.topk(q))
        ks_t = torch.tensor(list(ks), dtype=torch.int32, device=q.device)
        hits = torch.zeros(len(ks), dtype=torch.int64, device=q.device)
        ops.topk_hits(pos, scores, ks_t, hits)
        return hits
