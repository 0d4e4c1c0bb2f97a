"""Lean DeepFM training engine: one fused training step (hash -> gather+pool+FM -> MLP tower -> loss ->
tower backward -> scatter-add backward) over preallocated HBM buffers, every optimizer update fused into
the producing kernel (SGD: dst += -lr * grad inside the scatter / wgrad epilogues).

This is the path bench.py and __graft_entry__.smoke() time; the reference-shaped classes in
deep_recommenders_amd.keras / .estimator run the same kernels through autograd.  Model = the reference's
DeepFM (keras/models/ranking/deepfm.py:36-47 of the reference): sigmoid(FM(first-order, stacked embeddings)
+ Sequential(Dense(u, relu)..., Dense(1))(concat embeddings [+ dense features])); loss =
tf.losses.sigmoid_cross_entropy on the logits (examples/train_fm_on_movielens_estimator.py:46).
"""
import math
from typing import List, Optional, Sequence

import torch

from . import ops


_SIDE_STREAMS = {}


def _side_streams(device):
    """The engines' two side streams, ONE pair per device and process: HIP maps streams onto a few hardware queues, and a second engine's
    streams of its own ended up sharing a queue with the training stream (bench.py's strict-fp32 leg, a second engine in the process:
    1.63 - 1.67 ms per step against 1.37 with shared streams -- every refresh launch serialised the training stream behind it)."""
    key = str(torch.device(device))
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = (torch.cuda.Stream(device=device), torch.cuda.Stream(device=device))
    return _SIDE_STREAMS[key]


def _pad4(n):
    return (n + 3) // 4 * 4


def _os_env(name, default):
    import os
    return os.environ.get(name, default)


class DeepFMEngine:
    def __init__(self, num_fields: int, vocab_per_field: int, dim: int, dnn_units: Sequence[int], batch: int,
                 num_dense: int = 0, lr: float = 0.01, device="cuda", seed: int = 42, hashed: bool = True,
                 table_init_std: Optional[float] = None, lin_init_std: float = 0.0, sorted_bwd: bool = True,
                 optimizer: str = "sgd", beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8):
        self.F, self.V, self.D, self.B = num_fields, vocab_per_field, dim, batch
        self.Nd, self.lr, self.dev, self.hashed = num_dense, lr, device, hashed
        F, V, D, B = self.F, self.V, self.D, self.B
        g = torch.Generator(device=device)
        g.manual_seed(seed)
        self.R = F * V
        # ---- parameters (HBM resident; 26 x 10 M x 64 fp32 = 66.6 GB at config 3) -------------------
        self.table = torch.empty((self.R, D), dtype=torch.float32, device=device)
        std = table_init_std if table_init_std is not None else 1.0 / math.sqrt(D)     # [TF] B4
        chunk = 1 << 24
        for r0 in range(0, self.R, chunk):     # in place, chunked: no second 66 GB temporary
            self.table[r0:r0 + chunk].normal_(0.0, std, generator=g).clamp_(-2 * std, 2 * std)
        self.lin_w = torch.zeros(self.R, dtype=torch.float32, device=device)           # zeros init (fm.py:16-20)
        if lin_init_std > 0:
            self.lin_w.normal_(0.0, lin_init_std, generator=g)
        self.in_dim = F * D + num_dense
        self.ld = _pad4(self.in_dim)
        units = list(dnn_units) + [1]
        # dense parameters live in ONE flat buffer (weights, biases, the first-order bias): the Adam mode then needs one
        # gradient bucket and one dr_adam_step launch; the SGD mode updates the views in place from inside the wgrad kernels
        shapes, d = [], self.in_dim
        for u in units:
            shapes.append((d, u))
            d = u
        total = sum(k * _pad4(u) + _pad4(u) for k, u in shapes) + 4
        self.flat_params = torch.zeros(total, dtype=torch.float32, device=device)
        self.Ws: List[torch.Tensor] = []
        self.bs: List[torch.Tensor] = []
        self._views = []         # (offset, k, pu, u) of every weight, then biases, for the gradient-bucket views
        off = 0
        for k, u in shapes:                                                            # [TF] B8 glorot uniform
            pu = _pad4(u)
            limit = math.sqrt(6.0 / (k + u))
            Wfull = self.flat_params[off:off + k * pu].view(k, pu)
            Wfull.copy_((torch.rand((k, pu), device=device, generator=g) * 2 - 1) * limit)
            Wfull[:, u:].zero_()
            self.Ws.append(Wfull[:, :u])
            w_off = off
            off += k * pu
            self.bs.append(self.flat_params[off:off + u])
            self._views.append((w_off, k, pu, u, off))
            off += pu
        self.lin_bias = self.flat_params[off:off + 1]
        self._bias_off = off
        assert optimizer in ("sgd", "adam", "adam_tf") and (optimizer == "sgd" or sorted_bwd), "adam needs the sorted K4"
        # "adam_tf": tf.train.AdamOptimizer's NON-lazy behaviour on the tables (every row moves on every step while its first
        # moment is non-zero), evaluated lazily: a row's missed decay-only steps are replayed when the row is next
        # looked up (dr_adam_catchup_rows, before the forward), then the fused row-wise K4 applies the step itself.  Every value
        # the model reads equals TF's dense update; `adam_flush()` brings the whole table up to date (export / inspection).
        self.adam_tf = optimizer == "adam_tf"
        if self.adam_tf:
            optimizer = "adam"
        self.optimizer, self.beta1, self.beta2, self.eps, self.t = optimizer, beta1, beta2, eps, 0
        if optimizer == "adam":
            # gradient bucket + moments for the dense parameters; row-wise moments for the tables (2 x 66.6 GB at config 3:
            # table + m + v = 200 GB of the 288 GB HBM)
            self.flat_grads = torch.zeros_like(self.flat_params)
            self.flat_m = torch.zeros_like(self.flat_params)
            self.flat_v = torch.zeros_like(self.flat_params)
            self.gWs = [self.flat_grads[wo:wo + k * pu].view(k, pu)[:, :u] for wo, k, pu, u, bo in self._views]
            self.gbs = [self.flat_grads[bo:bo + u] for wo, k, pu, u, bo in self._views]
            self.g_lin_bias = self.flat_grads[self._bias_off:self._bias_off + 1]
            self.m_table = torch.zeros_like(self.table)
            self.v_table = torch.zeros_like(self.table)
            # first-order moments interleaved in ONE [R, 2] array (m_lin, v_lin are its two columns; the kernels see v_lin == m_lin + 1
            # and stride by 2): a row's first-order Adam state is then one line to read and one to write instead of two each -- the
            # fused Adam K4 is bound by line operations like the SGD one (round 4; DR_ADAM_LIN_PACKED=0: two arrays)
            if _os_env("DR_ADAM_LIN_PACKED", "1") == "1":
                self.mv_lin = torch.zeros((self.R, 2), dtype=torch.float32, device=device)
                self.m_lin, self.v_lin = self.mv_lin[:, 0], self.mv_lin[:, 1]
            else:
                self.m_lin = torch.zeros_like(self.lin_w)
                self.v_lin = torch.zeros_like(self.lin_w)
            self.row_step = torch.zeros(self.R, dtype=torch.int32, device=device) if self.adam_tf else None
        import os as _os
        self.acts = [1] * len(dnn_units) + [0]
        # Wide layers: forward and dgrad on pre-split weights (dr_bf3_linear_nt: the activation operand stays fp32 and is split
        # in registers, only the weight is staged through the LDS): 360 -> 264 us forward, 400 -> 306 us dgrad at the first layer
        # of config 3.  The planes are refreshed after every update of the weight.  DR_PLANES=0: the in-kernel-split GEMMs.
        self.wplanes = [ops.WeightPlanes(W) if (_os.environ.get("DR_PLANES", "1") == "1" and u > 1
                                                and ops.planes_worthwhile(B, W.shape[0], W.shape[1])) else None
                        for W, u in zip(self.Ws, units)]
        self.dw_ws = [ops.linear_bwd_dw_workspace(B, W.shape[0], W.shape[1], device) for W in self.Ws]
        # wide layers: the register-split wgrad (dr_bf3_wgrad) on the same condition as the planes forward / dgrad
        self.wg_ws = [ops.bf3_wgrad_workspace(B, W.shape[0], W.shape[1], device) if wp is not None else None
                      for W, wp in zip(self.Ws, self.wplanes)]
        # narrow layers (N <= 32, K in {128,256,512}) below the first: dx + dW + db fused in one pass over the activations
        self.fuse_narrow = _os.environ.get("DR_FUSE_NARROW", "1") == "1"
        # last hidden layer (<= 32 units, relu) + Dense(1) + loss + Dense(1) backward fused into one GEMM epilogue
        nl = len(self.Ws)
        self.fuse_head = (_os.environ.get("DR_FUSE_HEAD", "1") == "1" and nl >= 2 and self.Ws[-1].shape[1] == 1
                          and self.Ws[-2].shape[1] <= 32 and self.acts[-2] == 1)
        self.head_ws = ops.tower_head_workspace(B, device) if self.fuse_head else None
        self._head_done = False
        self._tail_done = False
        self.narrow_ws = [ops.linear_bwd_narrow_workspace(B, W.shape[0], W.shape[1], device)
                          if (i > 0 and self.acts[i - 1] and ops.linear_bwd_narrow_supported(B, W.shape[0], W.shape[1])) else None
                          for i, W in enumerate(self.Ws)]
        # The whole tower tail in ONE pass over the activations below the last hidden layer (round 5, dr_tower_tail_fused): the fused head
        # and the narrow backward of that SAME layer read h (67 MB at config 3) twice, in two latency-bound launches (51 + 60 us); one
        # kernel stages each 32-row chunk through the LDS once for the head's product and keeps it in registers for the backward.
        # Training steps only; needs the fused head, the narrow backward's shape domain (K in {128, 256}) and a ReLU layer below.
        # DR_FUSE_TAIL=0: the two launches of rounds 1-4.
        self.fuse_tail = (_os.environ.get("DR_FUSE_TAIL", "1") == "1" and self.fuse_head and self.fuse_narrow and nl >= 3
                          and self.narrow_ws[nl - 2] is not None and self.acts[nl - 3] == 1
                          and ops.tower_tail_supported(B, self.Ws[-2].shape[0], self.Ws[-2].shape[1]))
        self.tail_ws = ops.tower_tail_workspace(B, self.Ws[-2].shape[0], device) if self.fuse_tail else None
        # ---- constant metadata ---------------------------------------------------------------------
        self.row_base = torch.arange(F, device=device, dtype=torch.int64) * V
        self.col_start = torch.arange(F + 1, device=device, dtype=torch.int32)
        self.col_buckets = torch.full((F,), V if hashed else 0, dtype=torch.int64, device=device)
        # ---- activations / gradients (preallocated once) ---------------------------------------------
        f32 = dict(dtype=torch.float32, device=device)
        # ids / slot plan are double-buffered: with next_keys handed to train_step, the NEXT batch's hash + slot sort run on the side
        # stream beside this step's K4 (see _prefetch_next); `ids` / `plan` are the current buffers
        self._ids = [torch.empty((B, F), dtype=torch.int64, device=device) for _ in range(2)]
        self.cur = 0
        self._pref = None            # identity of the keys whose ids / plan sit in buffer cur ^ 1
        self._pref_dense = None      # identity of the dense features already placed in concat / dense_pad for that batch
        self._next_keys = None
        self._next_dense = None
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
        self._events = None      # name -> [bound, work, [(start, end), ...]]
        self._ev_every, self._ev_step, self._ev_live = 1, 0, False
        self._ev_pool = []
        self._ev_only = None
        # ---- deterministic backward: slots sorted by table row on a side stream (depends only on ids) ----
        # Schedule experiments and their outcomes are recorded in DESIGN.md section 3 (sort started with K3 / in the backward /
        # on a high-priority stream, K4 overlapped with the first-layer wgrad, FM gradient folded into the dgrad epilogue):
        # what remains here is the best measured schedule plus DR_SORT_INLINE=1 (sort on the training stream, for profiling).
        self.sorted_bwd = sorted_bwd
        self.sort_inline = _os.environ.get("DR_SORT_INLINE", "0") == "1"
        # Where the side-stream sort starts.  Round 1: after K3, hidden under the in-kernel-split GEMMs (2 blocks / CU left room for
        # it).  The register-split GEMMs are persistent 512-thread blocks that own their CU (144 KB of LDS, all registers): the
        # sort only advances in the gaps between them (1.1 ms instead of 0.17) and K4 ends up waiting for it.  Started with K3 it
        # shares HBM with the gather instead (round 1 measured +46 us on K3).  DR_SORT_WITH_K3=0 / 1 selects.
        self.sort_with_k3 = _os.environ.get("DR_SORT_WITH_K3", "1") == "1"
        # K3 fused into the first layer's GEMM (dr_bf3_emb_linear_fwd: the register-split kernel gathers its activation rows from
        # the tables and produces concat / sum_x / fm_logit on the way).  Needs D == 64 and the planes path; DR_FUSE_K3=0: K3 as
        # its own kernel.
        self.fuse_k3 = _os.environ.get("DR_FUSE_K3", "1") == "1" and D == 64 and self.Nd <= 32 and V <= (1 << 24)
        # (two buffers, indexed like the ids / plan buffers -- round 6: the next batch's dense features are placed with the early chain,
        # while this step's forward and wgrad still read the current buffer; see _prefetch_early)
        self._dense_pads = ([torch.zeros((B, 32), dtype=torch.float32, device=device) for _ in range(2)]
                            if (self.fuse_k3 and self.Nd) else None)
        self._dense_amaxs = None
        # `concat` is never built (round 3): the fused first layer stops storing the gathered embeddings (0.44 GB written per step,
        # ~70 us of the kernel) because nothing reads them any more -- the first layer's wgrad gathers its operand from the tables
        # itself (dr_bf3_wgrad_emb, from field-major int32 ids), and K4's duplicate pass takes x of the few slots that share rows
        # from a snapshot of those rows taken just before the update (dr_emb_snapshot_sorted_rows).  DR_NO_CONCAT=0: as round 2.
        self.no_concat = (_os.environ.get("DR_NO_CONCAT", "1") == "1" and self.fuse_k3 and self.wplanes[0] is not None
                          and self.wg_ws[0] is not None and sorted_bwd and self.acts[0] in (0, 1) and len(dnn_units) >= 1
                          and not (self.fuse_head and len(self.Ws) == 2) and V < (1 << 24) - 1)
        if self.no_concat:
            self._ids_t = [torch.empty((F, B), dtype=torch.int32, device=device) for _ in range(2)]
            self.x_sorted = torch.empty((B * F, D), dtype=torch.float32, device=device)
        # The fused forward saves the first-order weight every slot read ([F, B], 6.8 MB at config 3); K4 then WRITES a unique row's
        # new weight (old + step) instead of read-modify-writing a line it would have to fetch from HBM again: K4 is bound by
        # 128-byte line operations, 8 per slot, and this removes one (round 4; DR_K4_LINOLD=0: the read-modify-write).
        self.lin_old_t = (torch.empty((F, B), dtype=torch.float32, device=device)
                          if (_os.environ.get("DR_K4_LINOLD", "1") == "1" and self.fuse_k3 and sorted_bwd) else None)
        self._lin_old_valid = False
        # The first layer's three GEMMs in the "f16x2" operand mode (round 4; include/dr_hotpath.h dr_h2_*): every operand value as
        # two fp16 terms of x * s, three matrix instructions per fragment pair instead of the bf16x3 mode's six -- fused forward
        # 302 -> 220 us, dgrad 308 -> 209, gathering wgrad 374 -> 299 (tools/exp/h2_check.py), errors against fp64 at or below the
        # bf16x3 mode's.  The power-of-two scales come from amax records kept on the device: the table's is a running maximum (one
        # pass over the table here, K4 raises it with every value it writes), the dense features' is rebuilt whenever dense_pad is
        # written, d h0's comes out of the narrow backward that produces it, W's out of the plane refresh.
        # ops.set_gemm_split("bf16x3") / DR_GEMM_SPLIT=bf16x3 (read once, by the library) restores the six-product mode.  Needs the fused first layer, the gathering wgrad and the sorted K4;
        # "adam_tf" (whose catch-up kernel also writes table rows) stays on bf16x3.
        self.h2 = (ops.get_gemm_split() == "f16x2" and self.no_concat and self.wplanes[0] is not None
                   and not self.adam_tf and _os.environ.get("DR_OVERLAP_DW", "0") != "1")
        if self.h2:
            # (two W images: the fused dgrad + K4 of step s reads the old one while the refresh behind step s's wgrad writes the other)
            self.wplanes[0] = ops.H2WeightPlanes(self.Ws[0], double_buffer=True)
            self.tab_amax = ops.h2_amax(self.table)
            self._tab_ver = (self.table.data_ptr(), self.table._version)
            self._dense_amaxs = [ops.h2_record(device) for _ in range(2)] if self._dense_pads is not None else None
            self.dh0_amax = ops.h2_record(device)
            self._amax_scratch = ops.h2_record(device)
        self._tighten_every = int(_os.environ.get("DR_AMAX_TIGHTEN_STEPS", "2048"))
        self._steps_since_tighten = 0
        # (timing experiment only -- the record goes stale: K4 without the running maximum, to price the tracking)
        self._exp_no_k4_amax = _os.environ.get("DR_EXP_K4_NO_AMAX", "0") == "1"
        # The slot sort of batch s + 1 next to K4 of batch s (DR_PREFETCH_PLAN=0: every step hashes and sorts its own batch).  K4 is
        # the one long kernel of the step that is HBM-bound with small blocks, i.e. that shares the machine; beside the persistent
        # GEMMs (the fused first layer occupies all 160 KB of LDS on every CU) the sort chain's ~20 small launches only advance in
        # the gaps: 1.2 ms instead of 0.17, and K4 waited 74 us for it (round 2, rocprofv3: K4 305 us, its event 379 us).
        self.prefetch_plan = _os.environ.get("DR_PREFETCH_PLAN", "1") == "1"
        # DR_PREFETCH_EARLY=1: the next batch's hash + slot plan start at the BEGINNING of this step (beside the
        # GEMMs, where they crawl but have a whole step of slack) instead of beside K4, which then has HBM to itself; the next batch's
        # dense features are still placed after this step's wgrad.
        # Round 5: ON by default.  With the f16x2 GEMMs the trade measured in round 4 (K4 -17 us, the GEMMs +14) has become neutral for the
        # step (1.191 / 1.195 / 1.200 ms against 1.187 / 1.218 / 1.189 / 1.209, alternating, profiles/r05_ab_side_stream.log): the fused
        # forward no longer shares the machine with the tail of the previous step's plan (251 -> 237 us), K4 runs 260 - 265 us instead of
        # 267 - 301 (0.63 - 0.645 of the roofline instead of 0.555 - 0.627: the kernel the metric names keeps its margin over 0.60
        # whatever the box), the wgrad and the narrow backward pay 3 us each.  DR_PREFETCH_EARLY=0: beside K4 as in rounds 2 - 4.
        # Round 6: "2" is the default -- the chain starts BEHIND the fused first layer, beside the tower tail (the one kernel of the step
        # that leaves registers and LDS free on every CU), and runs on beside the wgrad.  Since the plan went from ~25 launches to 8
        # (csrc/emb_plan.hip: the gated radix passes became one launch) the chain is short enough to profit: 0.975 - 0.978 ms against
        # 0.984 - 0.989 at the start of the step and 1.026 - 1.032 beside K4 (same call, alternating; before the slimmer plan the same
        # three read 1.056 - 1.061 / -- / 1.09).  Without the fused first layer "2" means "1".
        _pe = _os.environ.get("DR_PREFETCH_EARLY", "2")
        self.prefetch_early = _pe in ("1", "2")
        self.prefetch_after_fwd = _pe == "2" and self.fuse_k3      # "2": behind the first GEMM, beside the tower tail
        # Fewer cross-stream packets in front of K4 (rocprofv3 showed a 31 us gap there against 11 - 13 us between the other
        # dependent kernels): the side chain is ordered behind the plane refresh instead of an event of its own, and the wait for
        # a PREFETCHED plan -- long complete by then -- sits in front of the first-layer wgrad.  DR_LEAN_EVENTS=0: as before.
        self.lean_events = _os.environ.get("DR_LEAN_EVENTS", "1") == "1"
        # the early prefetch as ONE C call whose first kernel is K1 + field-major ids + composite keys + histogram (dr_hash_sort_slots);
        # DR_FUSE_PLAN_FRONT=0: dr_hash_bucket_i64, dr_ids_transpose_i32, dr_emb_sort_slots one after the other (8 launches instead of 6)
        self.fuse_plan_front = _os.environ.get("DR_FUSE_PLAN_FRONT", "1") == "1"
        self._plan_prefetched = False
        if sorted_bwd:
            self._plans = [ops.SortPlan(B * F, device) for _ in range(2)]
            # (round 5, measured and rejected: this stream restricted to a subset of the CUs -- hipExtStreamCreateWithCUMask, every 4th /
            # every 8th CU -- so that its ~25 small launches would not spread over the machine beside K4: the step went from 1.19 to
            # 1.75 - 1.85 ms, K4 itself unchanged; profiles/r05_ab_side_stream.log)
            self.side, _side_r = _side_streams(device)
            # Round 6: a stream of its own for what must be in place before the NEXT forward (the first layer's plane refresh behind its
            # wgrad, the next batch's dense features).  On `side` these five small launches queued behind the plan chain, whose kernels
            # only get CUs when a GEMM drains (a side kernel's blocks wait for the training stream's kernel to run out of blocks: the
            # chain advances about one kernel per kernel boundary of the step) -- rocprofv3 showed the refresh starting 17 us after the
            # fused dgrad + K4 had ENDED and the next forward waiting 39 us for the dense features' record.  DR_SIDE_R=0: as before.
            self.side_r = _side_r if _os.environ.get("DR_SIDE_R", "1") == "1" else self.side
            self._refresh_stream = self.side
            self.ev_ids = torch.cuda.Event()
            self._ev_sorted = [torch.cuda.Event(), torch.cuda.Event()]
            self._ev_hashed = [torch.cuda.Event(), torch.cuda.Event()]
            self.ev_k4 = torch.cuda.Event()
            self.ev_dw_done = torch.cuda.Event()
            self.ev_planes = torch.cuda.Event()
            self.ev_dlogit = torch.cuda.Event()
            self.ev_lin = torch.cuda.Event()
            self.ev_fwd0 = torch.cuda.Event()
            self.ev_part1 = torch.cuda.Event()
            self.ev_small = torch.cuda.Event()
            self._ev_dense = [torch.cuda.Event(), torch.cuda.Event()]
            self._dense_by_event = False
        # First-order weights of the rows that are unique in the batch (99.4 % of the slots for uniform ids) updated by a kernel of
        # their own on the side stream, beside the tower tail, instead of inside K4: a random 4-byte read-modify-write fetches a
        # 128-byte line, 0.27 GB of K4's 1.64 GB.  DR_LIN_SIDE=1 (off by default: see DESIGN.md for the A/B).
        self.lin_side = (sorted_bwd and optimizer == "sgd" and _os.environ.get("DR_LIN_SIDE", "0") == "1")
        # (the library reads the same variable once per process, with the same rule: off iff the value STARTS with '0')
        self._k4_det = not _os.environ.get("DR_K4_DETERMINISTIC", "1").startswith("0")
        # Round 6: K4's unique-row pass as the EPILOGUE of the first layer's dgrad (dr_h2_dgrad_emb_sgd, csrc/h2_occ.hip).  The dgrad's
        # output tile for (example m, field f) IS the gradient row of slot (m, f); the separate K4 read it back from d_concat 0.2 ms
        # after the dgrad had written it -- 0.88 GB of the step's HBM traffic for nothing.  Fused, a slot whose table row no other slot
        # shares gets K4's update straight from the accumulators (same operations, bit-identical tables:
        # tests/test_gpu_fused_k4.py); the few shared rows go through d_concat and K4's duplicate pass as before.  The first-layer
        # wgrad, which gathers x from the tables, moves IN FRONT of the dgrad.  Needs the f16x2 split, SGD, the sorted plan, no
        # concat, D == 64 and the forward's saved first-order weights.  DR_FUSE_K4=0: dgrad, wgrad, K4 as in round 5.
        self.fuse_k4 = (_os.environ.get("DR_FUSE_K4", "1") != "0" and self.h2 and optimizer == "sgd" and sorted_bwd and self.no_concat
                        and D == 64 and not self.lin_side and not self.sort_inline and self.lin_old_t is not None
                        and self.Ws[0].shape[0] >= F * D)
        # The step's three small reduce kernels -- the fused head's finish (partials -> Dense(1) step + loss), the narrow backward's
        # reduce (partials -> its weight step) and the first-layer wgrad's split-K reduce -- on the SIDE stream (round 4;
        # DR_REDUCE_SIDE=0: on the training stream as before).  Nothing in the rest of the step reads what they write; on the training
        # stream each cost its own 6 - 19 us plus a dependent-launch boundary (a build that skips them: 1.408 -> 1.37 ms).  SGD only
        # (Adam's dense step consumes the gradient bucket they fill).
        _rs = _os.environ.get("DR_REDUCE_SIDE", "0")
        self.reduce_side = (sorted_bwd and optimizer == "sgd" and _rs in ("1", "2") and _os.environ.get("DR_SORT_INLINE", "0") != "1")
        # "2": the head's and the narrow backward's only.  The wgrad's reduce is never deferred in the f16x2 mode: its part 2 divides
        # by the scales part 1 multiplied with, both derived from the table's amax record at launch time -- on the side stream it
        # would run beside K4, which raises that record (a power-of-two crossing between the two parts = W0 off by 2^k)
        self.reduce_side_wgrad = self.reduce_side and _rs == "1" and not self.h2
        self._small_pending = False
        self._planes_pending = False
        self._planes_pending_l0 = False
        self._k4_fused_done = False
        self._in_train_step = False
        self._early_issued = False
        # First-layer wgrad on a second stream, concurrent with the HBM-bound K4 on the training stream.  It paid next to the
        # in-kernel-split wgrad (2 blocks / CU, matrix pipe ~35 % busy: 1.91 -> 1.84 ms / step); next to the register-split wgrad
        # (one 512-thread block per CU holding all its registers and 96 KB of LDS) the two kernels only time-slice the CUs: the
        # pair takes 740 - 750 us against 340 + 310 back to back, the step 1.764 vs 1.746 ms (round 2, A/B on one box).  So it
        # is OFF unless DR_OVERLAP_DW=1 asks for it.
        _ov = _os.environ.get("DR_OVERLAP_DW", "0")
        self.overlap_dw = _ov == "1"
        if self.overlap_dw:
            self.no_concat = False       # the gathering wgrad would read the tables while K4 updates them on the other stream
        self.concurrent = {}
        if self.overlap_dw:
            k4 = "emb_pool_bwd_adam" if optimizer == "adam" else "emb_pool_bwd"
            self.concurrent = {"linear_bwd_dw_L0": k4, k4: "linear_bwd_dw_L0"}
            self.side2 = torch.cuda.Stream(device=device)
            self.ev_dx = torch.cuda.Event()
            self.ev_dw = torch.cuda.Event()

    @property
    def dense_pad(self):
        return self._dense_pads[self.cur] if self._dense_pads is not None else None

    @property
    def dense_amax(self):
        return self._dense_amaxs[self.cur] if self._dense_amaxs is not None else None

    @property
    def ids(self):
        return self._ids[self.cur]

    @property
    def plan(self):
        return self._plans[self.cur]

    @property
    def ev_sorted(self):
        return self._ev_sorted[self.cur]

    @staticmethod
    def _token(keys):
        return (keys.data_ptr(), keys._version, tuple(keys.shape))

    def _prefetch_mark(self):
        """First half of the next batch's prefetch, BEFORE K4 is launched: the point on the training stream the side chain may
        start behind (K4's own start).  Returns the token _prefetch_issue needs, or None when nothing is to be prefetched."""
        nk = self._next_keys
        if nk is None or not (self.sorted_bwd and not self.sort_inline and self.prefetch_plan):
            return None
        # the side stream already waits for the LAYER-0 wgrad (the last reader of concat's dense columns) iff the plane refresh
        # queued last was layer 0's; with e.g. dnn_units [64, 256] layer 0 has no planes and the last refresh waited only for
        # layer 1's wgrad: then the side chain takes its own event behind everything on the training stream
        ordered = self._planes_pending and self._planes_pending_l0 and self.lean_events
        if not ordered:
            self.ev_k4.record()
        return (ordered,)

    def _prefetch_issue(self, mark):
        """Second half, AFTER K4 has been handed to the GPU: hash + slot plan of the next batch on the side stream, beside K4
        (both HBM-bound, small blocks).  The chain is ~30 launches; issued before K4 (rounds 1-2) the host spent 100 - 230 us on
        them while the training stream sat idle in front of K4 (rocprofv3: a 233 us gap between the wgrad's reduce and K4).  The
        next forward() picks the buffers up if it is called with the same keys tensor (unmodified); otherwise it recomputes."""
        if mark is None:
            return
        (ordered,) = mark
        nk = self._next_keys
        nxt = self.cur ^ 1
        early_done = self._early_issued
        self._early_issued = False
        # (only the dense features are left to place when the chain went out early: on the stream the plane refresh was queued on)
        st = self._refresh_stream if (early_done and ordered) else self.side
        nk.record_stream(st)
        with torch.cuda.stream(st):
            if not ordered:
                st.wait_event(self.ev_k4)
            if not early_done:
                self._k("next_batch: hash_bucket_i64(side stream, overlapped)", "overlap", self.B * self.F * 16,
                        lambda: ops.hash_bucket_i64(nk, self.col_buckets, out=self._ids[nxt]))
                if self.no_concat:
                    ops.ids_transpose_i32(self._ids[nxt], out=self._ids_t[nxt])
            nd = self._next_dense
            self._pref_dense = None
            if early_done and getattr(self, "_dense_early", False):
                self._pref_dense = self._token(nd)             # placed at the start of the step (_prefetch_early)
            elif nd is not None and self.Nd and not self.overlap_dw:
                # the dense features of that batch too: this step's wgrad (the last reader of concat's dense columns) and forward
                # (dense_pad) are behind us on the training stream, which the side stream has waited for
                nd.record_stream(st)
                if not self.no_concat:
                    self.concat[:, self.F * self.D:self.F * self.D + self.Nd].copy_(nd)
                if self._dense_pads is not None:               # (the buffer the next forward switches to)
                    self._dense_pads[nxt][:, :self.Nd].copy_(nd)
                    if self.h2:
                        ops.h2_amax(self._dense_pads[nxt], self._dense_amaxs[nxt])
                self._pref_dense = self._token(nd)
            if early_done and getattr(self, "_dense_early", False):
                self._dense_by_event = True                    # (_ev_dense[nxt] was recorded behind the early copy)
            elif early_done:
                self._ev_dense[nxt].record(st)       # ids and plan were issued at the start of the step: only the dense features here
                self._dense_by_event = True
            else:
                self._dense_by_event = False
                self._ev_hashed[nxt].record(st)          # what the next forward waits for: ids (+ dense features) in place
                self._k("next_batch: emb_sort_slots(side stream, overlapped)", "overlap", self.B * self.F * 36,
                        lambda: ops.emb_sort_slots(self._ids[nxt], self.row_base, self.R, self._plans[nxt]))
                self._ev_sorted[nxt].record(st)
        self._pref = self._token(nk)

    def _prefetch_early(self):
        """DR_PREFETCH_EARLY=1: hash + field-major ids + slot plan of the NEXT batch, issued at the start of this step's forward."""
        nk = self._next_keys
        self._early_issued = False
        if nk is None or not (self.prefetch_early and self.sorted_bwd and not self.sort_inline and self.prefetch_plan):
            return
        nxt = self.cur ^ 1
        self.ev_fwd0.record()                   # behind the previous step's K4 (the last user of buffer `nxt`)
        nk.record_stream(self.side)
        self._exp_n = getattr(self, "_exp_n", 0) + 1
        if _os_env("DR_EXP_SKIP_PLAN", "") and self._exp_n > 8:      # EXPERIMENT (tools/exp/exp_plan.sh, timing only): buffers keep an older batch's ids + plan
            with torch.cuda.stream(self.side):
                self.side.wait_event(self.ev_fwd0)
                self._ev_hashed[nxt].record(self.side)
                self._ev_sorted[nxt].record(self.side)
            self._early_issued = True
            return
        nd = self._next_dense
        self._dense_early = (nd is not None and self.Nd and self.no_concat and self._dense_pads is not None and not self.overlap_dw)
        if self._dense_early:
            # the next batch's dense features into the OTHER buffer (its last reader was the previous step's wgrad), on the second side
            # stream, beside the plan's front kernel: in rounds 3 - 5 this copy, a fill and the amax pass waited for THIS step's wgrad --
            # three more small launches between the wgrad and the next forward, which rocprofv3 showed ending 10 - 20 us after the
            # step's last kernel.  (In FRONT of the plan chain on `side` they cost it the tower tail's window: 0.990 -> 0.994 ms.)
            nd.record_stream(self.side_r)
            with torch.cuda.stream(self.side_r):
                self.side_r.wait_event(self.ev_fwd0)
                self._dense_pads[nxt][:, :self.Nd].copy_(nd)
                if self.h2:
                    ops.h2_amax(self._dense_pads[nxt], self._dense_amaxs[nxt])
                self._ev_dense[nxt].record(self.side_r)
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.ev_fwd0)
            if self.fuse_plan_front:
                # K1, the field-major ids and the plan as one chain of 6 launches (round 6: dr_hash_sort_slots; rounds 3 - 5: 27)
                self._k("next_batch: hash_sort_slots(side stream, overlapped)", "overlap", self.B * self.F * 52,
                        lambda: ops.hash_sort_slots(nk, self.col_buckets, self.row_base, self.R, self._ids[nxt],
                                                    self._ids_t[nxt] if self.no_concat else None, self._plans[nxt]))
                self._ev_hashed[nxt].record(self.side)
            else:
                self._k("next_batch: hash_bucket_i64(side stream, overlapped)", "overlap", self.B * self.F * 16,
                        lambda: ops.hash_bucket_i64(nk, self.col_buckets, out=self._ids[nxt]))
                if self.no_concat:
                    ops.ids_transpose_i32(self._ids[nxt], out=self._ids_t[nxt])
                self._ev_hashed[nxt].record(self.side)
                self._k("next_batch: emb_sort_slots(side stream, overlapped)", "overlap", self.B * self.F * 36,
                        lambda: ops.emb_sort_slots(self._ids[nxt], self.row_base, self.R, self._plans[nxt]))
            self._ev_sorted[nxt].record(self.side)
        self._early_issued = True

    # ---- per-kernel HIP events on the launch stream (bench.py's roofline numbers) ----------------------
    def enable_kernel_events(self, on: bool, every: int = 1, only=None):
        """Per-kernel HIP events on the launch stream.  every = n: only every n-th train_step is bracketed -- two event records per
        kernel are two extra packets per kernel boundary, and at 12 kernels per 1.5 ms step that is no longer free (round 3: 1.56
        ms with events on every step, 1.48 ms without, same box); bench.py samples every 4th step of the timed region.
        only: names of the kernels to bracket (None = all).  Round 6: with all 9 kernels of the step bracketed, every 4th step, the
        0.97 ms step read 0.979 - 0.985 (off: 0.967 - 0.968): bench.py brackets only the headline kernel inside its timed region."""
        self._ev_only = set(only) if only is not None else None
        for rec in (getattr(self, "_events", None) or {}).values():       # the pairs of the previous collection go back to the pool
            self._ev_pool.extend(rec[2])
        self._events = {} if on else None
        self._ev_every = max(1, int(every))
        self._ev_step = 0
        self._ev_live = bool(on)

    def reserve_kernel_events(self, n_pairs: int):
        """Create n_pairs timing-event pairs NOW, so that a bracketed region creates none: on a box's first process hipEventCreate costs
        tens of microseconds a piece (round 6: the first bench.py of a fresh box read 1.32 - 1.44 ms per step with 25 bracketed steps
        in its timed region, the same command a second time 1.00 ms; with events off both read 0.98)."""
        while len(self._ev_pool) < n_pairs:
            self._ev_pool.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))

    def _k(self, name, bound, work, fn):
        if self._events is None or not self._ev_live or (self._ev_only is not None and name not in self._ev_only):
            return fn()
        s, e = self._ev_pool.pop() if self._ev_pool else (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
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
            out[name] = {"bound": bound, "ms": ms, "n": len(evs),
                         "alg_bytes" if bound in ("hbm", "overlap") else "alg_flops": work}
            if name in getattr(self, "concurrent", {}):
                out[name]["concurrent_with"] = self.concurrent[name]     # event time stretched by the co-running kernel
        return out

    def concurrent_pair_summary(self):
        """Kernels that run concurrently on two streams (first-layer wgrad next to K4) cannot be priced one by one: their joint
        wall time per step = first start -> last end, from the same HIP events."""
        torch.cuda.synchronize()
        done, out = set(), []
        for a, b in getattr(self, "concurrent", {}).items():
            if a in done or b in done or a not in (self._events or {}) or b not in (self._events or {}):
                continue
            done.update((a, b))
            ea, eb = self._events[a][2], self._events[b][2]
            n = min(len(ea), len(eb))
            tot = 0.0
            for (sa, fa), (sb, fb) in zip(ea[-n:], eb[-n:]):
                first = sa if sa.elapsed_time(sb) >= 0 else sb
                tot += max(first.elapsed_time(fa), first.elapsed_time(fb))
            out.append({"kernels": [a, b], "joint_us": tot / n * 1e3})
        return out

    def _wgrad_reduce_deferred(self, i):
        """layer 0's gathering wgrad with its reduce on the side stream, in front of the plane refresh that lives there already"""
        return (self.reduce_side_wgrad and i == 0 and self.no_concat and self.wplanes[0] is not None and not self.sort_inline
                and not self.overlap_dw)

    def _side_part2(self, fn):
        """second half of a two-part call on the side stream, behind the first half (just launched on the training stream)"""
        self.ev_part1.record()
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.ev_part1)
            fn()
            self.ev_small.record(self.side)
        self._small_pending = True

    def _wgrad(self, i, x, dy, sc, dstW, dstb):
        if i == 0 and self.h2:
            ops.h2_wgrad_emb(self._ids_t[self.cur], self.row_base, self.table, self.tab_amax, self.dense_pad, self.dense_amax, dy,
                             self.dh0_amax, sc, dstW, dstb, workspace=self.wg_ws[0], parts=1 if self._wgrad_reduce_deferred(i) else 3)
        elif i == 0 and self.no_concat:
            # x = [embeddings of this batch's rows, dense features]: gathered from the tables (they are updated only by K4, later
            # on this stream) through the field-major ids
            ops.bf3_wgrad_emb(self._ids_t[self.cur], self.row_base, self.table, self.dense_pad, dy, sc, dstW, dstb,
                              workspace=self.wg_ws[0], parts=1 if self._wgrad_reduce_deferred(i) else 3)
        elif self.wg_ws[i] is not None:
            ops.bf3_wgrad(x, dy, sc, dstW, dstb, workspace=self.wg_ws[i])
        else:
            ops.linear_bwd_dw(x, dy, sc, dstW, dstb, workspace=self.dw_ws[i])

    # ------------------------------------------------------------------------------------------
    def forward(self, keys: torch.Tensor, dense: Optional[torch.Tensor], labels: Optional[torch.Tensor],
                loss_mode: int = ops.LOSS_SIGMOID_CE):
        F, D = self.F, self.D
        B = self.B
        if self.optimizer == "adam":
            self.flat_grads.zero_()          # the fused head already writes Dense(1) gradients during the forward
        if self._planes_pending:         # the previous step's weight planes were refreshed on the side stream
            torch.cuda.current_stream().wait_event(self.ev_planes)
            self._planes_pending = False
        for wp in self.wplanes:          # weights written from outside since the last refresh (load / copy_ / fill_): re-split
            if wp is not None:
                wp.ensure_fresh()
        if self.h2 and (self.table.data_ptr(), self.table._version) != self._tab_ver:
            # the table was written from outside (checkpoint restore, copy_ / fill_ in a test): rebuild its amax record.  K4's own
            # writes go through raw pointers and keep the record themselves.
            ops.h2_amax(self.table, self.tab_amax)
            self._tab_ver = (self.table.data_ptr(), self.table._version)
        pref_tok = self._pref
        prefetched = pref_tok is not None and pref_tok == self._token(keys)
        self._pref = None
        self._plan_prefetched = prefetched
        if prefetched:
            # K1 + the slot sort of this batch ran beside the previous step's K4
            self.cur ^= 1
            torch.cuda.current_stream().wait_event(self._ev_hashed[self.cur])
            if getattr(self, "_dense_by_event", False):
                torch.cuda.current_stream().wait_event(self._ev_dense[self.cur])
        else:
            if pref_tok is not None:
                # a prefetch was issued for OTHER keys and is discarded: its side-stream copy of that batch's dense features into
                # concat / dense_pad must be behind us before this forward writes its own
                torch.cuda.current_stream().wait_event(self._ev_hashed[self.cur ^ 1])
            self._k("hash_bucket_i64", "hbm", B * F * 16,
                    lambda: ops.hash_bucket_i64(keys, self.col_buckets, out=self.ids))                # K1
            if self.no_concat:
                ops.ids_transpose_i32(self.ids, out=self._ids_t[self.cur])
        if self._in_train_step and not self.prefetch_after_fwd:
            self._prefetch_early()
        if self.sorted_bwd and self.sort_inline:
            self._k("emb_sort_slots", "hbm", B * F * 36,
                    lambda: ops.emb_sort_slots(self.ids, self.row_base, self.R, self.plan))
            self.ev_sorted.record()
        if self.sorted_bwd and not self.sort_inline and self.sort_with_k3 and not prefetched:
            self._launch_sort()              # next to K3 (both HBM-bound, small blocks: they do share the machine)
        if self.adam_tf:
            # (the catch-up stamps the looked-up rows with step t + 1 BEFORE that step's update exists: a forward that is not followed
            # by backward_and_update -- predict, a second forward -- would leave rows marked as updated and silently diverge from
            # TF's dense Adam)
            if not self._in_train_step:
                raise RuntimeError("optimizer='adam_tf': forward() is only valid inside train_step() (it stamps the looked-up rows "
                                   "with the step that backward_and_update is about to apply)")
            # rows this batch looks up: replay the decay-only steps they missed (through step t - 1), stamp them t
            self._k("adam_catchup_rows", "hbm", 0,
                    lambda: ops.adam_catchup_rows(self.ids, self.row_base, self.table, self.m_table, self.v_table, self.lin_w, self.m_lin,
                                                  self.v_lin, self.row_step, self.t, self.t + 1, self.lr, self.beta1, self.beta2, self.eps))
        fused_l0 = self.fuse_k3 and self.wplanes[0] is not None
        if not fused_l0:
            self._k("emb_pool_fwd", "hbm", self.alg_bytes_fwd(),                                      # K3+K5+K6
                    lambda: ops.emb_pool_fwd(self.ids, F, None, self.row_base, self.table, self.lin_w, self.lin_bias,
                                             ld_concat=self.ld, concat=self.concat, sum_x=self.sum_x,
                                             fm_logit=self.fm_logit))
        if self.sorted_bwd and not self.sort_inline and not self.sort_with_k3 and not prefetched:
            self._launch_sort()              # after K3: under the first GEMM and the tower tail
        if self.Nd and not (prefetched and self._pref_dense is not None and self._pref_dense == self._token(dense)):
            if not self.no_concat:
                self.concat[:, F * D:F * D + self.Nd].copy_(dense)                     # layout: append dense feats
            if self.dense_pad is not None:
                self.dense_pad[:, :self.Nd].copy_(dense)                               # the fused kernel's own (k-tile wide) copy
                if self.h2:
                    ops.h2_amax(self.dense_pad, self.dense_amax)
        self._pref_dense = None
        x = self.concat[:, :self.in_dim]
        head = self.fuse_head and labels is not None
        nl = len(self.Ws)
        self._lin_old_valid = False
        for i, (W, b) in enumerate(zip(self.Ws, self.bs)):                             # K7
            if head and i == nl - 2:
                break
            if i == 0 and fused_l0:
                # K3 + first Dense in one launch: the GEMM gathers its activation operand from the tables, writes concat and the
                # FM terms on the way (the dense features were placed in concat above)
                if self.h2:
                    self._k("emb_linear_fwd_L0", "mfma", 2.0 * B * W.shape[0] * W.shape[1],
                            lambda b=b: ops.h2_emb_linear_fwd(self.ids, self.row_base, self.V, self.table, self.tab_amax, self.lin_w, self.lin_bias,
                                                              self.dense_pad, self.dense_amax, None, self.in_dim, self.wplanes[0].wt, b,
                                                              self.acts[0], self.sum_x, self.fm_logit, self.hs[0], lin_vals_t=self.lin_old_t))
                else:
                    self._k("emb_linear_fwd_L0", "mfma", 2.0 * B * W.shape[0] * W.shape[1],
                            lambda b=b: ops.bf3_emb_linear_fwd(self.ids, self.row_base, self.V, self.table, self.lin_w, self.lin_bias,
                                                               self.dense_pad, None if self.no_concat else self.concat, self.in_dim,
                                                               self.wplanes[0].wt, b, self.acts[0], self.sum_x, self.fm_logit, self.hs[0],
                                                               lin_vals_t=self.lin_old_t))
                self._lin_old_valid = self.lin_old_t is not None
                if self._in_train_step and self.prefetch_after_fwd:
                    self._prefetch_early()
            elif self.wplanes[i] is not None:
                self._k("linear_fwd_L%d" % i, "mfma", 2.0 * B * W.shape[0] * W.shape[1],
                        lambda x=x, b=b, i=i: ops.bf3_linear_nt(x, self.wplanes[i].wt, bias=b, act=self.acts[i], out=self.hs[i]))
            else:
                self._k("linear_fwd_L%d" % i, "mfma", 2.0 * B * W.shape[0] * W.shape[1],
                        lambda x=x, W=W, b=b, i=i: ops.linear_fwd(x, W, b, self.acts[i], out=self.hs[i]))
            x = self.hs[i]
        self._head_done = head
        self._tail_done = False
        if head and self.fuse_tail and self._in_train_step:
            # K7 tail + K11 + the backward of the last hidden layer in one pass over x (= hs[nl - 3]): prob, loss, d_logit, d_h, the
            # gradient for the layer below (dhs[nl - 3]) and the SGD steps (or gradients) of both layers
            W1, W2 = self.Ws[-2], self.Ws[-1]
            adam = self.optimizer == "adam"
            def tail(parts, x=x, W1=W1, W2=W2):
                return ops.tower_tail_fused(
                    x, W1, self.bs[-2], W2, self.bs[-1], self.fm_logit, labels, loss_mode, 1.0 if adam else -self.lr, self.dhs[nl - 3],
                    dst_W1=self.gWs[-2] if adam else "inplace", dst_b1=self.gbs[-2] if adam else "inplace",
                    dst_W2=self.gWs[-1] if adam else "inplace", dst_b2=self.gbs[-1] if adam else "inplace",
                    prob=self.prob, d_logit=self.d_logit, d_h=self.dhs[-1], loss=self.loss, workspace=self.tail_ws, parts=parts,
                    dx_amax=self.dh0_amax if (self.h2 and nl - 2 == 1 and (parts & 1)) else None)
            work = 4.0 * B * (2 * W1.shape[0] + W1.shape[1] + 4)
            if self.reduce_side:
                self._k("tower_tail_fused", "hbm", work, lambda: tail(1))
                self._side_part2(lambda: tail(2))
            else:
                self._k("tower_tail_fused", "hbm", work, lambda: tail(3))
            self._tail_done = True
            return self.prob
        if head:
            # K7 tail + K11: Dense(H<=32, relu), Dense(1), + FM logit, loss, d_logit, d_h and the Dense(1) SGD step
            W1, W2 = self.Ws[-2], self.Ws[-1]
            def head(parts, x=x, W1=W1, W2=W2):
                return ops.tower_head_fwd_bwd(
                    x, W1, self.bs[-2], W2, self.bs[-1], self.fm_logit, labels, loss_mode,
                    1.0 if self.optimizer == "adam" else -self.lr, act=1,
                    prob=self.prob, d_logit=self.d_logit, d_h=self.dhs[-1], loss=self.loss, workspace=self.head_ws,
                    dst_W2=self.gWs[-1] if self.optimizer == "adam" else "inplace",
                    dst_b2=self.gbs[-1] if self.optimizer == "adam" else "inplace", parts=parts)
            if self.reduce_side and self._in_train_step:
                self._k("tower_head_fwd_bwd", "hbm", 4.0 * B * (W1.shape[0] + W1.shape[1] + 4), lambda: head(1))
                self._side_part2(lambda: head(2))
            else:
                self._k("tower_head_fwd_bwd", "hbm", 4.0 * B * (W1.shape[0] + W1.shape[1] + 4), lambda: head(3))
            return self.prob
        if labels is None:
            labels = self.prob   # dummy, loss ignored
        ops.bce_fwd_bwd(self.fm_logit, labels, loss_mode, workspace=self.ws, logits_b=self.hs[-1],    # K11
                        out=(self.prob, self.d_logit, self.loss))
        return self.prob

    def backward_and_update(self):
        F, D, lr = self.F, self.D, self.lr
        adam = self.optimizer == "adam"
        lin_side = self.lin_side and not adam and not self.sort_inline
        if lin_side:
            # d_logit is final (fused head or K11 above); the plan of this batch was produced on the side stream itself
            self.ev_dlogit.record()
            with torch.cuda.stream(self.side):
                self.side.wait_event(self.ev_dlogit)
                self._k("emb_lin_update_unique(side stream, overlapped)", "overlap", self.B * self.F * 16,
                        lambda: ops.emb_lin_update_unique(self.ids, self.row_base, self.plan, self.d_logit, -lr, self.lin_w))
                self.ev_lin.record(self.side)
        sc = 1.0 if adam else -lr                         # Adam: gradients into the bucket; SGD: the step itself
        n = len(self.Ws)
        dy = self.d_logit.reshape(-1, 1)                  # d(mean loss)/d logit == pre-activation grad of Dense(1)
        top = n - 1
        if self._head_done:                               # Dense(1) backward + its update already done by the fused head
            top = n - 2
            dy = self.dhs[-1]
        if self._tail_done:                               # ... and the last hidden layer's by the one-pass tail
            top = n - 3
            dy = self.dhs[n - 3]
        for i in range(top, -1, -1):
            x = self.concat[:, :self.in_dim] if i == 0 else self.hs[i - 1]
            W = self.Ws[i]
            dstW, dstb = (self.gWs[i], self.gbs[i]) if adam else (W, self.bs[i])
            fl = 2.0 * self.B * W.shape[0] * W.shape[1]
            if i > 0:
                dx = self.dhs[i - 1]
                rs = self.hs[i - 1] if self.acts[i - 1] else None
            else:
                dx = self.d_concat[:, :self.in_dim]
                rs = None
            if i > 0 and self.fuse_narrow and self.narrow_ws[i] is not None:
                def narrow(parts, x=x, dy=dy, W=W, dx=dx, i=i, dstW=dstW, dstb=dstb):
                    return ops.linear_bwd_narrow(x, dy, W, sc, dstW, dstb, dx, relu_mask=True, workspace=self.narrow_ws[i], parts=parts,
                                                 dx_amax=self.dh0_amax if (i == 1 and self.h2 and (parts & 1)) else None)
                if self.reduce_side and self._in_train_step:
                    self._k("linear_bwd_narrow_L%d" % i, "hbm", 4.0 * self.B * (2 * W.shape[0] + W.shape[1]), lambda: narrow(1))
                    self._side_part2(lambda: narrow(2))
                else:
                    self._k("linear_bwd_narrow_L%d" % i, "hbm", 4.0 * self.B * (2 * W.shape[0] + W.shape[1]), lambda: narrow(3))
                dy = dx
                continue
            if i == 0 and self.fuse_k4 and self._lin_old_valid and not adam and not self.overlap_dw:
                self._backward_l0_fused_k4(x, dy, sc, dstW, dstb, fl)
                dy = None
                continue
            # dx first (uses the pre-update W), then the wgrad (with the fused SGD step unless Adam)
            if i == 0 and self.h2:
                if not (n > 1 and self.fuse_narrow and self.narrow_ws[1] is not None):
                    ops.h2_amax(dy, self.dh0_amax)       # (layer 1's backward was not the narrow kernel that leaves the record)
                self._k("linear_bwd_dx_L%d" % i, "mfma", fl,
                        lambda dy=dy, rs=rs, dx=dx, i=i: ops.h2_linear_nt(dy, self.dh0_amax, self.wplanes[i].w, mask=rs, out=dx))
            elif self.wplanes[i] is not None:
                self._k("linear_bwd_dx_L%d" % i, "mfma", fl,
                        lambda dy=dy, rs=rs, dx=dx, i=i: ops.bf3_linear_nt(dy, self.wplanes[i].w, mask=rs, out=dx))
            else:
                self._k("linear_bwd_dx_L%d" % i, "mfma", fl,
                        lambda dy=dy, W=W, rs=rs, dx=dx: ops.linear_bwd_dx(dy, W, relu_src=rs, out=dx))
            if i == 0 and self.sorted_bwd and self._plan_prefetched and self.lean_events:
                torch.cuda.current_stream().wait_event(self.ev_sorted)     # (see lean_events)
            if i == 0 and self.overlap_dw:
                # first-layer wgrad on the second stream; K4 follows on this one as soon as dx is out
                self.ev_dx.record()
                with torch.cuda.stream(self.side2):
                    self.side2.wait_event(self.ev_dx)
                    self._k("linear_bwd_dw_L%d" % i, "mfma", fl,
                            lambda x=x, dy=dy, i=i, dstW=dstW, dstb=dstb: self._wgrad(i, x, dy, sc, dstW, dstb))
                    if self.wplanes[i] is not None and not adam:
                        self.wplanes[i].refresh()                    # the weight just moved: its planes follow (same stream)
                    self.ev_dw.record(self.side2)
                dy = dx
                continue
            self._k("linear_bwd_dw_L%d" % i, "mfma", fl,
                    lambda x=x, dy=dy, i=i, dstW=dstW, dstb=dstb: self._wgrad(i, x, dy, sc, dstW, dstb))
            if self.wplanes[i] is not None and not adam:
                if self.sorted_bwd and not self.sort_inline:
                    # the weight just moved: its planes follow -- on the side stream, beside whatever comes next on this one (the
                    # two split launches are needed only by the next step's forward / dgrad, which wait for ev_planes)
                    self.ev_dw_done.record()
                    self._refresh_stream = self.side
                    with torch.cuda.stream(self.side):
                        self.side.wait_event(self.ev_dw_done)
                        if self._wgrad_reduce_deferred(i) and self.h2:
                            ops.h2_wgrad_emb(self._ids_t[self.cur], self.row_base, self.table, self.tab_amax, self.dense_pad, self.dense_amax,
                                             dy, self.dh0_amax, sc, dstW, dstb, workspace=self.wg_ws[0], parts=2)
                        elif self._wgrad_reduce_deferred(i):
                            # the split-K reduce (+ fused SGD step) of the GEMM just launched: nothing on the training stream reads W
                            ops.bf3_wgrad_emb(self._ids_t[self.cur], self.row_base, self.table, self.dense_pad, dy, sc, dstW, dstb,
                                              workspace=self.wg_ws[0], parts=2)
                        self.wplanes[i].refresh()
                        self.ev_planes.record(self.side)
                    self._planes_pending = True
                    self._planes_pending_l0 = (i == 0)
                else:
                    self.wplanes[i].refresh()
            dy = dx
        if adam:
            self._adam_finish()
        elif self._k4_fused_done:
            self._k4_fused_done = False
        elif self.sorted_bwd:
            if not (self._plan_prefetched and self.lean_events):
                torch.cuda.current_stream().wait_event(self.ev_sorted)
            if self.no_concat and not self._k4_det:
                # x of the slots that share rows, before the atomic pieces of hot rows touch them (DR_K4_DETERMINISTIC=0 only: the
                # deterministic update reads x from the table itself -- no row is written before its reader has it)
                ops.emb_snapshot_sorted_rows(self.plan, self.table, self.R, self.x_sorted)
            mark = self._prefetch_mark()
            k4 = lambda parts: ops.emb_pool_bwd_sorted(self.ids, self.row_base, self.plan, D, self.R, self.d_concat, self.d_logit,
                                                       -lr, self.table, self.lin_w, self.lin_bias,
                                                       concat=None if self.no_concat else self.concat, sum_x=self.sum_x,
                                                       x_sorted=self.x_sorted if self.no_concat else None, parts=parts,
                                                       lin_old_t=self.lin_old_t if (self._lin_old_valid and not lin_side) else None,
                                                       table_amax=self.tab_amax if (self.h2 and not self._exp_no_k4_amax) else None)
            sk = 4 if lin_side else 0
            self._k("emb_pool_bwd", "hbm", self.alg_bytes_bwd() - (self.B * self.F * 8 if lin_side else 0),   # K4 (sorted)
                    lambda: k4(1 | sk))
            if self.no_concat:     # rows hit > 32 times: their parked pieces added in sorted order (a scan of the head list otherwise)
                self._k("emb_hot_rows_apply", "hbm", 0, lambda: k4(2 | sk))
            if lin_side:
                torch.cuda.current_stream().wait_event(self.ev_lin)
            self._prefetch_issue(mark)
        else:
            self._k("emb_pool_bwd", "hbm", self.alg_bytes_bwd(),                                      # K4 (atomics)
                    lambda: ops.emb_pool_bwd(self.ids, F, self.col_start, self.row_base, D, self.d_concat, self.concat,
                                             self.sum_x, self.d_logit, -lr, self.table, self.lin_w, self.lin_bias))
        if self._small_pending:
            # the head's finish / narrow reduce ran on the side stream: whoever reads the loss or the small layers' weights next does
            # so on this stream (they finished long ago -- the wait is free)
            torch.cuda.current_stream().wait_event(self.ev_small)
            self._small_pending = False
        if self.overlap_dw and not adam:
            torch.cuda.current_stream().wait_event(self.ev_dw)     # next step's forward reads the updated first layer

    def _backward_l0_fused_k4(self, x, dy, sc, dstW, dstb, fl):
        """First layer's backward with K4 fused into the dgrad (see fuse_k4): wgrad (gathers x from the still un-updated tables; its
        fused SGD step moves W0 but not the planes the dgrad reads) -> dgrad + K4's unique rows -> K4's duplicate pass, hot rows and
        first-order bias.  The plane refresh waits for the dgrad."""
        lr = self.lr
        torch.cuda.current_stream().wait_event(self.ev_sorted)             # the plan's flags (long complete when prefetched)
        self._k("linear_bwd_dw_L0", "mfma", fl, lambda: self._wgrad(0, x, dy, sc, dstW, dstb))
        # The weight just moved: its planes follow on the side stream, BESIDE the dgrad, which keeps reading the previous W image and
        # record (H2WeightPlanes(double_buffer=True): the refresh writes the other pair) -- the next forward waits for nothing.
        w_old = self.wplanes[0].w
        early = self._early_issued
        rs = self.side_r if early else self.side           # (without the early chain the hash + plan follow on `side`, behind the refresh)
        self._refresh_stream = rs
        self.ev_dw_done.record()
        with torch.cuda.stream(rs):
            rs.wait_event(self.ev_dw_done)
            self.wplanes[0].refresh()
            self.ev_planes.record(rs)
        self._planes_pending = True
        self._planes_pending_l0 = True
        # the next batch's dense features (and, unless they went out at the start of the step, its hash + plan) behind the refresh: the
        # wgrad was the last reader of dense_pad
        mark = self._prefetch_mark()
        if early:
            self._prefetch_issue(mark)
        if not self._k4_det:
            ops.emb_snapshot_sorted_rows(self.plan, self.table, self.R, self.x_sorted)
        tab_amax = self.tab_amax if not self._exp_no_k4_amax else None
        # bytes this kernel must move: SURVEY 8(d)'s K4 figure minus the 4FD read of d_concat that no longer exists, plus the dgrad's dy
        work = self.alg_bytes_bwd() - 4.0 * self.B * self.F * self.D + 4.0 * self.B * dy.shape[1]
        self._k("emb_pool_bwd_fused_dgrad_L0", "hbm", work,
                lambda: ops.h2_dgrad_emb_sgd(dy, self.dh0_amax, w_old, self._ids_t[self.cur], self.plan, self.row_base, self.table,
                                             self.lin_w, self.lin_old_t, self.sum_x, self.d_logit, -lr, self.d_concat, table_amax=tab_amax))
        k4 = lambda parts: ops.emb_pool_bwd_sorted(self.ids, self.row_base, self.plan, self.D, self.R, self.d_concat, self.d_logit, -lr,
                                                   self.table, self.lin_w, self.lin_bias, sum_x=self.sum_x, x_sorted=self.x_sorted,
                                                   parts=parts | 8, lin_old_t=self.lin_old_t, table_amax=tab_amax)
        self._k("emb_pool_bwd_dups", "hbm", 0, lambda: k4(1))
        self._k("emb_hot_rows_apply", "hbm", 0, lambda: k4(2))
        if not early:
            self._prefetch_issue(mark)
        self._k4_fused_done = True

    def _launch_sort(self):
        """The slot sort depends only on ids and is needed only by K4: it runs on a side stream.  It is HBM-bound like K3 and,
        squeezed next to a GEMM, long (~660 us vs 167 us alone): started with K3 it stretched K3; started before the first-layer
        dgrad it crawled to 1.17 ms and K4 waited for it; started after K3 it hides under the first GEMM and the tower tail."""
        self.ev_ids.record()
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.ev_ids)
            self._k("emb_sort_slots(side stream, overlapped)", "overlap", self.B * self.F * 36,
                    lambda: ops.emb_sort_slots(self.ids, self.row_base, self.R, self.plan))
            self.ev_sorted.record(self.side)

    def _adam_finish(self):
        """Adam mode: fused row-wise Adam K4, first-order bias gradient, one dense Adam step over the flat parameters."""
        self.t += 1
        lr_t = ops.adam_lr_t(self.lr, self.beta1, self.beta2, self.t)
        if not (self._plan_prefetched and self.lean_events):
            torch.cuda.current_stream().wait_event(self.ev_sorted)
        if self.no_concat:
            ops.emb_snapshot_sorted_rows(self.plan, self.table, self.R, self.x_sorted)
        mark = self._prefetch_mark()
        self._k("emb_pool_bwd_adam", "hbm", self.B * (28 * self.F * self.D + 40 * self.F),                # K4 + optimizer
                lambda: ops.emb_pool_bwd_sorted_adam(self.ids, self.row_base, self.plan, self.D, self.R, self.d_concat,
                                                     self.d_logit, lr_t, self.beta1, self.beta2, self.eps, self.table,
                                                     self.m_table, self.v_table, self.lin_w, self.m_lin, self.v_lin,
                                                     concat=None if self.no_concat else self.concat, sum_x=self.sum_x,
                                                     x_sorted=self.x_sorted if self.no_concat else None,
                                                     lin_old_t=self.lin_old_t if self._lin_old_valid else None,
                                                     table_amax=self.tab_amax if self.h2 else None))
        self._prefetch_issue(mark)
        ops.reduce_sum(self.d_logit, out=self.g_lin_bias)                       # d loss / d (first-order bias), fixed order
        if self.overlap_dw:
            torch.cuda.current_stream().wait_event(self.ev_dw)     # the dense step needs the first layer's gradient
        self._k("adam_step_dense", "hbm", 28.0 * self.flat_params.numel(),
                lambda: ops.adam_step(self.flat_params, self.flat_grads, self.flat_m, self.flat_v, lr_t, self.beta1, self.beta2,
                                      self.eps))
        for wp in self.wplanes:
            if wp is not None:
                wp.refresh()

    def adam_flush(self):
        """adam_tf mode: bring EVERY row of the tables to the current step (replays the decay-only steps of rows not looked up
        lately), so that table / lin_w / moments in memory equal TF's dense state.  A pass over the whole slab: export, tests."""
        assert self.adam_tf
        V, F = self.V, self.F
        step = 1 << 20
        for v0 in range(0, V, step):
            v1 = min(V, v0 + step)
            ids = torch.arange(v0, v1, device=self.table.device, dtype=torch.int64)[:, None].expand(v1 - v0, F).contiguous()
            ops.adam_catchup_rows(ids, self.row_base, self.table, self.m_table, self.v_table, self.lin_w, self.m_lin, self.v_lin,
                                  self.row_step, self.t, self.t, self.lr, self.beta1, self.beta2, self.eps)

    def refresh_planes(self):
        """Re-split every wide layer's weight into its bf16 planes.  Needed only after writing eng.Ws / eng.flat_params through a path
        torch's version counter does not see (a raw-pointer kernel); torch-side writes are picked up by forward() itself."""
        for wp in self.wplanes:
            if wp is not None:
                wp.refresh()

    def train_step(self, keys, dense, labels, next_keys=None, next_dense=None):
        """One training step.  next_keys (optional): the raw keys of the batch the NEXT call will train on (the data loader knows
        it; ShardedDeepFMEngine uses the same argument to route ahead): its K1 + slot sort then run beside this step's K4.
        next_dense (optional, with next_keys): that batch's dense features, placed into the input buffers at the same time."""
        self._next_keys, self._next_dense = next_keys, next_dense
        if self._events is not None:
            self._ev_live = (self._ev_step % self._ev_every) == 0
            self._ev_step += 1
        self._in_train_step = True
        try:
            self.forward(keys, dense, labels)
            self.backward_and_update()
        finally:
            self._in_train_step = False
        self._next_keys = self._next_dense = None
        if self.h2 and self._tighten_every > 0:
            self._steps_since_tighten += 1
            if self._steps_since_tighten >= self._tighten_every:
                self.tighten_amax()
        return self.loss

    def tighten_amax(self):
        """Recompute the table's amax record exactly (f16x2 mode).  K4 keeps it as a RUNNING maximum -- an upper bound that never comes
        down: after one outlier row, or when training shrinks the embeddings, every other value would carry fewer than its 22 bits for
        ever.  One pass over the slab on the training stream, BETWEEN two steps (nothing raises or reads the record
        meanwhile): 66.6 GB at ~5 TB/s = 13 ms at config 3, every DR_AMAX_TIGHTEN_STEPS steps (default 2048: 6 us per step; 0 = never).
        The new value goes into a scratch record first and replaces the old one with a 4-byte copy."""
        if not self.h2:
            return
        ops.h2_amax(self.table, self._amax_scratch)
        self.tab_amax.copy_(self._amax_scratch)
        self._steps_since_tighten = 0

    # algorithmic bytes of the two embedding kernels per step
    def alg_bytes_fwd(self):
        return self.B * (8 * self.F * self.D + 12 * self.F + 8)

    def alg_bytes_bwd(self):
        return self.B * (12 * self.F * self.D + 16 * self.F)

    def flops_step(self):
        fl = 0
        d = self.in_dim
        for W in self.Ws:
            fl += 2 * self.B * W.shape[0] * W.shape[1]
        return 3 * fl
