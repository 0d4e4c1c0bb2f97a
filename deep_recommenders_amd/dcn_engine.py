"""Lean DCN training step (BASELINE.json config 4: Criteo-shaped, 3 cross layers + MLP [1024, 512, 256] + Dense(1)):

    x0 = concat(field embeddings [, dense features])                        K1 hash -> K3 gather+pool
    x_{l+1} = x0 * (x_l W_l + b_l + diag * x_l) + x_l,  l = 0..L-1           K8  keras/models/ranking/dcn.py:70-88 of the reference
    logit = Dense(1)(relu-MLP(x_L))                                          K7
    loss = sigmoid cross-entropy                                             K11

stacked exactly like the reference's own test stacks `Cross` layers (tests/keras/test_dcn.py:27-32), with the same kernels the
`Cross` / `dnn` classes use, no autograd tape, every update fused into the producing kernel (SGD: dst += -lr * grad).
The reference ships the layer, not a DCN model class: this engine exists to put an end-to-end number on config 4."""
import math
from typing import List, Sequence

import torch

from . import ops


def _pad4(n):
    return (n + 3) // 4 * 4


class DCNDense:
    """The dense half of a DCN step -- cross stack, MLP, loss, and their backward -- on given weights, shared by DCNEngine (one GPU) and
    sharded.ShardedDCNEngine (row-sharded tables, replicated weights): x0 in, d loss / d x0 out.

    Gradients: `grads` = None applies the SGD step inside the wgrad kernels (dst += -lr * grad on the weights themselves, planes re-split
    right behind each update); `grads` = (g_cross_W, g_cross_b, g_Ws, g_bs) accumulates dst += grad into those tensors instead (a
    data-parallel bucket the caller all-reduces and applies; it then calls refresh_planes()).

    Wide GEMMs run on pre-split weights (ops.planes_worthwhile) in the operand split the library reports at construction
    (ops.get_gemm_split(): "f16x2" -- three matrix instructions per fragment pair, every activation operand with its amax record: x0's
    from one dr_h2_amax pass, every later one from the epilogue of the kernel that produces it -- or "bf16x3")."""

    def __init__(self, cross_W, cross_b, Ws, bs, batch, in_dim, ld, diag_scale, device, grads=None, k=None):
        import os as _os
        self.cross_W, self.cross_b, self.Ws, self.bs = list(cross_W), list(cross_b), list(Ws), list(bs)
        self.B, self.in_dim, self.ld, self.diag = batch, in_dim, ld, diag_scale
        self.grads = grads
        self._k = k if k is not None else (lambda name, bound, work, fn: fn())
        B = batch
        units = [W.shape[1] for W in self.Ws]
        self.acts = [1] * (len(units) - 1) + [0]
        f32 = dict(dtype=torch.float32, device=device)
        on = _os.environ.get("DR_PLANES", "1") == "1"
        wide_cross = on and ops.planes_worthwhile(B, in_dim, in_dim)
        self.h2 = wide_cross and ops.get_gemm_split() == "f16x2"
        self.h2_all_wide = self.h2                               # (bench.py: every planes GEMM of this engine is priced as f16x2)
        WP = ops.H2WeightPlanes if self.h2 else ops.WeightPlanes
        self.cross_planes = [WP(W) if wide_cross else None for W in self.cross_W]
        self.wplanes = [WP(W) if on and u > 1 and ops.planes_worthwhile(B, W.shape[0], W.shape[1]) else None
                        for W, u in zip(self.Ws, units)]
        if self.h2:
            rec = lambda: ops.h2_record(device)
            self.x_amax = [rec() for _ in range(len(self.cross_W) + 1)]   # x_0 .. x_L
            self.h_amax = [rec() for _ in units]                          # MLP activations h_i
            self.dh_amax = [rec() for _ in units]                         # their gradients (dy of layer i)
            self.dp_amax = rec()                                          # d_prod of the cross layer being differentiated
        self.hs = [torch.empty((B, _pad4(u)), **f32)[:, :u] for u in units]
        self.dhs = [torch.empty((B, _pad4(u)), **f32)[:, :u] for u in units[:-1]]
        self.d_top = torch.zeros((B, ld), **f32)
        self.zero_logit = torch.zeros(B, **f32)
        self.prob, self.d_logit, self.loss = torch.empty(B, **f32), torch.empty(B, **f32), torch.zeros(1, **f32)
        self.ws = torch.empty(1024, **f32)
        self.dw_ws = [ops.linear_bwd_dw_workspace(B, W.shape[0], W.shape[1], device) for W in self.Ws]
        self.cross_ws = ops.linear_bwd_dw_workspace(B, in_dim, in_dim, device)
        # wide layers: the register-split wgrad (dr_bf3_wgrad / dr_h2_wgrad) on the same condition as the planes forward / dgrad
        self.wg_ws = [ops.bf3_wgrad_workspace(B, W.shape[0], W.shape[1], device) if wp is not None else None
                      for W, wp in zip(self.Ws, self.wplanes)]
        self.cross_wg_ws = ops.bf3_wgrad_workspace(B, in_dim, in_dim, device) if wide_cross else None

    def planes(self):
        return [wp for wp in list(self.cross_planes) + list(self.wplanes) if wp is not None]

    def refresh_planes(self):
        for wp in self.planes():
            wp.refresh()

    def ensure_fresh(self):
        for wp in self.planes():     # weights written from outside since the last refresh: re-split
            wp.ensure_fresh()

    def flops_step(self):
        fl = len(self.cross_W) * 2 * self.B * self.in_dim * self.in_dim
        for W in self.Ws:
            fl += 2 * self.B * W.shape[0] * W.shape[1]
        return 3 * fl

    def step(self, x0_buf, labels, lr):
        """x0_buf [B, ld] (columns >= in_dim zero) -> d loss / d x0 as a [B, in_dim] view; loss in self.loss (mean over the B rows)."""
        B, n_in = self.B, self.in_dim
        inplace = self.grads is None
        sc = -lr if inplace else 1.0
        gcW, gcb, gW, gb = (self.cross_W, self.cross_b, self.Ws, self.bs) if inplace else self.grads
        x0 = x0_buf[:, :n_in]
        h2 = self.h2
        if h2:
            self._k("h2_amax_x0", "hbm", 4.0 * B * n_in, lambda: ops.h2_amax(x0, self.x_amax[0]))
        xs, prods = [x0], []
        fl_c = 2.0 * B * n_in * n_in
        for l, (W, b) in enumerate(zip(self.cross_W, self.cross_b)):
            if h2:
                out, prod = self._k("cross_fwd_L%d" % l, "mfma", fl_c,
                                    lambda x=xs[-1], b=b, l=l: ops.h2_cross_fwd(x0, x, self.x_amax[l], self.cross_planes[l].wt, b, self.diag,
                                                                                want_prod=True, out_amax=self.x_amax[l + 1]))
            elif self.cross_planes[l] is not None:
                out, prod = self._k("cross_fwd_L%d" % l, "mfma", fl_c,
                                    lambda x=xs[-1], b=b, l=l: ops.bf3_cross_fwd(x0, x, self.cross_planes[l].wt, b, self.diag, want_prod=True))
            else:
                out, prod = self._k("cross_fwd_L%d" % l, "mfma", fl_c,
                                    lambda x=xs[-1], W=W, b=b: ops.cross_fwd(x0, x, W, b, self.diag, want_prod=True))
            xs.append(out)
            prods.append(prod)
        x = xs[-1]
        xam = self.x_amax[-1] if h2 else None                 # record of the current MLP input (None: not known)
        for i, (W, b) in enumerate(zip(self.Ws, self.bs)):
            if h2 and self.wplanes[i] is not None:
                if xam is None:
                    xam = ops.h2_amax(x, self.h_amax[i - 1])
                self._k("linear_fwd_L%d" % i, "mfma", 2.0 * B * W.shape[0] * W.shape[1],
                        lambda x=x, b=b, i=i, xam=xam: ops.h2_linear_nt(x, xam, self.wplanes[i].wt, bias=b, act=self.acts[i], out=self.hs[i],
                                                                        out_amax=self.h_amax[i]))
                xam = self.h_amax[i]
                x = self.hs[i]
                continue
            xam = None
            if self.wplanes[i] is not None:
                self._k("linear_fwd_L%d" % i, "mfma", 2.0 * B * W.shape[0] * W.shape[1],
                        lambda x=x, b=b, i=i: ops.bf3_linear_nt(x, self.wplanes[i].wt, bias=b, act=self.acts[i], out=self.hs[i]))
            else:
                self._k("linear_fwd_L%d" % i, "mfma", 2.0 * B * W.shape[0] * W.shape[1],
                        lambda x=x, W=W, b=b, i=i: ops.linear_fwd(x, W, b, self.acts[i], out=self.hs[i]))
            x = self.hs[i]
        ops.bce_fwd_bwd(self.zero_logit, labels, ops.LOSS_SIGMOID_CE, workspace=self.ws, logits_b=self.hs[-1],
                        out=(self.prob, self.d_logit, self.loss))
        # ---- backward: MLP ---------------------------------------------------------------------------------------------
        dy = self.d_logit.reshape(-1, 1)
        dyam = None                                           # record of dy (None: not known, built by a pass when a GEMM wants it)
        for i in range(len(self.Ws) - 1, -1, -1):
            xin = xs[-1] if i == 0 else self.hs[i - 1]
            W = self.Ws[i]
            fl = 2.0 * B * W.shape[0] * W.shape[1]
            if i > 0:
                dx, rs = self.dhs[i - 1], (self.hs[i - 1] if self.acts[i - 1] else None)
            else:
                dx, rs = self.d_top[:, :n_in], None
            if h2 and self.wplanes[i] is not None:
                if dyam is None:
                    dyam = ops.h2_amax(dy, self.dh_amax[i])
                xinam = self.x_amax[-1] if i == 0 else self.h_amax[i - 1]
                dxam = self.dh_amax[i - 1] if i > 0 else None
                self._k("linear_bwd_dx_L%d" % i, "mfma", fl,
                        lambda dy=dy, rs=rs, dx=dx, i=i, dyam=dyam, dxam=dxam: ops.h2_linear_nt(dy, dyam, self.wplanes[i].w, mask=rs, out=dx,
                                                                                                out_amax=dxam))
                self._k("linear_bwd_dw_L%d" % i, "mfma", fl,
                        lambda xin=xin, dy=dy, i=i, dyam=dyam, xinam=xinam: ops.h2_wgrad(xin, xinam, dy, dyam, sc, gW[i], gb[i],
                                                                                          workspace=self.wg_ws[i]))
                if inplace:
                    self.wplanes[i].refresh()
                dy, dyam = dx, dxam
                continue
            dyam = None
            if self.wplanes[i] is not None:
                self._k("linear_bwd_dx_L%d" % i, "mfma", fl,
                        lambda dy=dy, rs=rs, dx=dx, i=i: ops.bf3_linear_nt(dy, self.wplanes[i].w, mask=rs, out=dx))
            else:
                self._k("linear_bwd_dx_L%d" % i, "mfma", fl, lambda dy=dy, W=W, rs=rs, dx=dx: ops.linear_bwd_dx(dy, W, relu_src=rs, out=dx))
            if self.wg_ws[i] is not None:
                self._k("linear_bwd_dw_L%d" % i, "mfma", fl,
                        lambda xin=xin, dy=dy, i=i: ops.bf3_wgrad(xin, dy, sc, gW[i], gb[i], workspace=self.wg_ws[i]))
            else:
                self._k("linear_bwd_dw_L%d" % i, "mfma", fl,
                        lambda xin=xin, dy=dy, i=i: ops.linear_bwd_dw(xin, dy, sc, gW[i], gb[i], workspace=self.dw_ws[i]))
            if self.wplanes[i] is not None and inplace:
                self.wplanes[i].refresh()
            dy = dx
        # ---- backward: cross stack.  d_out of layer l -> (d_x0 +=, d_x_l), W_l / b_l updated (or their gradients accumulated) -----
        d_out = self.d_top[:, :n_in]
        d_x0 = torch.zeros((B, self.ld), dtype=torch.float32, device=d_out.device)[:, :n_in]
        for l in range(len(self.cross_W) - 1, -1, -1):
            if self.diag == 0.0:
                # d_x = d_out + d_prod W^T: the dgrad below ACCUMULATES into the d_out buffer itself (d_out's last reader is the
                # combine kernel) -- no zero-filled d_x, no read-modify-write of it in the combine pass: -1.3 GB per layer
                d_x = d_out
                d_prod = self._k("cross_combine_bwd_L%d" % l, "hbm", 6.0 * 4 * B * n_in,
                                 lambda l=l, d_out=d_out: ops.cross_combine_bwd(x0, prods[l], d_out, 0.0, d_x0, None,
                                                                                d_prod_amax=self.dp_amax if h2 else None))
            else:
                d_x = torch.zeros((B, self.ld), dtype=torch.float32, device=d_out.device)[:, :n_in]
                d_prod = self._k("cross_combine_bwd_L%d" % l, "hbm", 6.0 * 4 * B * n_in,
                                 lambda l=l, d_out=d_out, d_x=d_x: ops.cross_combine_bwd(x0, prods[l], d_out, self.diag, d_x0, d_x,
                                                                                         d_prod_amax=self.dp_amax if h2 else None))
            if h2:
                self._k("cross_bwd_dx_L%d" % l, "mfma", fl_c,
                        lambda d_prod=d_prod, l=l, d_x=d_x: ops.h2_linear_nt(d_prod, self.dp_amax, self.cross_planes[l].w, accumulate=True, out=d_x))
                self._k("cross_bwd_dw_L%d" % l, "mfma", fl_c,
                        lambda l=l, d_prod=d_prod: ops.h2_wgrad(xs[l], self.x_amax[l], d_prod, self.dp_amax, sc, gcW[l], gcb[l],
                                                                workspace=self.cross_wg_ws))
                if inplace:
                    self.cross_planes[l].refresh()
                d_out = d_x
                continue
            if self.cross_planes[l] is not None:
                self._k("cross_bwd_dx_L%d" % l, "mfma", fl_c,
                        lambda d_prod=d_prod, l=l, d_x=d_x: ops.bf3_linear_nt(d_prod, self.cross_planes[l].w, accumulate=True, out=d_x))
            else:
                self._k("cross_bwd_dx_L%d" % l, "mfma", fl_c,
                        lambda d_prod=d_prod, l=l, d_x=d_x: ops.linear_bwd_dx(d_prod, self.cross_W[l], None, accumulate=True, out=d_x))
            if self.cross_wg_ws is not None:
                self._k("cross_bwd_dw_L%d" % l, "mfma", fl_c,
                        lambda l=l, d_prod=d_prod: ops.bf3_wgrad(xs[l], d_prod, sc, gcW[l], gcb[l], workspace=self.cross_wg_ws))
            else:
                self._k("cross_bwd_dw_L%d" % l, "mfma", fl_c,
                        lambda l=l, d_prod=d_prod: ops.linear_bwd_dw(xs[l], d_prod, sc, gcW[l], gcb[l], workspace=self.cross_ws))
            if self.cross_planes[l] is not None and inplace:
                self.cross_planes[l].refresh()
            d_out = d_x
        d_x0.add_(d_out)                                     # the first layer's x IS x0
        return d_x0


class DCNEngine:
    def __init__(self, num_fields: int, vocab_per_field: int, dim: int, num_cross: int, dnn_units: Sequence[int], batch: int,
                 num_dense: int = 0, lr: float = 0.01, diag_scale: float = 0.0, device="cuda", seed: int = 42):
        self.F, self.V, self.D, self.B, self.Nd, self.lr, self.diag = num_fields, vocab_per_field, dim, batch, num_dense, lr, diag_scale
        F, V, D, B = num_fields, vocab_per_field, dim, batch
        g = torch.Generator(device=device)
        g.manual_seed(seed)
        self.R = F * V
        self.table = torch.empty((self.R, D), dtype=torch.float32, device=device)
        std = 1.0 / math.sqrt(D)
        for r0 in range(0, self.R, 1 << 24):
            self.table[r0:r0 + (1 << 24)].normal_(0.0, std, generator=g).clamp_(-2 * std, 2 * std)
        self.in_dim = F * D + num_dense
        self.ld = _pad4(self.in_dim)
        f32 = dict(dtype=torch.float32, device=device)
        # cross layers: full-rank kernels [in_dim, in_dim] (dcn.py:55-68), truncated-normal init like the reference default
        self.cross_W, self.cross_b = [], []
        for _ in range(num_cross):
            W = torch.empty((self.in_dim, self.ld), **f32)
            W.normal_(0.0, 0.01, generator=g).clamp_(-0.02, 0.02)
            self.cross_W.append(W[:, :self.in_dim])
            self.cross_b.append(torch.zeros(self.in_dim, **f32))
        units = list(dnn_units) + [1]
        self.Ws: List[torch.Tensor] = []
        self.bs: List[torch.Tensor] = []
        d = self.in_dim
        for u in units:                                                    # [TF] B8 glorot uniform
            limit = math.sqrt(6.0 / (d + u))
            W = (torch.rand((d, _pad4(u)), device=device, generator=g) * 2 - 1) * limit
            self.Ws.append(W[:, :u])
            self.bs.append(torch.zeros(u, **f32))
            d = u
        self._events = None
        # cross stack + MLP + loss and their backward, every SGD step fused into the wgrad kernels (see DCNDense): forward / dgrad GEMMs
        # on pre-split weights, in the "f16x2" operand split unless ops.set_gemm_split("bf16x3") / DR_GEMM_SPLIT=bf16x3 (read once, by
        # the library) asks for the six-product one; DR_PLANES=0: the in-kernel-split GEMMs
        self.dense = DCNDense(self.cross_W, self.cross_b, self.Ws, self.bs, B, self.in_dim, self.ld, diag_scale, device,
                              k=lambda name, bound, work, fn: self._k(name, bound, work, fn))
        self.row_base = torch.arange(F, device=device, dtype=torch.int64) * V
        self.col_buckets = torch.full((F,), V, dtype=torch.int64, device=device)
        self.ids = torch.empty((B, F), dtype=torch.int64, device=device)
        self.x0 = torch.zeros((B, self.ld), **f32)
        self.plan = ops.SortPlan(B * F, device)
        self.side = torch.cuda.Stream(device=device)
        self.ev_ids, self.ev_sorted = torch.cuda.Event(), torch.cuda.Event()

    # the dense half's state, under the names rounds 1-4 (tests, bench.py) know
    h2 = property(lambda self: self.dense.h2)
    h2_all_wide = property(lambda self: self.dense.h2_all_wide)
    cross_planes = property(lambda self: self.dense.cross_planes)
    wplanes = property(lambda self: self.dense.wplanes)
    hs = property(lambda self: self.dense.hs)
    x_amax = property(lambda self: self.dense.x_amax)
    h_amax = property(lambda self: self.dense.h_amax)
    loss = property(lambda self: self.dense.loss)
    acts = property(lambda self: self.dense.acts)

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
            out[name] = {"bound": bound, "ms": ms, "n": len(evs), "alg_bytes" if bound in ("hbm", "overlap") else "alg_flops": work}
        return out

    def flops_step(self):
        return self.dense.flops_step()

    def refresh_planes(self):
        """Re-split every cross / MLP weight into its planes (see ops.WeightPlanes.ensure_fresh for when this is needed)."""
        self.dense.refresh_planes()

    def train_step(self, keys, dense, labels, next_keys=None):
        F, D, B, lr = self.F, self.D, self.B, self.lr
        self.dense.ensure_fresh()                            # weights written from outside since the last refresh: re-split
        # ---- forward -----------------------------------------------------------------------------------------------
        self._k("hash_bucket_i64", "hbm", B * F * 16, lambda: ops.hash_bucket_i64(keys, self.col_buckets, out=self.ids))
        self._k("emb_pool_fwd", "hbm", B * (8 * F * D + 8 * F),
                lambda: ops.emb_pool_fwd(self.ids, F, None, self.row_base, self.table, None, None, ld_concat=self.ld, concat=self.x0,
                                         want_sum_x=False, want_fm=False))
        self.ev_ids.record()
        with torch.cuda.stream(self.side):                   # slot sort for K4: hidden under the first cross GEMM
            self.side.wait_event(self.ev_ids)
            ops.emb_sort_slots(self.ids, self.row_base, self.R, self.plan)
            self.ev_sorted.record(self.side)
        if self.Nd:
            self.x0[:, F * D:F * D + self.Nd].copy_(dense)
        # ---- cross stack, MLP, loss, their backward and SGD steps ---------------------------------------------------------
        d_x0 = self.dense.step(self.x0, labels, lr)
        # ---- K4: scatter the embedding part of d_x0 (sorted, plain read-modify-write, fused SGD) ---------------------------
        torch.cuda.current_stream().wait_event(self.ev_sorted)
        self._k("emb_pool_bwd", "hbm", B * (12 * F * D + 8 * F),
                lambda: ops.emb_pool_bwd_sorted(self.ids, self.row_base, self.plan, D, self.R, d_x0, None, -lr, self.table, None, None))
        return self.loss
