"""Oracle parity of the EXACT configurations bench.py times.

Every other engine-level test uses toy shapes, where the benched orchestration is switched off (first-layer wgrad on a
second stream racing the fused K4, the wide-tile GEMM at K = 1677 / N = 256, the LPR = 16 single-valued gather kernel,
the row-ordered K4 at 1.7 M slots, the fused tower head at 65 536 rows).  Here the engines run exactly as `bench.py` builds
them -- B = 65 536, F = 26, D = 64, 13 dense, DNN [256, 32], hashed raw keys, default GEMM mode (register-split GEMMs on
pre-split weights for the first layer; the wgrad / K4 overlap is covered by its own on / off test) -- for two
consecutive steps against the host oracle (oracle/torch_ref.py under torch autograd, fp32 like the reference's TF-CPU path).

The host never holds the tables: the rows the two batches touch are gathered from HBM before the first step (at most
2 x 1.7 M rows = 0.9 GB), the oracle trains that compact table, and afterwards (a) the touched rows are compared row by row,
(b) a sample of untouched rows must be bit-identical to their initial values.  That makes the full-size table (V = 10 M per
field, 66.6 GB: the bench's own) testable, not only V = 1 M.

Tolerances (written where they are used): ids bit-exact; loss 1e-5 relative (north_star); parameters compared through their
UPDATE (after - before) so that a wrong or missing gradient cannot hide behind the size of the weights: |d_gpu - d_cpu| <=
2 ulp(weight) + 2e-3 * |d_cpu| + 1e-3 * rms(d_cpu) for SGD; 2 % of one Adam step for Adam (as in test_gpu_models.py).
The SGD tests use lr = 1.0 instead of bench.py's 0.01: the learning rate is a scalar the kernels multiply by, and at 0.01 a row's
update (~1e-8) is below the fp32 spacing of the row itself (7e-9), i.e. invisible to any comparison.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tf_semantics as O
from oracle import torch_ref as T

F, D, ND, B, DNN = 26, 64, 13, 65536, [256, 32]


def _batches(n, kind, seed):
    """bench.py:synth_batches (same distributions), its own generator seed"""
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    out = []
    for _ in range(n):
        if kind == "uniform":
            keys = torch.randint(0, 10**16, (B, F), device="cuda", generator=g)
        else:
            u = torch.rand((B, F), device="cuda", generator=g, dtype=torch.float64)
            al, nn = 1.05, float(10**12)
            keys = (((nn ** (1 - al) - 1) * u + 1) ** (1 / (1 - al))).long().clamp(1, 10**12)
        dense = torch.log1p(torch.randn((B, ND), device="cuda", generator=g).abs())
        labels = (torch.rand(B, device="cuda", generator=g) < 0.25).float()
        out.append((keys, dense, labels))
    return out


def _oracle_ids(keys, V):
    k = keys.cpu().numpy()
    return np.stack([O.hash_bucket_i64(k[:, f], V) for f in range(F)], axis=1)


def _ulp(x):
    return np.spacing(np.abs(x).astype(np.float32)).astype(np.float64)


def _assert_update(name, before, after, want_after, rel=2e-3, outliers=1e-4):
    """after - before (device) against want_after - before (oracle), see the module docstring.

    `outliers`: fraction of elements allowed outside the tight tolerance, each still bounded by half its update + 2 rms.
    Two fp32 implementations of a ReLU network cannot agree on every element: a hidden unit whose pre-activation is within
    rounding of zero (a few of the 16.7 M per step) is "on" on one side and "off" on the other, which switches one of the
    ~128 active terms of that example's 1664 input gradients (first seen at 5e-6 of the table elements, errors of ~0.5 % of
    the update).  A wrong kernel moves every element, not 1e-5 of them."""
    b64 = before.astype(np.float64)
    d_gpu, d_cpu = after.astype(np.float64) - b64, want_after.astype(np.float64) - b64
    rms = float(np.sqrt(np.mean(d_cpu * d_cpu)))
    assert rms > 0, name + ": the oracle update is identically zero (test is vacuous)"
    err = np.abs(d_gpu - d_cpu)
    tol = 2 * np.maximum(_ulp(before), _ulp(want_after)) + rel * np.abs(d_cpu) + 1e-3 * rms
    bad = err > tol
    frac = float(bad.mean())
    assert frac <= outliers, "%s: %.2e of %d elements off (allowed %.0e); worst |err| %.3e at update %.3e (rms update %.3e)" % (
        name, frac, bad.size, outliers, float(err[bad].max()), float(np.abs(d_cpu)[bad].max()), rms)
    if bad.any():
        assert bool((err[bad] <= 0.5 * np.abs(d_cpu)[bad] + 2 * rms).all()), "%s: an outlier is not ReLU-tie sized: |err| %.3e (rms %.3e)" % (
            name, float(err[bad].max()), rms)


def _assert_close_adam(name, got, want, lr, outliers, mean_tol=2e-3):
    """Adam at this batch size is an amplifier: on step t the update is lr * g / (|g| + eps'), eps' = eps / sqrt(1 - beta2^t)
    ~ 3e-7, and the gradients of a 65 536-example mean are 1e-8 .. 1e-5 -- for |g| < ~1e-6 a change of 1e-8 in g moves the
    update by more than 2 % of a step.  fp32 rounding is far below that, a ReLU tie (see _assert_update) is not: one hidden
    unit switched for one example shifts that example's 1664 embedding-gradient elements by ~0.5 % and, through dy, a whole
    1677-element column of the first layer's weight gradient by ~1e-7.  So: all but `outliers` of the elements within 2 % of a
    step (measured: 4e-4 of the table rows, 7e-3 of W0 with ~1e2 ties per step), the MEAN difference within 0.2 % of a step,
    nothing off by more than one whole step.  A wrong gradient or update rule moves the mean by tens of per cent."""
    err = np.abs(got.astype(np.float64) - want.astype(np.float64))
    frac = float((err > 2e-2 * lr).mean())
    assert frac <= outliers, "%s: %.2e of %d elements differ by more than 2 %% of an Adam step (allowed %.0e)" % (name, frac, err.size, outliers)
    assert float(err.mean()) <= mean_tol * lr, "%s: mean difference %.3e of a step" % (name, float(err.mean()) / lr)
    assert float(err.max()) <= 1.0 * lr, "%s: worst difference %.3e exceeds one Adam step" % (name, float(err.max()))


class _CompactOracle:
    """DeepFM training steps on the rows a set of batches touches (oracle/torch_ref.py math, torch autograd for the gradients,
    [TF] B15 Adam from the same module)."""

    def __init__(self, eng, ids_list, V, optimizer, lr):
        base = np.arange(F, dtype=np.int64)[None, :] * V
        rows = [i + base for i in ids_list]
        self.U = np.unique(np.concatenate([r.reshape(-1) for r in rows]))
        self.cidx = [torch.from_numpy(np.searchsorted(self.U, r)) for r in rows]
        Ud = torch.from_numpy(self.U).cuda()
        self.Ud = Ud
        self.table = eng.table[Ud].cpu()
        self.lin = eng.lin_w[Ud].cpu()
        self.bias = eng.lin_bias.cpu().clone()
        self.Ws = [w.cpu().clone().contiguous() for w in eng.Ws]
        self.bs = [b.cpu().clone() for b in eng.bs]
        self.table0, self.lin0 = self.table.clone(), self.lin.clone()
        self.Ws0, self.bs0, self.bias0 = [w.clone() for w in self.Ws], [b.clone() for b in self.bs], self.bias.clone()
        self.opt, self.lr, self.t = optimizer, lr, 0
        if optimizer == "adam":
            z = torch.zeros_like
            self.mt, self.vt, self.ml, self.vl = z(self.table), z(self.table), z(self.lin), z(self.lin)
            self.mb, self.vb = z(self.bias), z(self.bias)
            self.mW, self.vW = [z(w) for w in self.Ws], [z(w) for w in self.Ws]
            self.mB, self.vB = [z(b) for b in self.bs], [z(b) for b in self.bs]

    def resync_from(self, eng):
        """Adopt the device's state -- touched table rows, first-order weights, bias, dense parameters and (Adam) every moment -- so that
        the next step starts from IDENTICAL state on both sides (compare first, then call this)."""
        Ud = self.Ud
        self.table, self.lin = eng.table[Ud].cpu(), eng.lin_w[Ud].cpu()
        self.bias = eng.lin_bias.cpu().clone()
        self.Ws = [w.cpu().clone().contiguous() for w in eng.Ws]
        self.bs = [b.cpu().clone() for b in eng.bs]
        if self.opt == "adam":
            self.mt, self.vt = eng.m_table[Ud].cpu(), eng.v_table[Ud].cpu()
            self.ml, self.vl = eng.m_lin[Ud].cpu().clone(), eng.v_lin[Ud].cpu().clone()
            fm, fv = eng.flat_m.cpu(), eng.flat_v.cpu()
            self.mb, self.vb = fm[eng._bias_off:eng._bias_off + 1].clone(), fv[eng._bias_off:eng._bias_off + 1].clone()
            self.mW = [fm[wo:wo + k * pu].view(k, pu)[:, :u].clone().contiguous() for wo, k, pu, u, bo in eng._views]
            self.vW = [fv[wo:wo + k * pu].view(k, pu)[:, :u].clone().contiguous() for wo, k, pu, u, bo in eng._views]
            self.mB = [fm[bo:bo + u].clone() for wo, k, pu, u, bo in eng._views]
            self.vB = [fv[bo:bo + u].clone() for wo, k, pu, u, bo in eng._views]

    def step(self, i, dense, labels, relu_masks=None):
        """relu_masks: the device's ReLU decisions of this step (oracle/torch_ref.py dnn: tie-aware comparison); the units where
        they differ from the oracle's own are recorded in self.ties and must be ties (T.check_ties)."""
        cidx = self.cidx[i]
        emb = self.table[cidx].requires_grad_(True)                      # [B, F, D]  single-valued fields: x = the row ([TF] B5)
        lw = self.lin[cidx].requires_grad_(True)
        bias = self.bias.clone().requires_grad_(True)
        ks = [k.clone().requires_grad_(True) for k in self.Ws]
        bs = [b.clone().requires_grad_(True) for b in self.bs]
        x = torch.cat([emb.reshape(B, F * D), dense], 1)
        self.ties = []
        # (Adam amplifies fp32 noise in the gradients -- d step / d g is up to lr / eps' = 3e4 where |g| ~ 3e-7, see _assert_close_adam --
        # so two sides that each carry their OWN state drift apart by ~1e-5 rms in the pre-activations after one step.  Round 5 widened
        # the tie width 20 x for step 2; since round 6 the oracle adopts the device's parameters and moments after every step
        # (resync_from), each step is a one-step comparison from identical state, and a tie is a tie at the step-1 width)
        eps = 1e-5
        logit = T.fm_second_order(emb) + lw.sum(1) + bias + T.dnn(x, ks, bs, relu_masks=relu_masks, ties=self.ties, tie_eps=eps).squeeze(1)   # deepfm.py:36-47
        loss = T.sigmoid_cross_entropy(labels, logit)                                        # train_fm_on_movielens_estimator.py:46
        grads = torch.autograd.grad(loss, [emb, lw, bias] + ks + bs)
        gt = torch.zeros_like(self.table).index_add_(0, cidx.reshape(-1), grads[0].reshape(-1, D))
        gl = torch.zeros_like(self.lin).index_add_(0, cidx.reshape(-1), grads[1].reshape(-1))
        n = len(ks)
        self.t += 1
        self.last_gW, self.last_gb = [g_.detach().clone() for g_ in grads[3:3 + n]], [g_.detach().clone() for g_ in grads[3 + n:3 + 2 * n]]
        # smallest |gradient| an element has seen over the steps: Adam's sensitivity to gradient round-off (see _run_deepfm)
        mins = [g_.detach().abs() for g_ in list(grads[3:3 + n]) + list(grads[3 + n:3 + 2 * n])]
        self.min_abs_g = mins if self.t == 1 else [torch.minimum(a, b_) for a, b_ in zip(self.min_abs_g, mins)]
        with torch.no_grad():
            if self.opt == "sgd":
                self.table -= self.lr * gt
                self.lin -= self.lr * gl
                self.bias -= self.lr * grads[2]
                for j in range(n):
                    self.Ws[j] -= self.lr * grads[3 + j]
                    self.bs[j] -= self.lr * grads[3 + n + j]
            else:
                touched = torch.unique(cidx.reshape(-1))
                T.adam_rows_step(self.table, gt, touched, self.mt, self.vt, self.lr, self.t)
                T.adam_rows_step(self.lin, gl, touched, self.ml, self.vl, self.lr, self.t)
                T.adam_dense_step(self.bias, grads[2], self.mb, self.vb, self.lr, self.t)
                for j in range(n):
                    T.adam_dense_step(self.Ws[j], grads[3 + j], self.mW[j], self.vW[j], self.lr, self.t)
                    T.adam_dense_step(self.bs[j], grads[3 + n + j], self.mB[j], self.vB[j], self.lr, self.t)
        return float(loss.detach())


def _device_relu_masks(eng):
    """The ReLU decisions the device made in the step that just ran, one bool [B, units] per hidden layer: h > 0 of the stored
    activations; the fused head (last hidden layer + Dense(1) + loss in one GEMM epilogue) never stores its h, there the decision is
    read off the gradient it wrote: d_h = d_logit * w2 * (h > 0) with d_logit != 0 and w2 != 0 (both asserted)."""
    torch.cuda.synchronize()
    n_hidden = len(eng.Ws) - 1
    masks = []
    for i in range(n_hidden):
        if eng._head_done and i == n_hidden - 1:
            assert bool((eng.d_logit != 0).all()) and bool((eng.Ws[-1] != 0).all())
            masks.append((eng.dhs[-1] != 0).cpu())
        else:
            masks.append((eng.hs[i] > 0).cpu())
    return masks


def _check_adam_step(eng, orc, lr, step):
    """One Adam step of the device against one Adam step of the oracle FROM THE SAME STATE (the oracle was re-synchronised after the
    previous step): the step-1 tolerances apply to every step -- all but 3e-3 of the table rows / first-order weights and 1e-2 of the
    dense weights whose |g| >= 1e-5 within 2 % of a step, mean difference within 0.2 % of a step, nothing off by more than one step
    (an element whose gradient is ~0 may take +lr on one side and -lr on the other: 2 lr) -- and the gradients themselves directly."""
    torch.cuda.synchronize()
    got_t, got_l = eng.table[orc.Ud].cpu().numpy(), eng.lin_w[orc.Ud].cpu().numpy()
    _assert_close_adam("step %d table rows" % step, got_t, orc.table.numpy(), lr, 3e-3)
    _assert_close_adam("step %d first-order weights" % step, got_l, orc.lin.numpy(), lr, 3e-3)
    _assert_close_adam("step %d first-order bias" % step, eng.lin_bias.cpu().numpy(), orc.bias.numpy(), lr, 0.0)
    nW = len(orc.Ws)
    for j in range(nW):
        # the dense gradients (the engine's Adam bucket holds this step's): the direct check.  A ReLU tie moves a whole column of W0's
        # gradient by ~1e-7, hence the rms-relative floor.
        for nm, got, want in (("gW%d" % j, eng.gWs[j], orc.last_gW[j]), ("gb%d" % j, eng.gbs[j], orc.last_gb[j])):
            got, want = got.cpu().numpy().astype(np.float64), want.numpy().astype(np.float64)
            rms = float(np.sqrt(np.mean(want * want)))
            bad = np.abs(got - want) > 1e-3 * np.abs(want) + 2e-2 * rms
            assert float(bad.mean()) <= 1e-3, "step %d %s: %.2e of the gradient elements off (rms %.3e, worst %.3e)" % (
                step, nm, float(bad.mean()), rms, float(np.abs(got - want).max()))
        # the updated weights.  A step moves an element by ~ lr * g / (|g| + 3e-7): where |g| is within an order of magnitude of 3e-7 a
        # 1e-8 difference in g -- fp32 summation order -- moves the step by per cents; those elements are covered by the gradient check,
        # the update itself is compared where this step's |g| >= 1e-5
        for nm, got, want, g_ in (("W%d" % j, eng.Ws[j], orc.Ws[j], orc.last_gW[j]), ("b%d" % j, eng.bs[j], orc.bs[j], orc.last_gb[j])):
            sel = (g_.abs() >= 1e-5).numpy()
            if sel.sum() >= 16:
                _assert_close_adam("step %d %s" % (step, nm), got.cpu().numpy()[sel], want.numpy()[sel], lr, 1e-2, mean_tol=2e-3)
            err = np.abs(got.cpu().numpy().astype(np.float64) - want.numpy().astype(np.float64))
            assert float(err.max()) <= 2.05 * lr, "step %d %s: worst difference %.3e exceeds what one Adam step can differ by" % (step, nm, float(err.max()))


def _make_engine(V, optimizer, lr, overlap=None):
    from deep_recommenders_amd.engine import DeepFMEngine
    old = os.environ.get("DR_OVERLAP_DW")
    if overlap is not None:
        os.environ["DR_OVERLAP_DW"] = overlap
    try:
        eng = DeepFMEngine(F, V, D, DNN, B, num_dense=ND, lr=lr, seed=42, lin_init_std=0.01, optimizer=optimizer)
    finally:
        if overlap is not None:
            if old is None:
                os.environ.pop("DR_OVERLAP_DW", None)
            else:
                os.environ["DR_OVERLAP_DW"] = old
    return eng


def _run_deepfm(V, optimizer, kind, lr, steps=2):
    from deep_recommenders_amd import ops
    assert ops.get_gemm_mode() == "bf16x3"                 # bench.py's default product mode
    eng = _make_engine(V, optimizer, lr)
    assert eng.fuse_head and eng.sorted_bwd and eng.wplanes[0] is not None and eng.wg_ws[0] is not None, "not the benched orchestration"
    assert eng.no_concat == (os.environ.get("DR_NO_CONCAT", "1") == "1")
    assert ops.linear_bwd_narrow_supported(B, 256, 32) and eng.narrow_ws[1] is not None
    batches = _batches(steps, kind, seed=1234)
    ids_list = [_oracle_ids(k, V) for k, _, _ in batches]
    orc = _CompactOracle(eng, ids_list, V, optimizer, lr)
    # untouched rows: a sample outside the touched set, must stay bit-identical
    g = torch.Generator().manual_seed(5)
    cand = torch.randint(0, F * V, (40000,), generator=g).numpy()
    untouched = torch.from_numpy(cand[~np.isin(cand, orc.U)][:20000]).cuda()
    assert untouched.numel() >= 10000
    un_t0, un_l0 = eng.table[untouched].clone(), eng.lin_w[untouched].clone()
    n_ties = 0
    for i, (keys, dense, labels) in enumerate(batches):
        # exactly bench.py's call: the next batch's keys / dense features ride along, so from the second step on the hash, the slot
        # plan and the dense-feature placement come from the side-stream prefetch (the full-size test used to call
        # train_step without them)
        nk, nd = (batches[i + 1][0], batches[i + 1][1]) if i + 1 < len(batches) else (None, None)
        loss = float(eng.train_step(keys, dense, labels, next_keys=nk, next_dense=nd).item())
        assert eng._plan_prefetched == (i > 0 and eng.prefetch_plan), "the prefetched plan was not picked up"
        np.testing.assert_array_equal(eng.ids.cpu().numpy(), ids_list[i])                     # integer path: bit-exact
        want = orc.step(i, dense.cpu(), labels.cpu(), relu_masks=_device_relu_masks(eng))
        T.check_ties(orc.ties, max_frac=1e-5 if optimizer == "sgd" else 1e-4)
        n_ties += sum(t["disagree"] for t in orc.ties)
        assert abs(loss - want) <= 1e-5 * abs(want), (i, loss, want)                          # north_star: 1e-5 relative
        if optimizer == "adam":
            # every Adam step is checked on its own, from identical state, then the oracle adopts the device's state
            _check_adam_step(eng, orc, lr, i)
            if i + 1 < len(batches):
                orc.resync_from(eng)
    torch.cuda.synchronize()
    print("ReLU ties resolved the device's way over %d steps: %d" % (steps, n_ties))
    assert torch.equal(eng.table[untouched], un_t0) and torch.equal(eng.lin_w[untouched], un_l0), "an untouched row changed"
    got_t, got_l = eng.table[orc.Ud].cpu().numpy(), eng.lin_w[orc.Ud].cpu().numpy()
    if optimizer == "sgd":
        _assert_update("table rows", orc.table0.numpy(), got_t, orc.table.numpy())
        _assert_update("first-order weights", orc.lin0.numpy(), got_l, orc.lin.numpy())
        _assert_update("first-order bias", orc.bias0.numpy(), eng.lin_bias.cpu().numpy(), orc.bias.numpy())
        for j in range(len(orc.Ws)):
            # (round 5: the oracle evaluates the ReLU branches the device took -- see _device_relu_masks -- so a tie no longer moves a
            # whole d h0 row of one side only, and W0 is held to the same 1e-4 as everything else; round 4 had widened it to 2e-2)
            _assert_update("W%d" % j, orc.Ws0[j].numpy(), eng.Ws[j].cpu().numpy(), orc.Ws[j].numpy())
            _assert_update("b%d" % j, orc.bs0[j].numpy(), eng.bs[j].cpu().numpy(), orc.bs[j].numpy())
    else:
        # (every step was checked by _check_adam_step on its way)
        # and the update must not be vacuous: most touched rows moved by about one Adam step
        moved = np.abs(got_t - orc.table0.numpy()).max(axis=1)
        assert np.median(moved) > 0.2 * lr
    return eng, batches


def test_deepfm_bench_config_full_vocab_sgd_uniform():
    """bench.py's default line: V = 10 M rows per field (66.6 GB slab), uniform keys, fused SGD."""
    free, _ = torch.cuda.mem_get_info()
    V = 10_000_000 if free > 90e9 else 1_000_000
    _run_deepfm(V, "sgd", "uniform", lr=1.0)


def test_deepfm_bench_config_with_concat_buffer(monkeypatch):
    """DR_NO_CONCAT=0: the round-2 data flow (the fused forward stores `concat`, the first-layer wgrad and K4's duplicate pass read
    it) stays a supported switch -- same oracle assertions as the default (concat never built: gathering wgrad + row snapshot)."""
    monkeypatch.setenv("DR_NO_CONCAT", "0")
    eng, _ = _run_deepfm(1_000_000, "sgd", "uniform", lr=1.0)
    assert not eng.no_concat


@pytest.mark.parametrize("kind", ["uniform", "zipf"])
def test_deepfm_bench_config_lin_side_stream(kind, monkeypatch):
    """DR_LIN_SIDE=1: the first-order weights of rows unique in the batch are updated by dr_emb_lin_update_unique on the side stream
    (beside the tower tail) and K4 skips them -- same oracle assertions as the default."""
    monkeypatch.setenv("DR_LIN_SIDE", "1")
    eng, _ = _run_deepfm(1_000_000, "sgd", kind, lr=1.0)
    assert eng.lin_side


def test_deepfm_bench_config_sgd_zipf():
    """Zipf(1.05) keys (bench.py --ids zipf): most slots are duplicates, the K4 segment-sum path carries the step."""
    _run_deepfm(1_000_000, "sgd", "zipf", lr=1.0)


@pytest.mark.parametrize("kind", ["uniform", "zipf"])
def test_deepfm_bench_config_adam(kind):
    """bench.py --optimizer adam (the reference examples' optimizer, lr 0.01): fused row-wise Adam K4 + dense Adam."""
    _run_deepfm(1_000_000, "adam", kind, lr=0.01)


@pytest.mark.parametrize("no_concat", ["1", "0"])
def test_deepfm_zipf_step_is_reproducible(no_concat, monkeypatch):
    """Run-to-run reproducibility of one SGD step on Zipf keys.  Default data flow (concat never built, K4 gets
    the x_sorted snapshot buffer): rows hit by more than 32 slots are summed in 32-slot pieces that PARK their sums and are added in
    sorted order by emb_bwd_hot_apply_kernel -- every table row and first-order weight must be BIT-identical between two runs from
    identical state, hot rows included.  DR_NO_CONCAT=0 (no x_sorted): the pieces combine with fp32 atomics, so rows hit <= 32
    times must be bit-identical and hotter rows may differ by summation order only (bounded by 1e-3 of the update's scale)."""
    monkeypatch.setenv("DR_NO_CONCAT", no_concat)
    V = 1_000_000
    batches = _batches(1, "zipf", seed=77)
    keys, dense, labels = batches[0]
    ids = _oracle_ids(keys, V) + np.arange(F, dtype=np.int64)[None, :] * V
    U, cnt = np.unique(ids.reshape(-1), return_counts=True)
    assert (cnt > 32).sum() > 10 and (cnt > 1).sum() > 1000            # the batch does exercise the hot-row path
    cool, hot = torch.from_numpy(U[cnt <= 32]).cuda(), torch.from_numpy(U[cnt > 32]).cuda()
    outs = []
    for _ in range(2):
        eng = _make_engine(V, "sgd", 1.0)
        assert eng.no_concat == (no_concat == "1")
        t0 = eng.table[hot].clone()
        l0 = eng.lin_w[hot].clone()
        eng.train_step(keys, dense, labels)
        torch.cuda.synchronize()
        outs.append((eng.table[cool].clone(), eng.lin_w[cool].clone(), eng.table[hot].clone(), t0, eng.lin_w[hot].clone(), l0))
        del eng
        torch.cuda.empty_cache()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), "a row hit <= 32 times is not bit-reproducible"
    upd = (outs[0][2] - outs[0][3]).abs().max().item()
    assert upd > 0 and (outs[0][4] - outs[0][5]).abs().max().item() > 0
    if no_concat == "1":
        assert torch.equal(outs[0][2], outs[1][2]), "a hot row is not bit-reproducible"
        assert torch.equal(outs[0][4], outs[1][4]), "a hot row's first-order weight is not bit-reproducible"
    else:
        # (first seen: 1.8e-7 on an update of 2.7e-3 -- a few ulps of the row's value, the re-association of 32-slot pieces)
        assert (outs[0][2] - outs[1][2]).abs().max().item() <= 1e-3 * upd + 1e-6


@pytest.mark.parametrize("kind", ["uniform", "zipf"])
def test_deepfm_three_steps_are_bit_reproducible(kind):
    """Three SGD steps of the benched orchestration (next batch's plan prefetched beside K4, as bench.py runs it) from identical
    state, twice: every loss and every parameter -- tables, first-order weights and bias, all Dense kernels and biases -- must come
    out BIT-identical.  Nothing in the step combines floating-point values in an arrival order: split-K partials are reduced in
    slice order, bias gradients by one block in a fixed order, shared rows in sorted order (hot rows included), and two streams
    never write the same word."""
    V = 1_000_000
    batches = _batches(3, kind, seed=4242)
    res = []
    for _ in range(2):
        eng = _make_engine(V, "sgd", 1.0)
        assert eng.no_concat and eng.sorted_bwd
        losses = []
        for i, (keys, dense, labels) in enumerate(batches):
            nk = batches[i + 1][0] if i + 1 < len(batches) else None
            losses.append(eng.train_step(keys, dense, labels, next_keys=nk).clone())
        torch.cuda.synchronize()
        res.append(([l.cpu() for l in losses], eng.table.clone(), eng.lin_w.clone(), eng.lin_bias.clone(),
                    [w.clone() for w in eng.Ws], [b.clone() for b in eng.bs]))
        del eng
        torch.cuda.empty_cache()
    a, b = res
    for la, lb in zip(a[0], b[0]):
        assert torch.equal(la, lb), (la, lb)
    assert torch.equal(a[1], b[1]), "embedding tables differ between two identical runs"
    assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]), "first-order weights / bias differ"
    for i, (wa, wb) in enumerate(zip(a[4], b[4])):
        assert torch.equal(wa, wb), "Dense kernel %d differs" % i
    for i, (ba, bb) in enumerate(zip(a[5], b[5])):
        assert torch.equal(ba, bb), "Dense bias %d differs" % i


def test_deepfm_bench_config_overlap_on_off_agree():
    """DR_OVERLAP_DW=0 / 1 (first-layer wgrad on the second stream next to K4, or in line) are the same computation: two steps
    from identical parameters must give the same losses and the same parameter updates.  Not bit-identical by construction:
    the first layer's bias gradient is combined with fp32 atomics by the split-K wgrad blocks (order varies run to run), which
    reaches everything in step 2 at the 1e-7 level; a race between the two streams would be orders of magnitude larger."""
    V = 1_000_000
    batches = _batches(2, "uniform", seed=99)
    base = np.arange(F, dtype=np.int64)[None, :] * V
    rows = torch.from_numpy(np.unique(np.concatenate([(_oracle_ids(k, V) + base).reshape(-1) for k, _, _ in batches]))).cuda()
    res = []
    for ov in ("0", "1"):
        eng = _make_engine(V, "sgd", 1.0, overlap=ov)
        assert eng.overlap_dw == (ov == "1")
        before = (eng.table[rows].cpu().numpy(), eng.Ws[0].cpu().numpy().copy(), eng.lin_w[rows].cpu().numpy())
        losses = [float(eng.train_step(*b).item()) for b in batches]
        torch.cuda.synchronize()
        res.append((losses, before, (eng.table[rows].cpu().numpy(), eng.Ws[0].cpu().numpy().copy(), eng.lin_w[rows].cpu().numpy())))
        del eng
        torch.cuda.empty_cache()
    for a, b in zip(res[0][0], res[1][0]):
        assert abs(a - b) <= 1e-6 * abs(a), (res[0][0], res[1][0])
    for name, b0, a0, a1 in zip(("table rows", "W0", "first-order weights"), res[0][1], res[0][2], res[1][2]):
        _assert_update(name + " (overlap off vs on)", b0, a1, a0, rel=1e-4, outliers=1e-4)


def _dcn_step_vs_oracle(Bd, V, tie_aware=True):
    """One DCNEngine step at bench.py --model dcn's shapes against the float64 oracle (T.cross / T.dnn under autograd).  Returns
    (loss_device, loss_oracle, ties, [(name, before, after_device, after_oracle), ...]); see the test below."""
    from deep_recommenders_amd.dcn_engine import DCNEngine
    lr = 1.0
    eng = DCNEngine(F, V, D, 3, [1024, 512, 256], Bd, num_dense=ND, lr=lr, seed=11)
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    with torch.no_grad():
        for b in eng.cross_b:
            b.normal_(0, 0.05, generator=g)
        eng.bs[-1].fill_(0.03)          # see test_gpu_models.py::test_dcn_engine_train_step_matches_oracle (loss kink at logit 0)
    keys = torch.randint(0, 10**16, (Bd, F), device="cuda", generator=g)
    dense = torch.log1p(torch.randn((Bd, ND), device="cuda", generator=g).abs())
    labels = (torch.rand(Bd, device="cuda", generator=g) < 0.25).float()
    k = keys.cpu().numpy()
    ids = np.stack([O.hash_bucket_i64(k[:, f], V) for f in range(F)], axis=1)
    rows = ids + np.arange(F, dtype=np.int64)[None, :] * V
    U = np.unique(rows.reshape(-1))
    cidx = torch.from_numpy(np.searchsorted(U, rows))
    Ud = torch.from_numpy(U).cuda()
    t0 = eng.table[Ud].cpu()
    cW0 = [w.cpu().clone() for w in eng.cross_W]
    cb0 = [b.cpu().clone() for b in eng.cross_b]
    Ws0 = [w.cpu().clone() for w in eng.Ws]
    bs0 = [b.cpu().clone() for b in eng.bs]
    loss = float(eng.train_step(keys, dense, labels).item())
    np.testing.assert_array_equal(eng.ids.cpu().numpy(), ids)
    dd = torch.float64
    emb = t0.to(dd)[cidx].requires_grad_(True)
    cW = [w.to(dd).requires_grad_(True) for w in cW0]
    cb = [b.to(dd).requires_grad_(True) for b in cb0]
    Ws = [w.to(dd).requires_grad_(True) for w in Ws0]
    bs = [b.to(dd).requires_grad_(True) for b in bs0]
    x0 = torch.cat([emb.reshape(Bd, F * D), dense.cpu().to(dd)], 1)
    x = x0
    for W, b in zip(cW, cb):
        x = T.cross(x0, x, W, b, 0.0)                                            # dcn.py:81-88
    # tie-aware: the fp64 oracle takes the ReLU branches the device took (its stored activations h > 0); every unit where that
    # differs from the oracle's own z > 0 must be within 1e-5 rms(z) of zero (T.check_ties, asserted by the caller)
    torch.cuda.synchronize()
    masks, ties = ([(h > 0).cpu() for h in eng.hs[:-1]] if tie_aware else None), []
    x = T.dnn(x, Ws, bs, relu_masks=masks, ties=ties).reshape(-1)
    lo = T.sigmoid_cross_entropy(labels.cpu().to(dd), x)
    grads = torch.autograd.grad(lo, [emb] + cW + cb + Ws + bs)
    gt = torch.zeros((len(U), D), dtype=dd).index_add_(0, cidx.reshape(-1), grads[0].reshape(-1, D))
    out = [("table rows", t0.numpy(), eng.table[Ud].cpu().numpy(), (t0.to(dd) - lr * gt).float().numpy())]
    n, m = 3, len(Ws)
    for j in range(n):
        out.append(("cross W%d" % j, cW0[j].numpy(), eng.cross_W[j].cpu().numpy(), (cW0[j].to(dd) - lr * grads[1 + j]).float().numpy()))
        out.append(("cross b%d" % j, cb0[j].numpy(), eng.cross_b[j].cpu().numpy(), (cb0[j].to(dd) - lr * grads[1 + n + j]).float().numpy()))
    for j in range(m):
        out.append(("mlp W%d" % j, Ws0[j].numpy(), eng.Ws[j].cpu().numpy(), (Ws0[j].to(dd) - lr * grads[1 + 2 * n + j]).float().numpy()))
        out.append(("mlp b%d" % j, bs0[j].numpy(), eng.bs[j].cpu().numpy(), (bs0[j].to(dd) - lr * grads[1 + 2 * n + m + j]).float().numpy()))
    del eng
    torch.cuda.empty_cache()
    return loss, lo.item(), ties, out


@pytest.mark.parametrize("Bd,V", [(8192, 200_000), (65536, 10_000_000)])
def test_dcn_bench_config_matches_oracle(Bd, V):
    """bench.py --model dcn (BASELINE config 4): Din = 26 * 64 + 13 = 1677, 3 full-rank cross layers, MLP [1024, 512, 256],
    fused SGD.  Batch 8192 keeps the host side short; batch 65 536 over the full 26 x 10 M-row slab (66.6 GB) is the configuration
    bench.py --model dcn times (the oracle test ran at V = 1 M), with the compact-table trick of the DeepFM tests --
    the touched rows are gathered to the host and the oracle trains that compact table.
    Oracle in float64 (T.cross / dense layers under autograd), as in test_gpu_models.py, tie-aware (oracle/torch_ref.py dnn): no
    allowance for ReLU ties in the tolerances below (round 4 had widened the table rows' to 1e-3)."""
    loss, want, ties, params = _dcn_step_vs_oracle(Bd, V)
    T.check_ties(ties)
    print("ReLU ties resolved the device's way: %s" % [(t["layer"], t["disagree"], t["worst_abs_z"]) for t in ties])
    assert abs(loss - want) <= 1e-5 * abs(want), (loss, want)
    # K = 1677-long fp32 reductions feeding three stacked cross layers: the device gradient carries ~1e-6 * sqrt(K) relative
    # round-off per layer; 1e-2 of each update (+ the ulp / rms floor) still catches any missing term
    for name, before, after, want_after in params:
        _assert_update(name, before, after, want_after, rel=1e-2)


@pytest.mark.parametrize("variant", ["plain", "corrections"])
def test_retrieval_bench_config_matches_oracle(variant):
    """Retrieval.call at BASELINE config 5's in-batch size: B = 8192 query / candidate pairs, D = 128, loss and both input
    gradients against the float64 restatement of sbcnm.py:120-151 (`corrections`: temperature, sampling-probability
    correction, accidental-negative removal and sample weights together)."""
    from deep_recommenders_amd.keras.models.retrieval.sbcnm import Retrieval
    Bq, Dq = 8192, 128
    rng = np.random.default_rng(42)
    q = (rng.standard_normal((Bq, Dq)) / np.sqrt(Dq)).astype(np.float32)
    c = (rng.standard_normal((Bq, Dq)) / np.sqrt(Dq)).astype(np.float32)
    kw, okw, temp = {}, {}, None
    if variant == "corrections":
        temp = 0.5
        w = rng.uniform(0.5, 1.5, Bq).astype(np.float32)
        p = rng.uniform(0.01, 0.9, Bq).astype(np.float32)
        cid = rng.integers(0, Bq // 2, Bq).astype(np.int64)             # ~half the candidates share an id with another one
        kw = dict(sample_weight=torch.tensor(w).cuda(), candidate_sampling_probability=torch.tensor(p).cuda(),
                  candidate_ids=torch.tensor(cid).cuda())
        okw = dict(sample_weight=torch.tensor(w, dtype=torch.float64), cand_prob=torch.tensor(p, dtype=torch.float64),
                   cand_ids=torch.tensor(cid), temperature=temp)
    qd = torch.tensor(q).cuda().requires_grad_(True)
    cd = torch.tensor(c).cuda().requires_grad_(True)
    task = Retrieval(temperature=temp)
    loss = task(qd, cd, compute_metrics=False, **kw)
    loss.backward()
    qo = torch.tensor(q, dtype=torch.float64, requires_grad=True)
    co = torch.tensor(c, dtype=torch.float64, requires_grad=True)
    lo = T.inbatch_softmax_loss(qo, co, **okw)
    lo.backward()
    assert abs(float(loss) - lo.item()) <= 1e-5 * abs(lo.item()), (float(loss), lo.item())
    for name, got, want in (("dq", qd.grad, qo.grad), ("dc", cd.grad, co.grad)):
        want = want.numpy()
        err = np.abs(got.cpu().numpy().astype(np.float64) - want)
        # fp32 GEMM over 8192 terms: 2e-6 * sqrt(8192) relative to the largest gradient entry
        assert err.max() <= 2e-6 * np.sqrt(Bq) * np.abs(want).max() + 1e-7, (name, err.max(), np.abs(want).max())
