"""Pins the oracle's [TF]-level rules against INDEPENDENT third-party implementations that ship in this image (VERDICT r2
item 8): TensorFlow itself cannot be installed here and the reference holds no vector for these rules, so until round 3 they
were checked only against a second restatement by the same author (tests/golden/make_semantics_fixtures.py).  PyTorch's own
operators were written by other people against the same published definitions; where TF's and torch's definitions differ
(Adam's epsilon, log_loss's epsilon) the difference is stated and the test pins the oracle through the exact algebraic map.

CPU only.  Provenance table: oracle/README.md."""
import numpy as np
import torch
import torch.nn.functional as Fn

from oracle import tf_semantics as O
from oracle import torch_ref as T


def _rng(seed):
    return np.random.default_rng(seed)


def test_mean_combiner_against_torch_embedding_bag():
    """[TF] B4/B5 safe_embedding_lookup_sparse(combiner='mean'): ids < 0 dropped, mean over the rest, empty bag -> zeros
    == torch.nn.functional.embedding_bag(mode='mean', padding_idx=...) (padding entries are excluded from the mean, an
    all-padding bag yields zeros)."""
    r = _rng(1)
    R, D, B, L = 50, 12, 64, 7
    table = r.standard_normal((R, D)).astype(np.float32)
    ids = r.integers(0, R, size=(B, L))
    ids[r.random((B, L)) < 0.35] = -1
    ids[3] = -1                                            # an empty bag
    ids[4, 1:] = -1                                        # a single id
    ids[5] = ids[5, 0]                                     # the same id L times
    got = O.embedding_mean_pool(table, ids)
    got_fast = O.embedding_mean_pool_fast(table, ids)
    w = torch.cat([torch.from_numpy(table), torch.zeros(1, D)])            # row R = the padding row
    idx = torch.from_numpy(np.where(ids >= 0, ids, R))
    want = Fn.embedding_bag(idx, w, mode="mean", padding_idx=R).numpy()
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(got_fast, want, rtol=1e-6, atol=1e-6)
    assert np.all(got[3] == 0)
    # the torch-side restatement used for gradients (oracle/torch_ref.py) against the same operator
    cs = [0, L]
    emb = T.pool_fields(torch.from_numpy(table), torch.from_numpy(ids), cs, [0])[0].numpy()
    np.testing.assert_allclose(emb, want, rtol=1e-6, atol=1e-6)


def test_first_order_gather_sum_against_torch_embedding_bag_sum():
    """[TF] B3 + B7: indicator (multi-hot, duplicates add, -1 dropped) x Dense(1) == sum of the looked-up weights
    == embedding_bag(mode='sum', padding_idx)."""
    r = _rng(2)
    V, B, L = 40, 33, 5
    w = r.standard_normal(V).astype(np.float32)
    ids = r.integers(0, V, size=(B, L))
    ids[r.random((B, L)) < 0.3] = -1
    ids[0] = 7                                             # duplicates of one id add
    mh = O.indicator_multi_hot(ids, V)
    dense_form = mh @ w + np.float32(0.25)
    wt = torch.cat([torch.from_numpy(w), torch.zeros(1)])[:, None]
    want = Fn.embedding_bag(torch.from_numpy(np.where(ids >= 0, ids, V)), wt, mode="sum", padding_idx=V).numpy()[:, 0] + 0.25
    np.testing.assert_allclose(dense_form, want, rtol=1e-5, atol=1e-6)
    got = O.first_order_gather([ids], [w], 0.25).reshape(-1)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)


def test_sigmoid_cross_entropy_against_torch_bce_with_logits():
    """[TF] B9 tf.losses.sigmoid_cross_entropy (mean of max(x,0) - x z + log1p(exp(-|x|))) == BCEWithLogitsLoss (mean)."""
    x = np.array([0.0, 1e-4, -1e-4, 0.5, -0.5, 3.0, -3.0, 20.0, -20.0, 50.0, -50.0, 88.0, -88.0, 7.5], dtype=np.float32)
    for z in (np.zeros_like(x), np.ones_like(x), (np.arange(x.size) % 2).astype(np.float32), np.full_like(x, 0.3)):
        want = torch.nn.BCEWithLogitsLoss()(torch.from_numpy(x).double(), torch.from_numpy(z).double()).item()
        assert abs(float(O.sigmoid_cross_entropy(z, x)) - want) <= 1e-6 * max(1.0, abs(want))
        got_t = T.sigmoid_cross_entropy(torch.from_numpy(z).double(), torch.from_numpy(x).double()).item()
        assert abs(got_t - want) <= 1e-12 * max(1.0, abs(want))
    # gradient of the torch-side restatement == torch's own (sigmoid(x) - z) / n
    xt = torch.from_numpy(x).double().requires_grad_(True)
    zt = torch.from_numpy((np.arange(x.size) % 2).astype(np.float64))
    T.sigmoid_cross_entropy(zt, xt).backward()
    xr = torch.from_numpy(x).double().requires_grad_(True)
    torch.nn.BCEWithLogitsLoss()(xr, zt).backward()
    np.testing.assert_allclose(xt.grad.numpy(), xr.grad.numpy(), rtol=1e-10, atol=1e-14)


def test_log_loss_and_keras_bce_against_torch_binary_cross_entropy():
    """[TF] B10 / B11 on probabilities.  torch's binary_cross_entropy is -z log p - (1 - z) log(1 - p) with the logs clamped at
    -100; TF adds eps = 1e-7 INSIDE the logs (and Keras clips p to [eps, 1 - eps] first).  Away from p in {0, 1} the two agree to
    O(eps / p); the test pins the formula there and checks the documented eps behaviour at the ends."""
    p = np.array([0.02, 0.1, 0.3, 0.5, 0.7, 0.9, 0.98], dtype=np.float32)
    z = np.array([0, 1, 1, 0, 1, 0, 1], dtype=np.float32)
    want = Fn.binary_cross_entropy(torch.from_numpy(p).double(), torch.from_numpy(z).double()).item()
    assert abs(float(O.log_loss(z, p)) - want) <= 1e-5 * want
    assert abs(float(O.keras_binary_crossentropy(z, p)) - want) <= 1e-5 * want
    # ends: TF's eps keeps the loss finite where the exact formula is infinite
    assert np.isfinite(O.log_loss(np.array([1.0], np.float32), np.array([0.0], np.float32)))
    assert abs(float(O.log_loss(np.array([1.0], np.float32), np.array([0.0], np.float32))) - (-np.log(1e-7))) < 1e-4
    assert abs(float(O.keras_binary_crossentropy(np.array([1.0], np.float32), np.array([0.0], np.float32))) - (-np.log(2e-7))) < 1e-4


def test_cce_from_logits_sum_against_torch_cross_entropy():
    """[TF] B12 CategoricalCrossentropy(from_logits=True, reduction=SUM) == CrossEntropyLoss(reduction='sum') -- hard (eye)
    labels as sbcnm.py:134 builds them, soft labels, and per-example sample weights (reduction='none' * w)."""
    r = _rng(3)
    B, C = 17, 23
    s = (r.standard_normal((B, C)) * 4).astype(np.float32)
    s[2, 5] = 80.0                                          # a dominant logit (log-sum-exp stability)
    s[3] = -60.0
    eye = np.zeros((B, C), dtype=np.float32)
    eye[np.arange(B), np.arange(B)] = 1.0
    st = torch.from_numpy(s).double()
    want = torch.nn.CrossEntropyLoss(reduction="sum")(st, torch.arange(B)).item()
    assert abs(float(O.categorical_crossentropy_from_logits_sum(eye, s)) - want) <= 1e-6 * abs(want)
    soft = r.random((B, C)).astype(np.float32)
    soft /= soft.sum(1, keepdims=True)
    want = torch.nn.CrossEntropyLoss(reduction="sum")(st, torch.from_numpy(soft).double()).item()
    assert abs(float(O.categorical_crossentropy_from_logits_sum(soft, s)) - want) <= 1e-6 * abs(want)
    w = r.random(B).astype(np.float32)
    want = (torch.nn.CrossEntropyLoss(reduction="none")(st, torch.arange(B)) * torch.from_numpy(w).double()).sum().item()
    assert abs(float(O.categorical_crossentropy_from_logits_sum(eye, s, sample_weight=w)) - want) <= 1e-6 * abs(want)


def test_retrieval_call_plain_branch_against_torch_cross_entropy():
    """Retrieval.call end to end without the optional branches (sbcnm.py:129-151): scores = q c^T, labels = eye, CCE SUM
    == CrossEntropyLoss(sum)(q c^T, arange); with a temperature the scores are divided by it first (sbcnm.py:137-138)."""
    r = _rng(4)
    B, D = 24, 16
    q = r.standard_normal((B, D)).astype(np.float32)
    c = r.standard_normal((B, D)).astype(np.float32)
    sc = torch.from_numpy(q).double() @ torch.from_numpy(c).double().t()
    want = torch.nn.CrossEntropyLoss(reduction="sum")(sc, torch.arange(B)).item()
    assert abs(float(O.retrieval_loss(q, c)) - want) <= 2e-6 * abs(want)
    got_t = T.inbatch_softmax_loss(torch.from_numpy(q).double(), torch.from_numpy(c).double()).item()
    assert abs(got_t - want) <= 1e-10 * abs(want)
    want_t = torch.nn.CrossEntropyLoss(reduction="sum")(sc / 0.4, torch.arange(B)).item()
    assert abs(float(O.retrieval_loss(q, c, temperature=0.4)) - want_t) <= 2e-6 * abs(want_t)


def test_top_k_tie_rule_against_torch_stable_sort():
    """[TF] B13 tf.math.top_k: descending, equal elements -> lower index first == torch.sort(descending=True, stable=True)."""
    r = _rng(5)
    x = r.integers(0, 6, size=(40, 30)).astype(np.float32)          # heavy ties
    x[0] = 1.0
    x[1, ::2] = np.float32(-np.inf)
    for k in (1, 5, 30):
        vals, idx = O.top_k(x, k)
        sv, si = torch.sort(torch.from_numpy(x), dim=1, descending=True, stable=True)
        np.testing.assert_array_equal(idx, si[:, :k].numpy())
        np.testing.assert_array_equal(vals, sv[:, :k].numpy())


def test_in_top_k_rule_against_rank_by_stable_sort():
    """[TF] B14 tf.math.in_top_k: the target is in the top k iff fewer than k entries are STRICTLY greater -- i.e. ties with the
    target never push it out.  Cross-checked through ranks from torch's stable sort: (number of entries strictly greater) =
    position of the first occurrence of the target's value in the descending sort."""
    r = _rng(6)
    p = r.integers(0, 5, size=(50, 12)).astype(np.float32)
    t = r.integers(0, 12, size=50)
    sv, _ = torch.sort(torch.from_numpy(p), dim=1, descending=True, stable=True)
    tv = torch.from_numpy(p)[torch.arange(50), torch.from_numpy(t)]
    first = (sv == tv[:, None]).float().argmax(dim=1)                       # index of the first equal value = #strictly greater
    for k in (1, 3, 12):
        np.testing.assert_array_equal(O.in_top_k(t, p, k), (first < k).numpy())


def _torch_adam_steps(p0, grads, lr, b1, b2, eps_for_step):
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([p], lr=lr, betas=(b1, b2), eps=1e-8)
    for t, g in enumerate(grads, start=1):
        opt.param_groups[0]["eps"] = eps_for_step(t)
        p.grad = g.clone()
        opt.step()
    return p.detach()


def test_adam_dense_step_against_torch_optim_adam_with_the_epsilon_map():
    """[TF] B15 tf.train.AdamOptimizer:  p -= lr sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps)            ("epsilon hat")
    torch.optim.Adam:                    p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
                                           = lr sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps sqrt(1 - b2^t)).
    Same m, v recurrences; the ONLY difference is where epsilon sits: TF's eps equals torch's eps' with
    eps' = eps / sqrt(1 - b2^t).  With that per-step map torch.optim.Adam must reproduce the oracle exactly; with the same
    nominal eps the two differ by a factor of up to 1 / sqrt(1 - b2) = 31.6 in the effective epsilon on step 1."""
    g = torch.Generator().manual_seed(7)
    lr, b1, b2, eps = 0.01, 0.9, 0.999, 1e-8
    p0 = torch.randn(200, generator=g, dtype=torch.float64)
    grads = [torch.randn(200, generator=g, dtype=torch.float64) * (10.0 ** float(-k)) for k in (0, 3, 6, 8, 9)]
    p, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    for t, gr in enumerate(grads, start=1):
        T.adam_dense_step(p, gr, m, v, lr, t, b1, b2, eps)
    want = _torch_adam_steps(p0, grads, lr, b1, b2, lambda t: eps / (1.0 - b2 ** t) ** 0.5)
    np.testing.assert_allclose(p.numpy(), want.numpy(), rtol=1e-12, atol=1e-15)
    # and the difference that the map removes is real: same nominal eps, tiny gradients -> visibly different steps
    same_eps = _torch_adam_steps(p0, grads, lr, b1, b2, lambda t: eps)
    assert float((same_eps - p).abs().max()) > 1e-5 * lr


def test_adam_rows_step_is_tf_adam_on_touched_rows():
    """The row-wise form applies exactly the dense [TF] B15 update to the touched rows and leaves the others alone (the documented
    divergence from TF's non-lazy sparse Adam: untouched rows' moments are not decayed) -- pinned through torch.optim.Adam with
    the epsilon map above, on a table whose untouched rows are excluded from the comparison."""
    g = torch.Generator().manual_seed(8)
    lr, b1, b2, eps = 0.01, 0.9, 0.999, 1e-8
    R, D = 30, 6
    p0 = torch.randn((R, D), generator=g, dtype=torch.float64)
    rows = torch.tensor([3, 7, 7, 12, 29, 3])
    dense_grads = []
    for _ in range(3):
        gr = torch.zeros((R, D), dtype=torch.float64)
        gr.index_add_(0, rows, torch.randn((rows.numel(), D), generator=g, dtype=torch.float64))
        dense_grads.append(gr)
    p, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    for t, gr in enumerate(dense_grads, start=1):
        T.adam_rows_step(p, gr, rows, m, v, lr, t, b1, b2, eps)
    touched = torch.unique(rows)
    want = _torch_adam_steps(p0[touched], [gr[touched] for gr in dense_grads], lr, b1, b2, lambda t: eps / (1.0 - b2 ** t) ** 0.5)
    np.testing.assert_allclose(p[touched].numpy(), want.numpy(), rtol=1e-12, atol=1e-15)
    untouched = torch.tensor([i for i in range(R) if i not in set(touched.tolist())])
    assert torch.equal(p[untouched], p0[untouched]) and float(m[untouched].abs().max()) == 0.0


def test_fm_second_order_against_explicit_pairwise_sum():
    """keras/models/ranking/fm.py:29-33: 0.5 * sum_d ((sum_f x)^2 - sum_f x^2) is the sum over field PAIRS of <x_i, x_j> -- checked
    against the pairwise definition computed by torch.einsum (an independent formula, not a restatement of the closed form)."""
    r = _rng(9)
    x = r.standard_normal((11, 6, 5)).astype(np.float32)
    xt = torch.from_numpy(x).double()
    gram = torch.einsum("bid,bjd->bij", xt, xt)
    want = (gram.sum(dim=(1, 2)) - torch.diagonal(gram, dim1=1, dim2=2).sum(1)) * 0.5
    np.testing.assert_allclose(O.fm_second_order(x).reshape(-1), want.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(T.fm_second_order(xt).reshape(-1).numpy(), want.numpy(), rtol=1e-12, atol=1e-12)


def test_ftrl_step_against_torch_adagrad_and_the_l1_threshold():
    """tf.train.FtrlOptimizer (examples/train_wdl_on_movielens_estimator.py:66-70; oracle/torch_ref.py ftrl_dense_step) had no
    third-party anchor (no FTRL implementation in the image).  Two properties pin it (round 5):

    (1) With l1 = l2 = 0, lr_power = -0.5 and weights that start at ZERO -- the wide / linear part of the reference's WDL starts from
    zeros -- FTRL-Proximal is per-coordinate AdaGrad exactly (McMahan et al. 2013, section 3): the update keeps the invariant
    linear = -w sqrt(accum) / lr, so  w' = w - lr g / sqrt(accum + g^2).  That is torch.optim.Adagrad(lr, eps=0,
    initial_accumulator_value=0.1) -- written by other people -- step for step (TF's default initial accumulator is 0.1).
    (2) The l1 clause: a coordinate whose |linear| stays <= l1 is EXACTLY zero, and above the threshold the weight is the
    soft-thresholded closed form -(linear - sign(linear) l1) / (sqrt(accum) / lr + 2 l2)."""
    torch.manual_seed(0)
    n, lr, a0 = 257, 0.05, 0.1
    p = torch.zeros(n, dtype=torch.float64)
    accum, linear = torch.full((n,), a0, dtype=torch.float64), torch.zeros(n, dtype=torch.float64)
    q = torch.zeros(n, dtype=torch.float64, requires_grad=True)
    opt = torch.optim.Adagrad([q], lr=lr, eps=0.0, initial_accumulator_value=a0, lr_decay=0.0)
    for step in range(6):
        g = torch.randn(n, dtype=torch.float64) * (10.0 ** torch.randint(-3, 2, (n,)).double())
        T.ftrl_dense_step(p, g, accum, linear, lr)
        q.grad = g.clone()
        opt.step()
        np.testing.assert_allclose(p.numpy(), q.detach().numpy(), rtol=1e-12, atol=1e-15, err_msg="step %d" % step)
    # (2) l1 threshold and soft-threshold closed form
    l1, l2 = 0.5, 0.01
    p = torch.zeros(n, dtype=torch.float64)
    accum, linear = torch.full((n,), a0, dtype=torch.float64), torch.zeros(n, dtype=torch.float64)
    small = torch.arange(n) % 2 == 0                           # even coordinates: gradients too small to ever cross l1
    for step in range(4):
        g = torch.where(small, torch.full((n,), 0.1, dtype=torch.float64), torch.randn(n, dtype=torch.float64) * 3 + 4)
        T.ftrl_dense_step(p, g, accum, linear, lr, l1=l1, l2=l2)
        assert bool((linear[small].abs() <= l1).all()) and bool((p[small] == 0).all())
        big = ~small & (linear.abs() > l1)
        want = -(linear[big] - torch.sign(linear[big]) * l1) / (accum[big].sqrt() / lr + 2 * l2)
        np.testing.assert_allclose(p[big].numpy(), want.numpy(), rtol=1e-13)
        assert bool((p[~small & ~big] == 0).all())
