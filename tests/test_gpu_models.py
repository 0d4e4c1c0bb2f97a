"""GPU tests of the reference-shaped classes.  Each test names the reference test it mirrors
(paths relative to the reference root) and checks results against the CPU oracle."""
import io
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tf_semantics as O
from oracle import torch_ref as T


def _movielens_columns(fc, dim=16):
    # examples/train_fm_on_movielens_estimator.py:10-34 (incl. the movie_genres/gender_vocab quirk, App. A2)
    user_id = fc.categorical_column_with_hash_bucket("user_id", 6040)
    user_gender = fc.categorical_column_with_vocabulary_list("user_gender", ["F", "M"])
    user_age = fc.categorical_column_with_vocabulary_list("user_age", [1, 18, 25, 35, 45, 50, 56])
    user_occupation = fc.categorical_column_with_vocabulary_list("user_occupation", list(range(21)))
    movie_id = fc.categorical_column_with_hash_bucket("movie_id", 3952)
    movie_genres = fc.categorical_column_with_vocabulary_list("movie_genres", ["F", "M"])
    base = [user_id, user_gender, user_age, user_occupation, movie_id, movie_genres]
    return [fc.indicator_column(c) for c in base], [fc.embedding_column(c, dimension=dim) for c in base]


def _movielens_batch(rng, B):
    genres = ["Action", "Adventure", "Animation", "Children's", "Comedy", "Crime", "Documentary", "Drama"]
    feats = {
        "user_id": np.array([str(v) for v in rng.integers(1, 6041, size=B)], dtype=object),
        "user_gender": np.array([["F", "M"][v] for v in rng.integers(0, 2, size=B)], dtype=object),
        "user_age": np.array([1, 18, 25, 35, 45, 50, 56])[rng.integers(0, 7, size=B)],
        "user_occupation": rng.integers(0, 21, size=B),
        "movie_id": np.array([str(v) for v in rng.integers(1, 3953, size=B)], dtype=object),
        "movie_genres": [list(rng.choice(genres, size=rng.integers(1, 4), replace=False)) for _ in range(B)],
    }
    labels = (rng.random(B) < 0.575).astype(np.float32)
    return feats, labels


def _oracle_ids(feats):
    B = len(feats["user_age"])
    L = max(len(g) for g in feats["movie_genres"])
    genres = np.full((B, L), "", dtype=object)
    for i, g in enumerate(feats["movie_genres"]):
        genres[i, :len(g)] = g
    g_ids = O.vocab_lookup(genres, ["F", "M"])
    g_ids[genres == ""] = -1
    return [
        O.hash_bucket_strings(feats["user_id"], 6040).reshape(B, 1),
        O.vocab_lookup(feats["user_gender"], ["F", "M"]).reshape(B, 1),
        O.vocab_lookup([int(v) for v in feats["user_age"]], [1, 18, 25, 35, 45, 50, 56]).reshape(B, 1),
        O.vocab_lookup([int(v) for v in feats["user_occupation"]], list(range(21))).reshape(B, 1),
        O.hash_bucket_strings(feats["movie_id"], 3952).reshape(B, 1),
        g_ids,
    ]


KEYS = ["user_id", "user_gender", "user_age", "user_occupation", "movie_id", "movie_genres"]


def test_fm_layer():
    # tests/keras/test_fm.py:17-26
    from deep_recommenders_amd.keras.models.ranking import FM
    rng = np.random.RandomState(1)
    sparse_inputs = rng.randint(0, 2, size=(10, 10)).astype(np.float32)
    embedding_inputs = rng.normal(size=(10, 5, 5)).astype(np.float32)
    x_sum = np.sum(embedding_inputs, axis=1)
    x_square_sum = np.sum(np.power(embedding_inputs, 2), axis=1)
    expected = 0.5 * np.sum(np.power(x_sum, 2) - x_square_sum, axis=1, keepdims=True)
    layer = FM()
    out = layer(sparse_inputs, embedding_inputs)
    np.testing.assert_allclose(out.detach().cpu().numpy(), expected, rtol=1e-6, atol=1e-6)
    lin_only = layer(sparse_inputs)
    assert lin_only.shape == (10, 1) and float(lin_only.abs().max()) == 0.0        # zero-init linear (fm.py:16-20)
    assert layer.get_config() == {}


def test_fm_layer_train():
    # tests/keras/test_fm.py:28-42 — FM -> Dense(1), mse, one fit pass must run and reduce the loss
    from deep_recommenders_amd.keras.models.ranking import FM
    from deep_recommenders_amd import layers as L
    rng = np.random.RandomState(2)
    sp = rng.randint(0, 2, size=(10, 10)).astype(np.float32)
    emb = torch.tensor(rng.uniform(size=(10, 5, 5)).astype(np.float32), device="cuda", requires_grad=True)
    y = torch.tensor(rng.uniform(size=(10, 1)).astype(np.float32), device="cuda")
    layer = FM()
    head_w = torch.nn.Parameter(torch.full((1, 1), 0.5, device="cuda"))
    head_b = torch.nn.Parameter(torch.zeros(1, device="cuda"))
    layer(sp, emb)
    opt = torch.optim.SGD(list(layer.parameters()) + [head_w, head_b], lr=1e-3)
    losses = []
    for _ in range(5):
        opt.zero_grad()
        out = L.mlp(layer(sp, emb), [head_w], [head_b], [0])
        loss = ((out - y) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0]
    assert emb.grad is not None and torch.isfinite(emb.grad).all()


@pytest.mark.parametrize("model_name", ["FactorizationMachine", "DeepFM"])
def test_model_train_and_state_roundtrip(model_name):
    # tests/keras/test_fm.py:67-107 and tests/keras/test_deepfm.py:16-56:
    # hash_bucket(100) columns on keys "1"/"2", dim 16, 100 Adam steps with binary_crossentropy, then
    # save -> load -> identical predictions and identical get_config()
    from deep_recommenders_amd import feature_column as fc
    from deep_recommenders_amd import losses
    from deep_recommenders_amd.keras.models.ranking import FactorizationMachine, DeepFM

    def build():
        base = [fc.categorical_column_with_hash_bucket("user_id", 100),
                fc.categorical_column_with_hash_bucket("movie_id", 100)]
        ind = [fc.indicator_column(c) for c in base]
        emb = [fc.embedding_column(c, dimension=16) for c in base]
        if model_name == "DeepFM":
            return DeepFM(ind, emb, dnn_units_size=[10, 5])
        return FactorizationMachine(ind, emb)

    torch.manual_seed(0)
    model = build()
    train = {"user_id": [["1"]] * 1000, "movie_id": [["2"]] * 1000}
    labels = torch.zeros((1000, 1), device="cuda")            # np.random.randint(0, 1) == all zeros in the reference
    model(train)                                               # build lazily created weights
    opt = torch.optim.Adam(model.parameters())
    first = None
    for _ in range(100):
        opt.zero_grad()
        loss = losses.binary_crossentropy(labels, model(train))
        loss.backward()
        opt.step()
        first = loss.item() if first is None else first
    assert loss.item() < first
    test_data = {"user_id": np.asarray([["1"], ["2"]]), "movie_id": np.asarray([["1"], ["2"]])}
    pred = model.predict(test_data)
    assert pred.shape == (2, 1) and np.isfinite(pred).all()
    # the row trained on ("1","2") labels 0 must have moved below 0.5 for user "1"
    buf = io.BytesIO()
    torch.save(model.state_dict(), buf)
    buf.seek(0)
    loaded = build()
    loaded(test_data)
    loaded.load_state_dict(torch.load(buf))
    np.testing.assert_array_equal(pred, loaded.predict(test_data))
    assert {k: v for k, v in model.get_config().items() if "columns" not in k} == \
           {k: v for k, v in loaded.get_config().items() if "columns" not in k}


def test_keras_deepfm_matches_oracle_on_movielens_shaped_batch():
    """End-to-end parity (north_star): ids bit-identical, loss within 1e-5 relative fp32."""
    from deep_recommenders_amd import feature_column as fc
    from deep_recommenders_amd import losses
    from deep_recommenders_amd.keras.models.ranking import DeepFM, FactorizationMachine
    rng = np.random.default_rng(42)
    B = 256
    feats, labels = _movielens_batch(rng, B)
    ind, emb = _movielens_columns(fc)
    torch.manual_seed(42)
    model = DeepFM(ind, emb, dnn_units_size=[256, 32])          # examples/train_deepfm_on_movielens_keras.py:42
    prob = model(feats)
    slab = model.slab
    slab.lin_w.data.normal_(0, 0.1)
    slab.lin_bias.data.fill_(0.05)
    prob = model(feats)
    # integer path bit-identical
    ids_dev, col_start, row_base = slab.transform(feats, KEYS)
    want_ids = np.concatenate(_oracle_ids(feats), axis=1)
    np.testing.assert_array_equal(ids_dev.cpu().numpy(), want_ids)
    # oracle forward with the same weights
    tabs = [slab.embedding_weights(k).cpu().numpy() for k in KEYS]
    lws = [slab.linear_weights(k).cpu().numpy() for k in KEYS]
    Ws = [w.detach().cpu().numpy() for w in model.dnn_kernels]
    bs = [b.detach().cpu().numpy() for b in model.dnn_biases]
    want_prob, want_logit, _ = O.deepfm_forward(_oracle_ids(feats), tabs, lws, 0.05, Ws, bs, return_parts=True)
    got_logit = model.logits(feats).detach().cpu().numpy()
    np.testing.assert_allclose(got_logit, want_logit, rtol=0, atol=1e-5 * (np.abs(want_logit).max() + 1))
    np.testing.assert_allclose(prob.detach().cpu().numpy(), want_prob, rtol=1e-5, atol=1e-6)
    lab = torch.tensor(labels, device="cuda").reshape(-1, 1)
    for ours, theirs in [(losses.binary_crossentropy(lab, prob), O.keras_binary_crossentropy(labels, want_prob)),
                         (losses.log_loss(lab, prob), O.log_loss(labels, want_prob)),
                         (losses.sigmoid_cross_entropy(lab, model.logits(feats)), O.sigmoid_cross_entropy(labels, want_logit))]:
        assert abs(ours.item() - float(theirs)) <= 1e-5 * abs(float(theirs))
    # FactorizationMachine = same minus the DNN (fm.py:54-64)
    fm_model = FactorizationMachine(ind, emb)
    fm_model.slab.load_state_dict(slab.state_dict())
    want_fm = O.deepfm_forward(_oracle_ids(feats), tabs, lws, 0.05, None, None)
    np.testing.assert_allclose(fm_model(feats).detach().cpu().numpy(), want_fm, rtol=1e-5, atol=1e-6)


def test_deepfm_gradients_match_autograd_oracle():
    from deep_recommenders_amd import feature_column as fc
    from deep_recommenders_amd import losses
    from deep_recommenders_amd.keras.models.ranking import DeepFM
    rng = np.random.default_rng(7)
    B = 200
    feats, labels = _movielens_batch(rng, B)
    ind, emb = _movielens_columns(fc)
    torch.manual_seed(1)
    model = DeepFM(ind, emb, dnn_units_size=[64, 32])           # examples/train_deepfm_on_movielens_estimator.py:40
    model(feats)
    model.slab.lin_w.data.normal_(0, 0.1)
    lab = torch.tensor(labels, device="cuda").reshape(-1, 1)
    loss = losses.sigmoid_cross_entropy(lab, model.logits(feats))
    loss.backward()
    # oracle: fp64 autograd over the torch restatement with the same weights
    slab = model.slab
    ids = torch.tensor(np.concatenate(_oracle_ids(feats), axis=1))
    widths = [a.shape[1] for a in _oracle_ids(feats)]
    cs = [0]
    for w in widths:
        cs.append(cs[-1] + w)
    rb = [slab.base[k] for k in KEYS]
    tt = slab.table.detach().cpu().double().requires_grad_(True)
    tl = slab.lin_w.detach().cpu().double().requires_grad_(True)
    tb = slab.lin_bias.detach().cpu().double().requires_grad_(True)
    Ws = [w.detach().cpu().double().requires_grad_(True) for w in model.dnn_kernels]
    bs = [b.detach().cpu().double().requires_grad_(True) for b in model.dnn_biases]
    logit = T.deepfm_logit(tt, tl, tb, ids, cs, rb, Ws, bs)
    lo = T.sigmoid_cross_entropy(torch.tensor(labels, dtype=torch.float64), logit)
    lo.backward()
    assert abs(loss.item() - lo.item()) <= 1e-5 * abs(lo.item())

    def close(a, b, name):
        a, b = a.cpu().numpy(), b.numpy()
        np.testing.assert_allclose(a, b, rtol=1e-3, atol=2e-6 * (np.abs(b).max() + 1e-3) + 1e-9, err_msg=name)

    close(slab.table.grad, tt.grad, "table")
    close(slab.lin_w.grad, tl.grad, "lin_w")
    close(slab.lin_bias.grad, tb.grad, "lin_bias")
    for i, (w, b) in enumerate(zip(model.dnn_kernels, model.dnn_biases)):
        close(w.grad, Ws[i].grad, "W%d" % i)
        close(b.grad, bs[i].grad, "b%d" % i)


def test_cross_layer_api():
    # tests/keras/test_dcn.py:16-45
    from deep_recommenders_amd.keras.models.ranking.dcn import Cross
    x0 = np.asarray([[0.1, 0.2, 0.3]]).astype(np.float32)
    x = np.asarray([[0.4, 0.5, 0.6]]).astype(np.float32)
    cross = Cross(projection_dim=None, kernel_init="ones")
    np.testing.assert_allclose(cross(x0, x).detach().cpu().numpy(), np.asarray([[0.55, 0.8, 1.05]]), rtol=1e-6)
    # stacking pattern + gradient flows to both layers (test_dcn.py:27-32)
    torch.manual_seed(0)
    c1, c2 = Cross(), Cross(diag_scale=0.1)
    inp = torch.tensor(np.random.default_rng(0).uniform(size=(10, 13)).astype(np.float32), device="cuda")
    x1 = c1(inp, inp)
    x2 = c2(inp, x1)
    want1 = O.cross(inp.cpu().numpy(), inp.cpu().numpy(), c1.kernel.detach().cpu().numpy(), c1.bias.detach().cpu().numpy())
    want2 = O.cross(inp.cpu().numpy(), want1, c2.kernel.detach().cpu().numpy(), c2.bias.detach().cpu().numpy(), 0.1)
    np.testing.assert_allclose(x2.detach().cpu().numpy(), want2, rtol=1e-5, atol=1e-6)
    x2.sum().backward()
    # gradient oracle
    tk1 = c1.kernel.detach().cpu().double().requires_grad_(True)
    tk2 = c2.kernel.detach().cpu().double().requires_grad_(True)
    tb1 = c1.bias.detach().cpu().double().requires_grad_(True)
    ti = inp.cpu().double()
    o = T.cross(ti, T.cross(ti, ti, tk1, tb1), tk2, c2.bias.detach().cpu().double(), 0.1)
    o.sum().backward()
    np.testing.assert_allclose(c1.kernel.grad.cpu().numpy(), tk1.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(c2.kernel.grad.cpu().numpy(), tk2.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(c1.bias.grad.cpu().numpy(), tb1.grad.numpy(), rtol=1e-4, atol=1e-5)
    # error conventions (dcn.py:32-33,48-53,75-78)
    with pytest.raises(ValueError):
        Cross()(np.zeros((1, 3), np.float32), np.zeros((1, 4), np.float32))
    with pytest.raises(ValueError):
        Cross(projection_dim=5)(np.zeros((1, 6), np.float32))
    with pytest.raises(AssertionError):
        Cross(diag_scale=-1.0)
    cfg = Cross(projection_dim=2, diag_scale=0.5).get_config()
    assert set(cfg) == {"projection_dim", "diag_scale", "use_bias", "kernel_init", "kernel_regu", "bias_init", "bias_regu"}
    # low-rank form (dcn.py:83)
    lr_layer = Cross(projection_dim=3)
    xx = np.random.default_rng(1).standard_normal((4, 8)).astype(np.float32)
    out = lr_layer(xx)
    want = O.cross(xx, None, lr_layer.kernel.detach().cpu().numpy(), lr_layer.bias.detach().cpu().numpy(), 0.0,
                   kernel_u=lr_layer.kernel_u.detach().cpu().numpy())
    np.testing.assert_allclose(out.detach().cpu().numpy(), want, rtol=1e-5, atol=1e-6)


def test_estimator_fm_and_deepfm():
    # tests/estimator/test_fm.py:18-26 (shape) + estimator FM/DeepFM surface (estimator/.../fm.py:29-56, ranking/deepfm.py)
    from deep_recommenders_amd import feature_column as fc
    from deep_recommenders_amd.estimator.models.feature_interaction import fm, FM, dnn
    from deep_recommenders_amd.estimator.models.ranking import DeepFM
    y = fm(torch.randn(10, 2, 3))
    assert tuple(y.shape) == (10, 1)
    with pytest.raises(ValueError):
        fm(torch.randn(10, 6))
    rng = np.random.default_rng(3)
    feats, labels = _movielens_batch(rng, 64)
    ind, emb = _movielens_columns(fc)
    torch.manual_seed(3)
    model = FM(ind, emb)
    logits = model(feats)
    assert tuple(logits.shape) == (64, 1) and len(model.embeddings) == 6 and tuple(model.embeddings[0].shape) == (64, 16)
    slab = model.slab
    tabs = [slab.embedding_weights(k).cpu().numpy() for k in KEYS]
    lws = [slab.linear_weights(k).cpu().numpy() for k in KEYS]
    _, want_logit, _ = O.deepfm_forward(_oracle_ids(feats), tabs, lws, 0.0, None, None, return_parts=True)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), want_logit, rtol=1e-5, atol=1e-5)   # logits, no sigmoid (fm.py:56)
    dm = DeepFM(ind, emb, dnn_units=[64, 32])
    p = dm(feats)
    assert tuple(p.shape) == (64, 1) and float(p.min()) > 0 and float(p.max()) < 1
    with pytest.raises(TypeError):
        dnn(torch.randn(4, 8), [4, 1], batch_normalization=True)      # reference quirk, dnn.py:23-24


@pytest.mark.parametrize("optimizer,D,V", [("sgd", 16, 1000), ("adam", 16, 1000), ("sgd", 64, 3000)])
def test_engine_next_batch_prefetch_is_bit_identical(optimizer, D, V):
    """train_step(..., next_keys=) runs the next batch's K1 + slot sort beside this step's K4 (double-buffered ids / plan).  The
    trained parameters must be bit-identical to the engine that hashes and sorts every batch in its own step (the sorted backward
    is deterministic), also when a prefetched batch is NOT the one that comes next, and when the keys tensor is modified in
    between.  D = 64 takes the fused first layer (dr_bf3_emb_linear_fwd reads the prefetched ids)."""
    from deep_recommenders_amd.engine import DeepFMEngine
    F, B, Nd = 4, 2304, 3
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    batches = [(torch.randint(0, 10**12, (B, F), device="cuda", generator=g), torch.rand((B, Nd), device="cuda", generator=g),
                (torch.rand(B, device="cuda", generator=g) < 0.3).float()) for _ in range(4)]
    engs = [DeepFMEngine(F, V, D, [256, 16], B, num_dense=Nd, lr=0.05 if optimizer == "sgd" else 0.01, seed=3, lin_init_std=0.1,
                         optimizer=optimizer) for _ in range(2)]
    order = [0, 1, 2, 3, 1, 0]
    for n, i in enumerate(order):
        engs[0].train_step(*batches[i])
        nxt = order[n + 1] if n + 1 < len(order) else None
        if n == 2:
            nxt = 0                                     # announced batch 0, batch 1 comes: the prefetched plan must be dropped
        engs[1].train_step(*batches[i], next_keys=None if nxt is None else batches[nxt][0],
                           next_dense=None if nxt is None else batches[nxt][1])
        if n == 0:
            # the announced tensor is modified before it is used: its version changes, the prefetched ids are stale
            batches[1][0].add_(1)
    torch.cuda.synchronize()
    assert engs[1].prefetch_plan
    assert torch.equal(engs[0].table, engs[1].table) and torch.equal(engs[0].lin_w, engs[1].lin_w)
    assert torch.equal(engs[0].flat_params, engs[1].flat_params)
    assert engs[0].loss.item() == engs[1].loss.item()


@pytest.mark.parametrize("switch,value", [("DR_REDUCE_SIDE", "1"), ("DR_REDUCE_SIDE", "2"), ("DR_PREFETCH_EARLY", "0"), ("DR_PREFETCH_EARLY", "1")])
def test_engine_schedule_switches_are_bit_identical(switch, value, monkeypatch):
    """Round 4's measured-and-rejected schedules stay correct: the step's small reduce kernels on the side stream
    (dr_tower_head_fwd_bwd_parts / dr_linear_bwd_narrow_parts / dr_bf3_wgrad_emb_parts: the two halves of each call on two streams),
    and the next batch's hash + plan issued at the start of the step (round 5's default) or beside K4 (rounds 2-4) instead of behind the
    fused first layer (round 6).  Same kernels, same arithmetic order: parameters and losses
    must be bit-identical to the default schedule over prefetched steps (D = 64: fused first layer, gathering wgrad, planes)."""
    from deep_recommenders_amd.engine import DeepFMEngine
    F, B, Nd, D, V = 4, 2304, 3, 64, 3000
    g = torch.Generator(device="cuda")
    g.manual_seed(12)
    batches = [(torch.randint(0, 10**12, (B, F), device="cuda", generator=g), torch.rand((B, Nd), device="cuda", generator=g),
                (torch.rand(B, device="cuda", generator=g) < 0.3).float()) for _ in range(3)]

    def run():
        eng = DeepFMEngine(F, V, D, [256, 16], B, num_dense=Nd, lr=0.05, seed=3, lin_init_std=0.1)
        losses = []
        for n in range(5):
            k, d, l = batches[n % 3]
            nk, nd = batches[(n + 1) % 3][0], batches[(n + 1) % 3][1]
            losses.append(float(eng.train_step(k, d, l, next_keys=nk, next_dense=nd).item()))
        torch.cuda.synchronize()
        return eng, losses
    base, l0 = run()
    monkeypatch.setenv(switch, value)
    alt, l1 = run()
    if switch == "DR_REDUCE_SIDE":
        # (the wgrad's reduce is deferred only in the bf16x3 split: in the f16x2 mode its second half would derive its scales from the
        # table's amax record while K4 raises it on the other stream -- engine.py)
        assert alt.reduce_side and alt.reduce_side_wgrad == (value == "1" and not alt.h2)
    else:
        assert base.prefetch_early and base.prefetch_after_fwd              # the default since round 6: issued behind the fused first layer
        assert alt.prefetch_early == (value != "0") and alt.prefetch_after_fwd == (value == "2")
    assert l0 == l1
    assert torch.equal(base.table, alt.table) and torch.equal(base.lin_w, alt.lin_w) and torch.equal(base.flat_params, alt.flat_params)


@pytest.mark.parametrize("optimizer", ["sgd", "adam"])
def test_engine_one_pass_tail_tracks_the_two_launch_tail(optimizer, monkeypatch):
    """dr_tower_tail_fused (the default since round 5: head + narrow backward of the last hidden layer in one pass over h) against
    DR_FUSE_TAIL=0 (dr_tower_head_fwd_bwd + dr_linear_bwd_narrow, rounds 1-4) over prefetched steps: the same arithmetic up to the order of
    the 256-long sums of the head's product -- losses to 1e-6 relative, parameters to fp32 noise; both paths stay under test."""
    from deep_recommenders_amd.engine import DeepFMEngine
    F, B, Nd, D, V = 4, 2304, 3, 64, 3000
    g = torch.Generator(device="cuda")
    g.manual_seed(12)
    batches = [(torch.randint(0, 10**12, (B, F), device="cuda", generator=g), torch.rand((B, Nd), device="cuda", generator=g),
                (torch.rand(B, device="cuda", generator=g) < 0.3).float()) for _ in range(3)]

    def run():
        eng = DeepFMEngine(F, V, D, [256, 16], B, num_dense=Nd, lr=0.05 if optimizer == "sgd" else 0.002, seed=3, lin_init_std=0.1,
                           optimizer=optimizer)
        losses = []
        for n in range(5):
            k, d, l = batches[n % 3]
            losses.append(float(eng.train_step(k, d, l, next_keys=batches[(n + 1) % 3][0], next_dense=batches[(n + 1) % 3][1]).item()))
        torch.cuda.synchronize()
        return eng, losses
    one, l1 = run()
    assert one.fuse_tail and one._tail_done
    monkeypatch.setenv("DR_FUSE_TAIL", "0")
    two, l2 = run()
    assert not two.fuse_tail and not two._tail_done and two._head_done
    for a, b in zip(l1, l2):
        assert abs(a - b) <= 1e-6 * abs(b), (l1, l2)
    tol = 2e-5 if optimizer == "sgd" else 2e-3       # (Adam divides by sqrt(v): a last-bit difference in a tiny gradient moves a whole step)
    assert (one.table - two.table).abs().max().item() <= tol * two.table.abs().max().item()
    assert (one.flat_params - two.flat_params).abs().max().item() <= tol * two.flat_params.abs().max().item()


def test_engine_train_step_matches_oracle():
    """The fused engine step (what bench.py times) against the host restatement: loss and updated weights."""
    from deep_recommenders_amd.engine import DeepFMEngine
    F, V, D, B, Nd = 5, 1000, 16, 512, 3
    lr = 0.05
    eng = DeepFMEngine(F, V, D, [32, 16], B, num_dense=Nd, lr=lr, seed=1, lin_init_std=0.1)
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    keys = torch.randint(0, 10**12, (B, F), device="cuda", generator=g)
    dense = torch.rand((B, Nd), device="cuda", generator=g)
    labels = (torch.rand(B, device="cuda", generator=g) < 0.25).float()
    # host copies of the parameters before the step
    table0, lin0 = eng.table.cpu().clone(), eng.lin_w.cpu().clone()
    Ws0 = [w.cpu().clone().contiguous() for w in eng.Ws]
    bs0 = [b.cpu().clone() for b in eng.bs]
    loss = eng.train_step(keys, dense, labels).item()
    # oracle step
    ids = np.stack([O.hash_bucket_i64(keys[:, f].cpu().numpy(), V) for f in range(F)], axis=1)
    np.testing.assert_array_equal(eng.ids.cpu().numpy(), ids)
    params = (table0.clone(), lin0.clone(), torch.zeros(()), [w.clone() for w in Ws0], [b.clone() for b in bs0])
    # the oracle applies SGD to bias too; mirror by hand
    tb = torch.zeros(1, requires_grad=True)
    tt = table0.clone().requires_grad_(True)
    tl = lin0.clone().requires_grad_(True)
    Ws = [w.clone().requires_grad_(True) for w in Ws0]
    bs = [b.clone().requires_grad_(True) for b in bs0]
    rb = [f * V for f in range(F)]
    logit = T.deepfm_logit(tt, tl, tb, torch.tensor(ids), list(range(F + 1)), rb, Ws, bs, dense.cpu())
    lo = T.sigmoid_cross_entropy(labels.cpu(), logit)
    lo.backward()
    assert abs(loss - lo.item()) <= 1e-5 * abs(lo.item())
    np.testing.assert_allclose(eng.table.cpu().numpy(), (table0 - lr * tt.grad).numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(eng.lin_w.cpu().numpy(), (lin0 - lr * tl.grad).numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(eng.lin_bias.cpu().numpy(), (-lr * tb.grad).numpy(), rtol=1e-4, atol=1e-7)
    for i in range(len(Ws)):
        np.testing.assert_allclose(eng.Ws[i].cpu().numpy(), (Ws0[i] - lr * Ws[i].grad).numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(eng.bs[i].cpu().numpy(), (bs0[i] - lr * bs[i].grad).numpy(), rtol=1e-4, atol=1e-6)


def _store_params(store, scope="dnn"):
    ks, bs, i = [], [], 0
    while True:
        name = ("%s/dense%s" % (scope, "" if i == 0 else "_%d" % i)).replace("/", "__")
        if name + "__kernel" not in store.vars:
            return ks, bs
        ks.append(store.vars[name + "__kernel"].detach().cpu().numpy())
        bs.append(store.vars[name + "__bias"].detach().cpu().numpy())
        i += 1


def test_estimator_wdl_and_fnn(tmp_path):
    """WDL (wide_and_deep.py:29-48) and FNN (fnn.py:50-90) on the MovieLens-shaped batch against their
    oracle restatements; gradients reach the wide weights; FNN warm-starts from an FM through the reference's TF variable
    names (object, dict and .npz file)."""
    from deep_recommenders_amd import feature_column as fc
    from deep_recommenders_amd.estimator.models.feature_interaction import FM
    from deep_recommenders_amd.estimator.models.ranking import WDL, FNN
    rng = np.random.default_rng(4)
    feats, labels = _movielens_batch(rng, 80)
    ind, emb = _movielens_columns(fc)
    ids = _oracle_ids(feats)
    # ---- WDL ----
    torch.manual_seed(5)
    wdl = WDL(ind, emb, dnn_units=[32, 16])
    with torch.no_grad():                                   # non-trivial first-order weights (zero-initialised otherwise)
        wdl.slab.lin_w.normal_(0, 0.1)
        wdl.slab.lin_bias.fill_(0.05)
    p = wdl(feats)
    assert tuple(p.shape) == (80, 1)
    tabs = [wdl.slab.embedding_weights(k).cpu().numpy() for k in KEYS]
    lws = [wdl.slab.linear_weights(k).cpu().numpy() for k in KEYS]
    ks, bs = _store_params(wdl.store)
    want = O.wdl_forward(ids, tabs, lws, 0.05, ks, bs)
    np.testing.assert_allclose(p.detach().cpu().numpy(), want, rtol=1e-5, atol=2e-6)
    loss = torch.nn.functional.binary_cross_entropy(p, torch.tensor(labels, device="cuda").reshape(-1, 1))
    loss.backward()
    assert float(wdl.slab.lin_w.grad.abs().sum()) > 0 and float(wdl.slab.table.grad.abs().sum()) > 0
    assert float(wdl.slab.lin_bias.grad.abs().sum()) > 0
    names = wdl.export_variables()
    assert "wide/linear_model/user_id_indicator/weights" in names and "wide/linear_model/bias_weights" in names
    assert "deep/input_layer/movie_id_embedding/embedding_weights" in names and "deep/dense/kernel" in names and "deep/dense_2/bias" in names
    assert names["wide/linear_model/user_id_indicator/weights"].shape == (6040, 1)
    wdl2 = WDL(ind, emb, dnn_units=[32, 16])
    wdl2(feats)                                             # materialise the dense variables
    wdl2.import_variables({k + ":0": v for k, v in names.items()})          # TF's ':0' suffix accepted
    np.testing.assert_array_equal(wdl2(feats).detach().cpu().numpy(), wdl(feats).detach().cpu().numpy())
    # ---- FM -> FNN warm start ----
    torch.manual_seed(6)
    fm_model = FM(ind, emb)
    with torch.no_grad():
        fm_model.slab.lin_w.normal_(0, 0.1)
        fm_model.slab.lin_bias.fill_(-0.2)
    fm_vars = fm_model.export_variables()
    assert set(n.split("/")[0] for n in fm_vars) == {"linear", "factorized"}
    npz = str(tmp_path / "fm_variables.npz")
    fm_model.save_variables(npz)
    outs = []
    for src in (fm_model, fm_vars, npz):
        torch.manual_seed(7)                                # same DNN initialisation for the three sources
        fnn = FNN(ind, emb, src, dnn_units=[24, 8])
        lin_v, fac_v = fnn.warm_up()
        assert set(lin_v) == set(KEYS) | {"bias"} and set(fac_v) == set(KEYS) and fac_v["user_id"].shape == (6040, 16)
        q = fnn(feats)
        outs.append(q.detach().cpu().numpy())
        np.testing.assert_array_equal(fnn.slab.table.detach().cpu().numpy(), fm_model.slab.table.detach().cpu().numpy())
        np.testing.assert_array_equal(fnn.slab.lin_w.detach().cpu().numpy(), fm_model.slab.lin_w.detach().cpu().numpy())
    np.testing.assert_array_equal(outs[0], outs[1])
    np.testing.assert_array_equal(outs[0], outs[2])
    tabs = [fm_model.slab.embedding_weights(k).cpu().numpy() for k in KEYS]
    lws = [fm_model.slab.linear_weights(k).cpu().numpy() for k in KEYS]
    ks, bs = _store_params(fnn.store)
    assert ks[0].shape == (1 + 6 + 6 * 16, 24)              # [bias | 6 first-order | 6 x 16 embeddings]  (fnn.py:83)
    want = O.fnn_forward(ids, tabs, lws, [-0.2], ks, bs)
    np.testing.assert_allclose(outs[0], want, rtol=1e-5, atol=2e-6)
    q.sum().backward()                                      # the warm-started first-order weights keep training (fnn.py:57-60)
    assert float(fnn.slab.lin_w.grad.abs().sum()) > 0 and float(fnn.slab.table.grad.abs().sum()) > 0


def test_wdl_example_training_recipe_ftrl_wide_adam_deep():
    """examples/train_wdl_on_movielens_estimator.py:60-77 of the reference: log_loss, FtrlOptimizer(0.01, l1=0.5) on the
    "wide" variables, AdamOptimizer(0.01) on the "deep" variables -- 3 steps with deep_recommenders_amd.optim against a
    float64 restatement (oracle forward + autograd, T.ftrl_dense_step / T.adam_dense_step)."""
    from deep_recommenders_amd import feature_column as fc, optim, losses
    from deep_recommenders_amd.estimator.models.ranking import WDL
    rng = np.random.default_rng(8)
    ind, emb = _movielens_columns(fc)
    torch.manual_seed(11)
    wdl = WDL(ind, emb, dnn_units=[16, 8])
    feats, labels = _movielens_batch(rng, 64)
    wdl(feats)                                                           # materialise the dense variables
    wide = [wdl.slab.lin_w, wdl.slab.lin_bias]
    deep = [wdl.slab.table] + [p for p in wdl.store.parameters()]
    opt_w = optim.Ftrl(wide, 0.01, l1_regularization_strength=0.5)
    opt_d = optim.Adam(deep, 0.01, epsilon=1e-8)
    # float64 mirror of every variable and optimizer slot
    ref = {id(p): p.detach().cpu().double().clone() for p in wide + deep}
    acc = {id(p): torch.full_like(ref[id(p)], 0.1) for p in wide}
    lin = {id(p): torch.zeros_like(ref[id(p)]) for p in wide}
    m = {id(p): torch.zeros_like(ref[id(p)]) for p in deep}
    v = {id(p): torch.zeros_like(ref[id(p)]) for p in deep}
    names = sorted(k for k in wdl.store.vars.keys())
    for step in range(1, 4):
        feats, labels = _movielens_batch(rng, 64)
        y = torch.tensor(labels, device="cuda").reshape(-1, 1)
        opt_w.zero_grad(); opt_d.zero_grad()
        p = wdl(feats)
        loss = losses.log_loss(y, p)                                     # :60
        loss.backward()
        opt_w.step(); opt_d.step()
        # ---- restatement ----
        ids = _oracle_ids(feats)
        tvars = {k: t.clone().requires_grad_(True) for k, t in ref.items()}
        tab, lw, lb = tvars[id(wdl.slab.table)], tvars[id(wdl.slab.lin_w)], tvars[id(wdl.slab.lin_bias)]
        base = [wdl.slab.base[k] for k in KEYS]
        embs, first = [], lb.clone()
        for f, idf in enumerate(ids):
            idt = torch.tensor(idf)
            mask = (idt >= 0).double()
            rows = base[f] + idt.clamp(min=0)
            e = (tab[rows] * mask[..., None]).sum(1) / mask.sum(1).clamp(min=1)[:, None]      # [TF] B5 mean combiner
            embs.append(e)
            first = first + (lw[rows] * mask).sum(1)
        x = torch.cat(embs, dim=1)
        nl = len(names) // 2
        for i in range(nl):
            nm = "dnn__dense" + ("" if i == 0 else "_%d" % i)
            x = x @ tvars[id(wdl.store.vars[nm + "__kernel"])] + tvars[id(wdl.store.vars[nm + "__bias"])]
            if i < nl - 1:
                x = torch.relu(x)
        prob = torch.sigmoid(first.reshape(-1, 1) + x)
        lo = T.log_loss(torch.tensor(labels, dtype=torch.float64).reshape(-1, 1), prob)
        lo.backward()
        assert abs(loss.item() - lo.item()) <= 1e-5 * abs(lo.item())
        for pw in wide:
            T.ftrl_dense_step(ref[id(pw)], tvars[id(pw)].grad, acc[id(pw)], lin[id(pw)], 0.01, -0.5, 0.5, 0.0)
        for pd in deep:
            T.adam_dense_step(ref[id(pd)], tvars[id(pd)].grad, m[id(pd)], v[id(pd)], 0.01, step, eps=1e-8)
    for pw in wide:                                                      # FTRL with l1 = 0.5 keeps tiny weights at exactly 0
        got, want = pw.detach().cpu().numpy(), ref[id(pw)].numpy()
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-6)
        assert (got == 0).mean() > 0.5
    for pd in deep:
        np.testing.assert_allclose(pd.detach().cpu().numpy(), ref[id(pd)].numpy(), rtol=0, atol=2e-2 * 0.01)


def test_tfrecord_batch_through_feature_columns_and_fm(tmp_path):
    """Input side -> hot path: a batch parsed by the native TFRecord / tf.Example reader (bytes features) through the
    reference-shaped feature columns and FM model equals the oracle evaluated on the same strings."""
    from oracle import tfrecord_py as W
    from deep_recommenders_amd import feature_column as fc
    from deep_recommenders_amd.datasets import MovielensRanking
    from deep_recommenders_amd.estimator.models.feature_interaction import FM
    rng = np.random.default_rng(12)
    feats0, _ = _movielens_batch(rng, 96)
    rows = [W.movielens_example(feats0["user_id"][i].encode(), feats0["movie_id"][i].encode(), int(rng.integers(1, 6)), 978300000 + i,
                                feats0["user_gender"][i].encode(), int(feats0["user_age"][i]), int(feats0["user_occupation"][i]),
                                b"94110", b"Title %d" % i, [g.encode() for g in feats0["movie_genres"][i]]) for i in range(96)]
    path = str(tmp_path / "movielens.tfrecords")
    W.write_tfrecords(path, [W.encode_example(r) for r in rows])
    feats, labels = next(iter(MovielensRanking(epochs=1, batch_size=96, filename=path).input_fn()))
    assert labels.shape == (96, 1) and isinstance(feats["user_id"][0], bytes)
    ind, emb = _movielens_columns(fc)
    torch.manual_seed(3)
    model = FM(ind, emb)
    logits = model(feats)                                                  # bytes in, as the reader produced them
    slab = model.slab
    tabs = [slab.embedding_weights(k).cpu().numpy() for k in KEYS]
    lws = [slab.linear_weights(k).cpu().numpy() for k in KEYS]
    _, want_logit, _ = O.deepfm_forward(_oracle_ids(feats0), tabs, lws, 0.0, None, None, return_parts=True)   # str oracle
    np.testing.assert_allclose(logits.detach().cpu().numpy(), want_logit, rtol=1e-5, atol=1e-5)


def test_dcn_engine_train_step_matches_oracle():
    """DCNEngine (config 4 shape in small: cross stack -> MLP -> sigmoid CE, fused SGD everywhere) against the host
    restatement: T.cross / dense layers under autograd, plain SGD."""
    from deep_recommenders_amd.dcn_engine import DCNEngine
    F, V, D, B, Nd, lr = 4, 60, 8, 96, 3, 0.05
    eng = DCNEngine(F, V, D, 2, [16, 8], B, num_dense=Nd, lr=lr, diag_scale=0.1, seed=2)
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    keys = torch.randint(0, 10**12, (B, F), device="cuda", generator=g)
    dense = torch.rand((B, Nd), device="cuda", generator=g)
    labels = (torch.rand(B, device="cuda", generator=g) < 0.3).float()
    with torch.no_grad():
        for b in eng.cross_b:
            b.normal_(0, 0.05, generator=g)
        # a non-zero output bias: with b = 0 a row whose last hidden layer is all-dead has logit exactly 0, where the loss
        # formula max(x,0) - xz + log1p(exp(-|x|)) has kinks and autograd (torch: 1 - z, TF: -z) disagrees with the analytic
        # sigmoid(0) - z = 0.5 - z the kernel uses
        eng.bs[-1].fill_(0.03)
    table0 = eng.table.cpu().double().clone()
    cW0 = [w.cpu().double().clone() for w in eng.cross_W]
    cb0 = [b.cpu().double().clone() for b in eng.cross_b]
    Ws0 = [w.cpu().double().clone() for w in eng.Ws]
    bs0 = [b.cpu().double().clone() for b in eng.bs]
    loss = eng.train_step(keys, dense, labels).item()
    ids = np.stack([O.hash_bucket_i64(keys[:, f].cpu().numpy(), V) for f in range(F)], axis=1)
    np.testing.assert_array_equal(eng.ids.cpu().numpy(), ids)
    tt = table0.clone().requires_grad_(True)
    cW = [w.clone().requires_grad_(True) for w in cW0]
    cb = [b.clone().requires_grad_(True) for b in cb0]
    Ws = [w.clone().requires_grad_(True) for w in Ws0]
    bs = [b.clone().requires_grad_(True) for b in bs0]
    rows = torch.tensor(ids + np.arange(F)[None, :] * V)
    x0 = torch.cat([tt[rows].reshape(B, F * D), dense.cpu().double()], dim=1)
    x = x0
    for W, b in zip(cW, cb):
        x = T.cross(x0, x, W, b, 0.1)                                          # dcn.py:81-88
    for i, (W, b) in enumerate(zip(Ws, bs)):
        x = x @ W + b
        if i < len(Ws) - 1:
            x = torch.relu(x)
    lo = T.sigmoid_cross_entropy(labels.cpu().double(), x.reshape(-1))
    lo.backward()
    assert abs(loss - lo.item()) <= 1e-5 * abs(lo.item())
    np.testing.assert_allclose(eng.table.cpu().numpy(), (table0 - lr * tt.grad).numpy(), rtol=1e-4, atol=2e-6)
    for got, w0, w in zip(eng.cross_W, cW0, cW):
        np.testing.assert_allclose(got.cpu().numpy(), (w0 - lr * w.grad).numpy(), rtol=1e-4, atol=2e-6)
    for got, b0, b in zip(eng.cross_b, cb0, cb):
        np.testing.assert_allclose(got.cpu().numpy(), (b0 - lr * b.grad).numpy(), rtol=1e-4, atol=2e-6)
    for got, w0, w in zip(eng.Ws, Ws0, Ws):
        np.testing.assert_allclose(got.cpu().numpy(), (w0 - lr * w.grad).numpy(), rtol=1e-4, atol=2e-6)
    for got, b0, b in zip(eng.bs, bs0, bs):
        np.testing.assert_allclose(got.cpu().numpy(), (b0 - lr * b.grad).numpy(), rtol=1e-4, atol=2e-6)


def test_engine_adam_steps_match_oracle():
    """optimizer="adam": 3 engine steps (fused row-wise Adam in K4 + dense Adam over the flat parameter buffer) against the
    host restatement ([TF] B15 formulas; tables updated on touched rows).  Parameters to 2 % of one Adam step."""
    from deep_recommenders_amd.engine import DeepFMEngine
    F, V, D, B, Nd = 5, 300, 16, 512, 3
    lr = 0.01
    eng = DeepFMEngine(F, V, D, [32, 16], B, num_dense=Nd, lr=lr, seed=1, lin_init_std=0.1, optimizer="adam")
    g = torch.Generator(device="cuda")
    g.manual_seed(6)
    tt, tl, tb = eng.table.cpu().double(), eng.lin_w.cpu().double(), torch.zeros(1, dtype=torch.float64)
    Ws = [w.cpu().double().contiguous() for w in eng.Ws]
    bs = [b.cpu().double() for b in eng.bs]
    zl = lambda t: torch.zeros_like(t)
    mt, vt, ml, vl, mb, vb = zl(tt), zl(tt), zl(tl), zl(tl), zl(tb), zl(tb)
    mW, vW, mB, vB = [zl(w) for w in Ws], [zl(w) for w in Ws], [zl(b) for b in bs], [zl(b) for b in bs]
    rb = [f * V for f in range(F)]
    for step in range(1, 4):
        keys = torch.randint(0, 10**12, (B, F), device="cuda", generator=g)
        dense = torch.rand((B, Nd), device="cuda", generator=g)
        labels = (torch.rand(B, device="cuda", generator=g) < 0.25).float()
        loss = eng.train_step(keys, dense, labels).item()
        ids = np.stack([O.hash_bucket_i64(keys[:, f].cpu().numpy(), V) for f in range(F)], axis=1)
        a = [t.clone().requires_grad_(True) for t in (tt, tl, tb)]
        aW = [w.clone().requires_grad_(True) for w in Ws]
        aB = [b.clone().requires_grad_(True) for b in bs]
        logit = T.deepfm_logit(a[0], a[1], a[2], torch.tensor(ids), list(range(F + 1)), rb, aW, aB, dense.cpu().double())
        lo = T.sigmoid_cross_entropy(labels.cpu().double(), logit)
        lo.backward()
        assert abs(loss - lo.item()) <= 2e-5 * abs(lo.item())
        rows = torch.tensor(ids + np.array(rb)[None, :]).reshape(-1)
        T.adam_rows_step(tt, a[0].grad, rows, mt, vt, lr, step)
        T.adam_rows_step(tl, a[1].grad, rows, ml, vl, lr, step)
        T.adam_dense_step(tb, a[2].grad, mb, vb, lr, step)
        for i in range(len(Ws)):
            T.adam_dense_step(Ws[i], aW[i].grad, mW[i], vW[i], lr, step)
            T.adam_dense_step(bs[i], aB[i].grad, mB[i], vB[i], lr, step)
    tol = 2e-2 * lr
    np.testing.assert_allclose(eng.table.cpu().numpy(), tt.numpy(), rtol=0, atol=tol)
    np.testing.assert_allclose(eng.lin_w.cpu().numpy(), tl.numpy(), rtol=0, atol=tol)
    np.testing.assert_allclose(eng.lin_bias.cpu().numpy(), tb.numpy(), rtol=0, atol=tol)
    for i in range(len(Ws)):
        np.testing.assert_allclose(eng.Ws[i].cpu().numpy(), Ws[i].numpy(), rtol=0, atol=tol)
        np.testing.assert_allclose(eng.bs[i].cpu().numpy(), bs[i].numpy(), rtol=0, atol=tol)


@pytest.mark.parametrize("alias_world1", [False, True])
def test_sharded_engine_world1_matches_unsharded_engine(alias_world1):
    """HIP prims through the exchange plan: steps must equal the plain engine's.  alias_world1=False: every exchange through an
    RCCL process group of size 1; True (the default of a one-rank job): the buffers alias and no collective is issued."""
    import torch.distributed as dist
    from deep_recommenders_amd.engine import DeepFMEngine
    from deep_recommenders_amd.sharded import ShardedDeepFMEngine
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        F, V, D, B, Nd = 6, 3000, 16, 768, 3
        ref = DeepFMEngine(F, V, D, [32, 16], B, num_dense=Nd, lr=0.05, seed=9, lin_init_std=0.1)
        sh = ShardedDeepFMEngine(F, V, D, [32, 16], B, num_dense=Nd, lr=0.05, device="cuda", world=1, rank=0, seed=9,
                                 init_tables=(ref.table.clone(), ref.lin_w.clone()), alias_world1=alias_world1)
        assert sh.mb == 2                                   # two micro-batches per step
        assert sh.ex.local == alias_world1
        for a, b in zip(sh.Ws, ref.Ws):
            a.copy_(b)
        g = torch.Generator(device="cuda")
        g.manual_seed(2)
        batches = []
        for _ in range(3):
            keys = torch.randint(0, 10**14, (B, F), device="cuda", generator=g)
            keys[5, 2] = -1
            dense = torch.rand((B, Nd), device="cuda", generator=g)
            labels = (torch.rand(B, device="cuda", generator=g) < 0.25).float()
            batches.append((keys, dense, labels))
        for t, (keys, dense, labels) in enumerate(batches):     # prefetched routes from the second step on
            nk = batches[t + 1][0] if t + 1 < len(batches) else None
            l_ref = ref.train_step(keys, dense, labels).item()
            l_sh = sh.train_step(keys, dense, labels, next_keys=nk).item()
            assert abs(l_ref - l_sh) <= 2e-6 * abs(l_ref)
        np.testing.assert_allclose(sh.table.cpu().numpy(), ref.table.cpu().numpy(), rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(sh.lin_w.cpu().numpy(), ref.lin_w.cpu().numpy(), rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(sh.lin_bias.cpu().numpy(), ref.lin_bias.cpu().numpy(), rtol=2e-5, atol=2e-7)
        for a, b in zip(sh.Ws, ref.Ws):
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-5, atol=2e-6)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("split", ["f16x2", "bf16x3"])
def test_sharded_dcn_world1_matches_unsharded_dcn(split):
    """ShardedDCNEngine runs the SAME dense half as DCNEngine (dcn_engine.DCNDense: pre-split weights, the library's operand split,
    amax records chained through the epilogues), with the gradients going through the all-reduce bucket instead of being applied in
    place: at world 1, from the same tables and weights, three steps must give the same losses and parameters -- at a shape where
    every wide GEMM takes the register-split path (B = 4096, input width 6 * 64 + 5 = 389, MLP [256, 128])."""
    import torch.distributed as dist
    from deep_recommenders_amd import ops
    from deep_recommenders_amd.dcn_engine import DCNEngine
    from deep_recommenders_amd.sharded import ShardedDCNEngine
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    prev = ops.set_gemm_split(split)
    try:
        F, V, D, B, Nd, lr = 6, 3000, 64, 4096, 5, 0.05
        ref = DCNEngine(F, V, D, 3, [256, 128], B, num_dense=Nd, lr=lr, seed=4)
        sh = ShardedDCNEngine(F, V, D, 3, [256, 128], B, num_dense=Nd, lr=lr, device="cuda", world=1, rank=0, seed=4,
                              init_tables=ref.table.clone())
        assert sh.core is not None and sh.h2 == (split == "f16x2") and ref.h2 == sh.h2
        assert all(wp is not None for wp in sh.core.cross_planes) and sh.core.wplanes[0] is not None and sh.core.wplanes[1] is not None
        with torch.no_grad():
            for a, b in zip(sh.cross_W + sh.cross_b + sh.Ws + sh.bs, ref.cross_W + ref.cross_b + ref.Ws + ref.bs):
                a.copy_(b)
        g = torch.Generator(device="cuda")
        g.manual_seed(7)
        for _ in range(3):
            keys = torch.randint(0, 10**14, (B, F), device="cuda", generator=g)
            dense = torch.rand((B, Nd), device="cuda", generator=g)
            labels = (torch.rand(B, device="cuda", generator=g) < 0.25).float()
            l_ref = ref.train_step(keys, dense, labels).item()
            l_sh = sh.train_step(keys, dense, labels).item()
            assert abs(l_ref - l_sh) <= 2e-6 * abs(l_ref), (l_ref, l_sh)
        np.testing.assert_allclose(sh.table.cpu().numpy(), ref.table.cpu().numpy(), rtol=2e-5, atol=2e-6)
        for a, b in zip(sh.cross_W + sh.cross_b + sh.Ws + sh.bs, ref.cross_W + ref.cross_b + ref.Ws + ref.bs):
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-5, atol=2e-6)
    finally:
        ops.set_gemm_split(prev)
        dist.destroy_process_group()


def test_engine_adam_tf_equals_tf_nonlazy_sparse_adam():
    """optimizer="adam_tf": tf.train.AdamOptimizer on sparse gradients is NON-lazy -- m / v of every row decay on every step and the row
    keeps moving while m != 0 (examples/train_fm_on_movielens_estimator.py:51-52).  The engine evaluates those
    decay-only steps lazily (replayed when a row is next looked up, dr_adam_catchup_rows) -- after adam_flush() the tables must
    equal the oracle's DENSE Adam applied to the whole variable every step (oracle/torch_ref.py adam_dense_step on the dense
    gradient, zeros for untouched rows; pinned against torch.optim.Adam in tests/test_oracle_third_party_pins.py).
    Small vocabulary + few distinct keys per step, so rows are touched, left alone for several steps, and touched again."""
    from deep_recommenders_amd.engine import DeepFMEngine
    F, V, D, B, Nd, lr = 3, 40, 16, 64, 2, 0.01
    eng = DeepFMEngine(F, V, D, [32, 16], B, num_dense=Nd, lr=lr, seed=7, lin_init_std=0.1, optimizer="adam_tf")
    lazy = DeepFMEngine(F, V, D, [32, 16], B, num_dense=Nd, lr=lr, seed=7, lin_init_std=0.1, optimizer="adam")
    tt, tl, tb = eng.table.cpu().double(), eng.lin_w.cpu().double(), torch.zeros(1, dtype=torch.float64)
    Ws = [w.cpu().double().contiguous() for w in eng.Ws]
    bs = [b.cpu().double() for b in eng.bs]
    z = torch.zeros_like
    mt, vt, ml, vl, mb, vb = z(tt), z(tt), z(tl), z(tl), z(tb), z(tb)
    mW, vW, mB, vB = [z(w) for w in Ws], [z(w) for w in Ws], [z(b) for b in bs], [z(b) for b in bs]
    rb = [f * V for f in range(F)]
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    nsteps = 9
    for step in range(1, nsteps + 1):
        # a narrow, shifting window of raw keys: most rows rest for a few steps between touches, some are never touched
        keys = torch.randint(1000 * (step % 3), 1000 * (step % 3) + 6, (B, F), device="cuda", generator=g)
        dense = torch.rand((B, Nd), device="cuda", generator=g)
        labels = (torch.rand(B, device="cuda", generator=g) < 0.3).float()
        loss = eng.train_step(keys, dense, labels).item()
        lazy.train_step(keys, dense, labels)
        ids = np.stack([O.hash_bucket_i64(keys[:, f].cpu().numpy(), V) for f in range(F)], axis=1)
        a = [t.clone().requires_grad_(True) for t in (tt, tl, tb)]
        aW = [w.clone().requires_grad_(True) for w in Ws]
        aB = [b.clone().requires_grad_(True) for b in bs]
        logit = T.deepfm_logit(a[0], a[1], a[2], torch.tensor(ids), list(range(F + 1)), rb, aW, aB, dense.cpu().double())
        lo = T.sigmoid_cross_entropy(labels.cpu().double(), logit)
        lo.backward()
        assert abs(loss - lo.item()) <= 2e-5 * abs(lo.item()), (step, loss, lo.item())
        # TF: the dense Adam rule on the WHOLE variable (autograd's dense gradient is zero on untouched rows)
        T.adam_dense_step(tt, a[0].grad, mt, vt, lr, step)
        T.adam_dense_step(tl, a[1].grad, ml, vl, lr, step)
        T.adam_dense_step(tb, a[2].grad, mb, vb, lr, step)
        for i in range(len(Ws)):
            T.adam_dense_step(Ws[i], aW[i].grad, mW[i], vW[i], lr, step)
            T.adam_dense_step(bs[i], aB[i].grad, mB[i], vB[i], lr, step)
    eng.adam_flush()
    torch.cuda.synchronize()
    tol = 2e-2 * lr
    np.testing.assert_allclose(eng.table.cpu().numpy(), tt.numpy(), rtol=0, atol=tol)
    np.testing.assert_allclose(eng.lin_w.cpu().numpy(), tl.numpy(), rtol=0, atol=tol)
    np.testing.assert_allclose(eng.m_table.cpu().numpy(), mt.numpy(), rtol=1e-3, atol=1e-7)
    np.testing.assert_allclose(eng.v_table.cpu().numpy(), vt.numpy(), rtol=1e-3, atol=1e-10)
    # and the distinction is real: the row-wise lazy engine leaves resting rows where they were
    assert float((lazy.table.cpu().double() - tt).abs().max()) > 20 * tol
    never = torch.tensor(sorted(set(range(F * V)) - set(int(r) for r in torch.nonzero(eng.row_step.cpu() > 0).reshape(-1).tolist())))
    if never.numel():
        pass    # (rows never looked up keep stamp 0 only until adam_flush stamps them: nothing to assert on their values beyond the table check)


def test_sharded_engine_through_the_cabi_exchange_library():
    """The sharded DeepFM step with every exchange issued through include/dr_collectives.h (sharded.CApiTransport: dr_coll_alltoallv
    / dr_coll_allreduce_f32 on the engine's streams, a communicator of one rank) equals the unsharded engine -- the C-ABI exchange
    library drives a real training step, not only a round trip."""
    from deep_recommenders_amd.engine import DeepFMEngine
    from deep_recommenders_amd.sharded import CApiTransport, ShardedDeepFMEngine
    F, V, D, B, Nd = 6, 3000, 16, 768, 3
    tr = CApiTransport(1, 0)
    try:
        ref = DeepFMEngine(F, V, D, [32, 16], B, num_dense=Nd, lr=0.05, seed=9, lin_init_std=0.1)
        sh = ShardedDeepFMEngine(F, V, D, [32, 16], B, num_dense=Nd, lr=0.05, device="cuda", world=1, rank=0, seed=9,
                                 init_tables=(ref.table.clone(), ref.lin_w.clone()), alias_world1=False, transport=tr)
        assert not sh.ex.local and sh.tr is tr
        for a, b in zip(sh.Ws, ref.Ws):
            a.copy_(b)
        g = torch.Generator(device="cuda")
        g.manual_seed(2)
        batches = []
        for _ in range(3):
            keys = torch.randint(0, 10**14, (B, F), device="cuda", generator=g)
            dense = torch.rand((B, Nd), device="cuda", generator=g)
            labels = (torch.rand(B, device="cuda", generator=g) < 0.25).float()
            batches.append((keys, dense, labels))
        for t, (keys, dense, labels) in enumerate(batches):
            nk = batches[t + 1][0] if t + 1 < len(batches) else None
            l_ref = ref.train_step(keys, dense, labels).item()
            l_sh = sh.train_step(keys, dense, labels, next_keys=nk).item()
            assert abs(l_ref - l_sh) <= 2e-6 * abs(l_ref)
        torch.cuda.synchronize()
        np.testing.assert_allclose(sh.table.cpu().numpy(), ref.table.cpu().numpy(), rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(sh.lin_w.cpu().numpy(), ref.lin_w.cpu().numpy(), rtol=2e-5, atol=2e-6)
        for a, b in zip(sh.Ws, ref.Ws):
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-5, atol=2e-6)
    finally:
        torch.cuda.synchronize()
        tr.close()


@pytest.mark.parametrize("split", ["f16x2", "bf16x3"])
def test_sharded_engine_fused_pack_tracks_the_two_launch_pack(split, monkeypatch):
    """DR_FUSE_PACK (default on since round 5: the first-layer dgrad writes every slot's gradient row straight into the send layout,
    dr_h2_linear_nt_pack / dr_bf3_linear_nt_pack) against the dgrad + dr_emb_pack_grads pair, at a shape that takes the fused path
    (D = 64, micro-batches of 2048 rows, DNN [256, 32]): same losses, same tables / weights after three steps up to the one fused
    multiply-add of the FM term -- and both equal the unsharded engine."""
    from deep_recommenders_amd import ops
    from deep_recommenders_amd.engine import DeepFMEngine
    from deep_recommenders_amd.sharded import ShardedDeepFMEngine
    F, V, D, B, Nd = 5, 3000, 64, 4096, 3                     # (the wide-GEMM planes want micro-batches of >= 2048 rows)
    prev = ops.set_gemm_split(split)
    try:
        ref = DeepFMEngine(F, V, D, [256, 32], B, num_dense=Nd, lr=0.05, seed=9, lin_init_std=0.1)
        engs = {}
        for fp in ("1", "0"):
            monkeypatch.setenv("DR_FUSE_PACK", fp)
            e = ShardedDeepFMEngine(F, V, D, [256, 32], B, num_dense=Nd, lr=0.05, device="cuda", world=1, rank=0, seed=9,
                                    init_tables=(ref.table.clone(), ref.lin_w.clone()))
            assert e.mb == 2 and e.fuse_pack == (fp == "1") and e.h2 == (split == "f16x2")
            for a, b in zip(e.Ws, ref.Ws):
                a.copy_(b)
            engs[fp] = e
        g = torch.Generator(device="cuda")
        g.manual_seed(2)
        batches = [(torch.randint(0, 10**14, (B, F), device="cuda", generator=g), torch.rand((B, Nd), device="cuda", generator=g),
                    (torch.rand(B, device="cuda", generator=g) < 0.25).float()) for _ in range(3)]
        for t, (keys, dense, labels) in enumerate(batches):
            nk = batches[t + 1][0] if t + 1 < len(batches) else None
            l_ref = ref.train_step(keys, dense, labels).item()
            l1 = engs["1"].train_step(keys, dense, labels, next_keys=nk).item()
            l0 = engs["0"].train_step(keys, dense, labels, next_keys=nk).item()
            assert abs(l1 - l0) <= 1e-6 * abs(l0) and abs(l1 - l_ref) <= 2e-6 * abs(l_ref), (l1, l0, l_ref)
        torch.cuda.synchronize()
        for name in ("table", "lin_w"):
            a, b, r = (getattr(e, name).cpu().numpy() for e in (engs["1"], engs["0"], ref))
            np.testing.assert_allclose(a, b, rtol=2e-6, atol=2e-7, err_msg=name)
            np.testing.assert_allclose(a, r, rtol=2e-5, atol=2e-6, err_msg=name)
        for a, b, r in zip(engs["1"].Ws, engs["0"].Ws, ref.Ws):
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-6, atol=2e-7)
            np.testing.assert_allclose(a.cpu().numpy(), r.cpu().numpy(), rtol=2e-5, atol=2e-6)
    finally:
        ops.set_gemm_split(prev)


def test_engine_picks_up_weights_written_from_outside():
    """The wide layers' forward / dgrad read bf16 PLANES of the weights, refreshed by the engine after its own updates.  A write to
    eng.Ws from outside (checkpoint restore, copy_) must reach them too: two engines, one constructed with the
    weights and one that receives them by copy_ after construction, must train identically."""
    from deep_recommenders_amd.engine import DeepFMEngine
    F, V, D, B, Nd = 4, 2000, 64, 2304, 3
    a = DeepFMEngine(F, V, D, [256, 32], B, num_dense=Nd, lr=0.05, seed=4, lin_init_std=0.1)
    b = DeepFMEngine(F, V, D, [256, 32], B, num_dense=Nd, lr=0.05, seed=5, lin_init_std=0.1)
    assert a.wplanes[0] is not None
    b.table.copy_(a.table)
    b.lin_w.copy_(a.lin_w)
    b.flat_params.copy_(a.flat_params)                   # weights, biases: the planes of b are now stale
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    for _ in range(2):
        keys = torch.randint(0, 10**14, (B, F), device="cuda", generator=g)
        dense = torch.rand((B, Nd), device="cuda", generator=g)
        labels = (torch.rand(B, device="cuda", generator=g) < 0.25).float()
        la, lb = a.train_step(keys, dense, labels).item(), b.train_step(keys, dense, labels).item()
        assert abs(la - lb) <= 1e-6 * abs(la)            # (stale planes = another seed's weights: the losses would differ in the first digit)
    torch.cuda.synchronize()
    # not bit-equal by construction: the first layer's bias gradient is combined with fp32 atomics
    np.testing.assert_allclose(a.flat_params.cpu().numpy(), b.flat_params.cpu().numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(a.table.cpu().numpy(), b.table.cpu().numpy(), rtol=1e-5, atol=1e-7)


def test_c1_example_script_trains_fm_on_movielens_shape():
    """BASELINE config 1 plumbing: the runnable equivalent of the reference's examples/train_fm_on_movielens_estimator.py
    (model_fn -> FM -> sigmoid-CE -> Adam(0.01), AUC) on MovieLens-shaped synthetic batches of 256: the loss falls and the AUC
    rises above chance."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "train_fm_on_movielens_estimator.py")
    spec = importlib.util.spec_from_file_location("c1_example", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    before, after = mod.main(["--steps", "150", "--batch", "256", "--eval-steps", "8"])
    assert after["examples"] == 8 * 256 and after["global_step"] == 150
    assert after["loss"] < before["loss"] - 0.01 and after["auc"] > 0.6 > before["auc"] - 0.15
    # the AUC helper against a hand-checked case (ties get half credit)
    assert mod.auc([0, 0, 1, 1], [0.1, 0.4, 0.35, 0.8]) == pytest.approx(0.75)
    assert mod.auc([0, 1], [0.5, 0.5]) == pytest.approx(0.5)


def test_surface_gaps_activations_dropout_regularizers_low_rank_cross():
    """what the reference's layer surface accepts and round 1 raised on."""
    from deep_recommenders_amd import layers as L, ops
    import importlib
    dnn_mod = importlib.import_module("deep_recommenders_amd.estimator.models.feature_interaction.dnn")   # (the package exports the function)
    from deep_recommenders_amd.keras.models.ranking.dcn import Cross
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn((300, 20), device="cuda", generator=g)
    # --- sigmoid / tanh Dense activations: forward and backward against float64 autograd
    for name, fn in (("sigmoid", torch.sigmoid), ("tanh", torch.tanh)):
        store = dnn_mod.VariableStore()
        xi = x.clone().requires_grad_(True)
        out = dnn_mod.dnn(xi, [16, 8, 3], activation=getattr(dnn_mod, name), store=store, scope=name)
        out.square().sum().backward()
        ps = {k: v for k, v in store.vars.items()}
        xd = x.double().cpu().requires_grad_(True)
        Ws = [ps["%s__dense%s__kernel" % (name, s)].detach().double().cpu().requires_grad_(True) for s in ("", "_1", "_2")]
        bs = [ps["%s__dense%s__bias" % (name, s)].detach().double().cpu().requires_grad_(True) for s in ("", "_1", "_2")]
        h = fn(fn(xd @ Ws[0] + bs[0]) @ Ws[1] + bs[1]) @ Ws[2] + bs[2]
        h.square().sum().backward()
        np.testing.assert_allclose(out.detach().cpu().numpy(), h.detach().numpy(), rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(xi.grad.cpu().numpy(), xd.grad.numpy(), rtol=2e-4, atol=2e-6)
        np.testing.assert_allclose(ps["%s__dense__kernel" % name].grad.cpu().numpy(), Ws[0].grad.numpy(), rtol=2e-4, atol=2e-6)
    # --- dropout: always on (no train / eval switch), kept elements scaled by 1 / (1 - rate), the backward uses the same mask
    xi = torch.ones((2000, 64), device="cuda", requires_grad=True)
    y = L.dropout(xi, 0.25, seed=7)
    keep = (y != 0)
    assert abs(float(keep.float().mean()) - 0.75) < 0.01
    assert torch.allclose(y[keep], torch.full_like(y[keep], 1.0 / 0.75))
    y.sum().backward()
    assert torch.equal(xi.grad != 0, keep) and torch.allclose(xi.grad[keep], torch.full_like(y[keep], 1.0 / 0.75))
    assert not torch.equal(L.dropout(xi, 0.25, seed=8) != 0, keep)                      # another seed, another mask
    store = dnn_mod.VariableStore()
    a = dnn_mod.dnn(x, [16, 4], dropout=0.5, store=store, scope="drop")
    b = dnn_mod.dnn(x, [16, 4], dropout=0.5, store=store, scope="drop")
    assert not torch.equal(a, b)                                                         # dnn.py:26-27: a fresh mask on every call
    # --- L2 regularizers on Cross (dcn.py:27-30,39-45): layer.losses, with gradients
    layer = Cross(kernel_regu=0.01, bias_regu={"l2": 0.1}, bias_init="ones")
    out = layer(x)
    terms = layer.losses
    assert len(terms) == 2
    want = 0.01 * float(layer.kernel.detach().double().square().sum()) + 0.1 * float(layer.bias.detach().double().square().sum())
    assert float(sum(terms)) == pytest.approx(want, rel=1e-5)
    (out.sum() * 0 + sum(terms)).backward()
    np.testing.assert_allclose(layer.kernel.grad.cpu().numpy(), 0.02 * layer.kernel.detach().cpu().numpy(), rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(layer.bias.grad.cpu().numpy(), 0.2 * layer.bias.detach().cpu().numpy(), rtol=1e-5)
    with pytest.raises(ValueError):
        Cross(kernel_regu="l1")
    # --- low-rank Cross (projection_dim) through dr_cross_fwd(W = NULL) / dr_cross_combine_bwd, against float64 autograd
    lr = Cross(projection_dim=5, diag_scale=0.2, bias_init="ones")
    x0 = torch.randn((300, 20), device="cuda", generator=g).requires_grad_(True)
    xi = x.clone().requires_grad_(True)
    out = lr(x0, xi)
    out.square().sum().backward()
    U, V, bb = (t.detach().double().cpu().requires_grad_(True) for t in (lr.kernel_u, lr.kernel, lr.bias))
    x0d, xd = x0.detach().double().cpu().requires_grad_(True), x.double().cpu().requires_grad_(True)
    ref = x0d * ((xd @ U) @ V + bb + 0.2 * xd) + xd                                       # dcn.py:83-88
    ref.square().sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-5, atol=2e-6)
    for got, want in ((x0.grad, x0d.grad), (xi.grad, xd.grad), (lr.kernel_u.grad, U.grad), (lr.kernel.grad, V.grad), (lr.bias.grad, bb.grad)):
        np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=3e-4, atol=3e-5)
    # --- the plain sum used for the first-order bias gradient
    v = torch.randn(100000, device="cuda", generator=g)
    assert float(ops.reduce_sum(v)) == pytest.approx(float(v.double().sum()), abs=1e-2)


def test_cabi_collectives_single_rank_roundtrip():
    """include/dr_collectives.h on one GPU: a communicator of one rank (what one process of N would create), the fixed and the
    variable all-to-all (to itself), all-reduce and all-gather deliver the right bytes on the caller's stream."""
    import ctypes
    from deep_recommenders_amd import _coll_lib as C
    L = C.lib()
    idb = ctypes.create_string_buffer(C.ID_BYTES)
    C.check(L.dr_coll_unique_id(idb), "dr_coll_unique_id")
    comm = ctypes.c_void_p()
    C.check(L.dr_coll_init(ctypes.byref(comm), 1, 0, idb), "dr_coll_init")
    try:
        assert L.dr_coll_world(comm) == 1 and L.dr_coll_rank(comm) == 0
        st = torch.cuda.current_stream().cuda_stream
        a = torch.arange(7, dtype=torch.int64, device="cuda")
        b = torch.zeros_like(a)
        C.check(L.dr_coll_alltoall_i64(comm, a.data_ptr(), b.data_ptr(), 7, st), "dr_coll_alltoall_i64")
        rows = torch.randn((5, 16), device="cuda")
        got = torch.zeros_like(rows)
        cnt = (ctypes.c_int64 * 1)(5)
        C.check(L.dr_coll_alltoallv(comm, rows.data_ptr(), cnt, got.data_ptr(), cnt, 64, st), "dr_coll_alltoallv")
        g = torch.randn(1000, device="cuda")
        g0 = g.clone()
        C.check(L.dr_coll_allreduce_f32(comm, g.data_ptr(), g.numel(), st), "dr_coll_allreduce_f32")
        ag = torch.zeros_like(rows)
        C.check(L.dr_coll_allgather(comm, rows.data_ptr(), ag.data_ptr(), rows.numel() * 4, st), "dr_coll_allgather")
        torch.cuda.synchronize()
        assert torch.equal(b, a) and torch.equal(got, rows) and torch.equal(g, g0) and torch.equal(ag, rows)
        assert L.dr_coll_alltoallv(comm, rows.data_ptr(), cnt, got.data_ptr(), cnt, 0, st) == -1      # DRC_EINVAL
    finally:
        C.check(L.dr_coll_destroy(comm), "dr_coll_destroy")
