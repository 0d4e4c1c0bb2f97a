"""Pins oracle/ against the reference's own known-answer tests."""
import json
import os

import numpy as np
import pytest

from oracle import tf_semantics as O

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))


def test_farmhash_upstream_vectors():
    for s, want in G["farmhash"]["fingerprint64"].items():
        assert O.fingerprint64(s.encode()) == want
        assert O.fingerprint64_py(s.encode()) == want
    t = G["farmhash"]["to_hash_bucket_fast"]
    got = O.hash_bucket_strings(np.array(t["inputs"], dtype=object), t["num_buckets"])
    assert got.tolist() == t["expected"]
    for row in G["farmhash"]["reference_test_keys"]:
        assert O.hash_bucket_strings(np.array([row["key"]], dtype=object), row["num_buckets"])[0] == row["id"]
        assert O.hash_bucket_i64(np.array([int(row["key"])]), row["num_buckets"])[0] == row["id"]


def test_farmhash_c_vs_python_restatement_all_short_branches():
    rng = np.random.default_rng(0)
    for n in range(0, 33):
        for _ in range(20):
            b = bytes(rng.integers(0, 256, size=n, dtype=np.uint8))
            assert O.fingerprint64(b) == O.fingerprint64_py(b), n


def test_hash_bucket_drops_missing():
    assert O.hash_bucket_i64(np.array([-1, 7]), 10)[0] == -1
    assert O.hash_bucket_strings(np.array(["", "x"], dtype=object), 10)[0] == -1
    # negative keys other than -1 hash their decimal text including the '-' sign
    assert O.hash_bucket_i64(np.array([-12]), 1000)[0] == O.fingerprint64(b"-12") % 1000


def test_vocab_lookup_oov_is_minus_one():
    # examples/train_fm_on_movielens_estimator.py:22-23 — movie_genres uses gender_vocab -> all OOV
    assert O.vocab_lookup(["Action", "M", "F"], ["F", "M"]).tolist() == [-1, 1, 0]
    assert O.vocab_lookup([1, 18, 99], [1, 18, 25, 35, 45, 50, 56]).tolist() == [0, 1, -1]


def test_fm_layer_matches_reference_numpy_expectation():
    # tests/keras/test_fm.py:17-26 (linear kernel zero-init -> output == interaction)
    rng = np.random.RandomState(0)
    sparse = rng.randint(0, 2, size=(10, 10)).astype(np.float32)
    emb = rng.normal(size=(10, 5, 5)).astype(np.float32)
    x_sum = np.sum(emb, axis=1)
    x_square_sum = np.sum(np.power(emb, 2), axis=1)
    expected = 0.5 * np.sum(np.power(x_sum, 2) - x_square_sum, axis=1, keepdims=True)
    out = O.fm_layer(sparse, emb, np.zeros((10, 1), np.float32), 0.0)
    np.testing.assert_allclose(out, expected, rtol=1e-6, atol=1e-6)
    # embedding_inputs=None -> linear only (fm.py:25-26)
    np.testing.assert_array_equal(O.fm_layer(sparse, None, np.zeros((10, 1), np.float32), 0.0), np.zeros((10, 1)))


def test_estimator_fm_shape_and_rank_check():
    # tests/estimator/test_fm.py:18-26 ; estimator/.../fm.py:19-20
    assert O.fm_second_order(np.random.randn(10, 2, 3).astype(np.float32)).shape == (10, 1)
    with pytest.raises(ValueError):
        O.fm_second_order(np.zeros((10, 6), np.float32))


def test_cross_known_answer():
    g = G["cross_kat"]
    out = O.cross(np.array(g["x0"], np.float32), np.array(g["x"], np.float32),
                  np.ones((3, 3), np.float32), np.zeros(3, np.float32))
    np.testing.assert_allclose(out, np.array(g["expected"]), rtol=1e-6, atol=1e-6)
    with pytest.raises(ValueError):
        O.cross(np.zeros((1, 3), np.float32), np.zeros((1, 4), np.float32), np.ones((4, 4), np.float32))


def test_take_long_axis_and_exclude():
    g = G["take_long_axis"]
    np.testing.assert_allclose(O.take_long_axis(np.array(g["arr"]), np.array(g["indices"])), np.array(g["expected"]))
    g = G["exclude"]
    s, i = O.exclude(np.array(g["scores"], np.float32), np.array(g["identifiers"]), np.array(g["exclude"]), g["k"])
    np.testing.assert_allclose(s, np.array(g["expected_scores"]), rtol=1e-6)
    assert i.tolist() == g["expected_ids"]


@pytest.mark.parametrize("index", ["streaming", "brute_force"])
def test_factorized_topk_metric_procedure(index):
    # tests/keras/test_factorized_top_k.py:86-130
    rng = np.random.RandomState(42)
    nc, nq, d = 100, 10, 4
    candidates = rng.normal(size=(nc, d)).astype(np.float32)
    queries = rng.normal(size=(nq, d)).astype(np.float32)
    true_c = rng.normal(size=(nq, d)).astype(np.float32)
    all_scores = np.concatenate([(queries * true_c).sum(1, keepdims=True), queries @ candidates.T], axis=1)
    ks = [1, 5, 10, 50]
    pos = (queries * true_c).sum(axis=1, keepdims=True)
    if index == "streaming":
        batches = [candidates[i:i + 32] for i in range(0, nc, 32)]
        topk, ids = O.streaming_top_k(queries, batches, k=max(ks))
        bs, bi = O.brute_force_top_k(queries, candidates, k=max(ks))
        np.testing.assert_array_equal(ids, bi)
    else:
        topk, _ = O.brute_force_top_k(queries, candidates, k=max(ks))
    y_pred = np.concatenate([pos, topk], axis=1)
    for k in ks:
        got = O.in_top_k(np.zeros(nq, np.int64), y_pred, k).mean()
        want = O.in_top_k(np.zeros(nq, np.int64), all_scores, k).mean()
        assert got == want


@pytest.mark.parametrize("h", [3, 5, 10, 15])
def test_hard_negative_mining_procedure(h):
    # tests/keras/test_sbcnm.py:16-41
    rng = np.random.RandomState(42)
    logits = rng.uniform(size=(2, 20)).astype(np.float32)
    labels = rng.permutation(np.eye(2, 20).T).T.astype(np.float32)
    ol, olab = O.hard_negative_mining(logits, labels, h)
    assert ol.shape[-1] == h + 1
    np.testing.assert_allclose((ol * olab).sum(-1), (logits * labels).sum(-1), rtol=1e-6)
    logits2 = logits + labels * 1000.0
    ol2, _ = O.hard_negative_mining(logits2, labels, h)
    np.testing.assert_allclose(np.sort(logits2, axis=1)[:, -h - 1:], np.sort(ol2), rtol=1e-6)


def test_remove_accidental_negative_procedure():
    # tests/keras/test_sbcnm.py:43-55
    rng = np.random.RandomState(42)
    logits = rng.uniform(size=(2, 4)).astype(np.float32)
    labels = rng.permutation(np.eye(2, 4).T).T.astype(np.float32)
    identifiers = rng.randint(0, 3, size=4)
    out = O.remove_accidental_negative(logits, labels, identifiers)
    np.testing.assert_allclose((out * labels).sum(1), (logits * labels).sum(1), rtol=1e-6)


def test_embedding_mean_pool_rules():
    # [TF] B5: drop ids<0, mean in id order, empty bag -> zeros; single id -> exact copy
    rng = np.random.default_rng(1)
    table = rng.normal(size=(7, 4)).astype(np.float32)
    ids = np.array([[3, -1, -1], [-1, -1, -1], [1, 1, 5], [-1, 6, 2]])
    out = O.embedding_mean_pool(table, ids)
    np.testing.assert_array_equal(out[0], table[3])
    np.testing.assert_array_equal(out[1], np.zeros(4, np.float32))
    np.testing.assert_array_equal(out[2], ((table[1] + table[1]) + table[5]) / np.float32(3))
    np.testing.assert_array_equal(out[3], (table[6] + table[2]) / np.float32(2))
    np.testing.assert_array_equal(out, O.embedding_mean_pool_fast(table, ids))


def test_first_order_gather_equals_dense_multi_hot():
    # keras/.../fm.py:47,55 builds multi-hot [B, sum V] @ kernel; gather-sum must be identical
    rng = np.random.default_rng(2)
    Vs = [5, 3]
    ids = [np.array([[0, 0], [4, -1], [-1, -1]]), np.array([[2], [-1], [1]])]
    ws = [rng.normal(size=v).astype(np.float32) for v in Vs]
    mh = np.concatenate([O.indicator_multi_hot(i, v) for i, v in zip(ids, Vs)], axis=1)
    dense = mh @ np.concatenate(ws)[:, None] + np.float32(0.25)
    np.testing.assert_allclose(O.first_order_gather(ids, ws, 0.25), dense, rtol=1e-6, atol=1e-7)


def test_losses_restatements_agree_on_easy_points():
    z = np.array([1, 0, 1, 0], np.float32)
    x = np.array([2.0, -1.0, 0.5, 3.0], np.float32)
    p = O.sigmoid(x)
    # sigmoid-CE on logits == -z log p - (1-z) log(1-p) exactly (up to eps terms)
    ref = np.mean(-z * np.log(p) - (1 - z) * np.log(1 - p))
    assert abs(O.sigmoid_cross_entropy(z, x) - ref) < 1e-6
    assert abs(O.log_loss(z, p) - ref) < 1e-5
    assert abs(O.keras_binary_crossentropy(z, p) - ref) < 1e-5


def test_retrieval_loss_identity_labels():
    rng = np.random.default_rng(3)
    q = rng.normal(size=(6, 4)).astype(np.float32)
    c = rng.normal(size=(6, 4)).astype(np.float32)
    s = (q @ c.T).astype(np.float64)
    want = (np.log(np.exp(s).sum(1)) - np.diag(s)).sum()
    assert abs(O.retrieval_loss(q, c) - want) < 1e-4
    s2 = s / 0.5
    want2 = (np.log(np.exp(s2).sum(1)) - np.diag(s2)).sum()
    assert abs(O.retrieval_loss(q, c, temperature=0.5) - want2) < 1e-4


def test_cin_oracle_reproduces_reference_known_answers():
    """tests/keras/test_xdeepfm.py:30-60 (kernel ones, relu; bias ones): pins oracle.cin, the checker of the HIP CIN kernel."""
    import json, os
    G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))
    for kat, bias in (("cin_outputs", None), ("cin_bias", np.ones(2, np.float32))):
        g = G[kat]
        x0, x = np.asarray(g["x0"], np.float32), np.asarray(g["x"], np.float32)
        out = O.cin(x0, x, np.ones((1, 4, g["feature_map"]), np.float32), bias, "relu")
        np.testing.assert_allclose(out, np.asarray(g["expected"], np.float32), rtol=1e-6, atol=1e-6)


def test_activation_unit_oracle_reproduces_reference_test_procedure():
    """tests/keras/test_din.py:17-48: ones kernels => output == reduce_sum(relu(concat @ ones)) (Subtract interacter included)."""
    rng = np.random.default_rng(1)
    x, y = rng.normal(size=(3, 5)).astype(np.float32), rng.normal(size=(3, 5)).astype(np.float32)
    for inter, cols in ((None, [x, y]), ((lambda xy: xy[0] - xy[1]), [x, y, x - y])):
        h = np.concatenate(cols, axis=1)
        want = np.maximum(h @ np.ones((h.shape[1], 10), np.float32), 0).sum(axis=1, keepdims=True)
        got = O.activation_unit(x, y, np.ones((h.shape[1], 10), np.float32), np.zeros(10, np.float32), np.ones((10, 1), np.float32),
                                np.zeros(1, np.float32), inter)
        np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6)
