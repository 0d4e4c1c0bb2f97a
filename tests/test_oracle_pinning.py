"""Pins the oracle beyond the reference tests' literals: CPU only.

1. Fingerprint64, <= 32-byte branches (every int64 key's decimal text): against third-party answers -- abseil's CityHash64
   as compiled into the pyarrow wheel (tests/golden/make_cityhash_vectors.py) -- for the C oracle AND the Python restatement.
2. Fingerprint64, all branches (33-64, > 64 bytes): the C oracle against the independent Python restatement, fuzzed over
   lengths 0..300.  Two restatements by one author agreeing is a transcription check, not an external pin (oracle/README.md).
3. The [TF] rules with no reference-held expectation (mean-combiner / -1 / empty bag, losses at edge values, CCE SUM,
   sparse Adam, top_k / in_top_k ties): oracle/tf_semantics.py and oracle/torch_ref.py against the fixtures of a second,
   scalar-loop restatement (tests/golden/make_semantics_fixtures.py).
"""
import json
import os
import random

import numpy as np
import torch

from oracle import tf_semantics as O
from oracle import torch_ref as T

HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    with open(os.path.join(HERE, "golden", name)) as f:
        return json.load(f)


def test_fingerprint64_le32_bytes_against_third_party_cityhash():
    v = _load("cityhash64_le32_vectors.json")
    assert len(v["bytes"]) >= 33 * 6
    seen = set()
    for e in v["bytes"]:
        s = bytes.fromhex(e["hex"])
        assert len(s) <= 32
        seen.add(len(s))
        assert O.fingerprint64(s) == e["fp"], ("C oracle", len(s), e["hex"])
        assert O.fingerprint64_py(s) == e["fp"], ("python restatement", len(s), e["hex"])
    assert seen == set(range(33))                         # every length of the 0 / 1-3 / 4-7 / 8-16 / 17-32 branches


def test_hash_bucket_i64_decimal_text_against_third_party_cityhash():
    """the integer id path end to end: as_string (sign, no padding) -> Fingerprint64 -> mod, for 1..20 character keys"""
    v = _load("cityhash64_le32_vectors.json")["int64_as_decimal"]
    keys = np.array([e["key"] for e in v], dtype=np.int64)
    lens = {len(str(int(k))) for k in keys}
    assert {1, 16, 17, 18, 19, 20} <= lens                # the 17-20 character keys were "no vectors" in SURVEY 8c
    for nb in (10_000_000, 6040, 2**40 + 7, 1):
        got = O.hash_bucket_i64(keys, nb)
        for e, g in zip(v, got):
            want = -1 if e["key"] == -1 else e["fp"] % nb          # [TF] B1: -1 entries are dropped
            assert int(g) == want, (e["key"], nb)


def test_fingerprint64_c_equals_python_restatement_all_branches():
    rnd = random.Random(99)
    for n in range(0, 301):
        for _ in range(12 if n > 32 else 4):
            s = bytes(rnd.getrandbits(8) for _ in range(n))
            assert O.fingerprint64(s) == O.fingerprint64_py(s), n
    for n in (63, 64, 65, 127, 128, 129, 191, 192, 193, 1000, 4096):       # block boundaries of the long loop
        for fill in (b"\x00", b"\xff", b"a"):
            s = fill * n
            assert O.fingerprint64(s) == O.fingerprint64_py(s), (n, fill)


# ---- second-restatement fixtures ------------------------------------------------------------------------------------
FX = None


def fx(name):
    global FX
    if FX is None:
        FX = _load("tf_semantics_fixtures.json")
    return FX[name]


def test_mean_pool_rules():
    c = fx("mean_pool")
    table = np.array(c["table"], dtype=np.float32)
    ids = np.array(c["ids"], dtype=np.int64)
    want = np.array(c["expected"])
    np.testing.assert_allclose(O.embedding_mean_pool(table, ids), want, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(O.embedding_mean_pool_fast(table, ids), want, rtol=1e-6, atol=1e-7)
    assert np.all(O.embedding_mean_pool(table, ids)[1] == 0)                  # empty bag -> exact zeros
    # torch restatement (the gradient oracle's forward): same rule
    tp = T.pool_fields(torch.tensor(table), torch.tensor(ids), [0, ids.shape[1]], [0])[0].numpy()
    np.testing.assert_allclose(tp, want, rtol=1e-6, atol=1e-7)


def test_linear_model_rules():
    c = fx("linear_model")
    ids = [np.array(col, dtype=np.int64) for col in c["ids_per_col"]]
    ws = [np.array(w, dtype=np.float32) for w in c["w_per_col"]]
    np.testing.assert_array_equal(O.indicator_multi_hot(ids[0], 3), np.array(c["multi_hot_col0"], dtype=np.float32))
    got = O.first_order_gather(ids, ws, c["bias"]).reshape(-1)
    np.testing.assert_allclose(got, np.array(c["expected"]), rtol=1e-6, atol=1e-7)
    dense = sum(O.indicator_multi_hot(i, len(w)) @ w for i, w in zip(ids, ws)) + np.float32(c["bias"])
    np.testing.assert_allclose(dense, np.array(c["expected"]), rtol=1e-6, atol=1e-7)      # multi-hot matmul form == gather-sum


def test_losses_at_edge_values():
    c = fx("sigmoid_ce")
    z, x = np.array(c["labels"], dtype=np.float32), np.array(c["logits"], dtype=np.float32)
    assert abs(float(O.sigmoid_cross_entropy(z, x)) - c["expected"]) <= 1e-6 * c["expected"]
    assert abs(float(T.sigmoid_cross_entropy(torch.tensor(z, dtype=torch.float64), torch.tensor(x, dtype=torch.float64)))
               - c["expected"]) <= 1e-12 * c["expected"]
    for zi, xi, want in zip(z, x, c["per_example"]):       # |x| = 100: no overflow, loss == |x| or ~0
        got = float(O.sigmoid_cross_entropy(np.array([zi]), np.array([xi])))
        assert abs(got - want) <= 1e-6 * max(want, 1e-30) + 1e-38
    for name, fo, ft in (("log_loss", O.log_loss, T.log_loss), ("keras_bce", O.keras_binary_crossentropy, T.keras_bce)):
        c = fx(name)
        z, p = np.array(c["labels"], dtype=np.float32), np.array(c["predictions"], dtype=np.float32)
        assert abs(float(fo(z, p)) - c["expected"]) <= 1e-6 * c["expected"], name
        got = float(ft(torch.tensor(z, dtype=torch.float64), torch.tensor(p, dtype=torch.float64)))
        assert abs(got - c["expected"]) <= 1e-12 * c["expected"], name


def test_cce_from_logits_sum():
    c = fx("cce_sum")
    Y, S = np.array(c["labels"], dtype=np.float32), np.array(c["scores"], dtype=np.float32)
    assert abs(float(O.categorical_crossentropy_from_logits_sum(Y, S)) - c["expected"]) <= 1e-6 * c["expected"]
    w = np.array(c["sample_weight"], dtype=np.float32)
    assert abs(float(O.categorical_crossentropy_from_logits_sum(Y, S, w)) - c["expected_weighted"]) <= 1e-6 * c["expected_weighted"]


def _adam_inputs(c):
    var = torch.tensor(c["var"], dtype=torch.float64)
    steps = [(torch.tensor(s["ids"]), torch.tensor(s["grads"], dtype=torch.float64)) for s in c["steps"]]
    return var, steps


def test_adam_tf_nonlazy_on_sparse_gradients():
    """[TF] B15: m / v of the WHOLE variable decay each step (rows without a gradient still move while m != 0)."""
    c = fx("adam_tf")
    var, steps = _adam_inputs(c)
    m, v = torch.zeros_like(var), torch.zeros_like(var)
    for t, (ids, g) in enumerate(steps, start=1):
        dense = torch.zeros_like(var).index_add_(0, ids, g)            # duplicates summed first
        T.adam_dense_step(var, dense, m, v, c["lr"], t)
    np.testing.assert_allclose(var.numpy(), np.array(c["expected_var"]), rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(m.numpy(), np.array(c["expected_m"]), rtol=1e-12, atol=1e-18)
    np.testing.assert_allclose(v.numpy(), np.array(c["expected_v"]), rtol=1e-12, atol=1e-24)


def test_adam_rowwise_lazy_form_and_where_it_departs_from_tf():
    c, ctf = fx("adam_lazy"), fx("adam_tf")
    var, steps = _adam_inputs(c)
    m, v = torch.zeros_like(var), torch.zeros_like(var)
    for t, (ids, g) in enumerate(steps, start=1):
        dense = torch.zeros_like(var).index_add_(0, ids, g)
        T.adam_rows_step(var, dense, ids, m, v, c["lr"], t)
    np.testing.assert_allclose(var.numpy(), np.array(c["expected_var"]), rtol=1e-12, atol=1e-15)
    lazy, tf_ = np.array(c["expected_var"]), np.array(ctf["expected_var"])
    # rows 2, 3 and 4 are touched and then left alone: TF keeps moving them (decaying m), the lazy form does not;
    # row 1 is first touched on the last step (moments zero before): identical
    assert not np.allclose(lazy[4], tf_[4], rtol=1e-9, atol=0) and not np.allclose(lazy[2], tf_[2], rtol=1e-9, atol=0)
    np.testing.assert_allclose(lazy[1], tf_[1], rtol=1e-12)


def test_top_k_and_in_top_k_tie_rules():
    c = fx("top_k")
    x = np.array(c["x"], dtype=np.float32)
    vals, idx = O.top_k(x, c["k"])
    for r, (wv, wi) in enumerate(c["expected"]):
        assert list(idx[r]) == wi and np.allclose(vals[r], wv)            # ties: lower index first
    c = fx("in_top_k")
    pred = np.array(c["predictions"], dtype=np.float32)
    for j, k in enumerate(c["ks"]):
        got = O.in_top_k(np.array(c["targets"]), pred, k)
        assert [bool(g) for g in got] == [row[j] for row in c["expected"]], k
