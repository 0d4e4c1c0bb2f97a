"""ORACLE — test infrastructure only (see oracle/README.md).

NumPy restatement of the reference's embedding-lookup + feature-interaction hot path
(SURVEY.md §8a rows a1..a12).  The arithmetic of the reference lives in TensorFlow, which
is third-party, un-vendored and absent from this container, so every "[TF]" rule below is
restated from SURVEY.md Appendix B and pinned by the reference's own known-answer tests
(tests/golden/*.json; see tests/test_oracle_golden.py).

Nothing in deep_recommenders_amd/ imports this module.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may use it, as the checker.

All citations are relative to /root/reference/.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
_LIB = None


def build_c_oracle(force: bool = False) -> str:
    """Compile oracle/farmhash_fp64.c with gcc into oracle/_build/liboracle_c.so."""
    os.makedirs(_BUILD, exist_ok=True)
    so = os.path.join(_BUILD, "liboracle_c.so")
    src = os.path.join(_HERE, "farmhash_fp64.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, src])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_BUILD, "liboracle_c.so")
        if not os.path.exists(so):
            so = build_c_oracle()
        L = ctypes.CDLL(so)
        L.oracle_fingerprint64.restype = ctypes.c_uint64
        L.oracle_fingerprint64.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
        L.oracle_hash_bucket_i64.restype = None
        L.oracle_hash_bucket_i64.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_void_p]
        L.oracle_hash_bucket_bytes.restype = None
        L.oracle_hash_bucket_bytes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                               ctypes.c_uint64, ctypes.c_void_p]
        _LIB = L
    return _LIB


# --------------------------------------------------------------------------------------
# a1  hash-bucket column  [TF] B1
#     call sites: examples/train_fm_on_movielens_estimator.py:12-13,20-21
# --------------------------------------------------------------------------------------
def fingerprint64(data: bytes) -> int:
    return int(_lib().oracle_fingerprint64(data, len(data)))


def fingerprint64_py(data: bytes) -> int:
    """Pure-Python second restatement of farmhashna::Hash64 (every length branch), written independently of the C one
    (Python big ints masked to 64 bits, the long loop as a state dict) and used to cross-check it: tests fuzz C against
    this over lengths 0..300.  Slow: small inputs only."""
    M = (1 << 64) - 1
    k0, k1, k2 = 0xC3A5C85C97CB3127, 0xB492B66FBE98F273, 0x9AE16A3B2F90404F
    f64 = lambda p: int.from_bytes(data[p:p + 8], "little")
    f32 = lambda p: int.from_bytes(data[p:p + 4], "little")
    rot = lambda v, s: v if s == 0 else ((v >> s) | (v << (64 - s))) & M
    smix = lambda v: v ^ (v >> 47)

    def hl16(u, v, mul):
        a = ((u ^ v) * mul) & M
        a ^= a >> 47
        b = ((v ^ a) * mul) & M
        b ^= b >> 47
        return (b * mul) & M

    n = len(data)
    if n == 0:
        return k2
    if n <= 3:
        y = (data[0] + (data[n >> 1] << 8)) & 0xFFFFFFFF
        z = (n + (data[n - 1] << 2)) & 0xFFFFFFFF
        return (smix(((y * k2) & M) ^ ((z * k0) & M)) * k2) & M
    if n <= 7:
        mul = (k2 + n * 2) & M
        return hl16((n + (f32(0) << 3)) & M, f32(n - 4), mul)
    if n <= 16:
        mul = (k2 + n * 2) & M
        a = (f64(0) + k2) & M
        b = f64(n - 8)
        c = (rot(b, 37) * mul + a) & M
        d = ((rot(a, 25) + b) * mul) & M
        return hl16(c, d, mul)
    if n <= 32:
        mul = (k2 + n * 2) & M
        a = (f64(0) * k1) & M
        b = f64(8)
        c = (f64(n - 8) * mul) & M
        d = (f64(n - 16) * k2) & M
        return hl16((rot((a + b) & M, 43) + rot(c, 30) + d) & M,
                    (a + rot((b + k2) & M, 18) + c) & M, mul)
    if n <= 64:
        mul = (k2 + n * 2) & M
        a = (f64(0) * k2) & M
        b = f64(8)
        c = (f64(n - 8) * mul) & M
        d = (f64(n - 16) * k2) & M
        y = (rot((a + b) & M, 43) + rot(c, 30) + d) & M
        z = hl16(y, (a + rot((b + k2) & M, 18) + c) & M, mul)
        e = (f64(16) * mul) & M
        f = f64(24)
        g = ((y + f64(n - 32)) * mul) & M
        h = ((z + f64(n - 24)) * mul) & M
        return hl16((rot((e + f) & M, 43) + rot(g, 30) + h) & M, (e + rot((f + a) & M, 18) + g) & M, mul)

    # > 64 bytes: 56 bytes of running state (x, y, z and the pairs v, w), 64-byte blocks, the last 64 bytes (overlapping)
    # mixed with a multiplier derived from z
    def weak(p, a, b):
        w_, x_, y_, z_ = f64(p), f64(p + 8), f64(p + 16), f64(p + 24)
        a = (a + w_) & M
        b = rot((b + a + z_) & M, 21)
        c = a
        a = (a + x_ + y_) & M
        b = (b + rot(a, 44)) & M
        return (a + z_) & M, (b + c) & M

    st = {"x": 81, "y": (81 * k1 + 113) & M}
    st["z"] = (smix((st["y"] * k2 + 113) & M) * k2) & M
    v, w = (0, 0), (0, 0)
    st["x"] = (st["x"] * k2 + f64(0)) & M
    end = ((n - 1) // 64) * 64
    last64 = end + ((n - 1) & 63) - 63

    def block(p, mul, tail):
        nonlocal v, w
        x, y, z = st["x"], st["y"], st["z"]
        x = (rot((x + y + v[0] + f64(p + 8)) & M, 37) * mul) & M
        y = (rot((y + v[1] + f64(p + 48)) & M, 42) * mul) & M
        if tail:
            x ^= (w[1] * 9) & M
            y = (y + v[0] * 9 + f64(p + 40)) & M
        else:
            x ^= w[1]
            y = (y + v[0] + f64(p + 40)) & M
        z = (rot((z + w[0]) & M, 33) * mul) & M
        v = weak(p, (v[1] * mul) & M, (x + w[0]) & M)
        w = weak(p + 32, (z + w[1]) & M, (y + f64(p + 16)) & M)
        st["x"], st["y"], st["z"] = z, y, x          # z and x trade places

    p = 0
    while True:
        block(p, k1, False)
        p += 64
        if p == end:
            break
    mul = (k1 + ((st["z"] & 0xFF) << 1)) & M
    w = ((w[0] + ((n - 1) & 63)) & M, w[1])
    v = ((v[0] + w[0]) & M, v[1])
    w = ((w[0] + v[0]) & M, w[1])
    block(last64, mul, True)
    x, y, z = st["x"], st["y"], st["z"]
    return hl16((hl16(v[0], w[0], mul) + (smix(y) * k0) + z) & M, (hl16(v[1], w[1], mul) + x) & M, mul)


def hash_bucket_i64(keys: np.ndarray, num_buckets: int) -> np.ndarray:
    """[TF] B1 on int64 input: as_string -> Fingerprint64 -> mod N; key == -1 is dropped (-> -1)."""
    k = np.ascontiguousarray(keys, dtype=np.int64)
    out = np.empty_like(k)
    _lib().oracle_hash_bucket_i64(k.ctypes.data, k.size, int(num_buckets), out.ctypes.data)
    return out


def hash_bucket_strings(values, num_buckets: int) -> np.ndarray:
    """[TF] B1 on string input: "" is dropped (-> -1)."""
    flat = [v if isinstance(v, bytes) else str(v).encode("utf-8") for v in np.asarray(values, dtype=object).ravel()]
    offs = np.zeros(len(flat) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(b) for b in flat])
    blob = np.frombuffer(b"".join(flat) + b"\0", dtype=np.uint8).copy()
    out = np.empty(len(flat), dtype=np.int64)
    _lib().oracle_hash_bucket_bytes(blob.ctypes.data, offs.ctypes.data, len(flat), int(num_buckets), out.ctypes.data)
    return out.reshape(np.asarray(values, dtype=object).shape)


# --------------------------------------------------------------------------------------
# a2  vocabulary-list column  [TF] B2
#     call sites: examples/train_fm_on_movielens_estimator.py:14-19,22-23
# --------------------------------------------------------------------------------------
def vocab_lookup(values, vocab) -> np.ndarray:
    """id = position in list; OOV -> -1 (default_value=-1, no OOV buckets); -1/"" dropped (-> -1)."""
    table = {v: i for i, v in enumerate(vocab)}
    arr = np.asarray(values, dtype=object)
    out = np.array([table.get(v, -1) for v in arr.ravel()], dtype=np.int64)
    return out.reshape(arr.shape)


# --------------------------------------------------------------------------------------
# a3  embedding lookup  [TF] B4/B5 safe_embedding_lookup_sparse(combiner='mean')
#     call sites: keras/models/ranking/fm.py:48-51,57-61 ; estimator/.../fm.py:48-52
# --------------------------------------------------------------------------------------
def embedding_mean_pool(table: np.ndarray, ids: np.ndarray) -> np.ndarray:
    """ids [B, L] int64 padded with -1 -> [B, D].  Drop ids<0; mean in id order; empty -> 0."""
    ids = np.asarray(ids, dtype=np.int64)
    if ids.ndim == 1:
        ids = ids[:, None]
    B, L = ids.shape
    D = table.shape[1]
    out = np.zeros((B, D), dtype=np.float32)
    for b in range(B):
        acc = np.zeros(D, dtype=np.float32)
        cnt = 0
        for l in range(L):
            i = ids[b, l]
            if i >= 0:
                acc = (acc + table[i]).astype(np.float32)
                cnt += 1
        if cnt:
            out[b] = acc / np.float32(cnt)
    return out


def embedding_mean_pool_fast(table: np.ndarray, ids: np.ndarray) -> np.ndarray:
    """Vectorised equivalent of embedding_mean_pool (same left-to-right fp32 sum order)."""
    ids = np.asarray(ids, dtype=np.int64)
    if ids.ndim == 1:
        ids = ids[:, None]
    B, L = ids.shape
    acc = np.zeros((B, table.shape[1]), dtype=np.float32)
    cnt = np.zeros(B, dtype=np.float32)
    for l in range(L):
        m = ids[:, l] >= 0
        acc[m] = acc[m] + table[ids[m, l]]
        cnt += m
    nz = cnt > 0
    acc[nz] = acc[nz] / cnt[nz, None]
    return acc


# --------------------------------------------------------------------------------------
# a5  first-order term  [TF] B3 (indicator multi-hot) + B7 (linear_model) / Dense(1, zeros)
#     keras/models/ranking/fm.py:16-20,26,37,47,55 ; estimator/.../fm.py:43-44
# --------------------------------------------------------------------------------------
def indicator_multi_hot(ids: np.ndarray, num_buckets: int) -> np.ndarray:
    """count vector [B, num_buckets] fp32; -1 contributes nothing; duplicates add."""
    ids = np.asarray(ids, dtype=np.int64)
    if ids.ndim == 1:
        ids = ids[:, None]
    out = np.zeros((ids.shape[0], num_buckets), dtype=np.float32)
    for b in range(ids.shape[0]):
        for i in ids[b]:
            if i >= 0:
                out[b, i] += 1.0
    return out


def first_order_gather(ids_per_col, w_per_col, bias: float) -> np.ndarray:
    """Gather-sum form of multi_hot @ w + b (mathematically identical; SURVEY §7 hard part 2)."""
    B = np.asarray(ids_per_col[0]).shape[0]
    out = np.zeros(B, dtype=np.float32)
    for ids, w in zip(ids_per_col, w_per_col):
        ids = np.asarray(ids, dtype=np.int64)
        if ids.ndim == 1:
            ids = ids[:, None]
        w = np.asarray(w, dtype=np.float32).reshape(-1)
        for l in range(ids.shape[1]):
            m = ids[:, l] >= 0
            out[m] = out[m] + w[ids[m, l]]
    return (out + np.float32(bias)).reshape(B, 1)


# --------------------------------------------------------------------------------------
# a6  FM second-order   keras/models/ranking/fm.py:28-35 ; estimator/.../fm.py:10-26
# --------------------------------------------------------------------------------------
def fm_second_order(x: np.ndarray) -> np.ndarray:
    if x.ndim != 3:
        raise ValueError("The rank of `x` should be 3. Got rank = {}.".format(x.ndim))  # estimator fm.py:19-20
    x = x.astype(np.float32)
    sum_square = np.square(np.sum(x, axis=1, dtype=np.float32))
    square_sum = np.sum(np.square(x), axis=1, dtype=np.float32)
    return (np.float32(0.5) * np.sum(sum_square - square_sum, axis=1, keepdims=True, dtype=np.float32)).astype(np.float32)


def fm_layer(sparse_inputs, embedding_inputs, linear_kernel, linear_bias) -> np.ndarray:
    """keras FM.call (fm.py:23-37): Dense(1)(sparse) [+ interaction]."""
    lin = sparse_inputs.astype(np.float32) @ linear_kernel.astype(np.float32) + np.float32(linear_bias)
    if embedding_inputs is None:
        return lin
    return lin + fm_second_order(embedding_inputs)


# --------------------------------------------------------------------------------------
# a7  MLP tower  keras/models/ranking/deepfm.py:30-34 ; estimator/.../dnn.py:9-31   [TF] B8
# --------------------------------------------------------------------------------------
def dense(x, kernel, bias=None, activation=None):
    y = x.astype(np.float32) @ kernel.astype(np.float32)
    if bias is not None:
        y = y + bias.astype(np.float32)
    if activation == "relu":
        y = np.maximum(y, np.float32(0))
    elif activation == "sigmoid":
        y = sigmoid(y)
    elif activation not in (None, "linear"):
        raise ValueError(activation)
    return y.astype(np.float32)


def dnn(x, kernels, biases, activation="relu"):
    """hidden layers with activation, last layer linear (dnn.py:17-29 ; deepfm.py:30-34)."""
    for W, b in zip(kernels[:-1], biases[:-1]):
        x = dense(x, W, b, activation)
    return dense(x, kernels[-1], biases[-1], None)


def sigmoid(x):
    x = x.astype(np.float32)
    return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(np.float32)


# --------------------------------------------------------------------------------------
# a8  DeepFM / FactorizationMachine forward   keras/.../deepfm.py:36-47 ; fm.py:54-64
# --------------------------------------------------------------------------------------
def deepfm_forward(ids_per_field, tables, lin_w, lin_b, dnn_kernels, dnn_biases,
                   dense_features=None, activation="relu", return_parts=False):
    """ids_per_field: list of F arrays [B, L_f] (post hash/vocab, -1 = missing).
    tables: list of F [V_f, D]; lin_w: list of F [V_f]; dnn_*: tower weights (None -> FM only).
    dense_features (extension): [B, Nd] appended to the DNN input only."""
    embs = [embedding_mean_pool_fast(t, i) for t, i in zip(tables, ids_per_field)]
    stack = np.stack(embs, axis=1)                      # deepfm.py:44
    concat = np.concatenate(embs, axis=1)               # deepfm.py:45
    logit = first_order_gather(ids_per_field, lin_w, lin_b) + fm_second_order(stack)
    if dnn_kernels is not None:
        x = concat if dense_features is None else np.concatenate([concat, dense_features.astype(np.float32)], axis=1)
        logit = logit + dnn(x, dnn_kernels, dnn_biases, activation)
    prob = sigmoid(logit)
    if return_parts:
        return prob, logit, stack
    return prob


def wdl_forward(ids_per_field, tables, lin_w, lin_b, dnn_kernels, dnn_biases, activation="relu"):
    """estimator/models/ranking/wide_and_deep.py:29-48 of the reference: sigmoid(linear_model(indicator columns) +
    dnn(concat(input_layer(embedding column) ...), units + [1]))."""
    embs = [embedding_mean_pool_fast(t, i) for t, i in zip(tables, ids_per_field)]
    wide = first_order_gather(ids_per_field, lin_w, lin_b)                    # :30-32
    deep = dnn(np.concatenate(embs, axis=1), dnn_kernels, dnn_biases, activation)   # :34-46
    return sigmoid(wide + deep)                                                # :48


def fnn_forward(ids_per_field, tables, lin_w, fm_bias, dnn_kernels, dnn_biases, activation="relu"):
    """estimator/models/ranking/fnn.py:50-90: dnn input = [FM bias (tiled) | per-field Dense(1, no bias)(multi-hot) |
    per-field embedding]; output sigmoid(dnn(...))."""
    B = ids_per_field[0].shape[0]
    weights = []
    for ids, w in zip(ids_per_field, lin_w):                                   # :52-63
        m = ids >= 0
        weights.append((np.where(m, np.asarray(w, np.float32)[np.where(m, ids, 0)], 0.0)).sum(axis=1, keepdims=True))
    concat_weights = np.concatenate(weights, axis=1).astype(np.float32)        # :64
    embs = [embedding_mean_pool_fast(t, i) for t, i in zip(tables, ids_per_field)]   # :66-77
    bias = np.tile(np.asarray(fm_bias, np.float32).reshape(1, -1), (B, 1))     # :80-81
    x = np.concatenate([bias, concat_weights, np.concatenate(embs, axis=1)], axis=1)   # :83
    return sigmoid(dnn(x, dnn_kernels, dnn_biases, activation))                # :85-90


# --------------------------------------------------------------------------------------
# a9  Cross layer   keras/models/ranking/dcn.py:70-88
# --------------------------------------------------------------------------------------
def cross(x0, x, kernel, bias=None, diag_scale=0.0, kernel_u=None):
    if x is None:
        x = x0                                           # dcn.py:72-73
    if x0.shape[-1] != x.shape[-1]:
        raise ValueError("`x0` and `x` dim mismatch. Got `x0` dim = {} and `x` dim = {}".format(
            x0.shape[-1], x.shape[-1]))                  # dcn.py:75-78
    x0 = x0.astype(np.float32)
    x = x.astype(np.float32)
    if kernel_u is None:
        prod = dense(x, kernel, bias)                    # dcn.py:81
    else:
        prod = dense(dense(x, kernel_u), kernel, bias)   # dcn.py:83
    if diag_scale:
        prod = prod + np.float32(diag_scale) * x         # dcn.py:85-86
    return (x0 * prod + x).astype(np.float32)            # dcn.py:88


# --------------------------------------------------------------------------------------
# a12 losses   [TF] B9 / B10 / B11
# --------------------------------------------------------------------------------------
def sigmoid_cross_entropy(labels, logits):
    """tf.losses.sigmoid_cross_entropy (examples/train_fm_on_movielens_estimator.py:46)."""
    x = logits.astype(np.float64).reshape(-1)
    z = labels.astype(np.float64).reshape(-1)
    per = np.maximum(x, 0) - x * z + np.log1p(np.exp(-np.abs(x)))
    return np.float32(per.mean())


def log_loss(labels, predictions, eps=1e-7):
    """tf.losses.log_loss (examples/train_deepfm_on_movielens_estimator.py:47)."""
    p = predictions.astype(np.float64).reshape(-1)
    z = labels.astype(np.float64).reshape(-1)
    per = -z * np.log(p + eps) - (1 - z) * np.log(1 - p + eps)
    return np.float32(per.mean())


def keras_binary_crossentropy(labels, predictions, eps=1e-7):
    """tf.keras.losses.binary_crossentropy on probabilities (examples/train_deepfm_on_movielens_keras.py:43)."""
    p = np.clip(predictions.astype(np.float64).reshape(-1), eps, 1 - eps)
    z = labels.astype(np.float64).reshape(-1)
    per = -(z * np.log(p + eps) + (1 - z) * np.log(1 - p + eps))
    return np.float32(per.mean())


# --------------------------------------------------------------------------------------
# a10 two-tower retrieval task   keras/models/retrieval/sbcnm.py
# --------------------------------------------------------------------------------------
MAX_FLOAT = np.finfo(np.float32).max / 100.0   # sbcnm.py:9
MIN_FLOAT = np.finfo(np.float32).min / 100.0   # sbcnm.py:10


def top_k(x, k):
    """[TF] B13: k largest per row, descending, ties -> lower index first."""
    x = np.asarray(x)
    idx = np.argsort(-x, axis=1, kind="stable")[:, :k]
    return np.take_along_axis(x, idx, axis=1), idx


def gather_elements_along_row(data, column_indices):
    assert data.shape[0] == column_indices.shape[0]      # sbcnm.py:18-19
    return np.take_along_axis(data, column_indices, axis=1)


def hard_negative_mining(logits, labels, num_hard_negatives):
    """sbcnm.py:33-49"""
    k = min(num_hard_negatives + 1, logits.shape[1])
    _, idx = top_k(logits + labels * np.float32(MAX_FLOAT), k)
    return gather_elements_along_row(logits, idx), gather_elements_along_row(labels, idx)


def remove_accidental_negative(logits, labels, identifiers):
    """sbcnm.py:52-75"""
    identifiers = np.asarray(identifiers).reshape(-1, 1)
    pos_idx = np.argmax(labels, axis=1)
    pos_ident = identifiers[pos_idx]
    dup = (pos_ident == identifiers.T).astype(labels.dtype)
    dup = dup - labels
    return logits + dup * np.float32(MIN_FLOAT)


def sampling_probability_correction(logits, candidate_sampling_probability):
    """sbcnm.py:78-86"""
    return logits - np.log(candidate_sampling_probability.astype(np.float32))


def categorical_crossentropy_from_logits_sum(labels, scores, sample_weight=None):
    """[TF] B12: CCE(from_logits=True, reduction=SUM) (sbcnm.py:100-102,151)."""
    s = scores.astype(np.float64)
    m = s.max(axis=1, keepdims=True)
    lse = m + np.log(np.exp(s - m).sum(axis=1, keepdims=True))
    per = -(labels.astype(np.float64) * (s - lse)).sum(axis=1)
    if sample_weight is not None:
        per = per * np.asarray(sample_weight, dtype=np.float64).reshape(-1)
    return np.float32(per.sum())


def retrieval_loss(q, c, sample_weight=None, candidate_sampling_probability=None, candidate_ids=None,
                   temperature=None, num_hard_negatives=None):
    """Retrieval.call (sbcnm.py:120-163) using the module-local helper layers (:33-86); the
    reference's own optional branches name a non-existent module."""
    scores = q.astype(np.float32) @ c.astype(np.float32).T                # :129
    labels = np.eye(scores.shape[0], scores.shape[1], dtype=np.float32)   # :134
    if candidate_sampling_probability is not None:
        scores = sampling_probability_correction(scores, candidate_sampling_probability)
    if candidate_ids is not None:
        scores = remove_accidental_negative(scores, labels, candidate_ids)
    if num_hard_negatives is not None:
        scores, labels = hard_negative_mining(scores, labels, num_hard_negatives)
    if temperature is not None:
        scores = scores / np.float32(temperature)                          # :148-149
    return categorical_crossentropy_from_logits_sum(labels, scores, sample_weight)


# --------------------------------------------------------------------------------------
# a11 top-K retrieval   keras/models/retrieval/factorized_top_k.py
# --------------------------------------------------------------------------------------
def take_long_axis(arr, indices):
    """factorized_top_k.py:26-41"""
    return np.take_along_axis(np.asarray(arr), np.asarray(indices), axis=1)


def exclude(scores, identifiers, exclude_ids, k):
    """factorized_top_k.py:44-67 (penalty 1e5 at :62)."""
    scores = np.asarray(scores, dtype=np.float32)
    identifiers = np.asarray(identifiers)
    isin = (identifiers[:, :, None] == np.asarray(exclude_ids)[:, None, :]).any(-1)
    adjusted = scores - isin.astype(np.float32) * np.float32(1.0e5)
    k = min(k, scores.shape[1])
    _, idx = top_k(adjusted, k)
    return take_long_axis(scores, idx), take_long_axis(identifiers, idx)


def brute_force_top_k(queries, candidates, identifiers=None, k=10):
    """BruteForce.call (factorized_top_k.py:316-334)."""
    scores = queries.astype(np.float32) @ candidates.astype(np.float32).T
    s, idx = top_k(scores, k)
    if identifiers is None:
        identifiers = np.arange(candidates.shape[0])
    return s, np.asarray(identifiers)[idx]


def streaming_top_k(queries, candidate_batches, identifier_batches=None, k=10, handle_incomplete_batches=True):
    """Streaming.call (factorized_top_k.py:178-260): per-batch top-k (map) then merge (reduce)."""
    B = queries.shape[0]
    state_s = np.zeros((B, 0), dtype=np.float32)
    state_i = None
    counter = 0
    for bi, cand in enumerate(candidate_batches):
        n = cand.shape[0]
        ids = np.arange(counter, counter + n) if identifier_batches is None else np.asarray(identifier_batches[bi])
        counter += n
        scores = queries.astype(np.float32) @ cand.astype(np.float32).T
        k_ = min(k, n) if handle_incomplete_batches else k
        if k_ > n:
            raise ValueError("Tried to retrieve k={k} top items, but candidate batch too small.".format(k=k))
        s, idx = top_k(scores, k_)
        x_i = ids[idx]
        if state_i is None:
            state_i = np.zeros((B, 0), dtype=x_i.dtype)
        js = np.concatenate([state_s, s], axis=1)
        ji = np.concatenate([state_i, x_i], axis=1)
        k2 = min(k, js.shape[1]) if handle_incomplete_batches else k
        state_s, idx2 = top_k(js, k2)
        state_i = np.take_along_axis(ji, idx2, axis=1)
    return state_s, state_i


def ivf_flat_search(queries, candidates, identifiers, centroids, assignments, nprobe, k):
    """IVF-Flat with inner product (the reference's `Faiss` index, factorized_top_k.py:337-461 -> faiss.IndexIVFFlat with
    METRIC_INNER_PRODUCT and `nprobe`), restated from faiss' published algorithm: probe the `nprobe` lists whose centroids
    have the largest inner product with the query, score their members exactly, keep the top k (score descending, ties by
    identifier ascending); missing results are (-inf, -1).  `centroids` / `assignments` are inputs: training is not part
    of the search semantics."""
    q = np.asarray(queries, np.float32)
    c = np.asarray(candidates, np.float32)
    ids = np.arange(c.shape[0], dtype=np.int64) if identifiers is None else np.asarray(identifiers, np.int64)
    cs = q.astype(np.float64) @ np.asarray(centroids, np.float64).T
    out_s = np.full((q.shape[0], k), -np.inf, np.float32)
    out_i = np.full((q.shape[0], k), -1, np.int64)
    for r in range(q.shape[0]):
        order = np.lexsort((np.arange(cs.shape[1]), -cs[r]))[:nprobe]
        members = np.nonzero(np.isin(assignments, order))[0]
        if members.size == 0:
            continue
        sc = (c[members].astype(np.float64) @ q[r].astype(np.float64)).astype(np.float32)
        best = np.lexsort((ids[members], -sc))[:k]
        out_s[r, :best.size] = sc[best]
        out_i[r, :best.size] = ids[members][best]
    return out_s, out_i


def in_top_k(targets, predictions, k):
    """[TF] B14: target in top-k iff fewer than k entries are strictly greater than its score."""
    predictions = np.asarray(predictions)
    t = predictions[np.arange(predictions.shape[0]), np.asarray(targets)]
    return (predictions > t[:, None]).sum(axis=1) < k


def factorized_top_k_accuracy(queries, true_candidates, candidates, ks, k=100):
    """FactorizedTopK.update_state/result (factorized_top_k.py:489-522) for one batch."""
    pos = (queries * true_candidates).sum(axis=1, keepdims=True).astype(np.float32)
    topk_scores, _ = brute_force_top_k(queries, candidates, k=min(k, candidates.shape[0]))
    y_pred = np.concatenate([pos, topk_scores], axis=1)
    return [float(in_top_k(np.zeros(len(y_pred), dtype=np.int64), y_pred, kk).mean()) for kk in ks]


# ---------------------------------------------------------------------------------------------------------------------------
# xDeepFM CIN (keras/models/ranking/xdeepfm.py:71-96) and DIN ActivationUnit (keras/models/ranking/din.py:59-70)
# ---------------------------------------------------------------------------------------------------------------------------
def _activation(name):
    if name in (None, "linear"):
        return lambda v: v
    if name == "relu":
        return lambda v: np.maximum(v, 0.0)
    if name == "sigmoid":
        return sigmoid
    if name == "tanh":
        return np.tanh
    raise ValueError(name)


def cin(x0, x, kernel, bias=None, activation="sigmoid"):
    """Follows the reference's op sequence literally: split along D (:80-82), matmul(x0_d, x_d^T) per d (:84), reshape to
    [D, B, H0 * Hk] (:85), transpose to [B, D, H0 * Hk] (:86), conv1d with a width-1 kernel [1, H0 * Hk, Fm] (:88),
    bias_add (:90-91), activation (:93), transpose to [B, Fm, D] (:94).  kernel: [H0 * Hk, Fm] or [1, H0 * Hk, Fm]."""
    x0 = np.asarray(x0)
    x = np.asarray(x)
    kernel = np.asarray(kernel)
    if kernel.ndim == 3:
        kernel = kernel[0]
    B, H0, D = x0.shape
    Hk = x.shape[1]
    x0s = [x0[:, :, d:d + 1] for d in range(D)]                   # tf.split(x0, field_dim, axis=-1): D x [B, H0, 1]
    xs = [x[:, :, d:d + 1] for d in range(D)]
    outer = np.stack([np.matmul(a, np.transpose(b, (0, 2, 1))) for a, b in zip(x0s, xs)], axis=0)   # [D, B, H0, Hk]
    outer = outer.reshape(D, B, H0 * Hk)
    outer = np.transpose(outer, (1, 0, 2))                         # [B, D, H0 * Hk]
    conv_out = outer @ kernel                                      # width-1 VALID conv1d == per-position matmul
    if bias is not None:
        conv_out = conv_out + np.asarray(bias)
    return np.transpose(_activation(activation)(conv_out), (0, 2, 1))


def activation_unit(x, y, kernel_w, kernel_b, output_w, output_b, interacter=None, activation="relu"):
    """din.py:59-70: y defaults to x; concat([x, y(, interacter([x, y]))], axis=1) -> Dense(units, activation) -> Dense(1)."""
    x = np.asarray(x)
    y = x if y is None else np.asarray(y)
    h = np.concatenate([x, y], axis=1)
    if interacter is not None:
        h = np.concatenate([h, interacter([x, y])], axis=1)
    h = dense(h, kernel_w, kernel_b, activation)
    return dense(h, output_w, output_b, None)
