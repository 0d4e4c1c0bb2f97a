"""Writes tests/golden/tf_semantics_fixtures.json: expected values for the [TF] rules the reference's own tests do not pin
(mean-combiner / -1 / empty bag, the three losses, CCE-from-logits SUM, tf.train Adam on sparse
gradients, top_k / in_top_k tie rules).

These are NOT outputs of TensorFlow (it cannot run here).  They come from a SECOND restatement of SURVEY.md Appendix B written
independently of oracle/tf_semantics.py and oracle/torch_ref.py and structured differently: scalar Python loops over
`math` / `fractions`-free float64, no NumPy broadcasting, no torch.  What the fixtures buy is protection against transcription
and vectorisation slips in the oracle the parity tests trust (two derivations of the same written rule must agree), and a
frozen record of the edge-value behaviour (p -> 0 / 1, |x| large, empty bags, duplicate ids).  They do not turn "parity
unpinned" into "pinned": that needs a TensorFlow-produced vector, which this container cannot make (oracle/README.md).

Each case stores its inputs, so the test feeds exactly these to the oracle.
"""
import json
import math
import os
import random

HERE = os.path.dirname(os.path.abspath(__file__))


# ---- B5: safe_embedding_lookup_sparse, combiner = mean -----------------------------------------------------------------
def mean_pool(table, bag):
    d = len(table[0])
    kept = [i for i in bag if i >= 0]                 # (1) drop ids < 0
    if not kept:
        return [0.0] * d                              # (2) empty row -> zeros
    acc = [0.0] * d
    for i in kept:                                    # (3) sum in id order, one divide
        for j in range(d):
            acc[j] += table[i][j]
    return [a / len(kept) for a in acc]


# ---- B3 / B7: indicator multi-hot count, linear model ------------------------------------------------------------------
def multi_hot(bag, depth):
    out = [0.0] * depth
    for i in bag:
        if i >= 0:
            out[i] += 1.0                             # duplicates add
    return out


def linear_model(bags_per_col, w_per_col, bias):
    n = len(bags_per_col[0])
    out = []
    for b in range(n):
        s = bias
        for col, w in zip(bags_per_col, w_per_col):
            mh = multi_hot(col[b], len(w))
            s += sum(m * wi for m, wi in zip(mh, w))  # multi_hot @ weights
        out.append(s)
    return out


# ---- B9 / B10 / B11: the three losses ----------------------------------------------------------------------------------
def sigmoid_ce(z, x):
    return sum(max(xi, 0.0) - xi * zi + math.log1p(math.exp(-abs(xi))) for zi, xi in zip(z, x)) / len(x)


def log_loss(z, p, eps=1e-7):
    return sum(-zi * math.log(pi + eps) - (1 - zi) * math.log(1 - pi + eps) for zi, pi in zip(z, p)) / len(p)


def keras_bce(z, p, eps=1e-7):
    tot = 0.0
    for zi, pi in zip(z, p):
        pi = min(max(pi, eps), 1 - eps)               # clip first
        tot += -(zi * math.log(pi + eps) + (1 - zi) * math.log(1 - pi + eps))
    return tot / len(p)


# ---- B12: CategoricalCrossentropy(from_logits=True, reduction=SUM) -----------------------------------------------------
def cce_sum(labels, scores, w=None):
    tot = 0.0
    for r, (y, s) in enumerate(zip(labels, scores)):
        m = max(s)
        lse = m + math.log(sum(math.exp(v - m) for v in s))
        per = -sum(yi * (si - lse) for yi, si in zip(y, s))
        tot += per * (w[r] if w is not None else 1.0)
    return tot


# ---- B15: tf.train.AdamOptimizer on IndexedSlices gradients (non-lazy) and the row-wise "lazy" form of the kernels -----
def adam_sparse(var, steps, lr, b1=0.9, b2=0.999, eps=1e-8, lazy=False):
    rows, d = len(var), len(var[0])
    var = [list(r) for r in var]
    m = [[0.0] * d for _ in range(rows)]
    v = [[0.0] * d for _ in range(rows)]
    for t, (ids, grads) in enumerate(steps, start=1):
        lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        g = {}
        for i, gi in zip(ids, grads):                 # duplicate ids: gradients summed first
            acc = g.setdefault(i, [0.0] * d)
            for j in range(d):
                acc[j] += gi[j]
        for r in range(rows):
            if lazy and r not in g:
                continue                              # lazy: untouched rows keep their moments and value
            gr = g.get(r, [0.0] * d)
            for j in range(d):
                m[r][j] = b1 * m[r][j] + (1 - b1) * gr[j]
                v[r][j] = b2 * v[r][j] + (1 - b2) * gr[j] * gr[j]
                var[r][j] -= lr_t * m[r][j] / (math.sqrt(v[r][j]) + eps)
    return var, m, v


# ---- B13 / B14: top_k (ties -> lower index) and in_top_k (strictly-greater count < k) ----------------------------------
def top_k_row(row, k):
    order = sorted(range(len(row)), key=lambda j: (-row[j], j))[:k]
    return [row[j] for j in order], order


def in_top_k_row(target, row, k):
    return sum(1 for v in row if v > row[target]) < k


def main():
    rnd = random.Random(7)
    f32 = lambda x: float(__import__("struct").unpack("f", __import__("struct").pack("f", x))[0])   # inputs are fp32-exact
    fx = {}
    table = [[f32(rnd.uniform(-1, 1)) for _ in range(3)] for _ in range(7)]
    bags = [[0, 1, 2, -1], [-1, -1, -1, -1], [3, 3, 3, 3], [-1, 6, -1, 5], [4, -1, -1, -1], [2, 2, 5, -1]]
    fx["mean_pool"] = {"table": table, "ids": bags, "expected": [mean_pool(table, b) for b in bags]}
    cols = [[[0, 2], [1, 1], [-1, -1], [2, -1]], [[3], [-1], [0], [3]]]
    ws = [[f32(rnd.uniform(-1, 1)) for _ in range(3)], [f32(rnd.uniform(-1, 1)) for _ in range(4)]]
    fx["linear_model"] = {"ids_per_col": cols, "w_per_col": ws, "bias": 0.25,
                          "multi_hot_col0": [multi_hot(b, 3) for b in cols[0]],
                          "expected": linear_model(cols, ws, 0.25)}
    x = [-100.0, -20.0, -1e-3, 0.0, 1e-3, 20.0, 100.0, 3.5, -3.5, 88.0]
    z = [0.0, 1.0, 1.0, 0.0, 1.0, 0.0, 1.0, 1.0, 0.0, 0.0]
    fx["sigmoid_ce"] = {"labels": z, "logits": x, "expected": sigmoid_ce(z, x),
                        "per_example": [sigmoid_ce([zi], [xi]) for zi, xi in zip(z, x)]}
    p = [0.0, 1e-9, 1e-7, 0.25, 0.5, 0.75, 1 - 1e-6, 1.0, 1.0, 0.0]
    p = [f32(v) for v in p]
    zp = [0.0, 0.0, 1.0, 1.0, 0.0, 1.0, 1.0, 1.0, 0.0, 1.0]
    fx["log_loss"] = {"labels": zp, "predictions": p, "expected": log_loss(zp, p)}
    fx["keras_bce"] = {"labels": zp, "predictions": p, "expected": keras_bce(zp, p)}
    S = [[f32(rnd.uniform(-4, 4)) for _ in range(5)] for _ in range(4)]
    S[2][1] = -3.0e36                                  # a masked logit (MIN_FLOAT-sized) must vanish from the softmax
    Y = [[1.0 if i == j else 0.0 for j in range(5)] for i in range(4)]
    Wt = [1.0, 0.5, 2.0, 0.0]
    fx["cce_sum"] = {"labels": Y, "scores": S, "expected": cce_sum(Y, S), "sample_weight": Wt,
                     "expected_weighted": cce_sum(Y, S, Wt)}
    var = [[f32(rnd.uniform(-1, 1)) for _ in range(2)] for _ in range(5)]
    steps = []
    for ids in ([0, 2, 2, 4], [2, 3], [0, 0, 0, 1]):      # duplicates, a row touched once then left alone, a late first touch
        steps.append((ids, [[f32(rnd.uniform(-1e-3, 1e-3)) for _ in range(2)] for _ in ids]))
    for lazy in (False, True):
        v_, m_, s_ = adam_sparse(var, steps, 0.01, lazy=lazy)
        fx["adam_lazy" if lazy else "adam_tf"] = {"var": var, "steps": [{"ids": i, "grads": g} for i, g in steps], "lr": 0.01,
                                                  "expected_var": v_, "expected_m": m_, "expected_v": s_}
    rows = [[0.5, 0.9, 0.9, 0.1, 0.9], [1.0, 1.0, 1.0, 1.0, 1.0], [0.3, -0.2, 0.3, 0.7, 0.3]]
    fx["top_k"] = {"x": rows, "k": 3, "expected": [top_k_row(r, 3) for r in rows]}
    fx["in_top_k"] = {"predictions": rows, "targets": [2, 4, 0], "ks": [1, 2, 3, 4],
                      "expected": [[in_top_k_row(t, r, k) for k in (1, 2, 3, 4)] for t, r in zip([2, 4, 0], rows)]}
    path = os.path.join(HERE, "tf_semantics_fixtures.json")
    with open(path, "w") as fh:
        json.dump(fx, fh, indent=0)
    print("wrote", path)


if __name__ == "__main__":
    main()
