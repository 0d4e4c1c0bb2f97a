"""ORACLE — test infrastructure only (see oracle/README.md).

torch-CPU restatement of the same math as oracle/tf_semantics.py, used for two things only:
  1. gradients: the reference has no backward code (TF autodiff) so the gradient oracle is
     torch autograd over this restatement (its forward is checked against tf_semantics in tests/);
  2. bench.py's `cpu_baseline` leg ("port"): one DeepFM training step on the host cores.
Never imported by deep_recommenders_amd/.
"""
import torch


def pool_fields(table, ids, col_start, row_base):
    """[TF] B5 mean-combiner per field.  ids [B, C] (-1 = missing) -> list of F tensors [B, D]."""
    out = []
    F = len(row_base)
    for f in range(F):
        cols = ids[:, col_start[f]:col_start[f + 1]]
        mask = cols >= 0
        rows = (cols.clamp(min=0) + row_base[f])
        e = table[rows] * mask.unsqueeze(-1).to(table.dtype)
        acc = e[:, 0]
        for l in range(1, e.shape[1]):          # left-to-right, id order
            acc = acc + e[:, l]
        cnt = mask.sum(1).clamp(min=1).to(table.dtype).unsqueeze(-1)
        out.append(acc / cnt)
    return out


def first_order(lin_w, ids, col_start, row_base, bias):
    B = ids.shape[0]
    lin = torch.zeros(B, dtype=lin_w.dtype)
    for f in range(len(row_base)):
        cols = ids[:, col_start[f]:col_start[f + 1]]
        mask = cols >= 0
        rows = cols.clamp(min=0) + row_base[f]
        lin = lin + (lin_w[rows] * mask.to(lin_w.dtype)).sum(1)
    return lin + bias


def fm_second_order(stack):
    """keras/models/ranking/fm.py:28-35"""
    s = stack.sum(1)
    return 0.5 * (s * s - (stack * stack).sum(1)).sum(1)


def emb_fm_forward(table, lin_w, bias, ids, col_start, row_base):
    embs = pool_fields(table, ids, col_start, row_base)
    stack = torch.stack(embs, 1)
    concat = torch.cat(embs, 1)
    logit = fm_second_order(stack)
    if lin_w is not None:
        logit = logit + first_order(lin_w, ids, col_start, row_base, bias)
    return concat, stack.sum(1), logit


def dnn(x, kernels, biases, relu_masks=None, ties=None, tie_eps=1e-5):
    """keras deepfm.py:30-34 / estimator dnn.py:17-29: relu hidden layers, linear last layer.

    Tie-aware comparison (relu_masks).  relu(z) has no derivative at z = 0, and two correct fp32
    implementations of z = x W + b differ in its last bits: of the ~10^7 hidden units of a 65 536-example batch a handful have
    |z| below that rounding, and each side picks "on" or "off" for them -- a different subgradient, an equally valid one, that
    moves a whole row of the example's input gradient.  relu_masks[i] (bool [B, units_i], the units the IMPLEMENTATION UNDER TEST
    treated as on in hidden layer i) makes the oracle evaluate h = z * mask instead of relu(z): the same function wherever
    mask == (z > 0).  Every unit where they disagree is recorded in `ties` (a list, one dict per layer) and must be a tie:
    |z_oracle| <= tie_eps * rms(z) -- the caller asserts that with check_ties(); a mask that is wrong anywhere else fails there,
    so the mask cannot hide an error larger than the rounding of z.  relu_masks = None: plain relu."""
    n_hidden = len(kernels) - 1
    for i, (W, b) in enumerate(zip(kernels[:-1], biases[:-1])):
        z = x @ W + b
        if relu_masks is None:
            x = torch.relu(z)
            continue
        m = relu_masks[i]
        assert m.dtype == torch.bool and m.shape == z.shape and len(relu_masks) == n_hidden
        zd = z.detach()
        dis = m != (zd > 0)
        if ties is not None:
            rms = float(zd.double().pow(2).mean().sqrt())
            worst = float(zd[dis].abs().max()) if bool(dis.any()) else 0.0
            ties.append({"layer": i, "units": dis.numel(), "disagree": int(dis.sum()), "worst_abs_z": worst, "rms_z": rms,
                         "examples": torch.nonzero(dis.any(1)).reshape(-1).tolist()[:64], "tie_eps": tie_eps})
        x = z * m.to(z.dtype)
    return x @ kernels[-1] + biases[-1]


def check_ties(ties, max_frac=1e-5):
    """The assertion half of dnn(relu_masks=...): every unit where the device's ReLU decision differs from the oracle's is within
    tie_eps * rms(z) of zero (the two sides' rounding of z), and there are at most max_frac of them (measured: 0 - 4 of 1.7e7)."""
    for t in ties:
        assert t["worst_abs_z"] <= t["tie_eps"] * t["rms_z"], "ReLU mask of layer %d is wrong at a unit with |z| = %.3e (rms z %.3e): not a tie" % (
            t["layer"], t["worst_abs_z"], t["rms_z"])
        assert t["disagree"] <= max(4, max_frac * t["units"]), "layer %d: %d of %d ReLU decisions differ from the oracle's" % (
            t["layer"], t["disagree"], t["units"])


def deepfm_logit(table, lin_w, bias, ids, col_start, row_base, kernels, biases, dense=None):
    concat, _, fm_logit = emb_fm_forward(table, lin_w, bias, ids, col_start, row_base)
    x = concat if dense is None else torch.cat([concat, dense], 1)
    return fm_logit + dnn(x, kernels, biases).squeeze(1)


def cross(x0, x, W, b, diag_scale=0.0):
    """keras/models/ranking/dcn.py:81-88"""
    prod = x @ W
    if b is not None:
        prod = prod + b
    if diag_scale:
        prod = prod + diag_scale * x
    return x0 * prod + x


def sigmoid_cross_entropy(labels, logits):
    """[TF] B9: mean of max(x, 0) - x z + log1p(exp(-|x|)).  max(x, 0) + log1p(exp(-|x|)) is the stable evaluation of
    softplus(x) = log(1 + e^x); written here as logaddexp(x, 0) -- the same value -- because autograd's subgradients of max / abs
    at EXACTLY x = 0 sum to 1 instead of sigmoid(0) = 1/2 (found by pinning the gradient against torch's BCEWithLogitsLoss,
    tests/test_oracle_third_party_pins.py; TF registers the analytic gradient sigmoid(x) - z, as the kernels compute)."""
    return (torch.logaddexp(logits, torch.zeros_like(logits)) - logits * labels).mean()


def log_loss(labels, p, eps=1e-7):
    """[TF] B10"""
    return (-labels * torch.log(p + eps) - (1 - labels) * torch.log(1 - p + eps)).mean()


def keras_bce(labels, p, eps=1e-7):
    """[TF] B11"""
    p = p.clamp(eps, 1 - eps)
    return (-(labels * torch.log(p + eps) + (1 - labels) * torch.log(1 - p + eps))).mean()


def inbatch_softmax_loss(q, c, sample_weight=None, cand_prob=None, cand_ids=None, temperature=None):
    """Retrieval.call all-optional-branches restatement (sbcnm.py:120-151)."""
    import numpy as np
    scores = q @ c.t()
    B = q.shape[0]
    labels = torch.eye(B, dtype=q.dtype)
    if cand_prob is not None:
        scores = scores - torch.log(cand_prob)
    if cand_ids is not None:
        ident = cand_ids.reshape(-1, 1)
        dup = (ident == ident.t()).to(q.dtype) - labels
        scores = scores + dup * float(np.finfo(np.float32).min / 100.0)
    if temperature is not None:
        scores = scores / temperature
    per = torch.logsumexp(scores, 1) - (scores * labels).sum(1)
    if sample_weight is not None:
        per = per * sample_weight
    return per.sum()


def deepfm_train_step_sgd(params, ids, dense, labels, col_start, row_base, lr):
    """One SGD step of DeepFM (forward, log-loss, backward, update) on the host — the cpu_baseline port.
    Embedding gradients are sparse (rows touched), like TF's IndexedSlices path."""
    table, lin_w, bias, kernels, biases = params
    F = len(row_base)
    rows = ids + torch.as_tensor(row_base)[None, :]              # single-valued fields
    emb = torch.nn.functional.embedding(rows, table)            # [B, F, D]
    emb.requires_grad_(True)
    lw = lin_w[rows]
    lw.requires_grad_(True)
    ks = [k.requires_grad_(True) for k in kernels]
    bs = [b.requires_grad_(True) for b in biases]
    concat = emb.reshape(emb.shape[0], -1)
    x = concat if dense is None else torch.cat([concat, dense], 1)
    logit = fm_second_order(emb) + lw.sum(1) + bias + dnn(x, ks, bs).squeeze(1)
    loss = sigmoid_cross_entropy(labels, logit)
    grads = torch.autograd.grad(loss, [emb, lw] + ks + bs)
    with torch.no_grad():
        table.index_add_(0, rows.reshape(-1), grads[0].reshape(-1, table.shape[1]), alpha=-lr)
        lin_w.index_add_(0, rows.reshape(-1), grads[1].reshape(-1), alpha=-lr)
        for p, g in zip(ks + bs, grads[2:]):
            p.sub_(lr * g)
    return float(loss)


def adam_lr_t(lr, beta1, beta2, step):
    """[TF] B15 (tf.train.AdamOptimizer, examples/train_fm_on_movielens_estimator.py:51 of the reference):
    lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t)."""
    return lr * (1.0 - beta2 ** step) ** 0.5 / (1.0 - beta1 ** step)


def adam_dense_step(p, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8):
    """[TF] B15 dense update, in place on torch tensors: m, v, then p -= lr_t * m / (sqrt(v) + eps)."""
    lr_t = adam_lr_t(lr, beta1, beta2, step)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    p.sub_(lr_t * m / (v.sqrt() + eps))


def adam_rows_step(p, dense_grad, touched_rows, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8):
    """Row-wise ("lazy") Adam: the [TF] B15 update applied only to `touched_rows` of p / m / v, with `dense_grad` the
    gradient already summed over duplicate ids (what autograd's index_add produces).  Equal to TF's non-lazy Adam on
    step 1 (all moments zero) and for rows touched on every step; documented divergence otherwise."""
    lr_t = adam_lr_t(lr, beta1, beta2, step)
    r = torch.unique(touched_rows[touched_rows >= 0])
    g = dense_grad[r]
    m[r] = beta1 * m[r] + (1 - beta1) * g
    v[r] = beta2 * v[r] + (1 - beta2) * g * g
    p[r] = p[r] - lr_t * m[r] / (v[r].sqrt() + eps)


def ftrl_dense_step(p, g, accum, linear, lr, lr_power=-0.5, l1=0.0, l2=0.0):
    """tf.train.FtrlOptimizer dense update (the reference's examples/train_wdl_on_movielens_estimator.py:66-70), in place:
    accum' = accum + g^2; linear += g - (accum'^-p - accum^-p)/lr * w; w = |linear| > l1 ? (sign(linear) l1 - linear) /
    (accum'^-p / lr + 2 l2) : 0."""
    a1 = accum + g * g
    pa0, pa1 = accum.pow(-lr_power), a1.pow(-lr_power)
    linear.add_(g - (pa1 - pa0) / lr * p)
    quad = pa1 / lr + 2 * l2
    p.copy_(torch.where(linear.abs() > l1, (torch.sign(linear) * l1 - linear) / quad, torch.zeros_like(p)))
    accum.copy_(a1)
