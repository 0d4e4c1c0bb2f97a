"""GPU: the f16x2 operand mode under an adversarial dynamic range, through the ENGINE.

The mode carries every table value as two fp16 terms of x * 2^k with ONE k per tensor, taken from the table's amax record.  One outlier row
therefore costs every other row bits, and a field whose embeddings are tiny next to it keeps almost none.  What must still hold is
north_star's bound -- the loss within 1e-5 relative of the fp32 reference (plain fp32 tf.matmul, keras/models/ranking/deepfm.py:30-34 of the
reference) -- because what is lost is ABSOLUTE precision at 2^-39 of the tensor's largest magnitude; and the record must come back down once
the outlier is gone (DeepFMEngine.tighten_amax)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tf_semantics as O
from oracle import torch_ref as T


def _oracle_loss(eng, keys, dense, labels, F, V):
    ids = np.stack([O.hash_bucket_i64(keys[:, f].cpu().numpy(), V) for f in range(F)], axis=1)
    logit = T.deepfm_logit(eng.table.cpu(), eng.lin_w.cpu(), eng.lin_bias.cpu().reshape(()), torch.tensor(ids), list(range(F + 1)),
                           [f * V for f in range(F)], [w.cpu().contiguous() for w in eng.Ws], [b.cpu() for b in eng.bs], dense.cpu())
    return float(T.sigmoid_cross_entropy(labels.cpu(), logit).item()), ids


def test_engine_loss_holds_1e5_with_an_outlier_row_and_a_tiny_field_and_the_record_tightens():
    from deep_recommenders_amd.engine import DeepFMEngine
    from deep_recommenders_amd import ops
    F, V, D, B, Nd = 5, 4000, 64, 2304, 3
    eng = DeepFMEngine(F, V, D, [256, 32], B, num_dense=Nd, lr=0.05, seed=7, lin_init_std=0.1)
    assert eng.h2 and eng.fuse_k4
    g = torch.Generator(device="cuda")
    g.manual_seed(31)
    def batch():
        u = torch.rand((B, F), device="cuda", generator=g, dtype=torch.float64)
        keys = (1.0 / (u + 1e-5)).long()                                  # Zipf-like: hot keys, long tail
        return keys, torch.rand((B, Nd), device="cuda", generator=g), (torch.rand(B, device="cuda", generator=g) < 0.3).float()
    batches = [batch() for _ in range(3)]
    looked_up = set()
    for keys, _, _ in batches:
        ids0 = O.hash_bucket_i64(keys[:, 0].cpu().numpy(), V)
        looked_up.update(int(i) for i in ids0)
    outlier = next(r for r in range(V) if r not in looked_up)             # a row of field 0 that no batch reads: it only moves the record
    base_amax = ops.h2_amax_value(eng.tab_amax)
    with torch.no_grad():
        eng.table[outlier] *= 2.0 ** 18
        eng.table[3 * V:4 * V] *= 2.0 ** -18                              # field 3: embeddings 2^-36 of the tensor's largest magnitude
    for n, (keys, dense, labels) in enumerate(batches):
        want, ids = _oracle_loss(eng, keys, dense, labels, F, V)          # the fp32 reference on the parameters as they stand
        got = float(eng.train_step(keys, dense, labels).item())
        assert np.array_equal(eng.ids.cpu().numpy(), ids)
        assert abs(got - want) <= 1e-5 * abs(want), (n, got, want)
        if n == 0:
            assert ops.h2_amax_value(eng.tab_amax) >= 2.0 ** 17 * base_amax          # the record follows the outlier (outside write detected)
    # the outlier goes away through a path the version counter cannot see (raw pointer write = what K4 itself does when training shrinks a row):
    # the running record stays high until it is tightened
    flat = eng.table.view(-1)
    ops.axpy(-(1.0 - 2.0 ** -18), flat[outlier * D:(outlier + 1) * D].clone(), flat[outlier * D:(outlier + 1) * D])
    stale = ops.h2_amax_value(eng.tab_amax)
    eng.tighten_amax()
    tight = ops.h2_amax_value(eng.tab_amax)
    assert tight <= stale * 2.0 ** -10 and tight == float(eng.table.abs().max().item())
    keys, dense, labels = batches[0]
    want, _ = _oracle_loss(eng, keys, dense, labels, F, V)
    got = float(eng.train_step(keys, dense, labels).item())
    assert abs(got - want) <= 1e-5 * abs(want)


def test_engine_tightens_its_record_every_n_steps(monkeypatch):
    from deep_recommenders_amd.engine import DeepFMEngine
    from deep_recommenders_amd import ops
    monkeypatch.setenv("DR_AMAX_TIGHTEN_STEPS", "3")
    F, V, D, B = 4, 3000, 64, 2304
    eng = DeepFMEngine(F, V, D, [256, 32], B, num_dense=0, lr=0.05, seed=7, lin_init_std=0.1)
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    keys = torch.randint(0, 10**12, (B, F), device="cuda", generator=g)
    labels = (torch.rand(B, device="cuda", generator=g) < 0.3).float()
    eng.tab_amax.copy_(torch.tensor([2.0 ** 20], dtype=torch.float32, device="cuda").view(torch.int32))     # a stale, far too large bound
    for n in range(3):
        assert ops.h2_amax_value(eng.tab_amax) >= 2.0 ** 20
        eng.train_step(keys, None, labels)
    assert ops.h2_amax_value(eng.tab_amax) == float(eng.table.abs().max().item())
