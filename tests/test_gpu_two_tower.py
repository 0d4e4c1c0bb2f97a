"""GPU tests of the composed two-tower (DSSM-style) engine, BASELINE config 5 — query / item towers + `Retrieval`'s in-batch
softmax (keras/models/retrieval/sbcnm.py:120-163 of the reference) + the FactorizedTopK metric pass
(factorized_top_k.py:489-512) — against the CPU oracle (torch autograd of the restated loss; NumPy metric procedure)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tf_semantics as O
from oracle import torch_ref as T


def _oracle_step(eng0, uid, iid, lr, inv_t, cand_ids, sample_weight=None):
    """Plain SGD on the restated loss, all parameters as float32 leaf tensors on the CPU."""
    P = {k: v.clone().requires_grad_(True) for k, v in eng0.items()}

    def tower(x, pre):
        n = sum(1 for k in P if k.startswith(pre + "W"))
        for i in range(n):
            x = x @ P["%sW%d" % (pre, i)] + P["%sb%d" % (pre, i)]
            if i < n - 1:
                x = torch.relu(x)
        return x
    q = tower(P["user_table"][uid], "q")
    c = tower(P["item_table"][iid], "c")
    loss = T.inbatch_softmax_loss(q, c, sample_weight=sample_weight, cand_ids=cand_ids,
                                  temperature=(1.0 / inv_t) if inv_t != 1.0 else None)
    loss.backward()
    return loss.item(), {k: (v - lr * v.grad).detach() for k, v in P.items()}, q.detach(), c.detach()


def _params(eng):
    d = {"user_table": eng.user_table.cpu().clone(), "item_table": eng.item_table.cpu().clone()}
    for pre, t in (("q", eng.q_tower), ("c", eng.c_tower)):
        for i, (W, b) in enumerate(zip(t.Ws, t.bs)):
            d["%sW%d" % (pre, i)] = W.cpu().clone().contiguous()
            d["%sb%d" % (pre, i)] = b.cpu().clone()
    return d


@pytest.mark.parametrize("units,temperature,accidental", [((64, 32), None, True), ((), 0.5, False), ((48,), 2.0, True)])
def test_two_tower_steps_match_oracle(units, temperature, accidental):
    from deep_recommenders_amd.two_tower_engine import TwoTowerEngine
    from oracle import tf_semantics as O
    Vu, Ni, D, B, lr = 3000, 2000, 32, 512, 0.05
    eng = TwoTowerEngine(Vu, Ni, D, units, B, lr=lr, temperature=temperature, remove_accidental_hits=accidental, seed=5)
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    for step in range(2):
        keys = torch.randint(0, 10**12, (B,), device="cuda", generator=g)
        items = torch.randint(0, Ni if step == 0 else 300, (B,), device="cuda", generator=g)   # step 1: many duplicate items
        p0 = _params(eng)
        loss = eng.train_step(keys, items).item()
        uid = O.hash_bucket_i64(keys.cpu().numpy(), Vu)
        assert np.array_equal(eng.uid[:, 0].cpu().numpy(), uid), "hashed user ids differ from the oracle"
        lo, want, _, _ = _oracle_step(p0, torch.tensor(uid), items.cpu(), lr, eng.inv_t, items.cpu() if accidental else None)
        assert abs(loss - lo) <= 1e-5 * abs(lo), (step, loss, lo)                        # north_star: 1e-5 relative on the loss
        got = _params(eng)
        for k in want:
            np.testing.assert_allclose(got[k].numpy(), want[k].numpy(), rtol=2e-4, atol=2e-6, err_msg="%s step %d" % (k, step))


def test_two_tower_metric_pass_matches_oracle():
    from deep_recommenders_amd.two_tower_engine import TwoTowerEngine
    Vu, Ni, D, B = 500, 5000, 16, 256
    eng = TwoTowerEngine(Vu, Ni, D, (32, 16), B, seed=2, k=100)
    with pytest.raises(AssertionError):
        eng.metric_step(torch.zeros(B, dtype=torch.int64, device="cuda"), torch.zeros(B, dtype=torch.int64, device="cuda"))
    corpus = eng.index_corpus(chunk=1024).cpu().numpy()
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    keys = torch.randint(0, 10**9, (B,), device="cuda", generator=g)
    items = torch.randint(0, Ni, (B,), device="cuda", generator=g)
    ks = (1, 5, 10, 50, 100)
    hits = eng.metric_step(keys, items, ks).cpu().numpy()
    q, c = eng.embeddings(keys, items)
    # corpus rows are the item tower's outputs (index over candidates.map(item_model))
    np.testing.assert_allclose(corpus[items.cpu().numpy()], c.cpu().numpy(), rtol=1e-5, atol=1e-6)
    want = O.factorized_top_k_accuracy(q.cpu().numpy(), c.cpu().numpy(), corpus, ks, k=100)
    got = (hits / B).tolist()
    # scores straddling the k-th place by an ulp may flip a hit: allow one example per k
    for a, b in zip(got, want):
        assert abs(a - b) <= 1.0 / B + 1e-9, (got, want)
    s, ids = eng.topk(q, 10)
    ws, wi = O.brute_force_top_k(q.cpu().numpy(), corpus, k=10)
    np.testing.assert_allclose(s.cpu().numpy(), ws, rtol=1e-5, atol=1e-6)
    assert (ids.cpu().numpy() == wi).mean() > 0.999


def test_two_tower_training_step_and_the_operand_split():
    """Where the operand-split switch (dr_set_gemm_split, f16x2 by default) reaches into the two-tower TRAINING step at config 5's shapes
    (B = 8192, towers 128 -> 256 -> 128), and where it does not.  The towers run the generic dr_linear_* GEMMs, whose products follow
    dr_set_gemm_mode: from identical state under both splits the first step's tower outputs are BIT-identical.  The in-batch softmax's
    two score passes (8192 x 8192 x 128) DO follow the switch since round 5 (register-split kernel with the LSE / softmax-gradient
    epilogues in the f16x2 split, the fp32 MFMA kernel otherwise): the losses agree to the products' rounding (1e-6 relative), the
    parameters after two steps to the run-to-run noise of the bias-gradient atomics.  (The metric pass's scan is on the switch too:
    test_gpu_h2_gemm.py.)"""
    from deep_recommenders_amd import ops
    from deep_recommenders_amd.two_tower_engine import TwoTowerEngine
    Vu, Ni, D, B = 50000, 20000, 128, 8192
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    batches = [(torch.randint(0, 10**12, (B,), device="cuda", generator=g), torch.randint(0, Ni, (B,), device="cuda", generator=g))
               for _ in range(2)]
    prev = ops.get_gemm_split()
    res = []
    try:
        for split in ("f16x2", "bf16x3"):
            ops.set_gemm_split(split)
            eng = TwoTowerEngine(Vu, Ni, D, (256, 128), B, lr=1e-5, seed=5)
            assert not hasattr(eng, "h2")
            l1 = eng.train_step(*batches[0]).item()
            q1, c1 = eng.q_tower.hs[-1].clone(), eng.c_tower.hs[-1].clone()
            l2 = eng.train_step(*batches[1]).item()
            torch.cuda.synchronize()
            res.append((l1, q1, c1, l2, _params(eng)))
    finally:
        ops.set_gemm_split(prev)
    a, b = res
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert abs(a[0] - b[0]) <= 2e-6 * abs(b[0]) and abs(a[3] - b[3]) <= 2e-6 * abs(b[3]), (a[0], b[0], a[3], b[3])
    for k in a[4]:
        np.testing.assert_allclose(a[4][k].numpy(), b[4][k].numpy(), rtol=1e-5, atol=1e-7, err_msg=k)
