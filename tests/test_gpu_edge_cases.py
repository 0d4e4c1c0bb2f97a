"""Edge cases of the hot path's entry points, through the C ABI: empty batches (B == 0 is DR_OK and launches nothing), batches in
which EVERY id is missing (-1: zero embeddings, no first-order term, no table row may change), and a batch of one example.
The reference reaches these through tf.feature_column's dropping of "" / -1 entries and through the last,
short batch of a dataset epoch (datasets/movielens.py:170-186 does not drop the remainder)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_empty_batch_is_a_no_op_everywhere():
    from deep_recommenders_amd import ops
    F, D, V = 5, 64, 100
    R = F * V
    rb = _dev((np.arange(F) * V).astype(np.int64))
    table = torch.randn((R, D), device="cuda")
    lin = torch.randn((R,), device="cuda")
    t0, l0 = table.clone(), lin.clone()
    ids = torch.empty((0, F), dtype=torch.int64, device="cuda")
    out = ops.hash_bucket_i64(ids, _dev(np.full(F, V, dtype=np.int64)))
    assert out.shape == (0, F)
    concat, sum_x, fm = ops.emb_pool_fwd(ids, F, None, rb, table, lin, None)
    assert concat.shape[0] == 0 and sum_x.shape[0] == 0
    plan = ops.emb_sort_slots(ids, rb, R, plan=ops.SortPlan(16, "cuda"))
    bias = torch.zeros(1, device="cuda")
    ops.emb_pool_bwd_sorted(ids, rb, plan, D, R, torch.empty((0, F * D), device="cuda"), torch.empty((0,), device="cuda"), -0.1,
                            table, lin, bias)
    W = torch.randn((F * D, 32), device="cuda")
    wp = ops.WeightPlanes(W)
    y = ops.bf3_linear_nt(torch.empty((0, F * D), device="cuda"), wp.wt)
    assert y.shape == (0, 32)
    torch.cuda.synchronize()
    assert torch.equal(table, t0) and torch.equal(lin, l0) and bias.item() == 0.0


@pytest.mark.parametrize("B", [1, 700])
def test_all_ids_missing_touches_no_row(B):
    """every slot -1: the pooled embeddings and FM terms are zero, the logit is the bias, K4 moves only the bias"""
    from deep_recommenders_amd import ops
    F, D, V = 7, 64, 50
    R = F * V
    g = torch.Generator(device="cuda").manual_seed(B)
    rb = _dev((np.arange(F) * V).astype(np.int64))
    table = torch.randn((R, D), device="cuda", generator=g)
    lin = torch.randn((R,), device="cuda", generator=g)
    lin_bias = torch.full((1,), 0.25, device="cuda")
    t0, l0 = table.clone(), lin.clone()
    ids = torch.full((B, F), -1, dtype=torch.int64, device="cuda")
    concat, sum_x, fm = ops.emb_pool_fwd(ids, F, None, rb, table, lin, lin_bias)
    assert concat.abs().max().item() == 0.0 and sum_x.abs().max().item() == 0.0
    assert torch.equal(fm, torch.full((B,), 0.25, device="cuda"))
    # the fused first layer agrees: output = act(bias), side outputs as above
    N = 64
    W = torch.randn((F * D, N), device="cuda", generator=g) * 0.1
    b = torch.randn((N,), device="cuda", generator=g)
    wp = ops.WeightPlanes(W)
    sx2, fm2 = torch.empty((B, D), device="cuda"), torch.empty((B,), device="cuda")
    y = torch.empty((B, N), device="cuda")
    ops.bf3_emb_linear_fwd(ids, rb, V, table, lin, lin_bias, None, None, F * D, wp.wt, b, 1, sx2, fm2, y)
    assert torch.equal(y, torch.relu(b).expand(B, N)) and sx2.abs().max().item() == 0.0 and torch.equal(fm2, fm)
    # backward: plan + K4 (with and without the snapshot buffer), only the bias moves
    plan = ops.emb_sort_slots(ids, rb, R)
    dl = torch.randn((B,), device="cuda", generator=g)
    grad = torch.randn((B, F * D), device="cuda", generator=g)
    for xs in (None, torch.full((B * F, D), float("nan"), device="cuda")):
        bias = torch.zeros(1, device="cuda")
        if xs is not None:
            ops.emb_snapshot_sorted_rows(plan, table, R, xs)
        ops.emb_pool_bwd_sorted(ids, rb, plan, D, R, grad, dl, -0.5, table, lin, bias, sum_x=sum_x, x_sorted=xs,
                                concat=concat if xs is None else None)
        torch.cuda.synchronize()
        assert torch.equal(table, t0) and torch.equal(lin, l0)
        assert abs(bias.item() - (-0.5 * dl.double().sum().item())) <= 1e-5 * (1 + dl.abs().sum().item())
