"""GPU parity: the first-layer dgrad with K4's unique-row pass as its epilogue (dr_h2_dgrad_emb_sgd + dr_emb_pool_bwd_sorted_ex parts | 8)
against the two launches it replaces (dr_h2_linear_nt into d_concat, then dr_emb_pool_bwd_sorted_ex).  The epilogue repeats K4's
arithmetic operation for operation, so tables, first-order weights and bias must come out BIT-identical -- which carries every
oracle-backed tolerance of the K4 / engine tests over to the fused step (autodiff of keras/models/ranking/deepfm.py:30-34,44-45 and
fm.py:23-37 of the reference w.r.t. the embedding tables)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from deep_recommenders_amd import ops as _ops
    return _ops


def _setup(ops, B, F, V, H, n_dense, seed, skew=False, missing=0.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    rng = np.random.default_rng(seed)
    D = 64
    if skew:
        ids = np.minimum((rng.pareto(1.05, size=(B, F)) * 3).astype(np.int64), V - 1)      # hot rows: shared by hundreds of slots
    else:
        ids = rng.integers(0, V, size=(B, F))
    if missing:
        ids[rng.random((B, F)) < missing] = -1
    ids = torch.as_tensor(ids).cuda()
    row_base = (torch.arange(F, dtype=torch.int64) * V).cuda()
    R = F * V
    table = torch.randn((R, D), device="cuda", generator=g) * 0.125
    lin = torch.randn(R, device="cuda", generator=g) * 0.01
    in_dim = F * D + n_dense
    W = torch.randn((in_dim, H), device="cuda", generator=g) / in_dim ** 0.5
    dy = torch.randn((B, H), device="cuda", generator=g) * (torch.rand((B, H), device="cuda", generator=g) > 0.5) / B
    dl = torch.randn(B, device="cuda", generator=g) / B
    # what the forward leaves behind: sum_x, the first-order weight every slot read
    idc = ids.clamp_min(0) + row_base[None, :]
    x = table[idc] * (ids >= 0)[..., None]
    sum_x = x.sum(1).contiguous()
    lin_old_t = lin[idc].t().contiguous()
    return dict(D=D, R=R, ids=ids, row_base=row_base, table=table, lin=lin, W=W, dy=dy, dl=dl, sum_x=sum_x, lin_old_t=lin_old_t, in_dim=in_dim)


def _run(ops, s, fused, lr=0.05):
    B, F = s["ids"].shape
    D, R = s["D"], s["R"]
    table, lin, bias = s["table"].clone(), s["lin"].clone(), torch.zeros(1, device="cuda")
    plan = ops.emb_sort_slots(s["ids"], s["row_base"], R)
    wp = ops.H2WeightPlanes(s["W"])
    dy_am = ops.h2_amax(s["dy"])
    tab_am = ops.h2_amax(table)
    ld = (s["in_dim"] + 3) // 4 * 4
    d_concat = torch.full((B, ld), float("nan"), device="cuda")
    xs = torch.full((B * F, D), float("nan"), device="cuda")
    ops.emb_snapshot_sorted_rows(plan, table, R, xs)
    k4 = lambda parts: ops.emb_pool_bwd_sorted(s["ids"], s["row_base"], plan, D, R, d_concat, s["dl"], -lr, table, lin, bias, sum_x=s["sum_x"],
                                               x_sorted=xs, parts=parts, lin_old_t=s["lin_old_t"], table_amax=tab_am)
    if fused:
        ids_t = ops.ids_transpose_i32(s["ids"])
        ops.h2_dgrad_emb_sgd(s["dy"], dy_am, wp.w, ids_t, plan, s["row_base"], table, lin, s["lin_old_t"], s["sum_x"], s["dl"], -lr, d_concat,
                             table_amax=tab_am)
        k4(1 | 8)
        k4(2 | 8)
    else:
        ops.h2_linear_nt(s["dy"], dy_am, wp.w, out=d_concat[:, :s["in_dim"]])
        k4(1)
        k4(2)
    torch.cuda.synchronize()
    return table, lin, bias, tab_am.clone(), plan


@pytest.mark.parametrize("B,F,V,H,n_dense,skew,missing", [
    (1024, 6, 5000, 64, 0, False, 0.0),          # interior row tiles, 1.5 column tiles
    (1000, 5, 300, 96, 13, False, 0.05),         # ragged last row tile, many shared rows, missing ids, dense columns behind the embeddings
    (2500, 9, 200000, 256, 13, False, 0.0),      # mostly unique rows, K = 256 (the bench layer's width), 3 column tiles
    (3000, 4, 4000, 32, 0, True, 0.02),          # Zipf-like ids: hot rows through the duplicate pass
])
def test_fused_dgrad_k4_is_bit_identical_to_dgrad_then_k4(ops, B, F, V, H, n_dense, skew, missing):
    s = _setup(ops, B, F, V, H, n_dense, seed=B + F, skew=skew, missing=missing)
    t0, l0, b0, a0, plan = _run(ops, s, fused=False)
    t1, l1, b1, a1, _ = _run(ops, s, fused=True)
    uniq = int(plan.flags[:B * F].sum().item())
    assert 0 < uniq <= B * F
    assert not torch.equal(t0, s["table"])                                  # the step did move the table
    assert torch.equal(t1, t0), "tables differ: %d elements" % int((t1 != t0).sum().item())
    assert torch.equal(l1, l0)
    assert torch.equal(b1, b0)
    assert torch.equal(a1, a0)                                              # the table's amax record as K4 leaves it


def test_fused_dgrad_k4_against_fp64(ops):
    """... and the pair itself against a float64 restatement of the update (unique rows only -- where the fused epilogue acts)."""
    s = _setup(ops, 1500, 7, 100000, 128, 13, seed=3)
    lr = 0.05
    t1, l1, _, _, plan = _run(ops, s, fused=True, lr=lr)
    B, F = s["ids"].shape
    flags = plan.flags[:B * F].reshape(B, F).bool()
    dx = (s["dy"].double() @ s["W"].double().t())[:, :F * 64].reshape(B, F, 64)
    idc = s["ids"] + s["row_base"][None, :]
    x = s["table"][idc].double()
    g = dx + s["dl"].double()[:, None, None] * (s["sum_x"].double()[:, None, :] - x)
    want = x - lr * g
    got = t1[idc].double()
    err = ((got - want).abs() * flags[..., None]).max().item()
    assert err <= 2e-6 * want.abs().max().item()
    wl = s["lin"][idc].double() - lr * s["dl"].double()[:, None]
    assert ((l1[idc].double() - wl).abs() * flags).max().item() <= 1e-7


@pytest.mark.parametrize("V,zipf", [(3000, False), (200000, False), (50000, True)])
def test_engine_with_fused_k4_is_bit_identical_to_the_three_kernel_backward(V, zipf, monkeypatch):
    """DeepFMEngine's default step (wgrad -> dgrad + K4's unique rows -> duplicate pass) against DR_FUSE_K4=0 (dgrad -> wgrad -> K4):
    same arithmetic in every kernel, so parameters and losses must be bit-identical over prefetched steps -- many shared rows
    (V = 3000), mostly unique rows, and a skewed batch whose hot rows take the parked-pieces path."""
    from deep_recommenders_amd.engine import DeepFMEngine
    F, B, Nd, D = 5, 2304, 3, 64
    g = torch.Generator(device="cuda")
    g.manual_seed(21)
    def keys():
        if zipf:
            u = torch.rand((B, F), device="cuda", generator=g)
            return (1.0 / (u + 1e-4)).long()                      # heavy head: a few keys hundreds of times
        return torch.randint(0, 10**12, (B, F), device="cuda", generator=g)
    batches = [(keys(), torch.rand((B, Nd), device="cuda", generator=g), (torch.rand(B, device="cuda", generator=g) < 0.3).float())
               for _ in range(3)]

    def run():
        eng = DeepFMEngine(F, V, D, [256, 16], B, num_dense=Nd, lr=0.05, seed=3, lin_init_std=0.1)
        losses = []
        for n in range(5):
            k, d, l = batches[n % 3]
            nk, nd = batches[(n + 1) % 3][0], batches[(n + 1) % 3][1]
            losses.append(float(eng.train_step(k, d, l, next_keys=nk, next_dense=nd).item()))
        torch.cuda.synchronize()
        return eng, losses
    fused, l1 = run()
    assert fused.fuse_k4 and fused.h2
    monkeypatch.setenv("DR_FUSE_K4", "0")
    plain, l0 = run()
    assert not plain.fuse_k4
    assert l1 == l0
    assert torch.equal(fused.table, plain.table) and torch.equal(fused.lin_w, plain.lin_w) and torch.equal(fused.lin_bias, plain.lin_bias)
    assert torch.equal(fused.flat_params, plain.flat_params)
    assert torch.equal(fused.tab_amax, plain.tab_amax)
