"""The "f16x2" operand mode of the first tower layer's GEMMs (dr_h2_*, include/dr_hotpath.h): fp32 in, fp32 accumulate, every operand
value carried as two fp16 terms of x * 2^k (k per tensor, from its amax record), three matrix instructions per fragment pair.

Reference: the same products in fp64 (keras/models/ranking/deepfm.py:30-34 of the reference is a plain fp32 Dense; its autodiff the
two backward products).  Tolerances are written against what the bf16x3 mode and a plain fp32 GEMM reach on the same inputs: the mode
must be no worse than 1.5 x the bf16x3 mode's error (it measures better on every case below) and within 4e-6 of the largest output."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from deep_recommenders_amd import ops as _ops
    return _ops


def _rel(got, ref):
    ref = ref.double()
    return ((got.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-300)).item()


def _rms(got, ref):
    ref = ref.double()
    return ((got.double() - ref).pow(2).mean() / ref.pow(2).mean().clamp_min(1e-300)).sqrt().item()


def _amax_bits(t):
    return int(t.abs().max().view(torch.int32).item()) if t.numel() else 0


def test_amax_record_is_the_largest_magnitude_as_float_bits(ops):
    g = torch.Generator(device="cuda").manual_seed(0)
    for shape, ld in (((1000, 37), 40), ((513, 256), 256), ((3, 5), 5), ((70000, 32), 32)):
        buf = torch.randn((shape[0], ld), device="cuda", generator=g)
        x = buf[:, :shape[1]]
        buf[:, shape[1]:] = 1e9                                   # padding outside the matrix must not be seen
        rec = ops.h2_amax(x)
        assert int(rec.item()) == _amax_bits(x)
        assert ops.h2_amax_value(rec) == float(x.abs().max().item())
        # running maximum: reset=False keeps a larger record, raises a smaller one
        big = ops.h2_record("cuda")
        big.fill_(int(torch.tensor(1e6).view(torch.int32).item()))
        ops.h2_amax(x, big, reset=False)
        assert ops.h2_amax_value(big) == 1e6
        ops.h2_amax(x * 1e8, big, reset=False)
        assert ops.h2_amax_value(big) == float((x * 1e8).abs().max().item())
    z = ops.h2_amax(torch.zeros((8, 8), device="cuda"))
    assert int(z.item()) == 0


CASES = ["randn", "rows spanning 1e-6..1", "all tiny (1e-20)", "all huge (1e15)", "one element 3e4 x the rest"]


@pytest.mark.parametrize("M,K,N", [(3000, 1677, 256), (4096, 256, 1677), (777, 96, 130)])
@pytest.mark.parametrize("case", CASES)
def test_h2_linear_nt_against_fp64_and_the_bf16x3_mode(ops, M, K, N, case):
    g = torch.Generator(device="cuda").manual_seed(M + K + len(case))
    a = torch.randn((M, (K + 3) // 4 * 4), device="cuda", generator=g)[:, :K] * 0.1
    W = torch.randn((K, N), device="cuda", generator=g) * 0.05
    if case == CASES[1]:
        a = a * torch.pow(10.0, -6 * torch.rand((M, 1), device="cuda", generator=g))
    elif case == CASES[2]:
        a, W = a * 1e-20, W * 1e-10
    elif case == CASES[3]:
        a, W = a * 1e15, W * 1e12
    elif case == CASES[4]:
        a = a.clone()
        a[0, 0] = 3000.0
    ref = a.double() @ W.double()
    wp3, wp2 = ops.WeightPlanes(W), ops.H2WeightPlanes(W)
    assert int(wp2.amax.item()) == _amax_bits(W)
    am = ops.h2_amax(a)
    y2 = ops.h2_linear_nt(a, am, wp2.wt)
    y3 = ops.bf3_linear_nt(a, wp3.wt)
    assert torch.isfinite(y2).all()
    e2, e3 = _rel(y2, ref), _rel(y3, ref)
    assert e2 <= 4e-6 and e2 <= 1.5 * e3 + 1e-7, (e2, e3)
    assert _rms(y2, ref) <= 1.5 * _rms(y3, ref) + 1e-8
    # a record that is an UPPER bound (a running maximum 8 x too large) costs three bits, not correctness
    loose = am.clone()
    loose.copy_((a.abs().max() * 8).view(torch.int32))
    assert _rel(ops.h2_linear_nt(a, loose, wp2.wt), ref) <= 3e-5


def test_h2_linear_nt_epilogues_and_edges(ops):
    """bias + ReLU, the ReLU' mask of the dgrad, accumulate; K not a multiple of 32, M / N not multiples of 256; M = 0."""
    g = torch.Generator(device="cuda").manual_seed(9)
    for (M, K, N) in ((1000, 83, 40), (515, 300, 257), (4096, 128, 520)):
        a = torch.randn((M, (K + 3) // 4 * 4), device="cuda", generator=g)[:, :K]
        W = torch.randn((K, N), device="cuda", generator=g) * 0.1
        b = torch.randn((N,), device="cuda", generator=g)
        mask = torch.randn((M, N), device="cuda", generator=g)
        wp = ops.H2WeightPlanes(W)
        am = ops.h2_amax(a)
        ref = a.double() @ W.double() + b.double()
        assert _rel(ops.h2_linear_nt(a, am, wp.wt, bias=b, act=1), ref.clamp_min(0)) <= 3e-6
        assert _rel(ops.h2_linear_nt(a, am, wp.wt, bias=b, mask=mask), ref * (mask > 0)) <= 3e-6
        o = torch.ones((M, N), device="cuda")
        ops.h2_linear_nt(a, am, wp.wt, bias=b, accumulate=True, out=o)
        assert _rel(o, ref + 1) <= 3e-6
        # the dgrad's operand: W planes (rows = input features)
        dy = torch.randn((M, N), device="cuda", generator=g) * 1e-3
        assert _rel(ops.h2_linear_nt(dy, ops.h2_amax(dy), wp.w), dy.double() @ W.double().t()) <= 3e-6
    out = ops.h2_linear_nt(torch.zeros((0, 64), device="cuda"), ops.h2_record("cuda"), ops.H2WeightPlanes(torch.ones((64, 128), device="cuda")).wt)
    assert out.shape == (0, 128)
    with pytest.raises(RuntimeError):        # a missing record is refused, not guessed
        ops.h2_linear_nt(torch.zeros((8, 64), device="cuda"), None, ops.H2WeightPlanes(torch.ones((64, 128), device="cuda")).wt)


@pytest.mark.parametrize("M,K,N", [(3000, 1677, 256), (5000, 300, 257), (777, 96, 130)])
def test_h2_wgrad_against_fp64(ops, M, K, N):
    g = torch.Generator(device="cuda").manual_seed(K)
    x = torch.randn((M, (K + 3) // 4 * 4), device="cuda", generator=g)[:, :K] * 0.1
    dy = torch.randn((M, N), device="cuda", generator=g) * 1e-3 * torch.pow(10.0, -4 * torch.rand((M, 1), device="cuda", generator=g))
    ref = -0.5 * (x.double().t() @ dy.double())
    d3, d2 = torch.zeros((K, N), device="cuda"), torch.zeros((K, N), device="cuda")
    b3, b2 = torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")
    ops.bf3_wgrad(x, dy, -0.5, d3, b3)
    ws = ops.bf3_wgrad_workspace(M, K, N, "cuda").fill_(float("nan"))
    ops.h2_wgrad(x, ops.h2_amax(x), dy, ops.h2_amax(dy), -0.5, d2, b2, workspace=ws)
    assert _rel(d2, ref) <= 3e-6 and _rel(d2, ref) <= 1.5 * _rel(d3, ref) + 1e-7
    assert torch.equal(b2, b3)                  # the column sums are taken from the unscaled fp32 values, as before
    # accumulates into dst (dst += scale * x^T dy), deterministic
    d2b = d2.clone()
    ops.h2_wgrad(x, ops.h2_amax(x), dy, ops.h2_amax(dy), -0.5, d2b, None, workspace=ws)
    assert _rel(d2b, 2 * ref) <= 3e-6
    d2c = torch.zeros((K, N), device="cuda")
    ops.h2_wgrad(x, ops.h2_amax(x), dy, ops.h2_amax(dy), -0.5, d2c, None)
    assert torch.equal(d2c, d2)


@pytest.mark.parametrize("M,F,Nd,N", [(3000, 26, 13, 256), (700, 3, 0, 64), (70000, 5, 7, 300), (66000, 2, 32, 512)])
def test_h2_emb_linear_fwd_and_wgrad_emb(ops, M, F, Nd, N):
    """The fused first layer in the f16x2 mode: everything that is not the product -- concat, sum_x, fm_logit, the saved first-order
    weights -- is bit-identical to the bf16x3 kernel's (the same loads, the same sums); the layer output is the fp64 product within
    the mode's tolerance; dense features 50 x larger than the embeddings share the activation scale (max of the two records)."""
    g = torch.Generator(device="cuda").manual_seed(M + F)
    D, V = 64, 997
    table = torch.randn((F * V, D), device="cuda", generator=g) * 0.3
    lin_w = torch.randn((F * V,), device="cuda", generator=g)
    lin_b = torch.tensor([0.37], device="cuda")
    row_base = (torch.arange(F, device="cuda") * V).to(torch.int64)
    ids = torch.randint(0, V, (M, F), device="cuda", generator=g)
    ids[torch.rand((M, F), device="cuda", generator=g) < 0.05] = -1
    K = F * D + Nd
    ld = (K + 3) // 4 * 4
    dense = torch.randn((M, Nd), device="cuda", generator=g) * 15
    W = torch.randn((K, N), device="cuda", generator=g) * 0.1
    b = torch.randn((N,), device="cuda", generator=g)
    wp3, wp2 = ops.WeightPlanes(W), ops.H2WeightPlanes(W)
    dpad = None
    if Nd:
        dpad = torch.zeros((M, 32), device="cuda")
        dpad[:, :Nd] = dense
    tam = ops.h2_amax(table)
    dam = ops.h2_amax(dpad) if Nd else None
    outs = []
    for mode in (3, 2):
        concat = torch.zeros((M, ld), device="cuda")
        concat[:, :F * D] = float("nan")
        concat[:, F * D:K] = dense
        sx, fm = torch.full((M, D), float("nan"), device="cuda"), torch.full((M,), float("nan"), device="cuda")
        y = torch.full((M, N), float("nan"), device="cuda")
        lv = torch.zeros((F, M), device="cuda")
        if mode == 3:
            ops.bf3_emb_linear_fwd(ids, row_base, V, table, lin_w, lin_b, dpad, concat, K, wp3.wt, b, 1, sx, fm, y, lin_vals_t=lv)
        else:
            ops.h2_emb_linear_fwd(ids, row_base, V, table, tam, lin_w, lin_b, dpad, dam, concat, K, wp2.wt, b, 1, sx, fm, y, lin_vals_t=lv)
        outs.append((concat, sx, fm, y, lv))
    c3, c2 = outs
    for i in (0, 1, 2, 4):
        assert torch.equal(c3[i], c2[i])
    ref = (c3[0][:, :K].double() @ W.double() + b.double()).clamp_min(0)
    assert _rel(c2[3], ref) <= 3e-6 and _rel(c2[3], ref) <= 1.5 * _rel(c3[3], ref) + 1e-7
    # concat == None (the engine's default: nothing stores the gathered rows) gives the same output
    y_nc = torch.empty((M, N), device="cuda")
    ops.h2_emb_linear_fwd(ids, row_base, V, table, tam, lin_w, lin_b, dpad, dam, None, K, wp2.wt, b, 1, c2[1], c2[2], y_nc)
    assert torch.equal(y_nc, c2[3])
    # the gathering wgrad
    dy = torch.randn((M, N), device="cuda", generator=g) * 1e-3
    ids_t = ids.t().contiguous().to(torch.int32)
    refw = c3[0][:, :K].double().t() @ dy.double()
    d3, d2 = torch.zeros((K, N), device="cuda"), torch.zeros((K, N), device="cuda")
    db3, db2 = torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")
    ops.bf3_wgrad_emb(ids_t, row_base, table, dpad, dy, 1.0, d3, db3)
    dyam = ops.h2_amax(dy)
    ops.h2_wgrad_emb(ids_t, row_base, table, tam, dpad, dam, dy, dyam, 1.0, d2, db2)
    assert _rel(d2, refw) <= 3e-6 and _rel(d2, refw) <= 1.5 * _rel(d3, refw) + 1e-7
    assert torch.equal(db2, db3)
    # in two halves (GEMM into the workspace, then the reduce): the same bits
    d2p, db2p = torch.zeros((K, N), device="cuda"), torch.zeros(N, device="cuda")
    ws = ops.bf3_wgrad_workspace(M, K, N, "cuda")
    ops.h2_wgrad_emb(ids_t, row_base, table, tam, dpad, dam, dy, dyam, 1.0, d2p, db2p, workspace=ws, parts=1)
    assert float(d2p.abs().max()) == 0.0
    ops.h2_wgrad_emb(ids_t, row_base, table, tam, dpad, dam, dy, dyam, 1.0, d2p, db2p, workspace=ws, parts=2)
    assert torch.equal(d2p, d2) and torch.equal(db2p, db2)
    if Nd:
        with pytest.raises(RuntimeError):    # dense features without their record
            ops.h2_wgrad_emb(ids_t, row_base, table, tam, dpad, None, dy, dyam, 1.0, d2p, db2p, workspace=ws)


def test_narrow_backward_leaves_the_amax_record_of_dx(ops):
    g = torch.Generator(device="cuda").manual_seed(4)
    for (M, K, N) in ((4096, 256, 32), (640, 128, 16), (2048, 512, 24)):
        x = torch.randn((M, K), device="cuda", generator=g)
        dy = torch.randn((M, N), device="cuda", generator=g) * 1e-3
        W = torch.randn((K, N), device="cuda", generator=g) * 0.1
        dW0, db0, dx0 = torch.zeros_like(W), torch.zeros(N, device="cuda"), torch.empty((M, K), device="cuda")
        dW1, db1, dx1 = torch.zeros_like(W), torch.zeros(N, device="cuda"), torch.empty((M, K), device="cuda")
        ops.linear_bwd_narrow(x, dy, W, -0.1, dW0, db0, dx0)
        rec = ops.h2_record("cuda")
        rec.fill_(0x7f000000)                                     # stale garbage from "the previous step": must be reset
        ops.linear_bwd_narrow(x, dy, W, -0.1, dW1, db1, dx1, dx_amax=rec)
        assert torch.equal(dx0, dx1) and torch.equal(dW0, dW1) and torch.equal(db0, db1)
        assert int(rec.item()) == _amax_bits(dx1)


def test_k4_atomic_pieces_also_raise_the_record(ops):
    """K4 without scratch rows (x_sorted = None): rows hotter than 32 slots are combined with fp32 atomics; every piece reports the
    value its own atomic produced, the last one applied is the row's final value -- the record still bounds the table."""
    g = torch.Generator(device="cuda").manual_seed(8)
    B, F, D, V = 4096, 3, 64, 40
    R = F * V
    row_base = (torch.arange(F, device="cuda") * V).to(torch.int64)
    ids = torch.randint(0, V, (B, F), device="cuda", generator=g)
    ids[: B // 2, 1] = 5
    table = torch.randn((R, D), device="cuda", generator=g) * 0.01
    lin = torch.zeros(R, device="cuda")
    bias = torch.zeros(1, device="cuda")
    plan = ops.emb_sort_slots(ids, row_base, R)
    grad = torch.randn((B, F * D), device="cuda", generator=g)
    concat = torch.randn((B, F * D), device="cuda", generator=g)
    sum_x = torch.randn((B, D), device="cuda", generator=g)
    dl = torch.randn(B, device="cuda", generator=g)
    rec = ops.h2_amax(table)
    before = ops.h2_amax_value(rec)
    ops.emb_pool_bwd_sorted(ids, row_base, plan, D, R, grad, dl, -0.05, table, lin, bias, concat=concat, sum_x=sum_x, table_amax=rec)
    torch.cuda.synchronize()
    true = float(table.abs().max().item())
    assert true > 10 * before and ops.h2_amax_value(rec) >= true and ops.h2_amax_value(rec) <= 1.5 * true


@pytest.mark.parametrize("optimizer", ["sgd", "adam"])
def test_k4_keeps_the_table_amax_record_an_upper_bound(ops, optimizer):
    """The engine's running record of the table (raised by K4 from the values it writes: unique rows, shared rows, rows hotter than
    32 slots) is >= the table's true largest magnitude after every step -- also when a step grows the largest value (lr large
    enough that the hottest rows move by more than the initial range)."""
    from deep_recommenders_amd.engine import DeepFMEngine
    F, B, Nd, D, V = 4, 2304, 3, 64, 50
    eng = DeepFMEngine(F, V, D, [256, 16], B, num_dense=Nd, lr=2.0 if optimizer == "sgd" else 0.05, seed=3, lin_init_std=0.1,
                       table_init_std=0.01, optimizer=optimizer)
    assert eng.h2
    g = torch.Generator(device="cuda").manual_seed(2)
    first = ops.h2_amax_value(eng.tab_amax)
    assert first == float(eng.table.abs().max().item())
    for n in range(6):
        keys = torch.randint(0, 10**12, (B, F), device="cuda", generator=g)
        keys[: B // 2, 0] = 7                                      # one row hit > 1000 times
        dense = torch.rand((B, Nd), device="cuda", generator=g) * 3
        labels = (torch.rand(B, device="cuda", generator=g) < 0.3).float()
        loss = float(eng.train_step(keys, dense, labels).item())
        assert loss == loss
        rec, true = ops.h2_amax_value(eng.tab_amax), float(eng.table.abs().max().item())
        assert rec >= true, (n, rec, true)
        assert rec <= 4 * max(true, first)                         # ... and not absurdly loose
    assert float(eng.table.abs().max().item()) > first             # the test did grow the table's range
    # a write from outside the engine is noticed (torch's version counter) and the record rebuilt
    eng.table.mul_(0.001)
    keys = torch.randint(0, 10**12, (B, F), device="cuda", generator=g)
    eng.train_step(keys, torch.rand((B, Nd), device="cuda", generator=g), (torch.rand(B, device="cuda", generator=g) < 0.3).float())
    torch.cuda.synchronize()
    assert ops.h2_amax_value(eng.tab_amax) <= 2 * float(eng.table.abs().max().item())


@pytest.mark.parametrize("optimizer", ["sgd", "adam"])
def test_engine_f16x2_step_tracks_the_bf16x3_step(optimizer, monkeypatch):
    """The default engine (first layer's GEMMs in the f16x2 mode) against DR_GEMM_SPLIT=bf16x3 over prefetched steps: same data,
    same seeds.  Both are fp32-accurate evaluations of the same arithmetic; losses agree to 1e-5 relative (north_star's loss
    tolerance) and parameters to fp32 noise amplified by a few steps."""
    from deep_recommenders_amd.engine import DeepFMEngine
    F, B, Nd, D, V = 6, 4608, 5, 64, 3000
    g = torch.Generator(device="cuda").manual_seed(12)
    batches = [(torch.randint(0, 10**12, (B, F), device="cuda", generator=g), torch.rand((B, Nd), device="cuda", generator=g) * 4,
                (torch.rand(B, device="cuda", generator=g) < 0.3).float()) for _ in range(3)]

    def run():
        eng = DeepFMEngine(F, V, D, [256, 32], B, num_dense=Nd, lr=0.05 if optimizer == "sgd" else 0.002, seed=3, lin_init_std=0.1,
                           optimizer=optimizer)
        losses = []
        for n in range(6):
            k, d, l = batches[n % 3]
            nk, nd = batches[(n + 1) % 3][0], batches[(n + 1) % 3][1]
            losses.append(float(eng.train_step(k, d, l, next_keys=nk, next_dense=nd).item()))
        torch.cuda.synchronize()
        return eng, losses
    h2, l2 = run()
    assert h2.h2
    with _split("bf16x3"):
        b3, l3 = run()
    assert not b3.h2
    for a, b in zip(l2, l3):
        assert abs(a - b) <= 1e-5 * abs(b), (l2, l3)
    tol = 2e-5 if optimizer == "sgd" else 2e-3       # (Adam divides by sqrt(v): a last-bit difference in a tiny gradient moves a whole step)
    assert (h2.table - b3.table).abs().max().item() <= tol * b3.table.abs().max().item()
    assert (h2.flat_params - b3.flat_params).abs().max().item() <= tol * b3.flat_params.abs().max().item()
    # the run is reproducible bit for bit (records, scales and the three-product sums are deterministic)
    again, l2b = run()
    assert l2b == l2 and torch.equal(again.table, h2.table) and torch.equal(again.flat_params, h2.flat_params)


def test_h2_gemm_epilogues_leave_the_amax_record_of_what_they_store(ops):
    """out_amax of dr_h2_linear_nt / dr_h2_cross_fwd and d_prod_amax of dr_cross_combine_bwd_amax: exactly max |stored value| as float
    bits (reset by the call) -- a chain of GEMMs hands each output to the next as an operand without a pass over it."""
    g = torch.Generator(device="cuda").manual_seed(21)
    for (M, K, N) in ((1000, 83, 40), (4100, 300, 520)):
        a = torch.randn((M, (K + 3) // 4 * 4), device="cuda", generator=g)[:, :K]
        W = torch.randn((K, N), device="cuda", generator=g) * 0.1
        b = torch.randn((N,), device="cuda", generator=g)
        mask = torch.randn((M, N), device="cuda", generator=g)
        wp, am = ops.H2WeightPlanes(W), ops.h2_amax(a)
        rec = ops.h2_record("cuda")
        for kw in (dict(bias=b, act=1), dict(bias=b, mask=mask), dict()):
            rec.fill_(0x7f000000)
            y = ops.h2_linear_nt(a, am, wp.wt, out_amax=rec, **kw)
            assert int(rec.item()) == _amax_bits(y), kw
        o = torch.randn((M, N), device="cuda", generator=g)
        ops.h2_linear_nt(a, am, wp.wt, accumulate=True, out=o, out_amax=rec)
        assert int(rec.item()) == _amax_bits(o)
    M, Dm = 1500, 333
    buf0 = torch.randn((M, 336), device="cuda", generator=g)
    buf1 = torch.randn((M, 336), device="cuda", generator=g)
    x0, x = buf0[:, :Dm], buf1[:, :Dm]
    W = torch.randn((Dm, Dm), device="cuda", generator=g) * 0.05
    b = torch.randn((Dm,), device="cuda", generator=g)
    rec = ops.h2_record("cuda")
    out, prod = ops.h2_cross_fwd(x0, x, ops.h2_amax(x), ops.H2WeightPlanes(W).wt, b, 0.3, want_prod=True, out_amax=rec)
    pr = x.double() @ W.double() + b.double() + 0.3 * x.double()
    assert _rel(prod, pr) <= 3e-6 and _rel(out, x0.double() * pr + x.double()) <= 3e-6
    o3, p3 = ops.bf3_cross_fwd(x0, x, ops.WeightPlanes(W).wt, b, 0.3, want_prod=True)
    assert _rel(out, x0.double() * pr + x.double()) <= 1.5 * _rel(o3, x0.double() * pr + x.double()) + 1e-7
    assert int(rec.item()) == _amax_bits(out)
    # the reference's known answer (tests/keras/test_dcn.py:16-23)
    x0k = torch.zeros((5, 4), device="cuda"); xk = torch.zeros((5, 4), device="cuda")
    x0k[:, :3] = torch.tensor([0.1, 0.2, 0.3], device="cuda"); xk[:, :3] = torch.tensor([0.4, 0.5, 0.6], device="cuda")
    ok, _ = ops.h2_cross_fwd(x0k[:, :3], xk[:, :3], ops.h2_amax(xk[:, :3]), ops.H2WeightPlanes(torch.ones((3, 3), device="cuda")).wt, torch.zeros(3, device="cuda"))
    assert torch.allclose(ok, torch.tensor([[0.55, 0.8, 1.05]], device="cuda").repeat(5, 1), rtol=1e-6)
    # the cross layer's backward combine
    d_out = (torch.randn((M, 336), device="cuda", generator=g) * 1e-3)[:, :Dm]
    dx0a, dx0b = torch.zeros((M, 336), device="cuda")[:, :Dm], torch.zeros((M, 336), device="cuda")[:, :Dm]
    dp0 = ops.cross_combine_bwd(x0, prod, d_out, 0.0, dx0a, None)
    rec.fill_(0x7f000000)
    dp1 = ops.cross_combine_bwd(x0, prod, d_out, 0.0, dx0b, None, d_prod_amax=rec)
    assert torch.equal(dp0, dp1) and torch.equal(dx0a, dx0b) and int(rec.item()) == _amax_bits(dp1)


def test_dcn_engine_f16x2_step_tracks_the_bf16x3_step(monkeypatch):
    """DCNEngine (config 4's stack: 3 cross layers + MLP) with every wide GEMM in the f16x2 mode against DR_GEMM_SPLIT=bf16x3: losses
    to 1e-5 relative, parameters to fp32 noise; the records chained through the epilogues equal the tensors' true amax."""
    from deep_recommenders_amd.dcn_engine import DCNEngine
    F, B, Nd, D, V = 6, 4096, 5, 64, 3000
    g = torch.Generator(device="cuda").manual_seed(12)
    batches = [(torch.randint(0, 10**12, (B, F), device="cuda", generator=g), torch.rand((B, Nd), device="cuda", generator=g) * 4,
                (torch.rand(B, device="cuda", generator=g) < 0.3).float()) for _ in range(3)]

    def run():
        eng = DCNEngine(F, V, D, 3, [512, 256, 128], B, num_dense=Nd, lr=0.05, seed=3)
        losses = [float(eng.train_step(*batches[n % 3]).item()) for n in range(5)]
        torch.cuda.synchronize()
        return eng, losses
    h2, l2 = run()
    assert h2.h2 and all(isinstance(p, ops_mod().H2WeightPlanes) for p in h2.cross_planes)
    with _split("bf16x3"):
        b3, l3 = run()
    assert not b3.h2
    for a, b in zip(l2, l3):
        assert abs(a - b) <= 1e-5 * abs(b), (l2, l3)
    assert (h2.table - b3.table).abs().max().item() <= 2e-5 * b3.table.abs().max().item()
    for Wa, Wb in zip(h2.cross_W + h2.Ws, b3.cross_W + b3.Ws):
        assert (Wa - Wb).abs().max().item() <= 2e-5 * Wb.abs().max().item()
    assert int(h2.x_amax[0].item()) == _amax_bits(h2.x0[:, :h2.in_dim])
    for i in range(len(h2.Ws) - 1):
        assert int(h2.h_amax[i].item()) == _amax_bits(h2.hs[i])


def ops_mod():
    from deep_recommenders_amd import ops as _ops
    return _ops


class _split:
    """`with _split("bf16x3"):` -- the library's one operand-split switch (dr_set_gemm_split) set for the block and restored after
    it.  The engines read it at construction, the exact top-K scan at every call; an environment variable patched after the
    library was loaded would be ignored by both."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.prev = ops_mod().set_gemm_split(self.name)
        assert ops_mod().get_gemm_split() == self.name

    def __exit__(self, *exc):
        ops_mod().set_gemm_split(self.prev)


def test_gemm_split_switch_is_the_librarys_and_is_read_where_documented(monkeypatch):
    """dr_set_gemm_split / dr_get_gemm_split: one switch for the engines (read at construction) and the top-K scan (read at every
    call).  The environment variable is parsed once, by the library, when it is loaded: patching it afterwards changes nothing."""
    from deep_recommenders_amd.engine import DeepFMEngine
    o = ops_mod()
    start = o.get_gemm_split()
    monkeypatch.setenv("DR_GEMM_SPLIT", "bf16x3" if start == "f16x2" else "f16x2")
    assert o.get_gemm_split() == start
    with pytest.raises((KeyError, ValueError)):
        o.set_gemm_split("fp8x4")
    mk = lambda: DeepFMEngine(6, 3000, 64, [256, 32], 4608, num_dense=5, lr=0.05, seed=3)
    with _split("f16x2"):
        assert mk().h2
        with _split("bf16x3"):
            assert not mk().h2
        assert o.get_gemm_split() == "f16x2"
        # the exact scan: the two splits give (slightly) different score bits for the same inputs, within fp32 noise of each other
        g = torch.Generator(device="cuda").manual_seed(5)
        q = torch.randn((256, 128), device="cuda", generator=g)
        c = torch.randn((20000, 128), device="cuda", generator=g)
        s_h2, i_h2 = o.topk_mips(q, c, 10)
        with _split("bf16x3"):
            s_b3, i_b3 = o.topk_mips(q, c, 10)
        s_again, i_again = o.topk_mips(q, c, 10)
        assert torch.equal(s_h2, s_again) and torch.equal(i_h2, i_again)
        assert not torch.equal(s_h2, s_b3), "the scan did not change its product mode with the switch"
        assert (s_h2 - s_b3).abs().max().item() <= 1e-5 * s_b3.abs().max().item()
    assert o.get_gemm_split() == start


@pytest.mark.parametrize("K,N,ld", [(1677, 256, 256), (1677, 1677, 1680), (300, 40, 44), (32, 32, 32)])
def test_refresh_weight_equals_amax_then_two_splits(K, N, ld):
    """dr_h2_refresh_weight (H2WeightPlanes.refresh since round 5: per-block maxima with plain stores, then one kernel that reduces
    them, stores the record and writes both orientations) == dr_h2_amax(reset) + dr_h2_split + dr_h2_split(transpose): the same
    record and the same planes, bit for bit -- also after the weight moved (a second refresh must not see anything of the first)."""
    from deep_recommenders_amd import ops
    g = torch.Generator(device="cuda").manual_seed(K + N)
    W = (torch.randn((K, ld), device="cuda", generator=g) * 0.3)[:, :N]
    wp = ops.H2WeightPlanes(W)
    for rep in range(2):
        rec = ops.h2_amax(W)
        wt = ops.h2_split(W, ops.H2Planes(N, K, W.device, rec), transpose=True)
        w = ops.h2_split(W, ops.H2Planes(K, N, W.device, rec))
        assert torch.equal(wp.amax, rec), (ops.h2_amax_value(wp.amax), ops.h2_amax_value(rec))
        assert torch.equal(wp.w.buf, w.buf) and torch.equal(wp.wt.buf, wt.buf)
        W.mul_(0.01 if rep == 0 else 1.0)                 # the largest magnitude SHRINKS: a stale record would show
        W[K // 2, N // 3] = 3.0
        wp.ensure_fresh()
