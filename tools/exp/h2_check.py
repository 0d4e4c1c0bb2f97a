"""EXPERIMENT driver (not product): the f16x2 operand mode (dr_h2_*) against bf16x3 and fp64 -- accuracy, then timing at config 3's
first-layer shape.  `python tools/exp/h2_check.py [acc] [time]`"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deep_recommenders_amd import ops
dev = "cuda"


def rel(a, b):
    b = b.double()
    return float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30))


def relrow(a, b):       # worst row-wise error relative to the row's own largest entry
    b = b.double()
    return float(((a.double() - b).abs().max(1).values / b.abs().max(1).values.clamp_min(1e-300)).max())


def rms(a, b):
    b = b.double()
    return float(((a.double() - b).pow(2).mean() / b.pow(2).mean()).sqrt())


def acc():
    g = torch.Generator(device=dev).manual_seed(5)
    for (M, K, N) in ((3000, 1677, 256), (4096, 256, 1677), (777, 96, 130)):
        for case in ("randn", "rows 1e-6..1", "tiny 1e-20", "huge 1e20", "one big element"):
            a = torch.randn((M, (K + 3) // 4 * 4), device=dev, generator=g)[:, :K] * 0.1
            W = torch.randn((K, N), device=dev, generator=g) * 0.05
            if case == "rows 1e-6..1":
                a = a * torch.pow(10.0, -6 * torch.rand((M, 1), device=dev, generator=g))
            elif case == "tiny 1e-20":
                a = a * 1e-20; W = W * 1e-10
            elif case == "huge 1e20":
                a = a * 1e15; W = W * 1e12
            elif case == "one big element":
                a = a.clone(); a[0, 0] = 3000.0
            ref = a.double() @ W.double()
            wp3, wp2 = ops.WeightPlanes(W), ops.H2WeightPlanes(W)
            am = ops.h2_amax(a)
            y3 = ops.bf3_linear_nt(a, wp3.wt)
            y2 = ops.h2_linear_nt(a, am, wp2.wt)
            y1 = a @ W
            print("NT  M=%d K=%d N=%d %-16s amax %.3e | max-rel: h2 %.2e bf3 %.2e torch-f32 %.2e | row-rel: h2 %.2e bf3 %.2e f32 %.2e | rms: h2 %.2e bf3 %.2e f32 %.2e"
                  % (M, K, N, case, ops.h2_amax_value(am), rel(y2, ref), rel(y3, ref), rel(y1, ref), relrow(y2, ref), relrow(y3, ref), relrow(y1, ref),
                     rms(y2, ref), rms(y3, ref), rms(y1, ref)), flush=True)
        # epilogues: bias + relu, mask, accumulate
        a = torch.randn((M, (K + 3) // 4 * 4), device=dev, generator=g)[:, :K] * 0.1
        W = torch.randn((K, N), device=dev, generator=g) * 0.05
        b = torch.randn((N,), device=dev, generator=g)
        mask = torch.randn((M, N), device=dev, generator=g)
        wp2 = ops.H2WeightPlanes(W)
        am = ops.h2_amax(a)
        ref = a.double() @ W.double() + b.double()
        print("  relu+bias %.2e  mask %.2e" % (rel(ops.h2_linear_nt(a, am, wp2.wt, bias=b, act=1), ref.clamp_min(0)),
                                                rel(ops.h2_linear_nt(a, am, wp2.wt, bias=b, mask=mask), ref * (mask > 0))), end="")
        o = torch.ones((M, N), device=dev)
        ops.h2_linear_nt(a, am, wp2.wt, bias=b, accumulate=True, out=o)
        print("  accumulate %.2e" % rel(o, ref + 1))
        # wgrad
        x = a
        dy = torch.randn((M, N), device=dev, generator=g) * 1e-3 * torch.pow(10.0, -4 * torch.rand((M, 1), device=dev, generator=g))
        refw = x.double().t() @ dy.double()
        d3, d2 = torch.zeros((K, N), device=dev), torch.zeros((K, N), device=dev)
        b3, b2 = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
        ops.bf3_wgrad(x, dy, -0.5, d3, b3)
        ops.h2_wgrad(x, ops.h2_amax(x), dy, ops.h2_amax(dy), -0.5, d2, b2)
        print("  wgrad: h2 %.2e bf3 %.2e torch-f32 %.2e | rms h2 %.2e bf3 %.2e | db h2 %.2e" % (
            rel(d2, -0.5 * refw), rel(d3, -0.5 * refw), rel(x.t() @ dy, refw), rms(d2, -0.5 * refw), rms(d3, -0.5 * refw),
            rel(b2, -0.5 * dy.double().sum(0))), flush=True)
    # fused forward + gathered wgrad on small tables
    for (M, F, Nd, N) in ((3000, 26, 13, 256), (700, 3, 0, 64), (70000, 5, 7, 300)):
        D, V = 64, 997
        table = torch.randn((F * V, D), device=dev, generator=g) * 0.3
        lin_w = torch.randn((F * V,), device=dev, generator=g)
        lin_b = torch.tensor([0.37], device=dev)
        row_base = (torch.arange(F, device=dev) * V).to(torch.int64)
        ids = torch.randint(0, V, (M, F), device=dev, generator=g)
        ids[torch.rand((M, F), device=dev, generator=g) < 0.05] = -1
        K = F * D + Nd
        ld = (K + 3) // 4 * 4
        dense = torch.randn((M, Nd), device=dev, generator=g) * 5
        W = torch.randn((K, N), device=dev, generator=g) * 0.1
        b = torch.randn((N,), device=dev, generator=g)
        wp3, wp2 = ops.WeightPlanes(W), ops.H2WeightPlanes(W)
        dpad = None
        if Nd:
            dpad = torch.zeros((M, 32), device=dev); dpad[:, :Nd] = dense
        outs = []
        for mode in (3, 2):
            concat = torch.zeros((M, ld), device=dev); concat[:, :F * D] = float("nan"); concat[:, F * D:K] = dense
            sx, fm = torch.full((M, D), float("nan"), device=dev), torch.full((M,), float("nan"), device=dev)
            y = torch.full((M, N), float("nan"), device=dev)
            lv = torch.zeros((F, M), device=dev)
            if mode == 3:
                ops.bf3_emb_linear_fwd(ids, row_base, V, table, lin_w, lin_b, dpad, concat, K, wp3.wt, b, 1, sx, fm, y, lin_vals_t=lv)
            else:
                tam = ops.h2_amax(table)
                dam = ops.h2_amax(dpad) if Nd else None
                ops.h2_emb_linear_fwd(ids, row_base, V, table, tam, lin_w, lin_b, dpad, dam, concat, K, wp2.wt, b, 1, sx, fm, y, lin_vals_t=lv)
            outs.append((concat, sx, fm, y, lv))
        c3, c2 = outs
        ref = (c3[0][:, :K].double() @ W.double() + b.double()).clamp_min(0)
        print("FUSED M=%d F=%d Nd=%d N=%d: concat equal %s sum_x equal %s fm equal %s lin_vals equal %s | y: h2 %.2e bf3 %.2e" % (
            M, F, Nd, N, torch.equal(c3[0], c2[0]), torch.equal(c3[1], c2[1]), torch.equal(c3[2], c2[2]), torch.equal(c3[4], c2[4]),
            rel(c2[3], ref), rel(c3[3], ref)), flush=True)
        dy = torch.randn((M, N), device=dev, generator=g) * 1e-3
        ids_t = ids.t().contiguous().to(torch.int32)
        refw = c3[0][:, :K].double().t() @ dy.double()
        d3, d2 = torch.zeros((K, N), device=dev), torch.zeros((K, N), device=dev)
        ops.bf3_wgrad_emb(ids_t, row_base, table, dpad, dy, 1.0, d3)
        ops.h2_wgrad_emb(ids_t, row_base, table, ops.h2_amax(table), dpad, ops.h2_amax(dpad) if Nd else None, dy, ops.h2_amax(dy), 1.0, d2)
        print("  wgrad_emb: h2 %.2e bf3 %.2e" % (rel(d2, refw), rel(d3, refw)), flush=True)


def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); evs.append((s, e))
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs)
    return ts[len(ts) // 2] * 1e3


def tim():
    g = torch.Generator(device=dev).manual_seed(1)
    M, F, Nd, N, D, V = 65536, 26, 13, 256, 64, 1_000_000
    K = F * D + Nd
    ld = (K + 3) // 4 * 4
    table = torch.empty((F * V, D), device=dev).normal_(0, 0.1, generator=g)
    lin_w = torch.zeros((F * V,), device=dev)
    lin_b = torch.zeros(1, device=dev)
    row_base = (torch.arange(F, device=dev) * V).to(torch.int64)
    ids = torch.randint(0, V, (M, F), device=dev, generator=g)
    ids_t = ids.t().contiguous().to(torch.int32)
    dpad = torch.zeros((M, 32), device=dev); dpad[:, :Nd] = torch.randn((M, Nd), device=dev, generator=g)
    W = torch.randn((K, N), device=dev, generator=g) * 0.05
    b = torch.zeros(N, device=dev)
    wp3, wp2 = ops.WeightPlanes(W), ops.H2WeightPlanes(W)
    concat = torch.zeros((M, ld), device=dev)
    sx, fm, y = torch.empty((M, D), device=dev), torch.empty((M,), device=dev), torch.empty((M, N), device=dev)
    lv = torch.zeros((F, M), device=dev)
    tam, dam = ops.h2_amax(table), ops.h2_amax(dpad)
    dy = torch.randn((M, N), device=dev, generator=g) * 1e-3
    mask = torch.randn((M, N), device=dev, generator=g)
    dyam = ops.h2_record(dev)
    dx = torch.empty((M, ld), device=dev)[:, :K]
    dW = torch.zeros((K, N), device=dev)
    db = torch.zeros(N, device=dev)
    ws = ops.bf3_wgrad_workspace(M, K, N, dev)
    for rep in range(2):
        print("---- pass", rep)
        print("fused fwd (no concat)  bf3 %.1f us" % timeit(lambda: ops.bf3_emb_linear_fwd(ids, row_base, V, table, lin_w, lin_b, dpad, None, K, wp3.wt, b, 1, sx, fm, y, lin_vals_t=lv)))
        print("fused fwd (no concat)  h2  %.1f us" % timeit(lambda: ops.h2_emb_linear_fwd(ids, row_base, V, table, tam, lin_w, lin_b, dpad, dam, None, K, wp2.wt, b, 1, sx, fm, y, lin_vals_t=lv)))
        print("plain fwd              bf3 %.1f us" % timeit(lambda: ops.bf3_linear_nt(concat[:, :K], wp3.wt, bias=b, act=1, out=y)))
        cam = ops.h2_amax(concat[:, :K])
        print("plain fwd              h2  %.1f us" % timeit(lambda: ops.h2_linear_nt(concat[:, :K], cam, wp2.wt, bias=b, act=1, out=y)))
        print("dgrad                  bf3 %.1f us" % timeit(lambda: ops.bf3_linear_nt(dy, wp3.w, out=dx)))
        print("amax(dy)                   %.1f us" % timeit(lambda: ops.h2_amax(dy, dyam)))
        print("dgrad                  h2  %.1f us" % timeit(lambda: ops.h2_linear_nt(dy, dyam, wp2.w, out=dx)))
        print("wgrad_emb              bf3 %.1f us" % timeit(lambda: ops.bf3_wgrad_emb(ids_t, row_base, table, dpad, dy, -1e-3, dW, db, workspace=ws)))
        print("wgrad_emb              h2  %.1f us" % timeit(lambda: ops.h2_wgrad_emb(ids_t, row_base, table, tam, dpad, dam, dy, dyam, -1e-3, dW, db, workspace=ws)))
        print("weight refresh         bf3 %.1f us   h2 %.1f us" % (timeit(wp3.refresh), timeit(wp2.refresh)))


if __name__ == "__main__":
    what = sys.argv[1:] or ["acc", "time"]
    if "acc" in what: acc()
    if "time" in what: tim()
