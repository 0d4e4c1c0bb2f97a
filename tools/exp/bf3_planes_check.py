"""GPU check + timing of the planes GEMMs (dr_bf3_gemm_nt / _tn) against fp64 and against the in-kernel-split bf16x3 GEMMs."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deep_recommenders_amd import ops


def timeit(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def relerr(got, ref):
    ref = ref.double()
    return ((got.double() - ref).abs().max() / ref.abs().max()).item()


def run(M, K, N, time=True):
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    x = torch.zeros((M, (K + 3) // 4 * 4), device="cuda")[:, :K]
    x.copy_(torch.relu(torch.randn((M, K), device="cuda", generator=g)))    # ReLU-sparse-ish like activations; 16-byte aligned rows
    W = torch.randn((K, N), device="cuda", generator=g) * 0.05
    b = torch.randn((N,), device="cuda", generator=g)
    dy = torch.randn((M, N), device="cuda", generator=g) * 1e-3
    xp = ops.bf3_split(x, ops.Planes(M, K, "cuda"))
    wtp = ops.bf3_split(W, ops.Planes(N, K, "cuda"), transpose=True)
    wp = ops.bf3_split(W, ops.Planes(K, N, "cuda"))
    dyp = ops.bf3_split(dy, ops.Planes(M, N, "cuda"))
    assert torch.equal(ops.bf3_join(xp), x) and torch.equal(ops.bf3_join(wtp), W.t().contiguous()), "split/join not exact"
    # forward
    y = ops.bf3_gemm_nt(xp, wtp, bias=b, act=1)
    ref = torch.relu(x.double() @ W.double() + b.double())
    y_old = ops.linear_fwd(x, W, b, 1)
    print("M=%d K=%d N=%d" % (M, K, N))
    y_rs = ops.bf3_linear_nt(x, wtp, bias=b, act=1)
    print("  fwd   err planes %.3e  in-kernel %.3e  reg-split %.3e" % (relerr(y, ref), relerr(y_old, ref), relerr(y_rs, ref)))
    # dgrad
    dx = ops.bf3_gemm_nt(dyp, wp)
    ref = dy.double() @ W.double().t()
    dx_old = ops.linear_bwd_dx(dy, W)
    dx_rs = ops.bf3_linear_nt(dy, wp)
    print("  dgrad err planes %.3e  in-kernel %.3e  reg-split %.3e" % (relerr(dx, ref), relerr(dx_old, ref), relerr(dx_rs, ref)))
    # wgrad
    dW = torch.zeros((K, N), device="cuda")
    ws = ops.bf3_gemm_tn_workspace(M, K, N, "cuda")
    ws.fill_(float("nan"))
    ops.bf3_gemm_tn(xp, dyp, 1.0, dW, workspace=ws)
    ref = x.double().t() @ dy.double()
    dW_old = torch.zeros((K, N), device="cuda")
    ws_old = ops.linear_bwd_dw_workspace(M, K, N, "cuda")
    ops.linear_bwd_dw(x, dy, 1.0, dW_old, None, workspace=ws_old)
    dW_rs = torch.zeros((K, N), device="cuda")
    db_rs = torch.zeros((N,), device="cuda")
    ws_rs = ops.bf3_wgrad_workspace(M, K, N, "cuda")
    ws_rs.fill_(float("nan"))
    ops.bf3_wgrad(x, dy, 1.0, dW_rs, db_rs, workspace=ws_rs)
    print("  wgrad err planes %.3e  in-kernel %.3e  reg-split %.3e  (db %.3e)" % (relerr(dW, ref), relerr(dW_old, ref), relerr(dW_rs, ref),
                                                                               relerr(db_rs, dy.double().sum(0))))
    if time:
        fl = 2.0 * M * K * N
        for name, fn in [("fwd planes", lambda: ops.bf3_gemm_nt(xp, wtp, bias=b, act=1, out=y)),
                         ("fwd in-kernel", lambda: ops.linear_fwd(x, W, b, 1, out=y_old)),
                         ("fwd reg-split", lambda: ops.bf3_linear_nt(x, wtp, bias=b, act=1, out=y_rs)),
                         ("dgrad reg-split", lambda: ops.bf3_linear_nt(dy, wp, out=dx_rs)),
                         ("dgrad planes", lambda: ops.bf3_gemm_nt(dyp, wp, out=dx)),
                         ("dgrad in-kernel", lambda: ops.linear_bwd_dx(dy, W, out=dx_old)),
                         ("wgrad planes", lambda: ops.bf3_gemm_tn(xp, dyp, 1e-9, dW, workspace=ws)),
                         ("wgrad in-kernel", lambda: ops.linear_bwd_dw(x, dy, 1e-9, dW_old, None, workspace=ws_old)),
                         ("wgrad reg-split", lambda: ops.bf3_wgrad(x, dy, 1e-9, dW_rs, db_rs, workspace=ws_rs)),
                         ("split x", lambda: ops.bf3_split(x, xp)),
                         ("split W^T", lambda: ops.bf3_split(W, wtp, transpose=True)),
                         ("split W", lambda: ops.bf3_split(W, wp))]:
            us = timeit(fn)
            print("  %-16s %8.1f us  %6.1f TF/s" % (name, us, fl / us / 1e6))


if __name__ == "__main__":
    run(200, 83, 40, time=False)
    run(1000, 300, 257, time=False)
    run(4096, 128, 520, time=False)
    run(300, 64, 600, time=False)
    run(65536, 1677, 256)
    run(65536, 1677, 1024)
    run(65536, 1677, 1677)
