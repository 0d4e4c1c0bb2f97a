"""EXPERIMENT: fp32 GEMMs with products emulated on the bf16 matrix pipe (bf16x3, 6 products) vs the native fp32 MFMA path.
Run with GEMM_MODE=native | bf16x3.  Prints time and error against fp64 on a row sample."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deep_recommenders_amd import ops
dev = "cuda"
g = torch.Generator(device=dev); g.manual_seed(0)
M, K, N = 65536, 1677, 256
x = torch.randn((M, 1680), device=dev, generator=g)[:, :K]
W = torch.randn((K, N), device=dev, generator=g) / K ** 0.5
b = torch.randn(N, device=dev, generator=g)
y = torch.empty((M, N), device=dev)
dy = torch.randn((M, N), device=dev, generator=g)
dx = torch.empty((M, 1680), device=dev)[:, :K]
dW = torch.zeros((K, N), device=dev)
db = torch.zeros(N, device=dev)
ws = ops.linear_bwd_dw_workspace(M, K, N, dev)
fl = 2.0 * M * K * N


def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


mode = os.environ.get("GEMM_MODE", "bf16x3")
ops.set_gemm_mode(mode)
tf = timeit(lambda: ops.linear_fwd(x, W, b, 0, out=y))
td = timeit(lambda: ops.linear_bwd_dx(dy, W, None, out=dx))
def dw():
    dW.zero_(); db.zero_()
    ops.linear_bwd_dw(x, dy, 1.0, dW, db, workspace=ws)
tw = timeit(dw)
S = 2048
e_f = ((y[:S].double() - (x[:S].double() @ W.double() + b.double())).abs().max() / y[:S].abs().max()).item()
e_d = ((dx[:S].double() - dy[:S].double() @ W.double().t()).abs().max() / dx[:S].abs().max()).item()
ref_w = x.double().t()[:128] @ dy.double()
e_w = ((dW[:128].double() - ref_w).abs().max() / ref_w.abs().max()).item()
r_f = ((y[:S].double() - (x[:S].double() @ W.double() + b.double())).pow(2).mean().sqrt() / y[:S].double().pow(2).mean().sqrt()).item()
print("BF3 mode=%s fwd %.1f TF (%.0f us) dx %.1f TF (%.0f us) dw %.1f TF (%.0f us)  max-rel err fwd %.2e dx %.2e dw %.2e  rms-rel fwd %.2e" % (
    mode, fl / tf / 1e12, tf * 1e6, fl / td / 1e12, td * 1e6, fl / tw / 1e12, tw * 1e6, e_f, e_d, e_w, r_f))
