"""EXPERIMENT (historical: the dr_debug_set_gemm_variant hook it drove was removed once the result was in DESIGN.md): in-process A/B of GEMM variants (dr_debug_set_gemm_variant) on the L0 shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deep_recommenders_amd import ops, _lib
dev = "cuda"
L = _lib.lib()
g = torch.Generator(device=dev); g.manual_seed(0)
M, K, N = 65536, 1677, 256
x = torch.randn((M, 1680), device=dev, generator=g)[:, :K]
W = torch.randn((K, N), device=dev, generator=g) / K ** 0.5
b = torch.zeros(N, device=dev)
y = torch.empty((M, N), device=dev)
dy = torch.randn((M, N), device=dev, generator=g)
dx = torch.empty((M, 1680), device=dev)[:, :K]
dW = torch.zeros((K, N), device=dev)
ws = ops.linear_bwd_dw_workspace(M, K, N, dev)
fl = 2.0 * M * K * N


def timeit(fn, iters=6, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


variants = [int(v) for v in os.environ.get("VARIANTS", "2,6").split(",")]
ref = {}
for rnd in range(3):
    for v in variants:
        L.dr_debug_set_gemm_variant(v)
        tf = timeit(lambda: ops.linear_fwd(x, W, b, 1, out=y))
        td = timeit(lambda: ops.linear_bwd_dx(dy, W, None, out=dx))
        tw = timeit(lambda: ops.linear_bwd_dw(x, dy, 1e-6, dW, b, workspace=ws))
        chk = (y.double().sum().item(), dx.double().sum().item())
        ref.setdefault("chk", chk)
        print("ABG2 variant=%d round %d  fwd %.1f  dx %.1f  dw %.1f TF   (%.0f %.0f %.0f us) same=%s" % (
            v, rnd, fl / tf / 1e12, fl / td / 1e12, fl / tw / 1e12, tf * 1e6, td * 1e6, tw * 1e6, chk == ref["chk"]))
