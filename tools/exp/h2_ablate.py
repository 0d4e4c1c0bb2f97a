"""EXPERIMENT driver (not product): the f16x2 register-split GEMM's ablations / wave shapes (library built with -DDR_BF3_ABLATE:
`bash tools/exp/rs_ablate.sh build`).  One setting per process: DR_BF3_RS_DBG (0, 32 no fragment reads, 64 no split, 96 both, 2 no
MFMA, 1 no weight DMA), DR_BF3_RS64 (0 / 1)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from deep_recommenders_amd import ops
dev, M = "cuda", 65536
torch.manual_seed(0)


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def case(name, K, N):
    buf = torch.zeros(M, (K + 3) // 4 * 4, device=dev)
    buf[:, :K] = torch.randn(M, K, device=dev)
    x = buf[:, :K]
    w = torch.randn(K, N, device=dev) / K ** 0.5
    wp = ops.H2WeightPlanes(w)
    am = ops.h2_amax(x)
    out = torch.zeros(M, (N + 3) // 4 * 4, device=dev)[:, :N]
    fn = lambda: ops.h2_linear_nt(x, am, wp.wt, out=out)
    fn()
    ref = x[:2048].double() @ w.double()
    err = (out[:2048].double() - ref).abs().max().item() / ref.abs().max().item()
    print("H2ABL rs64=%s dbg=%-3s %-18s %8.1f us   rel.err %.1e" % (os.environ.get("DR_BF3_RS64", "0"), os.environ.get("DR_BF3_RS_DBG", "0"), name, timeit(fn), err), flush=True)


case("forward K=1677 N=256", 1677, 256)
case("dgrad K=256 N=1677", 256, 1677)
case("square 1677", 1677, 1677)
