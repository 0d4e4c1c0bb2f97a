"""A/B of the register-split GEMM's two wave shapes: 8 waves x 32 rows (default) vs 4 waves x 64 rows (DR_BF3_RS64=1, one wave per
SIMD).  Run once per setting (the switch is read once per process):
    python tools/exp/rs64_bench.py            ; DR_BF3_RS64=1 python tools/exp/rs64_bench.py
Prints the time of the DeepFM config-3 shapes (forward layer M x 1677 -> 256, dgrad M x 256 -> 1677) and of a DCN cross-sized
square layer, and a checksum of each output (the two settings must print the same checksums: same splits, same k order)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from deep_recommenders_amd import ops

torch.manual_seed(0)
dev = "cuda"
M = 65536


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def case(name, K, N, mask=False, accumulate=False, act=0):
    x = torch.randn(M, K, device=dev)
    if K % 4:
        buf = torch.zeros(M, (K + 3) // 4 * 4, device=dev)
        buf[:, :K] = x
        x = buf[:, :K]
    w = torch.randn(K, N, device=dev) / K ** 0.5
    wp = ops.WeightPlanes(w)
    b = torch.randn(N, device=dev)
    m = torch.randn(M, N, device=dev) if mask else None
    out = torch.zeros(M, (N + 3) // 4 * 4, device=dev)[:, :N]
    fn = lambda: ops.bf3_linear_nt(x, wp.wt, bias=b, act=act, mask=m, accumulate=accumulate, out=out)
    out.zero_()
    fn()
    chk = out.double().sum().item(), out.double().abs().sum().item()
    ref = torch.relu(x[:4096].double() @ w.double() + b.double()) if act else x[:4096].double() @ w.double() + b.double()
    if mask:
        ref = torch.where(m[:4096] > 0, ref, torch.zeros_like(ref))
    err = (out[:4096].double() - ref).abs().max().item() / ref.abs().max().item()
    us = timeit(fn)
    tf = 2.0 * M * K * N / us / 1e6
    print("%-28s K=%5d N=%5d  %8.1f us  %6.1f TFLOP/s (fp32-equivalent)  rel.err %.2e  checksum %.9e %.9e" % (name, K, N, us, tf, err, *chk))


print("DR_BF3_RS64 =", os.environ.get("DR_BF3_RS64", "0"))
case("forward layer 0 (relu)", 1677, 256, act=1)
case("dgrad layer 0", 256, 1677)
case("dgrad with mask", 256, 1677, mask=True)
case("square 1677", 1677, 1677)
case("accumulate", 256, 1677, accumulate=True)
