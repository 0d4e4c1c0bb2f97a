"""EXPERIMENT: shader clock idle vs under the L0 GEMMs / K3 (tools/exp/clock_probe.hip)."""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deep_recommenders_amd import ops
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libclock_probe.so")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", os.path.join(here, "clock_probe.hip"), "-o", so])
L = ctypes.CDLL(so)
dev = "cuda"
out = torch.zeros(2, dtype=torch.int64, device=dev)
side = torch.cuda.Stream()
g = torch.Generator(device=dev); g.manual_seed(0)
M, K, N = 65536, 1677, 256
x = torch.randn((M, 1680), device=dev, generator=g)[:, :K]
W = torch.randn((K, N), device=dev, generator=g) / K ** 0.5
b = torch.zeros(N, device=dev); y = torch.empty((M, N), device=dev)


def probe(load_fn, dur_us=3000):
    torch.cuda.synchronize()
    if load_fn is not None:
        for _ in range(3): load_fn()              # warm the clocks under this load
        for _ in range(12): load_fn()             # ~6 ms of queued work
    with torch.cuda.stream(side):
        L.launch_clock_probe(ctypes.c_void_p(out.data_ptr()), ctypes.c_longlong(dur_us * 100), ctypes.c_void_p(side.cuda_stream))
    torch.cuda.synchronize()
    cyc, wall = out.tolist()
    return cyc / wall * 100e6 / 1e9


print("CLOCK idle            %.3f GHz (shader cycles per 100 MHz tick)" % probe(None))
print("CLOCK under fwd GEMM  %.3f GHz" % probe(lambda: ops.linear_fwd(x, W, b, 1, out=y)))
a = torch.empty(1 << 28, device=dev); c = torch.empty(1 << 28, device=dev)
print("CLOCK under HBM copy  %.3f GHz" % probe(lambda: c.copy_(a)))
print("CLOCK idle again      %.3f GHz" % probe(None))
