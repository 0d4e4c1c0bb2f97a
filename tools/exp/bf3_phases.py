"""EXPERIMENT: phase timeline (shader clock) of one block of the bf16x3 forward GEMM, via a temporary debug hook."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deep_recommenders_amd import ops, _lib
g = torch.Generator(device="cuda").manual_seed(0)
M, K, N = 65536, 1677, 256
x = torch.randn((M, 1680), device="cuda", generator=g)[:, :K]
W = torch.randn((K, N), device="cuda", generator=g) / K ** 0.5
b = torch.randn(N, device="cuda", generator=g)
y = torch.empty((M, N), device="cuda")
for _ in range(3):
    ops.linear_fwd(x, W, b, 0, out=y)
torch.cuda.synchronize()
L = ctypes.CDLL(_lib.SO_PATH if hasattr(_lib, "SO_PATH") else os.path.join(os.path.dirname(_lib.__file__), "lib", "libdr_hotpath.so"))
buf = (ctypes.c_ulonglong * 512)()
print("rc", L.dr_debug_clk(buf))
v = list(buf)
for w in range(4):
    for it in range(8):
        r = v[(w * 8 + it) * 8:(w * 8 + it) * 8 + 6]
        if it == 0: base = r[0]
        print("PH wave %d it %d: start %6d | store %5d | barrier1 %5d | issue loads %4d | frag+mfma %5d | barrier2 %5d | total %5d" % (
            w, it, r[0] - base, r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4], r[5] - r[0]))
