"""exact top-100 of 8192 queries over a 1 M x 128 corpus (BASELINE config 5's metric pass), timed; run under rocprofv3 for the split"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deep_recommenders_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
q = torch.randn((8192, 128), device="cuda", generator=g)
c = torch.randn((1_000_000, 128), device="cuda", generator=g)
ws = torch.empty(max(1, ops.lib().dr_topk_workspace_bytes(8192, 1_000_000, 100) // 4), dtype=torch.float32, device="cuda")
for _ in range(2):
    s, i = ops.topk_mips(q, c, 100, workspace=ws)
torch.cuda.synchronize()
t = time.perf_counter()
n = 3
for _ in range(n):
    s, i = ops.topk_mips(q, c, 100, workspace=ws)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / n
print("topk_mips 8192 x 1M x 128, k=100: %.2f ms  (%.1f TFLOP/s)" % (dt * 1e3, 2 * 8192 * 1e6 * 128 / dt / 1e12))
