"""EXPERIMENT driver: the exact top-100 scan of config 5 alone (8192 queries x 1 M items x 128), for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from deep_recommenders_amd import ops
g = torch.Generator(device="cuda"); g.manual_seed(0)
Bq, N, D, k = 8192, 1_000_000, 128, 100
q = torch.randn((Bq, D), device="cuda", generator=g) / D ** 0.5
c = torch.randn((N, D), device="cuda", generator=g) / D ** 0.5
if os.environ.get("TOPK_INDEX", "0") == "1":          # the corpus pre-split once (ops.TopKIndex), as BruteForce.index does
    c = ops.TopKIndex(c)
for _ in range(2):
    s, i = ops.topk_mips(q, c, k)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    s, i = ops.topk_mips(q, c, k)
e1.record(); torch.cuda.synchronize()
# a fingerprint of the result (A/B runs of two libraries must print the same one) + the IVF scan, which shares the list code
fp = (int(i.sum().item()) * 1000003 + int(s.double().sum().item() * 1e6)) & 0xFFFFFFFFFFFF
print("topk %.2f ms per call   result fingerprint %012x" % (e0.elapsed_time(e1) / 5, fp))
