"""Times the a1 hash kernel at the headline shape (65 536 x 26 int64 keys)."""
import torch
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from deep_recommenders_amd import ops
B, F, V = 65536, 26, 10_000_000
g = torch.Generator(device="cuda").manual_seed(0)
keys = torch.randint(0, 2**53, (B, F), device="cuda", generator=g)
bk = torch.full((F,), V, dtype=torch.int64, device="cuda")
for _ in range(5):
    ops.hash_bucket_i64(keys, bk)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(200):
    ops.hash_bucket_i64(keys, bk)
e1.record(); torch.cuda.synchronize()
print("hash_bucket_i64 %.2f us" % (e0.elapsed_time(e1) * 1000 / 200))
