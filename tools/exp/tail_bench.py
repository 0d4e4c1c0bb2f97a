"""EXPERIMENT driver: the one-pass tower tail alone at config 3's shape (M = 65 536, K = 256, H = 32) against the two launches it replaces;
main kernels only (parts = 1), HIP events over 30 launches."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from deep_recommenders_amd import ops
M, K, H = 65536, 256, 32
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn((M, K), device="cuda", generator=g).clamp_min(0)
W1 = torch.randn((K, H), device="cuda", generator=g) / K ** 0.5
b1 = torch.zeros(H, device="cuda")
W2 = torch.zeros((H, 4), device="cuda")[:, :1]; W2.copy_(torch.randn((H, 1), device="cuda", generator=g))
b2 = torch.zeros(1, device="cuda")
extra = torch.randn(M, device="cuda", generator=g)
z = (torch.rand(M, device="cuda", generator=g) < 0.3).float()
dx = torch.empty((M, K), device="cuda"); d_h = torch.empty((M, H), device="cuda")
prob = torch.empty(M, device="cuda"); d_logit = torch.empty(M, device="cuda"); loss = torch.empty(1, device="cuda")
ws = ops.tower_tail_workspace(M, K, "cuda"); hws = ops.tower_head_workspace(M, "cuda"); nws = ops.linear_bwd_narrow_workspace(M, K, H, "cuda")
gW1, gb1, gW2, gb2 = torch.zeros_like(W1), torch.zeros_like(b1), torch.zeros((H, 4), device="cuda")[:, :1], torch.zeros_like(b2)


def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


fused = lambda: ops.tower_tail_fused(x, W1, b1, W2, b2, extra, z, 0, 1.0, dx, dst_W1=gW1, dst_b1=gb1, dst_W2=gW2, dst_b2=gb2, prob=prob, d_logit=d_logit,
                                     d_h=d_h, loss=loss, workspace=ws, parts=1)
head = lambda: ops.tower_head_fwd_bwd(x, W1, b1, W2, b2, extra, z, 0, 1.0, prob=prob, d_logit=d_logit, d_h=d_h, loss=loss, workspace=hws, dst_W2=gW2,
                                      dst_b2=gb2, parts=1)
narrow = lambda: ops.linear_bwd_narrow(x, d_h, W1, 1.0, gW1, gb1, dx, relu_mask=True, workspace=nws, parts=1)
print("TAIL fused %.1f us   head %.1f us   narrow %.1f us   (hbm-ideal: %.1f us at 5.5 TB/s)" % (
    timeit(fused), timeit(head), timeit(narrow), (2 * M * K * 4 + M * H * 4) / 5.5e12 * 1e6), flush=True)
