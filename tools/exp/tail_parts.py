"""EXPERIMENT: standalone time of every small kernel in the DeepFM tower tail (layers [256->32], [32->1], loss)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deep_recommenders_amd import ops
dev = "cuda"
g = torch.Generator(device=dev); g.manual_seed(0)
M = 65536
h0 = torch.randn((M, 256), device=dev, generator=g).relu_()
W1 = torch.randn((256, 32), device=dev, generator=g) / 16
b1 = torch.zeros(32, device=dev)
h1 = torch.empty((M, 32), device=dev)
W2 = (torch.randn((32, 4), device=dev, generator=g) / 6)[:, :1]
b2 = torch.zeros(1, device=dev)
h2 = torch.empty((M, 4), device=dev)[:, :1]
fm = torch.randn(M, device=dev, generator=g)
labels = (torch.rand(M, device=dev, generator=g) > 0.5).float()
prob = torch.empty(M, device=dev); dlog = torch.empty(M, device=dev); loss = torch.zeros(1, device=dev); ws = torch.empty(1024, device=dev)
dh1 = torch.empty((M, 32), device=dev)
dh0 = torch.empty((M, 256), device=dev)
ws1 = ops.linear_bwd_dw_workspace(M, 256, 32, dev)
ws2 = ops.linear_bwd_dw_workspace(M, 32, 1, dev)


def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


parts = [
    ("fwd_L1", lambda: ops.linear_fwd(h0, W1, b1, 1, out=h1)),
    ("fwd_L2", lambda: ops.linear_fwd(h1, W2, b2, 0, out=h2)),
    ("bce", lambda: ops.bce_fwd_bwd(fm, labels, 0, workspace=ws, logits_b=h2, out=(prob, dlog, loss))),
    ("dx_L2", lambda: ops.linear_bwd_dx(dlog.reshape(-1, 1), W2, relu_src=h1, out=dh1)),
    ("dw_L2", lambda: ops.linear_bwd_dw(h1, dlog.reshape(-1, 1), -0.01, W2, b2, workspace=ws2)),
    ("dx_L1", lambda: ops.linear_bwd_dx(dh1, W1, relu_src=h0, out=dh0)),
    ("dw_L1", lambda: ops.linear_bwd_dw(h0, dh1, -0.01, W1, b1, workspace=ws1)),
]
wsh = ops.tower_head_workspace(M, dev)
parts.append(("head(fwd_L1..dw_L2 fused)", lambda: ops.tower_head_fwd_bwd(h0, W1, b1, W2, b2, fm, labels, 0, -0.01, prob=prob, d_logit=dlog,
                                                                           d_h=dh1, loss=loss, workspace=wsh)))
wsn = ops.linear_bwd_narrow_workspace(M, 256, 32, dev)
parts.append(("narrow_L1(dx+dw fused)", lambda: ops.linear_bwd_narrow(h0, dh1, W1, -0.01, W1, b1, dh0, relu_mask=True, workspace=wsn)))
tot = 0
for name, fn in parts:
    t = timeit(fn); tot += t
    print("TAIL %-24s %.1f us" % (name, t))
def allparts():
    for _, fn in parts: fn()
print("TAIL sum %.1f us ; back-to-back sequence %.1f us" % (tot, timeit(allparts)))
