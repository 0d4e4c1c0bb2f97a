"""EXPERIMENT driver: ReLU sign disagreements of the first layer's output against fp64, f16x2 vs bf16x3 vs a plain fp32 matmul."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from deep_recommenders_amd import ops
F, D, ND, B, N, V = 26, 64, 13, 65536, 256, 1_000_000
g = torch.Generator(device="cuda"); g.manual_seed(7)
std = 0.125
table = (torch.randn((F * V, D), device="cuda", generator=g) * std).clamp_(-2 * std, 2 * std)
row_base = (torch.arange(F, device="cuda") * V).to(torch.int64)
K = F * D + ND
lim = (6.0 / (K + N)) ** 0.5
W = (torch.rand((K, N), device="cuda", generator=g) * 2 - 1) * lim
b = torch.zeros(N, device="cuda")
for trial in range(3):
    ids = torch.randint(0, V, (B, F), device="cuda", generator=g)
    dense = torch.log1p(torch.randn((B, ND), device="cuda", generator=g).abs())
    dpad = torch.zeros((B, 32), device="cuda"); dpad[:, :ND] = dense
    x = torch.cat([table[(ids + row_base[None, :]).reshape(-1)].view(B, F * D), dense], 1)
    ref = torch.empty((B, N), dtype=torch.float64, device="cuda")
    for r0 in range(0, B, 8192):
        ref[r0:r0 + 8192] = x[r0:r0 + 8192].double() @ W.double()
    sx, fm = torch.empty((B, D), device="cuda"), torch.empty(B, device="cuda")
    y3, y2 = torch.empty((B, N), device="cuda"), torch.empty((B, N), device="cuda")
    wp3, wp2 = ops.WeightPlanes(W), ops.H2WeightPlanes(W)
    ops.bf3_emb_linear_fwd(ids, row_base, V, table, None, None, dpad, None, K, wp3.wt, b, 0, sx, fm, y3)
    ops.h2_emb_linear_fwd(ids, row_base, V, table, ops.h2_amax(table), None, None, dpad, ops.h2_amax(dpad), None, K, wp2.wt, b, 0, sx, fm, y2)
    y1 = x @ W
    y1c = (x.cpu() @ W.cpu()).cuda()
    for nm, y in (("f16x2", y2), ("bf16x3", y3), ("torch f32 gpu", y1), ("torch f32 cpu", y1c)):
        e = (y.double() - ref).abs()
        flips = ((y > 0) != (ref > 0)).sum().item()
        print("trial %d %-14s max|err| %.3e  rms err %.3e  mean |err| %.3e  sign flips vs fp64 %d   flips vs cpu-f32 %d" % (
            trial, nm, e.max().item(), e.pow(2).mean().sqrt().item(), e.mean().item(), flips, ((y > 0) != (y1c > 0)).sum().item()), flush=True)
