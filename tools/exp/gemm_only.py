"""EXPERIMENT: run only the L0 GEMMs a few times (target for rocprofv3 --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deep_recommenders_amd import ops
dev = "cuda"
g = torch.Generator(device=dev); g.manual_seed(0)
M, K, N = 65536, 1677, int(os.environ.get("GN", "256"))
x = torch.randn((M, 1680), device=dev, generator=g)[:, :K]
W = torch.randn((K, N), device=dev, generator=g) / K ** 0.5
b = torch.zeros(N, device=dev)
y = torch.empty((M, N), device=dev)
dy = torch.randn((M, N), device=dev, generator=g)
dx = torch.empty((M, 1680), device=dev)[:, :K]
dW = torch.zeros((K, N), device=dev)
ws = ops.linear_bwd_dw_workspace(M, K, N, dev)
for _ in range(3):
    ops.linear_fwd(x, W, b, 1, out=y)
    ops.linear_bwd_dx(dy, W, None, out=dx)
    ops.linear_bwd_dw(x, dy, 1e-6, dW, b, workspace=ws)
torch.cuda.synchronize()
