"""EXPERIMENT (historical): K4 unique-row kernel time vs grid size (DR_EXP_K4_GRID hook, since removed): 1024..16384 blocks all
338-353 us -> the kernel is bandwidth / request bound, not latency bound."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deep_recommenders_amd import ops
dev = "cuda"
B, F, D, V = 65536, 26, 64, 10_000_000
g = torch.Generator(device=dev); g.manual_seed(1)
R = F * V
table = torch.empty((R, D), device=dev).normal_(0, 0.1, generator=g)
lin = torch.zeros(R, device=dev); bias = torch.zeros(1, device=dev)
rb = torch.arange(F, device=dev, dtype=torch.int64) * V
sets = []
for _ in range(6):
    ids = torch.randint(0, V, (B, F), device=dev, generator=g)
    sets.append((ids, ops.emb_sort_slots(ids, rb, R)))
d_concat = torch.randn((B, 1680), device=dev, generator=g) * 1e-3
concat = torch.randn((B, 1680), device=dev, generator=g); sum_x = torch.randn((B, D), device=dev, generator=g)
dl = torch.randn(B, device=dev, generator=g) * 1e-3
def run(k):
    ids, plan = sets[k]
    ops.emb_pool_bwd_sorted(ids, rb, plan, D, R, d_concat, dl, -1e-3, table, lin, bias, concat=concat, sum_x=sum_x)
for i in range(3): run(i)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for i in range(18): run(i % 6)
e.record(); torch.cuda.synchronize()
print("K4GRID %s  %.1f us" % (os.environ.get("DR_EXP_K4_GRID", "8192"), s.elapsed_time(e) / 18 * 1e3))
