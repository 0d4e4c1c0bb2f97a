"""EXPERIMENT: cost split of the sorted K4 (with / without first-order weights, with / without FM term)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deep_recommenders_amd import ops
dev = "cuda"
B, F, D, V = 65536, 26, 64, 10_000_000
g = torch.Generator(device=dev); g.manual_seed(1)
R = F * V
table = torch.empty((R, D), device=dev).normal_(0, 0.1, generator=g)
lin = torch.zeros(R, device=dev)
rb = torch.arange(F, device=dev, dtype=torch.int64) * V
ld = 1680
NB = 6
sets = []
for _ in range(NB):
    ids = torch.randint(0, V, (B, F), device=dev, generator=g)
    sets.append((ids, ops.emb_sort_slots(ids, rb, R)))
d_concat = torch.randn((B, ld), device=dev, generator=g) * 1e-3
concat = torch.randn((B, ld), device=dev, generator=g)
sum_x = torch.randn((B, D), device=dev, generator=g)
dl = torch.randn(B, device=dev, generator=g) * 1e-3
bias = torch.zeros(1, device=dev)


def timeit(fn, iters=12, warm=3):
    for i in range(warm): fn(i % NB)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters): fn(i % NB)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def run(k, lin_on, fm_on):
    ids, plan = sets[k]
    ops.emb_pool_bwd_sorted(ids, rb, plan, D, R, d_concat, dl, -1e-3, table, lin if lin_on else None,
                            bias if lin_on else None, concat=concat if fm_on else None, sum_x=sum_x if fm_on else None)


for lin_on in (True, False):
    for fm_on in (True, False):
        t = timeit(lambda k: run(k, lin_on, fm_on))
        print("K4PARTS lin=%d fm=%d  %.1f us" % (lin_on, fm_on, t))
