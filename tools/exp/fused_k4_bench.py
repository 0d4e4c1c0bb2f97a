"""EXPERIMENT driver (not product): the fused first-layer dgrad + K4 (dr_h2_dgrad_emb_sgd) against dgrad + K4 at config 3's shape
(B = 65 536, F = 26, D = 64, 256-wide layer; V rows per field from argv, default 2 M = 13 GB of tables: random rows miss every cache)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from deep_recommenders_amd import ops

dev = "cuda"
B, F, D, H = 65536, 26, 64, 256
V = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
zipf = len(sys.argv) > 2 and sys.argv[2] == "zipf"
torch.manual_seed(0)
R = F * V
if zipf:
    import numpy as np
    rng = np.random.default_rng(0)
    ids = torch.as_tensor(np.minimum(rng.zipf(1.05, size=(B, F)) - 1, V - 1)).to(dev)
else:
    ids = torch.randint(0, V, (B, F), device=dev)
row_base = (torch.arange(F, dtype=torch.int64) * V).to(dev)
table = torch.randn((R, D), device=dev) * 0.125
lin = torch.zeros(R, device=dev)
bias = torch.zeros(1, device=dev)
in_dim = F * D + 13
W = torch.randn((in_dim, H), device=dev) / in_dim ** 0.5
dy = torch.randn((B, H), device=dev) * (torch.rand((B, H), device=dev) > 0.5) / B
dl = torch.randn(B, device=dev) / B
sum_x = torch.randn((B, D), device=dev)
lin_old_t = torch.zeros((F, B), device=dev)
plan = ops.emb_sort_slots(ids, row_base, R)
ids_t = ops.ids_transpose_i32(ids)
wp = ops.H2WeightPlanes(W)
dy_am = ops.h2_amax(dy)
tab_am = ops.h2_amax(table)
ld = (in_dim + 3) // 4 * 4
d_concat = torch.zeros((B, ld), device=dev)
xs = torch.zeros((B * F, D), device=dev)
lr = 1e-3


def timeit(fn, n=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


k4 = lambda parts: ops.emb_pool_bwd_sorted(ids, row_base, plan, D, R, d_concat, dl, -lr, table, lin, bias, sum_x=sum_x, x_sorted=xs, parts=parts,
                                           lin_old_t=lin_old_t, table_amax=tab_am)
dgrad = lambda: ops.h2_linear_nt(dy, dy_am, wp.w, out=d_concat[:, :in_dim])
fused = lambda: ops.h2_dgrad_emb_sgd(dy, dy_am, wp.w, ids_t, plan, row_base, table, lin, lin_old_t, sum_x, dl, -lr, d_concat, table_amax=tab_am)
uniq = int(plan.flags[:B * F].sum().item())
print("FUSEDK4 V=%d %s  unique slots %.4f  occ=%s" % (V, "zipf" if zipf else "uniform", uniq / (B * F), os.environ.get("DR_H2_OCC", "1")), flush=True)
for rep in range(2):
    t_d = timeit(dgrad)
    t_k = timeit(lambda: k4(1))
    t_pair = timeit(lambda: (dgrad(), k4(1), k4(2)))
    t_f = timeit(fused)
    t_kd = timeit(lambda: k4(1 | 8))
    t_fpair = timeit(lambda: (fused(), k4(1 | 8), k4(2 | 8)))
    print("FUSEDK4 rep %d: dgrad %.1f  K4 %.1f  dgrad+K4+hot %.1f  |  fused %.1f  K4 dups-only %.1f  fused+dups+hot %.1f us" %
          (rep, t_d, t_k, t_pair, t_f, t_kd, t_fpair), flush=True)
