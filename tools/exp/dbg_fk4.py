import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from deep_recommenders_amd import ops
import test_gpu_fused_k4 as T
B, F, V, H = 1024, 6, 5000, 64
s = T._setup(ops, B, F, V, H, 0, seed=B + F)
t0, l0, b0, a0, plan = T._run(ops, s, fused=False)
t1, l1, b1, a1, _ = T._run(ops, s, fused=True)
d = (t1 - t0).abs()
rows = (d.max(1).values > 0).nonzero().flatten()
print("rows differing", rows.numel(), "max abs", d.max().item(), "rel to update", (d.max() / (t0 - s["table"]).abs().max()).item())
flags = plan.flags[:B * F].reshape(B, F).bool()
idc = s["ids"] + s["row_base"][None, :]
uniq_rows = idc[flags]
isu = torch.isin(rows, uniq_rows)
print("differing rows that are unique-slot rows:", int(isu.sum()), "of", rows.numel())
# per-column pattern
print("cols differing histogram (first 64):", (d > 0).sum(0).tolist()[:64])
print("lin equal", torch.equal(l1, l0), "amax", a0.item(), a1.item())
# which m,f
mf = (d[idc].max(-1).values > 0)
print("per field counts", mf.sum(0).tolist(), "rows m%256 hist:", torch.bincount((mf.any(1).nonzero().flatten() % 64), minlength=64).tolist())
lr = 0.05
x = s["table"][idc].double()
g0 = (t0[idc].double() - x) / (-lr)
g1 = (t1[idc].double() - x) / (-lr)
dx = (s["dy"].double() @ s["W"].double().t())[:, :F * 64].reshape(B, F, 64)
fm = s["dl"].double()[:, None, None] * (s["sum_x"].double()[:, None, :] - x)
bad = (flags & mf)
m_idx, f_idx = bad.nonzero()[0].tolist()
print("example bad slot m=%d f=%d" % (m_idx, f_idx))
print(" g0-dx-fm", (g0 - dx - fm)[m_idx, f_idx, :4].tolist())
print(" g1-dx-fm", (g1 - dx - fm)[m_idx, f_idx, :4].tolist())
print(" g1-dx   ", (g1 - dx)[m_idx, f_idx, :4].tolist(), " fm", fm[m_idx, f_idx, :4].tolist())
# is g1 = dx of another row + fm?
res = g1[m_idx, f_idx] - fm[m_idx, f_idx]
cand = (dx[:, f_idx, :] - res[None, :]).abs().max(1).values
print(" closest dx row for (g1 - fm):", int(cand.argmin()), float(cand.min()))
res2 = g1[m_idx, f_idx] - dx[m_idx, f_idx]
dlv = s["dl"].double()
for mm in range(max(0, m_idx - 40), min(B, m_idx + 40)):
    c = (dlv[mm] * (s["sum_x"][mm].double() - x[m_idx, f_idx]) - res2).abs().max().item()
    if c < 1e-6: print(" fm term matches dl/sx of row", mm)
    c2 = (dlv[mm] * (s["sum_x"][m_idx].double() - x[m_idx, f_idx]) - res2).abs().max().item()
    if c2 < 1e-6: print(" fm term matches dl of row", mm, "with own sx")
print("---- hypotheses")
dlm = dlv[m_idx]
sxp = (g1[m_idx, f_idx] - dx[m_idx, f_idx]) / dlm + x[m_idx, f_idx]
c = (s["sum_x"].double() - sxp[None, :]).abs().max(1).values
print(" sx' matches sum_x row", int(c.argmin()), float(c.min()), " own row err", float(c[m_idx]))
# v wrong AND in which way: residual vs dx rows of OTHER fields / other k-halves
res = g1[m_idx, f_idx] - fm[m_idx, f_idx]
dxa = (s["dy"].double() @ s["W"].double().t())
best = None
for ff in range(dxa.shape[1] // 64):
    cc = (dxa[:, ff * 64:(ff + 1) * 64] - res[None, :]).abs().max(1).values
    if best is None or cc.min() < best[0]: best = (float(cc.min()), ff, int(cc.argmin()))
print(" residual closest to dx[m=%d, f=%d] err %.2e" % (best[2], best[1], best[0]))
# partial sums? ratio
print(" ratio res/dx", (res / dx[m_idx, f_idx])[:8].tolist())
print("---- column-wise matches of residual (g1 - fm) against all dx entries of the same field")
col_match = []
for d in range(64):
    cc = (dxa[:, :].reshape(B, -1) - res[d]).abs()
    mm, kk = divmod(int(cc.argmin()), cc.shape[1])
    col_match.append((d, mm, kk, float(cc.min())))
print(col_match[:12])
# direct: compare d_concat written by the fused kernel for NON-unique slots with dx
import test_gpu_fused_k4 as T2
