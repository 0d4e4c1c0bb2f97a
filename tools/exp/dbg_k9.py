import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from deep_recommenders_amd import ops
rng = np.random.default_rng(8)
B, D = 513, 128
q = (rng.standard_normal((B, D)) / np.sqrt(D)).astype(np.float32)
c = (rng.standard_normal((B, D)) / np.sqrt(D)).astype(np.float32)
w = rng.uniform(0.5, 1.5, size=B).astype(np.float32)
p = rng.uniform(0.05, 0.9, size=B).astype(np.float32)
ids = rng.integers(0, max(2, B // 3), size=B)
tq, tc = torch.tensor(q).cuda(), torch.tensor(c).cuda()
loss, lse, pos = ops.inbatch_softmax_fwd(tq, tc, None, torch.tensor(ids).cuda(), None, 1.0)
print("loss", loss.item(), "nan lse", torch.isnan(lse).sum().item(), "nan pos", torch.isnan(pos).sum().item(), "inf lse", torch.isinf(lse).sum().item())
bad = torch.nonzero(torch.isnan(lse) | torch.isinf(lse)).reshape(-1).tolist()[:10]
print("bad rows", bad)
s = q @ c.T
ident = ids.reshape(-1, 1)
dup = (ident == ident.T).astype(np.float32) - np.eye(B, dtype=np.float32)
s2 = s + dup * np.float32(np.finfo(np.float32).min / 100)
import scipy.special as sp
ref = sp.logsumexp(s2.astype(np.float64), axis=1)
for r in bad[:3]:
    print(r, lse[r].item(), ref[r], "ndups", int(dup[r].sum()))
print("max abs diff (finite)", np.nanmax(np.abs(np.where(np.isfinite(lse.cpu().numpy()), lse.cpu().numpy() - ref, np.nan))))
print("---- combined")
tp, tw, ti = torch.tensor(p).cuda(), torch.tensor(w).cuda(), torch.tensor(ids).cuda()
for name, args in [("p+ids", (tp, ti, None, 1.0)), ("p+ids+T", (tp, ti, None, 1 / 0.7)), ("ids+w", (None, ti, tw, 1.0)), ("all", (tp, ti, tw, 1 / 0.7))]:
    loss, lse, pos = ops.inbatch_softmax_fwd(tq, tc, *args)
    print(name, "loss", loss.item(), "nan lse", torch.isnan(lse).sum().item(), "nan pos", torch.isnan(pos).sum().item(),
          "inf lse", torch.isinf(lse).sum().item(), "inf pos", torch.isinf(pos).sum().item())
    bad = torch.nonzero(torch.isnan(lse) | torch.isinf(lse) | torch.isnan(pos)).reshape(-1).tolist()[:10]
    print("   bad rows", bad, [ (lse[r].item(), pos[r].item()) for r in bad[:3]])
