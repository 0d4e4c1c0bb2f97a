"""Per-row measurements at the BASELINE.json sizes that bench.py does not cover (configs 4 and 5):
DCN cross layer (a9), in-batch softmax (a10), exact top-K MIPS (a11).  HIP events on the launch stream."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deep_recommenders_amd import ops

dev = "cuda"
g = torch.Generator(device=dev); g.manual_seed(42)


def timeit(fn, iters=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


res = {}
# ---- config 4: DCN cross layer, B = 65536, Din = 26*64 + 13 = 1677 ---------------------------------------
B, Dm = 65536, 1677
ld = 1680
x0 = torch.randn((B, ld), device=dev, generator=g)[:, :Dm]
x = torch.randn((B, ld), device=dev, generator=g)[:, :Dm]
W = torch.randn((Dm, Dm), device=dev, generator=g) * 0.05
b = torch.zeros(Dm, device=dev)
fl = 2.0 * B * Dm * Dm
t = timeit(lambda: ops.cross_fwd(x0, x, W, b, 0.0, want_prod=True))
res["cross_fwd_65536x1677 (GEMM + x0*(.)+x epilogue, prod saved)"] = {"ms": t * 1e3, "TF/s": fl / t / 1e12, "frac_of_157.3": fl / t / 157.3e12}
out, prod = ops.cross_fwd(x0, x, W, b, 0.0, want_prod=True)
d_out = torch.randn((B, ld), device=dev, generator=g)[:, :Dm]
d_x0 = torch.zeros((B, ld), device=dev)[:, :Dm]
d_x = torch.zeros((B, ld), device=dev)[:, :Dm]
gW = torch.zeros_like(W); gb = torch.zeros_like(b)
ws = ops.linear_bwd_dw_workspace(B, Dm, Dm, dev)


def cross_bwd():
    d_prod = ops.cross_combine_bwd(x0, prod, d_out, 0.0, d_x0, d_x)
    ops.linear_bwd_dx(d_prod, W, None, accumulate=True, out=d_x)
    ops.linear_bwd_dw(x, d_prod, 1.0, gW, gb, workspace=ws)


t = timeit(cross_bwd)
res["cross_bwd_65536x1677 (combine + dgrad + wgrad)"] = {"ms": t * 1e3, "TF/s": 2 * fl / t / 1e12, "frac_of_157.3": 2 * fl / t / 157.3e12}
del x0, x, W, out, prod, d_out, d_x0, d_x, gW, ws
torch.cuda.empty_cache()

# ---- config 5: two-tower, B = 8192 in-batch, D = 128, corpus 1 M items, k = 100 -----------------------------
Bq, D, N, k = 8192, 128, 1_000_000, 100
q = torch.randn((Bq, D), device=dev, generator=g) / D ** 0.5
c = torch.randn((Bq, D), device=dev, generator=g) / D ** 0.5
t = timeit(lambda: ops.inbatch_softmax_fwd(q, c))
fl = 2.0 * Bq * Bq * D
res["inbatch_softmax_fwd_8192x128 (scores never materialised)"] = {"ms": t * 1e3, "TF/s": fl / t / 1e12}
loss, lse, pos = ops.inbatch_softmax_fwd(q, c)


def sm_bwd():
    G = ops.inbatch_softmax_grad_scores(q, c, lse, 1.0)
    ops.linear_fwd(G, c)
    dc = torch.zeros_like(c)
    ops.linear_bwd_dw(G, q, 1.0, dc)


t = timeit(sm_bwd)
res["inbatch_softmax_bwd_8192x128 (G + 2 GEMMs)"] = {"ms": t * 1e3, "TF/s": 3 * fl / t / 1e12}
corpus = torch.randn((N, D), device=dev, generator=g) / D ** 0.5
state = ops.topk_state(Bq, k, dev)
wsz = ops.lib().dr_topk_workspace_bytes(Bq, N, k)
wsb = torch.empty(wsz // 4, device=dev)
t = timeit(lambda: ops.topk_mips(q, corpus, k, state=state, workspace=wsb), iters=2, warm=1)
fl = 2.0 * Bq * N * D
res["topk_mips_8192q_x_1M_x128_k100 (chunked scores + register top-k)"] = {"ms": t * 1e3, "TF/s": fl / t / 1e12, "frac_of_157.3": fl / t / 157.3e12,
                                                                          "workspace_MB": wsz / 1e6}
# split of the search: scores only
chunk = wsz // (Bq * 4)
t2 = timeit(lambda: ops.scores_nt(q, corpus[:chunk], out=wsb[:Bq * chunk].view(Bq, chunk)), iters=3, warm=1)
res["  of which one chunk of scores (%d candidates)" % chunk] = {"ms": t2 * 1e3, "TF/s": 2.0 * Bq * chunk * D / t2 / 1e12}
t3 = timeit(lambda: ops.topk_select(wsb[:Bq * chunk].view(Bq, chunk), k, 0, False, state), iters=3, warm=1)
res["  of which one chunk of selection (steady state)"] = {"ms": t3 * 1e3, "GB/s": Bq * chunk * 4 / t3 / 1e9}
# IVF-Flat approximate search on the same corpus: build once, then time the search and measure recall
import time
from deep_recommenders_amd.keras.models.retrieval import factorized_top_k as ftk
for nlist, nprobe in ((1024, 8), (1024, 32)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ivf = ftk.Faiss(k=k, nlist=nlist, nprobe=nprobe).index(corpus)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    t = timeit(lambda: ivf(q), iters=3, warm=1)
    _, got = ivf(q)
    exact = state[1]
    hit = (got[:256, :, None] == exact[:256, None, :]).any(-1).float().mean().item()
    scanned = Bq * nprobe * (N / nlist)
    res["ivf_flat_8192q_x_1M_x128_k100 nlist=%d nprobe=%d" % (nlist, nprobe)] = {
        "ms": t * 1e3, "recall@100_vs_exact": hit, "build_s": build_s, "scan_GB/s": scanned * D * 4 / t / 1e9}
for kname, v in res.items():
    print("ROWS", kname, json.dumps({a: round(b, 4) for a, b in v.items()}))
