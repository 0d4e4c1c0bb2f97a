"""Per-kernel timing on the GPU (HIP events on the launch stream) at the Criteo-shaped config-3 sizes.
Usage: python tools/microbench.py [--V 10000000] [--B 65536] [--what emb,gemm,hash,loss]"""
import argparse
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deep_recommenders_amd import ops


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--V", type=int, default=10_000_000)
    ap.add_argument("--B", type=int, default=65536)
    ap.add_argument("--F", type=int, default=26)
    ap.add_argument("--D", type=int, default=64)
    ap.add_argument("--what", default="hash,emb,gemm,loss")
    a = ap.parse_args()
    B, F, D, V = a.B, a.F, a.D, a.V
    dev = "cuda"
    res = {}
    what = a.what.split(",")
    g = torch.Generator(device=dev)
    g.manual_seed(42)
    if "hash" in what:
        keys = torch.randint(0, 10**15, (B, F), device=dev, generator=g)
        buckets = torch.full((F,), V, dtype=torch.int64, device=dev)
        out = torch.empty_like(keys)
        t = timeit(lambda: ops.hash_bucket_i64(keys, buckets, out))
        res["hash_bucket_i64"] = {"us": t * 1e6, "GB/s": B * F * 16 / t / 1e9}
    if "emb" in what:
        R = V * F
        table = torch.empty((R, D), dtype=torch.float32, device=dev)
        table.normal_(0, 0.125, generator=g)
        lin_w = torch.zeros(R, dtype=torch.float32, device=dev)
        row_base = torch.arange(F, device=dev, dtype=torch.int64) * V
        col_start = torch.arange(F + 1, device=dev, dtype=torch.int32)
        ld = (F * D + 13 + 3) // 4 * 4
        for dist in ("uniform", "zipf"):
            if dist == "uniform":
                ids = torch.randint(0, V, (B, F), device=dev, generator=g)
            else:
                u = torch.rand((B, F), device=dev, generator=g, dtype=torch.float64)
                # Zipf-like (alpha = 1.05) via inverse CDF of a continuous power law on [1, V]
                al = 1.05
                ids = (((V ** (1 - al) - 1) * u + 1) ** (1 / (1 - al))).long().clamp(1, V) - 1
            concat = torch.zeros((B, ld), dtype=torch.float32, device=dev)
            sum_x = torch.empty((B, D), dtype=torch.float32, device=dev)
            fm = torch.empty((B,), dtype=torch.float32, device=dev)
            t = timeit(lambda: ops.emb_pool_fwd(ids, F, None, row_base, table, lin_w, None, ld_concat=ld, concat=concat,
                                                sum_x=sum_x, fm_logit=fm))
            alg = B * (8 * F * D + 12 * F + 8)
            res["emb_pool_fwd_" + dist] = {"us": t * 1e6, "alg_GB/s": alg / t / 1e9, "frac_of_8TB/s": alg / t / 8e12}
            d_concat = torch.randn((B, ld), device=dev, generator=g) * 1e-3
            d_fm = torch.randn((B,), device=dev, generator=g) * 1e-3
            alg = B * (12 * F * D + 16 * F)
            t = timeit(lambda: ops.emb_pool_bwd(ids, F, col_start, row_base, D, d_concat, concat, sum_x, d_fm, -1e-3, table, lin_w))
            res["emb_pool_bwd_atomic_%s" % dist] = {"us": t * 1e6, "alg_GB/s": alg / t / 1e9, "frac_of_8TB/s": alg / t / 8e12}
            NB = 4
            idsl = [ids] + [torch.randint(0, V, (B, F), device=dev, generator=g) for _ in range(NB - 1)] if dist == "uniform" else [ids]
            plan = ops.SortPlan(B * F, dev); fl = plan.flags
            cnt = [0]
            def do_sort():
                cnt[0] += 1
                ops.emb_sort_slots(idsl[cnt[0] % len(idsl)], row_base, R, plan)
            t = timeit(do_sort)
            res["emb_sort_slots_" + dist] = {"us": t * 1e6}
            ops.emb_sort_slots(ids, row_base, R, plan)
            t = timeit(lambda: ops.emb_pool_bwd_sorted(ids, row_base, plan, D, R, d_concat, d_fm, -1e-3, table, lin_w))
            res["emb_pool_bwd_sorted_" + dist] = {"us": t * 1e6, "alg_GB/s": alg / t / 1e9, "frac_of_8TB/s": alg / t / 8e12,
                                                   "unique_frac": float(fl.float().mean())}
        # copy ceiling for reference: device-to-device copy of the concat-sized buffer
        src = torch.empty(256 * 1024 * 1024 // 4, device=dev)
        dst = torch.empty_like(src)
        t = timeit(lambda: dst.copy_(src))
        res["d2d_copy_256MB"] = {"us": t * 1e6, "GB/s": 2 * src.numel() * 4 / t / 1e9}
        del table, lin_w
    if "gemm" in what:
        for (M, K, N) in [(B, 1677, 256), (B, 256, 32), (B, 1677, 1024), (B, 1024, 512), (B, 1677, 1677)]:
            ldx = (K + 3) // 4 * 4
            x = torch.randn((M, ldx), device=dev, generator=g)[:, :K]
            W = torch.randn((K, N), device=dev, generator=g) / K ** 0.5
            b = torch.zeros(N, device=dev)
            y = torch.empty((M, N), device=dev)
            t = timeit(lambda: ops.linear_fwd(x, W, b, 1, out=y), iters=5, warmup=2)
            fl = 2.0 * M * K * N
            res["linear_fwd_%dx%dx%d" % (M, K, N)] = {"us": t * 1e6, "TF/s": fl / t / 1e12}
            t = timeit(lambda: torch.relu(torch.addmm(b, x, W)), iters=5, warmup=2)
            res["torch_addmm_relu_%dx%dx%d" % (M, K, N)] = {"us": t * 1e6, "TF/s": fl / t / 1e12}
            dy = torch.randn((M, N), device=dev, generator=g)
            dx = torch.empty((M, ldx), device=dev)[:, :K]
            t = timeit(lambda: ops.linear_bwd_dx(dy, W, x, out=dx), iters=5, warmup=2)
            res["linear_bwd_dx_%dx%dx%d" % (M, K, N)] = {"us": t * 1e6, "TF/s": fl / t / 1e12}
            dW = torch.zeros((K, N), device=dev)
            db = torch.zeros(N, device=dev)
            t = timeit(lambda: ops.linear_bwd_dw(x, dy, 1e-6, dW, db), iters=5, warmup=2)
            res["linear_bwd_dw_%dx%dx%d" % (M, K, N)] = {"us": t * 1e6, "TF/s": fl / t / 1e12}
    if "loss" in what:
        x = torch.randn(B, device=dev, generator=g)
        z = (torch.rand(B, device=dev, generator=g) < 0.25).float()
        ws = torch.empty(1024, device=dev)
        t = timeit(lambda: ops.bce_fwd_bwd(x, z, 0, workspace=ws))
        res["bce_fwd_bwd"] = {"us": t * 1e6}
    for k, v in res.items():
        print(k, json.dumps({kk: round(vv, 3) for kk, vv in v.items()}))


if __name__ == "__main__":
    main()
