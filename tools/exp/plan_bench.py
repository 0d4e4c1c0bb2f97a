"""Times the slot plan (dr_emb_sort_slots) alone at the bench size: uniform keys (claim path) and Zipf keys (radix path)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deep_recommenders_amd import ops

B, F, V = 65536, 26, 10_000_000
g = torch.Generator(device="cuda").manual_seed(1)
rb = torch.arange(F, device="cuda", dtype=torch.int64) * V
buckets = torch.full((F,), V, dtype=torch.int64, device="cuda")
for kind in ("uniform", "zipf"):
    if kind == "uniform":
        keys = torch.randint(0, 10**16, (B, F), device="cuda", generator=g)
    else:
        u = torch.rand((B, F), device="cuda", generator=g, dtype=torch.float64)
        al, nn = 1.05, float(10**12)
        keys = (((nn ** (1 - al) - 1) * u + 1) ** (1 / (1 - al))).long().clamp(1, 10**12)
    ids = ops.hash_bucket_i64(keys, buckets)
    plan = ops.SortPlan(B * F, "cuda")
    for forced in (False, True):
        prev = ops.emb_plan_set_small_limit(0 if forced else 16384)
        for _ in range(3):
            ops.emb_sort_slots(ids, rb, F * V, plan)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        n = 20
        for _ in range(n):
            ops.emb_sort_slots(ids, rb, F * V, plan)
        e.record()
        torch.cuda.synchronize()
        ops.emb_plan_set_small_limit(prev)
        print("%s keys, %s: %.1f us per plan (sorted list %d of %d slots, %d work-list heads)" % (
            kind, "radix path forced" if forced else "default path", s.elapsed_time(e) / n * 1e3, plan.sorted_len(), B * F,
            int(plan.dup_count[0].item())))
