#!/usr/bin/env python
"""Headline benchmark: examples/sec of one DeepFM training step (Criteo-shaped synthetic, BASELINE.json
configs[2]: 26 sparse + 13 dense, 10 M vocab per field, dim 64, batch 65 536 per GPU) on N MI355X GPUs.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched under
torch.distributed.run, one rank per GPU.  Rank 0 prints ONE JSON line.

A "step" = hash -> fused gather+pool+FM forward -> DNN forward -> sigmoid-CE -> DNN backward ->
scatter-add backward, with the SGD update fused into the wgrad / scatter kernels (no work skipped).
Inputs (raw keys, dense features, labels) are resident in HBM before the timed region.
"""
import argparse
import json
import os
import re
import sys
import time

# HIP multiplexes streams onto 4 hardware queues by default; the sharded engine uses 5+ (training, routing, communication,
# RCCL's own, sort) and two that share a queue serialize (seen in the kernel trace: the routing stream ran in-line with
# the GEMMs).  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# dmabuf IPC for RCCL (the host driver has no legacy IPC): read when the ROCm runtime initialises, so set it here too
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.3 TB/s achievable copy)
MFMA_F32_PEAK_TF = 157.3  # dense fp32 MFMA peak (same guide)
# the default GEMM mode forms every fp32 product from six v_mfma_f32_32x32x16_bf16 products: its ceiling is the dense bf16
# peak (2.5 PFLOP/s, same guide) divided by six
MFMA_BF16_PEAK_TF = 2500.0
MFMA_BF16X3_PEAK_TF = MFMA_BF16_PEAK_TF / 6.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=65536, help="per-GPU batch (weak scaling)")
    ap.add_argument("--fields", type=int, default=26)
    ap.add_argument("--vocab", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--dense", type=int, default=13)
    ap.add_argument("--dnn", type=str, default="256,32")
    ap.add_argument("--lr", type=float, default=0.01)
    ap.add_argument("--ids", choices=["uniform", "zipf"], default="uniform")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gemm", choices=["bf16x3", "native"], default="bf16x3",
                    help="fp32 GEMM products: six bf16 MFMA products of exact three-way splits (default) or v_mfma_f32_32x32x2_f32")
    ap.add_argument("--optimizer", choices=["sgd", "adam"], default="sgd",
                    help="sgd: fused SGD (the headline line); adam: fused row-wise Adam in K4 + dense Adam (single GPU only)")
    ap.add_argument("--micro-batches", type=int, default=int(os.environ.get("DR_MICRO_BATCHES", "2")),
                    help="sharded engine only: micro-batches per step (exchange of one overlaps the tower of the other)")
    ap.add_argument("--model", choices=["deepfm", "dcn"], default="deepfm",
                    help="deepfm: the headline (BASELINE config 3); dcn: config 4 (3 cross layers + MLP 1024,512,256), single GPU")
    ap.add_argument("--pool", type=int, default=8, help="number of distinct synthetic batches cycled through")
    return ap.parse_args()


def synth_batches(a, device, rank):
    """Criteo-shaped synthetic batches: raw int64 keys < 1e16 (<= 16 decimal chars),
    13 dense fp32 ~ log1p(|N(0,1)|), labels ~ Bernoulli(0.25); seed 42 (+rank)."""
    g = torch.Generator(device=device)
    g.manual_seed(42 + rank)
    out = []
    for _ in range(a.pool):
        if a.ids == "uniform":
            keys = torch.randint(0, 10**16, (a.batch, a.fields), device=device, generator=g)
        else:   # Zipf(1.05)-like raw keys: few hot keys, long tail (hashing spreads them over rows)
            u = torch.rand((a.batch, a.fields), device=device, generator=g, dtype=torch.float64)
            al, n = 1.05, float(10**12)
            keys = (((n ** (1 - al) - 1) * u + 1) ** (1 / (1 - al))).long().clamp(1, 10**12)
        dense = torch.log1p(torch.randn((a.batch, a.dense), device=device, generator=g).abs()) if a.dense else None
        labels = (torch.rand(a.batch, device=device, generator=g) < 0.25).float()
        out.append((keys, dense, labels))
    return out


def cpu_baseline(a, dnn_units):
    """The oracle port (oracle/torch_ref.py) of the same training step on the host cores, bounded sample."""
    from oracle import torch_ref as T
    import math
    ncores = os.cpu_count() or 1
    torch.set_num_threads(ncores)
    V = min(a.vocab, 1_000_000)         # 26 x 1 M x 64 fp32 = 6.7 GB: the full 66.6 GB does not fit typical host RAM
    try:
        import psutil
        if psutil.virtual_memory().available < 24e9:
            V = min(V, 100_000)
    except Exception:
        pass
    F, D, B = a.fields, a.dim, a.batch
    g = torch.Generator().manual_seed(42)
    table = torch.empty((F * V, D)).normal_(0, 1 / math.sqrt(D), generator=g)
    lin_w = torch.zeros(F * V)
    d = F * D + a.dense
    kernels, biases = [], []
    for u in dnn_units + [1]:
        lim = math.sqrt(6.0 / (d + u))
        kernels.append((torch.rand((d, u), generator=g) * 2 - 1) * lim)
        biases.append(torch.zeros(u))
        d = u
    row_base = [f * V for f in range(F)]
    ids = torch.randint(0, V, (B, F), generator=g)
    dense = torch.log1p(torch.randn((B, a.dense), generator=g).abs()) if a.dense else None
    labels = (torch.rand(B, generator=g) < 0.25).float()
    params = (table, lin_w, torch.zeros(()), kernels, biases)
    T.deepfm_train_step_sgd(params, ids, dense, labels, None, row_base, a.lr)       # warm-up
    t0 = time.perf_counter()
    n = 0
    while n < 20 and (n < 2 or time.perf_counter() - t0 < 10.0):
        ids = torch.randint(0, V, (B, F), generator=g)
        T.deepfm_train_step_sgd(params, ids, dense, labels, None, row_base, a.lr)
        n += 1
    dt = (time.perf_counter() - t0) / n
    return {"value": B / dt, "unit": "examples/sec", "cores": ncores, "kind": "port",
            "sample": "%d DeepFM SGD steps of batch %d on torch-CPU (oracle/torch_ref.py), tables scaled to "
                      "V=%d rows/field (%.1f GB); ids pre-hashed" % (n, B, V, F * V * D * 4 / 1e9)}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus %d needs a torch.distributed.run launch with %d ranks" % (a.gpus, a.gpus))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dnn_units = [int(x) for x in a.dnn.split(",") if x]
    from deep_recommenders_amd import ops as dr_ops
    dr_ops.set_gemm_mode(a.gemm)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=device)

    force_sharded = os.environ.get("DR_FORCE_SHARDED", "0") == "1"      # exercise the N>1 code path on one GPU
    if world == 1 and force_sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    if a.model == "dcn":
        if world != 1:
            raise SystemExit("--model dcn is a single-GPU line")
        from deep_recommenders_amd.dcn_engine import DCNEngine
        dnn_units = [int(x) for x in a.dnn.split(",") if x] if a.dnn != "256,32" else [1024, 512, 256]
        eng = DCNEngine(a.fields, a.vocab, a.dim, 3, dnn_units, a.batch, num_dense=a.dense, lr=a.lr, device=device)
    elif world == 1 and not force_sharded:
        from deep_recommenders_amd.engine import DeepFMEngine
        eng = DeepFMEngine(a.fields, a.vocab, a.dim, dnn_units, a.batch, num_dense=a.dense, lr=a.lr, device=device,
                           optimizer=a.optimizer)
    else:
        from deep_recommenders_amd.sharded import ShardedDeepFMEngine
        eng = ShardedDeepFMEngine(a.fields, a.vocab, a.dim, dnn_units, a.batch, num_dense=a.dense, lr=a.lr,
                                  device=device, world=world, rank=rank, micro_batches=a.micro_batches)
    batches = synth_batches(a, device, rank)

    nb = len(batches)
    for i in range(a.warmup):
        eng.train_step(*batches[i % nb], next_keys=batches[(i + 1) % nb][0])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    # ---- timed region: exactly K steps; per-kernel HIP events ride along on the launch stream -------
    prof = getattr(eng, "enable_kernel_events", None)
    if os.environ.get("DR_BENCH_EVENTS", "1") == "0":      # measurement-overhead check only: no per-kernel numbers
        prof = None
    if prof is not None:
        eng.enable_kernel_events(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):          # the data loader knows the next batch: its keys are handed over for route prefetch
        eng.train_step(*batches[(a.warmup + i) % nb], next_keys=batches[(a.warmup + i + 1) % nb][0])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    loss = float(eng.loss.item())
    ms = dt / a.steps * 1e3
    value = a.batch * world * a.steps / dt

    kernels = eng.kernel_event_summary() if prof is not None else {}
    # HBM traffic per launch from the PMC counters (collected in separate rocprofv3 --pmc passes of this same command,
    # corrected as MI355X_MICROARCH.md prescribes; committed under profiles/): event name -> profiled kernel name
    traffic = {}
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))["kernels"]
        bf3 = "true" if a.gemm == "bf16x3" else "false"
        name_map = {"emb_pool_fwd": "emb_pool_fwd_sv_kernel<16,8>", "emb_pool_bwd": "emb_bwd_unique_kernel<16,4,false>",
                    "hash_bucket_i64": "hash_bucket_i64_kernel",
                    "linear_fwd_L0": "gemm_f32_mfma_kernel<true,false,0,false,false,%s>" % bf3,
                    "linear_bwd_dx_L0": "gemm_f32_mfma_kernel<true,true,2,false,false,%s>" % bf3,
                    "linear_bwd_dw_L0": "gemm_f32_mfma_kernel<false,false,3,false,false,%s>" % bf3}
        default_cfg = (a.batch, a.fields, a.vocab, a.dim, a.dense, a.dnn) == (65536, 26, 10_000_000, 64, 13, "256,32")
        if default_cfg and world == 1:
            traffic = {ev: int(pmc[k]["hbm_bytes_corrected"]) for ev, k in name_map.items() if k in pmc}
    except Exception:
        traffic = {}
    roof_all = []
    comm_phases = {}
    overlapped = {}
    for name, k in kernels.items():
        sec = k["ms"] * 1e-3
        if k["bound"] == "xgmi":      # exchange phases (all-to-all / all-reduce + their local halves): reported apart
            comm_phases[name] = {"avg_us": round(k["ms"] * 1e3, 2), "bytes_per_rank": k.get("alg_bytes"), "launches": k["n"]}
            continue
        if k["bound"] == "overlap":   # side-stream work hidden under the main stream: its event time is stretched by the
            notes = {"emb_sort_slots": (167.0, "rocPRIM radix sort of B*F slots + unique flags; runs concurrently with "
                                               "emb_pool_fwd / linear_fwd_L0, not on the critical path"),
                     "emb_pool_bwd": (336.0, "sorted scatter-add backward (HBM-bound) on its own stream, concurrent with the "
                                             "MFMA-bound linear_bwd_dw_L0; standalone = in-process A/B at this shape")}
            alone, note = next((v for kk, v in notes.items() if name.startswith(kk)), (None, ""))
            overlapped[name] = {"event_us_while_overlapped": round(k["ms"] * 1e3, 2), "launches": k["n"],
                                "standalone_us": alone, "alg_bytes": k.get("alg_bytes"), "note": note}
            continue
        if k["bound"] == "hbm":
            ach, peak, unit = k["alg_bytes"] / sec / 1e9, HBM_PEAK_GBS, "GB/s"
        else:
            # the wide-tile GEMMs run on the bf16 pipe in the default mode (6 products per fp32 product); the narrow-tile
            # and fused-tail kernels always use the fp32 MFMA
            on_bf16 = a.gemm == "bf16x3" and re.match(r"^(linear|cross)_(fwd|bwd_dx|bwd_dw)_L\d+$", name) is not None
            ach, peak, unit = k["alg_flops"] / sec / 1e12, (MFMA_BF16X3_PEAK_TF if on_bf16 else MFMA_F32_PEAK_TF), "TFLOP/s"
        row = {"kernel": name, "bound": k["bound"], "achieved": round(ach, 2), "peak": round(peak, 1), "unit": unit,
               "frac": round(ach / peak, 4), "traffic": traffic.get(name), "avg_us": round(k["ms"] * 1e3, 2),
               "launches": k["n"]}
        if k["bound"] == "mfma":
            row["peak_basis"] = ("dense bf16 MFMA peak 2500 TFLOP/s / 6 products per fp32 product" if on_bf16
                                 else "dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32)")
            row["frac_of_f32_mfma_peak"] = round(ach / MFMA_F32_PEAK_TF, 4)
        if k.get("concurrent_with"):
            row["concurrent_with"] = k["concurrent_with"]     # runs on a second stream next to that kernel: its event time
                                                              # (and so `achieved`) is stretched by the sharing
        roof_all.append(row)
    roof_all.sort(key=lambda r: -r["avg_us"])
    # the dominant kernel for `roofline` is the longest one that has the chip to itself (a pair of concurrent kernels cannot
    # be priced one by one); the concurrent ones stay in roofline_all, flagged
    solo = [r for r in roof_all if "concurrent_with" not in r]
    roofline = dict((solo or roof_all)[0]) if roof_all else None

    if rank == 0:
        out = {
            "metric": "examples/sec/node DeepFM (26 sparse feats, 10M vocab, dim 64)" if a.model == "deepfm"
            else "examples/sec DCN (26 sparse + 13 dense, 10M vocab, dim 64, 3 cross layers + MLP %s) -- BASELINE config 4" % dnn_units,
            "value": round(value, 1), "unit": "examples/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "DeepFM Criteo-shape synthetic training step (BASELINE.json configs[2]): %d sparse + %d "
                                   "dense, %d vocab/field hashed on device, dim %d, DNN %s+[1] relu, sigmoid-CE, fused %s; "
                                   "batch %d per GPU, ids %s; tables %s"
                                   % (a.fields, a.dense, a.vocab, a.dim, dnn_units, "SGD" if a.optimizer == "sgd" else "Adam (row-wise on the tables)", a.batch, a.ids,
                                      "on one GPU (%.1f GB)" % (a.fields * a.vocab * a.dim * 4 / 1e9) if world == 1
                                      else "row-sharded over %d GPUs (id %% N), RCCL all-to-all, %d micro-batches per step" % (world, getattr(eng, "mb", 1))),
                       "gemm_products": ("fp32 in / fp32 accumulate; products = 6 bf16 MFMA products of exact 3-way bf16 splits (error vs fp64 <= native fp32 MFMA, tests/test_gpu_kernels.py)"
                                         if a.gemm == "bf16x3" else "native v_mfma_f32_32x32x2_f32"),
                       "global_batch": a.batch * world, "parallelism": "single" if world == 1 else "dp%d+row-sharded-tables" % world,
                       "final_loss": round(loss, 6)},
            "roofline": roofline,
            "roofline_all": roof_all,
        }
        if comm_phases:
            out["exchange_phases"] = comm_phases
        if overlapped:
            out["overlapped_side_stream"] = overlapped
        if world == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(a, dnn_units)
            except Exception as e:     # the baseline is reported, never the thing measured
                out["cpu_baseline"] = {"value": None, "unit": "examples/sec", "cores": os.cpu_count(), "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        try:        # RCCL's version banner sits in the C stdio buffer until exit: push it out BEFORE the JSON line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
