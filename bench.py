#!/usr/bin/env python
"""Headline benchmark: examples/sec of one DeepFM training step (Criteo-shaped synthetic, BASELINE.json
configs[2]: 26 sparse + 13 dense, 10 M vocab per field, dim 64, batch 65 536 per GPU) on N MI355X GPUs.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`.  For N > 1 it runs one rank per GPU over RCCL: either the
caller wraps it (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N`;
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) or, when WORLD_SIZE is not set, bench.py starts the N ranks
itself (same launcher, a free port on 127.0.0.1) after checking that N devices are visible.  Rank 0 prints ONE JSON line.
`--launch-check`: rendezvous + one small exchange per collective, no training (runs on CPU over gloo: the launch path's unit
test).  `--share-device`: all ranks on cuda:0 with host-staged collectives -- verification of the N > 1 data path on a 1-GPU
box, never a measurement.

A "step" = hash -> fused gather+pool+FM forward -> DNN forward -> sigmoid-CE -> DNN backward ->
scatter-add backward, with the SGD update fused into the wgrad / scatter kernels (no work skipped).
Inputs (raw keys, dense features, labels) are resident in HBM before the timed region.
"""
import argparse
import json
import os
import re
import subprocess
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
# the first layer's three GEMMs of the DeepFM engine run in the "f16x2" operand mode by default (round 4): two fp16 terms per value,
# THREE v_mfma_f32_32x32x16_f16 products per fp32 product (the fp16 pipe has the bf16 pipe's dense rate)
MFMA_F16X2_PEAK_TF = MFMA_BF16_PEAK_TF / 3.0
H2_KERNELS = ("emb_linear_fwd_L0", "linear_bwd_dx_L0", "linear_bwd_dw_L0")


def _lib_sha256():
    try:
        import hashlib
        p = os.path.join(ROOT, "deep_recommenders_amd", "lib", "libdr_hotpath.so")
        return hashlib.sha256(open(p, "rb").read()).hexdigest()
    except Exception:
        return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults (round 5: 100 / 20, were 20 / 5): the chip needs ~20 steps to reach its steady state -- with 5 warm-up steps the first timed
    # steps run the three GEMMs 5-8 % slow (clock / power ramp; profiles/r05_steps_warmup.log: 1.185-1.205 ms at 20 / 5 against 1.148 at
    # 100 / 20 and 1.140-1.145 at 400 / 20, same box) -- and a 24 ms timed region is short for a wall-clock measurement.  120 steps = 0.14 s.
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
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
    ap.add_argument("--model", choices=["deepfm", "dcn", "dssm"], default="deepfm",
                    help="deepfm: the headline (BASELINE config 3); dcn: config 4 (3 cross layers + MLP 1024,512,256); dssm: config 5 "
                         "(two towers + in-batch softmax over the batch's 8192 candidates + FactorizedTopK pass over a 1 M corpus)")
    ap.add_argument("--preset", choices=["c2"], default=None,
                    help="c2: BASELINE config 2, the MovieLens-1M shape (7 sparse features, 10 K vocab, dim 16, batch 4096, no dense)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong scaling: the batch of the WHOLE job (BASELINE config 3 as worded: 65536 over 8 GPUs); per-GPU batch = this / N")
    ap.add_argument("--events", choices=["auto", "on", "off"], default="auto",
                    help="per-kernel HIP events in the timed region (roofline rows).  auto: off for steps too short to carry two event "
                         "records per kernel (first-layer GEMM < 4 GFLOP, e.g. the MovieLens shape)")
    ap.add_argument("--items", type=int, default=1_000_000, help="dssm: corpus size")
    ap.add_argument("--users", type=int, default=1_000_000, help="dssm: user hash buckets")
    ap.add_argument("--towers", type=str, default="256,128", help="dssm: tower widths (last = embedding dim of the retrieval space)")
    ap.add_argument("--topk", type=int, default=100, help="dssm: k of the FactorizedTopK metric pass")
    ap.add_argument("--pool", type=int, default=8, help="number of distinct synthetic batches cycled through")
    ap.add_argument("--launch-check", action="store_true",
                    help="start / join the N ranks, run one small exchange per collective through the engine's transport, print one JSON "
                         "line; no model, no timing (gloo on a box without GPUs)")
    ap.add_argument("--share-device", action="store_true",
                    help="N > 1 ranks all on cuda:0, collectives staged through the host over gloo (sharded.HostStagedTransport): checks "
                         "the N > 1 data path with the real kernels on a 1-GPU box; the printed rate is NOT a measurement")
    a = ap.parse_args()
    if a.preset == "c2":
        a.fields, a.vocab, a.dim, a.dense, a.batch = 7, 10_000, 16, 0, 4096
    if a.model == "dssm":
        if a.batch == 65536:
            a.batch = 8192
        if a.dim == 64:
            a.dim = 128
        if a.lr == 0.01:
            a.lr = 1e-5        # Retrieval's loss is a SUM over the batch (sbcnm.py:100-102): plain SGD at 0.01 diverges within a few steps
    if a.global_batch:
        world = int(os.environ.get("WORLD_SIZE", "1"))
        assert a.global_batch % world == 0, "--global-batch must divide by the number of GPUs"
        a.batch = a.global_batch // world
    return a


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


def _median_rate(step_fn, batch, warmup, min_steps, cap_s):
    """examples/s of step_fn: `warmup` untimed calls, then at least `min_steps` timed calls (more until `cap_s` seconds of
    timed work, at most 20), median over the per-call times."""
    for _ in range(warmup):
        step_fn()
    times = []
    t_all = time.perf_counter()
    while len(times) < 20 and (len(times) < min_steps or time.perf_counter() - t_all < cap_s):
        t0 = time.perf_counter()
        step_fn()
        times.append(time.perf_counter() - t0)
    times.sort()
    return batch / times[len(times) // 2], len(times)


def cpu_baseline(a, dnn_units):
    """The oracle port (oracle/torch_ref.py) of the same training step on the host cores, on a bounded sample of the workload
    (SURVEY.md §8d / BASELINE.md §2: warm-up, median, forward-only and forward+backward separately).  The true reference
    (TF-CPU) cannot run here (no TensorFlow): `kind` is "port"."""
    from oracle import torch_ref as T
    import math
    ncores = os.cpu_count() or 1
    torch.set_num_threads(ncores)
    g = torch.Generator().manual_seed(42)
    if a.model == "dssm":
        D, B, units = a.dim, a.batch, [int(x) for x in a.towers.split(",") if x]
        ut = torch.empty((a.users, D)).normal_(0, 1 / math.sqrt(D), generator=g)
        it = torch.empty((a.items, D)).normal_(0, 1 / math.sqrt(D), generator=g)

        def mk():
            Ws, d = [], D
            for u in units:
                Ws.append(((torch.rand((d, u), generator=g) * 2 - 1) * math.sqrt(6.0 / (d + u)), torch.zeros(u)))
                d = u
            return Ws
        qW, cW = mk(), mk()

        def tower(x, Ws):
            for i, (W, b) in enumerate(Ws):
                x = x @ W + b
                if i < len(Ws) - 1:
                    x = torch.relu(x)
            return x

        def step(train):
            uid, iid = torch.randint(0, a.users, (B,), generator=g), torch.randint(0, a.items, (B,), generator=g)
            ue, ie = ut[uid].requires_grad_(train), it[iid].requires_grad_(train)
            ps = [t.requires_grad_(train) for W, b in qW + cW for t in (W, b)]
            loss = T.inbatch_softmax_loss(tower(ue, qW), tower(ie, cW), cand_ids=iid)
            if train:
                gs = torch.autograd.grad(loss, [ue, ie] + ps)
                with torch.no_grad():
                    ut.index_add_(0, uid, gs[0], alpha=-a.lr)
                    it.index_add_(0, iid, gs[1], alpha=-a.lr)
                    for t, gr in zip(ps, gs[2:]):
                        t.sub_(a.lr * gr)
            for t in ps:
                t.requires_grad_(False)
        full, n1 = _median_rate(lambda: step(True), B, 2, 5, 10.0)
        with torch.no_grad():
            fwd, n2 = _median_rate(lambda: step(False), B, 2, 5, 5.0)
        return {"value": full, "unit": "examples/sec", "cores": ncores, "kind": "port", "forward_only_value": fwd,
                "sample": "median of %d two-tower SGD steps (and %d forward-only passes) of batch %d on torch-CPU autograd over the "
                          "oracle's restated Retrieval loss (oracle/torch_ref.py), 2 warm-up calls each; full-size tables (%d users, "
                          "%d items, dim %d); ids pre-hashed; top-K metric pass not included" % (n1, n2, B, a.users, a.items, D)}
    V = min(a.vocab, 1_000_000)         # 26 x 1 M x 64 fp32 = 6.7 GB: the full 66.6 GB does not fit typical host RAM
    try:
        import psutil
        if psutil.virtual_memory().available < 24e9:
            V = min(V, 100_000)
    except Exception:
        pass
    F, D = a.fields, a.dim
    B = min(a.batch, 8192)              # bounded sample: an eighth of the GPU batch keeps (5 + 20) + (5 + 10) calls in ~15 s
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
    dense = torch.log1p(torch.randn((B, a.dense), generator=g).abs()) if a.dense else None
    labels = (torch.rand(B, generator=g) < 0.25).float()
    params = (table, lin_w, torch.zeros(()), kernels, biases)

    def train():
        ids = torch.randint(0, V, (B, F), generator=g)
        T.deepfm_train_step_sgd(params, ids, dense, labels, None, row_base, a.lr)

    rb_t = torch.as_tensor(row_base)[None, :]

    def fwd_only():     # the forward of deepfm_train_step_sgd (same single-valued formulation), no autograd
        ids = torch.randint(0, V, (B, F), generator=g)
        with torch.no_grad():
            rows = ids + rb_t
            emb = torch.nn.functional.embedding(rows, table)
            x = emb.reshape(B, -1)
            x = x if dense is None else torch.cat([x, dense], 1)
            logit = T.fm_second_order(emb) + lin_w[rows].sum(1) + params[2] + T.dnn(x, kernels, biases).squeeze(1)
            T.sigmoid_cross_entropy(labels, logit)
    # Thread count: torch's default (one thread per logical CPU) oversubscribes this workload badly on a many-core host (round 2:
    # 2.5 K examples/s on "256 cores", one tenth of what 8 threads of a small box reach) -- probe a few counts with two calls each
    # and keep the best; `cores` reports the threads actually used.
    best_nt, best_t = ncores, None
    probe = {}
    for nt in sorted({min(ncores, c) for c in (8, 16, 32, 64, ncores)}):
        torch.set_num_threads(nt)
        train()
        t0 = time.perf_counter()
        train()
        train()
        dt = time.perf_counter() - t0
        probe[str(nt)] = round(2 * B / dt, 1)
        if best_t is None or dt < best_t:
            best_nt, best_t = nt, dt
    torch.set_num_threads(best_nt)
    full, n1 = _median_rate(train, B, 5, 10, 10.0)
    fwd, n2 = _median_rate(fwd_only, B, 5, 20, 8.0)
    return {"value": full, "unit": "examples/sec", "cores": best_nt, "kind": "port", "forward_only_value": fwd, "host_logical_cpus": ncores,
            "thread_probe_examples_per_sec": probe,      # two training calls per thread count; `cores` is the winner
            "sample": "median of %d DeepFM SGD steps (and %d forward-only passes) of batch %d on torch-CPU (oracle/torch_ref.py), 5 warm-up "
                      "calls each (BASELINE.md section 2: 5 + 20; the training leg stops at its 10 s cap); torch threads = %d, the best of a "
                      "probe over {8, 16, 32, 64, %d}; tables scaled to V=%d rows/field (%.1f GB); ids pre-hashed; DeepFM tower also for "
                      "--model dcn" % (n1, n2, B, best_nt, ncores, V, F * V * D * 4 / 1e9)}


def _visible_gpus():
    try:
        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:
        return 0


def _device_count_message(need, have):
    return ("bench.py --gpus %d: %d GPU(s) visible on this node (torch.cuda.device_count()), %d needed -- one rank per GPU.  "
            "Use --gpus <= %d, or --share-device to push the N > 1 data path through ONE GPU (verification only)."
            % (need, have, need, max(have, 1)))


def self_launch(a):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks ourselves (torch.distributed.run, one process per
    GPU, rendezvous on a free port of 127.0.0.1 -- the container hostname may not resolve) and exit with the launcher's status."""
    have = _visible_gpus()
    if not a.share_device and not (a.launch_check and have == 0) and have < a.gpus:
        raise SystemExit(_device_count_message(a.gpus, have))
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % a.gpus, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "8")
    env["DR_BENCH_SELF_LAUNCHED"] = "1"
    print("[bench] starting %d ranks: %s" % (a.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    raise SystemExit(subprocess.call(cmd, env=env))


def launch_check(a, world, rank, device, dist, transport):
    """One small instance of every collective the sharded engines issue, through the transport they would use; rank 0 prints the line."""
    ok = {}
    t = torch.full((4,), float(rank + 1), device=device)
    transport.allreduce(t)
    ok["allreduce"] = bool(abs(float(t[0]) - world * (world + 1) / 2) < 1e-6)
    send = (torch.arange(world * 3, device=device, dtype=torch.int64) + 100 * rank).view(world * 3, 1).contiguous()
    recv = torch.empty_like(send)
    transport.alltoall(recv, send, [3] * world, [3] * world)
    want = torch.cat([torch.arange(rank * 3, rank * 3 + 3, dtype=torch.int64) + 100 * r for r in range(world)])
    ok["alltoall"] = bool(torch.equal(recv.view(-1).cpu(), want))
    g = torch.empty((world * 2,), device=device)
    transport.allgather(g, torch.full((2,), float(rank), device=device))
    ok["allgather"] = bool(torch.equal(g.view(world, 2)[:, 0].cpu(), torch.arange(world, dtype=torch.float32)))
    seen = [None] * world
    dist.all_gather_object(seen, {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "device": str(device),
                                  "pid": os.getpid(), "ok": ok})
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": dist.get_world_size(), "backend": dist.get_backend(), "gpus_visible": _visible_gpus(),
                          "self_launched": os.environ.get("DR_BENCH_SELF_LAUNCHED") == "1", "transport": type(transport).__name__,
                          "ranks": seen, "all_ok": all(all(r["ok"].values()) for r in seen)}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        self_launch(a)                  # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%d: the launcher's --nproc-per-node must equal --gpus" % (a.gpus, world))
    have = _visible_gpus()
    cpu_only = a.launch_check and have == 0
    if a.share_device:
        local_rank = 0
    if not cpu_only and local_rank >= have:
        raise SystemExit(_device_count_message(a.gpus, have))
    device = torch.device("cpu") if cpu_only else torch.device("cuda", local_rank)
    if not cpu_only:
        torch.cuda.set_device(local_rank)
    dist = None
    transport = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        from deep_recommenders_amd import sharded as _sh
        if cpu_only or a.share_device:
            dist.init_process_group("gloo")
            transport = _sh.TorchDistTransport() if cpu_only else _sh.HostStagedTransport()
        else:
            dist.init_process_group("nccl", device_id=device)
            transport = _sh.default_transport(world, rank)
    if a.launch_check:
        if dist is None:
            print(json.dumps({"launch_check": True, "n_gpus": 1, "gpus_visible": have, "note": "one rank: nothing to rendezvous"}), flush=True)
            return
        return launch_check(a, world, rank, device, dist, transport)
    dnn_units = [int(x) for x in a.dnn.split(",") if x]
    from deep_recommenders_amd import ops as dr_ops
    dr_ops.set_gemm_mode(a.gemm)

    force_sharded = os.environ.get("DR_FORCE_SHARDED", "0") == "1"      # exercise the N>1 code path on one GPU
    if world == 1 and force_sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    sharded = world > 1 or force_sharded
    extra = {}
    if a.model == "dssm":
        towers = [int(x) for x in a.towers.split(",") if x]
        if sharded:
            from deep_recommenders_amd.sharded_retrieval import ShardedTwoTowerEngine
            eng = ShardedTwoTowerEngine(a.users, a.items, a.dim, towers, a.batch, lr=a.lr, k=a.topk, device=device, world=world, rank=rank,
                                        transport=transport)
        else:
            from deep_recommenders_amd.two_tower_engine import TwoTowerEngine
            eng = TwoTowerEngine(a.users, a.items, a.dim, towers, a.batch, lr=a.lr, k=a.topk, device=device)
        g = torch.Generator(device=device)
        g.manual_seed(42 + rank)
        batches = [(torch.randint(0, 10**16, (a.batch,), device=device, generator=g),
                    torch.randint(0, a.items, (a.batch,), device=device, generator=g)) for _ in range(a.pool)]
        step = lambda i: eng.train_step(*batches[i % len(batches)])
    else:
        if a.model == "dcn":
            dnn_units = [int(x) for x in a.dnn.split(",") if x] if a.dnn != "256,32" else [1024, 512, 256]
            if sharded:
                from deep_recommenders_amd.sharded import ShardedDCNEngine
                eng = ShardedDCNEngine(a.fields, a.vocab, a.dim, 3, dnn_units, a.batch, num_dense=a.dense, lr=a.lr, device=device,
                                       world=world, rank=rank, transport=transport)
            else:
                from deep_recommenders_amd.dcn_engine import DCNEngine
                eng = DCNEngine(a.fields, a.vocab, a.dim, 3, dnn_units, a.batch, num_dense=a.dense, lr=a.lr, device=device)
        elif not sharded:
            from deep_recommenders_amd.engine import DeepFMEngine
            eng = DeepFMEngine(a.fields, a.vocab, a.dim, dnn_units, a.batch, num_dense=a.dense, lr=a.lr, device=device,
                               optimizer=a.optimizer)
        else:
            from deep_recommenders_amd.sharded import ShardedDeepFMEngine
            eng = ShardedDeepFMEngine(a.fields, a.vocab, a.dim, dnn_units, a.batch, num_dense=a.dense, lr=a.lr,
                                      device=device, world=world, rank=rank, micro_batches=a.micro_batches, transport=transport,
                                      **({"optimizer": a.optimizer} if a.optimizer != "sgd" else {}))
        batches = synth_batches(a, device, rank)
        nb = len(batches)
        # the data loader knows the next batch: its keys are handed over for route prefetch
        if isinstance(eng, __import__("deep_recommenders_amd.engine", fromlist=["DeepFMEngine"]).DeepFMEngine):
            # (single-GPU engine: the next batch's dense features ride along, they are placed while this step's K4 runs)
            step = lambda i: eng.train_step(*batches[i % nb], next_keys=batches[(i + 1) % nb][0], next_dense=batches[(i + 1) % nb][1])
        else:
            step = lambda i: eng.train_step(*batches[i % nb], next_keys=batches[(i + 1) % nb][0])

    # per-kernel HIP events: decided here, because the untimed steps below must run the code path of the timed region
    prof = getattr(eng, "enable_kernel_events", None)
    first_gemm_flops = 2.0 * a.batch * (a.fields * a.dim + a.dense) * (dnn_units[0] if dnn_units else 1)
    events_on = a.events == "on" or (a.events == "auto" and (a.model != "deepfm" or first_gemm_flops >= 4e9))
    if os.environ.get("DR_BENCH_EVENTS", "1") == "0" or not events_on:
        prof = None
    ev_every = 1
    # The sharded step drives four streams; a step with event records on them takes anywhere from 1.1 x to 5 x as long as one without
    # (round 4, same box, same flags: 1.96 and 3.85 ms per step with every 4th step bracketed, 1.68 - 1.71 ms with none).  So with
    # --events auto the timed region of a sharded DeepFM run carries NO events; the per-phase durations, the exchange report and the
    # exposed waits are taken from EV_EXTRA bracketed steps run after the timed region and its loss read-out (same batches, same
    # prefetch), and the line says so.  --events on: every 4th step of the timed region, as before.
    EV_EXTRA = 8
    # (round 5: the two-tower step too -- ~120 small launches, bound by the host: bracketing every kernel of the timed region took the
    # line from 1.07-1.09 to 2.75-4.0 ms on this round's boxes, profiles/r05_bench_line_dssm*.json; its per-kernel rows come from
    # bracketed steps after the timed region as well)
    events_after = (prof is not None and a.events == "auto" and
                    ((a.model == "deepfm" and hasattr(eng, "exchange_report")) or a.model == "dssm"))
    ev_steps = a.steps
    if events_after:
        prof_after, prof = prof, None

    # Round 6: inside the timed region of the single-GPU DeepFM run only the HEADLINE kernel is bracketed (the embedding kernel `roofline`
    # is quoted on: two event records per bracketed step instead of eighteen -- with every kernel bracketed every 4th step the 0.97 ms
    # step read 0.979 - 0.985, with none 0.967 - 0.968); the other rows of roofline_all come from EV_EXTRA fully bracketed steps run after
    # the timed region and say so.  DR_BENCH_EVENTS_ALL=1: every kernel inside the timed region, as in rounds 1 - 5.
    HEADLINE_KERNELS = ("emb_pool_bwd_fused_dgrad_L0", "emb_pool_bwd", "emb_pool_bwd_adam", "emb_pool_fwd")
    headline_only = (prof is not None and a.model == "deepfm" and hasattr(eng, "reserve_kernel_events")
                     and os.environ.get("DR_BENCH_EVENTS_ALL", "0") != "1")

    def events_begin():
        # the DeepFM engines bracket every 4th step of the timed region (steps 0, 4, 8, ...): two event records per kernel
        # on EVERY step cost ~5 % of a 1.5 ms step (round 3: 1.56 vs 1.48 ms; sharded engine, round 4: 2.05 vs 1.94 ms); the per-kernel
        # averages are over those launches
        try:
            every = 4 if (a.model == "deepfm" and a.steps >= 8) else 1
            if headline_only:
                eng.enable_kernel_events(True, every=every, only=HEADLINE_KERNELS)
            else:
                eng.enable_kernel_events(True, every=every)
        except TypeError:
            every = 1
            eng.enable_kernel_events(True)
        return every

    if prof is not None and hasattr(eng, "reserve_kernel_events"):
        # every event pair the timed region will record exists before the first step (round 6: creating them inside the timed region cost
        # the FIRST process of a fresh box 0.3 - 0.45 ms per step -- lines of 1.32 / 1.44 ms whose kernel rows were normal)
        eng.reserve_kernel_events(((a.steps + 3) // 4 + 8) * 32)
    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    if prof is not None:
        ev_every = events_begin()        # (the settling blocks run what the timed region runs; their samples are dropped below)
    # Settling (single process only; untimed, like the warm-up): further blocks of 20 steps until two consecutive blocks agree within 2 %
    # (or 5 s have passed).  The W warm-up steps cover the chip's ramp; this covers the HOST: the step needs 0.33 ms of launch work per
    # 1.14 ms of GPU work, and on a box that is still busy with its own start-up (seen twice as the first command of a call: 1.52 and
    # 2.69 ms lines whose per-kernel times were normal) the launches fall behind for a few seconds.  Normally 2 blocks = 46 ms.
    settle_steps = 0
    if dist is None and a.warmup > 0 and os.environ.get("DR_BENCH_SETTLE", "1") == "1":
        prev_t, best_t, t_begin = None, None, time.perf_counter()
        while settle_steps < 6000 and time.perf_counter() - t_begin < 5.0:
            t1 = time.perf_counter()
            for i in range(20):
                step(a.warmup + settle_steps + i)
            torch.cuda.synchronize()
            cur_t = time.perf_counter() - t1
            settle_steps += 20
            best_t = cur_t if best_t is None else min(best_t, cur_t)
            # (round 6: two blocks that agree are not enough -- on a host busy with its neighbours' start-up the launches fall behind at
            # a STABLE slow rate for a second or two, seen as a 1.31 ms line between 1.03 ms ones; the block must also be within 5 % of
            # the fastest block seen so far)
            if prev_t is not None and abs(cur_t - prev_t) <= 0.02 * prev_t and cur_t <= 1.05 * best_t:
                break
            prev_t = cur_t
    if dist is not None:
        dist.barrier()
    # ---- timed region: exactly K steps; per-kernel HIP events ride along on the launch stream -------
    if prof is not None:
        ev_every = events_begin()        # (drops the samples of the settling blocks; their event pairs are reused)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + settle_steps + i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device="cpu" if a.share_device else device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    loss = float(eng.loss.item())
    ms = dt / a.steps * 1e3
    value = a.batch * world * a.steps / dt
    kernels_timed = None
    if headline_only:
        # the headline kernel's samples of the timed region; then every kernel of EV_EXTRA further steps for the other rows
        kernels_timed = eng.kernel_event_summary()
        eng.enable_kernel_events(True, every=1)
        for i in range(EV_EXTRA):
            step(a.warmup + settle_steps + a.steps + i)
        torch.cuda.synchronize()
    if events_after:
        prof = prof_after
        ev_every, ev_steps = 1, EV_EXTRA
        try:
            eng.enable_kernel_events(True, every=1)
        except TypeError:
            eng.enable_kernel_events(True)
        for i in range(EV_EXTRA):
            step(a.warmup + settle_steps + a.steps + i)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # N > 1, weak line: the SAME job at SURVEY 8(e)'s split as well -- a global batch of `--batch` examples divided over the ranks (strong
    # scaling) -- so that one driver command yields both scaling curves.  A second engine on every rank (per-rank
    # batch = batch / N), 5 warm-up + 20 timed steps bracketed like the main region; failures are reported, never fatal.
    strong = None
    if (world > 1 and not a.global_batch and a.model == "deepfm" and sharded and os.environ.get("DR_BENCH_STRONG", "1") == "1"
            and a.batch % world == 0):
        try:
            from deep_recommenders_amd.sharded import ShardedDeepFMEngine
            pb = a.batch // world
            eng_s = ShardedDeepFMEngine(a.fields, a.vocab, a.dim, dnn_units, pb, num_dense=a.dense, lr=a.lr, device=device, world=world,
                                        rank=rank, micro_batches=a.micro_batches, transport=transport,
                                        **({"optimizer": a.optimizer} if a.optimizer != "sgd" else {}))
            sb = [(k[:pb].contiguous(), d[:pb].contiguous() if d is not None else None, l[:pb].contiguous()) for k, d, l in batches]
            step_s = lambda i: eng_s.train_step(*sb[i % nb], next_keys=sb[(i + 1) % nb][0])
            n_w, n_t = 5, min(20, a.steps)
            for i in range(n_w):
                step_s(i)
            torch.cuda.synchronize()
            dist.barrier()
            t1 = time.perf_counter()
            for i in range(n_t):
                step_s(n_w + i)
            torch.cuda.synchronize()
            dist.barrier()
            dts = time.perf_counter() - t1
            tt = torch.tensor([dts], device="cpu" if a.share_device else device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dts = tt.item()
            strong = {"scaling": "strong", "global_batch": a.batch, "per_gpu_batch": pb, "steps": n_t, "warmup": n_w,
                      "ms_per_step": round(dts / n_t * 1e3, 4), "value": round(a.batch * n_t / dts, 1), "unit": "examples/sec",
                      "final_loss": round(float(eng_s.loss.item()), 6),
                      "note": "SURVEY 8(e)'s partitioning (global batch divided over the ranks), measured after the weak timed region in the same process group"}
            del eng_s, step_s, sb
        except Exception as e_s:
            strong = {"failed": repr(e_s)}

    # the slot plan K4 needs, ALONE on the idle chip (after the timed region): what it would add to K4 if it did not run beside the
    # previous step's K4 -- the overlapped event time in `overlapped_side_stream` is stretched by that sharing
    plan_alone_us = None
    if a.model == "deepfm" and not sharded and getattr(eng, "sorted_bwd", False) and hasattr(eng, "plan"):
        from deep_recommenders_amd import ops as _ops
        torch.cuda.synchronize()
        for _ in range(3):
            _ops.emb_sort_slots(eng.ids, eng.row_base, eng.R, plan=eng.plan)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            _ops.emb_sort_slots(eng.ids, eng.row_base, eng.R, plan=eng.plan)
        e1.record()
        torch.cuda.synchronize()
        plan_alone_us = e0.elapsed_time(e1) / 10 * 1e3
    # ... and K4 itself alone (same batch, its plan just rebuilt above; parts = 1: the update kernel only).  In the step its event
    # time includes what sharing the chip with the next batch's hash + plan costs it, which varies from run to run; this is the
    # kernel by itself.  (It applies the last batch's update ten more times -- after the timed region and the loss read-out.)
    k4_alone_us = None
    if plan_alone_us is not None and a.optimizer == "sgd" and getattr(eng, "d_concat", None) is not None:
        k4 = lambda: _ops.emb_pool_bwd_sorted(eng.ids, eng.row_base, eng.plan, eng.D, eng.R, eng.d_concat, eng.d_logit, -eng.lr,
                                              eng.table, eng.lin_w, eng.lin_bias,
                                              concat=None if eng.no_concat else eng.concat, sum_x=eng.sum_x,
                                              x_sorted=eng.x_sorted if eng.no_concat else None, parts=1,
                                              lin_old_t=getattr(eng, "lin_old_t", None) if getattr(eng, "_lin_old_valid", False) else None,
                                              table_amax=getattr(eng, "tab_amax", None) if getattr(eng, "h2", False) else None)
        for _ in range(3):
            k4()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            k4()
        e1.record()
        torch.cuda.synchronize()
        k4_alone_us = e0.elapsed_time(e1) / 10 * 1e3

    # ... and the bare forward gather + pool kernel (K3 + K5 + K6: dr_emb_pool_fwd, what DR_FUSE_K3=0 runs) on the same batch, alone:
    # in the default step K3 lives inside the first layer's MFMA-bound GEMM and has no HBM roofline of its own; north_star asks for
    # the forward AND the backward fraction.  It writes concat / sum_x / fm_logit of the last batch again.
    k3_alone_us = None
    if plan_alone_us is not None and getattr(eng, "concat", None) is not None and hasattr(eng, "alg_bytes_fwd"):
        k3 = lambda: _ops.emb_pool_fwd(eng.ids, eng.F, None, eng.row_base, eng.table, eng.lin_w, eng.lin_bias, ld_concat=eng.ld,
                                       concat=eng.concat, sum_x=eng.sum_x, fm_logit=eng.fm_logit)
        for _ in range(3):
            k3()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            k3()
        e1.record()
        torch.cuda.synchronize()
        k3_alone_us = e0.elapsed_time(e1) / 10 * 1e3

    kernels = eng.kernel_event_summary() if prof is not None else {}
    if kernels_timed is not None:
        for name in kernels:
            kernels[name]["after_timed_region"] = name not in kernels_timed
        kernels.update({n: dict(v, after_timed_region=False) for n, v in kernels_timed.items()})
    pairs = eng.concurrent_pair_summary() if (prof is not None and hasattr(eng, "concurrent_pair_summary")) else []
    exchange = None
    if prof is not None and hasattr(eng, "exchange_report"):
        exchange = eng.exchange_report(ev_steps)                 # (before the events are dropped below)
        if events_after:
            exchange["measured_in"] = "%d bracketed steps run AFTER the timed region (a bracketed sharded step is slower; ms_per_step has none)" % EV_EXTRA
        exchange["ranks_in_process_group"] = dist.get_world_size() if dist is not None else 1
        exchange["gpus_visible"] = have
        exchange["backend"] = dist.get_backend() if dist is not None else None
    if prof is not None:
        eng.enable_kernel_events(False)
    if a.model == "dssm":
        # FactorizedTopK pass (factorized_top_k.py:489-512): corpus index = item tower over all items, then exact top-k of
        # every query against it; reported next to the training step (Retrieval.call(compute_metrics=True) does both)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        eng.index_corpus()
        torch.cuda.synchronize()
        t_index = time.perf_counter() - t1
        eng.metric_step(*batches[0])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        nm = 3
        for i in range(nm):
            hits = eng.metric_step(*batches[i % len(batches)])
        torch.cuda.synchronize()
        t_metric = (time.perf_counter() - t1) / nm
        extra["metric_pass"] = {"index_corpus_ms": round(t_index * 1e3, 3), "topk_ms_per_batch": round(t_metric * 1e3, 3),
                                "k": a.topk, "corpus_items": a.items,
                                "scan_tflops": round(2.0 * a.batch * world * a.items * eng.out_dim / t_metric / 1e12, 2),
                                "top_k_hits_last_batch": [int(h) for h in hits.cpu().tolist()],
                                "note": "exact MIPS top-%d of %d queries against the item tower's outputs for all %d items%s"
                                        % (a.topk, a.batch * world, a.items, " (corpus row-sharded, local top-k -> all-gather -> merge)" if sharded else "")}
    # HBM traffic per launch from the PMC counters: NOT measured in this run -- replayed from the committed rocprofv3 --pmc
    # passes of this same command (profiles/, see `traffic_source`), only for the default configuration they were taken on
    traffic, traffic_source = {}, None
    traffic_note = "HBM counters need the profiler (rocprofv3 --pmc passes of this command, tools/collect_profiles.sh); not collected inside bench.py"
    try:
        src = next(("profiles/%s_pmc_traffic.json" % r for r in ("r06", "r05", "r04", "r03", "r02", "r01")
                    if os.path.exists(os.path.join(ROOT, "profiles", "%s_pmc_traffic.json" % r))), "profiles/r01_pmc_traffic.json")
        pmc = json.load(open(os.path.join(ROOT, src)))
        name_map = pmc.get("event_names") or {}
        # the static figures describe the LIBRARY they were collected on (tools/collect_profiles.sh records its sha256): a kernel change
        # that doubled the traffic must not keep printing the old number -- a different library gets no figure
        lib_sha = _lib_sha256()
        if not pmc.get("lib_sha256"):
            raise RuntimeError("%s records no library hash (collected before round 6): re-collect with tools/collect_profiles.sh" % src)
        if not lib_sha or pmc["lib_sha256"] != lib_sha:
            raise RuntimeError("%s was collected on libdr_hotpath.so %s..., this run loads %s...: re-collect with tools/collect_profiles.sh"
                               % (src, pmc["lib_sha256"][:12], (lib_sha or "?")[:12]))
        default_cfg = (a.model, a.batch, a.fields, a.vocab, a.dim, a.dense, a.dnn, a.optimizer, a.gemm) == \
            ("deepfm", 65536, 26, 10_000_000, 64, 13, "256,32", "sgd", "bf16x3")
        if default_cfg and world == 1 and not force_sharded:
            traffic = {ev: int(pmc["kernels"][k]["hbm_bytes_corrected"]) for ev, k in name_map.items() if k in pmc["kernels"]}
            traffic_source = src + (" (static: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command by tools/collect_profiles.sh, "
                                    "not this run; collected on the library this run loads: sha256 %s)" % (lib_sha or "?")[:400])
            if getattr(eng, "h2", False) and pmc.get("gemm_split", "bf16x3") != "f16x2":
                # the committed passes were taken with the first layer's GEMMs in the bf16x3 mode: their byte counts (three weight planes
                # per k-tile, not two) do not describe the f16x2 kernels this run timed.  K4 and the tail kernels are the same code.
                for ev in H2_KERNELS:
                    traffic.pop(ev, None)
                traffic_source += "; the three first-layer GEMM rows carry no figure: the passes predate the f16x2 mode (profiles/README.md)"
        else:
            traffic_note = ("the committed PMC passes (%s) were taken on the default configuration (deepfm, B 65536, 26 x 10 M x 64, DNN 256,32, sgd, "
                            "bf16x3, one GPU); this run differs, so no traffic figure applies" % src)
    except Exception as e:
        traffic = {}
        traffic_note = "no static PMC figure applies (%s)" % (e,)
    roof_all = []
    comm_phases = {}
    overlapped = {}
    for name, k in kernels.items():
        sec = k["ms"] * 1e-3
        if k["bound"] == "xgmi":      # exchange phases (all-to-all / all-reduce + their local halves): reported apart
            comm_phases[name] = {"avg_us": round(k["ms"] * 1e3, 2), "bytes_per_rank": k.get("alg_bytes"), "launches": k["n"]}
            continue
        if k["bound"] == "stall":     # training-stream waits on exchange events: reported in `exchange`
            continue
        if k["bound"] == "overlap":   # side-stream work hidden under the main stream: its event time is stretched by the sharing
            overlapped[name] = {"event_us_while_overlapped": round(k["ms"] * 1e3, 2), "launches": k["n"], "alg_bytes": k.get("alg_bytes")}
            continue
        if k["bound"] == "hbm":
            ach, peak, unit = k["alg_bytes"] / sec / 1e9, HBM_PEAK_GBS, "GB/s"
        else:
            # the wide-tile GEMMs run on the bf16 pipe in the default mode (6 products per fp32 product); the narrow-tile
            # and fused-tail kernels always use the fp32 MFMA
            on_bf16 = a.gemm == "bf16x3" and re.match(r"^(linear|emb_linear|cross|[qc]_tower)_(fwd|bwd_dx|bwd_dw)_L\d+$", name) is not None
            on_h2 = on_bf16 and bool(getattr(eng, "h2", False)) and (name in H2_KERNELS or bool(getattr(eng, "h2_all_wide", False)))
            ach, peak, unit = k["alg_flops"] / sec / 1e12, (MFMA_F16X2_PEAK_TF if on_h2 else MFMA_BF16X3_PEAK_TF if on_bf16 else MFMA_F32_PEAK_TF), "TFLOP/s"
        row = {"kernel": name, "bound": k["bound"], "achieved": round(ach, 2), "peak": round(peak, 1), "unit": unit,
               "frac": round(ach / peak, 4), "traffic": traffic.get(name), "avg_us": round(k["ms"] * 1e3, 2),
               "launches": k["n"]}
        if k.get("after_timed_region"):
            row["events"] = "%d fully bracketed steps after the timed region" % EV_EXTRA
        if k["bound"] == "mfma":
            row["peak_basis"] = ("dense fp16 MFMA peak 2500 TFLOP/s / 3 products per fp32 product (f16x2 operand mode)" if on_h2
                                 else "dense bf16 MFMA peak 2500 TFLOP/s / 6 products per fp32 product" if on_bf16
                                 else "dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32)")
            if on_h2:     # continuity with rounds 2-3, which priced these kernels against the six-product ceiling
                row["frac_of_bf16x3_ceiling"] = round(ach / MFMA_BF16X3_PEAK_TF, 4)
        if name == "emb_linear_fwd_L0":      # K3 fused into the first layer's GEMM: priced on the MFMA side, its HBM side stated too
            # bytes this kernel MUST move: ids + table rows + first-order weights + dense features in, layer output + sum_x + fm_logit
            # out (+ the gathered embeddings only when `concat` is still stored, DR_NO_CONCAT=0).  SURVEY section 8(d)'s K3 figure
            # (8FD + 12F + 8 per example) includes the 4FD concat WRITE, which the no-concat kernel does not perform: kept beside it,
            # labelled, not used for the fraction.
            Fq, Dq, N0 = a.fields, a.dim, int(dnn_units[0]) if dnn_units else 1
            stores_concat = not getattr(eng, "no_concat", False)
            must = a.batch * (8 * Fq + 4 * Fq * Dq + 4 * Fq + 4 * a.dense + 4 * N0 + 4 * Dq + 4 + (4 * Fq * Dq if stores_concat else 0))
            sv = eng.alg_bytes_fwd() if hasattr(eng, "alg_bytes_fwd") else None
            row["hbm_side"] = {"bytes_that_must_move": int(must), "GBps": round(must / sec / 1e9, 1), "frac_of_hbm_peak": round(must / sec / 1e9 / HBM_PEAK_GBS, 4),
                               "concat_stored": stores_concat,
                               "survey_8d_k3_bytes_plus_output": int(sv + 4.0 * a.batch * N0) if sv else None,
                               "note": "K3 (gather + first-order + FM) runs inside this GEMM; the fraction is on the bytes the kernel must move, "
                                       "the section-8(d) figure beside it still counts a concat write this kernel no longer does"}
        if name == "emb_pool_bwd_fused_dgrad_L0":
            # the first layer's dgrad with K4's unique-row pass as its epilogue: priced on the bytes it must move; its matrix side stated too
            Fq, Dq, N0 = a.fields, a.dim, int(dnn_units[0]) if dnn_units else 1
            fl = 2.0 * a.batch * Fq * Dq * N0
            row["mfma_side"] = {"flops": fl, "TFLOPs": round(fl / sec / 1e12, 1), "frac_of_f16x2_ceiling": round(fl / sec / 1e12 / MFMA_F16X2_PEAK_TF, 4)}
            row["note"] = ("dx = dy W^T for the 64 F embedding columns (f16x2 split) + K4's SGD update of every table row unique in the batch, applied from "
                           "the accumulators: bytes = SURVEY 8(d)'s K4 figure 12FD + 16F minus the 4FD read of d_concat, which no longer exists, plus the "
                           "read of dy; the rows several slots share go through emb_pool_bwd_dups.  DR_FUSE_K4=0 runs linear_bwd_dx_L0 + emb_pool_bwd instead "
                           "(roofline_bwd_bare)")
        if k.get("note"):
            row["note"] = k["note"]
        if k.get("concurrent_with"):
            row["concurrent_with"] = k["concurrent_with"]     # runs on a second stream next to that kernel: its event time
                                                              # (and so `achieved`) is stretched by the sharing -- see roofline_pairs
        roof_all.append(row)
    roof_all.sort(key=lambda r: -r["avg_us"])
    # kernels that run concurrently on two streams are priced as a PAIR: joint wall time (first start -> last end) against the
    # sum of the two kernels' roofline times
    roof_pairs = []
    for pr in pairs:
        ideal_us = 0.0
        for nm_ in pr["kernels"]:
            k = kernels[nm_]
            if k["bound"] == "hbm":
                ideal_us += k["alg_bytes"] / (HBM_PEAK_GBS * 1e9) * 1e6
            else:
                pk = MFMA_F16X2_PEAK_TF if (a.gemm == "bf16x3" and getattr(eng, "h2", False) and nm_ in H2_KERNELS) else \
                    (MFMA_BF16X3_PEAK_TF if a.gemm == "bf16x3" else MFMA_F32_PEAK_TF)
                ideal_us += k["alg_flops"] / (pk * 1e12) * 1e6
        roof_pairs.append({"kernels": pr["kernels"], "joint_us": round(pr["joint_us"], 2), "roofline_sum_us": round(ideal_us, 2),
                           "frac": round(ideal_us / pr["joint_us"], 4)})
    solo = [r for r in roof_all if "concurrent_with" not in r]
    roofline = dict((solo or roof_all)[0]) if roof_all else None
    roofline_mfma = None
    if roofline is not None:
        roofline["selection"] = "longest kernel of the timed region that has the chip to itself; concurrent kernels are priced jointly in roofline_pairs"
    if a.model == "deepfm" and roof_all:
        # BASELINE.json's metric is "... % HBM roofline" of the fused gather+pool forward / backward: the headline object is the
        # HBM-bound embedding kernel of the step (K4, the transposed scatter-add backward; K3 runs INSIDE the first GEMM in the
        # default configuration and is priced in that row's note).  The longest MFMA-bound kernel stays next to it.
        hbm_rows = [r for r in (solo or roof_all) if r["bound"] == "hbm" and r["kernel"].startswith(("emb_pool_bwd", "emb_pool_fwd"))]
        mfma_rows = [r for r in (solo or roof_all) if r["bound"] == "mfma"]
        if hbm_rows:
            roofline = dict(hbm_rows[0])
            roofline["selection"] = ("the HBM-bound embedding kernel the metric names (longest of K3 / K4 in the timed region); the "
                                     "longest MFMA-bound kernel is in roofline_mfma, every kernel in roofline_all")
            # its ordering step (the slot plan) runs on a side stream beside the PREVIOUS step's K4: this row's event time already
            # contains what that sharing costs K4; charged a second way below -- plan event time added to the kernel's own
            plan = [v for k, v in overlapped.items() if "sort_slots" in k]        # (round 6: the event spans K1 + the plan, one chain)
            if plan:
                pu = plan[0]["event_us_while_overlapped"]
                roofline["plan_event_us_while_overlapped"] = pu
                roofline["frac_with_plan_charged"] = round(roofline["achieved"] * roofline["avg_us"] / (roofline["avg_us"] + pu) / roofline["peak"], 4)
            if k4_alone_us is not None and roofline["kernel"] == "emb_pool_bwd":
                roofline["kernel_alone_us"] = round(k4_alone_us, 2)
                roofline["frac_kernel_alone"] = round(roofline["achieved"] * roofline["avg_us"] / k4_alone_us / roofline["peak"], 4)
            if plan_alone_us is not None and roofline["kernel"].startswith("emb_pool_bwd"):
                roofline["plan_alone_us"] = round(plan_alone_us, 2)
                roofline["frac_with_plan_alone_charged"] = round(
                    roofline["achieved"] * roofline["avg_us"] / (roofline["avg_us"] + plan_alone_us) / roofline["peak"], 4)
        if mfma_rows:
            roofline_mfma = dict(mfma_rows[0])
    roofline_fwd_bare = None
    if k3_alone_us is not None:
        fb = float(eng.alg_bytes_fwd())
        ach = fb / (k3_alone_us * 1e-6) / 1e9
        roofline_fwd_bare = {"kernel": "emb_pool_fwd", "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(ach / HBM_PEAK_GBS, 4), "avg_us": round(k3_alone_us, 2), "launches": 10,
                             "alg_bytes": int(fb), "traffic": traffic.get("emb_pool_fwd"),
                             "scope": "the bare gather + pool forward (dr_emb_pool_fwd: K3 + first-order + FM terms, concat stored; SURVEY 8(d) "
                                      "bytes 8FD + 12F + 8 per example) on the last batch, ALONE after the timed region -- in the timed step K3 runs "
                                      "inside emb_linear_fwd_L0 (see that row's hbm_side); DR_FUSE_K3=0 puts this kernel into the step"}
    roofline_bwd_bare = None
    if k4_alone_us is not None and getattr(eng, "fuse_k4", False):
        bb = float(eng.alg_bytes_bwd())
        ach = bb / (k4_alone_us * 1e-6) / 1e9
        roofline_bwd_bare = {"kernel": "emb_pool_bwd", "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(ach / HBM_PEAK_GBS, 4), "avg_us": round(k4_alone_us, 2), "launches": 10, "alg_bytes": int(bb),
                             "traffic": traffic.get("emb_pool_bwd"),
                             "scope": "K4 as a kernel of its own (dr_emb_pool_bwd_sorted_ex, SURVEY 8(d) bytes 12FD + 16F per example: it reads the gradient "
                                      "rows back from d_concat) on the last batch, ALONE after the timed region -- in the timed step its unique-row pass is "
                                      "the epilogue of the first-layer dgrad (emb_pool_bwd_fused_dgrad_L0), which never writes those rows; DR_FUSE_K4=0 puts "
                                      "this kernel into the step"}
    # measured copy ceiling next to the spec peak: the library's own streaming copy (dr_copy_nt: 16-byte
    # nontemporal loads / stores, the guide's "float4 copy") over 1 GiB, read + write bytes.  (Rounds 1-4 timed a torch copy here,
    # which K4 exceeded: not a ceiling.)
    copy_gbs = None
    try:
        if world == 1 or not a.share_device:
            src_c = torch.empty(1 << 28, dtype=torch.float32, device=device)
            dst_c = torch.empty_like(src_c)
            src_c.zero_()
            for _ in range(3):
                dr_ops.copy_nt(src_c, dst_c)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                dr_ops.copy_nt(src_c, dst_c)
            e1.record()
            torch.cuda.synchronize()
            copy_gbs = 2.0 * src_c.numel() * 4 / (e0.elapsed_time(e1) / 10 * 1e-3) / 1e9
            del src_c, dst_c
    except Exception:
        copy_gbs = None
    if roofline is not None:
        roofline["traffic_source"] = traffic_source
        if traffic_source is None:
            roofline["traffic_note"] = traffic_note
        roofline["peak_basis"] = roofline.get("peak_basis") or "HBM3E spec peak 8.0 TB/s (MI355X_MICROARCH.md)"
        if copy_gbs is not None and roofline.get("unit") == "GB/s":
            roofline["measured_copy_ceiling_GBps"] = round(copy_gbs, 1)      # dr_copy_nt over 1 GiB on this box, read + write bytes
            roofline["frac_of_measured_copy_ceiling"] = round(roofline["achieved"] / copy_gbs, 4)
        roofline["event_scope"] = (("HIP events around each phase in %d bracketed steps run after the timed region (the timed region itself carries none)" % EV_EXTRA) if events_after else
                                   "HIP events on the launch stream around %s, every %d-th step of the timed region; an event pair also "
                                   "spans cross-stream waits queued in front of the kernel, so it reads 3-10 %% above rocprofv3's kernel-only "
                                   "duration (profiles/)" % ("this kernel (the other rows of roofline_all: %d fully bracketed steps after the timed "
                                                             "region)" % EV_EXTRA if headline_only else "each kernel", ev_every))

    if rank == 0:
        opt_s = "SGD" if a.optimizer == "sgd" else "Adam (row-wise on the tables)"
        par = "single" if world == 1 and not force_sharded else "dp%d+row-sharded-tables" % world
        where = ("on one GPU (%.1f GB)" % (a.fields * a.vocab * a.dim * 4 / 1e9) if not sharded
                 else "row-sharded over %d GPUs (id %% N), RCCL all-to-all" % world)
        if a.model == "deepfm":
            metric = "examples/sec/node DeepFM (26 sparse feats, 10M vocab, dim 64)"
            cfgname = "BASELINE.json configs[1], MovieLens-1M shape" if a.preset == "c2" else "BASELINE.json configs[2]"
            workload = ("DeepFM %s synthetic training step (%s): %d sparse + %d dense, %d vocab/field hashed on device, dim %d, DNN %s+[1] "
                        "relu, sigmoid-CE, fused %s; batch %d per GPU, ids %s; tables %s%s"
                        % ("MovieLens-shape" if a.preset == "c2" else "Criteo-shape", cfgname, a.fields, a.dense, a.vocab, a.dim, dnn_units, opt_s, a.batch,
                           a.ids, where, ", %d micro-batches per step" % getattr(eng, "mb", 1) if sharded else ""))
        elif a.model == "dcn":
            metric = "examples/sec DCN (26 sparse + 13 dense, 10M vocab, dim 64, 3 cross layers + MLP %s) -- BASELINE config 4" % dnn_units
            workload = ("DCN Criteo-shape synthetic training step (BASELINE.json configs[3]): %d sparse + %d dense, %d vocab/field hashed on "
                        "device, dim %d, 3 full-rank cross layers %dx%d + MLP %s+[1] relu, sigmoid-CE, fused SGD; batch %d per GPU, ids %s; "
                        "tables %s" % (a.fields, a.dense, a.vocab, a.dim, a.fields * a.dim + a.dense, a.fields * a.dim + a.dense, dnn_units,
                                       a.batch, a.ids, where))
        else:
            metric = "examples/sec two-tower retrieval training step (1M-item corpus, dim 128, in-batch softmax over 8192 candidates) -- BASELINE config 5"
            workload = ("DSSM-style two-tower synthetic training step (BASELINE.json configs[4]): user keys hashed on device into %d rows, %d "
                        "items, embedding dim %d, towers %s (relu hidden, linear output), Retrieval loss = in-batch softmax over the batch's "
                        "%d candidates with accidental-hit removal, CCE SUM, fused SGD; batch %d per GPU; FactorizedTopK pass over the corpus "
                        "reported in metric_pass%s" % (a.users, a.items, a.dim, a.towers, a.batch * world, a.batch,
                                                       "; item table and corpus row-sharded over %d GPUs" % world if sharded else ""))
        out = {
            "metric": metric,
            "value": round(value, 1), "unit": "examples/sec", "n_gpus": dist.get_world_size() if (dist is not None and world > 1) else 1, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "strong" if a.global_batch else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload,
                       "gemm_products": (("fp32 in / fp32 accumulate; %s: 3 fp16 MFMA products of two-term fp16 splits of x * 2^k, "
                                          "k per tensor from its amax record (f16x2 split; error vs fp64 at or below the bf16x3 split's: tests/test_gpu_h2_gemm.py, "
                                          "profiles/r05_dcn_parity_diag.log); other GEMMs: 6 bf16 MFMA products of exact 3-way bf16 splits"
                                          % ("every wide GEMM (cross layers, MLP layers of >= 128 outputs)" if getattr(eng, "h2_all_wide", False)
                                             else "first layer's three GEMMs"))
                                         if (a.gemm == "bf16x3" and getattr(eng, "h2", False)) else
                                         "fp32 in / fp32 accumulate; products = 6 bf16 MFMA products of exact 3-way bf16 splits (error vs fp64 <= native fp32 MFMA, tests/test_gpu_kernels.py)"
                                         + ("; the in-batch softmax's two 8192 x 8192 x 128 score passes and the exact top-K scan of the metric pass follow "
                                            "dr_get_gemm_split (%s: 3 fp16 MFMA products of two-term fp16 splits)" % dr_ops.get_gemm_split() if a.model == "dssm" else "")
                                         if a.gemm == "bf16x3" else "native v_mfma_f32_32x32x2_f32"),
                       "gemm_split": dr_ops.get_gemm_split() if a.gemm == "bf16x3" else None,      # what the LIBRARY reports (dr_get_gemm_split)
                       "global_batch": a.batch * world, "parallelism": par, "final_loss": round(loss, 6),
                       "per_kernel_events": prof is not None, "per_kernel_events_every_n_steps": ev_every if prof is not None else None,
                       "per_kernel_events_in_timed_region": (prof is not None and not events_after),
                       **({"per_kernel_events_scope": "the headline kernel inside the timed region; the other rows from %d fully bracketed steps after it" % EV_EXTRA}
                          if headline_only else {}),
                       "settle_steps_untimed": settle_steps},
            "roofline": roofline,
            "roofline_all": roof_all,
        }
        if roofline_mfma is not None:
            out["roofline_mfma"] = roofline_mfma
        if roofline_fwd_bare is not None:
            if copy_gbs is not None:
                roofline_fwd_bare["frac_of_measured_copy_ceiling"] = round(roofline_fwd_bare["achieved"] / copy_gbs, 4)
            out["roofline_fwd_bare"] = roofline_fwd_bare
        if roofline_bwd_bare is not None:
            if copy_gbs is not None:
                roofline_bwd_bare["frac_of_measured_copy_ceiling"] = round(roofline_bwd_bare["achieved"] / copy_gbs, 4)
            out["roofline_bwd_bare"] = roofline_bwd_bare
        if roof_pairs:
            out["roofline_pairs"] = roof_pairs
        out.update(extra)
        if comm_phases:
            out["exchange_phases"] = comm_phases
        if exchange is not None:
            out["exchange"] = exchange
        if strong is not None:
            out["strong"] = strong
        if a.share_device:
            out["NOT_A_MEASUREMENT"] = "--share-device: all ranks on cuda:0, collectives staged through the host over gloo (data-path check only)"
        if overlapped:
            out["overlapped_side_stream"] = overlapped
        # The same job with every product formed from EXACT operand splits (six bf16 MFMA products: 24-bit operands, i.e. fp32 operand
        # arithmetic) beside the headline, whose first-layer GEMMs carry 22-bit operands (f16x2 split): a second engine, same seed, same
        # batches, as many steps as the headline engine had run when its loss was read; the last 20 of them timed.
        if (world == 1 and not force_sharded and a.model == "deepfm" and a.gemm == "bf16x3" and getattr(eng, "h2", False)
                and a.optimizer == "sgd" and os.environ.get("DR_BENCH_STRICT", "1") == "1"):
            prev_split = dr_ops.set_gemm_split("bf16x3")
            try:
                from deep_recommenders_amd.engine import DeepFMEngine
                eng2 = DeepFMEngine(a.fields, a.vocab, a.dim, dnn_units, a.batch, num_dense=a.dense, lr=a.lr, device=device, optimizer=a.optimizer)
                step2 = lambda i: eng2.train_step(*batches[i % nb], next_keys=batches[(i + 1) % nb][0], next_dense=batches[(i + 1) % nb][1])
                total, n_t = a.warmup + settle_steps + a.steps, min(20, a.steps)
                for i in range(total - n_t):
                    step2(i)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for i in range(total - n_t, total):
                    step2(i)
                torch.cuda.synchronize()
                ms2 = (time.perf_counter() - t1) / n_t * 1e3
                loss2 = float(eng2.loss.item())
                out["strict_fp32"] = {"gemm_split": "bf16x3", "ms_per_step": round(ms2, 4), "steps_timed": n_t, "steps_run": total,
                                      "final_loss": round(loss2, 6), "headline_final_loss": round(loss, 6),
                                      "rel_diff_of_final_loss": float("%.3g" % (abs(loss2 - loss) / max(abs(loss2), 1e-30))),
                                      "note": "same seed, batches and step count as the headline engine; every GEMM product from exact three-way bf16 splits "
                                              "(fp32 operand arithmetic, fp32 accumulate) -- the headline's first-layer GEMMs use the f16x2 split (22-bit operands)"}
                del eng2, step2
            except Exception as e:
                out["strict_fp32"] = {"failed": repr(e)}
            finally:
                dr_ops.set_gemm_split(prev_split)
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
