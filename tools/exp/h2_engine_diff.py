"""EXPERIMENT driver: the DeepFM engine in the f16x2 mode against the bf16x3 mode, tensor by tensor, at config 3's batch (V = 1 M)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from deep_recommenders_amd.engine import DeepFMEngine
F, D, ND, B, DNN, V = 26, 64, 13, 65536, [256, 32], 1_000_000
g = torch.Generator(device="cuda"); g.manual_seed(1234)
batches = []
for _ in range(2):
    keys = torch.randint(0, 10**16, (B, F), device="cuda", generator=g)
    dense = torch.log1p(torch.randn((B, ND), device="cuda", generator=g).abs())
    labels = (torch.rand(B, device="cuda", generator=g) < 0.25).float()
    batches.append((keys, dense, labels))


def mk(split):
    os.environ["DR_GEMM_SPLIT"] = split
    return DeepFMEngine(F, V, D, DNN, B, num_dense=ND, lr=1.0, seed=42, lin_init_std=0.01)


def cmp(name, a, b):
    a, b = a.double(), b.double()
    d = (a - b).abs()
    print("  %-10s max|d| %.3e  max|b| %.3e  rms(d)/rms(b) %.3e  frac(|d| > 1e-4 rms b) %.2e" % (
        name, d.max().item(), b.abs().max().item(), (d.pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item(),
        (d > 1e-4 * b.pow(2).mean().sqrt()).double().mean().item()), flush=True)


e2, e3 = mk("f16x2"), mk("bf16x3")
assert e2.h2 and not e3.h2
for i, (k, d, l) in enumerate(batches):
    nk, nd = (batches[i + 1][0], batches[i + 1][1]) if i + 1 < len(batches) else (None, None)
    w2, w3 = e2.Ws[0].clone(), e3.Ws[0].clone()
    l2 = e2.train_step(k, d, l, next_keys=nk, next_dense=nd).item()
    l3 = e3.train_step(k, d, l, next_keys=nk, next_dense=nd).item()
    torch.cuda.synchronize()
    print("step %d loss %.9f %.9f  records: table %.4g dense %.4g dh0 %.4g W0 %.4g" % (
        i, l2, l3, e2.tab_amax.view(torch.float32).item(), e2.dense_amax.view(torch.float32).item(), e2.dh0_amax.view(torch.float32).item(),
        e2.wplanes[0].amax.view(torch.float32).item()))
    print("   true: dense %.4g dh0 %.4g W0(now) %.4g" % (e2.dense_pad.abs().max().item(), e2.dhs[0].abs().max().item(), e2.Ws[0].abs().max().item()))
    cmp("h0", e2.hs[0], e3.hs[0])
    cmp("dh0", e2.dhs[0], e3.dhs[0])
    cmp("d_concat", e2.d_concat[:, :e2.in_dim], e3.d_concat[:, :e3.in_dim])
    cmp("dW0", e2.Ws[0] - w2, e3.Ws[0] - w3)
    cmp("dW0[emb]", (e2.Ws[0] - w2)[:1664], (e3.Ws[0] - w3)[:1664])
    cmp("dW0[dense]", (e2.Ws[0] - w2)[1664:], (e3.Ws[0] - w3)[1664:])
    cmp("W1", e2.Ws[1], e3.Ws[1])
    # the wgrad alone on identical inputs: e3's tensors through both kernels
    from deep_recommenders_amd import ops
    dy = e3.dhs[0]
    da, db_ = torch.zeros_like(e3.Ws[0]), torch.zeros_like(e3.Ws[0])
    ops.bf3_wgrad_emb(e3._ids_t[e3.cur], e3.row_base, e3.table, e3.dense_pad, dy, 1.0, da)
    ops.h2_wgrad_emb(e3._ids_t[e3.cur], e3.row_base, e3.table, ops.h2_amax(e3.table), e3.dense_pad, ops.h2_amax(e3.dense_pad), dy, ops.h2_amax(dy), 1.0, db_)
    cmp("wgrad same-in", db_, da)
