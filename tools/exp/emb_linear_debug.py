"""Where does dr_bf3_emb_linear_fwd differ from dr_emb_pool_fwd + dr_bf3_linear_nt?  (debug aid for the fused first layer)"""
import sys
import torch
from deep_recommenders_amd import ops


def run(M, F, Nd, N, miss=0.05):
    g = torch.Generator(device="cuda").manual_seed(M + F)
    D, V = 64, 97
    table = torch.randn((F * V, D), device="cuda", generator=g) * 0.3
    lin_w = torch.randn((F * V,), device="cuda", generator=g)
    lin_b = torch.tensor([0.37], device="cuda")
    row_base = (torch.arange(F, device="cuda") * V).to(torch.int64)
    ids = torch.randint(0, V, (M, F), device="cuda", generator=g)
    if miss:
        ids[torch.rand((M, F), device="cuda", generator=g) < miss] = -1
    K = F * D + Nd
    ld = (K + 3) // 4 * 4
    dense = torch.randn((M, Nd), device="cuda", generator=g)
    W = torch.randn((K, N), device="cuda", generator=g) * 0.1
    b = torch.randn((N,), device="cuda", generator=g)
    wp = ops.WeightPlanes(W)
    concat0 = torch.zeros((M, ld), device="cuda")
    sum0, fm0 = torch.empty((M, D), device="cuda"), torch.empty((M,), device="cuda")
    ops.emb_pool_fwd(ids, F, None, row_base, table, lin_w, lin_b, ld_concat=ld, concat=concat0, sum_x=sum0, fm_logit=fm0)
    concat0[:, F * D:K] = dense
    y0 = ops.bf3_linear_nt(concat0[:, :K], wp.wt, bias=b, act=1)
    concat1 = torch.zeros((M, ld), device="cuda")
    concat1[:, :F * D] = float("nan")
    concat1[:, F * D:K] = dense
    sum1, fm1 = torch.full((M, D), float("nan"), device="cuda"), torch.full((M,), float("nan"), device="cuda")
    y1 = torch.full((M, N), float("nan"), device="cuda")
    dpad = None
    if Nd:
        dpad = torch.zeros((M, 32), device="cuda")
        dpad[:, :Nd] = dense
    ops.bf3_emb_linear_fwd(ids, row_base, V, table, lin_w, lin_b, dpad, concat1, K, wp.wt, b, 1, sum1, fm1, y1)
    torch.cuda.synchronize()
    print("M %d F %d Nd %d N %d miss %.2f" % (M, F, Nd, N, miss))
    for name, a, r in (("concat", concat1, concat0), ("sum_x", sum1, sum0), ("fm", fm1.view(-1, 1), fm0.view(-1, 1)), ("y", y1, y0)):
        bad = ~((a == r) | ((a - r).abs() <= 1e-4 * (1 + r.abs())))
        nb = int(bad.sum())
        print("  %-7s mismatches %d of %d  nan %d" % (name, nb, a.numel(), int(torch.isnan(a).sum())))
        if nb:
            idx = bad.nonzero()[:6].tolist()
            rows = sorted(set(bad.nonzero()[:, 0].tolist()))
            cols = sorted(set(bad.nonzero()[:, 1].tolist()))
            print("     first", idx, "rows", rows[:12], "n_rows", len(rows), "cols", cols[:12], "n_cols", len(cols))
            if name == "concat":
                r0, c0 = idx[0]
                print("     got", a[r0, c0:c0 + 4].tolist(), "want", r[r0, c0:c0 + 4].tolist(), "id", ids[r0, min(c0 // 64, F - 1)].item())


if __name__ == "__main__":
    run(300, 3, 0, 40, miss=0.0)
    run(300, 3, 0, 40)
    run(257, 1, 2, 64)
    run(2085, 26, 13, 256)


def big():
    """slab larger than 4 GB: rows past byte offset 2^32 must be reachable"""
    R, F, M, N = 20_000_000, 2, 1024, 64
    table = torch.empty((R, 64), device="cuda")
    table.copy_(torch.arange(R, device="cuda", dtype=torch.float32).view(-1, 1).expand(R, 64) * 1e-6)
    row_base = torch.tensor([0, R // 2], device="cuda", dtype=torch.int64)
    g = torch.Generator(device="cuda").manual_seed(1)
    ids = torch.randint(0, R // 2, (M, F), device="cuda", generator=g)
    K = 128
    W = torch.randn((K, N), device="cuda", generator=g) * 0.1
    wp = ops.WeightPlanes(W)
    b = torch.zeros((N,), device="cuda")
    concat = torch.zeros((M, K), device="cuda")
    sx, fm, y = torch.empty((M, 64), device="cuda"), torch.empty((M,), device="cuda"), torch.empty((M, N), device="cuda")
    ops.bf3_emb_linear_fwd(ids, row_base, R // 2, table, None, None, None, concat, K, wp.wt, b, 0, sx, fm, y)
    want = torch.cat([table[ids[:, 0]], table[ids[:, 1] + R // 2]], dim=1)
    bad = (concat != want)
    print("big slab: mismatching rows", int(bad.any(dim=1).sum()), "of", M, "| first field", int(bad[:, :64].any(dim=1).sum()), "second", int(bad[:, 64:].any(dim=1).sum()))
    if bad.any():
        r = bad.any(dim=1).nonzero()[0].item()
        print("   row", r, "ids", ids[r].tolist(), "got", concat[r, 0].item() * 1e6, concat[r, 64].item() * 1e6)


if __name__ == "__main__":
    big()
