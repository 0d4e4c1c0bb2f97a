"""world_size-2 `gloo` tests (CPU) of the N>1 exchange logic in deep_recommenders_amd/sharded.py.

The HIP kernels cannot run here, so the engine is driven with an oracle-backed `prims` object (NumPy /
torch-CPU restatements of each primitive, tests-only): what is under test is the bucketing / all-to-all /
inverse-permutation / gradient-return / all-reduce plan.  The result of one sharded training step on two
ranks must equal one single-process oracle step on the concatenated global batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import tf_semantics as O
from oracle import torch_ref as T


class OraclePrims:
    @staticmethod
    def hash_bucket_i64(keys, col_buckets, out=None):
        k = keys.numpy()
        res = np.stack([O.hash_bucket_i64(k[:, c], int(col_buckets[c])) for c in range(k.shape[1])], axis=1)
        res = torch.from_numpy(res)
        if out is not None:
            out.copy_(res)
            return out
        return res

    @staticmethod
    def shard_bucket_ids(ids, rows_per_shard, world, rep=None):
        B, C = ids.shape
        flat = ids.reshape(-1).numpy()
        p = np.arange(flat.size)
        owner = np.where(flat >= 0, flat % world, p % world)
        local = np.where(flat >= 0, (p % C) * rows_per_shard + flat // world, -1)
        if rep is None:
            order = np.argsort(owner, kind="stable")
            pos = np.empty_like(order)
            pos[order] = np.arange(order.size)
            counts = np.bincount(owner, minlength=world).astype(np.int64)
            return torch.from_numpy(counts), torch.from_numpy(local[order].astype(np.int64)), \
                torch.from_numpy(pos.astype(np.int64)).reshape(B, C)
        # de-duplicated: only the representatives get a send slot, every other slot shares its representative's position
        rep = rep.numpy()
        is_rep = rep == p
        keep = p[is_rep]
        order = keep[np.argsort(owner[keep], kind="stable")]
        pos = np.full(flat.size, -1, dtype=np.int64)
        pos[order] = np.arange(order.size)
        pos = pos[rep]
        counts = np.bincount(owner[keep], minlength=world).astype(np.int64)
        send = np.full(flat.size, -7, dtype=np.int64)          # (entries past the distinct rows are never sent)
        send[:order.size] = local[order]
        return torch.from_numpy(counts), torch.from_numpy(send), torch.from_numpy(pos).reshape(B, C)

    @staticmethod
    def new_sort_plan(n, device):
        return "oracle-plan"

    @staticmethod
    def shard_dedup_slots(ids, row_base, num_rows, plan, out=None):
        """rep[p] = lowest slot with the same (field, id); missing ids keep their own slot; flags = 1 iff no other slot shares the row"""
        B, F = ids.shape
        rows = (ids + row_base[None, :]).reshape(-1).numpy()
        miss = ids.reshape(-1).numpy() < 0
        p = np.arange(rows.size)
        key = np.where(miss, -1 - p, rows)                       # every missing slot is its own key
        uniq, first, cnt = np.unique(key, return_index=True, return_counts=True)
        idx = np.searchsorted(uniq, key)
        rep = first[idx]
        flags = (cnt[idx] == 1).astype(np.uint8)
        return torch.from_numpy(rep.astype(np.int64)), torch.from_numpy(flags)

    @staticmethod
    def rows_gather(rows, table, lin_w=None):
        m = rows >= 0
        out = torch.zeros((rows.numel(), table.shape[1]))
        out[m] = table[rows[m]]
        lin = None
        if lin_w is not None:
            lin = torch.zeros(rows.numel())
            lin[m] = lin_w[rows[m]]
        return out, lin

    @staticmethod
    def rows_scatter_add(rows, grads, lin_grads, scale, table, lin_w):
        m = rows >= 0
        table.index_add_(0, rows[m], grads[m], alpha=scale)
        if lin_w is not None and lin_grads is not None:
            lin_w.index_add_(0, rows[m], lin_grads[m], alpha=scale)

    @staticmethod
    def emb_pool_fwd(ids, F, col_start, row_base, table, lin_w, lin_bias, ld_concat=None, concat=None, sum_x=None,
                     fm_logit=None, want_sum_x=True, want_fm=True, **kw):
        cs = list(range(F + 1)) if col_start is None else col_start.tolist()
        if not want_fm:                                    # gather + pool only (DCN: no first-order / FM term)
            embs = T.pool_fields(table, ids, cs, row_base.tolist())
            c = torch.cat(embs, 1)
            concat[:, :c.shape[1]].copy_(c)
            return concat, None, None
        c, s, l = T.emb_fm_forward(table, lin_w, lin_bias[0] if lin_bias is not None else 0.0, ids, cs, row_base.tolist())
        concat[:, :c.shape[1]].copy_(c)
        sum_x.copy_(s)
        fm_logit.copy_(l)
        return concat, sum_x, fm_logit

    @staticmethod
    def emb_pool_bwd(ids, F, col_start, row_base, D, d_concat, concat, sum_x, d_fm, scale, dst_table, dst_lin,
                     dst_bias=None):
        for f in range(F):        # single-valued fields in these tests
            rows = ids[:, f] + row_base[f]
            g = d_concat[:, f * D:(f + 1) * D] + d_fm[:, None] * (sum_x - concat[:, f * D:(f + 1) * D])
            dst_table.index_add_(0, rows, g, alpha=scale)
            if dst_lin is not None:
                dst_lin.index_add_(0, rows, d_fm, alpha=scale)
        if dst_bias is not None:
            dst_bias += scale * d_fm.sum()

    @staticmethod
    def emb_pack_grads(pos, D, d_concat, concat, sum_x, d_fm, out_rows, out_lin=None, bias_sum=None, unique_flags=None):
        B, F = pos.shape
        for f in range(F):
            g = d_concat[:, f * D:(f + 1) * D]
            if d_fm is not None:
                g = g + d_fm[:, None] * (sum_x - concat[:, f * D:(f + 1) * D])
            if unique_flags is not None:                     # de-duplicated: slots of a shared row accumulate (zero-filled buffers)
                out_rows.index_add_(0, pos[:, f], g)
                if out_lin is not None:
                    out_lin.index_add_(0, pos[:, f], d_fm)
                continue
            out_rows[pos[:, f]] = g
            if out_lin is not None:
                out_lin[pos[:, f]] = d_fm
        if bias_sum is not None:
            bias_sum += d_fm.sum()

    @staticmethod
    def emb_sort_slots(ids, row_base, num_rows, plan=None):
        return "oracle-plan"       # the oracle applies gradients with index_add_: no plan needed

    @staticmethod
    def emb_pool_bwd_sorted(ids, row_base, plan, D, num_rows, grad, d_fm, scale, dst_table, dst_lin=None, dst_bias=None,
                            concat=None, sum_x=None, slot_lin_grad=None):
        flat = ids.reshape(-1)
        m = flat >= 0
        g = grad.reshape(-1, D)
        dst_table.index_add_(0, flat[m], g[m], alpha=scale)
        if dst_lin is not None and slot_lin_grad is not None:
            dst_lin.index_add_(0, flat[m], slot_lin_grad[m], alpha=scale)

    @staticmethod
    def emb_pool_bwd_sorted_adam(ids, row_base, plan, D, num_rows, grad, d_fm, lr_t, beta1, beta2, eps, table, m_table, v_table,
                                 lin_w=None, m_lin=None, v_lin=None, concat=None, sum_x=None, slot_lin_grad=None):
        """one row-wise Adam update per touched row from the sum of its slots' gradients ([TF] B15 with lr_t given)"""
        flat = ids.reshape(-1)
        m = flat >= 0
        rows = torch.unique(flat[m])
        g = torch.zeros_like(table).index_add_(0, flat[m], grad.reshape(-1, D)[m])[rows]
        m_table[rows] = beta1 * m_table[rows] + (1 - beta1) * g
        v_table[rows] = beta2 * v_table[rows] + (1 - beta2) * g * g
        table[rows] = table[rows] - lr_t * m_table[rows] / (v_table[rows].sqrt() + eps)
        if lin_w is not None and slot_lin_grad is not None:
            gl = torch.zeros_like(lin_w).index_add_(0, flat[m], slot_lin_grad[m])[rows]
            m_lin[rows] = beta1 * m_lin[rows] + (1 - beta1) * gl
            v_lin[rows] = beta2 * v_lin[rows] + (1 - beta2) * gl * gl
            lin_w[rows] = lin_w[rows] - lr_t * m_lin[rows] / (v_lin[rows].sqrt() + eps)

    @staticmethod
    def adam_step(param, grad, m, v, lr_t, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
        g = grad * grad_scale
        m.mul_(beta1).add_(g, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        param.sub_(lr_t * m / (v.sqrt() + eps))

    @staticmethod
    def linear_fwd(x, W, b, act, out=None):
        y = x @ W + b
        if act:
            y = torch.relu(y)
        out.copy_(y)
        return out

    @staticmethod
    def linear_bwd_dx(dy, W, relu_src=None, accumulate=False, out=None):
        dx = dy @ W.t()
        if relu_src is not None:
            dx = dx * (relu_src > 0)
        if accumulate:
            out.add_(dx)
        else:
            out.copy_(dx)
        return out

    @staticmethod
    def cross_fwd(x0, x, W, b, diag_scale=0.0, want_prod=False, prod=None):
        pr = x @ W + b + diag_scale * x                     # keras/models/ranking/dcn.py:81,85-86
        return x0 * pr + x, pr                              # :88

    @staticmethod
    def cross_combine_bwd(x0, prod, d_out, diag_scale, d_x0_accum, d_x_accum):
        d_prod = d_out * x0
        d_x0_accum.add_(d_out * prod)
        if d_x_accum is not None:
            d_x_accum.add_(d_out + diag_scale * d_prod)
        return d_prod

    @staticmethod
    def linear_bwd_dw(x, dy, scale, dstW, dstb=None, workspace=None):
        dstW += scale * (x.t() @ dy)
        if dstb is not None:
            dstb += scale * dy.sum(0)

    @staticmethod
    def linear_bwd_narrow_supported(M, K, N):
        return N <= 32 and K in (128, 256, 512) and M > 0 and M % 32 == 0        # same domain as the HIP kernel

    @staticmethod
    def linear_bwd_narrow(x, dy, W, scale, dstW, dstb, dx, relu_mask=True, workspace=None):
        g = dy @ W.t()                                    # pre-update W (dstW may alias it)
        dx.copy_(g * (x > 0) if relu_mask else g)
        dstW += scale * (x.t() @ dy)
        if dstb is not None:
            dstb += scale * dy.sum(0)
        return dx

    @staticmethod
    def tower_head_fwd_bwd(x, W1, b1, W2, b2, extra_logit, labels, loss_mode, scale, act=1, h_out=None, prob=None,
                           d_logit=None, d_h=None, loss=None, workspace=None, dst_W2="inplace", dst_b2="inplace", n_total=0):
        assert loss_mode == 0
        nt = n_total if n_total else x.shape[0]
        dst_W2 = W2 if isinstance(dst_W2, str) else dst_W2
        dst_b2 = b2 if isinstance(dst_b2, str) else dst_b2
        h = x @ W1 + b1
        if act:
            h = torch.relu(h)
        z = (h @ W2)[:, 0] + b2[0] + (extra_logit if extra_logit is not None else 0.0)
        prob.copy_(torch.sigmoid(z))
        d_logit.copy_((torch.sigmoid(z) - labels) / nt)
        loss.copy_((T.sigmoid_cross_entropy(labels, z) * z.numel() / nt).reshape(1))
        d_h.copy_(d_logit[:, None] * W2[:, 0][None, :] * ((h > 0).float() if act else 1.0))
        if dst_W2 is not None:
            dst_W2 += scale * (h.t() @ d_logit[:, None])
        if dst_b2 is not None:
            dst_b2 += scale * d_logit.sum()
        return loss, prob, d_logit, d_h

    @staticmethod
    def bce_fwd_bwd(logits, labels, mode, workspace=None, logits_b=None, out=None, **kw):
        x = logits + logits_b[:, 0]
        prob, d_logit, loss = out
        prob.copy_(torch.sigmoid(x))
        d_logit.copy_((torch.sigmoid(x) - labels) / x.numel())
        loss.copy_(T.sigmoid_cross_entropy(labels, x).reshape(1))
        return loss, prob, d_logit

    @staticmethod
    def axpy(alpha, x, y):
        y.add_(x, alpha=alpha)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


CFG = dict(F=4, V=37, D=8, B=24, Nd=3, units=[16, 8], lr=0.1, key_max=10**9, steps=2)


NSTEPS = 2


def _global_problem(cfg=None):
    """`steps` global batches of world * B examples (rank r trains on rows [r*B, (r+1)*B) of each); world = cfg["world"], default 2."""
    g = torch.Generator().manual_seed(123)
    c = cfg or CFG
    Wn = c.get("world", 2)
    table = torch.randn((c["F"] * c["V"], c["D"]), generator=g) * 0.3
    lin = torch.randn(c["F"] * c["V"], generator=g) * 0.1
    batches = []
    for _ in range(c.get("steps", NSTEPS)):
        keys = torch.randint(0, c.get("key_max", 10**9), (Wn * c["B"], c["F"]), generator=g)
        keys[3, 1] = -1                            # a missing id travels through the exchange as a zero row
        dense = torch.rand((Wn * c["B"], c["Nd"]), generator=g)
        labels = (torch.rand(Wn * c["B"], generator=g) < 0.3).float()
        batches.append((keys, dense, labels))
    return table, lin, batches


def _engine_kwargs(gpu):
    """CPU: oracle-backed primitives over gloo.  gpu=True (tests/test_gpu_sharded_two_rank.py): the HIP kernels, both ranks on
    cuda:0, device buffers staged through the host over the same gloo group (RCCL refuses two ranks on one device)."""
    if not gpu:
        return dict(device="cpu", prims=OraclePrims)
    from deep_recommenders_amd.sharded import HostStagedTransport
    torch.cuda.set_device(0)
    return dict(device="cuda", transport=HostStagedTransport())


def _worker(rank, world, port, outdir, micro_batches, optimizer="sgd", units=None, gpu=False, cfg=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from deep_recommenders_amd.sharded import ShardedDeepFMEngine
        c = cfg or CFG
        table, lin, batches = _global_problem(c)
        kw = _engine_kwargs(gpu)
        dev = kw["device"]
        eng = ShardedDeepFMEngine(c["F"], c["V"], c["D"], units or c["units"], c["B"], num_dense=c["Nd"], lr=c["lr"],
                                  world=world, rank=rank, seed=5, init_tables=(table, lin),
                                  micro_batches=micro_batches, optimizer=optimizer, dedup=bool(c.get("dedup", False)), **kw)
        assert eng.ex.dedup == bool(c.get("dedup", False))
        assert eng.mb == (1 if (optimizer == "adam" or not eng.fuse_head) else micro_batches)
        assert eng.fuse_head == (units is None)         # the [16, 40] tower of the unfused-head test cannot take the fused head
        if gpu and c.get("expect_fused_l0") is not None:
            assert bool(eng.fuse_k3) == c["expect_fused_l0"]
        sl = slice(rank * c["B"], (rank + 1) * c["B"])
        Ws0 = [w.cpu().clone() for w in eng.Ws]
        bs0 = [b.cpu().clone() for b in eng.bs]
        local = [(k[sl].contiguous().to(dev), d[sl].contiguous().to(dev), l[sl].contiguous().to(dev)) for k, d, l in batches]
        losses = []
        for t, (k, d, l) in enumerate(local):       # every step but the last hands over the next batch's keys (route prefetch)
            nk = local[t + 1][0] if t + 1 < len(local) else None
            losses.append(eng.train_step(k, d, l, next_keys=nk).item())
        torch.save((rank, losses, eng.table.cpu().clone(), eng.lin_w.cpu().clone(), [w.cpu().clone() for w in eng.Ws],
                    [b.cpu().clone() for b in eng.bs], eng.lin_bias.cpu().clone(), Ws0, bs0), os.path.join(outdir, "rank%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def _spawn(worker, args, tmp_path, world=2, timeout=240):
    """runs `worker(rank, world, port, outdir, *args)` on `world` spawned processes, returns {rank: saved tuple[1:]}"""
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=worker, args=(r, world, port, str(tmp_path)) + tuple(args)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout)
        assert p.exitcode == 0
    res = {}
    for r in range(world):
        item = torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r))
        res[item[0]] = item[1:]
    return res


def _reference_deepfm(cfg, Ws0, bs0, optimizer="sgd", dtype=torch.float32):
    """Single-process oracle: plain SGD (or row-wise Adam, [TF] B15 -- oracle/torch_ref.py adam_rows_step / adam_dense_step) on the
    GLOBAL batches.  Returns (per-step losses, table, lin, bias, Ws, bs) after the last step."""
    c = cfg
    table, lin, batches = _global_problem(c)
    F, V, lr = c["F"], c["V"], c["lr"]
    tab, li, bias = table.to(dtype), lin.to(dtype), torch.zeros(1, dtype=dtype)
    Wc, bc = [w.to(dtype).clone() for w in Ws0], [b.to(dtype).clone() for b in bs0]
    z = torch.zeros_like
    mt, vt, ml, vl, mb_, vb_ = z(tab), z(tab), z(li), z(li), z(bias), z(bias)
    mW, vW, mB, vB = [z(w) for w in Wc], [z(w) for w in Wc], [z(b) for b in bc], [z(b) for b in bc]
    losses = []
    for t, (keys, dense, labels) in enumerate(batches):
        ids = np.stack([O.hash_bucket_i64(keys[:, f].numpy(), V) for f in range(F)], axis=1)
        tt, tl, tb = tab.clone().requires_grad_(True), li.clone().requires_grad_(True), bias.clone().requires_grad_(True)
        Ws = [w.clone().requires_grad_(True) for w in Wc]
        bs = [b.clone().requires_grad_(True) for b in bc]
        logit = T.deepfm_logit(tt, tl, tb, torch.tensor(ids), list(range(F + 1)), [f * V for f in range(F)], Ws, bs, dense.to(dtype))
        lo = T.sigmoid_cross_entropy(labels.to(dtype), logit)
        lo.backward()
        losses.append(lo.item())
        if optimizer == "sgd":
            tab, li, bias = (tab - lr * tt.grad).detach(), (li - lr * tl.grad).detach(), (bias - lr * tb.grad).detach()
            Wc = [(w - lr * g.grad).detach() for w, g in zip(Wc, Ws)]
            bc = [(b - lr * g.grad).detach() for b, g in zip(bc, bs)]
        else:
            rows = torch.tensor(ids + np.arange(F)[None, :] * V).reshape(-1)
            rows = rows[torch.tensor(ids.reshape(-1) >= 0)]
            T.adam_rows_step(tab, tt.grad, rows, mt, vt, lr, t + 1)
            T.adam_rows_step(li, tl.grad, rows, ml, vl, lr, t + 1)
            T.adam_dense_step(bias, tb.grad, mb_, vb_, lr, t + 1)
            for i in range(len(Wc)):
                T.adam_dense_step(Wc[i], Ws[i].grad, mW[i], vW[i], lr, t + 1)
                T.adam_dense_step(bc[i], bs[i].grad, mB[i], vB[i], lr, t + 1)
    return losses, tab, li, bias, Wc, bc


def _shard_views(cfg, world, r, tab_r, lin_r, tab, li):
    """(got, want) pairs of rank r's table / first-order shard against the global oracle tables (row id -> rank id % world)"""
    F, V = cfg["F"], cfg["V"]
    rps = (V + world - 1) // world
    out = []
    for f in range(F):
        gid = torch.arange(r, V, world)
        out.append((tab_r[f * rps:f * rps + len(gid)], tab[f * V + gid]))
        out.append((lin_r[f * rps:f * rps + len(gid)], li[f * V + gid]))
    return out


@pytest.mark.timeout(300)
@pytest.mark.parametrize("micro_batches", [1, 2])
def test_two_rank_sharded_steps_equal_single_process_oracle(tmp_path, micro_batches):
    """2 ranks x NSTEPS steps (prefetched routes, micro-batches) == plain SGD on the global batches in one process."""
    world = 2
    res = _spawn(_worker, (micro_batches,), tmp_path)
    c = CFG
    Ws0, bs0 = res[0][6], res[0][7]
    for a, b in zip(Ws0, res[1][6]):
        assert torch.equal(a, b)                    # replicas start identical
    losses, tab, li, bias, Wc, bc = _reference_deepfm(c, Ws0, bs0, "sgd")
    for t, lo in enumerate(losses):                 # global loss = mean of the per-rank means (equal batch sizes)
        assert abs(0.5 * (res[0][0][t] + res[1][0][t]) - lo) < 2e-6
    for r in range(world):
        _, tab_r, lin_r, Ws_r, bs_r, bias_r, _, _ = res[r]
        for got, want in _shard_views(c, world, r, tab_r, lin_r, tab, li):
            np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=2e-5, atol=2e-6)
        for i in range(len(Wc)):
            np.testing.assert_allclose(Ws_r[i].numpy(), Wc[i].numpy(), rtol=2e-5, atol=2e-6)
            np.testing.assert_allclose(bs_r[i].numpy(), bc[i].numpy(), rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(bias_r.numpy(), bias.numpy(), rtol=2e-5, atol=2e-7)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("micro_batches,optimizer", [(1, "sgd"), (2, "sgd"), (1, "adam")])
def test_two_rank_sharded_dedup_exchange_equals_single_process_oracle(tmp_path, micro_batches, optimizer):
    """Requester-side de-duplication (dedup=True; [TF] `unique` inside the lookup, keras/models/ranking/fm.py:57-61): rows that several
    slots of a micro-batch share travel once each way and their gradients are summed before they are sent.  Same assertion as the plain
    exchange -- the result equals the single-process oracle on the global batches -- on a problem where sharing is the rule (24
    examples x 4 fields over 37 ids per field), and the wire really carries fewer rows."""
    world = 2
    cfg = dict(CFG, dedup=True)
    res = _spawn(_worker, (micro_batches, optimizer, None, False, cfg), tmp_path)
    Ws0, bs0 = res[0][6], res[0][7]
    losses, tab, li, bias, Wc, bc = _reference_deepfm(cfg, Ws0, bs0, optimizer)
    for t, lo in enumerate(losses):
        assert abs(0.5 * (res[0][0][t] + res[1][0][t]) - lo) < 2e-6
    tol = dict(rtol=2e-5, atol=2e-6) if optimizer == "sgd" else dict(rtol=0, atol=2e-3 * cfg["lr"])
    for r in range(world):
        _, tab_r, lin_r, Ws_r, bs_r, bias_r, _, _ = res[r]
        for got, want in _shard_views(cfg, world, r, tab_r, lin_r, tab, li):
            np.testing.assert_allclose(got.numpy(), want.numpy(), **tol)
        for i in range(len(Wc)):
            np.testing.assert_allclose(Ws_r[i].numpy(), Wc[i].numpy(), **tol)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("dedup", [False, True])
def test_four_rank_sharded_steps_equal_single_process_oracle(tmp_path, dedup):
    """The same assertion on FOUR ranks (every earlier multi-process test has world 2): owner = id % 4, uneven split sizes in every
    all-to-all, ranks 2 and 3 as both requesters and owners, two micro-batches with prefetched routes; with and without the
    de-duplicated exchange."""
    world = 4
    cfg = dict(CFG, world=world, dedup=dedup)
    res = _spawn(_worker, (2, "sgd", None, False, cfg), tmp_path, world=world)
    Ws0, bs0 = res[0][6], res[0][7]
    losses, tab, li, bias, Wc, bc = _reference_deepfm(cfg, Ws0, bs0, "sgd")
    for t, lo in enumerate(losses):
        assert abs(sum(res[r][0][t] for r in range(world)) / world - lo) < 2e-6
    for r in range(world):
        _, tab_r, lin_r, Ws_r, bs_r, bias_r, _, _ = res[r]
        for got, want in _shard_views(cfg, world, r, tab_r, lin_r, tab, li):
            np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=2e-5, atol=2e-6)
        for i in range(len(Wc)):
            np.testing.assert_allclose(Ws_r[i].numpy(), Wc[i].numpy(), rtol=2e-5, atol=2e-6)
            np.testing.assert_allclose(bs_r[i].numpy(), bc[i].numpy(), rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(bias_r.numpy(), bias.numpy(), rtol=2e-5, atol=2e-7)


def test_dedup_bucketing_sends_each_distinct_row_once():
    """The representative map and the de-duplicated bucketing of the oracle-backed primitives (the HIP kernels are checked against
    exactly these in tests/test_gpu_kernels.py): counts sum to the number of distinct rows (+ one slot per missing id), every slot's
    position points at its row, representatives are the lowest slots."""
    g = torch.Generator().manual_seed(4)
    B, F, V, world = 40, 3, 11, 2
    ids = torch.randint(0, V, (B, F), generator=g)
    ids[2, 1] = -1
    ids[7, 1] = -1
    base = torch.arange(F) * V
    rep, flags = OraclePrims.shard_dedup_slots(ids, base, F * V, None)
    flat = (ids + base[None, :]).reshape(-1)
    for p_ in range(B * F):
        r = int(rep[p_])
        if ids.reshape(-1)[p_] < 0:
            assert r == p_ and flags[p_] == 1
            continue
        assert r <= p_ and flat[r] == flat[p_] and not bool((flat[:r] == flat[p_]).any())
        assert bool(flags[p_]) == (int((flat == flat[p_]).sum()) == 1)
    rps = (V + world - 1) // world
    counts, send, pos = OraclePrims.shard_bucket_ids(ids, rps, world, rep=rep)
    n_distinct = len(set(flat[ids.reshape(-1) >= 0].tolist())) + 2
    assert int(counts.sum()) == n_distinct
    cflat, pflat = ids.reshape(-1), pos.reshape(-1)
    for p_ in range(B * F):
        want = -1 if cflat[p_] < 0 else (p_ % F) * rps + int(cflat[p_]) // world
        assert int(send[pflat[p_]]) == want
    owner_of_pos = np.repeat(np.arange(world), counts.numpy())
    for p_ in range(B * F):
        if cflat[p_] >= 0:
            assert owner_of_pos[pflat[p_]] == int(cflat[p_]) % world


@pytest.mark.timeout(300)
@pytest.mark.parametrize("units", [None, [16, 40]])
def test_two_rank_sharded_adam_steps_equal_single_process_oracle(tmp_path, units):
    """Adam in the sharded engine: 2 ranks x NSTEPS steps == row-wise Adam on the global batches in one process
    (oracle/torch_ref.py adam_rows_step / adam_dense_step, [TF] B15): every touched row gets ONE update from the gradient summed
    over both ranks' slots, the dense tower one update from the all-reduced gradient.  units=[16, 40]: a last hidden layer the
    fused head does not take (> 32 units) -- the stand-alone loss kernel normalises by the rank batch and the engine rescales
    (that branch used to leave every gradient W times too large)."""
    world = 2
    res = _spawn(_worker, (2, "adam", units), tmp_path)
    c = CFG
    lr = c["lr"]
    Ws0, bs0 = res[0][6], res[0][7]
    losses, tab, li, bias, Wc, bc = _reference_deepfm(c, Ws0, bs0, "adam")
    for t, lo in enumerate(losses):
        assert abs(0.5 * (res[0][0][t] + res[1][0][t]) - lo) < 2e-6
    # Adam divides by sqrt(v): tolerances are a fraction of one step (lr), not of the weight
    tol = dict(rtol=0, atol=2e-3 * lr)
    for r in range(world):
        _, tab_r, lin_r, Ws_r, bs_r, bias_r, _, _ = res[r]
        for got, want in _shard_views(c, world, r, tab_r, lin_r, tab, li):
            np.testing.assert_allclose(got.numpy(), want.numpy(), **tol)
        for i in range(len(Wc)):
            np.testing.assert_allclose(Ws_r[i].numpy(), Wc[i].numpy(), **tol)
            np.testing.assert_allclose(bs_r[i].numpy(), bc[i].numpy(), **tol)
        np.testing.assert_allclose(bias_r.numpy(), bias.numpy(), **tol)


DCN_CFG = dict(F=3, V=29, D=4, B=16, Nd=2, units=[8, 4], L=2, lr=0.2, diag=0.1)


def _dcn_problem(cfg=None):
    g = torch.Generator().manual_seed(9)
    c = cfg or DCN_CFG
    table = torch.randn((c["F"] * c["V"], c["D"]), generator=g) * 0.4
    batches = []
    for _ in range(NSTEPS):
        keys = torch.randint(0, 10**9, (2 * c["B"], c["F"]), generator=g)
        dense = torch.rand((2 * c["B"], c["Nd"]), generator=g)
        labels = (torch.rand(2 * c["B"], generator=g) < 0.3).float()
        batches.append((keys, dense, labels))
    return table, batches


def _dcn_worker(rank, world, port, outdir, gpu=False, cfg=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from deep_recommenders_amd.sharded import ShardedDCNEngine
        c = cfg or DCN_CFG
        table, batches = _dcn_problem(c)
        kw = _engine_kwargs(gpu)
        dev = kw["device"]
        eng = ShardedDCNEngine(c["F"], c["V"], c["D"], c["L"], c["units"], c["B"], num_dense=c["Nd"], lr=c["lr"], diag_scale=c["diag"],
                               world=world, rank=rank, seed=5, init_tables=table, **kw)
        with torch.no_grad():
            for b in eng.cross_b:
                b.fill_(0.05)
            eng.bs[-1].fill_(0.03)        # keeps logits off the kink of the restated loss at exactly 0 (dead-ReLU examples would sit
                                          # on it with a zero bias; autograd's subgradient there is not the kernels' sigmoid(0) - z)
        params0 = eng.flat_params.cpu().clone()
        sl = slice(rank * c["B"], (rank + 1) * c["B"])
        losses = [eng.train_step(k[sl].contiguous().to(dev), d[sl].contiguous().to(dev), l[sl].contiguous().to(dev)).item()
                  for k, d, l in batches]
        torch.save((rank, losses, eng.table.cpu().clone(), eng.flat_params.cpu().clone(), params0), os.path.join(outdir, "rank%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def _reference_dcn(cfg, flat0, dtype=torch.float32):
    """Single-process oracle of the DCN step (cross stack of keras/models/ranking/dcn.py:81-88 = T.cross, under torch autograd):
    plain SGD on the global batches.  Returns (losses, flat parameters, table) after the last step."""
    c = cfg
    table, batches = _dcn_problem(c)
    F, V, D, lr, L = c["F"], c["V"], c["D"], c["lr"], c["L"]
    n_in = F * D + c["Nd"]
    ld = (n_in + 3) // 4 * 4

    def unpack(P):
        off, cW, cb, Ws, bs = 0, [], [], [], []
        for _ in range(L):
            cW.append(P[off:off + n_in * ld].view(n_in, ld)[:, :n_in]); off += n_in * ld
            cb.append(P[off:off + n_in]); off += ld
        d = n_in
        for u in c["units"] + [1]:
            pu = (u + 3) // 4 * 4
            Ws.append(P[off:off + d * pu].view(d, pu)[:, :u]); off += d * pu
            bs.append(P[off:off + u]); off += pu
            d = u
        return cW, cb, Ws, bs
    flat = flat0.to(dtype).clone()
    tab = table.to(dtype).clone()
    losses = []
    for t, (keys, dense, labels) in enumerate(batches):
        ids = np.stack([O.hash_bucket_i64(keys[:, f].numpy(), V) for f in range(F)], axis=1)
        P = flat.clone().requires_grad_(True)
        tt = tab.clone().requires_grad_(True)
        cW, cb, Ws, bs = unpack(P)
        emb = T.pool_fields(tt, torch.tensor(ids), list(range(F + 1)), [f * V for f in range(F)])
        x0 = torch.cat(emb + [dense.to(dtype)], 1)
        x = x0
        for Wc, bc in zip(cW, cb):
            x = T.cross(x0, x, Wc, bc, c["diag"])
        logit = T.dnn(x, Ws, bs).squeeze(1)
        lo = T.sigmoid_cross_entropy(labels.to(dtype), logit)
        lo.backward()
        losses.append(lo.item())
        flat = (flat - lr * P.grad).detach()
        tab = (tab - lr * tt.grad).detach()
    return losses, flat, tab


@pytest.mark.timeout(300)
def test_two_rank_sharded_dcn_equals_single_process_oracle(tmp_path):
    """ShardedDCNEngine: 2 ranks x NSTEPS steps == plain SGD on the global batches in one process, with the
    cross stack of keras/models/ranking/dcn.py:81-88 (T.cross) under torch autograd."""
    world = 2
    res = _spawn(_dcn_worker, (), tmp_path)
    c = DCN_CFG
    F, V = c["F"], c["V"]
    assert torch.equal(res[0][3], res[1][3])
    losses, flat, tab = _reference_dcn(c, res[0][3])
    for t, lo in enumerate(losses):
        assert abs(0.5 * (res[0][0][t] + res[1][0][t]) - lo) < 2e-6
    rps = (V + world - 1) // world
    for r in range(world):
        _, tab_r, flat_r, _ = res[r]
        np.testing.assert_allclose(flat_r.numpy(), flat.numpy(), rtol=2e-5, atol=2e-6)
        for f in range(F):
            gid = torch.arange(r, V, world)
            np.testing.assert_allclose(tab_r[f * rps:f * rps + len(gid)].numpy(), tab[f * V + gid].numpy(), rtol=2e-5, atol=2e-6)


def test_bucketing_oracle_matches_owner_rule():
    ids = torch.tensor([[5, -1, 2], [0, 7, 9]])
    counts, send_rows, pos = OraclePrims.shard_bucket_ids(ids, 10, 2)
    assert counts.tolist() == [2, 4]
    flat_owner = [1, 1, 0, 0, 1, 1]       # 5%2, missing p=1 -> 1%2, 2%2, 0%2, 7%2, 9%2
    for p, o in enumerate(flat_owner):
        start, size = (0, 2) if o == 0 else (2, 4)
        assert start <= pos.reshape(-1)[p] < start + size
    assert send_rows[pos[0, 0]].item() == 0 * 10 + 5 // 2
    assert send_rows[pos[0, 1]].item() == -1
    assert send_rows[pos[1, 2]].item() == 2 * 10 + 9 // 2


def _amax_worker(rank, world, port, outdir, micro_batches):
    """DR_SH_TRACK_AMAX=1: the table-bound bookkeeping of the f16x2 mode (sharded.py) without its kernels"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["DR_SH_TRACK_AMAX"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from deep_recommenders_amd.sharded import ShardedDeepFMEngine
        c = dict(CFG, lr=30.0, steps=4, world=world)            # a step size that grows the tables' range every step
        table, lin, batches = _global_problem(c)
        eng = ShardedDeepFMEngine(c["F"], c["V"], c["D"], c["units"], c["B"], num_dense=c["Nd"], lr=c["lr"], world=world, rank=rank, seed=5,
                                  init_tables=(table, lin), micro_batches=micro_batches, device="cpu", prims=OraclePrims)
        assert eng.track_amax and not eng.h2 and not eng.ex.local
        sl = slice(rank * c["B"], (rank + 1) * c["B"])
        local = [(k[sl].contiguous(), d[sl].contiguous(), l[sl].contiguous()) for k, d, l in batches]
        rows = []

        def true_global():
            t = eng.table.abs().max().reshape(1).clone()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        rows.append((float(eng.tab_amax.view(torch.float32).item()), true_global(), float(eng.table.abs().max().item())))
        for t, (k, d, l) in enumerate(local):
            nk = local[t + 1][0] if t + 1 < len(local) else None
            eng.train_step(k, d, l, next_keys=nk)
            # the bound the NEXT step will read: the buffer filled behind this step's last owner-side update
            assert eng._tab_swap
            nxt = eng._tab_bufs[eng._tab_i ^ 1]
            rows.append((float(nxt.view(torch.float32).item()), true_global(), float(eng.table.abs().max().item())))
        torch.save((rank, rows), os.path.join(outdir, "rank%d.pt" % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("micro_batches", [1, 2])
def test_two_rank_table_bound_of_the_f16x2_mode_is_global_and_current(tmp_path, micro_batches):
    """The f16x2 GEMMs of the sharded engine scale the rows a rank RECEIVES by a bound over ALL shards (sharded.py: local running
    record, one all-reduce(MAX) behind the step's last owner-side update, two buffers).  With DR_SH_TRACK_AMAX=1 that bookkeeping runs
    on the CPU engine: after every step the bound the next step will read equals the largest magnitude in ANY rank's shard (here the
    records are recomputed from the shards, so equality, not just >=), is the same on both ranks, and follows a growing table."""
    res = _spawn(_amax_worker, (micro_batches,), tmp_path)
    r0, r1 = res[0][0], res[1][0]
    assert len(r0) == 5 and len(r1) == 5
    for (b0, g0, l0), (b1, g1, l1) in zip(r0, r1):
        assert b0 == b1 and g0 == g1, (r0, r1)       # one bound, on every rank
        assert b0 == g0 == max(l0, l1), (r0, r1)     # ... the largest magnitude over both shards
    assert r0[-1][0] > 1.5 * r0[0][0], (r0, r1)      # the tables' range did grow (1.18 -> 4.72 here), and the bound followed
