"""EXPERIMENT driver (not product): K4's ablation ladder (k4_ladder.hip) at config 3's shape, all variants on the same tensors in one
process.  Build first (in the container):
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -mllvm -pragma-unroll-threshold=131072 -Iinclude -shared \
        -o tools/exp/libk4_ladder.so tools/exp/k4_ladder.hip
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deep_recommenders_amd import ops
HERE = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(HERE, "libk4_ladder.so"))
p, i64, i32, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_float
L.exp_k4_m.argtypes = [i32, i32, p, p, i64, i32, p, p, i64, p, p, f32, p, p, p, p]
L.exp_k4_n.argtypes = [i32, i32, p, p, i64, i32, p, p, i64, p, p, f32, p, p, p, p, p]
L.exp_k4_s.argtypes = [i32, i32, p, p, i64, i32, p, p, i64, p, p, f32, p, p, p, p, p, p, p, i64, p, p]
dev = "cuda"
X = dict(LIN=1, FM=2, FLAGS=4, BIAS=8, NTG=16, NTS=32, REV=64, NTT=128, LINW=256, NTL=512)


def ptr(t):
    return None if t is None else t.data_ptr()


def st():
    return torch.cuda.current_stream().cuda_stream


LIN_OLD = {}


def run_variant(kind, feat, grid, ids, plan, rb, R, F, B, grad, ld, sum_x, dl, table, lin, bias, x_sorted=None, grid_d=2048, flags=None):
    flags = plan.flags if flags is None else flags
    if kind == "S":
        lin_on, fm_on, bias_on = feat & 1, feat & 2, feat & 8
        L.exp_k4_s(grid, grid_d, ptr(ids), ptr(flags), B, F, ptr(rb), ptr(grad), ld, ptr(sum_x) if fm_on else None, ptr(dl), -1e-3,
                   ptr(table), ptr(lin) if lin_on else None, ptr(bias) if bias_on else None, ptr(plan.rows), ptr(plan.slots),
                   ptr(plan.dup_heads), ptr(plan.dup_count), R, ptr(x_sorted), st())
    else:
        if kind == "M":
            rc = L.exp_k4_m(feat, grid, ptr(ids), ptr(flags), B, F, ptr(rb), ptr(grad), ld, ptr(sum_x), ptr(dl), -1e-3, ptr(table), ptr(lin),
                            ptr(bias), st())
        else:
            lo = LIN_OLD.get(B * F)
            if lo is None:
                lo = LIN_OLD[B * F] = torch.zeros(B * F, device=dev)
            rc = L.exp_k4_n(feat, grid, ptr(ids), ptr(flags), B, F, ptr(rb), ptr(grad), ld, ptr(sum_x), ptr(dl), -1e-3, ptr(table), ptr(lin),
                            ptr(bias), ptr(lo), st())
        assert rc == 0, (kind, feat)


def check_small():
    """N (all features) must equal the shipped kernel bit for bit on the rows that are unique in the batch"""
    B, F, D, V = 4096, 26, 64, 100_000
    g = torch.Generator(device=dev); g.manual_seed(3)
    R = F * V
    rb = torch.arange(F, device=dev, dtype=torch.int64) * V
    ids = torch.randint(0, V, (B, F), device=dev, generator=g)
    ids[5, 3] = -1
    plan = ops.emb_sort_slots(ids, rb, R)
    ld = 1680
    grad = torch.randn((B, ld), device=dev, generator=g) * 1e-3
    sum_x = torch.randn((B, D), device=dev, generator=g)
    dl = torch.randn(B, device=dev, generator=g) * 1e-3
    t0 = torch.randn((R, D), device=dev, generator=g) * 0.1
    l0 = torch.randn(R, device=dev, generator=g) * 0.1
    outs = {}
    xs = torch.zeros((B * F, D), device=dev)
    LIN_OLD[B * F] = l0[(ids.clamp(min=0) + rb[None, :]).reshape(-1)].contiguous()
    for kind, feat in (("S", 15), ("N", 15), ("M", 15), ("N", 63), ("N", 79), ("N", 255), ("N", 271), ("N", 511)):
        t, l, bias = t0.clone(), l0.clone(), torch.zeros(1, device=dev)
        run_variant(kind, feat, 1024, ids, plan, rb, R, F, B, grad, ld, sum_x, dl, t, l, bias, x_sorted=xs)
        torch.cuda.synchronize()
        outs[(kind, feat)] = (t, l, bias)
    del LIN_OLD[B * F]
    fl = plan.flags[:B * F].view(B, F).bool() & (ids >= 0)
    rows = (ids + rb[None, :])[fl]
    ref = outs[("S", 15)]
    for k, v in outs.items():
        same_t = bool((v[0][rows] == ref[0][rows]).all())
        same_l = bool((v[1][rows] == ref[1][rows]).all())
        touched = int((v[0] != t0).any(1).sum())
        print("CHECK %s feat=%d rows equal to shipped: table %s lin %s bias %s (rows changed %d, unique slots %d)"
              % (k[0], k[1], same_t, same_l, bool(v[2][0] == ref[2][0]), touched, rows.numel()))


def main():
    check_small()
    B, F, D, V = 65536, 26, 64, 10_000_000
    g = torch.Generator(device=dev); g.manual_seed(1)
    R = F * V
    table = torch.empty((R, D), device=dev)
    for r0 in range(0, R, 1 << 24):
        table[r0:r0 + (1 << 24)].normal_(0, 0.1, generator=g)
    lin = torch.zeros(R, device=dev)
    bias = torch.zeros(1, device=dev)
    rb = torch.arange(F, device=dev, dtype=torch.int64) * V
    ld = 1680
    NB = 6
    sets = []
    for _ in range(NB):
        ids = torch.randint(0, V, (B, F), device=dev, generator=g)
        sets.append((ids, ops.emb_sort_slots(ids, rb, R)))
    grad = torch.randn((B, ld), device=dev, generator=g) * 1e-3
    grad_src = grad.clone()
    sum_x = torch.randn((B, D), device=dev, generator=g)
    dl = torch.randn(B, device=dev, generator=g) * 1e-3
    x_sorted = torch.empty((B * F, D), device=dev)
    alg = B * (12 * F * D + 16 * F)

    def timeit(fn, iters=12, warm=3, pre=None):
        for i in range(warm):
            if pre: pre()
            fn(i % NB)
        torch.cuda.synchronize()
        evs = []
        for i in range(iters):
            if pre: pre()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); fn(i % NB); e.record()
            evs.append((s, e))
        torch.cuda.synchronize()
        ts = sorted(s.elapsed_time(e) for s, e in evs)
        return ts[len(ts) // 2] * 1e3, ts[0] * 1e3

    def name(feat):
        return "+".join(k for k, v in X.items() if feat & v) or "none"

    def go(kind, feat, grid, pre=None, tag="", **kw):
        t, tmin = timeit(lambda k: run_variant(kind, feat, grid, sets[k][0], sets[k][1], rb, R, F, B, grad, ld, sum_x, dl, table, lin, bias, **kw),
                         pre=pre)
        print("K4LADDER %s %-28s grid=%5d %s median %7.1f us (min %7.1f)  alg %.0f GB/s  frac %.3f"
              % (kind, name(feat), grid, tag, t, tmin, alg / t / 1e3, alg / t / 1e3 / 8000), flush=True)

    for rep in range(3):
        print("---- pass %d" % rep)
        go("S", 11, 8192, tag="grid_d=2048", x_sorted=x_sorted)
        go("S", 10, 8192, tag="grid_d=2048", x_sorted=x_sorted)
        go("S", 11, 8192, tag="grid_d=0   ", x_sorted=x_sorted, grid_d=0)
        go("S", 11, 16384, tag="grid_d=2048", x_sorted=x_sorted)
        for feat in (15, 191, 703, 447, 959):
            go("N", feat, 8192)
        go("N", 191, 16384); go("N", 191, 4096)
    print("---- gradient freshly written before each launch (436 MB copy)")
    pre = lambda: grad.copy_(grad_src)
    for rep in range(2):
        go("S", 11, 8192, pre=pre, tag="fresh", x_sorted=x_sorted)
        for feat in (191, 447):
            go("N", feat, 8192, pre=pre, tag="fresh")


main()
