"""EXPERIMENT driver (not product): times exp_emb.hip variants with distinct id batches per iteration."""
import ctypes, os, sys, json
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(HERE, "libexp_emb.so"))
p, i64, i32, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_float
L.exp_fwd.argtypes = [i32, i32, p, i64, i32, p, i64, p, i64, p, p]
L.exp_bwd.argtypes = [i32, i32, p, i64, i32, i64, p, p, i64, p, p, f32, p, p]
dev = "cuda"
B, F, D = 65536, 26, 64
ld = 1680
g = torch.Generator(device=dev); g.manual_seed(1)


def timeit(fn, nb, iters=16, warm=4):
    for i in range(warm): fn(i % nb)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters): fn(i % nb)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for V in (10_000_000, 1_000_000):
    table = torch.empty((F * V, D), device=dev).normal_(0, 0.1, generator=g)
    NB = 8
    ids = [torch.randint(0, V, (B, F), device=dev, generator=g) for _ in range(NB)]
    concat = torch.zeros((B, ld), device=dev)
    d_concat = torch.randn((B, ld), device=dev, generator=g) * 1e-3
    sum_x = torch.randn((B, D), device=dev, generator=g)
    dl = torch.randn(B, device=dev, generator=g) * 1e-3
    sink = torch.zeros(4, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    names_f = {0: "U8 plain", 1: "U8 gather-only(no store)", 2: "U8 nt-store", 3: "U8 nt-load", 4: "U8 nt-load+nt-store",
               5: "U4 plain", 6: "U4 gather-only"}
    for var in range(7):
        for grid in (2048, 4096, 16384):
            t = timeit(lambda k: L.exp_fwd(var, grid, ids[k].data_ptr(), B, F, table.data_ptr(), V, concat.data_ptr(), ld,
                                           sink.data_ptr(), st), NB)
            print("V=%d fwd %-26s grid=%5d  %7.1f us  alg %.0f GB/s" % (V, names_f[var], grid, t, B * (8 * F * D + 12 * F + 8) / t / 1e3))
    names_b = {0: "atomic strided + FM", 1: "plain RMW + FM", 2: "atomic strided, no FM read", 3: "plain RMW, no FM read"}
    for var in range(4):
        for grid in (2048, 8192):
            t = timeit(lambda k: L.exp_bwd(var, grid, ids[k].data_ptr(), B, F, V, d_concat.data_ptr(), concat.data_ptr(), ld,
                                           sum_x.data_ptr(), dl.data_ptr(), -1e-3, table.data_ptr(), st), NB)
            print("V=%d bwd %-26s grid=%5d  %7.1f us  alg %.0f GB/s" % (V, names_b[var], grid, t, B * (12 * F * D + 16 * F) / t / 1e3))
    del table
    torch.cuda.empty_cache()
