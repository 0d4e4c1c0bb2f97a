"""EXPERIMENT: first-order weight as a separate gather vs packed behind the embedding row (pitch 68 floats)."""
import ctypes, os
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(HERE, "libexp_emb.so"))
p, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
L.exp_fwd_lin.argtypes = [i32, i32, p, i64, i32, p, p, i64, p, i64, p, p]
dev = "cuda"
B, F, V = 65536, 26, 10_000_000
g = torch.Generator(device=dev); g.manual_seed(1)
NB = 8
ids = [torch.randint(0, V, (B, F), device=dev, generator=g) for _ in range(NB)]
concat = torch.zeros((B, 1680), device=dev)
out_lin = torch.zeros(B, device=dev)
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, iters=16, warm=4):
    for i in range(warm): fn(i % NB)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters): fn(i % NB)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for packed, pitch in ((0, 64), (1, 68)):
    table = torch.empty((F * V, pitch), device=dev).normal_(0, 0.1, generator=g)
    lin = torch.zeros(F * V, device=dev)
    for grid in (4096, 16384):
        t = timeit(lambda k: L.exp_fwd_lin(packed, grid, ids[k].data_ptr(), B, F, table.data_ptr(), lin.data_ptr(), V,
                                           concat.data_ptr(), 1680, out_lin.data_ptr(), st))
        print("EXPLIN packed=%d grid=%d  %.1f us" % (packed, grid, t))
    del table, lin
    torch.cuda.empty_cache()
