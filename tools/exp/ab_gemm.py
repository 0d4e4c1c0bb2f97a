"""EXPERIMENT (historical: the dr_debug_set_gemm_variant hook it drove was removed once the result was in DESIGN.md): A/B of the first GEMM kernel (git history) vs the current one, same process, interleaved rounds."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deep_recommenders_amd import _lib
HERE = os.path.dirname(os.path.abspath(__file__))
old = ctypes.CDLL(os.path.join(HERE, "libdense_old.so"))
p, i64, i32, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_float
old.dr_linear_fwd.argtypes = [p, i64, p, i64, p, i64, i32, i32, i32, p, i64, p]
old.dr_linear_bwd_dx.argtypes = [p, i64, p, i64, i64, i32, i32, p, i64, i32, p, i64, p]
old.dr_linear_bwd_dw.argtypes = [p, i64, p, i64, i64, i32, i32, f32, p, i64, p, p]
new = _lib.lib()
dev = "cuda"
g = torch.Generator(device=dev); g.manual_seed(0)
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, iters=4, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


M = 65536
for (K, N) in [(1677, 256), (1677, 1677), (1677, 1024)]:
    ldx = (K + 3) // 4 * 4
    x = torch.randn((M, ldx), device=dev, generator=g)
    W = torch.randn((K, N), device=dev, generator=g) / K ** 0.5
    b = torch.zeros(N, device=dev)
    ldn = (N + 3) // 4 * 4
    y = torch.empty((M, ldn), device=dev)
    dy = torch.randn((M, ldn), device=dev, generator=g)
    dx = torch.empty((M, ldx), device=dev)
    dW = torch.zeros((K, N), device=dev)
    fl = 2.0 * M * K * N
    for rnd in range(3):
        res = {}
        for name, L in (("old", old), ("new", new)):
            new.dr_debug_set_gemm_variant(2)
            tf = timeit(lambda: L.dr_linear_fwd(x.data_ptr(), ldx, W.data_ptr(), N, b.data_ptr(), M, K, N, 1, y.data_ptr(), ldn, st))
            td = timeit(lambda: L.dr_linear_bwd_dx(dy.data_ptr(), ldn, W.data_ptr(), N, M, K, N, x.data_ptr(), ldx, 0, dx.data_ptr(), ldx, st))
            if name == "old":
                tw = timeit(lambda: L.dr_linear_bwd_dw(x.data_ptr(), ldx, dy.data_ptr(), ldn, M, K, N, 1e-6, dW.data_ptr(), N, b.data_ptr(), st))
            else:
                tw = timeit(lambda: L.dr_linear_bwd_dw(x.data_ptr(), ldx, dy.data_ptr(), ldn, M, K, N, 1e-6, dW.data_ptr(), N, b.data_ptr(), None, 0, st))
            res[name] = (fl / tf / 1e12, fl / td / 1e12, fl / tw / 1e12)
        tt = timeit(lambda: torch.relu(torch.addmm(b, x[:, :K], W)))
        print("ABOLD %dx%d round %d  first-kernel fwd/dx/dw %.1f %.1f %.1f | current %.1f %.1f %.1f | rocblas fwd %.1f" % (
            K, N, rnd, *res["old"], *res["new"], fl / tt / 1e12))
