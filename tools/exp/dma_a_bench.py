"""EXPERIMENT driver (round 5): the plain f16x2 NT GEMM with its activation operand fetched by LDS-DMA (bf3_emb_linear_kernel<1, 1>, the fused
forward's kernel with the gather switched off: dr_exp_h2_linear_nt_dma, C++ linkage) against the shipped register-operand kernel
(dr_h2_linear_nt), M = 65 536: forward shape (K 1677 -> N 256), dgrad shape (256 -> 1677), DCN's cross shape (1677 -> 1677)."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from deep_recommenders_amd import ops, _lib
from deep_recommenders_amd._lib import ptr, stream_ptr
L = ctypes.CDLL(os.path.join(os.path.dirname(_lib.__file__), "lib", "libdr_hotpath.so"))
fn = getattr(L, "_Z23dr_exp_h2_linear_nt_dmaPKflPKjPKvllS2_liiS0_iPflPv")
fn.restype = ctypes.c_int
P, I64, I32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
fn.argtypes = [P, I64, P, P, I64, I64, P, I64, I32, I32, P, I32, P, I64, P]
dev, M = "cuda", 65536
torch.manual_seed(0)


def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def case(name, K, N):
    ldk = (K + 31) // 32 * 32 + 32                      # rows padded so that the DMA's reads past column K stay inside the row / allocation
    buf = torch.zeros(M + 8, ldk, device=dev)
    buf[:M, :K] = torch.randn(M, K, device=dev)
    x = buf[:M, :K]
    w = torch.randn(K, N, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    wp = ops.H2WeightPlanes(w)
    am = ops.h2_amax(x)
    out0 = torch.zeros(M, (N + 3) // 4 * 4, device=dev)[:, :N]
    out1 = torch.zeros(M, (N + 3) // 4 * 4, device=dev)[:, :N]
    f0 = lambda: ops.h2_linear_nt(x, am, wp.wt, bias=b, act=1, out=out0)
    def f1():
        rc = fn(ptr(x), x.stride(0), ptr(am), ptr(wp.wt.buf), wp.wt.plane_stride, wp.wt.ld, ptr(wp.wt.amax), M, N, K, ptr(b), 1, ptr(out1), out1.stride(0),
                stream_ptr())
        assert rc == 0, rc
    f0(); f1()
    torch.cuda.synchronize()
    ref = (x[:2048].double() @ w.double() + b.double()).clamp_min(0)
    e0 = (out0[:2048].double() - ref).abs().max().item() / ref.abs().max().item()
    e1 = (out1[:2048].double() - ref).abs().max().item() / ref.abs().max().item()
    same = float((out0 - out1).abs().max())
    print("DMAA %-22s registers %8.1f us (err %.1e)   lds-dma %8.1f us (err %.1e)   max |diff| %.2e" % (name, timeit(f0), e0, timeit(f1), e1, same), flush=True)


case("forward K=1677 N=256", 1677, 256)
case("dgrad K=256 N=1677", 256, 1677)
case("square 1677", 1677, 1677)
case("mlp 1677->1024", 1677, 1024)
