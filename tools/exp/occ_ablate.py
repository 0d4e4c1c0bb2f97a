"""EXPERIMENT driver (not product): compile-time ablations of the 16-wave GEMM (library built with DR_HIPCC_EXTRA=-DDR_OCC_ABLATE).
DR_OCC_DBG bits: 2 no MFMAs, 32 no fragment reads, 1 no weight DMA after the prologue, 8 A from cache, 64 no split / image writes."""
import os, sys, subprocess
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))


def child():
    import torch
    from deep_recommenders_amd import ops
    dev, M = "cuda", 65536

    def timeit(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    res = []
    for name, K, N in (("fwd", 1677, 256), ("dgrad", 256, 1677)):
        torch.manual_seed(1)
        buf = torch.zeros(M, (K + 3) // 4 * 4, device=dev)
        buf[:, :K] = torch.randn(M, K, device=dev)
        x = buf[:, :K]
        w = torch.randn(K, N, device=dev) / K ** 0.5
        wp = ops.H2WeightPlanes(w)
        am = ops.h2_amax(x)
        out = torch.zeros(M, (N + 3) // 4 * 4, device=dev)[:, :N]
        res.append("%s %7.1f" % (name, timeit(lambda: ops.h2_linear_nt(x, am, wp.wt, out=out))))
    print("OCCABL occ=%s dbg=%-3s  %s" % (os.environ.get("DR_H2_OCC", "1"), os.environ.get("DR_OCC_DBG", "0"), "   ".join(res)), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for dbg in sys.argv[1:] or ["0", "2", "32", "34", "1", "8", "9", "64", "73"]:
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, DR_OCC_DBG=dbg), check=False)
