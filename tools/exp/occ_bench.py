"""EXPERIMENT driver (not product): the 16-wave f16x2 NT GEMM (csrc/h2_occ.hip) against the 8-wave register-split kernel.
One setting per process (DR_H2_OCC=0 / 1 is read once by the library); `python tools/exp/occ_bench.py` runs both as subprocesses,
compares their outputs bit for bit and prints the times."""
import os, sys, subprocess, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))

CASES = [("forward K=1677 N=256", 65536, 1677, 256, {}), ("dgrad K=256 N=1677", 65536, 256, 1677, {}),
         ("dgrad+mask", 65536, 256, 1677, {"mask": True}), ("accumulate 1677^2", 65536, 1677, 1677, {"acc": True}),
         ("edges M=1000 K=77 N=300 bias relu", 1000, 77, 300, {"bias": True, "act": 1}), ("tiny M=5", 5, 64, 40, {"bias": True}),
         ("mlp 1024->512", 65536, 1024, 512, {"bias": True, "act": 1})]


def child():
    import torch
    from deep_recommenders_amd import ops
    dev = "cuda"

    def timeit(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    for name, M, K, N, kw in CASES:
        torch.manual_seed(1)
        buf = torch.zeros(M, (K + 3) // 4 * 4, device=dev)
        buf[:, :K] = torch.randn(M, K, device=dev)
        x = buf[:, :K]
        w = torch.randn(K, N, device=dev) / K ** 0.5
        wp = ops.H2WeightPlanes(w)
        am = ops.h2_amax(x)
        ldn = (N + 3) // 4 * 4
        out = torch.zeros(M, ldn, device=dev)[:, :N]
        bias = torch.randn(N, device=dev) if kw.get("bias") else None
        mask = (torch.randn(M, ldn, device=dev)[:, :N]) if kw.get("mask") else None
        rec = ops.h2_record(dev)
        acc = bool(kw.get("acc"))
        if acc: out.normal_()
        base = out.clone()

        def fn():
            return ops.h2_linear_nt(x, am, wp.wt, bias=bias, act=kw.get("act", 0), mask=mask, accumulate=acc, out=out, out_amax=rec)
        fn()
        rows = min(M, 1024)
        ref = x[:rows].double() @ w.double()
        if bias is not None: ref = ref + bias.double()
        if kw.get("act"): ref = ref.clamp_min(0)
        if mask is not None: ref = ref * (mask[:rows] > 0)
        if acc: ref = ref + base[:rows].double()
        err = (out[:rows].double() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)
        digest = hashlib.sha1(out.contiguous().cpu().numpy().tobytes()).hexdigest()[:12]
        recv = rec.cpu().view(torch.float32).item()
        t = timeit(fn) if M >= 4096 else 0.0
        print("OCC=%s %-36s %8.1f us  rel.err %.1e  amax %.6e  sha %s" % (os.environ.get("DR_H2_OCC", "1"), name, t, err, recv, digest), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for occ in ("0", "1"):
            env = dict(os.environ, DR_H2_OCC=occ)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False)
