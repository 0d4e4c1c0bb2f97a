"""EXPERIMENT: N random read-modify-writes of 4..128 bytes into a 1 GiB buffer (one per distinct 128-byte line)."""
import ctypes, os, subprocess, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libexp_emb.so")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics",
                       os.path.join(here, "exp_emb.hip"), "-o", so])
L = ctypes.CDLL(so)
dev = "cuda"
lines = (1 << 30) // 128
buf = torch.zeros(lines * 32, device=dev)
n = 1_700_000
g = torch.Generator(device=dev); g.manual_seed(3)
idxs = [torch.randperm(lines, device=dev, generator=g)[:n].contiguous() for _ in range(6)]
st = torch.cuda.current_stream().cuda_stream
for read in (1, 0):
    for w in (4, 16, 32, 64, 128):
        def run(k): L.exp_rmw(w, read, ctypes.c_void_p(idxs[k].data_ptr()), ctypes.c_int64(n), ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(st))
        for k in range(3): run(k)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for k in range(12): run(k % 6)
        e.record(); torch.cuda.synchronize()
        print("EXPRMW read=%d width=%3d B  %.1f us" % (read, w, s.elapsed_time(e) / 12 * 1e3))
