"""EXPERIMENT: is the FIRST GPU process on a freshly started box slower, and for how long?  Builds bench.py's default engine and prints the wall
time per step of consecutive 50-step blocks for ~6 s (same batches, same prefetch as bench.py)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
t_imp = time.perf_counter()
import torch
from deep_recommenders_amd.engine import DeepFMEngine
print("import %.1f s" % (time.perf_counter() - t_imp), flush=True)
B, F, V, D, Nd = 65536, 26, 10_000_000, 64, 13
t0 = time.perf_counter()
eng = DeepFMEngine(F, V, D, [256, 32], B, num_dense=Nd, lr=0.01, device="cuda")
torch.cuda.synchronize()
print("engine %.1f s" % (time.perf_counter() - t0), flush=True)
g = torch.Generator(device="cuda"); g.manual_seed(42)
bs = [(torch.randint(0, 10**16, (B, F), device="cuda", generator=g), torch.log1p(torch.randn((B, Nd), device="cuda", generator=g).abs()),
       (torch.rand(B, device="cuda", generator=g) < 0.25).float()) for _ in range(8)]
nb = len(bs)
i = 0
series = []
t_start = time.perf_counter()
while time.perf_counter() - t_start < float(os.environ.get("PROBE_S", "6")):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50):
        eng.train_step(*bs[i % nb], next_keys=bs[(i + 1) % nb][0], next_dense=bs[(i + 1) % nb][1]); i += 1
    torch.cuda.synchronize()
    series.append((time.perf_counter() - t) / 50 * 1e3)
print("blocks of 50 steps, ms/step: first 12:", " ".join("%.3f" % x for x in series[:12]))
print("  every 10th block:", " ".join("%.3f" % x for x in series[::10]))
print("  min %.3f  median %.3f  last %.3f  n %d" % (min(series), sorted(series)[len(series) // 2], series[-1], len(series)), flush=True)
# host time per step: enqueue 40 steps without waiting for the GPU (the queues are deep enough), then wait
for rep in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(40):
        eng.train_step(*bs[i % nb], next_keys=bs[(i + 1) % nb][0], next_dense=bs[(i + 1) % nb][1]); i += 1
    t_enq = time.perf_counter() - t
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t
    print("host: 40 steps enqueued in %.2f ms (%.3f ms/step), finished in %.2f ms (%.3f ms/step)" % (t_enq * 1e3, t_enq / 40 * 1e3, t_all * 1e3, t_all / 40 * 1e3), flush=True)
