"""EXPERIMENT: DeepFMEngine.train_step captured in a HIP graph (torch.cuda.CUDAGraph) vs eager launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deep_recommenders_amd.engine import DeepFMEngine

shape = os.environ.get("SHAPE", "ml")
if shape == "ml":
    B, F, V, D, Nd = 4096, 7, 10000, 16, 0
else:
    B, F, V, D, Nd = 65536, 26, 10_000_000, 64, 13
dev = "cuda"
eng = DeepFMEngine(F, V, D, [256, 32], B, num_dense=Nd, lr=0.01, device=dev)
g = torch.Generator(device=dev); g.manual_seed(0)
batches = [(torch.randint(0, 10**16, (B, F), device=dev, generator=g),
            torch.rand((B, Nd), device=dev, generator=g) if Nd else None,
            (torch.rand(B, device=dev, generator=g) < 0.25).float()) for _ in range(4)]
skeys = torch.empty_like(batches[0][0]); sdense = torch.empty((B, Nd), device=dev) if Nd else None; slabels = torch.empty(B, device=dev)


def feed(i):
    k, d, l = batches[i % 4]
    skeys.copy_(k); slabels.copy_(l)
    if Nd: sdense.copy_(d)


def timeit(fn, n):
    for i in range(5): fn(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


n = 300 if shape == "ml" else 30
def eager(i):
    feed(i); eng.train_step(skeys, sdense, slabels)
t_e = timeit(eager, n)
loss_e = eng.loss.item()
# capture
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for i in range(3): eager(i)
torch.cuda.current_stream().wait_stream(s)
graph = torch.cuda.CUDAGraph()
feed(0)
with torch.cuda.graph(graph):
    eng.train_step(skeys, sdense, slabels)
def replay(i):
    feed(i); graph.replay()
t_g = timeit(replay, n)
print("GRAPH shape=%s eager %.4f ms/step  graph %.4f ms/step  (%.1f -> %.1f M ex/s) loss %.5f / %.5f" % (
    shape, t_e, t_g, B / t_e / 1e3, B / t_g / 1e3, loss_e, eng.loss.item()))
