"""EXPERIMENT: where the host's 0.33 ms per training step goes (cProfile over 300 steps of bench.py's default engine)."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from deep_recommenders_amd.engine import DeepFMEngine
B, F, V, D, Nd = 65536, 26, 10_000_000, 64, 13
eng = DeepFMEngine(F, V, D, [256, 32], B, num_dense=Nd, lr=0.01, device="cuda")
g = torch.Generator(device="cuda"); g.manual_seed(42)
bs = [(torch.randint(0, 10**16, (B, F), device="cuda", generator=g), torch.log1p(torch.randn((B, Nd), device="cuda", generator=g).abs()),
       (torch.rand(B, device="cuda", generator=g) < 0.25).float()) for _ in range(8)]
nb = len(bs)
def run(n, i0):
    for i in range(i0, i0 + n):
        eng.train_step(*bs[i % nb], next_keys=bs[(i + 1) % nb][0], next_dense=bs[(i + 1) % nb][1])
run(40, 0); torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable(); run(300, 40); pr.disable()
torch.cuda.synchronize()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(28)
print(st.getvalue()[:6000])
