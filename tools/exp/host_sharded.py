"""EXPERIMENT: is the sharded DeepFM step (world 1 through the N > 1 code path) bound by the host?  Enqueue time of N steps (the Python
loop, no synchronisation inside) against the device time of the same N steps; then cProfile of the loop (top entries by own time)."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29591")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
from deep_recommenders_amd.sharded import ShardedDeepFMEngine
B, F, V, D, Nd = 65536, 26, 10_000_000, 64, 13
eng = ShardedDeepFMEngine(F, V, D, [256, 32], B, num_dense=Nd, lr=0.01, device=torch.device("cuda:0"), world=1, rank=0, micro_batches=2)
g = torch.Generator(device="cuda"); g.manual_seed(42)
bs = [(torch.randint(0, 10**16, (B, F), device="cuda", generator=g), torch.log1p(torch.randn((B, Nd), device="cuda", generator=g).abs()),
       (torch.rand(B, device="cuda", generator=g) < 0.25).float()) for _ in range(8)]
nb = len(bs)
def run(n, i0):
    for i in range(i0, i0 + n):
        eng.train_step(*bs[i % nb], next_keys=bs[(i + 1) % nb][0])
run(40, 0); torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); run(100, 40); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("HOST sharded: enqueue %.3f ms/step, until the device is done %.3f ms/step" % ((t1 - t0) * 10, (t2 - t0) * 10), flush=True)
pr = cProfile.Profile()
pr.enable(); run(200, 40); pr.disable()
torch.cuda.synchronize()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(32)
print(st.getvalue()[:7000])
dist.destroy_process_group()
