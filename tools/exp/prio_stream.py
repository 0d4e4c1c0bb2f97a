"""EXPERIMENT: does a high-priority training stream keep the side streams' HBM-bound kernels (next batch's plan / routing, owner-side
gathers) from slowing the step's own kernels?  The engine's training stream is "whatever stream the caller is on": run the same steps on
the default stream and on a stream created with the highest priority.  MODE=single | sharded."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
mode = os.environ.get("MODE", "single")
B, F, V, D, Nd = 65536, 26, 10_000_000, 64, 13
dev = torch.device("cuda:0")
try:
    print("priority range", torch.cuda.Stream.priority_range())
except Exception as e:
    print("priority_range:", e)
if mode == "sharded":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29592")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from deep_recommenders_amd.sharded import ShardedDeepFMEngine
    eng = ShardedDeepFMEngine(F, V, D, [256, 32], B, num_dense=Nd, lr=0.01, device=dev, world=1, rank=0, micro_batches=2)
else:
    from deep_recommenders_amd.engine import DeepFMEngine
    eng = DeepFMEngine(F, V, D, [256, 32], B, num_dense=Nd, lr=0.01, device="cuda")
g = torch.Generator(device="cuda"); g.manual_seed(42)
bs = [(torch.randint(0, 10**16, (B, F), device="cuda", generator=g), torch.log1p(torch.randn((B, Nd), device="cuda", generator=g).abs()),
       (torch.rand(B, device="cuda", generator=g) < 0.25).float()) for _ in range(8)]
nb = len(bs)
def run(n, i0):
    for i in range(i0, i0 + n):
        if mode == "sharded":
            eng.train_step(*bs[i % nb], next_keys=bs[(i + 1) % nb][0])
        else:
            eng.train_step(*bs[i % nb], next_keys=bs[(i + 1) % nb][0], next_dense=bs[(i + 1) % nb][1])
def timed(tag):
    run(60, 0); torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter(); run(100, 60); torch.cuda.synchronize()
        print("PRIO %s %s: %.4f ms/step" % (mode, tag, (time.perf_counter() - t0) * 10), flush=True)
timed("default stream")
hp = torch.cuda.Stream(priority=int(os.environ.get("PRIO", "-1")))
hp.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(hp):
    timed("high-priority stream")
torch.cuda.synchronize()
timed("default stream again")
if mode == "sharded":
    dist.destroy_process_group()
