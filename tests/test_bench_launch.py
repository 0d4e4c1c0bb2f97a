"""bench.py's launch contract (INTEGRATION.md "Launching N ranks"): `python bench.py --gpus N` starts its N ranks itself when no
launcher wrapped it, refuses with a device-count message when the node has fewer GPUs, and prints ONE JSON line from rank 0.

CPU part: the spawn path end to end over gloo (`--launch-check`: rendezvous on a free 127.0.0.1 port, one small instance of every
collective the sharded engines issue, through the transport class they use).  GPU part (one GPU): the device-count refusal, and
two ranks on the one device through `--share-device` (sharded.HostStagedTransport) running the real sharded step and reporting
the exchange (collective sizes / durations, exposed wait of the training stream)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, BENCH] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    return p.returncode, p.stdout, p.stderr


def _json_lines(out):
    lines = []
    for ln in out.splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                lines.append(json.loads(ln))
            except ValueError:
                pass
    return lines


def _no_gpu():
    import torch
    return not torch.cuda.is_available()


@pytest.mark.timeout(300)
def test_gpus2_without_launcher_spawns_two_ranks_and_prints_one_line():
    if not _no_gpu():
        pytest.skip("CPU form of the check (gloo); the GPU form is test_share_device_two_ranks_on_one_gpu")
    rc, out, err = _run(["--gpus", "2", "--launch-check"])
    assert rc == 0, err[-2000:]
    lines = _json_lines(out)
    assert len(lines) == 1, out
    d = lines[0]
    assert d["launch_check"] is True and d["n_gpus"] == 2 and d["self_launched"] is True and d["all_ok"] is True
    assert d["backend"] == "gloo" and d["transport"] == "TorchDistTransport"
    assert sorted(r["rank"] for r in d["ranks"]) == [0, 1]
    assert len({r["pid"] for r in d["ranks"]}) == 2                       # two processes
    assert "starting 2 ranks" in err and "--master-addr 127.0.0.1" in err


@pytest.mark.timeout(300)
def test_wrapped_by_the_drivers_launcher_line():
    """The driver's own form for N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` -- bench.py must NOT spawn again, and rank 0 prints the one line."""
    if not _no_gpu():
        pytest.skip("CPU form of the check (gloo)")
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "2", "--launch-check"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["self_launched"] is False and lines[0]["all_ok"] is True
    assert "starting 2 ranks" not in p.stderr


@pytest.mark.timeout(120)
def test_gpus2_on_a_box_without_enough_gpus_names_the_device_count():
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    need = have + 1 if have else 2
    rc, out, err = _run(["--gpus", str(need), "--steps", "1", "--warmup", "0"])
    assert rc != 0
    assert not _json_lines(out)
    assert ("%d GPU(s) visible" % have) in err and ("--gpus %d" % need) in err and "--share-device" in err


@pytest.mark.timeout(120)
def test_launcher_world_size_must_match_gpus():
    rc, out, err = _run(["--gpus", "2", "--launch-check"], env_extra={"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc != 0 and "WORLD_SIZE=4" in err and "--nproc-per-node" in err


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_share_device_two_ranks_on_one_gpu():
    """The self-launched N = 2 bench through the REAL sharded step (HIP kernels, two micro-batches, routes prefetched) with both
    ranks on cuda:0 and the collectives staged through the host: the line must carry n_gpus = 2, the exchange report and a finite
    loss.  (Timing is meaningless here and flagged as such in the line.)"""
    rc, out, err = _run(["--gpus", "2", "--share-device", "--steps", "3", "--warmup", "1", "--batch", "4096", "--vocab", "20000",
                         "--no-cpu-baseline", "--events", "on"], timeout=500)
    assert rc == 0, err[-3000:]
    lines = _json_lines(out)
    assert len(lines) == 1, out
    d = lines[0]
    assert d["n_gpus"] == 2 and "NOT_A_MEASUREMENT" in d
    assert d["config"]["parallelism"] == "dp2+row-sharded-tables"
    loss = d["config"]["final_loss"]
    assert loss == loss and 0.0 < loss < 5.0
    ex = d["exchange"]
    assert ex["world"] == 2 and ex["ranks_in_process_group"] == 2 and ex["transport"] == "HostStagedTransport"
    for tag in ("a2a_rows", "a2a_grads"):
        c = ex["collectives"][tag]
        assert c["bytes_sent_per_rank"] > 0 and c["avg_us"] > 0 and c["GBps_per_link"] > 0
    assert ex["exposed_us_per_step"] is not None and ex["exposed_us_per_step"] >= 0
    # the weak line carries SURVEY 8(e)'s split too: the same global batch divided over the two ranks
    st = d["strong"]
    assert "failed" not in st, st
    assert st["global_batch"] == 4096 and st["per_gpu_batch"] == 2048 and st["ms_per_step"] > 0 and 0.0 < st["final_loss"] < 5.0
