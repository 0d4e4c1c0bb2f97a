"""The N > 1 path with the REAL kernels: two ranks (two processes) share cuda:0 and run the row-sharded engines with `HipPrims`;
the collectives go through `sharded.HostStagedTransport` (device buffers staged through the host over a gloo group -- RCCL
refuses two ranks on one device).  Same assertion as tests/test_sharded_gloo.py / test_sharded_retrieval_gloo.py -- N steps on two
ranks equal the single-process oracle on the concatenated global batches -- but what runs between the exchanges is the shipped
HIP code: bucketing, owner-side gather, the fused first layer over the RECEIVED rows, pack, the owner-side sorted K4 (SGD / Adam),
the register-split GEMMs, the cross kernels, the in-batch softmax over the all-gathered candidates, the sharded top-K merge
(reference semantics of the merge: keras/models/retrieval/factorized_top_k.py:215-233 of the reference).

(until round 3 the HIP kernels had only ever run through an RCCL group of ONE.)"""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import test_sharded_gloo as SG                     # noqa: E402  (workers, problems and single-process references)
import test_sharded_retrieval_gloo as SR           # noqa: E402

pytestmark = pytest.mark.gpu

# generic kernels: K3 / K4 / GEMMs at a small odd shape, two micro-batches, one missing id
SMALL = dict(F=6, V=3001, D=16, B=768, Nd=3, units=[32, 16], lr=0.05, key_max=10**14, steps=3, expect_fused_l0=False)
# the bench's kernels: D = 64 and a wide first layer -> K3 runs inside the register-split GEMM over the received row buffer,
# wgrad / dgrad on the register-split kernels (2048-row micro-batches)
FUSED = dict(F=5, V=2003, D=64, B=4096, Nd=3, units=[128, 16], lr=0.05, key_max=10**14, steps=2, expect_fused_l0=True)


# the de-duplicated exchange on a problem where most slots share their row (500 ids per field, 2048 examples per micro-batch)
DEDUP = dict(F=5, V=503, D=64, B=4096, Nd=3, units=[128, 16], lr=0.05, key_max=10**14, steps=2, expect_fused_l0=True, dedup=True)


def _assert_close(name, got, want, before=None, rel=2e-3):
    """parameters after N steps: |got - want| <= rel * |want - before| + a few ulp + 1e-3 rms(update) for all but 1e-4 of the
    elements (ReLU ties, see tests/test_gpu_benchcfg.py::_assert_update); without `before`: plain allclose."""
    got, want = got.double().numpy(), want.double().numpy()
    if before is None:
        np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-6, err_msg=name)
        return
    b = before.double().numpy()
    d = want - b
    rms = float(np.sqrt(np.mean(d * d)))
    assert rms > 0, name + ": vacuous (zero oracle update)"
    err = np.abs(got - want)
    tol = 4 * np.spacing(np.abs(want).astype(np.float32)).astype(np.float64) + rel * np.abs(d) + 1e-3 * rms
    frac = float((err > tol).mean())
    assert frac <= 1e-4, "%s: %.2e of %d elements off; worst |err| %.3e (rms update %.3e)" % (name, frac, err.size, float(err.max()), rms)
    assert float(err.max()) <= 0.5 * float(np.abs(d).max()) + 2 * rms, name


@pytest.mark.timeout(600)
@pytest.mark.parametrize("cfg_name,micro_batches", [("SMALL", 2), ("FUSED", 2), ("FUSED", 1), ("DEDUP", 2)])
def test_two_ranks_one_gpu_deepfm_sgd_equals_oracle(tmp_path, cfg_name, micro_batches):
    cfg = dict({"SMALL": SMALL, "FUSED": FUSED, "DEDUP": DEDUP}[cfg_name])
    world = 2
    res = SG._spawn(SG._worker, (micro_batches, "sgd", None, True, cfg), tmp_path, timeout=500)
    Ws0, bs0 = res[0][6], res[0][7]
    for a, b in zip(Ws0, res[1][6]):
        assert torch.equal(a, b)
    losses, tab, li, bias, Wc, bc = SG._reference_deepfm(cfg, Ws0, bs0, "sgd", dtype=torch.float64)
    table0, lin0, _ = SG._global_problem(cfg)
    for t, lo in enumerate(losses):                 # global loss = mean of the per-rank means; north_star: 1e-5 relative
        got = 0.5 * (res[0][0][t] + res[1][0][t])
        assert abs(got - lo) <= 1e-5 * abs(lo), (t, got, lo)
    for r in range(world):
        _, tab_r, lin_r, Ws_r, bs_r, bias_r, _, _ = res[r]
        pairs = SG._shard_views(cfg, world, r, tab_r, lin_r, tab, li)
        before = SG._shard_views(cfg, world, r, tab_r, lin_r, table0, lin0)
        _assert_close("table shard r%d" % r, torch.cat([g.reshape(-1) for g, _ in pairs[0::2]]),
                      torch.cat([w.reshape(-1) for _, w in pairs[0::2]]), torch.cat([w.reshape(-1) for _, w in before[0::2]]))
        _assert_close("first-order shard r%d" % r, torch.cat([g for g, _ in pairs[1::2]]), torch.cat([w for _, w in pairs[1::2]]),
                      torch.cat([w for _, w in before[1::2]]))
        for i in range(len(Wc)):
            _assert_close("W%d r%d" % (i, r), Ws_r[i], Wc[i], Ws0[i])
            _assert_close("b%d r%d" % (i, r), bs_r[i], bc[i], bs0[i])
        _assert_close("first-order bias r%d" % r, bias_r, bias, torch.zeros(1))


@pytest.mark.timeout(600)
def test_two_ranks_one_gpu_deepfm_adam_equals_oracle(tmp_path):
    cfg = dict(FUSED)
    world = 2
    lr = cfg["lr"]
    res = SG._spawn(SG._worker, (2, "adam", None, True, cfg), tmp_path, timeout=500)
    Ws0, bs0 = res[0][6], res[0][7]
    losses, tab, li, bias, Wc, bc = SG._reference_deepfm(cfg, Ws0, bs0, "adam", dtype=torch.float64)
    for t, lo in enumerate(losses):
        got = 0.5 * (res[0][0][t] + res[1][0][t])
        assert abs(got - lo) <= 1e-5 * abs(lo), (t, got, lo)

    def close(name, got, want):
        # Adam divides by sqrt(v): tolerances are fractions of one step (see tests/test_gpu_benchcfg.py::_assert_close_adam)
        err = (got.double() - want.double()).abs().numpy()
        assert float((err > 2e-2 * lr).mean()) <= 2e-2, "%s: %.2e of the elements off by > 2 %% of a step" % (name, float((err > 2e-2 * lr).mean()))
        assert float(err.mean()) <= 2e-3 * lr and float(err.max()) <= lr, "%s: mean %.3e max %.3e of a step" % (name, err.mean() / lr, err.max() / lr)
    for r in range(world):
        _, tab_r, lin_r, Ws_r, bs_r, bias_r, _, _ = res[r]
        pairs = SG._shard_views(cfg, world, r, tab_r, lin_r, tab, li)
        close("table shard r%d" % r, torch.cat([g.reshape(-1) for g, _ in pairs[0::2]]), torch.cat([w.reshape(-1) for _, w in pairs[0::2]]))
        close("first-order shard r%d" % r, torch.cat([g for g, _ in pairs[1::2]]), torch.cat([w for _, w in pairs[1::2]]))
        for i in range(len(Wc)):
            close("W%d r%d" % (i, r), Ws_r[i], Wc[i])
            close("b%d r%d" % (i, r), bs_r[i], bc[i])
        close("first-order bias r%d" % r, bias_r, bias)


DCN_GPU = dict(F=4, V=1501, D=16, B=512, Nd=3, units=[32, 16], L=2, lr=0.1, diag=0.1)


@pytest.mark.timeout(600)
def test_two_ranks_one_gpu_dcn_equals_oracle(tmp_path):
    cfg = dict(DCN_GPU)
    world = 2
    res = SG._spawn(SG._dcn_worker, (True, cfg), tmp_path, timeout=500)
    assert torch.equal(res[0][3], res[1][3])
    flat0 = res[0][3]
    table0, _ = SG._dcn_problem(cfg)
    losses, flat, tab = SG._reference_dcn(cfg, flat0, dtype=torch.float64)
    for t, lo in enumerate(losses):
        got = 0.5 * (res[0][0][t] + res[1][0][t])
        assert abs(got - lo) <= 1e-5 * abs(lo), (t, got, lo)
    F, V = cfg["F"], cfg["V"]
    rps = (V + world - 1) // world
    for r in range(world):
        _, tab_r, flat_r, _ = res[r]
        _assert_close("dense parameters r%d" % r, flat_r, flat, flat0)
        got, want, before = [], [], []
        for f in range(F):
            gid = torch.arange(r, V, world)
            got.append(tab_r[f * rps:f * rps + len(gid)].reshape(-1))
            want.append(tab[f * V + gid].reshape(-1))
            before.append(table0[f * V + gid].reshape(-1))
        _assert_close("table shard r%d" % r, torch.cat(got), torch.cat(want), torch.cat(before))


@pytest.mark.timeout(600)
def test_two_ranks_one_gpu_two_tower_equals_oracle(tmp_path):
    SR.run_and_check(tmp_path, gpu=True)
