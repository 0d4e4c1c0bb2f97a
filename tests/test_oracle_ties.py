"""The tie-aware half of the gradient oracle (oracle/torch_ref.py dnn(relu_masks=...) / check_ties), on the CPU.

The GPU parity tests hand the oracle the ReLU decisions the device made; these tests pin what that does and what it cannot hide:
  * with the oracle's own decisions the masked network IS the relu network (value and every gradient, bit for bit);
  * a decision flipped at a unit that is NOT a tie is reported by check_ties (the mask cannot cover a wrong pre-activation);
  * a decision flipped at a genuine tie (|z| at rounding level) passes and changes the input gradient of that one example only."""
import pytest
import torch

from oracle import torch_ref as T


def _net(seed=0, B=64, K=24, units=(16, 8)):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((B, K), generator=g, dtype=torch.float64)
    dims = [K] + list(units) + [1]
    Ws = [torch.randn((a, b), generator=g, dtype=torch.float64) / a ** 0.5 for a, b in zip(dims[:-1], dims[1:])]
    bs = [0.1 * torch.randn(b, generator=g, dtype=torch.float64) for b in dims[1:]]
    return x, Ws, bs


def _own_masks(x, Ws, bs):
    masks, h = [], x
    for W, b in zip(Ws[:-1], bs[:-1]):
        z = h @ W + b
        masks.append(z > 0)
        h = torch.relu(z)
    return masks


def test_masked_dnn_equals_relu_dnn_with_the_oracles_own_decisions():
    x, Ws, bs = _net()
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    Wa, Wb = [w.clone().requires_grad_(True) for w in Ws], [w.clone().requires_grad_(True) for w in Ws]
    ties = []
    ya = T.dnn(xa, Wa, bs)
    yb = T.dnn(xb, Wb, bs, relu_masks=_own_masks(x, Ws, bs), ties=ties)
    assert torch.equal(ya, yb)
    ga = torch.autograd.grad(ya.sum(), [xa] + Wa)
    gb = torch.autograd.grad(yb.sum(), [xb] + Wb)
    for a, b in zip(ga, gb):
        assert torch.equal(a, b)
    assert [t["disagree"] for t in ties] == [0, 0]
    T.check_ties(ties)


def test_a_wrong_decision_away_from_zero_is_not_a_tie():
    x, Ws, bs = _net(seed=1)
    masks = _own_masks(x, Ws, bs)
    z0 = x @ Ws[0] + bs[0]
    b_, u_ = divmod(int(z0.abs().argmax()), z0.shape[1])       # the unit farthest from zero
    masks[0][b_, u_] = ~masks[0][b_, u_]
    ties = []
    T.dnn(x, Ws, bs, relu_masks=masks, ties=ties)
    assert ties[0]["disagree"] == 1 and ties[0]["examples"] == [b_]
    with pytest.raises(AssertionError, match="not a tie"):
        T.check_ties(ties)


def test_a_flipped_tie_passes_and_moves_one_examples_gradient_only():
    x, Ws, bs = _net(seed=2)
    z0 = x @ Ws[0] + bs[0]
    b_, u_ = 5, 3
    bs[0] = bs[0].clone()
    # put unit (5, 3) within rounding of zero: shift example 5's input along W[:, 3] (other units of that example move too, they are
    # not near zero; the other examples are untouched)
    x = x.clone()
    x[b_] -= (z0[b_, u_] - 1e-12) * Ws[0][:, u_] / Ws[0][:, u_].pow(2).sum()
    masks = _own_masks(x, Ws, bs)
    assert bool(masks[0][b_, u_])                               # z = +1e-12: "on" for the oracle
    flipped = [m.clone() for m in masks]
    flipped[0][b_, u_] = False                                  # ... "off" for the implementation under test
    outs = []
    for m in (masks, flipped):
        xi = x.clone().requires_grad_(True)
        ties = []
        y = T.dnn(xi, Ws, bs, relu_masks=m, ties=ties)
        T.check_ties(ties)
        outs.append((y.detach(), torch.autograd.grad(y.sum(), xi)[0], ties))
    assert outs[1][2][0]["disagree"] == 1 and outs[1][2][0]["worst_abs_z"] < 1e-10
    assert (outs[0][0] - outs[1][0]).abs().max() < 1e-10        # the value does not care
    d = (outs[0][1] - outs[1][1]).abs().amax(1)
    assert d[b_] > 1e-3 and float(d.sum() - d[b_]) == 0.0       # the subgradient of that one example does
