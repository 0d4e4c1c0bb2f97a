"""GPU tests of the CIN (xDeepFM) and DIN ActivationUnit layers (second half), mirroring the reference's
tests/keras/test_xdeepfm.py and tests/keras/test_din.py, checked against the oracle (oracle/tf_semantics.py)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tf_semantics as O

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))


def test_cin_invalid_inputs():
    from deep_recommenders_amd.keras.models.ranking.xdeepfm import CIN
    with pytest.raises(ValueError, match=r"`CIN` layer's inputs type should be `tuple`."):      # test_xdeepfm.py:16-21
        CIN(feature_map=3)(np.random.normal(size=(2, 3, 5)).astype(np.float32))
    with pytest.raises(ValueError, match=r"`x0` and `x` dim should be 3."):                     # :23-28
        inputs = np.random.normal(size=(2, 15)).astype(np.float32)
        CIN(feature_map=3)((inputs, inputs))


@pytest.mark.parametrize("kat", ["cin_outputs", "cin_bias"])
def test_cin_reference_known_answers(kat):
    from deep_recommenders_amd.keras.models.ranking.xdeepfm import CIN
    g = G[kat]                                                                                  # test_xdeepfm.py:30-60
    x0, x = np.asarray(g["x0"], np.float32), np.asarray(g["x"], np.float32)
    layer = CIN(feature_map=g["feature_map"], activation="relu", kernel_init="ones",
                **(dict(use_bias=True, bias_init="ones") if kat == "cin_bias" else {}))
    out = layer((x0, x)).detach().cpu().numpy()
    np.testing.assert_allclose(out, np.asarray(g["expected"], np.float32), rtol=1e-6, atol=1e-6)   # assertAllClose defaults


@pytest.mark.parametrize("B,H0,Hk,D,Fm,act,bias", [(10, 12, 12, 10, 3, "sigmoid", False),     # the reference's train/save test shape
                                                    (33, 7, 5, 16, 70, "relu", True),
                                                    (64, 39, 40, 8, 100, "tanh", True),
                                                    (5, 3, 9, 64, 33, None, False)])
def test_cin_forward_backward_match_oracle(B, H0, Hk, D, Fm, act, bias):
    from deep_recommenders_amd import ops
    rng = np.random.default_rng(B + Fm)
    x0 = rng.standard_normal((B, H0, D)).astype(np.float32)
    x = rng.standard_normal((B, Hk, D)).astype(np.float32)
    W = (rng.standard_normal((H0 * Hk, Fm)) * 0.2).astype(np.float32)
    b = rng.standard_normal(Fm).astype(np.float32) if bias else None
    want = O.cin(x0, x, W, b, act)
    t = lambda a: None if a is None else torch.tensor(a).cuda()
    code = ops.ACT_CODES[act]
    out = ops.cin_fwd(t(x0), t(x), t(W), t(b), code)
    # fp32 MFMA over H0 * Hk terms: 1e-6 * sqrt(terms) of the pre-activation scale
    scale = np.abs(x0).max() * np.abs(x).max() * np.abs(W).max() * np.sqrt(H0 * Hk)
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-5, atol=2e-6 * scale)
    # backward against float64 autograd of the same formula
    dd = torch.float64
    X0, X, Wt = (torch.tensor(a, dtype=dd, requires_grad=True) for a in (x0, x, W))
    Bt = torch.tensor(b, dtype=dd, requires_grad=True) if bias else None
    pre = torch.einsum("bid,bjd,ijf->bfd", X0, X, Wt.reshape(H0, Hk, Fm))
    if bias:
        pre = pre + Bt[None, :, None]
    o = {"sigmoid": torch.sigmoid, "relu": torch.relu, "tanh": torch.tanh, None: (lambda v: v)}[act](pre)
    gout = rng.standard_normal((B, Fm, D)).astype(np.float32)
    o.backward(torch.tensor(gout, dtype=dd))
    d_x0, d_x, dW, dbias = ops.cin_bwd(t(x0), t(x), t(W), code, out, t(gout), want_bias=bias)
    for name, got, ref in (("d_x0", d_x0, X0.grad), ("d_x", d_x, X.grad), ("dW", dW, Wt.grad)) + ((("dbias", dbias, Bt.grad),) if bias else ()):
        ref = ref.numpy()
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=2e-4, atol=2e-5 * np.abs(ref).max(), err_msg=name)


def test_cin_stack_trains_like_the_reference_model():
    """test_xdeepfm.py:62-79: x = CIN(3)((x0, x0)); x = CIN(3)((x0, x)); Dense(1)(Flatten(x)), mse -- one SGD step lowers the loss and
    autograd reaches every parameter."""
    from deep_recommenders_amd.keras.models.ranking.xdeepfm import CIN
    rng = np.random.default_rng(0)
    x0 = torch.tensor(rng.uniform(size=(10, 12, 10)).astype(np.float32)).cuda()
    y = torch.tensor(rng.uniform(size=(10,)).astype(np.float32)).cuda()
    c1, c2 = CIN(feature_map=3), CIN(feature_map=3)
    head = torch.nn.Linear(30, 1).cuda()

    def loss_fn():
        h = c2((x0, c1((x0, x0))))
        return ((head(h.flatten(1)).squeeze(1) - y) ** 2).mean()
    l0 = loss_fn()
    l0.backward()
    params = list(c1.parameters()) + list(c2.parameters()) + list(head.parameters())
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in params)
    with torch.no_grad():
        for p in params:
            p -= 0.05 * p.grad
    assert loss_fn().item() < l0.item()
    assert c1.get_config()["feature_map"] == 3 and c1.get_config()["activation"] == "sigmoid"


@pytest.mark.parametrize("interact", [None, "subtract", "multiply"])
def test_activation_unit_matches_reference_test_and_oracle(interact):
    """tests/keras/test_din.py:17-48: kernel_init ones, ActivationUnit(10)(x, y) == reduce_sum(Dense(10, relu, ones)(concat[x, y(, x - y)]))."""
    from deep_recommenders_amd.keras.models.ranking import din
    rng = np.random.default_rng(7)
    x = rng.normal(size=(3, 5)).astype(np.float32)
    y = rng.normal(size=(3, 5)).astype(np.float32)
    inter = {None: None, "subtract": din.Subtract(), "multiply": din.Multiply()}[interact]
    unit = din.ActivationUnit(10, interacter=inter, kernel_init="ones")
    out = unit(x, y).detach().cpu().numpy()
    cols = [x, y] + ([x - y] if interact == "subtract" else [x * y] if interact == "multiply" else [])
    h = np.concatenate(cols, axis=1)
    expected = np.maximum(h @ np.ones((h.shape[1], 10), np.float32), 0).sum(axis=1, keepdims=True)
    np.testing.assert_allclose(out, expected, rtol=1e-6, atol=1e-6)
    o_inter = {None: None, "subtract": (lambda xy: xy[0] - xy[1]), "multiply": (lambda xy: xy[0] * xy[1])}[interact]
    want = O.activation_unit(x, y, np.ones((h.shape[1], 10), np.float32), np.zeros(10, np.float32), np.ones((10, 1), np.float32),
                             np.zeros(1, np.float32), o_inter)
    np.testing.assert_allclose(out, want, rtol=1e-6, atol=1e-6)


def test_activation_unit_gradients_and_defaults():
    from deep_recommenders_amd.keras.models.ranking import din
    rng = np.random.default_rng(3)
    x = torch.tensor(rng.normal(size=(64, 8)).astype(np.float32)).cuda().requires_grad_(True)
    y = torch.tensor(rng.normal(size=(64, 8)).astype(np.float32)).cuda().requires_grad_(True)
    unit = din.ActivationUnit(16, interacter=din.Subtract())
    out = unit(x, y)
    assert out.shape == (64, 1)
    with torch.no_grad():
        unit.dense_kernel_b.normal_(0, 0.1)
    out = unit(x, y)
    out.sum().backward()
    # float64 restatement under autograd
    dd = torch.float64
    X, Y = x.detach().cpu().to(dd).requires_grad_(True), y.detach().cpu().to(dd).requires_grad_(True)
    Wk, bk = unit.dense_kernel_w.detach().cpu().to(dd).requires_grad_(True), unit.dense_kernel_b.detach().cpu().to(dd).requires_grad_(True)
    Wo, bo = unit.dense_output_w.detach().cpu().to(dd).requires_grad_(True), unit.dense_output_b.detach().cpu().to(dd)
    h = torch.relu(torch.cat([X, Y, X - Y], 1) @ Wk + bk) @ Wo + bo
    h.sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), h.detach().numpy(), rtol=1e-5, atol=1e-6)
    for name, got, ref in (("dx", x.grad, X.grad), ("dy", y.grad, Y.grad), ("dWk", unit.dense_kernel_w.grad, Wk.grad),
                           ("dbk", unit.dense_kernel_b.grad, bk.grad), ("dWo", unit.dense_output_w.grad, Wo.grad)):
        np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-6, err_msg=name)
    # y defaults to x (din.py:59-60)
    u2 = din.ActivationUnit(4, kernel_init="ones")
    xs = x.detach()
    np.testing.assert_allclose(u2(xs).detach().cpu().numpy(), u2(xs, xs).detach().cpu().numpy())
    assert u2.get_config()["units"] == 4 and u2.get_config()["activation"] == "relu"
