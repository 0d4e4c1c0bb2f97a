"""Estimator-style MLP — same signature as the reference's estimator/models/feature_interaction/dnn.py:9-31.

TF1 `tf.layers.dense` creates variables in the ambient variable scope; here they live in a VariableStore
(AUTO_REUSE semantics: the same scope + layer index returns the same parameters on the next call)."""
import torch
from torch import nn

from deep_recommenders_amd import layers as L


class VariableStore(nn.Module):
    def __init__(self):
        super().__init__()
        self.vars = nn.ParameterDict()

    def get(self, name, shape, init, device):
        key = name.replace("/", "__")
        if key not in self.vars:
            t = torch.empty(shape, dtype=torch.float32, device=device)
            init(t)
            self.vars[key] = nn.Parameter(t)
        p = self.vars[key]
        if tuple(p.shape) != tuple(shape):
            raise ValueError("variable {} exists with shape {}, requested {}".format(name, tuple(p.shape), shape))
        return p


_DEFAULT_STORE = VariableStore()


def default_store():
    return _DEFAULT_STORE


def relu(x):           # activation tokens: the Dense kernels implement them (relu in the GEMM epilogue)
    return x


def sigmoid(x):
    return x


def tanh(x):
    return x


_ACT_CODES = {"relu": 1, "sigmoid": 2, "tanh": 3}
_dropout_calls = [0]


def dnn(inputs, hidden_units, activation=relu, batch_normalization=False, dropout=None, store=None, scope="dnn", seed=0,
        **kwargs):
    """x -> [Dense(u, activation) -> dropout] for u in hidden_units[:-1] -> Dense(hidden_units[-1]) (dnn.py:17-29).

    Reference quirks kept: batch_normalization=True raises (the reference calls tf.nn.batch_normalization with
    missing arguments, dnn.py:23-24); dropout has NO train / eval switch (dnn.py:26-27: `tf.nn.dropout(x, rate=dropout)` on every
    call) -- every call draws a fresh mask (counter-based; `seed` offsets the stream, TF's own stream is not reproducible)."""
    if batch_normalization is True:
        raise TypeError("batch_normalization() missing required arguments (the reference's dnn.py:23-24 "
                        "calls tf.nn.batch_normalization(x) and raises too)")
    if activation is None:
        act = 0
    else:
        name = activation if isinstance(activation, str) else getattr(activation, "__name__", "")
        if name not in _ACT_CODES:
            raise ValueError("activation must be relu / sigmoid / tanh / None, got {!r}".format(activation))
        act = _ACT_CODES[name]
    if dropout is not None and not (0.0 <= float(dropout) < 1.0):
        raise ValueError("dropout rate must be in [0, 1), got {}".format(dropout))
    store = store if store is not None else _DEFAULT_STORE
    x = torch.as_tensor(inputs, dtype=torch.float32).cuda()
    n = len(hidden_units)
    d = x.shape[1]
    params = []
    for i, units in enumerate(hidden_units):
        name = "{}/dense{}".format(scope, "" if i == 0 else "_%d" % i)
        params.append((store.get(name + "/kernel", (d, units), L.glorot_uniform_, x.device),                 # [TF] B8
                       store.get(name + "/bias", (units,), torch.nn.init.zeros_, x.device)))
        d = units
    if dropout is None:
        return L.mlp(x, [k for k, _ in params], [b for _, b in params], [act] * (n - 1) + [0])
    for i, (k, b) in enumerate(params):          # dropout sits between the layers: one Dense launch sequence per layer
        x = L.mlp(x, [k], [b], [act if i < n - 1 else 0])
        if i < n - 1:
            _dropout_calls[0] += 1
            x = L.dropout(x, dropout, seed * 1000003 + _dropout_calls[0])
    return x
