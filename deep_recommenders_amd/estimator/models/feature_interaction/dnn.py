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


def relu(x):           # activation token: the fused Dense kernels implement relu in their epilogue
    return x


def dnn(inputs, hidden_units, activation=relu, batch_normalization=False, dropout=None, store=None, scope="dnn",
        **kwargs):
    """x -> Dense(u, activation) for u in hidden_units[:-1] -> Dense(hidden_units[-1]) (dnn.py:17-29).

    Reference quirks kept: batch_normalization=True raises (the reference calls
    tf.nn.batch_normalization with missing arguments, dnn.py:23-24); dropout has no train/eval switch and
    is not supported by the fused kernels."""
    if batch_normalization is True:
        raise TypeError("batch_normalization() missing required arguments (the reference's dnn.py:23-24 "
                        "calls tf.nn.batch_normalization(x) and raises too)")
    if dropout is not None:
        raise NotImplementedError("dropout inside dnn() is always-on in the reference (dnn.py:26-27); unsupported")
    if activation not in (relu, None) and getattr(activation, "__name__", "") != "relu":
        raise ValueError("only relu / None activations are fused")
    store = store if store is not None else _DEFAULT_STORE
    x = torch.as_tensor(inputs, dtype=torch.float32).cuda()
    kernels, biases, acts = [], [], []
    d = x.shape[1]
    for i, units in enumerate(hidden_units):
        name = "{}/dense{}".format(scope, "" if i == 0 else "_%d" % i)
        kernels.append(store.get(name + "/kernel", (d, units), L.glorot_uniform_, x.device))     # [TF] B8
        biases.append(store.get(name + "/bias", (units,), torch.nn.init.zeros_, x.device))
        acts.append(1 if (i < len(hidden_units) - 1 and activation is not None) else 0)
        d = units
    return L.mlp(x, kernels, biases, acts)
