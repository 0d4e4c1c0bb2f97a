"""DCN cross layer — same constructor / call / get_config surface as the reference's
keras/models/ranking/dcn.py:9-108."""
from typing import Optional

import torch
from torch import nn

from deep_recommenders_amd import layers as L

_INITS = ("truncated_normal", "zeros", "ones", "glorot_uniform")


def _init(name, shape, device):
    t = torch.empty(shape, dtype=torch.float32, device=device)
    if callable(name):
        name(t)
    elif name == "truncated_normal":          # Keras string alias: TruncatedNormal(mean=0, stddev=0.05)  [TF] B8
        L.truncated_normal_(t, 0.05)
    elif name == "zeros":
        t.zero_()
    elif name == "ones":
        t.fill_(1.0)
    elif name == "glorot_uniform":
        L.glorot_uniform_(t) if t.dim() == 2 else t.zero_()
    else:
        raise ValueError("unknown initializer {!r}; supported: {} or a callable".format(name, _INITS))
    return t


class Cross(nn.Module):
    """Cross(projection_dim=None, diag_scale=0.0, use_bias=True, kernel_init="truncated_normal", kernel_regu=None,
    bias_init="zeros", bias_regu=None).call(x0, x=None) = x0 * (W x + b + diag_scale * x) + x   (dcn.py:70-88)."""

    def __init__(self, projection_dim: Optional[int] = None, diag_scale: Optional[float] = 0.0, use_bias: bool = True,
                 kernel_init="truncated_normal", kernel_regu=None, bias_init="zeros", bias_regu=None, **kwargs):
        super().__init__()
        self._projection_dim = projection_dim
        self._diag_scale = diag_scale
        self._use_bias = use_bias
        self._kernel_init = kernel_init
        self._kernel_regu = kernel_regu
        self._bias_init = bias_init
        self._bias_regu = bias_regu
        self._kwargs = kwargs
        if kernel_regu is not None or bias_regu is not None:
            raise NotImplementedError("regularizers are not used by any reference model/test")
        assert self._diag_scale >= 0, ValueError(
            "diag scale must be non-negative, got {}".format(self._diag_scale))       # dcn.py:32-33
        self.built = False

    def build(self, input_shape, device="cuda"):
        last_dim = int(input_shape[-1])
        if self._projection_dim is None:
            self.kernel = nn.Parameter(_init(self._kernel_init, (last_dim, last_dim), device))
            self.kernel_u = None
        else:
            if self._projection_dim < 0 or self._projection_dim > last_dim / 2:       # dcn.py:48-53
                raise ValueError(
                    "`projection_dim` should be smaller than last_dim / 2 to improve "
                    "the model efficiency, and should be positive. Got "
                    "`projection_dim` {}, and last dimension of input {}".format(self._projection_dim, last_dim))
            self.kernel_u = nn.Parameter(_init(self._kernel_init, (last_dim, self._projection_dim), device))
            self.kernel = nn.Parameter(_init(self._kernel_init, (self._projection_dim, last_dim), device))
        self.bias = nn.Parameter(_init(self._bias_init, (last_dim,), device)) if self._use_bias else None
        self.built = True

    def call(self, x0, x=None, **kwargs):
        x0 = torch.as_tensor(x0, dtype=torch.float32).cuda()
        if x is None:
            x = x0                                                                     # dcn.py:72-73
        else:
            x = torch.as_tensor(x, dtype=torch.float32).cuda()
        if x0.shape[-1] != x.shape[-1]:                                                # dcn.py:75-78
            raise ValueError("`x0` and `x` dim mismatch. "
                             "Got `x0` dim = {} and `x` dim = {}".format(x0.shape[-1], x.shape[-1]))
        if not self.built:
            self.build(x0.shape, x0.device)
        if self._projection_dim is None:
            return L.cross(x0, x, self.kernel, self.bias, self._diag_scale or 0.0)     # dcn.py:81,85-88
        # low-rank: prod = Dense_v(Dense_u(x)) (dcn.py:83); the combine is elementwise in autograd-visible form
        u = L.mlp(x, [self.kernel_u], [None], [0])
        prod = L.mlp(u, [self.kernel], [self.bias], [0])
        if self._diag_scale:
            prod = prod + self._diag_scale * x
        return x0 * prod + x

    forward = call

    def get_config(self):
        config = {
            "projection_dim": self._projection_dim,
            "diag_scale": self._diag_scale,
            "use_bias": self._use_bias,
            "kernel_init": self._kernel_init,
            "kernel_regu": self._kernel_regu,
            "bias_init": self._bias_init,
            "bias_regu": self._bias_regu,
        }
        return {**self._kwargs, **config}
