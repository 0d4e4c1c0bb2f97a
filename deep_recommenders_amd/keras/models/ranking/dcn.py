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


def _l2_coeff(regu):
    """None -> 0; 'l2' -> Keras' default factor 0.01; a float -> that factor; {'l2': f} -> f; anything else -> None (rejected)"""
    if regu is None:
        return 0.0
    if regu == "l2":
        return 0.01
    if isinstance(regu, (int, float)):
        return float(regu)
    if isinstance(regu, dict) and set(regu) == {"l2"}:
        return float(regu["l2"])
    return None


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
        for r in (kernel_regu, bias_regu):
            if r is not None and _l2_coeff(r) is None:
                raise ValueError("regularizer must be None, 'l2', a float (the l2 factor) or {'l2': factor}; got %r" % (r,))
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
        # low-rank: prod = (x U) V + b (dcn.py:83), combined by dr_cross_fwd(W = NULL)
        return L.cross_low_rank(x0, x, self.kernel_u, self.kernel, self.bias, self._diag_scale or 0.0)

    forward = call

    @property
    def losses(self):
        """Regularization terms like a Keras layer's `.losses` (kernel_regularizer on every kernel, bias_regularizer on the
        bias; dcn.py:39-45,55-68): a list of scalar tensors to add to the training loss."""
        out = []
        if not self.built:
            return out
        kc, bc = _l2_coeff(self._kernel_regu), _l2_coeff(self._bias_regu)
        if kc:
            out.append(L.l2_penalty(self.kernel, kc))
            if self.kernel_u is not None:
                out.append(L.l2_penalty(self.kernel_u, kc))
        if bc and self.bias is not None:
            out.append(L.l2_penalty(self.bias, bc))
        return out

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
