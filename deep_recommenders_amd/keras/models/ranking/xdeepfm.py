"""xDeepFM's Compressed Interaction Network layer -- same constructor / call / get_config surface as the reference's
keras/models/ranking/xdeepfm.py:9-117, backed by dr_cin_fwd / dr_cin_bwd (deep_recommenders_amd/csrc/cin.hip)."""
from typing import Optional, Tuple

import torch
from torch import nn

from deep_recommenders_amd import ops
from deep_recommenders_amd.keras.models.ranking.dcn import _init


class _CinFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0, x, W, bias, act):
        out = ops.cin_fwd(x0, x, W, bias, act)
        ctx.act = act
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x0, x, W, out)
        return out

    @staticmethod
    def backward(ctx, d_out):
        x0, x, W, out = ctx.saved_tensors
        d_x0, d_x, dW, dbias = ops.cin_bwd(x0, x, W, ctx.act, out, d_out.contiguous(), want_bias=ctx.has_bias)
        return d_x0, d_x, dW, dbias, None


class CIN(nn.Module):
    """CIN(feature_map=3, use_bias=False, activation="sigmoid", kernel_init="truncated_normal", kernel_regu=None,
    bias_init="zeros", bias_regu=None)((x0, x)) -> [B, feature_map, D]   (xdeepfm.py:71-96)."""

    def __init__(self, feature_map: Optional[int] = 3, use_bias: bool = False, activation="sigmoid",
                 kernel_init="truncated_normal", kernel_regu=None, bias_init="zeros", bias_regu=None, **kwargs):
        super().__init__()
        self._feature_map = feature_map
        self._use_bias = use_bias
        if activation is not None and not callable(activation) and activation not in ops.ACT_CODES:
            raise ValueError("unknown activation {!r}; supported: {}".format(activation, sorted(k for k in ops.ACT_CODES if k)))
        self._activation = activation
        self._kernel_init, self._kernel_regu = kernel_init, kernel_regu
        self._bias_init, self._bias_regu = bias_init, bias_regu
        if kernel_regu is not None or bias_regu is not None:
            raise NotImplementedError("regularizers are not used by any reference model/test")
        self._kwargs = kwargs
        self.built = False

    def build(self, input_shape, device="cuda"):
        if not isinstance(input_shape, tuple):                                      # xdeepfm.py:41-44
            raise ValueError("`CIN` layer's inputs type should be `tuple`."
                             "Got `CIN` layer's inputs type = `{}`".format(type(input_shape)))
        if len(input_shape) != 2:                                                   # :46-48
            raise ValueError("`CIN` Layer inputs tuple length should be 2."
                             "Got `length` = {}".format(len(input_shape)))
        x0_shape, x_shape = input_shape
        self._x0_fields, self._x_fields = int(x0_shape[1]), int(x_shape[1])
        # conv1d kernel [1, H0 * Hk, feature_map] (:54-60), stored without the leading width-1 axis
        self.kernel = nn.Parameter(_init(self._kernel_init, (self._x0_fields * self._x_fields, self._feature_map), device))
        self.bias = nn.Parameter(_init(self._bias_init, (self._feature_map,), device)) if self._use_bias is True else None
        self.built = True

    def call(self, inputs: Tuple[torch.Tensor, torch.Tensor], **kwargs):
        if not isinstance(inputs, tuple):
            raise ValueError("`CIN` layer's inputs type should be `tuple`."
                             "Got `CIN` layer's inputs type = `{}`".format(type(inputs)))
        if len(inputs) != 2:
            raise ValueError("`CIN` Layer inputs tuple length should be 2."
                             "Got `length` = {}".format(len(inputs)))
        x0, x = (torch.as_tensor(t, dtype=torch.float32).cuda() for t in inputs)
        if x0.dim() != 3 or x.dim() != 3:                                           # :75-80
            raise ValueError("`x0` and `x` dim should be 3."
                             "Got `x0` dim = {}, `x` dim = {}".format(x0.dim(), x.dim()))
        if not self.built:
            self.build((tuple(x0.shape), tuple(x.shape)), x0.device)
        if callable(self._activation):                                              # a user-supplied layer: linear kernel + it
            return self._activation(_CinFn.apply(x0, x, self.kernel, self.bias, 0))
        return _CinFn.apply(x0, x, self.kernel, self.bias, ops.ACT_CODES[self._activation])

    forward = call

    def get_config(self):
        config = {
            "feature_map": self._feature_map,
            "use_bias": self._use_bias,
            "activation": self._activation,
            "kernel_init": self._kernel_init,
            "kernel_regu": self._kernel_regu,
            "bias_init": self._bias_init,
            "bias_regu": self._bias_regu,
        }
        return {**self._kwargs, **config}
