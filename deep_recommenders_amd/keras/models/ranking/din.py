"""DIN's ActivationUnit -- same constructor / call / get_config surface as the reference's
keras/models/ranking/din.py:8-88: Dense(1)(Dense(units, activation)(concat([x, y, interacter([x, y])], axis=1))).
The concat (+ Subtract / Multiply interaction) is one kernel (dr_din_concat_fwd), the two Dense layers are the MFMA GEMM path
(deep_recommenders_amd.layers.mlp).  `Dice` (din.py:91-130) is outside SURVEY.md section 8 and not provided."""
import torch
from torch import nn

from deep_recommenders_amd import layers as L
from deep_recommenders_amd import ops
from deep_recommenders_amd.keras.models.ranking.dcn import _init


class Subtract:
    """keras.layers.Subtract restricted to what ActivationUnit feeds it: [x, y] -> x - y (tests/keras/test_din.py:33)."""
    mode = 1

    def __call__(self, inputs):
        x, y = inputs
        return _DinConcatFn.apply(x, y, 1)[:, 2 * x.shape[1]:]


class Multiply:
    """keras.layers.Multiply: [x, y] -> x * y."""
    mode = 2

    def __call__(self, inputs):
        x, y = inputs
        return _DinConcatFn.apply(x, y, 2)[:, 2 * x.shape[1]:]


class _DinConcatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, mode):
        ctx.mode = mode
        ctx.save_for_backward(x, y)
        return ops.din_concat_fwd(x, y, mode)

    @staticmethod
    def backward(ctx, d_out):
        x, y = ctx.saved_tensors
        if d_out.stride(1) != 1:
            d_out = d_out.contiguous()
        d_x, d_y = ops.din_concat_bwd(x, y, ctx.mode, d_out)
        return d_x, d_y, None


class ActivationUnit(nn.Module):
    def __init__(self, units, interacter=None, use_bias=True, activation="relu", kernel_init="truncated_normal",
                 kernel_regu=None, bias_init="zeros", bias_regu=None, **kwargs):
        super().__init__()
        self._kernel_units = units
        self._interacter = interacter
        self._use_bias = use_bias
        if activation not in ("relu", None, "linear"):
            raise NotImplementedError("ActivationUnit activation %r: the GEMM epilogue provides relu / linear" % (activation,))
        self._kernel_activation = activation
        self._kernel_init, self._kernel_regu = kernel_init, kernel_regu
        self._bias_init, self._bias_regu = bias_init, bias_regu
        if kernel_regu is not None or bias_regu is not None:
            raise NotImplementedError("regularizers are not used by any reference model/test")
        self._kwargs = kwargs
        self.built = False

    def build(self, in_dim, device="cuda"):
        u = self._kernel_units
        self.dense_kernel_w = nn.Parameter(_init(self._kernel_init, (in_dim, u), device))       # din.py:37-45
        self.dense_output_w = nn.Parameter(_init(self._kernel_init, (u, 1), device))            # :46-54
        self.dense_kernel_b = nn.Parameter(_init(self._bias_init, (u,), device)) if self._use_bias else None
        self.dense_output_b = nn.Parameter(_init(self._bias_init, (1,), device)) if self._use_bias else None
        self.built = True

    def call(self, x_embeddings, y_embeddings=None, **kwargs):
        x = torch.as_tensor(x_embeddings, dtype=torch.float32).cuda()
        y = x if y_embeddings is None else torch.as_tensor(y_embeddings, dtype=torch.float32).cuda()      # din.py:59-60
        mode = getattr(self._interacter, "mode", None) if self._interacter is not None else 0
        if mode is not None:
            h = _DinConcatFn.apply(x, y, mode)                                                           # :62-66 in one pass
        else:   # a user-supplied interacter: its output is appended as a third column block
            h = torch.cat([_DinConcatFn.apply(x, y, 0), self._interacter([x, y])], dim=1)
        if not self.built:
            self.build(h.shape[1], h.device)
        act = 1 if self._kernel_activation == "relu" else 0
        return L.mlp(h, [self.dense_kernel_w, self.dense_output_w], [self.dense_kernel_b, self.dense_output_b], [act, 0])   # :68-69

    forward = call

    def get_config(self):
        config = {
            "units": self._kernel_units,
            "interacter": self._interacter,
            "use_bias": self._use_bias,
            "activation": self._kernel_activation,
            "kernel_init": self._kernel_init,
            "kernel_regu": self._kernel_regu,
            "bias_init": self._bias_init,
            "bias_regu": self._bias_regu,
        }
        return {**self._kwargs, **config}
