"""FM layer and FactorizationMachine model — same call/config surface as the reference's
keras/models/ranking/fm.py (FM :8-37, FactorizationMachine :40-72), backed by the gfx950 kernels."""
from typing import Dict

import torch
from torch import nn

from deep_recommenders_amd import layers as L
from deep_recommenders_amd import losses


class FM(nn.Module):
    """Factorization Machine layer.  call(sparse_inputs[B, SV], embedding_inputs[B, F, D]=None) -> [B, 1]
    = Dense(1, kernel_initializer="zeros")(sparse_inputs) [+ 0.5 * sum_d((sum_f x)^2 - sum_f x^2)]."""

    def __init__(self, **kwargs):
        super().__init__()
        self._kwargs = kwargs
        self.built = False

    def build(self, input_shape, device="cuda"):
        sv = int(input_shape[-1])
        # reference: Dense(units=1, kernel_initializer="zeros") -> zero kernel, zero bias (fm.py:16-20)
        self.linear_kernel = nn.Parameter(torch.zeros((sv, 1), dtype=torch.float32, device=device))
        self.linear_bias = nn.Parameter(torch.zeros(1, dtype=torch.float32, device=device))
        self.built = True

    def call(self, sparse_inputs, embedding_inputs=None, **kwargs):
        sparse_inputs = torch.as_tensor(sparse_inputs, dtype=torch.float32).cuda()
        if not self.built:
            self.build(sparse_inputs.shape, sparse_inputs.device)
        linear = L.mlp(sparse_inputs, [self.linear_kernel], [self.linear_bias], [0])
        if embedding_inputs is None:
            return linear                                      # fm.py:25-26
        embedding_inputs = torch.as_tensor(embedding_inputs, dtype=torch.float32).cuda()
        return linear + L.fm_second_order(embedding_inputs)   # fm.py:28-37

    forward = call

    def get_config(self):
        return dict(self._kwargs)


class FactorizationMachine(nn.Module):
    """FactorizationMachine(indicator_columns, embedding_columns).call(inputs: dict) -> sigmoid prob [B, 1].

    Reference pipeline (fm.py:54-64): DenseFeatures(indicator) -> per-input DenseFeatures(embedding) ->
    stack -> FM -> sigmoid.  Here: one fused gather+pool+first-order+FM kernel, then the sigmoid kernel.
    As in the reference, fields are taken in `inputs.items()` order (fm.py:57)."""

    def __init__(self, indicator_columns, embedding_columns, device="cuda", **kwargs):
        super().__init__()
        self._indicator_columns = indicator_columns
        self._embedding_columns = embedding_columns
        self._kwargs = kwargs
        self.slab = L.EmbeddingSlab(embedding_columns, indicator_columns, device=device)

    def _field_keys(self, inputs: Dict[str, object]):
        return [k for k in inputs.keys() if k in self.slab.columns]   # dict insertion order

    def logits(self, inputs):
        _, fm_logit, _ = self.slab(inputs, self._field_keys(inputs))
        return fm_logit.reshape(-1, 1)

    def call(self, inputs, training=None, mask=None):
        return losses.sigmoid(self.logits(inputs))

    forward = call

    def predict(self, inputs):
        with torch.no_grad():
            return self.call(inputs).cpu().numpy()

    def get_config(self):
        config = {"indicator_columns": self._indicator_columns, "embedding_columns": self._embedding_columns}
        return {**self._kwargs, **config}
