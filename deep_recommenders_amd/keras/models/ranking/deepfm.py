"""DeepFM — same call/config surface as the reference's keras/models/ranking/deepfm.py:9-55."""
from typing import Dict, Optional

import torch
from torch import nn

from deep_recommenders_amd import layers as L
from deep_recommenders_amd import losses

_ACT = {"relu": 1, None: 0, "linear": 0, "sigmoid": 2, "tanh": 3}


class DeepFM(nn.Module):
    """DeepFM(indicator_columns, embedding_columns, dnn_units_size, dnn_activation="relu").call(inputs) -> prob.

    sigmoid( FM(indicator, stacked embeddings) + Sequential(Dense(u, act)..., Dense(1))(concat embeddings) )
    (reference deepfm.py:36-47).  `dense_features_key` is an extension for Criteo-shaped inputs (SURVEY.md
    §8d): a float [B, Nd] feature appended to the DNN input only; the reference has no numeric-column path."""

    def __init__(self, indicator_columns, embedding_columns, dnn_units_size, dnn_activation="relu",
                 dense_features_key: Optional[str] = None, device="cuda", **kwargs):
        super().__init__()
        if dnn_activation not in _ACT:
            raise ValueError("dnn_activation must be one of {}, got {!r}".format(sorted(k for k in _ACT if k), dnn_activation))
        self._indicator_columns = indicator_columns
        self._embedding_columns = embedding_columns
        self._dnn_units_size = list(dnn_units_size)
        self._dnn_activation = dnn_activation
        self._dense_key = dense_features_key
        self._kwargs = kwargs
        self.slab = L.EmbeddingSlab(embedding_columns, indicator_columns, device=device)
        self.dnn_kernels = nn.ParameterList()
        self.dnn_biases = nn.ParameterList()
        self._dnn_built = False

    def _build_dnn(self, in_dim, device):
        units = self._dnn_units_size + [1]                       # deepfm.py:30-34
        d = in_dim
        for u in units:                                          # [TF] B8: glorot-uniform kernel, zero bias
            W = torch.empty((d, u), dtype=torch.float32, device=device)
            L.glorot_uniform_(W)
            self.dnn_kernels.append(nn.Parameter(W))
            self.dnn_biases.append(nn.Parameter(torch.zeros(u, dtype=torch.float32, device=device)))
            d = u
        self._dnn_built = True

    def _field_keys(self, inputs: Dict[str, object]):
        return [k for k in inputs.keys() if k in self.slab.columns]   # deepfm.py:39 iterates inputs.items()

    def logits(self, inputs):
        keys = self._field_keys(inputs)
        FD = len(keys) * self.slab.D
        dense = None
        in_dim = FD
        if self._dense_key is not None:
            dense = torch.as_tensor(inputs[self._dense_key], dtype=torch.float32).to(self.slab.table.device)
            in_dim = FD + dense.shape[1]
        ld = L._pad4(in_dim)
        concat, fm_logit, _ = self.slab(inputs, keys, ld_concat=ld)
        if dense is not None:
            concat.data[:, FD:in_dim].copy_(dense)                 # layout only: append to the DNN input
        if not self._dnn_built:
            self._build_dnn(in_dim, concat.device)
        acts = [_ACT[self._dnn_activation]] * len(self._dnn_units_size) + [0]
        dnn_out = L.mlp(concat[:, :in_dim], list(self.dnn_kernels), list(self.dnn_biases), acts)
        return fm_logit.reshape(-1, 1) + dnn_out                   # deepfm.py:46

    def call(self, inputs, **kwargs):
        return losses.sigmoid(self.logits(inputs))                 # deepfm.py:47

    forward = call

    def predict(self, inputs):
        with torch.no_grad():
            return self.call(inputs).cpu().numpy()

    def get_config(self):
        config = {"dnn_units_size": self._dnn_units_size, "dnn_activation": self._dnn_activation}
        return {**self._kwargs, **config}
