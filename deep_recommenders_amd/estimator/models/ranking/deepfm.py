"""Estimator-style DeepFM — same surface as the reference's estimator/models/ranking/deepfm.py:9-43."""
from torch import nn

from deep_recommenders_amd import losses
from deep_recommenders_amd.estimator.models.feature_interaction import FM
from deep_recommenders_amd.estimator.models.feature_interaction.dnn import VariableStore, dnn, relu


class DeepFM(nn.Module):
    """DeepFM(indicator_columns, embedding_columns, dnn_units, dnn_activation=relu, dnn_batch_normalization=False,
    dnn_dropout=None, **dnn_kwargs)(features) -> sigmoid(fm_outputs + dnn_outputs)  (deepfm.py:30-43)."""

    def __init__(self, indicator_columns, embedding_columns, dnn_units, dnn_activation=relu,
                 dnn_batch_normalization=False, dnn_dropout=None, device="cuda", **dnn_kwargs):
        super().__init__()
        self._indicator_columns = indicator_columns
        self._embedding_columns = embedding_columns
        self._dnn_hidden_units = list(dnn_units)
        self._dnn_activation = dnn_activation
        self._dnn_batch_norm = dnn_batch_normalization
        self._dnn_dropout = dnn_dropout
        self._dnn_kwargs = dnn_kwargs
        # the reference builds a fresh FM(...) inside call() under TF variable scopes that make the
        # variables persistent; the object-owned equivalents:
        self.fm = FM(indicator_columns, embedding_columns, device=device)
        self.store = VariableStore()

    def logits(self, features):
        fm_outputs = self.fm(features)                                    # deepfm.py:31-33
        F, D = len(self.fm.embeddings), self.fm.slab.D
        concat_embeddings = self.fm.concat_embeddings[:, :F * D]          # deepfm.py:34 (zero-copy)
        dnn_outputs = dnn(concat_embeddings, self._dnn_hidden_units + [1], activation=self._dnn_activation,
                          batch_normalization=self._dnn_batch_norm, dropout=self._dnn_dropout, store=self.store,
                          **self._dnn_kwargs)                             # deepfm.py:36-41
        return fm_outputs + dnn_outputs

    def call(self, features):
        return losses.sigmoid(self.logits(features))                      # deepfm.py:43

    forward = call
