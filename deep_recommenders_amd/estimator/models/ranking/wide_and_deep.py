"""Estimator-style Wide & Deep -- same surface as the reference's estimator/models/ranking/wide_and_deep.py:9-48.

wide: tf.feature_column.linear_model over the indicator columns = first-order term  bias + sum w[id]  (multi-hot sums);
deep: every embedding column's input_layer, concatenated, through dnn(units + [1]); output sigmoid(wide + deep).
Both parts come out of ONE launch of the fused gather kernel (first-order-only logit, DR_POOL_FIRST_ORDER_ONLY)."""
from torch import nn

from deep_recommenders_amd import layers as L
from deep_recommenders_amd import losses
from deep_recommenders_amd.estimator.models import variables as V
from deep_recommenders_amd.estimator.models.feature_interaction.dnn import VariableStore, dnn, relu


class WDL(nn.Module):
    """WDL(indicator_columns, embedding_columns, dnn_units, dnn_activation=relu, dnn_batch_normalization=False,
    dnn_dropout=None, **dnn_kwargs)(features) -> sigmoid(linear_outputs + dnn_outputs)  (wide_and_deep.py:29-48)."""

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
        self.slab = L.EmbeddingSlab(embedding_columns, indicator_columns, device=device)
        self.store = VariableStore()

    def logits(self, features):
        keys = [c.name.replace("_embedding", "") for c in self._embedding_columns]          # wide_and_deep.py:36
        concat, linear_outputs, _ = self.slab(features, keys, second_order=False)            # "wide" (:30-32) + embeddings
        F, D = len(keys), self.slab.D
        dnn_outputs = dnn(concat[:, :F * D], self._dnn_hidden_units + [1], activation=self._dnn_activation,
                          batch_normalization=self._dnn_batch_norm, dropout=self._dnn_dropout, store=self.store,
                          **self._dnn_kwargs)                                                  # "deep" (:41-46)
        return linear_outputs.reshape(-1, 1) + dnn_outputs

    def call(self, features):
        return losses.sigmoid(self.logits(features))                                          # :48

    forward = call

    # TF1 names: wide/linear_model/<key>_indicator/weights, wide/linear_model/bias_weights,
    #            deep/input_layer/<key>_embedding/embedding_weights, deep/dense[_i]/{kernel,bias}
    def export_variables(self):
        out = V.export_slab(self.slab, linear_scope="wide", factorized_scope="deep")
        out.update(V.export_store(self.store, "dnn", "deep"))
        return out

    def import_variables(self, variables, strict=True):
        used = V.import_slab(self.slab, variables, linear_scope="wide", factorized_scope="deep", strict=strict)
        return used + V.import_store(self.store, variables, "deep", "dnn", strict=strict)
