"""Estimator-style FM — same surface as the reference's estimator/models/feature_interaction/fm.py
(`fm` :10-26, `FM` :29-56).  TF1 variable scopes become parameters owned by the object."""
import torch
from torch import nn

from deep_recommenders_amd import layers as L


def fm(x):
    """Second order interaction in Factorization Machine.  x: (batch_size, num_features, embedding_dim)."""
    x = torch.as_tensor(x, dtype=torch.float32)
    if x.dim() != 3:                                                           # fm.py:19-20
        raise ValueError("The rank of `x` should be 3. Got rank = {}.".format(x.dim()))
    return L.fm_second_order(x.cuda())


class FM(nn.Module):
    """FM(indicator_columns, embedding_columns)(features) -> LOGITS [B, 1] (no sigmoid, fm.py:56);
    side effect: `.embeddings` = list of F [B, D] tensors in `embedding_columns` order (fm.py:47-52)."""

    def __init__(self, indicator_columns, embedding_columns, device="cuda"):
        super().__init__()
        self._indicator_columns = indicator_columns
        self._embedding_columns = embedding_columns
        self.slab = L.EmbeddingSlab(embedding_columns, indicator_columns, device=device)
        self.embeddings = []
        self.concat_embeddings = None

    def call(self, features, ld_concat=None):
        # fm.py:48-50: feature name = embedding column name minus the "_embedding" suffix, in column order
        keys = [c.name.replace("_embedding", "") for c in self._embedding_columns]
        concat, fm_logit, _ = self.slab(features, keys, ld_concat=ld_concat)
        D = self.slab.D
        self.concat_embeddings = concat                      # == tf.concat(self.embeddings, axis=1), zero-copy
        self.embeddings = [concat[:, f * D:(f + 1) * D] for f in range(len(keys))]
        return fm_logit.reshape(-1, 1)                       # linear_outputs + factorized_outputs

    forward = call

    # TF1 names of fm.py:43-52: linear/linear_model/<key>_indicator/weights, linear/linear_model/bias_weights,
    # factorized/input_layer/<key>_embedding/embedding_weights -- the contract FNN.warm_up reads (ranking/fnn.py:32-48)
    def export_variables(self):
        from deep_recommenders_amd.estimator.models import variables as V
        return V.export_slab(self.slab)

    def import_variables(self, variables, strict=True):
        from deep_recommenders_amd.estimator.models import variables as V
        return V.import_slab(self.slab, variables, strict=strict)

    def save_variables(self, path):
        import numpy as np
        np.savez(path, **self.export_variables())
