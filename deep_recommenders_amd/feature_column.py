"""The slice of `tf.feature_column` that the reference's hot-path classes consume.

The reference code only touches `col.categorical_column.key` (keras/models/ranking/fm.py:49,
deepfm.py:26), `col.name` with the `<key>_embedding` / `<key>_indicator` suffix convention
(estimator/models/feature_interaction/fm.py:49), `col.dimension` and `col.categorical_column`
(estimator/models/ranking/fnn.py:70-72); call sites that construct the columns:
examples/train_fm_on_movielens_estimator.py:10-34.  Semantics follow SURVEY.md Appendix B1-B4.

The id transformation itself (hash / vocabulary lookup) runs on the GPU through the C-ABI kernels
dr_hash_bucket_* / dr_vocab_lookup_* — these objects only carry configuration.
"""

import numpy as np
import torch

from . import ops


class CategoricalColumn:
    key: str
    num_buckets: int

    @property
    def name(self):
        return self.key

    def get_config(self):
        raise NotImplementedError

    # -- input normalisation ---------------------------------------------------------------
    @staticmethod
    def _as_2d_list(value):
        """host-side ragged / nested input -> list of rows (each a list)"""
        rows = []
        for v in value:
            if isinstance(v, (list, tuple, np.ndarray)):
                rows.append(list(np.asarray(v, dtype=object).reshape(-1)))
            else:
                rows.append([v])
        return rows

    def _is_string_input(self, value):
        if isinstance(value, torch.Tensor):
            return False
        arr = np.asarray(value, dtype=object) if not isinstance(value, np.ndarray) else value
        if arr.dtype.kind in ("U", "S"):
            return True
        if arr.dtype.kind == "O":
            flat = arr.reshape(-1)
            for v in flat:
                if isinstance(v, (list, tuple, np.ndarray)):
                    for w in np.asarray(v, dtype=object).reshape(-1):
                        return isinstance(w, (str, bytes))
                return isinstance(v, (str, bytes))
        return False

    def _int_matrix(self, value, device):
        """int input -> int64 [B, L] on device, padded with -1."""
        if isinstance(value, torch.Tensor):
            t = value.to(device=device, dtype=torch.int64)
            return t.reshape(t.shape[0], -1)
        arr = np.asarray(value)
        if arr.dtype.kind == "O":      # ragged
            rows = self._as_2d_list(value)
            L = max(1, max(len(r) for r in rows))
            out = np.full((len(rows), L), -1, dtype=np.int64)
            for i, r in enumerate(rows):
                out[i, :len(r)] = r
            arr = out
        arr = arr.astype(np.int64).reshape(arr.shape[0], -1)
        return torch.from_numpy(np.ascontiguousarray(arr)).to(device)

    def _string_rows(self, value):
        arr = np.asarray(value, dtype=object)
        if arr.ndim >= 2:
            return [[x for x in row.reshape(-1)] for row in arr]
        return self._as_2d_list(value)

    def ids(self, value, device="cuda") -> torch.Tensor:
        """feature value -> int64 ids [B, L] on `device`, -1 = missing/OOV."""
        raise NotImplementedError


class HashedCategoricalColumn(CategoricalColumn):
    """[TF] categorical_column_with_hash_bucket: id = Fingerprint64(as_string(x)) mod N (B1)."""

    def __init__(self, key, hash_bucket_size, dtype=str):
        if hash_bucket_size is None or hash_bucket_size < 1:
            raise ValueError("hash_bucket_size must be at least 1. hash_bucket_size: {}, key: {}".format(
                hash_bucket_size, key))
        self.key = key
        self.hash_bucket_size = int(hash_bucket_size)
        self.dtype = dtype
        self.num_buckets = self.hash_bucket_size

    def get_config(self):
        return {"key": self.key, "hash_bucket_size": self.hash_bucket_size}

    def ids(self, value, device="cuda"):
        if self._is_string_input(value):
            rows = self._string_rows(value)
            L = max(1, max(len(r) for r in rows))
            flat = []
            for r in rows:
                flat.extend(list(r) + [""] * (L - len(r)))      # "" is dropped by TF -> -1
            return ops.hash_bucket_strings(flat, self.hash_bucket_size, device).reshape(len(rows), L)
        keys = self._int_matrix(value, device)
        buckets = torch.full((keys.shape[1],), self.hash_bucket_size, dtype=torch.int64, device=device)
        return ops.hash_bucket_i64(keys, buckets)


class VocabularyListCategoricalColumn(CategoricalColumn):
    """[TF] categorical_column_with_vocabulary_list, default_value=-1, num_oov_buckets=0 (B2)."""

    def __init__(self, key, vocabulary_list, dtype=None, default_value=-1, num_oov_buckets=0):
        if vocabulary_list is None or len(vocabulary_list) < 1:
            raise ValueError("vocabulary_list {} must be non-empty, column_name: {}".format(vocabulary_list, key))
        if len(set(vocabulary_list)) != len(vocabulary_list):
            raise ValueError("Duplicate keys in vocabulary_list: {}, column_name: {}".format(vocabulary_list, key))
        if default_value != -1 or num_oov_buckets != 0:
            raise NotImplementedError("only default_value=-1 / num_oov_buckets=0 (what the reference uses)")
        self.key = key
        self.vocabulary_list = tuple(vocabulary_list)
        self.default_value = default_value
        self.num_oov_buckets = num_oov_buckets
        self.num_buckets = len(self.vocabulary_list)
        self._is_str = isinstance(self.vocabulary_list[0], (str, bytes))
        self._dev_vocab = {}

    def get_config(self):
        return {"key": self.key, "vocabulary_list": list(self.vocabulary_list)}

    def ids(self, value, device="cuda"):
        if self._is_str:
            rows = self._string_rows(value)
            L = max(1, max(len(r) for r in rows))
            flat = []
            for r in rows:
                flat.extend(list(r) + [""] * (L - len(r)))
            return ops.vocab_lookup_strings(flat, self.vocabulary_list, device).reshape(len(rows), L)
        keys = self._int_matrix(value, device)
        dv = self._dev_vocab.get(str(device))
        if dv is None:
            dv = torch.tensor(self.vocabulary_list, dtype=torch.int64, device=device)
            self._dev_vocab[str(device)] = dv
        return ops.vocab_lookup_i64(keys, dv)


class IdentityCategoricalColumn(CategoricalColumn):
    """Ids already in [0, num_buckets) (used for pre-hashed synthetic batches); -1 = missing."""

    def __init__(self, key, num_buckets):
        self.key = key
        self.num_buckets = int(num_buckets)

    def get_config(self):
        return {"key": self.key, "num_buckets": self.num_buckets}

    def ids(self, value, device="cuda"):
        ids = self._int_matrix(value, device)
        # [TF] categorical_column_with_identity asserts 0 <= id < num_buckets (negative ids mean "missing" here, as in the
        # hashed / vocabulary columns).  The gather / scatter kernels trust the range: an id past num_buckets would read or
        # update another field's rows of the shared slab, so it is checked here (one small reduction + host read).
        if ids.numel() and int(ids.max()) >= self.num_buckets:
            raise ValueError("IdentityCategoricalColumn(%r): id %d is out of range [0, %d)"
                             % (self.key, int(ids.max()), self.num_buckets))
        return ids


class IndicatorColumn:
    """[TF] indicator_column: multi-hot count vector (B3); `.name == "<key>_indicator"`."""

    def __init__(self, categorical_column):
        self.categorical_column = categorical_column

    @property
    def name(self):
        return "{}_indicator".format(self.categorical_column.name)

    @property
    def variable_shape(self):
        return (self.categorical_column.num_buckets,)

    def get_config(self):
        return {"categorical_column": self.categorical_column.get_config()}


class EmbeddingColumn:
    """[TF] embedding_column (B4): table [num_buckets, dimension], truncated-normal(0, 1/sqrt(dim)),
    combiner 'mean'; `.name == "<key>_embedding"`."""

    def __init__(self, categorical_column, dimension, combiner="mean", initializer=None, max_norm=None,
                 trainable=True):
        if dimension is None or dimension < 1:
            raise ValueError("Invalid dimension {}.".format(dimension))
        if combiner != "mean":
            raise NotImplementedError("only combiner='mean' (the default the reference relies on)")
        if max_norm is not None:
            raise NotImplementedError("max_norm is not used by the reference")
        self.categorical_column = categorical_column
        self.dimension = int(dimension)
        self.combiner = combiner
        self.initializer = initializer
        self.trainable = trainable

    @property
    def name(self):
        return "{}_embedding".format(self.categorical_column.name)

    @property
    def variable_shape(self):
        return (self.dimension,)

    def get_config(self):
        return {"categorical_column": self.categorical_column.get_config(), "dimension": self.dimension,
                "combiner": self.combiner}


def categorical_column_with_hash_bucket(key, hash_bucket_size, dtype=str):
    return HashedCategoricalColumn(key, hash_bucket_size, dtype)


def categorical_column_with_vocabulary_list(key, vocabulary_list, dtype=None, default_value=-1, num_oov_buckets=0):
    return VocabularyListCategoricalColumn(key, vocabulary_list, dtype, default_value, num_oov_buckets)


def categorical_column_with_identity(key, num_buckets):
    return IdentityCategoricalColumn(key, num_buckets)


def indicator_column(categorical_column):
    return IndicatorColumn(categorical_column)


def embedding_column(categorical_column, dimension, combiner="mean", initializer=None, max_norm=None, trainable=True):
    return EmbeddingColumn(categorical_column, dimension, combiner, initializer, max_norm, trainable)
