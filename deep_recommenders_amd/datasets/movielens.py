"""Input side of the hot path: the reference's MovieLens TFRecord pipeline without TensorFlow.

Mirrors deep_recommenders/datasets/movielens.py of the reference:
  MovieLens          :96-131   (file name, column list, vocabularies, `dataset(epochs, batch_size)`)
  MovielensRanking   :134-186  (epochs 10, batch 1024, train 0.8; step counts; `input_fn` renames the six model features
                                and labels = rating > 3 as float32 [B, 1])
tf.data.TFRecordDataset + tf.io.parse_example are replaced by the native reader of include/dr_input.h
(deep_recommenders_amd/csrc_host/tfrecord_reader.cpp); a dataset is a Python generator of (features, ratings).
String features arrive as `BytesColumn` (one byte blob + offsets [+ CSR row splits for VarLenFeature]) -- the layout
dr_hash_bucket_bytes / dr_vocab_lookup_bytes take -- and convert to (nested) lists of bytes for the feature columns.
Writing the file (`serialize_tfrecords`, :54-92) needs the MovieLens-1M download and is not reproduced; tests write
synthetic files of the same schema with an independent encoder (oracle/tfrecord_py.py).
"""
import ctypes
import os

import numpy as np

from .. import _input_lib as L


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


class TFRecordFile:
    """Index of a TFRecord file: payload offset / length of every record (CRCs verified once, at open)."""

    def __init__(self, path, verify_crc=True):
        self.path = os.fspath(path)
        bpath = self.path.encode()
        cnt = ctypes.c_int64(0)
        rc = L.lib().dri_tfrecord_index(bpath, 1 if verify_crc else 0, None, None, 0, ctypes.byref(cnt))
        if rc not in (L.DRI_OK, L.DRI_ECAPACITY):
            L.check(rc, "dri_tfrecord_index(%s)" % self.path)
        n = cnt.value
        self.offsets = np.empty(n, dtype=np.int64)
        self.lengths = np.empty(n, dtype=np.int64)
        if n:
            L.check(L.lib().dri_tfrecord_index(bpath, 0, _ptr(self.offsets), _ptr(self.lengths), n, ctypes.byref(cnt)),
                    "dri_tfrecord_index(%s)" % self.path)

    def __len__(self):
        return len(self.offsets)

    def read(self, index):
        """Serialized records `index` (array of record numbers) -> (records uint8[total], rec_offsets int64[n+1])."""
        index = np.asarray(index, dtype=np.int64)
        offs = np.ascontiguousarray(self.offsets[index])
        lens = np.ascontiguousarray(self.lengths[index])
        out = np.empty(int(lens.sum()), dtype=np.uint8)
        rec_offsets = np.empty(len(index) + 1, dtype=np.int64)
        L.check(L.lib().dri_tfrecord_read(self.path.encode(), _ptr(offs), _ptr(lens), len(index), _ptr(out), out.size,
                                          _ptr(rec_offsets)), "dri_tfrecord_read")
        return out, rec_offsets


def parse_int64(records, rec_offsets, key):
    """tf.io.FixedLenFeature([], tf.int64) (movielens.py:119-120 of the reference)."""
    n = len(rec_offsets) - 1
    out = np.empty(n, dtype=np.int64)
    L.check(L.lib().dri_example_int64(_ptr(records), _ptr(rec_offsets), n, key.encode(), _ptr(out)),
            "dri_example_int64(%r)" % key)
    return out


class BytesColumn:
    """A string feature of a batch: `blob` (uint8), `value_offsets` (int64[values+1]), `row_splits` (int64[n+1])."""

    def __init__(self, blob, value_offsets, row_splits, ragged):
        self.blob, self.value_offsets, self.row_splits, self.ragged = blob, value_offsets, row_splits, ragged

    def __len__(self):
        return len(self.row_splits) - 1

    def values(self):
        b, o = self.blob.tobytes(), self.value_offsets
        return [b[o[i]:o[i + 1]] for i in range(len(o) - 1)]

    def to_list(self):
        """list[bytes] (FixedLenFeature) or list[list[bytes]] (VarLenFeature) -- what the feature columns accept."""
        v = self.values()
        if not self.ragged:
            return v
        rs = self.row_splits
        return [v[rs[i]:rs[i + 1]] for i in range(len(rs) - 1)]


def parse_bytes(records, rec_offsets, key, varlen=False):
    """tf.io.FixedLenFeature([], tf.string) / tf.io.VarLenFeature(tf.string) (movielens.py:121-123 of the reference)."""
    n = len(rec_offsets) - 1
    totals = np.zeros(2, dtype=np.int64)
    fn = L.lib().dri_example_bytes
    L.check(fn(_ptr(records), _ptr(rec_offsets), n, key.encode(), 1 if varlen else 0, None, 0, None, 0, None, _ptr(totals)),
            "dri_example_bytes(%r)" % key)
    nv, nb = int(totals[0]), int(totals[1])
    blob = np.empty(max(nb, 1), dtype=np.uint8)
    value_offsets = np.empty(nv + 1, dtype=np.int64)
    row_splits = np.empty(n + 1, dtype=np.int64)
    L.check(fn(_ptr(records), _ptr(rec_offsets), n, key.encode(), 1 if varlen else 0, _ptr(blob), nb, _ptr(value_offsets), nv,
               _ptr(row_splits), _ptr(totals)), "dri_example_bytes(%r)" % key)
    return BytesColumn(blob[:nb], value_offsets, row_splits, varlen)


class MovieLens(object):

    def __init__(self, filename="movielens.tfrecords"):
        self._filename = filename if os.path.isabs(filename) else os.path.join(os.path.dirname(__file__), filename)
        self._columns = ["UserID", "MovieID", "Rating", "Timestamp",
                         "Gender", "Age", "Occupation", "Zip-code",
                         "Title", "Genres"]
        self.num_ratings = 1000209
        self.num_users = 6040
        self.num_movies = 3952
        self.gender_vocab = ["F", "M"]
        self.age_vocab = [1, 18, 25, 35, 45, 50, 56]
        self.occupation_vocab = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10,
                                 11, 12, 13, 14, 15, 16, 17, 18, 19, 20]
        self.genres_vocab = ["Action", "Adventure", "Animation", "Children's", "Comedy",
                             "Crime", "Documentary", "Drama", "Fantasy", "Film-Noir", "Horror",
                             "Musical", "Mystery", "Romance", "Sci-Fi", "Thriller", "War", "Western"]
        self._file = None

    def _open(self):
        if self._file is None:
            self._file = TFRecordFile(self._filename)
        return self._file

    def dataset(self, epochs=1, batch_size=256):
        """Generator of (example dict, ratings) -- `TFRecordDataset(f).repeat(epochs).batch(batch_size).map(parse)`
        (movielens.py:127-131): records are repeated FIRST, so batches run across epoch boundaries and only the very last
        batch may be short."""
        f = self._open()
        n = len(f)
        total = n * epochs
        for start in range(0, total, batch_size):
            index = np.arange(start, min(start + batch_size, total)) % n
            records, rec_offsets = f.read(index)
            example = {}
            for c in ["Age", "Occupation", "Timestamp"]:
                example[c] = parse_int64(records, rec_offsets, c)
            for c in ["UserID", "MovieID", "Gender", "Zip-code", "Title"]:
                example[c] = parse_bytes(records, rec_offsets, c)
            example["Genres"] = parse_bytes(records, rec_offsets, "Genres", varlen=True)
            ratings = parse_int64(records, rec_offsets, "Rating")
            yield example, ratings


class MovielensRanking(MovieLens):

    def __init__(self,
                 epochs: int = 10,
                 batch_size: int = 1024,
                 buffer_size: int = 1024,
                 train_size: float = 0.8,
                 *args, **kwargs):
        super(MovielensRanking, self).__init__(*args, **kwargs)
        self._epochs = epochs
        self._batch_size = batch_size
        self._buffer_size = buffer_size
        self._train_size = train_size

    @property
    def train_steps(self):
        num_train_ratings = self.num_ratings * self._epochs * self._train_size
        return int(num_train_ratings // self._batch_size)

    @property
    def train_steps_per_epoch(self):
        num_train_ratings = self.num_ratings * self._train_size
        return int(num_train_ratings // self._batch_size)

    @property
    def test_steps(self):
        return self.num_ratings // self._batch_size - self.train_steps_per_epoch

    @property
    def training_input_fn(self):
        return self._take(self.input_fn(), 0, self.train_steps)

    @property
    def testing_input_fn(self):
        return self._take(self.input_fn(), self.train_steps, self.test_steps)

    @staticmethod
    def _take(gen, skip, take):
        for i, item in enumerate(gen):
            if i >= skip + take:
                break
            if i >= skip:
                yield item

    def input_fn(self):
        """(features, labels): the six model features under the reference's names (movielens.py:172-179) as host lists /
        arrays the feature columns accept; labels = 1.0 where rating > 3 else 0.0, float32 [B, 1] (:180-182)."""
        for x, y in self.dataset(self._epochs, self._batch_size):
            features = {
                "user_id": x["UserID"].to_list(),
                "user_gender": x["Gender"].to_list(),
                "user_age": x["Age"],
                "user_occupation": x["Occupation"],
                "movie_id": x["MovieID"].to_list(),
                "movie_genres": x["Genres"].to_list(),
            }
            labels = np.where(y > 3, np.float32(1.0), np.float32(0.0)).astype(np.float32)[:, None]
            yield features, labels
