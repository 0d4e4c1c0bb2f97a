"""Input side: native TFRecord / tf.Example reader + the MovieLens / MovielensRanking mirror.
CPU only.  The wire formats are pinned three ways: the CRC-32C known answer, byte equality of the oracle encoder with
google.protobuf's serializer, and agreement of the native decoder with both."""
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import tfrecord_py as W                      # noqa: E402
from deep_recommenders_amd import _input_lib as L        # noqa: E402
from deep_recommenders_amd import datasets as DS         # noqa: E402

GENRES = [b"Action", b"Adventure", b"Animation", b"Children's", b"Comedy", b"Crime", b"Documentary", b"Drama", b"Fantasy",
          b"Film-Noir", b"Horror", b"Musical", b"Mystery", b"Romance", b"Sci-Fi", b"Thriller", b"War", b"Western"]


def _synthetic(n, seed=42):
    rng = np.random.default_rng(seed)
    rows = []
    for i in range(n):
        ng = int(rng.integers(1, 7))
        rows.append(W.movielens_example(
            user_id=str(int(rng.integers(1, 6041))).encode(), movie_id=str(int(rng.integers(1, 3953))).encode(),
            rating=int(rng.integers(1, 6)), timestamp=int(rng.integers(9 * 10**8, 10**9)),
            gender=[b"F", b"M"][int(rng.integers(0, 2))], age=int(rng.choice([1, 18, 25, 35, 45, 50, 56])),
            occupation=int(rng.integers(0, 21)), zipcode=("%05d" % rng.integers(0, 99999)).encode(),
            title=("Movie \xe9 %d (19%02d)" % (i, rng.integers(0, 99))).encode("utf-8"),
            genres=[GENRES[j] for j in rng.choice(18, size=ng, replace=False)]))
    return rows


def test_library_exports_every_declared_symbol():
    import re
    hdr = open(os.path.join(ROOT, "include", "dr_input.h")).read()
    declared = set(re.findall(r"\b(dri_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)
    lib = L.lib()
    for name in declared:
        assert getattr(lib, name) is not None
    assert b"dr_input" in lib.dri_version()


def test_crc32c_known_answers():
    data = np.frombuffer(b"123456789", dtype=np.uint8).copy()
    assert L.lib().dri_crc32c(data.ctypes.data, 9) == 0xE3069283 == W.crc32c(b"123456789")
    blob = np.random.default_rng(0).integers(0, 256, size=100003, dtype=np.uint8)      # slicing-by-8 + tail
    assert L.lib().dri_crc32c(blob.ctypes.data, blob.size) == W.crc32c(blob.tobytes())


def _protobuf_example_class():
    """tensorflow/core/example/{feature,example}.proto declared on the fly with google.protobuf (TF is not installed)."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name="dr_test_example.proto", package="drtest", syntax="proto3")
    T = descriptor_pb2.FieldDescriptorProto

    def msg(name):
        m = fd.message_type.add()
        m.name = name
        return m
    bl = msg("BytesList"); f = bl.field.add(name="value", number=1, type=T.TYPE_BYTES, label=T.LABEL_REPEATED)
    fl = msg("FloatList"); f = fl.field.add(name="value", number=1, type=T.TYPE_FLOAT, label=T.LABEL_REPEATED)
    il = msg("Int64List"); f = il.field.add(name="value", number=1, type=T.TYPE_INT64, label=T.LABEL_REPEATED)
    ft = msg("Feature")
    ft.oneof_decl.add(name="kind")
    ft.field.add(name="bytes_list", number=1, type=T.TYPE_MESSAGE, label=T.LABEL_OPTIONAL, type_name=".drtest.BytesList", oneof_index=0)
    ft.field.add(name="float_list", number=2, type=T.TYPE_MESSAGE, label=T.LABEL_OPTIONAL, type_name=".drtest.FloatList", oneof_index=0)
    ft.field.add(name="int64_list", number=3, type=T.TYPE_MESSAGE, label=T.LABEL_OPTIONAL, type_name=".drtest.Int64List", oneof_index=0)
    fs = msg("Features")
    entry = fs.nested_type.add(name="FeatureEntry")
    entry.options.map_entry = True
    entry.field.add(name="key", number=1, type=T.TYPE_STRING, label=T.LABEL_OPTIONAL)
    entry.field.add(name="value", number=2, type=T.TYPE_MESSAGE, label=T.LABEL_OPTIONAL, type_name=".drtest.Feature")
    fs.field.add(name="feature", number=1, type=T.TYPE_MESSAGE, label=T.LABEL_REPEATED, type_name=".drtest.Features.FeatureEntry")
    ex = msg("Example")
    ex.field.add(name="features", number=1, type=T.TYPE_MESSAGE, label=T.LABEL_OPTIONAL, type_name=".drtest.Features")
    del f
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("drtest.Example"))


def test_oracle_encoder_is_byte_identical_to_protobuf():
    Example = _protobuf_example_class()
    for row in _synthetic(50, seed=7):
        ex = Example()
        for k, v in row.items():
            if isinstance(v[0], bytes):
                ex.features.feature[k].bytes_list.value.extend(v)
            else:
                ex.features.feature[k].int64_list.value.extend(v)
        assert ex.SerializeToString(deterministic=True) == W.encode_example(row)
    # negative int64 (10-byte varint) and an empty list
    ex = Example()
    ex.features.feature["a"].int64_list.value.extend([-5, 0, 2**62])
    ex.features.feature["b"].bytes_list.value.extend([b""])
    assert ex.SerializeToString(deterministic=True) == W.encode_example({"a": [-5, 0, 2**62], "b": [b""]})


def _records(rows, packed=True):
    recs = [W.encode_example(r, packed=packed) for r in rows]
    blob = np.frombuffer(b"".join(recs), dtype=np.uint8).copy()
    offs = np.zeros(len(recs) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(r) for r in recs])
    return blob, offs


@pytest.mark.parametrize("packed", [True, False])
def test_native_example_parser(packed):
    rows = _synthetic(300, seed=11)
    blob, offs = _records(rows, packed)
    for key in ("Age", "Occupation", "Rating", "Timestamp"):
        np.testing.assert_array_equal(DS.parse_int64(blob, offs, key), [r[key][0] for r in rows])
    for key in ("UserID", "MovieID", "Gender", "Zip-code", "Title"):
        col = DS.parse_bytes(blob, offs, key)
        assert col.to_list() == [r[key][0] for r in rows]
        np.testing.assert_array_equal(np.diff(col.row_splits), 1)
    g = DS.parse_bytes(blob, offs, "Genres", varlen=True)
    assert g.to_list() == [r["Genres"] for r in rows]
    assert g.row_splits[-1] == sum(len(r["Genres"]) for r in rows) == len(g.value_offsets) - 1
    # negative values, a missing VarLen key (= empty row), unknown extra keys
    rows2 = [{"x": [-7], "s": [b"ab", b""]}, {"x": [2**40], "zzz": [1, 2, 3]}]
    blob2, offs2 = _records(rows2, packed)
    np.testing.assert_array_equal(DS.parse_int64(blob2, offs2, "x"), [-7, 2**40])
    assert DS.parse_bytes(blob2, offs2, "s", varlen=True).to_list() == [[b"ab", b""], []]
    # FixedLenFeature([]) contract: missing key, two values, wrong kind -> error (tf.io.parse_example raises too)
    for key, fn in (("s", lambda: DS.parse_bytes(blob2, offs2, "s")), ("zzz", lambda: DS.parse_int64(blob2, offs2, "zzz")),
                    ("x", lambda: DS.parse_bytes(blob2, offs2, "x")), ("nope", lambda: DS.parse_int64(blob2, offs2, "nope"))):
        with pytest.raises(ValueError, match="DRI_EPARSE"):
            fn()
    # truncated protobuf
    with pytest.raises(ValueError, match="DRI_EPARSE"):
        DS.parse_int64(blob[:offs[1] - 3].copy(), np.array([0, offs[1] - 3], dtype=np.int64), "Age")


def test_tfrecord_framing_roundtrip_and_corruption(tmp_path):
    rows = _synthetic(257, seed=3)
    path = str(tmp_path / "ml.tfrecords")
    W.write_tfrecords(path, [W.encode_example(r) for r in rows])
    f = DS.TFRecordFile(path)
    assert len(f) == 257
    recs, offs = f.read(np.arange(257))
    assert recs.tobytes() == b"".join(W.encode_example(r) for r in rows)
    recs, offs = f.read([5, 5, 0])                                  # arbitrary order / repeats
    assert recs[offs[0]:offs[1]].tobytes() == recs[offs[1]:offs[2]].tobytes() == W.encode_example(rows[5])
    # an empty file is a valid, empty dataset
    open(str(tmp_path / "empty"), "wb").close()
    assert len(DS.TFRecordFile(str(tmp_path / "empty"))) == 0
    # flip one payload byte -> data CRC mismatch; flip a length byte -> length CRC mismatch; truncate -> corrupt
    raw = bytearray(open(path, "rb").read())
    for pos in (40, 3):
        bad = bytearray(raw)
        bad[pos] ^= 0x10
        p2 = str(tmp_path / ("bad%d" % pos))
        open(p2, "wb").write(bad)
        with pytest.raises(ValueError, match="DRI_ECORRUPT"):
            DS.TFRecordFile(p2)
    open(str(tmp_path / "trunc"), "wb").write(raw[:-2])
    for verify in (True, False):
        with pytest.raises(ValueError, match="DRI_ECORRUPT"):
            DS.TFRecordFile(str(tmp_path / "trunc"), verify_crc=verify)
    with pytest.raises(ValueError, match="DRI_EIO"):
        DS.TFRecordFile(str(tmp_path / "does-not-exist"))
    # the masked-CRC constant itself: header of a zero-length record
    hdr = struct.pack("<Q", 0)
    assert W.masked_crc(hdr) == struct.unpack("<I", raw[8:12])[0] or len(rows) > 0


def test_movielens_dataset_and_ranking_input_fn(tmp_path):
    rows = _synthetic(100, seed=5)
    path = str(tmp_path / "movielens.tfrecords")
    W.write_tfrecords(path, [W.encode_example(r) for r in rows])
    ml = DS.MovieLens(path)
    assert (ml.num_ratings, ml.num_users, ml.num_movies) == (1000209, 6040, 3952) and len(ml.genres_vocab) == 18
    # repeat(2).batch(32): 200 records -> 6 full batches + one of 8; batch 4 straddles the epoch boundary (movielens.py:127-129)
    batches = list(ml.dataset(epochs=2, batch_size=32))
    assert [len(y) for _, y in batches] == [32] * 6 + [8]
    x3, y3 = batches[3]
    want = [rows[i % 100] for i in range(96, 128)]
    np.testing.assert_array_equal(y3, [r["Rating"][0] for r in want])
    assert x3["UserID"].to_list() == [r["UserID"][0] for r in want] and x3["Genres"].to_list() == [r["Genres"] for r in want]
    assert set(x3) == {"Age", "Occupation", "Timestamp", "UserID", "MovieID", "Gender", "Zip-code", "Title", "Genres"}
    # MovielensRanking: defaults and step arithmetic of the reference (movielens.py:136-164)
    r = DS.MovielensRanking(filename=path)
    assert (r._epochs, r._batch_size, r._buffer_size, r._train_size) == (10, 1024, 1024, 0.8)
    assert r.train_steps == int(1000209 * 10 * 0.8 // 1024) == 7814
    assert r.train_steps_per_epoch == int(1000209 * 0.8 // 1024) == 781
    assert r.test_steps == 1000209 // 1024 - 781 == 195
    r = DS.MovielensRanking(epochs=1, batch_size=16, filename=path)
    feats, labels = next(iter(r.input_fn()))
    assert set(feats) == {"user_id", "user_gender", "user_age", "user_occupation", "movie_id", "movie_genres"}
    assert labels.dtype == np.float32 and labels.shape == (16, 1)
    np.testing.assert_array_equal(labels[:, 0], [1.0 if rows[i]["Rating"][0] > 3 else 0.0 for i in range(16)])   # :180-182
    assert feats["movie_genres"][0] == rows[0]["Genres"] and feats["user_age"].dtype == np.int64
