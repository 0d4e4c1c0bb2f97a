"""TEST INFRASTRUCTURE (oracle): an independent pure-Python restatement of the two on-disk formats the reference's
deep_recommenders/datasets/movielens.py:54-92 writes through TensorFlow -- tf.train.Example protobuf encoding and TFRecord
framing -- used to WRITE synthetic files for the native reader's tests and to decode them a second way.  Pinned by
(a) the CRC-32C known answer crc32c(b"123456789") == 0xE3069283 and (b) byte equality with google.protobuf's own
serializer on dynamically declared Example/Features/Feature messages (tests/test_input_pipeline.py).
Only tests/ may import this module."""
import struct

_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1
    _TABLE.append(_c)


def crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    for b in data:
        c = (c >> 8) ^ _TABLE[(c ^ b) & 0xFF]
    return c ^ 0xFFFFFFFF


def masked_crc(data: bytes) -> int:
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(v: int) -> bytes:
    v &= (1 << 64) - 1                 # int64 negatives: two's complement, 10 bytes
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(field: int, payload: bytes) -> bytes:          # length-delimited field
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def encode_feature(value, packed=True) -> bytes:
    """value: list of ints -> Int64List (field 3); list of bytes -> BytesList (field 1)."""
    if len(value) and isinstance(value[0], (bytes, bytearray)):
        return _ld(1, b"".join(_ld(1, bytes(v)) for v in value))
    if packed:
        body = _ld(1, b"".join(_varint(int(v)) for v in value)) if len(value) else b""
    else:
        body = b"".join(_varint((1 << 3) | 0) + _varint(int(v)) for v in value)
    return _ld(3, body)


def encode_example(features: dict, packed=True) -> bytes:
    """dict name -> list of ints | list of bytes, serialized like tf.train.Example (map entries in sorted key order,
    which is what protobuf's deterministic serialization produces)."""
    entries = b""
    for k in sorted(features):
        entry = _ld(1, k.encode()) + _ld(2, encode_feature(features[k], packed))
        entries += _ld(1, entry)
    return _ld(1, entries)


def write_tfrecords(path, serialized_examples):
    with open(path, "wb") as f:
        for rec in serialized_examples:
            hdr = struct.pack("<Q", len(rec))
            f.write(hdr + struct.pack("<I", masked_crc(hdr)) + rec + struct.pack("<I", masked_crc(rec)))


def movielens_example(user_id, movie_id, rating, timestamp, gender, age, occupation, zipcode, title, genres):
    """Same keys / kinds as `_serialize_example` (movielens.py:54-62)."""
    return {"Age": [age], "Occupation": [occupation], "Rating": [rating], "Timestamp": [timestamp],
            "UserID": [user_id], "MovieID": [movie_id], "Gender": [gender], "Zip-code": [zipcode], "Title": [title],
            "Genres": list(genres)}
