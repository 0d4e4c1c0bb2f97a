"""Writes tests/golden/cityhash64_le32_vectors.json: third-party known answers for the <= 32-byte branches of Fingerprint64.

Source of the answers: abseil's `absl::hash_internal::CityHash64(const char*, size_t)` (CityHash v1.1), an exported symbol of
the pyarrow wheel's `libarrow_compute.so` in the build container -- compiled third-party code, not written by the author of
oracle/farmhash_fp64.c.  FarmHash's `farmhashna::Hash64` (= TensorFlow's Fingerprint64) shares
CityHash v1.1's HashLen0to16 and HashLen17to32 verbatim, so for inputs of 0..32 bytes the two functions are the same function
(confirmed here: the upstream TensorFlow vectors Fingerprint64("a".."d") are reproduced by this symbol).  From 33 bytes on
FarmHash uses its own mixing (HashLen33to64, the seeded long loop) and CityHash64 is NOT a reference for it.

What this pins: every decimal rendering of an int64 key (<= 20 bytes) -- i.e. the whole integer id path of the hot path,
including the 17-20 digit keys SURVEY.md listed as "no vectors" -- and byte strings up to 32 bytes.

Run in the build container (needs pyarrow); the GPU box only reads the committed JSON.
"""
import ctypes
import glob
import json
import os
import random

HERE = os.path.dirname(os.path.abspath(__file__))
SYM = "_ZN4absl12lts_2026010713hash_internal10CityHash64EPKcm"


def city64():
    import pyarrow
    for so in sorted(glob.glob(os.path.join(os.path.dirname(pyarrow.__file__), "libarrow_compute.so*"))):
        try:
            f = getattr(ctypes.CDLL(so), SYM)
        except (OSError, AttributeError):
            continue
        f.restype = ctypes.c_uint64
        f.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
        return f, os.path.basename(so), pyarrow.__version__
    raise SystemExit("no libarrow_compute with an exported absl CityHash64 found")


def main():
    f, so, ver = city64()
    assert f(b"a", 1) == 12917804110809363939 and f(b"d", 1) == 4470636696479570465      # upstream TF Fingerprint64 vectors
    rnd = random.Random(20260925)
    vecs = []
    for n in range(0, 33):                          # every length of the shared branches, 6 random contents each
        for _ in range(6):
            s = bytes(rnd.getrandbits(8) for _ in range(n))
            vecs.append({"hex": s.hex(), "fp": f(s, n)})
    keys = [0, 1, 2, 9, 10, 99, 6040, 3952, 10**7, 10**8 - 1, 10**8, 10**15, 10**16 - 1, 10**16, 10**16 + 1, 10**17,
            10**18, 2**63 - 1, -1, -2, -10**15, -10**16, -10**17, -10**18, -2**63, 1234567890123456789, 99999999999999999]
    keys += [rnd.randrange(-2**63, 2**63) for _ in range(60)] + [rnd.randrange(10**16, 2**63) for _ in range(40)]
    ints = [{"key": k, "fp": f(str(k).encode(), len(str(k)))} for k in keys]
    out = {"source": "absl::hash_internal::CityHash64 exported by pyarrow %s %s" % (ver, so),
           "valid_for": "inputs of 0..32 bytes (farmhashna::Hash64 == CityHash64 v1.1 there)",
           "bytes": vecs, "int64_as_decimal": ints}
    path = os.path.join(HERE, "cityhash64_le32_vectors.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=0)
    print("wrote", path, len(vecs), "+", len(ints), "vectors")


if __name__ == "__main__":
    main()
