"""ctypes binding of the host-side input library (include/dr_input.h -> lib/libdr_input.so)."""
import ctypes
import os

_p, _i64, _i32, _u32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_uint32
_cp = ctypes.c_char_p
SIGNATURES = {
    "dri_version": [],
    "dri_crc32c": [_p, _i64],
    "dri_tfrecord_index": [_cp, _i32, _p, _p, _i64, _p],
    "dri_tfrecord_read": [_cp, _p, _p, _i64, _p, _i64, _p],
    "dri_example_int64": [_p, _p, _i64, _cp, _p],
    "dri_example_bytes": [_p, _p, _i64, _cp, _i32, _p, _i64, _p, _i64, _p, _p],
}
_RESTYPE = {"dri_version": ctypes.c_char_p, "dri_crc32c": ctypes.c_uint32}
DRI_OK, DRI_EINVAL, DRI_EIO, DRI_ECORRUPT, DRI_EPARSE, DRI_ECAPACITY = 0, -1, -10, -11, -12, -13
_ERR = {DRI_EINVAL: "DRI_EINVAL (bad argument)", DRI_EIO: "DRI_EIO (cannot open / short read)",
        DRI_ECORRUPT: "DRI_ECORRUPT (TFRecord framing or CRC mismatch)",
        DRI_EPARSE: "DRI_EPARSE (malformed Example, or a fixed-length key missing / wrong kind / not one value)",
        DRI_ECAPACITY: "DRI_ECAPACITY (output too small)"}
_LIB = None


class InputLibraryMissing(RuntimeError):
    pass


def path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libdr_input.so")


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(path()):
            raise InputLibraryMissing("%s not found: run `python -m deep_recommenders_amd.build`" % path())
        L = ctypes.CDLL(path())
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = _RESTYPE.get(name, ctypes.c_int)
        _LIB = L
    return _LIB


def check(status, what):
    if status != DRI_OK:
        raise ValueError("%s failed: %s" % (what, _ERR.get(status, status)))
