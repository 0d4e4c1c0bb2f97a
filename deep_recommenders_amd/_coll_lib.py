"""ctypes binding of the C-ABI exchange library (include/dr_collectives.h -> lib/libdr_collectives.so, RCCL underneath).

The Python engines in this package exchange through torch.distributed (backend "nccl" == RCCL); this library is the same plan
for a host that is not PyTorch.  Bound here for the tests (a single-rank communicator on one GPU)."""
import ctypes
import os

_p, _i64, _i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
SIGNATURES = {
    "dr_coll_unique_id": [_p],
    "dr_coll_init": [ctypes.POINTER(ctypes.c_void_p), _i32, _i32, _p],
    "dr_coll_destroy": [_p],
    "dr_coll_world": [_p],
    "dr_coll_rank": [_p],
    "dr_coll_last_error": [],
    "dr_coll_alltoall_i64": [_p, _p, _p, _i64, _p],
    "dr_coll_alltoallv": [_p, _p, ctypes.POINTER(_i64), _p, ctypes.POINTER(_i64), _i64, _p],
    "dr_coll_allreduce_f32": [_p, _p, _i64, _p],
    "dr_coll_allgather": [_p, _p, _p, _i64, _p],
}
_RESTYPE = {"dr_coll_last_error": ctypes.c_char_p, "dr_coll_world": ctypes.c_int32, "dr_coll_rank": ctypes.c_int32}
ID_BYTES = 128
_LIB = None


def path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libdr_collectives.so")


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(path()):
            raise RuntimeError("%s not found: run `python -m deep_recommenders_amd.build`" % path())
        L = ctypes.CDLL(path())
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = _RESTYPE.get(name, ctypes.c_int)
        _LIB = L
    return _LIB


def check(status, what):
    if status != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, status, (lib().dr_coll_last_error() or b"").decode()))
