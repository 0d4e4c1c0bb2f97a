"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, and exports exactly the
symbols include/dr_hotpath.h declares; the ctypes binding covers all of them; the product path refuses to
run without the library or without a GPU (no silent fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "dr_hotpath.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dr_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    from deep_recommenders_amd import build
    so = build.build()
    L = ctypes.CDLL(so)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "libdr_hotpath.so does not export %s" % n


def test_ctypes_binding_covers_header():
    from deep_recommenders_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    _lib.lib()
    assert b"gfx950" in _lib.lib().dr_version()


def test_no_cpu_fallback():
    from deep_recommenders_amd import ops
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        ops.fm2_fwd(torch.zeros(2, 3, 4))


def test_missing_library_fails_loudly(monkeypatch):
    from deep_recommenders_amd import _lib
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "SO_PATH", "/nonexistent/libdr_hotpath.so")
    with pytest.raises(_lib.HotPathLibraryMissing):
        _lib.lib()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "deep_recommenders_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, fn)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, fn


def test_collectives_library_exports_every_declared_symbol():
    """include/dr_collectives.h (the exchange steps over RCCL, SURVEY section 8b's proposal): builds, loads, exports what it declares."""
    from deep_recommenders_amd import build, _coll_lib
    so = build.build_collectives()
    src = open(os.path.join(ROOT, "include", "dr_collectives.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(dr_coll_[a-z0-9_]+)\s*\(", src)))
    assert len(names) >= 9 and sorted(_coll_lib.SIGNATURES) == names
    L = ctypes.CDLL(so)
    for n in names:
        assert hasattr(L, n), "libdr_collectives.so does not export %s" % n
    assert _coll_lib.lib().dr_coll_world(None) == 0 and _coll_lib.lib().dr_coll_rank(None) == -1
