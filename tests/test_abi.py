"""The C-ABI library builds for gfx950, loads, and exports every symbol include/cafehip.h
declares.  No compute calls (CPU only)."""
import ctypes as C
import os
import re

import pytest

import cafe_amd
from cafe_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="cafehip.h", prefix="cafehip_"):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z_0-9]+)\s*\(", txt)))


def test_header_and_binding_agree():
    syms = declared_symbols()
    assert len(syms) >= 15
    assert sorted(_lib.SIGNATURES) == syms


def test_host_header_and_binding_agree():
    syms = declared_symbols("cafehost.h", "cafehost_")
    assert len(syms) >= 10
    assert sorted(_lib.HOST_SIGNATURES) == syms


def test_library_exports_every_declared_symbol():
    L = cafe_amd.load()
    for s in declared_symbols() + declared_symbols("cafehost.h", "cafehost_"):
        assert hasattr(L, s), s
    assert L.cafehip_abi_version() == 1


def test_library_is_in_tree_and_gfx950():
    path = cafe_amd.lib_path()
    assert path.startswith(ROOT) and os.path.exists(path)
    blob = open(path, "rb").read()
    assert b"gfx950" in blob


def test_no_cpu_fallback_without_gpu():
    L = cafe_amd.load()
    h = C.c_void_p()
    rc = L.cafehip_create(C.byref(h), 0)
    if rc == 0:  # a GPU is present: nothing to check here
        L.cafehip_destroy(h)
        pytest.skip("GPU present")
    assert rc < 0
    assert b"no CPU fallback" in L.cafehip_last_error()


def test_product_never_imports_oracle():
    # only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/
    pkg = os.path.join(ROOT, "cafe_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dp, fn), errors="replace").read()
                assert "oracle" not in txt.replace("NOT the oracle", ""), os.path.join(dp, fn)
