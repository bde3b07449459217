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


def test_headers_are_plain_c_and_link(tmp_path):
    # the boundary must be bindable from C (cgo / JNI / ctypes style): compile a C translation unit that includes
    # both headers with -std=c99 -pedantic, takes the address of every declared entry point, and link it against
    # the built libraries (no call is made: no GPU here)
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = []
    for h in ("cafehip.h", "cafehost.h"):
        text = open(os.path.join(root, "include", h)).read()
        names += re.findall(r"\b(cafe(?:hip|host)_[a-z_0-9]+)\s*\(", text)
    names = sorted(set(n for n in names if not n.endswith("_fn")))
    assert len(names) > 30
    src = tmp_path / "abi.c"
    src.write_text('#include "cafehip.h"\n#include "cafehost.h"\n#include <stdio.h>\n'
                   "int main(void) {\n  const void *p[] = {" + ", ".join("(const void *)" + n for n in names) +
                   "};\n  printf(\"%d\\n\", (int)(sizeof p / sizeof p[0]));\n  return p[0] == 0;\n}\n")
    exe = tmp_path / "abi"
    libdir = os.path.join(root, "cafe_amd", "lib")
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-Wno-pedantic", "-I", os.path.join(root, "include"), str(src), "-o", str(exe),
           "-L", libdir, "-lcafehip", "-Wl,-rpath," + libdir]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
