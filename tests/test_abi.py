"""The C-ABI library builds for gfx950, loads, and exports every symbol include/*.h declares.
No compute calls here (no GPU needed)."""
import ctypes
import os
import re

import pytest

from hjtest import ROOT, has_gpu, pkg


def _declared_functions():
    names = set()
    for h in ("hj.h", "hj_reference_abi.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", text):
            n = m.group(1)
            if n.startswith("hj_") or n == "hashJoinClusteredProbe":
                names.add(n)
    return names


def test_library_builds_and_exports_every_declared_symbol():
    p = pkg()
    L = p._lib.lib()
    declared = _declared_functions()
    assert len(declared) >= 35
    for n in declared:
        assert hasattr(L, n), "libhj.so does not export %s" % n
    # and the Python binding covers exactly the header
    assert declared == set(p._lib.SIGNATURES), declared ^ set(p._lib.SIGNATURES)
    assert b"gfx950" in L.hj_version()


def test_code_object_is_gfx950():
    so = pkg()._lib.LIB_PATH
    blob = open(so, "rb").read()
    assert b"gfx950" in blob and b"k_scatter" in blob and b"k_join" in blob


def test_struct_layout_matches_reference_args_block():
    # src/common-host.h:39-52: int* S; size_t S_els; char[50]; int* R; size_t R_els; char[50]; int; uint; uint
    A = pkg()._lib.Args
    assert A.S.offset == 0 and A.S_els.offset == 8 and A.S_filename.offset == 16
    assert A.R.offset == 72 and A.R_els.offset == 80 and A.R_filename.offset == 88
    assert A.threadsNum.offset == 140 and A.sharedMem.offset == 144 and A.pivotsNum.offset == 148
    assert ctypes.sizeof(A) == 152


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode")
def test_fails_loudly_without_gpu():
    p = pkg()
    with pytest.raises(p.HJError):
        p.HashJoin(0)
    # the reference entry point reports the failure instead of falling back to a CPU join
    import numpy as np
    r = p.hashJoinClusteredProbe(np.arange(8, dtype=np.int32), np.arange(8, dtype=np.int32))
    assert r["status"] != 0 and r["matches"] == 0


def _build_c_client(tmp_path):
    """gcc (not hipcc), C99, only include/*.h: the ABI is consumable from plain C."""
    import subprocess
    p = pkg()
    p._lib.build()
    exe = str(tmp_path / "c_abi_client")
    libdir = os.path.dirname(p._lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_abi_client.c"), "-o", exe, "-L", libdir, "-lhj",
                           "-Wl,-rpath," + libdir])
    return exe


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode")
def test_c_client_builds_and_fails_loudly_without_gpu(tmp_path):
    import subprocess
    r = subprocess.run([_build_c_client(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 77, r.stdout + r.stderr


@pytest.mark.gpu
def test_c_client_on_gpu(tmp_path):
    import subprocess
    r = subprocess.run([_build_c_client(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "c_abi_client ok: 150000 matches" in r.stdout
