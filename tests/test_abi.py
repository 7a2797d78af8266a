"""CPU tests of the drop-in boundary: the shared library loads without a GPU, exports every symbol
include/rls_mi355x.h declares, the ctypes binding covers exactly that set, and a context cannot be
created without a device (no silent CPU fallback)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rls_mi355x.h")


def header_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rls_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_documented_surface():
    syms = header_symbols()
    for must in ("rls_ctx_create", "rls_gemv", "rls_nrm2", "rls_dotc", "rls_axpy", "rls_prox_l1", "rls_prox_l21",
                 "rls_prox_tv_fgp", "rls_prox_positive", "rls_cgnr_init", "rls_cgnr_step", "rls_fista_step",
                 "rls_cg_solve", "rls_cgnr_step_local_a", "rls_operator_mul_normal"):
        assert must in syms


def test_library_exports_every_header_symbol(rls):
    lib = rls.load()
    missing = [s for s in header_symbols() if not hasattr(lib, s)]
    assert not missing, f"declared in the header but not exported: {missing}"
    out = subprocess.run(["nm", "-D", "--defined-only", rls.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (rls_[a-z0-9_]+)", out))
    assert set(header_symbols()) <= exported


def test_binding_covers_exactly_the_header(rls):
    from rls_amd import _lib

    assert sorted(_lib.PROTOTYPES) == header_symbols()


def test_abi_version_and_status_codes(rls):
    lib = rls.load()
    assert lib.rls_abi_version() == 2
    assert lib.rls_ctx_sync(None) == -1          # RLS_E_INVALID on a null context
    assert lib.rls_cgnr_step(None, 1) == -1
    assert lib.rls_last_error_string(None) == b"null context"
    import ctypes as C
    shape = (C.c_int64 * 2)(8, 8)
    dims = (C.c_int32 * 2)(0, 1)
    assert lib.rls_tv_grad_len(2, shape, 2, dims) == 2 * 8 * 7
    assert lib.rls_prox_tv_workspace_bytes(0, 2, shape, 2, dims) == (2 * 112 + 64) * 4
    assert lib.rls_tv_grad_len(2, shape, 1, (C.c_int32 * 1)(5)) == -1  # bad dim


def test_no_cpu_fallback_without_a_device(rls):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(rls.RLSError, match="no CPU fallback"):
        rls.Context(0)


def test_product_package_does_not_import_the_oracle():
    """the oracle is test infrastructure: nothing shipped may import, include, link or execute it"""
    pkg = os.path.join(ROOT, "regularizedleastsquares.jl_amd")
    pat_py = re.compile(r"^\s*(import|from)\s+\S*(rls_oracle|oracle)\b|__import__\(|importlib.*oracle", re.M)
    pat_c = re.compile(r"#\s*include\s*[\"<][^\">]*oracle|dlopen\([^)]*oracle", re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            text = open(os.path.join(dirpath, f), errors="ignore").read() if f.endswith((".py", ".hip", ".hpp", "Makefile")) else ""
            if f.endswith(".py"):
                assert not pat_py.search(text), f
            elif text:
                assert not pat_c.search(text) and "oracle/" not in "".join(l for l in text.splitlines() if not l.lstrip().startswith("//")), f
    for f in ("rls_amd.py",):
        assert not pat_py.search(open(os.path.join(ROOT, f)).read())
