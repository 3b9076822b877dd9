"""CPU checks of the C-ABI boundary: the library builds/loads, exports every symbol declared in
include/fxctr.h, and the ctypes table in fuxictr_amd/_lib.py covers exactly that set.
No compute call is made here (no GPU in the build container)."""
import ctypes
import os
import re

from conftest import ROOT
from fuxictr_amd import _lib


def _header_functions():
    text = open(os.path.join(ROOT, "include", "fxctr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(fx_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_declares_functions():
    names = _header_functions()
    assert "fx_emb_gather_fwd" in names and "fx_gemm_f32" in names and "fx_sparse_adam" in names
    assert len(names) >= 25


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _header_functions():
        assert hasattr(lib, name), "libfxctr.so does not export %s" % name


def test_ctypes_table_matches_header():
    assert sorted(_lib.SIGNATURES.keys()) == _header_functions()


def test_load_and_version_and_error_string():
    lib = _lib.load()
    assert lib.fx_abi_version() == 1
    assert isinstance(lib.fx_last_error(), bytes)


def test_argument_validation_without_gpu():
    """Entry points validate arguments before touching the device, so bad calls fail cleanly."""
    lib = _lib.load()
    st = lib.fx_emb_gather_fwd(None, 0, None, 0, None, None, None, 0, None, 0, None, None, 0,
                               None, 0, 4, None, 0, None)
    assert st == 1 and b"D=0" in lib.fx_last_error()
    st = lib.fx_gemm_f32(0, 0, 4, 4, 4, None, 4, None, 4, None, 4, None, 1, None, None)
    assert st == 1 and b"null matrix" in lib.fx_last_error()
    assert lib.fx_emb_grad_reduce_partials(1024, 16) == 16
    assert lib.fx_emb_grad_reduce_partials(1024, 128) == 128
    assert lib.fx_emb_grad_reduce_scratch_ints(3300) == 102


def test_struct_layout_matches_header():
    assert ctypes.sizeof(_lib.GemmEpilogue) == 88
    assert _lib.SC_WORDS * 4 == 64


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        _lib.load()
    except _lib.FxError as e:
        assert "no fallback" in str(e)
    else:
        raise AssertionError("load() must raise when the .so is missing")


def test_integration_doc_names_every_entry_point():
    """INTEGRATION.md maps each C-ABI entry point to the reference code it replaces."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [n for n in _header_functions() if n not in doc]
    assert not missing, missing


def test_every_entry_point_cites_the_reference_in_the_header():
    """include/fxctr.h: the comment in front of each declaration group cites reference file:line
    (or says the function is new functionality / housekeeping)."""
    text = open(os.path.join(ROOT, "include", "fxctr.h")).read()
    assert len(re.findall(r"[a-z_]+\.py:\d+", text)) >= 30
