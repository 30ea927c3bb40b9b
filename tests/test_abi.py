"""The C-ABI library loads and exports every symbol include/*.h declares
(no compute call is made here: there is no GPU on this side)."""
import ctypes
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = open(h).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(mcx?_[a-z0-9_]+)\s*\(", text))
    return names


def test_library_exports_every_declared_symbol():
    from montecarlo_amd import binding

    lib = binding.load_library()
    names = declared_symbols()
    assert len(names) >= 20
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing


def test_error_strings_and_missing_device():
    from montecarlo_amd import binding

    lib = binding.load_library()
    assert lib.mcx_strerror(0) == b"ok"
    assert b"pattern" in lib.mcx_strerror(-5)
    import torch

    if not torch.cuda.is_available():
        # no CPU fallback: creating a context without a device must fail loudly
        h = ctypes.c_void_p()
        rc = lib.mcx_ctx_create(0, None, ctypes.byref(h))
        assert rc == -1 and not h.value


def test_binding_refuses_missing_library(monkeypatch, tmp_path):
    from montecarlo_amd import binding

    monkeypatch.setattr(binding, "_LIB", None)
    monkeypatch.setenv("MONTECARLO_LIB", str(tmp_path / "nope.so"))
    try:
        binding.load_library()
    except OSError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("a missing library must raise")
    finally:
        monkeypatch.setattr(binding, "_LIB", None)
