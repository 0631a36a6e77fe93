"""CPU tests of the drop-in boundary: libclpgpu.so builds for gfx950, loads, exports every symbol
include/clpgpu.h declares, and refuses to run without a HIP device (no CPU fallback)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "clpgpu.h")).read()
    return sorted(set(re.findall(r"\b(clpgpu_[a-z_]+)\s*\(", text)))


def test_header_symbols_exported(built):
    from clp_amd import engine

    lib = engine.lib()
    declared = declared_symbols()
    assert len(declared) >= 20
    assert sorted(engine.ABI_SYMBOLS) == declared, "engine.ABI_SYMBOLS out of sync with include/clpgpu.h"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in clpgpu.h but not exported"


def test_library_is_gfx950_only(built):
    from clp_amd import engine

    blob = open(engine.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_90", b"nvptx"):
        assert other not in blob


def test_no_cpu_fallback(built):
    import torch

    from clp_amd import engine

    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the -m gpu tests")
    with pytest.raises(RuntimeError):
        engine.ClpGpuSimplex(0)


def test_product_does_not_import_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "clp_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                for needle in ("import oracle", "from oracle", "oracle/", "oracle.", "libclporacle", "orc_", "clp_dual_oracle"):
                    assert needle not in text, f"{f} references the oracle ({needle!r})"
