"""The Clp-side adapters of include/adapters/ (ClpGpuPackedMatrix, CoinGpuFactorization,
ClpGpuDualRowSteepest, clpGpuDual) cannot be linked here -- CoinUtils is absent -- but they can be
type-checked: g++ -fsyntax-only against tests/stubs/, minimal headers that restate the virtual
signatures of the reference's plug-in classes.  Every `override` in the adapters then fails to compile if
its signature drifts from the stubs, and (when the reference tree is mounted) every stub declaration is
looked up verbatim in the reference header it restates."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ADAPTERS = ["ClpGpuPackedMatrix.hpp", "CoinGpuFactorization.hpp", "ClpGpuDualRowSteepest.hpp", "ClpGpuDual.cpp"]


@pytest.mark.parametrize("name", ADAPTERS)
def test_adapter_compiles_against_stub_headers(name):
    src = os.path.join(ROOT, "include", "adapters", name)
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wno-unused", "-Wno-undefined-inline", "-x", "c++",
           "-I", os.path.join(ROOT, "tests", "stubs"), "-I", os.path.join(ROOT, "include"), src]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def _norm(text):
    return re.sub(r"\s+", " ", re.sub(r"//[^\n]*", "", text)).strip()


@pytest.mark.parametrize("stub,ref", [("ClpMatrixBase.hpp", "src/ClpMatrixBase.hpp"), ("ClpDualRowPivot.hpp", "src/ClpDualRowPivot.hpp"),
                                      ("ClpPackedMatrix.hpp", "src/ClpPackedMatrix.hpp")])
def test_stub_virtuals_are_the_reference_declarations(stub, ref):
    ref_path = os.path.join("/root/reference", ref)
    if not os.path.exists(ref_path):
        pytest.skip("reference tree not mounted (GPU box)")
    reference = _norm(open(ref_path).read())
    text = open(os.path.join(ROOT, "tests", "stubs", stub)).read()
    decls = re.findall(r"virtual[^;{]*?\)\s*(?:const)?\s*(?:=\s*0)?\s*[;{]", text, flags=re.S)
    assert len(decls) >= 5
    for d in decls:
        d = _norm(d).rstrip(";{").strip()
        if d.startswith("virtual ~"):
            continue
        assert d in reference, f"{stub}: `{d}` is not a declaration of {ref}"


def test_stub_accessors_exist_in_the_reference():
    if not os.path.exists("/root/reference/src/ClpSimplex.hpp"):
        pytest.skip("reference tree not mounted (GPU box)")
    reference = _norm(open("/root/reference/src/ClpSimplex.hpp").read() + open("/root/reference/src/ClpModel.hpp").read()
                      + open("/root/reference/src/ClpFactorization.hpp").read())
    text = open(os.path.join(ROOT, "tests", "stubs", "ClpSimplex.hpp")).read()
    decls = re.findall(r"^[ \t]*((?:inline |mutable )?[A-Za-z_][^;{}()\n]*\([^;{}]*\)(?: const)?);", text, flags=re.M)
    decls += re.findall(r"^[ \t]*(mutable [^;\n]+);", text, flags=re.M)
    assert len(decls) >= 30
    for d in decls:
        assert _norm(d) in reference, f"`{_norm(d)}` not found in ClpSimplex.hpp / ClpModel.hpp / ClpFactorization.hpp"


def test_steepest_stub_is_the_reference_declaration():
    """tests/stubs/ClpDualRowSteepest.hpp (what clpGpuDual reads to forward the pivot rule's mode): the constructor's default mode and
    the accessor, looked up verbatim in src/ClpDualRowSteepest.hpp."""
    ref_path = "/root/reference/src/ClpDualRowSteepest.hpp"
    if not os.path.exists(ref_path):
        pytest.skip("reference tree not mounted (GPU box)")
    reference = _norm(open(ref_path).read())
    for decl in ("ClpDualRowSteepest(int mode = 3);", "inline int mode() const", "public ClpDualRowPivot {"):
        assert _norm(decl) in reference, decl
        assert _norm(decl).rstrip(";") in _norm(open(os.path.join(ROOT, "tests", "stubs", "ClpDualRowSteepest.hpp")).read()), decl
