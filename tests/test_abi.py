"""CPU-side checks of the drop-in boundary: libnoize_hip.so loads, exports every symbol that
include/noize_hip.h declares, and fails loudly (no fallback) when there is no GPU."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "noize_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nz_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(nj):
    lib = ctypes.CDLL(nj._native.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 45
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_python_binding_covers_the_header(nj):
    assert sorted(nj._native.SIGNATURES) == declared_symbols()


def test_header_cites_the_reference_interfaces():
    text = open(HEADER).read()
    for cite in ("Noise/Fractal/Fractal.cs:76-88", "Filter/Kernel/KernelJob.cs:308-314", "BlurJob.cs:23-30",
                 "KernelJob.cs:350", "Geologic/FlowMap/FlowMapJob.cs:82-98", "FlowMapJob.cs:154-165",
                 "FlowMapJob.cs:220-228", "Filter/NormalizeJob.cs:94-100", "Mesh/Job/HeightMapMeshJob.cs:55-65"):
        assert cite in text, cite


def test_version_and_error_string(nj):
    assert nj._native.lib.nz_version() == 100
    assert isinstance(nj._native.lib.nz_last_error(), bytes)


def test_no_silent_fallback_without_a_gpu(nj):
    if nj.Context.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(nj.NoizeError) as e:
        nj.Context(0)
    assert e.value.status == nj._native.NZ_ERR_NO_DEVICE


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "noize_job_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                pats = [r"^\s*(import|from)\s+oracle", r"#include\s*[<\"][^>\"]*oracle", r"libnoize_oracle"]
                if f != "nz_comm.cpp":  # the one file that opens a library at run time: RCCL, checked below
                    pats.append(r"dlopen")
                for pat in pats:
                    assert not re.search(pat, src, re.M), (pat, os.path.join(dirpath, f))
    # nz_comm.cpp opens librccl (and nothing else) lazily: every library name it can hand to dlopen is an RCCL one or the
    # NZ_RCCL_LIB override, and the word "oracle" does not occur in the file
    src = open(os.path.join(pkg, "csrc", "nz_comm.cpp")).read()
    assert "oracle" not in src
    names = re.search(r"const char \*names\[\] = \{([^}]*)\}", src).group(1)
    assert [n.strip() for n in names.split(",")] == ["env", '"librccl.so.1"', '"/opt/rocm/lib/librccl.so.1"', '"librccl.so"']
    assert len(re.findall(r"\bdlopen\(", src)) == 1  # the one call, over those names


def test_stripe_struct_layout(nj):
    assert ctypes.sizeof(nj.Stripe) == 28
    assert [f[0] for f in nj.Stripe._fields_] == ["cols", "rows", "grow0", "grows", "own0", "own1", "pitch"]
