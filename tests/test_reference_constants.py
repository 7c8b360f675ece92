"""The constants the reference holds beside the hot path -- enum orders, stage defaults, ErosionSettings' fields and
Reset() values, the demo assets -- against the product's.  tests/golden/reference_constants.json was extracted from the
reference's text by tests/golden/make_reference_constants.py (numbers and names only); nothing here is hand-typed."""
import inspect
import json
import os
import re

import numpy as np
import pytest

from conftest import ROOT

REF = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_constants.json")))


def _names(enum_cls):
    return [m.name for m in sorted(enum_cls, key=lambda m: m.value)]


def test_enum_orders_are_the_reference_s(nj):
    from noize_job_amd import live_erosion as le
    assert _names(nj.FractalNoise) == REF["enums"]["FractalNoise"]          # Noise/NoiseStage.cs:15-24
    assert _names(nj.KernelFilterType) == REF["enums"]["KernelFilterType"]  # Filter/Kernel/KernelJob.cs:79-94
    assert _names(nj.GaussSigma) == REF["enums"]["GaussSigma"]              # Filter/Kernel/Blur/BlurKernels.cs:8-25
    assert _names(le.ErosionMode) == REF["enums"]["ErosionMode"]
    assert [m.value for m in sorted(nj.FractalNoise, key=lambda m: m.value)] == list(range(len(REF["enums"]["FractalNoise"])))
    assert nj.BlurHelper.limitWidth(1000) == REF["BlurHelper.max_width"]
    # the C ABI's enumerators (include/noize_hip.h) carry the same numbers: an enum body lists them in order from 0
    hdr = re.sub(r"/\*.*?\*/", " ", open(os.path.join(ROOT, "include", "noize_hip.h")).read(), flags=re.S)

    def c_enum(first):
        body = re.search(r"\{\s*(%s\s*=\s*0[^}]*)\}" % first, hdr, re.S).group(1)
        return [t.split("=")[0].strip() for t in body.split(",") if t.strip()]

    def c_name(prefix, name):
        return prefix + re.sub(r"(?<=[a-z])(?=[A-Z])|(?<=[0-9])(?=[A-Z][a-z])", "_", name).upper()
    assert c_enum("NZ_GAUSS9_S1") == [c_name("NZ_", n) for n in REF["enums"]["KernelFilterType"]]
    assert c_enum("NZ_NOISE_SIN") == [c_name("NZ_NOISE_", n) for n in REF["enums"]["FractalNoise"]]


def test_stage_defaults_are_the_reference_s(nj):
    def defaults(cls):
        sig = inspect.signature(cls.__init__)
        return {k: v.default for k, v in sig.parameters.items() if v.default is not inspect.Parameter.empty}
    got = defaults(nj.NoiseStage)
    for k, v in REF["NoiseStage.defaults"].items():          # Noise/NoiseStage.cs:37-54
        assert got[k] == pytest.approx(v), k
    assert got["noiseType"] == nj.FractalNoise[REF["enums"]["FractalNoise"][0]]  # a C# enum field defaults to its first member
    got = defaults(nj.FlowMapStage)
    for k, v in REF["FlowMapStage.defaults"].items():        # Geologic/Stage/FlowMapStage.cs:18-23
        assert np.float32(got[k]) == np.float32(v), k


def test_erosion_settings_follow_the_reference(nj):
    from noize_job_amd import live_erosion as le
    es = nj.ErosionSettings()
    # every field of the ScriptableObject exists, Reset() values are the defaults (ErosionSettings.cs:8-92)
    assert set(REF["ErosionSettings.fields"]) == set(vars(es))
    for k, v in REF["ErosionSettings.Reset"].items():
        got = getattr(es, k)
        if isinstance(v, str):
            assert got == le.ErosionMode[v], k
        else:
            assert np.float32(got) == np.float32(v), k
    assert np.float32(es.FLOW_LOSS_RATE) == np.float32(REF["ErosionSettings.initialisers"]["FLOW_LOSS_RATE"])
    # ErosionParameters: the struct's field order is the ABI's (LiveErosionDataTypes.cs:78-100)
    assert [n for n, _ in nj._native.ErosionParameters._fields_] == REF["ErosionParameters.fields"]
    hdr = open(os.path.join(ROOT, "include", "noize_hip.h")).read()
    body = re.search(r"typedef struct nz_erosion_params \{(.*?)\}", hdr, re.S).group(1)
    c_fields = re.findall(r"\b([A-Z_]{3,})\b(?=\s*[,;])", re.sub(r"/\*.*?\*/", " ", body, flags=re.S))
    assert c_fields == REF["ErosionParameters.fields"]


def test_demo_parameters_used_by_the_tests_are_the_assets(nj):
    # tests/test_gpu_parity.py (test_demo_pipeline_with_invert_curve, test_fractal_matches_oracle) and DESIGN.md quote the
    # BasicDemo~ assets: the numbers come from here
    perl, simplex = REF["demo_assets"]["Perl"], REF["demo_assets"]["Simplex"]
    assert (perl["noiseType"], perl["octaves"], perl["noiseSize"]) == (int(nj.FractalNoise.Perlin), 6, 658)
    assert (perl["hurst"], perl["stepdown"], perl["detuneRate"]) == (0.5938, 1.9168, 0.0317)
    assert (simplex["noiseType"], simplex["hurst"], simplex["octaves"], simplex["noiseSize"]) == (int(nj.FractalNoise.Simplex), 0.9001, 6, 7475)
    src = open(os.path.join(ROOT, "tests", "test_gpu_parity.py")).read()
    assert "nj.FractalNoise.Perlin, %s, 1.0, %d, %s, %s, %d" % (perl["hurst"], perl["octaves"], perl["stepdown"], perl["detuneRate"],
                                                                 perl["noiseSize"]) in src
    assert REF["demo_assets"]["GaussLF"] == {"filter": int(nj.KernelFilterType.Gauss9_S1), "iterations": 2}
    assert REF["demo_assets"]["GaussHF"] == {"filter": int(nj.KernelFilterType.Gauss3_S1), "iterations": 3}
    assert REF["demo_assets"]["Sobel2D"]["filter"] == int(nj.KernelFilterType.Sobel3_2D)
    fm = REF["demo_assets"]["FlowMapStage"]
    assert (fm["iterations"], fm["normMin"], fm["normMax"]) == (1, 0, 0.005)   # the flow normalisation bench.py uses
    assert REF["demo_assets"]["MeshTileStage"]["meshType"] == int(nj.MeshType.OvershootSquareGridHeightMap)
    g = REF["demo_generator"]
    assert g["generatorResolution"] - 2 * g["margin"] > g["tileResolution"] - 2 * g["margin"] > 0 and g["tileHeight"] == 2000
