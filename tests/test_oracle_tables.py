"""Pins the oracle (and the library's host-side table builder, via test_gpu_filters) against the
only known answers the reference holds on this path: its Gaussian coefficient literals."""
import json
import os

import numpy as np

from conftest import GOLDEN

SIGMAS = ["s0d50", "s1d00", "s1d50", "s2d00", "s2d50", "s3d00", "s3d50", "s4d00", "s4d50", "s5d00", "s5d50",
          "s6d00", "s6d50", "s7d00", "s7d50", "s8d00"]


def _tables():
    with open(os.path.join(GOLDEN, "gauss_tables.json")) as f:
        return json.load(f)


def test_blur_kernels_match_reference_literals(oracle):
    # Filter/Kernel/Blur/BlurKernels.cs:59-318: 16 sigmas x 12 widths, every coefficient bit-equal in fp32
    t = _tables()
    n = 0
    for si, s in enumerate(SIGMAS):
        rows = t["by_sigma"][s]
        assert len(rows) == 12
        for wi, row in enumerate(rows):
            want = np.array(row, np.float64).astype(np.float32)
            got = oracle.gauss_kernel(si, 3 + 2 * wi)
            assert np.array_equal(got, want), (s, 3 + 2 * wi)
            n += len(row)
    assert n == 2688


def test_kernel_filter_tables_match_reference_literals(oracle):
    # Filter/Kernel/KernelJob.cs:97-136
    fx = _tables()["fixed"]
    names = {oracle.GAUSS9_S1: "gauss9_s1", oracle.GAUSS7_S1: "gauss7_s1", oracle.GAUSS5_S1: "gauss5_s1",
             oracle.GAUSS3_S1: "gauss3_s1", oracle.GAUSS9_S2: "gauss9_s2", oracle.GAUSS7_S2: "gauss7_s2",
             oracle.GAUSS5_S2: "gauss5_s2", oracle.GAUSS3_S2: "gauss3_s2"}
    for ft, name in names.items():
        kx, kz, factor, ks = oracle.kernel_filter_table(ft)
        want = np.array(fx[name]).astype(np.float32)
        assert ks == len(want) and factor == 1.0
        assert np.array_equal(kx, want) and np.array_equal(kz, want)
    pairs = {oracle.SMOOTH3: ("smooth3", "smooth3"), oracle.SOBEL3_H: ("sobel3_HX", "sobel3_HZ"),
             oracle.SOBEL3_V: ("sobel3_VX", "sobel3_VZ"), oracle.PREWITT3_H: ("prewitt3_HX", "prewitt3_HZ"),
             oracle.PREWITT3_V: ("prewitt3_VX", "prewitt3_VZ")}
    for ft, (nx, nz) in pairs.items():
        kx, kz, factor, ks = oracle.kernel_filter_table(ft)
        assert ks == 3
        assert np.array_equal(kx, np.array(fx[nx], np.float32)) and np.array_equal(kz, np.array(fx[nz], np.float32))
    assert oracle.kernel_filter_table(oracle.SMOOTH3)[2] == np.float32(1.0) / np.float32(3.0)
    assert fx["smooth3Factor"] == [1.0, 3.0] and fx["sobel3Factor"] == [1.0] and fx["prewitt3Factor"] == [1.0]


def test_limit_width(oracle):
    # BlurHelper.limitWidth BlurKernels.cs:29-36
    assert [oracle.limit_width(w) for w in (1, 2, 3, 4, 5, 24, 25, 26, 40)] == [3, 3, 3, 5, 5, 25, 25, 25, 25]
    assert len(oracle.gauss_kernel(3, 4)) == 5 and len(oracle.gauss_kernel(3, 40)) == 25
