#!/usr/bin/env python3
"""Generates tests/golden/fixtures.npz from the CPU oracle (oracle/noize_oracle.c).

The reference ships no golden vectors and cannot run here, so these fixtures
are ORACLE outputs ("parity unpinned"): they freeze the restatement so that later edits to either
the oracle or the kernels show up as diffs.  Small on purpose: 64^2 tiles per stage, a 16^2 mesh
with its full index buffer, and per-basis noise probes at fixed coordinates (including
coordinates > 702 where psrnoise hashes un-reduced values)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle as O  # noqa: E402


def main():
    out = {}
    rng = np.random.default_rng(1234)
    tile = rng.random((64, 64), dtype=np.float32)
    out["tile"] = tile
    probes = np.array([[0.0, 0.0], [0.5, 0.25], [3.75, 9.125], [123.456, 78.9], [701.5, 50.25], [705.25, 99.5],
                       [1009.75, 101.5], [1060.5, 3.0], [9870.123, 4321.5], [-3.5, 7.25]], np.float32)
    out["probes"] = probes
    for b in range(8):
        out["noise_value_%d" % b] = np.array([O.noise_value(b, float(x), float(z)) for x, z in probes], np.float32)
        out["fractal_%d" % b] = O.fractal(b, 64, 64, 0.4, 1.0, 2.0, 0.0, 13, 12288, 20480, 1700)
    out["fractal_detuned"] = O.fractal(O.PERLIN, 64, 64, 0.5938, 1.0, 1.9168, 0.0317, 6, 0, 0, 658)
    for ft in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 13):
        out["kernel_filter_%d" % ft] = O.kernel_filter(tile, ft)
    out["gauss5_x17"] = O.kernel_filter(tile, O.GAUSS5_S1, 17)
    out["gauss_s4d00_w25"] = O.gauss(tile, 25, 7)
    out["smooth_w7_x2"] = O.smooth(tile, 7, 2)
    out["erosion_x1"] = O.erosion_min(tile)
    out["erosion_x5"] = O.erosion_min(tile, 5)
    h = O.kernel_filter(out["fractal_3"], O.GAUSS5_S1, 4)
    out["flow_height"] = h
    out["flowmap_x5_demo"] = O.flowmap(h, 5, 0.0, 0.005)
    out["flowmap_x2_default"] = O.flowmap(h, 2, -0.1, 0.1)
    hm = rng.random((20, 20), dtype=np.float32)
    out["mesh_heights"] = hm
    for mt, name in ((O.MESH_SQUARE, "square"), (O.MESH_OVERSHOOT, "overshoot")):
        v, i = O.mesh_heightmap(mt, hm, 16, 2, 1000.0, 1000.0)
        out["mesh_%s_vtx" % name], out["mesh_%s_idx" % name] = v, i
    out["pipeline_64"] = O.pipeline(64, 64)
    # lattice cells where fp32 mod289 returns 289 (negative multiples of 289), every basis
    for b in range(8):
        out["fractal_neg_%d" % b] = O.fractal(b, 64, 64, 0.5, 1.0, 2.0, 0.0, 2, -8990, -18230, 1)
    # stages either side of the metric path
    other = rng.random((64, 64), dtype=np.float32)
    out["other"] = other
    out["constant_mul"] = O.constant(tile, O.CONST_MULTIPLY, 0.37)
    out["constant_bin"] = O.constant(tile, O.CONST_BINARIZE, 0.37)
    for op in range(5):
        out["reduce_%d" % op] = O.reduce(tile, other, op)
    lut = np.array([1.0 - np.float32(i) / np.float32(256) for i in range(256)], np.float32)
    out["curve_lut"] = lut
    out["curve_invert"] = O.curve(tile, lut)
    out["thermal_x2"] = O.thermal_erosion(tile, 45.0, 0.5, 0.75, 2)
    out["crop_40"] = O.crop(tile, 40)
    out["sobel_2d"] = O.kernel_filter(tile, O.SOBEL3_2D)
    v, i = O.mesh_square_grid(5)
    out["mesh_planar_vtx"], out["mesh_planar_idx"] = v, i
    np.savez_compressed(os.path.join(HERE, "fixtures.npz"), **out)
    print("wrote fixtures.npz with %d arrays, %.1f KiB" %
          (len(out), os.path.getsize(os.path.join(HERE, "fixtures.npz")) / 1024))


if __name__ == "__main__":
    main()
