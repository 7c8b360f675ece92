"""Committed golden fixtures (tests/golden/fixtures.npz, made by tests/golden/make_fixtures.py from
the oracle): the oracle must still reproduce them (CPU), and the HIP path must match them (GPU)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_parity

f32 = np.float32


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(GOLDEN, "fixtures.npz"))


def test_oracle_reproduces_fixtures(oracle, fx):
    t = fx["tile"]
    for b in range(8):
        got = np.array([oracle.noise_value(b, float(x), float(z)) for x, z in fx["probes"]], f32)
        assert np.array_equal(got, fx["noise_value_%d" % b]), b
        assert np.array_equal(oracle.fractal(b, 64, 64, 0.4, 1.0, 2.0, 0.0, 13, 12288, 20480, 1700), fx["fractal_%d" % b])
    for ft in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 13):
        assert np.array_equal(oracle.kernel_filter(t, ft), fx["kernel_filter_%d" % ft])
    assert np.array_equal(oracle.kernel_filter(t, oracle.GAUSS5_S1, 17), fx["gauss5_x17"])
    assert np.array_equal(oracle.erosion_min(t, 5), fx["erosion_x5"])
    assert np.array_equal(oracle.flowmap(fx["flow_height"], 5, 0.0, 0.005), fx["flowmap_x5_demo"])
    v, i = oracle.mesh_heightmap(oracle.MESH_OVERSHOOT, fx["mesh_heights"], 16, 2, 1000.0, 1000.0)
    assert np.array_equal(v, fx["mesh_overshoot_vtx"]) and np.array_equal(i, fx["mesh_overshoot_idx"])
    assert np.array_equal(oracle.pipeline(64, 64), fx["pipeline_64"])
    for b in range(8):
        assert np.array_equal(oracle.fractal(b, 64, 64, 0.5, 1.0, 2.0, 0.0, 2, -8990, -18230, 1), fx["fractal_neg_%d" % b])
    o = fx["other"]
    assert np.array_equal(oracle.constant(t, 0, 0.37), fx["constant_mul"])
    assert np.array_equal(oracle.constant(t, 1, 0.37), fx["constant_bin"])
    for op in range(5):
        assert np.array_equal(oracle.reduce(t, o, op), fx["reduce_%d" % op])
    assert np.array_equal(oracle.curve(t, fx["curve_lut"]), fx["curve_invert"])
    assert np.array_equal(oracle.thermal_erosion(t, 45.0, 0.5, 0.75, 2), fx["thermal_x2"])
    assert np.array_equal(oracle.crop(t, 40), fx["crop_40"])
    assert np.array_equal(oracle.kernel_filter(t, oracle.SOBEL3_2D), fx["sobel_2d"])
    v, i = oracle.mesh_square_grid(5)
    assert np.array_equal(v, fx["mesh_planar_vtx"]) and np.array_equal(i, fx["mesh_planar_idx"])


def _run(nj, stage, d):
    stage.ReceiveHandledInput(nj.PipelineWorkItem(d), nj.JobHandle())
    stage.jobHandle.Complete()
    return d.data.ToArray((d.resolution, d.resolution))


@pytest.mark.gpu
def test_hip_path_matches_fixtures(nj, ctx, fx):
    t = fx["tile"]

    def gd(host=None, xpos=0, zpos=0):
        return nj.GeneratorData("g", ctx.alloc(64 * 64) if host is None else ctx.from_host(host), 64, xpos, zpos)

    for b in range(8):
        st = nj.NoiseStage(ctx, nj.FractalNoise(b), 0.4, 1.0, 13, 2.0, 0.0, 1700)
        assert_parity(_run(nj, st, gd(xpos=12288, zpos=20480)), fx["fractal_%d" % b], "fractal %d" % b)
    st = nj.NoiseStage(ctx, nj.FractalNoise.Perlin, 0.5938, 1.0, 6, 1.9168, 0.0317, 658)
    assert_parity(_run(nj, st, gd()), fx["fractal_detuned"], "detuned perlin")
    for ft in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 13):
        got = _run(nj, nj.KernelFilterStage(ctx, nj.KernelFilterType(ft), 1), gd(t))
        assert_parity(got, fx["kernel_filter_%d" % ft], "filter %d" % ft)
    assert_parity(_run(nj, nj.KernelFilterStage(ctx, nj.KernelFilterType.Gauss5_S1, 17), gd(t)), fx["gauss5_x17"], "g17")
    assert_parity(_run(nj, nj.StageGaussianBlur(ctx, 1, nj.GaussSigma.s4d00, 25), gd(t)), fx["gauss_s4d00_w25"], "w25")
    assert_parity(_run(nj, nj.StageSmoothBlur(ctx, 2, 7), gd(t)), fx["smooth_w7_x2"], "smooth")
    assert_parity(_run(nj, nj.ErosionStage(ctx, 1), gd(t)), fx["erosion_x1"], "erosion1")
    assert_parity(_run(nj, nj.ErosionStage(ctx, 5), gd(t)), fx["erosion_x5"], "erosion5")
    assert_parity(_run(nj, nj.FlowMapStage(ctx, 5, 0.0, 0.005), gd(fx["flow_height"])), fx["flowmap_x5_demo"], "flow5")
    assert_parity(_run(nj, nj.FlowMapStage(ctx, 2, -0.1, 0.1), gd(fx["flow_height"])), fx["flowmap_x2_default"], "flow2")
    for mt, name in ((0, "square"), (1, "overshoot")):
        d = nj.MeshStageData("m", ctx.from_host(fx["mesh_heights"]), 16, 20, 2, 1000.0, 1000.0)
        st = nj.MeshTileStage(ctx, nj.MeshType(mt))
        st.ReceiveHandledInput(nj.PipelineWorkItem(d), nj.JobHandle())
        st.jobHandle.Complete()
        assert np.array_equal(d.mesh.index_array(), fx["mesh_%s_idx" % name])
        assert_parity(d.mesh.vertices.ToArray().reshape(-1, 12), fx["mesh_%s_vtx" % name], "mesh " + name)
    # fixtures added with the stages either side of the path (bit-exact: the kernels follow the oracle op for op)
    for b in range(1, 8):
        st = nj.NoiseStage(ctx, nj.FractalNoise(b), 0.5, 1.0, 2, 2.0, 0.0, 1)
        assert np.array_equal(_run(nj, st, gd(xpos=-8990, zpos=-18230)), fx["fractal_neg_%d" % b]), b
    assert np.array_equal(_run(nj, nj.ConstantStage(ctx, nj.ConstantOperationType.MULTIPLY, 0.37), gd(t)), fx["constant_mul"])
    assert np.array_equal(_run(nj, nj.ConstantStage(ctx, nj.ConstantOperationType.BINARIZE, 0.37), gd(t)), fx["constant_bin"])
    for op in range(5):
        d = nj.ReduceData("r", ctx.from_host(t), ctx.from_host(fx["other"]), 64)
        st = nj.ReduceStage(ctx, nj.ReductionType(op))
        st.ReceiveHandledInput(nj.PipelineWorkItem(d), nj.JobHandle())
        st.jobHandle.Complete()
        assert np.array_equal(d.data.ToArray((64, 64)), fx["reduce_%d" % op]), op
    lut = fx["curve_lut"]
    assert np.array_equal(_run(nj, nj.CurveStage(ctx, lambda x: lut[int(round(float(x) * 256))], 256), gd(t)), fx["curve_invert"])
    assert np.array_equal(_run(nj, nj.StageThermalErosion(ctx, 2, 45, 0.5, 0.75), gd(t)), fx["thermal_x2"])
    d = nj.DownsampleData("c", ctx.alloc(40 * 40), ctx.from_host(t), 40, 64)
    st = nj.CropStage(ctx)
    st.ReceiveHandledInput(nj.PipelineWorkItem(d), nj.JobHandle())
    st.jobHandle.Complete()
    assert np.array_equal(d.data.ToArray((40, 40)), fx["crop_40"])
    assert np.array_equal(_run(nj, nj.KernelFilterStage(ctx, nj.KernelFilterType.Sobel3_2D, 1), gd(t)), fx["sobel_2d"])
    nv, ni = nj._native.lib.nz_mesh_vertex_count(5), nj._native.lib.nz_mesh_index_count(5)
    v, i = ctx.alloc(nv * 12), ctx.alloc(ni, dtype=np.uint32)
    ctx.call("nz_square_grid_mesh", v.ptr, i.ptr, 5).Complete()
    assert np.array_equal(i.ToArray(), fx["mesh_planar_idx"]) and np.array_equal(v.ToArray().reshape(-1, 12), fx["mesh_planar_vtx"])
