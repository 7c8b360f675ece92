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
