"""Parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs.  Bar: 1e-5 relative (abs floor 1e-6) for float planes, bit-exact for mesh indices
(BASELINE.json); where the kernels reproduce the oracle's operation order the planes are
additionally required to be equal bit for bit."""
import ctypes as C

import numpy as np
import pytest

from conftest import adversarial_tiles, assert_parity

pytestmark = pytest.mark.gpu
f32 = np.float32

BASES = ["Sin", "Perlin", "PeriodicPerlin", "Simplex", "RotatedSimplex", "Cellular", "DomainRotatedPerlin",
         "DomainRotatedSimplex"]


def gen(nj, ctx, res, uuid="t", xpos=0, zpos=0, host=None):
    data = ctx.alloc(res * res) if host is None else ctx.from_host(host)
    return nj.GeneratorData(uuid, data, res, xpos, zpos)


def run(stage, nj, d):
    stage.ReceiveHandledInput(nj.PipelineWorkItem(d), nj.JobHandle())
    stage.jobHandle.Complete()
    return d.data.ToArray((d.resolution, d.resolution))


# ---- noise ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("basis", range(8), ids=BASES)
def test_fractal_matches_oracle(nj, ctx, oracle, basis):
    res = 96
    for (octv, hurst, amp, step, det, ns, xp, zp) in [(8, 0.5, 1.0, 2.0, 0.0, 1000, 0, 0),
                                                       (13, 0.4, 1.0, 2.0, 0.0, 1700, 12288, 20480),
                                                       (6, 0.5938, 2.5, 1.9168, 0.0317, 658, 300, 77),
                                                       (1, 0.0, 1.0, 2.0, 0.0, 5, 0, 0)]:
        st = nj.NoiseStage(ctx, nj.FractalNoise(basis), hurst, amp, octv, step, det, ns)
        d = gen(nj, ctx, res, xpos=xp, zpos=zp)
        got = run(st, nj, d)
        want = oracle.fractal(basis, res, res, hurst, amp, step, det, octv, xp, zp, ns)
        assert_parity(got, want, "%s oct=%d" % (BASES[basis], octv))
        if basis != 0:  # every basis but Sin is op-for-op the oracle's arithmetic (Sin calls device sinf)
            assert np.array_equal(got, want), "%s oct=%d not bit-equal" % (BASES[basis], octv)
        d.data.Dispose()


@pytest.mark.parametrize("basis", range(1, 8), ids=BASES[1:])
def test_negative_and_huge_coordinates(nj, ctx, oracle, basis):
    # lattice tables and the psrnoise kernels' branch-free fmod are exact while the coordinates keep their
    # fractional bits; beyond that (and for negative lattice cells) the kernels must still follow the
    # reference's fp32 arithmetic: negative positions, exact multiples of the psrnoise periods (1010, 102),
    # coordinates up to ~4e7 where mod289 / fmod arguments are rounded integers
    res = 64
    for (octv, ns, xp, zp) in [(3, 1, -2100, -250), (3, 1, 1980, 60), (13, 7, -70000, -33333), (13, 3, 50000, 90000),
                               (2, 1, -4194400, 4194200)]:
        st = nj.NoiseStage(ctx, nj.FractalNoise(basis), 0.5, 1.0, octv, 2.0, 0.0, ns)
        d = gen(nj, ctx, res, xpos=xp, zpos=zp)
        got = run(st, nj, d)
        want = oracle.fractal(basis, res, res, 0.5, 1.0, 2.0, 0.0, octv, xp, zp, ns)
        assert np.array_equal(got, want), "%s ns=%d pos=(%d,%d)" % (BASES[basis], ns, xp, zp)
        d.data.Dispose()


@pytest.mark.parametrize("basis", range(1, 8), ids=BASES[1:])
def test_negative_lattice_cells_where_fp32_mod289_returns_289(nj, ctx, oracle, basis):
    # x - floor(x * (1/289)) * 289 in fp32 is 289, not 0, for some negative multiples of 289 (first: -8959,
    # -17629, -17918, -18207): the lattice tables must carry that index through like the reference's arithmetic
    res = 128
    for (octv, ns, xp, zp) in [(1, 1, -8990, -18260), (1, 1, -9020, 8930), (3, 1, -18000, -9000), (2, 2, -36000, -35900)]:
        st = nj.NoiseStage(ctx, nj.FractalNoise(basis), 0.5, 1.0, octv, 2.0, 0.0, ns)
        d = gen(nj, ctx, res, xpos=xp, zpos=zp)
        got = run(st, nj, d)
        want = oracle.fractal(basis, res, res, 0.5, 1.0, 2.0, 0.0, octv, xp, zp, ns)
        assert np.array_equal(got, want), "%s ns=%d pos=(%d,%d): %d cells differ" % (
            BASES[basis], ns, xp, zp, int((got != want).sum()))
        d.data.Dispose()


@pytest.mark.parametrize("basis", [2, 4], ids=[BASES[2], BASES[4]])
def test_psrnoise_where_the_lattice_wraps_begin(nj, ctx, oracle, basis):
    # the psrnoise kernels leave the wrap by the periods (1010, 102) out where no lattice coordinate of a workgroup's cells
    # can reach them, per octave and per row: tiles whose octaves sit on either side of both thresholds, rows and columns
    # that cross them inside the tile (coordinate = (cell + pos) / noiseSize * 2^octave), wider than one workgroup
    for (res, octv, ns, xp, zp) in [(640, 8, 4, 0, 0),          # x reaches 160 * 2^o: wraps from octave 3; y: from octave 0 on
                                    (300, 6, 3, 2400, -250),   # |x| in [800, 900]: at the x bound (1000 - |y| - 3.5) from octave 0
                                    (300, 6, 3, -3290, 280),   # negative x just short of -1010, y just short of 102
                                    (1100, 5, 11, -6000, -600),  # the thresholds cross the tile in both directions
                                    (257, 13, 1700, 0, 0)]:    # the default scale: no wrap in the first octaves
        st = nj.NoiseStage(ctx, nj.FractalNoise(basis), 0.5, 1.0, octv, 2.0, 0.0, ns)
        d = gen(nj, ctx, res, xpos=xp, zpos=zp)
        got = run(st, nj, d)
        want = oracle.fractal(basis, res, res, 0.5, 1.0, 2.0, 0.0, octv, xp, zp, ns)
        assert np.array_equal(got, want), "%s ns=%d pos=(%d,%d): %d cells differ" % (
            BASES[basis], ns, xp, zp, int((got != want).sum()))
        d.data.Dispose()


def test_fractal_config1_plumbing(nj, ctx, oracle):
    # BASELINE config 1: 1024^2 Perlin, 8 octaves, hurst 0.5
    st = nj.NoiseStage(ctx, nj.FractalNoise.Perlin, 0.5, 1.0, 8, 2.0, 0.0, 1000)
    d = gen(nj, ctx, 1024)
    got = run(st, nj, d)
    assert np.array_equal(got, oracle.fractal(oracle.PERLIN, 1024, 1024, 0.5, 1.0, 2.0, 0.0, 8, 0, 0, 1000))
    d.data.Dispose()


def test_fractal_odd_resolution_and_tile_seams(nj, ctx, oracle):
    # resolutions that are not multiples of the vector width; neighbouring tiles continue each other (B3)
    for res in (8, 37, 130):
        st = nj.NoiseStage(ctx, nj.FractalNoise.Simplex, 0.4, 1.0, 13, 2.0, 0.0, 1700)
        a = run(st, nj, gen(nj, ctx, res, xpos=0, zpos=0))
        b = run(st, nj, gen(nj, ctx, res, xpos=res, zpos=0))
        mono = oracle.fractal(oracle.SIMPLEX, res, 2 * res, 0.4, 1.0, 2.0, 0.0, 13, 0, 0, 1700)
        assert np.array_equal(np.hstack([a, b]), mono)


# ---- separable filters ----------------------------------------------------------------------------
@pytest.mark.parametrize("ft", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 13])
def test_kernel_filter_delegate_matches_oracle(nj, ctx, oracle, ft):
    res = 150
    for name, t in adversarial_tiles(res).items():
        d = gen(nj, ctx, res, host=t)
        got = run(nj.KernelFilterStage(ctx, nj.KernelFilterType(ft), 1), nj, d)
        assert np.array_equal(got, oracle.kernel_filter(t, ft)), (ft, name)
        d.data.Dispose()


@pytest.mark.parametrize("ft,iters", [(2, 2), (2, 3), (2, 4), (2, 17), (2, 32), (3, 5), (3, 13), (0, 2), (0, 5),
                                      (1, 3), (6, 7), (8, 6), (12, 2)])
def test_kernel_filter_stage_iterations(nj, ctx, oracle, ft, iters):
    # fused launches (T applications per launch, halo T*(K-1)/2) must equal the reference's chain of
    # single applications, including clamp-to-edge at every pass; resolutions straddle tile sizes
    for res in (64, 203, 300):
        tiles = adversarial_tiles(res)
        for name in ("uniform", "impulse_corner", "ramp_z"):
            d = gen(nj, ctx, res, host=tiles[name])
            got = run(nj.KernelFilterStage(ctx, nj.KernelFilterType(ft), iters), nj, d)
            want = oracle.kernel_filter(tiles[name], ft, iters)
            assert_parity(got, want, "ft=%d it=%d res=%d %s" % (ft, iters, res, name))
            assert np.array_equal(got, want), "ft=%d it=%d res=%d %s not bit-equal" % (ft, iters, res, name)
            d.data.Dispose()


@pytest.mark.parametrize("res,iters", [(16, 1), (90, 1), (257, 2)])
def test_sobel_2d_runs_both_filters_and_reduces(nj, ctx, oracle, res, iters):
    # SeparableKernelFilter.ScheduleReduce<RootSumSquaresTiles> (KernelJob.cs:187-215)
    t = np.random.default_rng(res).random((res, res), dtype=f32)
    got = run(nj.KernelFilterStage(ctx, nj.KernelFilterType.Sobel3_2D, iters), nj, gen(nj, ctx, res, host=t))
    assert np.array_equal(got, oracle.kernel_filter(t, oracle.SOBEL3_2D, iters))
    flat = np.full((res, res), 0.25, f32)       # no gradient anywhere, clamped borders included
    got = run(nj.KernelFilterStage(ctx, nj.KernelFilterType.Sobel3_2D, 1), nj, gen(nj, ctx, res, host=flat))
    assert np.array_equal(got, np.zeros((res, res), f32))


def test_edge_filter_delegates(nj, ctx, oracle):
    # Edge1DFilter / Edge2DFilter (Filter/Kernel/Edge/EdgeJob.cs): the Sobel / Prewitt kernels with kernelFactor 1
    res = 75
    t = np.random.default_rng(9).random((res, res), dtype=f32)
    for algo, (fh, fv) in enumerate(((oracle.SOBEL3_H, oracle.SOBEL3_V), (oracle.PREWITT3_H, oracle.PREWITT3_V))):
        for dirn, ft in enumerate((fh, fv)):
            src, tmp = ctx.from_host(t), ctx.alloc(res * res)
            ctx.call("nz_edge_1d_filter", src.ptr, tmp.ptr, algo, dirn, res).Complete()
            kx, kz, _, _ = oracle.kernel_filter_table(ft)
            assert np.array_equal(src.ToArray((res, res)), oracle.separable(t, 3, kx, kz, 1.0)), (algo, dirn)
        src, tmp = ctx.from_host(t), ctx.alloc(res * res)
        ctx.call("nz_edge_2d_filter", src.ptr, tmp.ptr, algo, res).Complete()
        hx, hz, _, _ = oracle.kernel_filter_table(fh)
        vx, vz, _, _ = oracle.kernel_filter_table(fv)
        want = oracle.reduce(oracle.separable(t, 3, hx, hz, 1.0), oracle.separable(t, 3, vx, vz, 1.0), 2)
        assert np.array_equal(src.ToArray((res, res)), want), algo
    with pytest.raises(nj.NoizeError):
        ctx.call("nz_edge_2d_filter", src.ptr, tmp.ptr, 2, res)


@pytest.mark.parametrize("sigma,width,iters", [(0, 3, 1), (1, 5, 3), (3, 9, 2), (7, 13, 1), (15, 25, 2), (5, 4, 1),
                                               (2, 40, 1)])
def test_gaussian_blur_stage(nj, ctx, oracle, sigma, width, iters):
    res = 90
    t = adversarial_tiles(res)["uniform"]
    d = gen(nj, ctx, res, host=t)
    got = run(nj.StageGaussianBlur(ctx, iters, nj.GaussSigma(sigma), width), nj, d)
    want = oracle.gauss(t, oracle.limit_width(width), sigma, iters)  # the stage passes limitWidth(width)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("width", [11, 13, 15, 17, 19, 21, 23, 25])
def test_wide_blur_kernels_interior_and_edge_tiles(nj, ctx, oracle, width):
    # 11..25 taps run the one-launch-per-application kernel: 32 x 128 output tiles, so 300 / 301 cells have
    # interior tiles (16-byte loads), edge tiles (clamped loads) and, at 301, unaligned rows
    for res, iters in ((300, 2), (301, 1)):
        t = np.random.default_rng(width * 1000 + res).random((res, res), dtype=f32)
        d = gen(nj, ctx, res, host=t)
        got = run(nj.StageGaussianBlur(ctx, iters, nj.GaussSigma(width % 16), width), nj, d)
        assert np.array_equal(got, oracle.gauss(t, width, width % 16, iters))
        d.data.Dispose()
    res = 300
    t = np.random.default_rng(width).random((res, res), dtype=f32)
    d = gen(nj, ctx, res, host=t)
    got = run(nj.StageSmoothBlur(ctx, 3, width), nj, d)   # factor 1, taps 1/width; odd count: copy back
    assert np.array_equal(got, oracle.smooth(t, width, 3))
    d.data.Dispose()


def test_gauss_filter_delegate_even_width_quirk(nj, ctx, oracle):
    # GaussFilter.Schedule called directly with an even width: 5-tap body, kernelSize 4 (BlurJob.cs:11-21)
    res = 40
    t = adversarial_tiles(res)["uniform"]
    src, tmp = ctx.from_host(t), ctx.alloc(res * res)
    ctx.call("nz_gauss_filter", src.ptr, tmp.ptr, 4, 3, res).Complete()
    assert np.array_equal(src.ToArray((res, res)), oracle.gauss(t, 4, 3))
    with pytest.raises(nj.NoizeError):
        ctx.call("nz_gauss_filter", src.ptr, tmp.ptr, 40, 3, res)  # indexes outside the 25-tap body


@pytest.mark.parametrize("width,iters", [(3, 1), (7, 2), (25, 1)])
def test_smooth_blur_stage(nj, ctx, oracle, width, iters):
    res = 77
    t = adversarial_tiles(res)["uniform"]
    d = gen(nj, ctx, res, host=t)
    got = run(nj.StageSmoothBlur(ctx, iters, width), nj, d)
    assert np.array_equal(got, oracle.smooth(t, width, iters))


def test_separable_series_custom_kernels(nj, ctx, oracle):
    res = 65
    t = adversarial_tiles(res)["uniform"]
    kx = np.array([0.1, -0.3, 0.5, 0.2, 0.7], f32)
    kz = np.array([-1.0, 0.25, 0.5, 0.125, 2.0], f32)
    src, tmp = ctx.from_host(t), ctx.alloc(res * res)
    ctx.call("nz_separable_series", src.ptr, tmp.ptr, res, 5, kx.ctypes.data_as(nj._native.f32p),
             kz.ctypes.data_as(nj._native.f32p), 0.37).Complete()
    assert np.array_equal(src.ToArray((res, res)), oracle.separable(t, 5, kx, kz, 0.37))


@pytest.mark.parametrize("iters", [1, 2, 3, 5, 8, 16, 33])
def test_erosion_stage(nj, ctx, oracle, iters):
    for res in (50, 257):
        for name, t in adversarial_tiles(res).items():
            if name.startswith("impulse"):
                t = 1.0 - t  # a low pixel that spreads towards +x, +z (B9)
            d = gen(nj, ctx, res, host=t)
            got = run(nj.ErosionStage(ctx, iters), nj, d)
            assert np.array_equal(got, oracle.erosion_min(t, iters)), (iters, res, name)
            d.data.Dispose()


def test_erosion_delegate(nj, ctx, oracle):
    res = 100
    t = adversarial_tiles(res)["uniform"]
    src = ctx.from_host(t)
    ctx.call("nz_erosion_kernel", src.ptr, res).Complete()
    assert np.array_equal(src.ToArray((res, res)), oracle.erosion_min(t))


# ---- flow map -------------------------------------------------------------------------------------
def test_flow_delegates_match_oracle(nj, ctx, oracle):
    res = 70
    rng = np.random.default_rng(11)
    h = rng.random((res, res), dtype=f32)
    w = (rng.random((res, res), dtype=f32) * f32(0.01)).astype(f32)
    fl = [(rng.random((res, res), dtype=f32) * f32(0.02)).astype(f32) for _ in range(4)]  # N,S,E,W
    dh, dw = ctx.from_host(h), ctx.from_host(w)
    dfl = [ctx.from_host(x) for x in fl]
    buf = [ctx.alloc(res * res) for _ in range(5)]
    ctx.call("nz_flowmap_compute_flow", dh.ptr, dw.ptr, dfl[0].ptr, buf[0].ptr, dfl[1].ptr, buf[1].ptr, dfl[2].ptr,
             buf[2].ptr, dfl[3].ptr, buf[3].ptr, res).Complete()
    want = oracle.flow_step(h, w, *fl)
    for g, wv, n in zip(dfl, want, "NSEW"):
        assert np.array_equal(g.ToArray((res, res)), wv), n
    ctx.call("nz_flowmap_update_water", dw.ptr, buf[4].ptr, dfl[0].ptr, dfl[1].ptr, dfl[2].ptr, dfl[3].ptr,
             res).Complete()
    assert np.array_equal(dw.ToArray((res, res)), oracle.water_step(w, *want))
    ctx.call("nz_flowmap_write_values", dh.ptr, dfl[0].ptr, dfl[1].ptr, dfl[2].ptr, dfl[3].ptr, res).Complete()
    v = oracle.velocity(*want)
    assert np.array_equal(dh.ToArray((res, res)), v)
    args = np.array([0.0, 0.005, 0.005], f32)
    ctx.call("nz_map_normalize_values", dh.ptr, buf[0].ptr, args.ctypes.data_as(nj._native.f32p), res).Complete()
    assert np.array_equal(dh.ToArray((res, res)), oracle.normalize(v, 0.0, 0.005))
    fill = ctx.alloc(res * res)
    ctx.call("nz_fill_array", fill.ptr, res, 0.0001).Complete()
    assert np.array_equal(fill.ToArray(), np.full(res * res, 0.0001, f32))


@pytest.mark.parametrize("iters", [1, 2, 5, 9])
def test_flowmap_stage_matches_oracle(nj, ctx, oracle, iters):
    for res in (16, 67, 200):
        h = oracle.kernel_filter(oracle.fractal(oracle.SIMPLEX, res, res, 0.4, 1.0, 2.0, 0.0, 8, 0, 0, 170),
                                 oracle.GAUSS5_S1, 2)
        for nmin, nmax in ((0.0, 0.005), (-0.1, 0.1)):
            d = gen(nj, ctx, res, host=h)
            got = run(nj.FlowMapStage(ctx, iters, nmin, nmax), nj, d)
            want = oracle.flowmap(h, iters, nmin, nmax)
            assert_parity(got, want, "flowmap it=%d res=%d" % (iters, res))
            assert np.array_equal(got, want)
            d.data.Dispose()


def test_flowmap_flat_and_ramp(nj, ctx, oracle):
    res = 64
    flat = np.full((res, res), 0.3, f32)
    got = run(nj.FlowMapStage(ctx, 5, -0.1, 0.1), nj, gen(nj, ctx, res, host=flat))
    assert np.array_equal(got, np.full((res, res), 0.5, f32))  # zero velocity -> (0 + 0.1) / 0.2 (B12)
    ramp = adversarial_tiles(res)["ramp_x"]
    got = run(nj.FlowMapStage(ctx, 3, 0.0, 0.005), nj, gen(nj, ctx, res, host=ramp))
    assert np.array_equal(got, oracle.flowmap(ramp, 3, 0.0, 0.005))


def test_flowmap_stage_reuses_its_planes_across_work_items(nj, ctx, oracle):
    # the stage keeps its planes between runs (FlowMapStage.cs:52-62); flux is defined zero per run
    res = 48
    st = nj.FlowMapStage(ctx, 3, 0.0, 0.005)
    for seed in (1, 2):
        h = np.random.default_rng(seed).random((res, res), dtype=f32)
        got = run(st, nj, gen(nj, ctx, res, host=h))
        assert np.array_equal(got, oracle.flowmap(h, 3, 0.0, 0.005))


# ---- mesh -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("mesh_type", [0, 1], ids=["SquareGrid", "Overshoot"])
def test_mesh_matches_oracle(nj, ctx, oracle, mesh_type):
    for in_res, res, margin in ((40, 32, 4), (70, 63, 3), (21, 16, 2)):
        h = np.random.default_rng(3).random((in_res, in_res), dtype=f32)
        d = nj.MeshStageData("m", ctx.from_host(h), res, in_res, margin, 500.0, 2000.0)
        st = nj.MeshTileStage(ctx, nj.MeshType(mesh_type))
        st.ReceiveHandledInput(nj.PipelineWorkItem(d), nj.JobHandle())
        st.jobHandle.Complete()
        vtx, idx = oracle.mesh_heightmap(mesh_type, h, res, margin, 2000.0, 500.0)
        assert np.array_equal(d.mesh.index_array(), idx)  # bit-exact integer stream
        got = d.mesh.vertices.ToArray().reshape(-1, 12)
        assert_parity(got, vtx, "vertices")
        assert np.array_equal(got, vtx)
        va = d.mesh.vertex_array()
        assert va.dtype.itemsize == 48 and np.array_equal(va["tangent"][:, 3], np.zeros(len(va), f32))


def test_mesh_rejects_unsafe_margins(nj, ctx):
    h = ctx.alloc(64)
    st = nj.MeshTileStage(ctx, nj.MeshType.OvershootSquareGridHeightMap)
    with pytest.raises(nj.NoizeError):
        st.ReceiveHandledInput(nj.PipelineWorkItem(nj.MeshStageData("m", h, 8, 8, 0, 1.0, 1.0)), nj.JobHandle())


# ---- whole pipeline -------------------------------------------------------------------------------
def test_metric_pipeline_matches_oracle(nj, ctx, oracle):
    # README example: simplex 13 oct -> Gauss5 x17 -> FlowMap -> value erosion (README.md:23-32), then mesh
    res = 512
    data = ctx.alloc(res * res)
    stages = [nj.NoiseStage(ctx, nj.FractalNoise.Simplex, 0.4, 1.0, 13, 2.0, 0.0, 1700),
              nj.KernelFilterStage(ctx, nj.KernelFilterType.Gauss5_S1, 17),
              nj.FlowMapStage(ctx, 5, 0.0, 0.005),
              nj.ErosionStage(ctx, 5)]
    pipe = nj.BasePipeline(stages, "metric")
    done = []
    pipe.Enqueue(nj.GeneratorData("tile-0", data, res, 4096 * 3, 4096 * 5), completeAction=done.append)
    pipe.RunToCompletion()
    assert len(done) == 1 and done[0].uuid == "tile-0"
    got = data.ToArray((res, res))
    want = oracle.pipeline(res, res, xpos=4096 * 3, zpos=4096 * 5)
    assert_parity(got, want, "metric pipeline")
    assert np.array_equal(got, want)
    pipe.Destroy()


def test_job_handles(nj, ctx):
    res = 256
    d = gen(nj, ctx, res)
    st = nj.NoiseStage(ctx, nj.FractalNoise.Simplex, 0.4, 1.0, 13, 2.0, 0.0, 1700)
    st.ReceiveHandledInput(nj.PipelineWorkItem(d), nj.JobHandle())
    h = st.jobHandle
    assert h.id > 0
    h.Complete()
    assert h.IsCompleted and nj.JobHandle().IsCompleted
    with pytest.raises(nj.NoizeError):
        ctx.call("nz_fill_array", d.data.ptr, res, 0.0, dep=10 ** 12)  # a handle this context never issued
    a, b = ctx.record(), ctx.record()
    b.Complete()
    assert ctx.elapsed_ms(a, b) >= 0.0


def test_cpp_host_mirror_runs_the_metric_pipeline(oracle, tmp_path):
    # noize_job_amd/host/noize_pipeline.hpp: the same stage graph from a compiled host, C ABI only
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "noize_job_amd", "host", "host_demo")
    assert os.path.exists(exe), "host_demo not built (run __graft_entry__.build())"
    out = str(tmp_path / "plane.f32")
    subprocess.check_call([exe, "256", out, "17", "5", "5"])
    got = np.fromfile(out, dtype=np.float32).reshape(256, 256)
    assert np.array_equal(got, oracle.pipeline(256, 256))
    # the same pipeline on a READ / WRITE plane pair (nz_*_rw entries: SWAP_RWTILE as a pointer swap)
    subprocess.check_call([exe, "256", out, "rw"])
    assert np.array_equal(np.fromfile(out, dtype=np.float32).reshape(256, 256), got)
    # ReducePipeline of the C++ mirror: simplex (left) x cellular (right), MULTIPLY
    subprocess.check_call([exe, "200", out, "reduce"])
    got = np.fromfile(out, dtype=np.float32).reshape(200, 200)
    a = oracle.fractal(oracle.SIMPLEX, 200, 200, 0.4, 1.0, 2.0, 0.0, 6, 37, 11, 300)
    b = oracle.fractal(oracle.CELLULAR, 200, 200, 0.5, 1.0, 2.0, 0.0, 3, 37, 11, 90)
    assert np.array_equal(got, oracle.reduce(a, b, 1))
    # context stages of the C++ mirror: the consumer waits for the producer's buffer, then reads it
    subprocess.check_call([exe, "128", out, "context"])
    got = np.fromfile(out, dtype=np.float32).reshape(128, 128)
    noise = oracle.fractal(oracle.SIMPLEX, 128, 128, 0.4, 1.0, 2.0, 0.0, 8, 64, 32, 200)
    assert np.array_equal(got, oracle.kernel_filter(noise, 2, 3))
    # the live erosion driver of the C++ mirror (LiveErosion::TriggerQueuedBeyerMT): config 4's shape at 384^2
    res, particles, cycles = 384, 4000, 2
    subprocess.check_call([exe, str(res), out, "live", str(particles), str(cycles)])
    got = np.fromfile(out, dtype=np.float32).reshape(3, res, res)
    h = oracle.fractal(oracle.CELLULAR, res, res, 0.4, 1.0, 2.0, 0.0, 13, 0, 0, 1700)
    nj_settings = dict(PARTICLES_PER_CYCLE=particles, CYCLES=cycles, WATER_STEPS=5)
    import noize_job_amd as nj
    from test_live_erosion import _params
    st = nj.ErosionSettings(**nj_settings)
    L = oracle.LiveErosionOracle(h, _params(oracle, st), tile_height=1000, patch_res=float(np.float32(2000.0) / np.float32(res - 16)))
    for c in range(cycles):
        L.cycle(0, particles, 11 * c + 3, water_steps=5, thermal=(st.TALUS, st.THERMAL_STEP, 2.0, st.THERMAL_CYCLES))
    assert np.array_equal(got[0], L.height) and np.array_equal(got[1], L.pool) and np.array_equal(got[2], L.flow)
    # free-running pipelines of the C++ mirror (what tools/bench_tiles.py does in Python): 3 pipelines on 3 contexts take 11
    # tiles of 160^2 in turn, nobody waits until the end; the file holds pipeline 0's last tile (tile 9, xpos 160 * 9)
    subprocess.check_call([exe, "160", out, "tiles", "3", "11"], stdout=subprocess.DEVNULL)
    assert np.array_equal(np.fromfile(out, dtype=np.float32).reshape(160, 160), oracle.pipeline(160, 160, xpos=160 * 9))
    # batched stage bodies from the C++ mirror: 3 tiles of 96^2 at (k * 96, -3 k)
    subprocess.check_call([exe, "96", out, "batch", "3"])
    got = np.fromfile(out, dtype=np.float32).reshape(3, 96, 96)
    for k in range(3):
        assert np.array_equal(got[k], oracle.pipeline(96, 96, xpos=96 * k, zpos=-3 * k)), k


# ---- element-wise stages ------------------------------------------------------
def test_constant_reduce_curve_stages(nj, ctx, oracle):
    res = 130
    rng = np.random.default_rng(21)
    a = (rng.random((res, res), dtype=f32) * f32(1.4) - f32(0.2)).astype(f32)
    b = rng.random((res, res), dtype=f32)
    for op in (0, 1):
        got = run(nj.ConstantStage(ctx, nj.ConstantOperationType(op), 0.37), nj, gen(nj, ctx, res, host=a))
        assert np.array_equal(got, oracle.constant(a, op, 0.37)), op
    for op in range(5):
        d = nj.ReduceData("r", ctx.from_host(a), ctx.from_host(b), res)
        wi = nj.PipelineWorkItem(d)
        st = nj.ReduceStage(ctx, nj.ReductionType(op))
        st.ReceiveHandledInput(wi, nj.JobHandle())
        st.jobHandle.Complete()
        assert isinstance(wi.data, nj.GeneratorData)  # TransformData, ReduceStage.cs:53-62
        assert np.array_equal(wi.data.data.ToArray((res, res)), oracle.reduce(a, b, op)), op
    # math.max / math.min of Unity.Mathematics (`float.IsNaN(y) || x > y ? x : y`): a NaN operand loses, and on a TIE the second
    # operand stays -- which shows where the two are +0 and -0 (round 6: looked for after the soak's clamp finding)
    res2 = 16
    sa = np.zeros((res2, res2), f32)
    sb = np.zeros((res2, res2), f32)
    sa[0, :8] = [0.0, -0.0, 0.0, -0.0, 1.0, np.nan, np.nan, np.inf]
    sb[0, :8] = [-0.0, 0.0, 0.0, -0.0, np.nan, 1.0, np.nan, -np.inf]
    for op in (3, 4):
        d = nj.ReduceData("r", ctx.from_host(sa), ctx.from_host(sb), res2)
        wi = nj.PipelineWorkItem(d)
        st = nj.ReduceStage(ctx, nj.ReductionType(op))
        st.ReceiveHandledInput(wi, nj.JobHandle())
        st.jobHandle.Complete()
        got, want = wi.data.data.ToArray((res2, res2)), oracle.reduce(sa, sb, op)
        assert np.array_equal(got, want, equal_nan=True), (op, got[0, :8], want[0, :8])
        # the SIGN of a zero result is where "bit-equal" ends in this repository: Unity's max(+0, -0) is its second operand (-0),
        # v_max_f32 orders the zeros (+0).  np.array_equal does not see it, no tolerance does; recorded, not claimed
        print("reduce op %d: zero signs GPU %s reference %s" % (op, np.signbit(got[0, :4]).tolist(), np.signbit(want[0, :4]).tolist()))
    for fn, samples in ((lambda t: 1.0 - t, 256), (lambda t: t * t * (3.0 - 2.0 * t), 64), (lambda t: 1.5 * t - 0.1, 7)):
        st = nj.CurveStage(ctx, fn, samples)
        got = run(st, nj, gen(nj, ctx, res, host=a))
        host = np.array([fn(f32(i) / f32(samples)) for i in range(samples)], f32)
        assert np.array_equal(got, oracle.curve(a, host)), samples
    with pytest.raises(Exception, match="Unhandled stageio"):
        run(nj.ReduceStage(ctx, nj.ReductionType.MAX), nj, gen(nj, ctx, res, host=a))


def test_reduce_pipeline_fans_in_two_upstream_pipelines(nj, ctx, oracle):
    # ReducePipeline.cs:82-148: the same tile is requested from both upstreams (the right one into the reduce
    # pipeline's own plane), then the reduce stages run on the pair; two queued tiles go through one by one
    res = 160
    left = nj.BasePipeline([nj.NoiseStage(ctx, nj.FractalNoise.Simplex, 0.4, 1.0, 6, 2.0, 0.0, 300),
                            nj.KernelFilterStage(ctx, nj.KernelFilterType.Gauss5_S1, 2)], "left")
    right = nj.BasePipeline([nj.NoiseStage(ctx, nj.FractalNoise.Cellular, 0.5, 1.0, 3, 2.0, 0.0, 90)], "right")
    red = nj.ReducePipeline(ctx, [nj.ReduceStage(ctx, nj.ReductionType.MULTIPLY),
                                  nj.CurveStage(ctx, lambda t: 1.0 - t, 256)], left, right, "reduce")
    assert red.GetDependencies()[:3] == [left, right, red]
    done = []
    tiles = [nj.GeneratorData("t%d" % i, ctx.alloc(res * res), res, 37 * i, 11) for i in range(2)]
    for t in tiles:
        red.Enqueue(t, completeAction=lambda d: done.append(d.uuid))
    red.RunToCompletion()
    assert done == ["t0", "t1"]
    lut = np.array([1.0 - f32(i) / f32(256) for i in range(256)], f32)
    for i, t in enumerate(tiles):
        a = oracle.kernel_filter(oracle.fractal(oracle.SIMPLEX, res, res, 0.4, 1.0, 2.0, 0.0, 6, 37 * i, 11, 300), 2, 2)
        b = oracle.fractal(oracle.CELLULAR, res, res, 0.5, 1.0, 2.0, 0.0, 3, 37 * i, 11, 90)
        assert np.array_equal(t.data.ToArray((res, res)), oracle.curve(oracle.reduce(a, b, 1), lut)), i
    red.Destroy()
    left.Destroy()
    right.Destroy()


def test_context_buffers_keep_a_tile_resident_between_pipelines(nj, ctx, oracle, tmp_path):
    # WriteGeneratorContextStage parks the producer's tile in a named device buffer; a consumer pipeline
    # queued first waits for it (dependencyHell) and reads it back with ReadGeneratorContextStage
    res = 128
    mgr = nj.PipelineStateManager(ctx)
    mgr.SetSavePath(str(tmp_path), "terrain", "v1")
    producer = nj.BasePipeline([nj.NoiseStage(ctx, nj.FractalNoise.Simplex, 0.4, 1.0, 8, 2.0, 0.0, 200),
                                nj.WriteGeneratorContextStage(ctx, "height")], "producer", contextManager=mgr)
    consumer = nj.BasePipeline([nj.ReadGeneratorContextStage(ctx, "height"),
                                nj.KernelFilterStage(ctx, nj.KernelFilterType.Gauss5_S1, 3)], "consumer",
                               contextManager=mgr)
    out = nj.GeneratorData("out", ctx.alloc(res * res), res, 64, 32)
    consumer.Enqueue(out)
    consumer.Update()
    assert not consumer.pipelineRunning and len(consumer.dependencyHell) == 1
    producer.Enqueue(nj.GeneratorData("src", ctx.alloc(res * res), res, 64, 32))
    producer.RunToCompletion()
    consumer.RunToCompletion()
    noise = oracle.fractal(oracle.SIMPLEX, res, res, 0.4, 1.0, 2.0, 0.0, 8, 64, 32, 200)
    assert np.array_equal(mgr.GetBufferNoLoad("64_32__128__height").ToArray((res, res)), noise)
    assert np.array_equal(out.data.ToArray((res, res)), oracle.kernel_filter(noise, 2, 3))
    # the buffer in the reference's on-disk form, picked up by a fresh manager
    mgr.SaveBufferToDisk("64_32__128__height")
    mgr2 = nj.PipelineStateManager(ctx)
    mgr2.SetSavePath(str(tmp_path), "terrain", "v1")
    assert np.array_equal(mgr2.GetBuffer("64_32__128__height", res * res).ToArray((res, res)), noise)
    for m in (mgr, mgr2):
        m.OnDestroy()
    producer.Destroy()
    consumer.Destroy()


def test_crop_stage(nj, ctx, oracle):
    a = np.random.default_rng(5).random((150, 150), dtype=f32)
    src = ctx.from_host(a)
    for out_res in (64, 150, 161):
        d = nj.DownsampleData("c", ctx.alloc(out_res * out_res), src, out_res, 150)
        st = nj.CropStage(ctx)
        st.ReceiveHandledInput(nj.PipelineWorkItem(d), nj.JobHandle())
        st.jobHandle.Complete()
        assert np.array_equal(d.data.ToArray((out_res, out_res)), oracle.crop(a, out_res)), out_res
    with pytest.raises(Exception, match="Unhandled stageio"):
        nj.CropStage(ctx).Schedule(nj.PipelineWorkItem(gen(nj, ctx, 8)), nj.JobHandle())


@pytest.mark.parametrize("res,count", [(100, 5), (128, 9), (37, 3), (256, 2)])
def test_batched_tiles_equal_single_tiles(nj, ctx, oracle, res, count):
    # `count` independent tiles through ONE launch sequence (nz_*_batch): each tile must come out exactly as it
    # does alone -- clamped at its own border, its own world position
    rng = np.random.default_rng(res)
    positions = [(int(rng.integers(-5000, 5000)), int(rng.integers(-5000, 5000))) for _ in range(count)]
    batch = nj.GeneratorDataBatch.create(ctx, "b", res, positions)
    stages = [nj.NoiseStage(ctx, nj.FractalNoise.Simplex, 0.4, 1.0, 9, 2.0, 0.0, 300),
              nj.KernelFilterStage(ctx, nj.KernelFilterType.Gauss5_S1, 7), nj.FlowMapStage(ctx, 5, 0.0, 0.005),
              nj.ErosionStage(ctx, 5)]
    pipe = nj.BasePipeline(stages, "batched")
    done = []
    pipe.Enqueue(batch, completeAction=lambda d: done.append(d.count))
    pipe.RunToCompletion()
    assert done == [count]
    for k, (xp, zp) in enumerate(positions):
        want = oracle.pipeline(res, res, octaves=9, noise_size=300, gauss_iterations=7, xpos=xp, zpos=zp)
        assert np.array_equal(batch.tile(k).ToArray((res, res)), want), (k, xp, zp)
    pipe.Destroy()
    # other stage bodies: every basis, a wide blur, the even-width quirk (grid-by-grid fallback), one application
    t = rng.random((count, res, res), dtype=f32)
    for basis in (1, 2, 5, 6):
        _ = run_batch(nj, nj.NoiseStage(ctx, nj.FractalNoise(basis), 0.5, 1.0, 4, 2.0, 0.0, 50), batch)
        for k, (xp, zp) in enumerate(positions):
            assert np.array_equal(batch.tile(k).ToArray((res, res)),
                                  oracle.fractal(basis, res, res, 0.5, 1.0, 2.0, 0.0, 4, xp, zp, 50)), (basis, k)
    for stage, fn in ((nj.StageGaussianBlur(ctx, 2, nj.GaussSigma(5), 13), lambda a: oracle.gauss(a, 13, 5, 2)),
                      (nj.StageGaussianBlur(ctx, 1, nj.GaussSigma(3), 5), lambda a: oracle.gauss(a, 5, 3, 1)),
                      (nj.KernelFilterStage(ctx, nj.KernelFilterType.Gauss9_S2, 4), lambda a: oracle.kernel_filter(a, 4, 4)),
                      (nj.ErosionStage(ctx, 1), lambda a: oracle.erosion_min(a, 1)),
                      (nj.FlowMapStage(ctx, 7, -0.1, 0.1), lambda a: oracle.flowmap(a, 7, -0.1, 0.1))):
        batch.data.CopyFrom(t)
        run_batch(nj, stage, batch)
        for k in range(count):
            assert np.array_equal(batch.tile(k).ToArray((res, res)), fn(t[k])), (type(stage).__name__, k)
        stage.OnDestroy()
    batch.data.Dispose()
    batch.positions.Dispose()


@pytest.mark.parametrize("mesh_type,res,in_res,margin", [(1, 64, 72, 4), (0, 33, 35, 1), (1, 127, 131, 2)])
def test_batched_meshes_equal_single_meshes(nj, ctx, oracle, mesh_type, res, in_res, margin):
    count = 4
    h = np.random.default_rng(res).random((count, in_res, in_res), dtype=f32)
    md = nj.MeshStageData("m", ctx.from_host(h), res, in_res, margin, 100.0, 25.0, count=count)
    st = nj.MeshTileStage(ctx, nj.MeshType(mesh_type))
    st.ReceiveHandledInput(nj.PipelineWorkItem(md), nj.JobHandle())
    st.jobHandle.Complete()
    vtx = md.mesh.vertices.ToArray().reshape(count, -1, 12)
    idx = md.mesh.index_array().reshape(count, -1)
    for k in range(count):
        v, i = oracle.mesh_heightmap(mesh_type, h[k], res, margin, 25.0, 100.0)
        assert np.array_equal(idx[k], i), k
        assert np.array_equal(vtx[k], v), k


def run_batch(nj, stage, batch):
    stage.ReceiveHandledInput(nj.PipelineWorkItem(batch), nj.JobHandle())
    stage.jobHandle.Complete()


@pytest.mark.parametrize("res", [1, 2, 17, 255, 256])
def test_square_planar_mesh(nj, ctx, oracle, res):
    # MeshJob<SharedSquareGridPosition, PositionStream32> (MeshHelper.makeSquarePlanarMesh)
    nv, ni = nj._native.lib.nz_mesh_vertex_count(res), nj._native.lib.nz_mesh_index_count(res)
    v, i = ctx.alloc(nv * 12), ctx.alloc(ni, dtype=np.uint32)
    ctx.call("nz_square_grid_mesh", v.ptr, i.ptr, res).Complete()
    vtx, idx = oracle.mesh_square_grid(res)
    assert np.array_equal(i.ToArray(), idx)
    assert np.array_equal(v.ToArray().reshape(-1, 12), vtx)
    v.Dispose()
    i.Dispose()


def test_mesh_from_a_context_buffer(nj, ctx, oracle):
    # MeshTileReferenceDataStage: heights come from the state manager's buffer named after the tile, not from the
    # work item; the stage waits until that buffer exists
    res, in_res, margin = 56, 64, 4
    mgr = nj.PipelineStateManager(ctx)
    h = np.random.default_rng(2).random((in_res, in_res), dtype=f32)
    mesher = nj.BasePipeline([nj.MeshTileReferenceDataStage(ctx, nj.MeshType.OvershootSquareGridHeightMap, "height")],
                             "mesher", contextManager=mgr)
    md = nj.MeshStageData("m", ctx.alloc(4), res, in_res, margin, 100.0, 30.0, xpos=7, zpos=-3)
    mesher.Enqueue(md)
    mesher.Update()
    assert not mesher.pipelineRunning and len(mesher.dependencyHell) == 1      # no buffer yet
    writer = nj.BasePipeline([nj.WriteGeneratorContextStage(ctx, "height")], "writer", contextManager=mgr)
    writer.Enqueue(nj.GeneratorData("h", ctx.from_host(h), in_res, 7, -3))
    writer.RunToCompletion()
    mesher.RunToCompletion()
    vtx, idx = oracle.mesh_heightmap(oracle.MESH_OVERSHOOT, h, res, margin, 30.0, 100.0)
    assert np.array_equal(md.mesh.index_array(), idx)
    assert np.array_equal(md.mesh.vertices.ToArray().reshape(-1, 12), vtx)
    mgr.OnDestroy()


@pytest.mark.parametrize("res", [2, 9, 64, 255])
def test_live_erosion_grid_jobs(nj, ctx, oracle, res):
    # UpdateFlowFromTrackJob and PoolAutomataJob (drainParticles == false): the deterministic part of the live erosion
    rng = np.random.default_rng(res)
    height = rng.random((res, res), dtype=f32)
    pool = np.where(rng.random((res, res)) < 0.35, rng.random((res, res), dtype=f32) * f32(0.3), 0).astype(f32)
    pool[0, :] = f32(0.05)          # border cells: a clamped neighbour is the cell itself
    flow = rng.random((res, res), dtype=f32)
    track = np.where(rng.random((res, res)) < 0.5, rng.random((res, res), dtype=f32), 0).astype(f32)
    d_pool, d_flow, d_track = ctx.from_host(pool), ctx.from_host(flow), ctx.from_host(track)
    ctx.call("nz_update_flow_from_track", d_pool.ptr, d_flow.ptr, d_track.ptr, 0.05, 0.1, 700.0, res).Complete()
    p, fl, tr = oracle.update_flow_from_track(pool, flow, track, 0.05, 0.1, 700.0)
    assert np.array_equal(d_pool.ToArray((res, res)), p) and np.array_equal(d_flow.ToArray((res, res)), fl)
    assert np.array_equal(d_track.ToArray((res, res)), tr)
    for iters in (1, 3):
        d_pool, d_h = ctx.from_host(pool), ctx.from_host(height)
        ctx.call("nz_pool_automata", d_pool.ptr, d_h.ptr, iters, res).Complete()
        assert np.array_equal(d_pool.ToArray((res, res)), oracle.pool_automata(pool, height, iters)), iters
    with pytest.raises(nj.NoizeError):
        ctx.call("nz_pool_automata", d_pool.ptr, d_pool.ptr, 1, res)


@pytest.mark.parametrize("res,cover", [(66, 1.0), (130, 0.6), (257, 1.0), (257, 0.03), (512, 0.9)])
def test_pool_automata_runs_match_the_row_walk(nj, ctx, oracle, res, cover):
    # the pass runs as parallel runs of acting steps (pool_runs_kernel): lakes that span mask words and whole rows,
    # isolated puddles, water below the 1E-3 threshold between them, odd sizes -- all equal to the oracle's row walk
    rng = np.random.default_rng(res + int(cover * 100))
    height = (rng.random((res, res), dtype=f32) * f32(0.2)).astype(f32)
    wet = rng.random((res, res)) < cover
    pool = np.where(wet, f32(0.002) + rng.random((res, res), dtype=f32) * f32(0.3), 0).astype(f32)
    pool[rng.random((res, res)) < 0.1] = f32(0.0005)      # standing water that does not act
    pool[:, res // 2] = f32(0.25)                          # one full line of water either way
    pool[res // 3, :] = f32(0.25)
    d_pool, d_h = ctx.from_host(pool), ctx.from_host(height)
    ctx.call("nz_pool_automata", d_pool.ptr, d_h.ptr, 2, res).Complete()
    assert np.array_equal(d_pool.ToArray((res, res)), oracle.pool_automata(pool, height, 2))


@pytest.mark.parametrize("res,wet", [(512, 1.0), (640, 0.93), (512, 0.6)])
def test_pool_automata_on_a_plane_under_water(nj, ctx, oracle, res, wet):
    # At least seven steps in eight acting: the job's launches walk whole rows, one lane each (the sparse launch decides
    # that on the device, from this job's own plane: ctl[3]); fewer: parallel runs.  Twice on the same context,
    # so that the second job also meets the first one's report.  All equal to the oracle's row walk.
    rng = np.random.default_rng(res + int(wet * 10))
    height = (rng.random((res, res), dtype=f32) * f32(0.2)).astype(f32)
    pool = np.where(rng.random((res, res)) < wet, f32(0.002) + rng.random((res, res), dtype=f32) * f32(0.3), 0).astype(f32)
    want = oracle.pool_automata(pool, height, 3)
    d_h = ctx.from_host(height)
    for _ in range(2):
        d_pool = ctx.from_host(pool)
        ctx.call("nz_pool_automata", d_pool.ptr, d_h.ptr, 3, res).Complete()
        assert np.array_equal(d_pool.ToArray((res, res)), want)
        d_pool.Dispose()
    d_h.Dispose()


@pytest.mark.parametrize("n", [1, 3, 255, 1024, 4099, 300 * 300, 2048 * 2048 + 5])
def test_get_map_range_and_device_args_normalise(nj, ctx, oracle, n):
    # GetMapRangeJob (Filter/NormalizeJob.cs:17-55) -> {min, max, range} in device memory -> NormalizeMap reading them
    rng = np.random.default_rng(n)
    for case in range(6):
        a = (rng.random(n, dtype=f32) * f32(4) - f32(1.5)).astype(f32)
        lim = (np.inf, -np.inf)
        if case == 1:            # NaN cells are skipped
            a[rng.random(n) < 0.3] = np.nan
        elif case == 2:          # minimum zero: the sign of the LAST zero cell stays
            a = np.abs(a); a[rng.integers(0, n, 5)] = f32(0.0); a[rng.integers(0, n, 5)] = f32(-0.0)
        elif case == 3:          # maximum zero
            a = -np.abs(a); a[rng.integers(0, n, 5)] = f32(-0.0); a[rng.integers(0, n, 5)] = f32(0.0)
        elif case == 4:          # limits inside the data's range, and a zero limit with no zero cell
            a = np.abs(a) + f32(0.25); lim = (-0.0, 1.0)
        elif case == 5:          # nothing but NaN
            a[:] = np.nan; lim = (2.0, -3.0)
        d, res = ctx.from_host(a), ctx.alloc(3)
        ctx.call("nz_get_map_range", d.ptr, n, res.ptr, lim[0], lim[1]).Complete()
        want = oracle.get_map_range(a, *lim)
        got = res.ToArray((3,))
        assert got.view(np.uint32).tolist() == want.view(np.uint32).tolist(), (case, got, want)
    side = int(np.sqrt(n))
    if side >= 2 and side * side == n:
        a = rng.random((side, side), dtype=f32)
        d, res = ctx.from_host(a), ctx.alloc(3)
        h = ctx.call("nz_get_map_range", d.ptr, n, res.ptr, np.inf, -np.inf)
        ctx.call("nz_map_normalize_values_dev", d.ptr, d.ptr, res.ptr, side, dep=h).Complete()
        assert np.array_equal(d.ToArray((side, side)), oracle.normalize_args(a, oracle.get_map_range(a)))


def test_demo_pipeline_with_invert_curve(nj, ctx, oracle):
    # BasicDemo "ParallelFlowMap": Perlin fBm -> Invert (curve) -> FlowMapStage -> CurveBoostContrast
    res = 128
    data = ctx.alloc(res * res)
    invert, boost = (lambda t: 1.0 - t), (lambda t: min(1.0, 4.0 * t))
    stages = [nj.NoiseStage(ctx, nj.FractalNoise.Perlin, 0.5938, 1.0, 6, 1.9168, 0.0317, 658), nj.CurveStage(ctx, invert, 256),
              nj.FlowMapStage(ctx, 1, 0.0, 0.005), nj.CurveStage(ctx, boost, 256)]
    pipe = nj.BasePipeline(stages)
    pipe.Enqueue(nj.GeneratorData("demo", data, res, 0, 0))
    pipe.RunToCompletion()
    want = oracle.fractal(oracle.PERLIN, res, res, 0.5938, 1.0, 1.9168, 0.0317, 6, 0, 0, 658)
    want = oracle.curve(want, np.array([invert(f32(i) / f32(256)) for i in range(256)], f32))
    want = oracle.flowmap(want, 1, 0.0, 0.005)
    want = oracle.curve(want, np.array([boost(f32(i) / f32(256)) for i in range(256)], f32))
    assert np.array_equal(data.ToArray((res, res)), want)


def test_thermal_erosion_stage(nj, ctx, oracle):
    for res in (2, 3, 4, 5, 8, 65, 256, 1030, 2048):   # 1030: rows not 16-byte aligned; 2048: more blocks than threads
        t = np.random.default_rng(res).random((res, res), dtype=f32)
        for iters, talus, inc, ratio in ((1, 45, 0.5, 0.75), (3, 20, 0.25, 0.3), (2, 80, 0.5, 2.0)):
            got = run(nj.StageThermalErosion(ctx, iters, talus, inc, ratio), nj, gen(nj, ctx, res, host=t))
            assert np.array_equal(got, oracle.thermal_erosion(t, float(talus), inc, ratio, iters)), (res, iters)


# ---- READ / WRITE plane pairs (nz_rw_tile): TileHelpers.SWAP_RWTILE as a swap ---------------------------------
def _pair(nj, ctx, res, host):
    d = nj.GeneratorData("rw", ctx.from_host(host), res, 0, 0, write=ctx.from_host(np.full((res, res), np.nan, f32)))
    return d, {d.data.ptr, d.write.ptr}


@pytest.mark.parametrize("res", [50, 257])
def test_rw_pair_stages_match_the_oracle(nj, ctx, oracle, res):
    t = adversarial_tiles(res)["uniform"]
    cases = [(nj.KernelFilterStage(ctx, nj.KernelFilterType(ft), it), lambda a, ft=ft, it=it: oracle.kernel_filter(a, ft, it))
             for ft, it in ((2, 1), (2, 17), (0, 3), (6, 7), (3, 4), (12, 2), (13, 1))]
    cases += [(nj.StageGaussianBlur(ctx, it, nj.GaussSigma(4), w),
               lambda a, w=w, it=it: oracle.gauss(a, oracle.limit_width(w), 4, it)) for w, it in ((5, 1), (13, 3), (25, 2), (8, 2))]
    cases += [(nj.StageSmoothBlur(ctx, 3, 11), lambda a: oracle.smooth(a, oracle.limit_width(11), 3))]
    cases += [(nj.ErosionStage(ctx, it), lambda a, it=it: oracle.erosion_min(a, it)) for it in (1, 2, 5, 8, 9, 21)]
    cases += [(nj.FlowMapStage(ctx, it, 0.0, 0.005), lambda a, it=it: oracle.flowmap(a, it, 0.0, 0.005)) for it in (1, 5, 7, 12)]
    for stage, want in cases:
        d, planes = _pair(nj, ctx, res, t)
        got = run(stage, nj, d)  # reads d.data after the stage: whichever plane holds the result
        assert np.array_equal(got, want(t)), (type(stage).__name__, vars(stage).get("iterations"))
        assert {d.data.ptr, d.write.ptr} == planes and d.data.ptr != d.write.ptr
        d.data.Dispose(); d.write.Dispose()


def test_rw_pair_pipeline_equals_the_in_place_pipeline(nj, ctx, oracle):
    res = 600

    def stages():
        return [nj.NoiseStage(ctx, nj.FractalNoise.Simplex, 0.4, 1.0, 13, 2.0, 0.0, 1700),
                nj.KernelFilterStage(ctx, nj.KernelFilterType.Gauss5_S1, 17), nj.FlowMapStage(ctx, 5, 0.0, 0.005),
                nj.ErosionStage(ctx, 5), nj.ConstantStage(ctx, nj.ConstantOperationType.MULTIPLY, 0.5)]
    outs = []
    for pair in (False, True):
        d = nj.GeneratorData("p", ctx.alloc(res * res), res, 40, -7, write=ctx.alloc(res * res) if pair else None)
        pipe = nj.BasePipeline(stages())
        seen = []
        pipe.Enqueue(d, completeAction=lambda x: seen.append(x.data.ToArray((res, res))))
        pipe.RunToCompletion()
        outs.append(seen[0])
        pipe.Destroy()
    assert np.array_equal(outs[0], outs[1])
    assert np.array_equal(outs[0], oracle.constant(oracle.pipeline(res, res, xpos=40, zpos=-7), 0, 0.5))


@pytest.mark.parametrize("res", [256, 280, 512, 768, 1000, 1024, 1536])
def test_metric_pipeline_at_the_references_tile_sizes(nj, ctx, oracle, res):
    """The reference's own tile sizes take launch shapes of their own (64-row filter tiles of 1024 x 2 or 512 x 4 rows, chained
    from two launches on; 32-, 48- or 64-row flow tiles, whichever covers the grid in one round of the CUs): the stock stage
    list on a single plane and on a READ / WRITE pair, bit for bit the oracle's tile."""
    want = oracle.pipeline(res, res, xpos=3 * res, zpos=-res)
    for pair in (False, True):
        d = nj.GeneratorData("p", ctx.alloc(res * res), res, 3 * res, -res, write=ctx.alloc(res * res) if pair else None)
        pipe = nj.BasePipeline([nj.NoiseStage(ctx, nj.FractalNoise.Simplex, 0.4, 1.0, 13, 2.0, 0.0, 1700),
                                nj.KernelFilterStage(ctx, nj.KernelFilterType.Gauss5_S1, 17),
                                nj.FlowMapStage(ctx, 5, 0.0, 0.005), nj.ErosionStage(ctx, 5)])
        seen = []
        pipe.Enqueue(d, completeAction=lambda x: seen.append(x.data.ToArray((res, res))))
        pipe.RunToCompletion()
        pipe.Destroy()
        assert np.array_equal(seen[0], want), (res, pair, int((seen[0] != want).sum()))
        d.data.Dispose()
        if pair:
            d.write.Dispose()


def test_rw_pair_batch_and_errors(nj, ctx, oracle):
    res, count = 96, 3
    b = nj.GeneratorDataBatch.create(ctx, "b", res, [(96 * k, -3 * k) for k in range(count)])
    b.write = ctx.alloc(count * res * res)
    pipe = nj.BasePipeline([nj.NoiseStage(ctx, nj.FractalNoise.Simplex, 0.4, 1.0, 13, 2.0, 0.0, 1700),
                            nj.KernelFilterStage(ctx, nj.KernelFilterType.Gauss5_S1, 17),
                            nj.FlowMapStage(ctx, 5, 0.0, 0.005), nj.ErosionStage(ctx, 5)])
    pipe.Enqueue(b)
    pipe.RunToCompletion()
    got = b.data.ToArray((count, res, res))
    for k in range(count):
        assert np.array_equal(got[k], oracle.pipeline(res, res, xpos=96 * k, zpos=-3 * k)), k
    N = nj._native
    same = N.RWTile(b.data.ptr, b.data.ptr, res, 1)
    with pytest.raises(nj.NoizeError):
        ctx.call("nz_erosion_stage_rw", C.byref(same), 2)
    t = N.RWTile(b.data.ptr, b.write.ptr, res, 1)
    with pytest.raises(nj.NoizeError):
        ctx.call("nz_kernel_filter_stage_rw", C.byref(t), int(nj.KernelFilterType.Sobel3_2D), 1)
    with pytest.raises(nj.NoizeError):
        ctx.call("nz_flowmap_stage_rw", C.byref(t), None, 5, 0.0, 0.005)
    assert (t.read, t.write) == (b.data.ptr, b.write.ptr)  # a refused call leaves the pair alone


def test_mesh_16_bit_index_stream(nj, ctx, oracle):
    # PositionStream16 / TriangleUInt16 (Mesh/Streams/PositionStream.cs:11-74, Triangle.cs:7-17): same vertices, indices
    # cast to ushort; 255^2 quads is the largest mesh whose (R + 1)^2 vertices fit, 300 shows the documented wrap-around
    rng = np.random.default_rng(16)
    for res in (64, 255, 300):
        in_res = res + 8
        h = rng.random((in_res, in_res), dtype=f32)
        heights = ctx.from_host(h)
        nv, ni = (res + 1) ** 2, 6 * res * res
        vtx, idx = ctx.alloc(nv * 12), ctx.alloc((ni + 1) // 2, dtype=np.uint32)
        ctx.call("nz_heightmap_mesh16", int(nj.MeshType.OvershootSquareGridHeightMap), vtx.ptr, idx.ptr, res, in_res, 4, 50.0,
                 100.0, heights.ptr).Complete()
        wv, wi = oracle.mesh_heightmap(oracle.MESH_OVERSHOOT, h, res, 4, 50.0, 100.0)
        got = idx.ToArray().view(np.uint16)[:ni]
        assert np.array_equal(got, wi.astype(np.uint16)) and np.array_equal(vtx.ToArray().reshape(-1, 12), wv)
        assert (got.astype(np.uint32) == wi).all() == (nv <= 65536)
        for t in (heights, vtx, idx):
            t.Dispose()
