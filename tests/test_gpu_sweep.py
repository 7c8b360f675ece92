"""Seeded random sweep: every stencil stage against the oracle over many tile sizes, iteration counts
and (for the stripe entry points) random stripe geometry with a row pitch wider than the row.  All
comparisons are bit-exact.  Sizes straddle the kernels' tile shapes (64 x 128 / 48 x 128 / 32 x 128
workgroup tiles, 4-cell vectors) so interior, edge and unaligned paths are all taken."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32

SIZES = [4, 7, 16, 31, 33, 63, 64, 65, 100, 127, 128, 129, 131, 191, 200, 257, 300, 385]


def _run(nj, stage, d):
    stage.ReceiveHandledInput(nj.PipelineWorkItem(d), nj.JobHandle())
    stage.jobHandle.Complete()
    stage.OnDestroy()
    out = d.data.ToArray((d.resolution, d.resolution))
    d.data.Dispose()
    return out


def _tile(rng, res):
    kind = rng.integers(0, 4)
    if kind == 0:
        return rng.random((res, res), dtype=f32)
    if kind == 1:  # large dynamic range, both signs
        return ((rng.random((res, res), dtype=f32) - f32(0.5)) * f32(10.0) ** rng.integers(-3, 4)).astype(f32)
    if kind == 2:  # plateaus: ties for the min filter, zero slopes for the flow
        return np.round(rng.random((res, res), dtype=f32) * f32(4.0)).astype(f32) * f32(0.25)
    t = np.zeros((res, res), f32)  # sparse impulses incl. the corners
    t[0, 0] = t[-1, -1] = t[0, -1] = 1.0
    t[rng.integers(0, res), rng.integers(0, res)] = -2.0
    return t


@pytest.mark.parametrize("seed", range(10))
def test_filters_erosion_flow_random_sizes(nj, ctx, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    for _ in range(7):
        res = int(rng.choice(SIZES))
        t = _tile(rng, res)
        ft = int(rng.choice([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13]))
        it = int(rng.integers(1, 12)) if ft != 11 else int(rng.integers(1, 3))
        got = _run(nj, nj.KernelFilterStage(ctx, nj.KernelFilterType(ft), it), nj.GeneratorData("k", ctx.from_host(t), res))
        assert np.array_equal(got, oracle.kernel_filter(t, ft, it)), ("filter", res, ft, it)
        width, sigma, it = int(rng.integers(1, 27)), int(rng.integers(0, 16)), int(rng.integers(1, 4))
        got = _run(nj, nj.StageGaussianBlur(ctx, it, nj.GaussSigma(sigma), width), nj.GeneratorData("g", ctx.from_host(t), res))
        assert np.array_equal(got, oracle.gauss(t, oracle.limit_width(width), sigma, it)), ("gauss", res, width, sigma, it)
        width, it = int(rng.integers(1, 26)), int(rng.integers(1, 4))
        got = _run(nj, nj.StageSmoothBlur(ctx, it, width), nj.GeneratorData("s", ctx.from_host(t), res))
        assert np.array_equal(got, oracle.smooth(t, oracle.limit_width(width), it)), ("smooth", res, width, it)  # the stage passes limitWidth(width)
        it = int(rng.integers(1, 14))
        got = _run(nj, nj.ErosionStage(ctx, it), nj.GeneratorData("e", ctx.from_host(t), res))
        assert np.array_equal(got, oracle.erosion_min(t, it)), ("erosion", res, it)
        it = int(rng.integers(1, 13))
        lo, hi = [(0.0, 0.005), (-0.1, 0.1), (0.0, 0.0)][int(rng.integers(0, 3))]
        got = _run(nj, nj.FlowMapStage(ctx, it, lo, hi), nj.GeneratorData("f", ctx.from_host(t), res))
        # a zero range divides 0 by 0 (NormalizeMap, FlowMapComponents.cs:157-165): NaN planes on both sides
        assert np.array_equal(got, oracle.flowmap(t, it, lo, hi), equal_nan=True), ("flow", res, it, lo, hi)
        it = int(rng.integers(1, 4))
        if res >= 4:
            got = _run(nj, nj.StageThermalErosion(ctx, it, 45, 0.5, 0.75), nj.GeneratorData("t", ctx.from_host(t), res))
            assert np.array_equal(got, oracle.thermal_erosion(t, 45.0, 0.5, 0.75, it)), ("thermal", res, it)


@pytest.mark.parametrize("seed", range(6))
def test_rw_pair_random_stage_chains(nj, ctx, oracle, seed):
    # chains of stencil stages on a READ / WRITE plane pair (nz_*_rw: the pair is swapped, never flushed), with an
    # in-place element-wise stage thrown in, against the oracle's composition
    rng = np.random.default_rng(9000 + seed)
    for _ in range(5):
        res = int(rng.choice(SIZES))
        t = _tile(rng, res)
        d = nj.GeneratorData("rw", ctx.from_host(t), res, write=ctx.from_host(np.full((res, res), np.nan, f32)))
        planes = {d.data.ptr, d.write.ptr}
        want, desc = t, []
        for _ in range(int(rng.integers(1, 5))):
            kind = int(rng.integers(0, 6))
            if kind == 0:
                ft = int(rng.choice([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 13]))
                it = int(rng.integers(1, 20))
                stage, want = nj.KernelFilterStage(ctx, nj.KernelFilterType(ft), it), oracle.kernel_filter(want, ft, it)
            elif kind == 1:
                w, sg, it = int(rng.integers(1, 27)), int(rng.integers(0, 16)), int(rng.integers(1, 4))
                stage, want = nj.StageGaussianBlur(ctx, it, nj.GaussSigma(sg), w), oracle.gauss(want, oracle.limit_width(w), sg, it)
            elif kind == 2:
                w, it = int(rng.integers(1, 26)), int(rng.integers(1, 4))
                stage, want = nj.StageSmoothBlur(ctx, it, w), oracle.smooth(want, oracle.limit_width(w), it)
            elif kind == 3:
                it = int(rng.integers(1, 20))
                stage, want = nj.ErosionStage(ctx, it), oracle.erosion_min(want, it)
            elif kind == 4:
                it = int(rng.integers(1, 13))
                stage, want = nj.FlowMapStage(ctx, it, -0.1, 0.1), oracle.flowmap(want, it, -0.1, 0.1)
            else:
                stage, want = nj.ConstantStage(ctx, nj.ConstantOperationType.MULTIPLY, 0.75), oracle.constant(want, 0, 0.75)
            desc.append((type(stage).__name__, vars(stage).get("iterations")))
            stage.ReceiveHandledInput(nj.PipelineWorkItem(d), nj.JobHandle())
            stage.jobHandle.Complete()
            stage.OnDestroy()
            assert {d.data.ptr, d.write.ptr} == planes
        assert np.array_equal(d.data.ToArray((res, res)), want, equal_nan=True), (res, desc)
        d.data.Dispose(); d.write.Dispose()


@pytest.mark.parametrize("seed", range(3))
def test_non_finite_cells_propagate_like_the_oracle(nj, ctx, oracle, seed):
    # NaN / +-inf cells: sums and products spread them the IEEE way, min / max drop a NaN operand
    # (Unity.Mathematics math.min / math.max, C fminf / fmaxf, v_min_f32 / v_max_f32)
    rng = np.random.default_rng(5000 + seed)
    for res in (33, 130, 200):
        t = rng.random((res, res), dtype=f32)
        for v in (np.nan, np.inf, -np.inf, np.nan):
            t[rng.integers(0, res), rng.integers(0, res)] = v
        t[0, 0] = np.nan
        t[-1, -1] = np.inf
        eq = lambda a, b: np.array_equal(a, b, equal_nan=True)  # noqa: E731
        got = _run(nj, nj.KernelFilterStage(ctx, nj.KernelFilterType.Gauss5_S1, 4), nj.GeneratorData("k", ctx.from_host(t), res))
        assert eq(got, oracle.kernel_filter(t, 2, 4)), ("filter", res)
        got = _run(nj, nj.StageGaussianBlur(ctx, 1, nj.GaussSigma(3), 21), nj.GeneratorData("g", ctx.from_host(t), res))
        assert eq(got, oracle.gauss(t, 21, 3, 1)), ("wide blur", res)
        for it in (1, 3, 6):
            got = _run(nj, nj.ErosionStage(ctx, it), nj.GeneratorData("e", ctx.from_host(t), res))
            assert eq(got, oracle.erosion_min(t, it)), ("erosion", res, it)
        for it in (1, 5, 7):
            got = _run(nj, nj.FlowMapStage(ctx, it, 0.0, 0.005), nj.GeneratorData("f", ctx.from_host(t), res))
            assert eq(got, oracle.flowmap(t, it, 0.0, 0.005)), ("flow", res, it)
        got = _run(nj, nj.StageThermalErosion(ctx, 2, 45, 0.5, 0.75), nj.GeneratorData("t", ctx.from_host(t), res))
        assert eq(got, oracle.thermal_erosion(t, 45.0, 0.5, 0.75, 2)), ("thermal", res)
        # math.clamp on a NaN (max(0, min(1, NaN)) = 1): the curve's input clamp, and the flow map's outflow scale where a cell
        # without water faces a flux sum that underflows (round 6: the soak's seed 62067 -- sparse impulses on a plane of zeros,
        # twelve iterations: the water differences that drive the flow there are denormal)
        sq = lambda u: u * u  # noqa: E731
        got = _run(nj, nj.CurveStage(ctx, sq, 9), nj.GeneratorData("c", ctx.from_host(t), res))
        assert eq(got, oracle.curve(t, np.array([sq(f32(i) / f32(9)) for i in range(9)], f32))), ("curve", res)
        z = np.zeros((res, res), f32)
        z[0, 0] = z[-1, -1] = z[0, -1] = 1.0
        z[res // 3, 4] = -2.0
        z[res // 2, res // 2] = 1e-45
        for it in (3, 12):
            got = _run(nj, nj.FlowMapStage(ctx, it, 0.0, 0.005), nj.GeneratorData("f", ctx.from_host(z), res))
            want = oracle.flowmap(z, it, 0.0, 0.005)
            assert eq(got, want) and np.isfinite(want).all(), ("flow on impulses", res, it)


@pytest.mark.parametrize("basis", range(1, 8))
def test_noise_random_parameters(nj, ctx, oracle, basis):
    # every basis but Sin (device sinf) is bit-exact: random fBm parameters, positions of both signs around the
    # lattice cells where fp32 mod289 misbehaves, small and negative noise sizes, one tile beyond the table range
    rng = np.random.default_rng(4000 + basis)
    for k in range(10):
        res = int(rng.choice([17, 64, 96, 130]))
        octv = int(rng.integers(1, 14))
        hurst = float(f32(rng.random()))
        amp = float(f32(rng.random() * 3.0 + 0.1))
        step = float(f32(rng.choice([2.0, 1.9168, 2.5, 1.5, 0.7])))
        det = float(f32(rng.choice([0.0, 0.0, 0.0317])))
        ns = int(rng.choice([1, 2, 7, 100, 658, 1700, -13]))
        span = 20000 * abs(ns) if k < 9 else 3000000 * abs(ns)
        xp, zp = int(rng.integers(-span, span)), int(rng.integers(-span, span))
        st = nj.NoiseStage(ctx, nj.FractalNoise(basis), hurst, amp, octv, step, det, ns)
        d = nj.GeneratorData("n", ctx.alloc(res * res), res, xp, zp)
        got = _run(nj, st, d)
        want = oracle.fractal(basis, res, res, hurst, amp, step, det, octv, xp, zp, ns)
        assert np.array_equal(got, want, equal_nan=True), (basis, res, octv, hurst, amp, step, det, ns, xp, zp,
                                                            int((got != want).sum()))


@pytest.mark.parametrize("seed", range(4))
def test_mesh_random_shapes(nj, ctx, oracle, seed):
    rng = np.random.default_rng(2000 + seed)
    for _ in range(6):
        res = int(rng.choice([2, 3, 7, 16, 33, 64, 100, 255, 256, 300]))
        margin = int(rng.integers(2, 6))  # overshoot with margin 1 reads past the row end in the reference (B17)
        mesh_type = int(rng.integers(0, 2))
        in_res = res + 2 * margin if mesh_type == 1 else res + int(rng.choice([1, 2, 2 * margin]))
        h = rng.random((in_res, in_res), dtype=f32)
        height, size = float(rng.choice([1.0, 10.0, 1000.0])), float(rng.choice([1.0, 512.0, 1000.0]))
        md = nj.MeshStageData("m", ctx.from_host(h), res, in_res, margin, size, height)
        st = nj.MeshTileStage(ctx, nj.MeshType(mesh_type))
        st.ReceiveHandledInput(nj.PipelineWorkItem(md), nj.JobHandle())
        st.jobHandle.Complete()
        vtx, idx = oracle.mesh_heightmap(mesh_type, h, res, margin, height, size)
        assert np.array_equal(md.mesh.index_array(), idx), (mesh_type, res, in_res)
        assert np.array_equal(md.mesh.vertices.ToArray().reshape(-1, 12), vtx), (mesh_type, res, in_res)
        md.data.Dispose()
        md.mesh.vertices.Dispose()
        md.mesh.indices.Dispose()


@pytest.mark.parametrize("seed", range(4))
def test_stripe_entry_points_random_geometry_and_pitch(nj, ctx, oracle, seed):
    # a stripe of a larger grid: buffer rows [grow0, grow0 + rows), produced rows [own0, own1), row pitch >= cols
    rng = np.random.default_rng(3000 + seed)
    for _ in range(6):
        cols = int(rng.choice([5, 33, 64, 130, 200, 257]))
        grows = int(rng.choice([40, 97, 150, 260]))
        pitch = cols + int(rng.choice([0, 0, 1, 4, 13]))
        ft = int(rng.choice([0, 1, 2, 3, 8]))
        T = int(rng.integers(1, 1 + nj._native.lib.nz_kernel_filter_max_fused(ft)))
        halo = nj._native.lib.nz_kernel_filter_halo_rows(ft, T)
        grid = rng.random((grows, cols), dtype=f32)
        want = oracle.kernel_filter(grid, ft, T)
        g0 = int(rng.integers(0, grows - 1))
        g1 = int(rng.integers(g0 + 1, grows + 1))
        b0, b1 = max(0, g0 - halo - int(rng.integers(0, 3))), min(grows, g1 + halo + int(rng.integers(0, 3)))
        rows = b1 - b0
        buf = np.full((rows, pitch), np.nan, f32)
        buf[:, :cols] = grid[b0:b1]
        src, dst = ctx.from_host(buf), ctx.from_host(np.full((rows, pitch), np.nan, f32))
        st = nj.Stripe(cols, rows, b0, grows, g0 - b0, g1 - b0, pitch)
        ctx.call("nz_kernel_filter_stripe", src.ptr, dst.ptr, C.byref(st), ft, T).Complete()
        out = dst.ToArray((rows, pitch))
        assert np.array_equal(out[g0 - b0:g1 - b0, :cols], want[g0:g1]), ("conv", cols, grows, pitch, ft, T, g0, g1)
        assert np.isnan(out[:g0 - b0]).all() and np.isnan(out[g1 - b0:]).all() and np.isnan(out[:, cols:]).all()
        # erosion: E applications reach E rows upwards only
        E = int(rng.integers(1, 1 + nj._native.lib.nz_erosion_max_fused_iterations()))
        want = oracle.erosion_min(grid, E)
        b0 = max(0, g0 - E)
        rows = g1 - b0
        buf = np.full((rows, pitch), np.nan, f32)
        buf[:, :cols] = grid[b0:g1]
        src.Dispose(); dst.Dispose()
        src, dst = ctx.from_host(buf), ctx.from_host(np.full((rows, pitch), np.nan, f32))
        st = nj.Stripe(cols, rows, b0, grows, g0 - b0, g1 - b0, pitch)
        ctx.call("nz_erosion_stripe", src.ptr, dst.ptr, C.byref(st), E).Complete()
        out = dst.ToArray((rows, pitch))
        assert np.array_equal(out[g0 - b0:, :cols], want[g0:g1]), ("erosion", cols, grows, pitch, E, g0, g1)
        src.Dispose(); dst.Dispose()


def test_chained_filter_launches_tolerate_a_straggling_tile(nj, ctx, oracle):
    # The launches of a filter stage run as ONE grid whose tiles wait for the previous launch's tiles they read -- and
    # overwrite the plane that launch read.  A tile held up before its loads (nz_debug_chain_delay: ~0.3 ms, several
    # launches' worth) must still find its input intact: the tiles of the next launch that store into its window are made
    # to wait for it (write after read), whatever the fusion depths along the chain are.
    lib = nj._native.lib
    res = 2816   # 7.9 M cells: below ~7 M a stage runs 64-row tiles as separate launches and nothing is chained
    h = oracle.fractal(oracle.SIMPLEX, res, res, 0.4, 1.0, 2.0, 0.0, 6, 0, 0, 300)
    want = oracle.kernel_filter(h, int(nj.KernelFilterType.Gauss5_S1), 17)
    tiles0 = -(-res // 112) ** 2   # launch 0 of 17 = 4 + 4 + 4 + 5 applications: 112 x 112 interiors
    data, write = ctx.from_host(h), ctx.alloc(res * res)
    stage = nj.KernelFilterStage(ctx, nj.KernelFilterType.Gauss5_S1, 17)
    try:
        for item in (0, 7, tiles0 // 2 + 3, tiles0 - 1, tiles0 + 5, 2 * tiles0 + 11, 3 * tiles0 - 2):
            assert lib.nz_debug_chain_delay(item, 90) == 0
            data.CopyFrom(h)
            gd = nj.GeneratorData("t", data, res, 0, 0, write=write)
            stage.ReceiveHandledInput(nj.PipelineWorkItem(gd), nj.JobHandle())
            stage.jobHandle.Complete()
            assert np.array_equal(gd.data.ToArray((res, res)), want), item
    finally:
        lib.nz_debug_chain_delay(-1, 0)
    stage.OnDestroy()


def test_a_chained_launch_that_times_out_is_reported_and_the_pipeline_runs_again(nj, oracle):
    # The chained form's wait is bounded: a tile whose producer does not come raises the context's error word (mapped
    # host memory: no device-to-host copy on the wait path).  The host's wait reports NZ_ERR_RETRY, the context switches
    # to separate launches for good, and a BasePipeline that regenerates its tile (NoiseStage first) schedules the work
    # item again on its own -- its result is the oracle's.  Forced here with a poll limit of 8 (~16 us) and one
    # launch-0 tile held up for ~1.7 ms.
    lib = nj._native.lib
    res = 2816   # big enough for the chained form (7 M cells and more)
    want = oracle.pipeline(res, res, octaves=6, noise_size=300, gauss_iterations=17, flow_iterations=0, erosion_iterations=0)
    env = os.environ.get
    if (env("NZ_CONV_CHAIN", "1") == "0" or env("NZ_CONV_STREAM", "1") == "2"
            or (env("NZ_CONV_SMALL") == "2" and env("NZ_CONV_CHAIN", "1") != "2")):
        pytest.skip("knob matrix: the chained form is switched off for this grid (every launch a streaming one, every grid "
                    "on the small-grid tiles, or more launches than a chain holds)")
    with nj.Context(0) as c:
        data, write = c.alloc(res * res), c.alloc(res * res)
        stages = [nj.NoiseStage(c, nj.FractalNoise.Simplex, 0.4, 1.0, 6, 2.0, 0.0, 300),
                  nj.KernelFilterStage(c, nj.KernelFilterType.Gauss5_S1, 17)]
        try:
            assert lib.nz_debug_chain_poll_limit(8) == 0 and lib.nz_debug_chain_delay(3, 500) == 0
            # the delegate-level call: the error reaches the caller
            gd = nj.GeneratorData("t", data, res, 0, 0, write=write)
            h = nj.JobHandle()
            for st in stages:
                st.Schedule(nj.PipelineWorkItem(gd), h)
                h = st.jobHandle
            with pytest.raises(nj.NoizeError) as e:
                h.Complete()
            assert e.value.status == nj._native.NZ_ERR_RETRY
            # ... and from now on the context runs the stage as separate launches: no time-out, the right plane
            gd = nj.GeneratorData("t", data, res, 0, 0, write=write)
            h = nj.JobHandle()
            for st in stages:
                st.Schedule(nj.PipelineWorkItem(gd), h)
                h = st.jobHandle
            h.Complete()
            assert np.array_equal(gd.data.ToArray((res, res)), want)
        finally:
            lib.nz_debug_chain_poll_limit(0)
            lib.nz_debug_chain_delay(-1, 0)
        for st in stages:
            st.OnDestroy()
    # a fresh context (chained form on again), the same time-out, through BasePipeline: the retry is the pipeline's
    with nj.Context(0) as c:
        data, write = c.alloc(res * res), c.alloc(res * res)
        pipe = nj.BasePipeline([nj.NoiseStage(c, nj.FractalNoise.Simplex, 0.4, 1.0, 6, 2.0, 0.0, 300),
                                nj.KernelFilterStage(c, nj.KernelFilterType.Gauss5_S1, 17)])
        try:
            assert lib.nz_debug_chain_poll_limit(8) == 0 and lib.nz_debug_chain_delay(3, 500) == 0
            gd = nj.GeneratorData("t", data, res, 0, 0, write=write)
            done = []
            pipe.Enqueue(gd, completeAction=lambda d: done.append(d))
            pipe.RunToCompletion()
            assert len(done) == 1 and np.array_equal(gd.data.ToArray((res, res)), want)
        finally:
            lib.nz_debug_chain_poll_limit(0)
            lib.nz_debug_chain_delay(-1, 0)
        pipe.Destroy()
    # The failure belongs to the work that caused it.  (a) A handle recorded on the same context BEFORE the stage that fails
    # (another job's fence) completes clean, even when it is waited for after the launch has given up; the stage's own handle
    # and every handle issued after it report NZ_ERR_RETRY, however often they are waited for; handles issued after the host
    # has noticed are clean again.  (b) A pipeline whose handles are in other hands (a scheduleAction) does not re-run on its
    # own: the status goes to the caller.
    import time
    with nj.Context(0) as c:
        data, write = c.alloc(res * res), c.alloc(res * res)
        stages = [nj.NoiseStage(c, nj.FractalNoise.Simplex, 0.4, 1.0, 6, 2.0, 0.0, 300),
                  nj.KernelFilterStage(c, nj.KernelFilterType.Gauss5_S1, 17)]
        try:
            assert lib.nz_debug_chain_poll_limit(8) == 0 and lib.nz_debug_chain_delay(3, 500) == 0
            before = c.record()
            gd = nj.GeneratorData("t", data, res, 0, 0, write=write)
            h = nj.JobHandle()
            for st in stages:
                st.Schedule(nj.PipelineWorkItem(gd), h)
                h = st.jobHandle
            after = c.record()
            time.sleep(0.05)     # the launch has given up by now
            before.Complete()    # an unrelated, older handle: no error
            for hh in (h, after, h):
                with pytest.raises(nj.NoizeError) as e:
                    hh.Complete()
                assert e.value.status == nj._native.NZ_ERR_RETRY
            c.synchronize()      # (reported through the handles already)
            later = c.record()
            later.Complete()
        finally:
            lib.nz_debug_chain_poll_limit(0)
            lib.nz_debug_chain_delay(-1, 0)
        for st in stages:
            st.OnDestroy()
    # (a') The host first notices on a wait for an OLDER handle (clean), has not been told yet, and schedules a dependent of the
    # failed stage: that handle -- issued after the notice -- reports the failure too (the window closes at the first report).
    with nj.Context(0) as c:
        data, write = c.alloc(res * res), c.alloc(res * res)
        stages = [nj.NoiseStage(c, nj.FractalNoise.Simplex, 0.4, 1.0, 6, 2.0, 0.0, 300),
                  nj.KernelFilterStage(c, nj.KernelFilterType.Gauss5_S1, 17)]
        ero = nj.ErosionStage(c, 2)
        try:
            assert lib.nz_debug_chain_poll_limit(8) == 0 and lib.nz_debug_chain_delay(3, 500) == 0
            before = c.record()
            gd = nj.GeneratorData("t", data, res, 0, 0, write=write)
            h = nj.JobHandle()
            for st in stages:
                st.Schedule(nj.PipelineWorkItem(gd), h)
                h = st.jobHandle
            time.sleep(0.05)
            before.Complete()                            # the notice: clean, nothing reported
            ero.Schedule(nj.PipelineWorkItem(gd), h)     # a dependent of the invalid plane, issued after the notice
            with pytest.raises(nj.NoizeError) as e:
                ero.jobHandle.Complete()
            assert e.value.status == nj._native.NZ_ERR_RETRY
            later = c.record()                           # issued after the report: clean
            later.Complete()
        finally:
            lib.nz_debug_chain_poll_limit(0)
            lib.nz_debug_chain_delay(-1, 0)
        for st in stages + [ero]:
            st.OnDestroy()
    with nj.Context(0) as c:
        data, write = c.alloc(res * res), c.alloc(res * res)
        pipe = nj.BasePipeline([nj.NoiseStage(c, nj.FractalNoise.Simplex, 0.4, 1.0, 6, 2.0, 0.0, 300),
                                nj.KernelFilterStage(c, nj.KernelFilterType.Gauss5_S1, 17)])
        try:
            assert lib.nz_debug_chain_poll_limit(8) == 0 and lib.nz_debug_chain_delay(3, 500) == 0
            gd = nj.GeneratorData("t", data, res, 0, 0, write=write)
            scheduled, done = [], []
            pipe.Enqueue(gd, scheduleAction=lambda d, hdl: scheduled.append(hdl), completeAction=lambda d: done.append(d))
            with pytest.raises(nj.NoizeError) as e:
                pipe.RunToCompletion()
            assert e.value.status == nj._native.NZ_ERR_RETRY and len(scheduled) == 1 and not done
        finally:
            lib.nz_debug_chain_poll_limit(0)
            lib.nz_debug_chain_delay(-1, 0)
        pipe.Destroy()
