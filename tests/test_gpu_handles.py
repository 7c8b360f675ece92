"""JobHandles across contexts (HIP streams): a handle names the context that issued it, and a dependency on another
context's handle is a device-side wait (hipStreamWaitEvent), as Unity JobHandles cross pipelines in the reference
(Pipeline/Executable/ReducePipeline.cs:82-148, Pipeline/PipelineState/PipelineStateLock.cs:12-39)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32


def test_handles_name_their_context(nj, ctx):
    with nj.Context(0) as other:
        a, b = ctx.record(), other.record()
        ida, idb = nj._native.lib.nz_handle_context_id(a.id), nj._native.lib.nz_handle_context_id(b.id)
        assert ida == nj._native.lib.nz_ctx_id(ctx._h) and idb == nj._native.lib.nz_ctx_id(other._h) and ida != idb
        # query / wait resolve the owner from the handle, whichever live context is passed
        b.Complete()
        assert nj.JobHandle(ctx, b.id).IsCompleted
        c = nj.JobHandle.CombineDependencies(ctx, a, b, nj.JobHandle())
        c.Complete()
        assert c.IsCompleted and nj._native.lib.nz_handle_context_id(c.id) == ida
        stale = b
    # the issuing context is gone: nz_ctx_destroy synchronised its stream, so its handles read as completed
    assert nj.JobHandle(ctx, stale.id).IsCompleted
    nj.JobHandle(ctx, stale.id).Complete()
    with pytest.raises(nj.NoizeError):  # a value no context of this process ever issued
        ctx.call("nz_fill_array", ctx.alloc(16).ptr, 4, 0.0, dep=(1000 << 40) | 5)
    with pytest.raises(nj.NoizeError):  # right context, sequence number from the future
        ctx.call("nz_fill_array", ctx.alloc(16).ptr, 4, 0.0, dep=a.id + 10 ** 6)


def test_dependency_on_another_contexts_handle_orders_the_streams(nj, ctx, oracle):
    # producer (context A): 13-octave cellular noise, 2048^2 -- ~0.2 ms of GPU work; consumer (context B): Gauss5 x3 on
    # the same plane, enqueued immediately with the producer's handle as dependency and NO host wait in between
    res = 2048
    with nj.Context(0) as a, nj.Context(0) as b:
        plane = a.alloc(res * res)
        want = oracle.kernel_filter(oracle.fractal(oracle.CELLULAR, res, res, 0.5, 1.0, 2.0, 0.0, 13, 5, 9, 400), 2, 3)
        noise = nj.NoiseStage(a, nj.FractalNoise.Cellular, 0.5, 1.0, 13, 2.0, 0.0, 400)
        blur = nj.KernelFilterStage(b, nj.KernelFilterType.Gauss5_S1, 3)
        for rep in range(5):  # repeated: a race would not lose every time
            a.call("nz_fill_array", plane.ptr, res, float("nan"))
            a.synchronize()
            d = nj.GeneratorData("x", plane, res, 5, 9)
            noise.ReceiveHandledInput(nj.PipelineWorkItem(d), nj.JobHandle())
            blur.ReceiveHandledInput(nj.PipelineWorkItem(d), noise.jobHandle)
            blur.jobHandle.Complete()
            assert np.array_equal(plane.ToArray((res, res)), want), rep
        blur.OnDestroy()
        plane.Dispose()


def test_handles_that_ride_on_a_launch_complete_with_it(nj, oracle):
    # The stage entries of the metric pipeline hand out handles that ride on their last kernel launch (no event record of
    # their own).  Producer (context A): noise -> Gauss5 x17 -> flow x5 -> erosion x5 on a READ / WRITE pair, ~0.2 ms of GPU
    # work at 2048^2; consumer (context B): a copy of the result plane, enqueued at once with the LAST stage's handle as
    # its dependency and no host wait in between -- a handle that completed before its kernel would copy a half-made plane.
    # Then the same with the filter stage's handle and a copy of what the filter left.
    res = 2048
    want = oracle.pipeline(res, res, octaves=6, noise_size=500, xpos=11, zpos=-7)
    want_f = oracle.kernel_filter(oracle.fractal(oracle.SIMPLEX, res, res, 0.4, 1.0, 2.0, 0.0, 6, 11, -7, 500), 2, 17)
    with nj.Context(0) as a, nj.Context(0) as b:
        p0, p1, snap = a.alloc(res * res), a.alloc(res * res), b.alloc(res * res)
        stages = [nj.NoiseStage(a, nj.FractalNoise.Simplex, 0.4, 1.0, 6, 2.0, 0.0, 500),
                  nj.KernelFilterStage(a, nj.KernelFilterType.Gauss5_S1, 17), nj.FlowMapStage(a, 5, 0.0, 0.005),
                  nj.ErosionStage(a, 5)]
        for upto, expect in ((4, want), (2, want_f)):
            for rep in range(4):  # repeated: a race would not lose every time
                b.call("nz_fill_array", snap.ptr, res, float("nan"))
                b.synchronize()
                d = nj.GeneratorData("x", p0, res, 11, -7, write=p1)
                h = nj.JobHandle()
                for st in stages[:upto]:
                    st.Schedule(nj.PipelineWorkItem(d), h)
                    h = st.jobHandle
                assert h.id != 0
                done = b.call("nz_flush_write_slice", snap.ptr, d.data.ptr, res * res, dep=h)
                done.Complete()
                assert h.IsCompleted
                assert np.array_equal(snap.ToArray((res, res)), expect), (upto, rep)
                assert a.elapsed_ms(stages[0].jobHandle, h) > 0.0  # the riding events carry time stamps like recorded ones
        for st in stages:
            st.OnDestroy()
        for t in (p0, p1, snap):
            t.Dispose()


def test_reduce_pipeline_joins_two_contexts_on_the_device(nj, ctx, oracle):
    # ReducePipeline.cs:82-148 with the two upstream pipelines on their own contexts (streams) and the reduce stages
    # on a third: deviceJoin schedules the reduce stages behind CombineDependencies(left, right) -- no host wait
    # between the three pipelines -- and the result equals the one-stream run bit for bit
    res = 512
    lut = np.array([1.0 - f32(i) / f32(256) for i in range(256)], f32)

    def build(cl, cr, cj, device_join):
        left = nj.BasePipeline([nj.NoiseStage(cl, nj.FractalNoise.Simplex, 0.4, 1.0, 13, 2.0, 0.0, 300),
                                nj.KernelFilterStage(cl, nj.KernelFilterType.Gauss5_S1, 5)], "left")
        right = nj.BasePipeline([nj.NoiseStage(cr, nj.FractalNoise.Cellular, 0.5, 1.0, 13, 2.0, 0.0, 90)], "right")
        red = nj.ReducePipeline(cj, [nj.ReduceStage(cj, nj.ReductionType.MULTIPLY),
                                     nj.CurveStage(cj, lambda t: 1.0 - t, 256)], left, right, "reduce",
                                deviceJoin=device_join)
        return left, right, red

    with nj.Context(0) as cl, nj.Context(0) as cr:
        results = []
        for (a, b, c, dj) in ((ctx, ctx, ctx, False), (cl, cr, ctx, True)):
            left, right, red = build(a, b, c, dj)
            tiles = [nj.GeneratorData("t%d" % i, c.alloc(res * res), res, 37 * i, 11) for i in range(3)]
            done = []
            for t in tiles:
                red.Enqueue(t, completeAction=lambda d: done.append(d.uuid))
            if dj:
                # drive the frame loop WITHOUT completing anything until every pipeline has been scheduled once
                for pl in (red, left, right):
                    pl.Update()
                assert red.pipelineRunning and left.pipelineRunning and right.pipelineRunning
                assert nj._native.lib.nz_handle_context_id(left.pipelineHandle.id) != \
                    nj._native.lib.nz_handle_context_id(right.pipelineHandle.id)
            red.RunToCompletion()
            assert done == ["t0", "t1", "t2"]
            results.append([t.data.ToArray((res, res)) for t in tiles])
            for pl in (red, left, right):
                pl.Destroy()
        for i in range(3):
            x = oracle.kernel_filter(oracle.fractal(oracle.SIMPLEX, res, res, 0.4, 1.0, 2.0, 0.0, 13, 37 * i, 11, 300), 2, 5)
            y = oracle.fractal(oracle.CELLULAR, res, res, 0.5, 1.0, 2.0, 0.0, 13, 37 * i, 11, 90)
            want = oracle.curve(oracle.reduce(x, y, 1), lut)
            assert np.array_equal(results[0][i], want) and np.array_equal(results[1][i], want), i


def test_contexts_report_their_device(nj):
    # nz_ctx_device: what a host needs to put a second context (another pipeline, a tile server) on the first context's device
    n = nj.Context.device_count()
    for dev in range(min(n, 2)):
        with nj.Context(dev) as c:
            assert nj._native.lib.nz_ctx_device(c._h) == dev
    if n < 2:
        pytest.skip("one device visible: the second-device half needs two")
