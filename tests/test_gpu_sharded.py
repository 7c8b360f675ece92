"""Row-stripe sharding on the real kernels: the stripes of a grid computed by the C-ABI stripe entry
points (all ranks rehearsed in one process on one GPU, ghost rows copied device-to-device) must equal
the monolithic single-tile run of the same kernels bit for bit, and the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run_lockstep(nj, world, grows, cols, p):
    import torch
    from noize_job_amd import sharded as sh
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ctx = nj.Context(0, stream=stream.cuda_stream)
        ops = sh.HipStripeOps(ctx)
        halo = sh.halo_rows_needed(ops, p)
        plans = [sh.StripePlan(r, world, grows, cols, halo, neighbours_own_halo=p.haloMode != "recompute")
                 for r in range(world)]
        nan = float("nan")
        bufs = [(torch.full((pl.rows, cols), nan, device="cuda"), torch.full((pl.rows, cols), nan, device="cuda"),
                 torch.full((5, pl.rows, cols), nan, device="cuda"), torch.full((5, pl.rows, cols), nan, device="cuda"))
                for pl in plans]

        def copy_rows(dst, d0, src, s0, n):
            dst[d0:d0 + n].copy_(src[s0:s0 + n])

        res = sh.run_pipeline_lockstep([ops] * world, plans, p, bufs, copy_rows)
        stream.synchronize()
        out = np.concatenate([r[pl.own0:pl.own1].cpu().numpy() for r, pl in zip(res, plans)], axis=0)
        ctx.close()
    return out


@pytest.mark.parametrize("world,mode", [(1, "exchange"), (2, "exchange"), (4, "exchange"), (8, "exchange"),
                                        (1, "recompute"), (2, "recompute"), (8, "recompute"), (16, "recompute"),
                                        (2, "exchange_once"), (8, "exchange_once")])
def test_sharded_equals_monolithic_and_oracle(nj, ctx, oracle, world, mode):
    # "recompute" at 16 ranks: 32-row stripes with 49 ghost rows, so a rank recomputes rows of two neighbours
    from noize_job_amd import sharded as sh
    res = 512
    p = sh.PipelineParams(octaves=13, gaussIterations=17, flowIterations=5, erosionIterations=5, xpos=100, zpos=900,
                          haloMode=mode)
    got = _run_lockstep(nj, world, res, res, p)
    # monolithic run of the same kernels through the tile API
    data = ctx.alloc(res * res)
    stages = [nj.NoiseStage(ctx, nj.FractalNoise.Simplex, 0.4, 1.0, 13, 2.0, 0.0, 1700),
              nj.KernelFilterStage(ctx, nj.KernelFilterType.Gauss5_S1, 17), nj.FlowMapStage(ctx, 5, 0.0, 0.005),
              nj.ErosionStage(ctx, 5)]
    pipe = nj.BasePipeline(stages)
    pipe.Enqueue(nj.GeneratorData("mono", data, res, 100, 900))
    pipe.RunToCompletion()
    mono = data.ToArray((res, res))
    assert np.array_equal(got, mono)
    assert np.array_equal(got, oracle.pipeline(res, res, xpos=100, zpos=900))
    pipe.Destroy()
    data.Dispose()


@pytest.mark.parametrize("mode", ["exchange", "recompute"])
def test_rectangular_stripes_uneven_split(nj, oracle, mode):
    from noize_job_amd import sharded as sh
    grows, cols = 333, 200  # rows do not divide by the world size, cols are not a multiple of the tile width
    p = sh.PipelineParams(octaves=8, noiseSize=300, gaussIterations=5, flowIterations=3, erosionIterations=7,
                          haloMode=mode)
    got = _run_lockstep(nj, 3, grows, cols, p)
    want = oracle.pipeline(grows, cols, octaves=8, noise_size=300, gauss_iterations=5, flow_iterations=3,
                           erosion_iterations=7)
    assert np.array_equal(got, want)


def test_stripe_entry_points_validate_ghost_rows(nj, ctx):
    import ctypes as C
    a, b = ctx.alloc(64 * 16), ctx.alloc(64 * 16)
    st = nj.Stripe(16, 64, 100, 1000, 1, 63, 0)  # only one ghost row: a 3-application launch needs 6
    with pytest.raises(nj.NoizeError):
        ctx.call("nz_kernel_filter_stripe", a.ptr, b.ptr, C.byref(st), 2, 3)
    st = nj.Stripe(16, 64, 100, 1000, 6, 58, 0)
    ctx.call("nz_kernel_filter_stripe", a.ptr, b.ptr, C.byref(st), 2, 3).Complete()
