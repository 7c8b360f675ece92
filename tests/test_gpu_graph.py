"""nz_pipeline_graph: a tile request as ONE replayed HIP graph of the stock stage list (include/noize_hip.h; the reference's
BasePipeline.Schedule per tile, Scripts/MeshTileGenerator.cs:181-211, Pipeline/Executable/Pipeline.cs:104-128).
The replay must be what the four stage entries compute -- bit for bit, for every request, also with requests in flight."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _stages(nj, c, g=17, f=5, e=5, noise=None, octaves=13):
    st = [nj.NoiseStage(c, noise if noise is not None else nj.FractalNoise.Simplex, 0.4, 1.0, octaves, 2.0, 0.0, 1700)]
    if g:
        st.append(nj.KernelFilterStage(c, nj.KernelFilterType.Gauss5_S1, g))
    if f:
        st.append(nj.FlowMapStage(c, f, 0.0, 0.005))
    if e:
        st.append(nj.ErosionStage(c, e))
    return st


@pytest.mark.parametrize("res", [256, 280, 512, 1000])
def test_replayed_requests_equal_stage_by_stage_and_the_oracle(nj, ctx, oracle, res):
    eager = nj.BasePipeline(_stages(nj, ctx), "eager")
    graph = nj.BasePipeline(_stages(nj, ctx), "graph", replay=True)
    a, b = ctx.alloc(res * res), ctx.alloc(res * res)
    c, d = ctx.alloc(res * res), ctx.alloc(res * res)
    ge = nj.GeneratorData("e", a, res, 0, 0, write=b)
    gg = nj.GeneratorData("g", c, res, 0, 0, write=d)
    positions = [(0, 0), (res, 0), (7 * res, -3 * res), (0, 0), (123, 456), (res, 0)]
    for k, (x, z) in enumerate(positions):
        for gd, pipe in ((ge, eager), (gg, graph)):
            gd.xpos, gd.zpos = x, z
            pipe.Schedule(gd)
            pipe.pipelineRunning = False
        eager.pipelineHandle.Complete()
        graph.pipelineHandle.Complete()
        want = ge.data.ToArray((res, res))
        got = gg.data.ToArray((res, res))
        assert np.array_equal(got, want), (res, k, x, z)
        # the pair comes back the way the stage entries leave it
        assert (gg.data.ptr == c.ptr) == (ge.data.ptr == a.ptr)
        if k in (0, 2):
            assert np.array_equal(got, oracle.pipeline(res, res, oracle.SIMPLEX, 0.4, 1.0, 2.0, 0.0, 13, x, z, 1700,
                                                       oracle.GAUSS5_S1, 17, 5, 0.0, 0.005, 5))
    g = graph._graphs[res]
    assert 1 <= g.captures <= 8          # one per orientation of the pair and mailbox slot, never one per request
    eager.Destroy(); graph.Destroy()
    for t in (a, b, c, d):
        t.Dispose()


def test_requests_in_flight_keep_their_own_position(nj, ctx, oracle):
    # three replays enqueued back to back, each followed by an asynchronous download on the same stream: the position
    # patched for request k + 1 must not reach request k
    import ctypes as C
    N = nj._native
    res = 384
    g = nj.PipelineGraph(ctx, _stages(nj, ctx, g=6, f=2, e=3, octaves=8), res)
    a, b = ctx.alloc(res * res), ctx.alloc(res * res)
    gd = nj.GeneratorData("t", a, res, 0, 0, write=b)
    for warm in range(8):                # both orientations x both slots captured (each capture serves its request eagerly)
        g.Launch(gd)
    ctx.synchronize()
    n0 = g.captures
    outs, pos = [], [(1000, 0), (0, 2000), (-300, 77), (5, 5), (1 << 20, -(1 << 21)), (3, 4), (0, 0)]
    for x, z in pos:
        gd.xpos, gd.zpos = x, z
        g.Launch(gd)
        host = np.empty(res * res, np.float32)
        N.check(N.lib.nz_tile_download(ctx._h, gd.data.ptr, host.ctypes.data, res * res, 0, None), "download")
        outs.append(host)
    ctx.synchronize()
    assert g.captures == n0 <= 8
    for (x, z), host in zip(pos, outs):
        want = oracle.pipeline(res, res, oracle.SIMPLEX, 0.4, 1.0, 2.0, 0.0, 8, x, z, 1700, oracle.GAUSS5_S1, 6, 2, 0.0, 0.005, 3)
        assert np.array_equal(host.reshape(res, res), want), (x, z)
    g.Destroy()
    a.Dispose(); b.Dispose()


@pytest.mark.parametrize("stages", [dict(g=17, f=0, e=0), dict(g=0, f=5, e=0), dict(g=0, f=0, e=5), dict(g=3, f=0, e=2),
                                    dict(g=17, f=5, e=5, noise="Cellular"), dict(g=9, f=7, e=1, noise="Perlin")])
def test_partial_stage_lists_and_other_bases(nj, ctx, stages):
    res = 320
    kw = dict(stages)
    noise = getattr(nj.FractalNoise, kw.pop("noise", "Simplex"))
    eager = nj.BasePipeline(_stages(nj, ctx, noise=noise, octaves=6, **kw), "eager")
    graph = nj.BasePipeline(_stages(nj, ctx, noise=noise, octaves=6, **kw), "graph", replay=True)
    planes = [ctx.alloc(res * res) for _ in range(4)]
    ge = nj.GeneratorData("e", planes[0], res, 0, 0, write=planes[1])
    gg = nj.GeneratorData("g", planes[2], res, 0, 0, write=planes[3])
    for x, z in [(0, 0), (640, 320), (-5, 9)]:
        for gd, pipe in ((ge, eager), (gg, graph)):
            gd.xpos, gd.zpos = x, z
            pipe.Schedule(gd)
            pipe.pipelineRunning = False
        ctx.synchronize()
        assert np.array_equal(gg.data.ToArray((res, res)), ge.data.ToArray((res, res))), (stages, x, z)
    assert graph._graphs, "the stage list did not take the graph path"
    eager.Destroy(); graph.Destroy()
    for t in planes:
        t.Dispose()


def test_what_is_not_the_stock_list_runs_stage_by_stage(nj, ctx):
    res = 256
    st = _stages(nj, ctx) + [nj.ConstantStage(ctx, nj.ConstantOperationType.MULTIPLY, 2.0)]
    assert nj.stock_list_params(st) is None and nj.stock_list_params(st[:-1]) is not None
    pipe = nj.BasePipeline(st, "custom", replay=True)
    gd = nj.GeneratorData("t", ctx.alloc(res * res), res, 0, 0, write=ctx.alloc(res * res))
    pipe.Schedule(gd)
    pipe.pipelineHandle.Complete()
    assert not pipe._graphs
    # ... and so does the stock list on a single plane (no WRITE plane to swap with)
    pipe2 = nj.BasePipeline(_stages(nj, ctx), "single", replay=True)
    g1 = nj.GeneratorData("t", ctx.alloc(res * res), res, 0, 0)
    pipe2.Schedule(g1)
    pipe2.pipelineHandle.Complete()
    assert not pipe2._graphs
    with pytest.raises(Exception):
        nj.PipelineGraph(ctx, st, res)
    pipe.Destroy(); pipe2.Destroy()


def test_a_change_of_float_mode_captures_again(nj, oracle):
    res = 512
    with nj.Context(0) as c:
        g = nj.PipelineGraph(c, _stages(nj, c), res)
        gd = nj.GeneratorData("t", c.alloc(res * res), res, 0, 0, write=c.alloc(res * res))
        want = oracle.pipeline(res, res, oracle.SIMPLEX, 0.4, 1.0, 2.0, 0.0, 13, 0, 0, 1700, oracle.GAUSS5_S1, 17, 5, 0.0, 0.005, 5)
        for _ in range(10):
            g.Launch(gd)
        c.synchronize()
        n0 = g.captures
        assert np.array_equal(gd.data.ToArray((res, res)), want)
        c.float_mode = 1
        for _ in range(10):
            g.Launch(gd)
        c.synchronize()
        fast = gd.data.ToArray((res, res))
        assert g.captures > n0 and not np.array_equal(fast, want) and np.abs(fast - want).max() < 2e-3
        c.float_mode = 0
        for _ in range(10):
            g.Launch(gd)
        c.synchronize()
        assert np.array_equal(gd.data.ToArray((res, res)), want)
        g.Destroy()


def test_a_chained_launch_that_times_out_inside_a_replay_is_reported_and_captured_again(nj, oracle):
    # the filter stage of a replay is the chained grid (two launches and more on a small grid): its bounded wait, the error word
    # and the retry window work as for the stage entry; the next launch captures the separate launches the context now runs
    lib = nj._native.lib
    res = 1024
    if os.environ.get("NZ_CONV_CHAIN", "1") == "0":
        pytest.skip("knob matrix: no chained form")
    want = oracle.pipeline(res, res, oracle.SIMPLEX, 0.4, 1.0, 2.0, 0.0, 6, 0, 0, 300, oracle.GAUSS5_S1, 17, 0, 0.0, 0.005, 0)
    with nj.Context(0) as c:
        st = [nj.NoiseStage(c, nj.FractalNoise.Simplex, 0.4, 1.0, 6, 2.0, 0.0, 300),
              nj.KernelFilterStage(c, nj.KernelFilterType.Gauss5_S1, 17)]
        g = nj.PipelineGraph(c, st, res)
        gd = nj.GeneratorData("t", c.alloc(res * res), res, 0, 0, write=c.alloc(res * res))
        for _ in range(10):
            g.Launch(gd)
        c.synchronize()
        assert np.array_equal(gd.data.ToArray((res, res)), want)
        n0 = g.captures
        try:
            assert lib.nz_debug_chain_poll_limit(8) == 0 and lib.nz_debug_chain_delay(3, 500) == 0
            # (the debug fields travel in the chained kernel's argument block, which a capture bakes in: capture with them)
            g2 = nj.PipelineGraph(c, st, res)
            with pytest.raises(nj.NoizeError) as e:
                for _ in range(8):       # the captures' eager passes time out already; so do the replays
                    g2.Launch(gd).Complete()
            assert e.value.status == nj._native.NZ_ERR_RETRY
        finally:
            lib.nz_debug_chain_poll_limit(0)
            lib.nz_debug_chain_delay(-1, 0)
        # both graphs capture again (separate launches from now on) and give the right plane
        for gr in (g, g2):
            for _ in range(9):
                gr.Launch(gd).Complete()
                assert np.array_equal(gd.data.ToArray((res, res)), want)
        assert g.captures > n0
        g.Destroy(); g2.Destroy()
