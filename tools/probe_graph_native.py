#!/usr/bin/env python3
"""Probe (round 6): where a replayed tile request's time goes -- nz_pipeline_graph_launch with / without a handle, against the four
stage entries with / without handles.  usage: probe_graph_native.py [res ...]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import noize_job_amd as nj  # noqa: E402
N = nj._native

for res in [int(a) for a in sys.argv[1:]] or [512, 1024]:
    with nj.Context(0) as ctx:
        stages = [nj.NoiseStage(ctx, nj.FractalNoise.Simplex, 0.4, 1.0, 13, 2.0, 0.0, 1700),
                  nj.KernelFilterStage(ctx, nj.KernelFilterType.Gauss5_S1, 17), nj.FlowMapStage(ctx, 5, 0.0, 0.005),
                  nj.ErosionStage(ctx, 5)]
        g = nj.PipelineGraph(ctx, stages, res)
        a, b = ctx.alloc(res * res), ctx.alloc(res * res)
        n = 2000

        def timed(fn):
            for _ in range(20):
                fn(0)
            ctx.synchronize()
            t0 = time.perf_counter()
            for k in range(n):
                fn(k)
            th = time.perf_counter() - t0
            ctx.synchronize()
            return (time.perf_counter() - t0) / n * 1e6, th / n * 1e6

        t = N.RWTile(a.ptr, b.ptr, res, 1)
        out = N.handle_t()

        def replay_handle(k):
            N.check(N.lib.nz_pipeline_graph_launch(ctx._h, g._h, C.byref(t), res * k, 0, 0, C.byref(out)), "launch")

        def replay_nohandle(k):
            N.check(N.lib.nz_pipeline_graph_launch(ctx._h, g._h, C.byref(t), res * k, 0, 0, None), "launch")

        pipe = nj.BasePipeline(stages, "eager")
        gd = nj.GeneratorData("t", a, res, 0, 0, write=b)

        def eager(k):
            gd.xpos = res * k
            pipe.Schedule(gd)
            pipe.pipelineRunning = False

        work = ctx.alloc(10 * res * res)
        t2 = N.RWTile(a.ptr, b.ptr, res, 1)

        def entries(last_handle):
            def fn(k):
                N.check(N.lib.nz_fractal(ctx._h, 3, t2.read, res, 0.4, 1.0, 2.0, 0.0, 13, res * k, 0, 1700, 0, None), "n")
                N.check(N.lib.nz_kernel_filter_stage_rw(ctx._h, C.byref(t2), 2, 17, 0, None), "g")
                N.check(N.lib.nz_flowmap_stage_rw(ctx._h, C.byref(t2), work.ptr, 5, 0.0, 0.005, 0, None), "f")
                N.check(N.lib.nz_erosion_stage_rw(ctx._h, C.byref(t2), 5, 0, C.byref(out) if last_handle else None), "e")
            return fn

        for name, fn in (("entries, no handles", entries(False)), ("entries, last handle only", entries(True)),
                         ("stage entries (handles)", eager), ("graph replay + handle", replay_handle),
                         ("graph replay, no handle", replay_nohandle), ("stage entries (handles)", eager)):
            us, host = timed(fn)
            print("res %5d  %-26s %7.2f us per tile (host %5.2f us)  %8.0f tiles/s" % (res, name, us, host, 1e6 / us))
        g.Destroy()
