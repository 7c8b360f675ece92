#!/usr/bin/env python3
"""Probe (round 6): tools/probe_graph.py on a READ / WRITE plane pair (the form the tile pipelines use): eager stage calls
against a replayed capture of the same calls."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import noize_job_amd as nj  # noqa: E402

for res in (256, 512, 1024):
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ctx = nj.Context(0, stream=stream.cuda_stream)
        data = torch.empty(res * res, dtype=torch.float32, device="cuda")
        wr = torch.empty(res * res, dtype=torch.float32, device="cuda")
        t0_, t1_ = ctx.wrap(data.data_ptr(), res * res), ctx.wrap(wr.data_ptr(), res * res)
        stages = [nj.NoiseStage(ctx, nj.FractalNoise.Simplex, 0.4, 1.0, 13, 2.0, 0.0, 1700),
                  nj.KernelFilterStage(ctx, nj.KernelFilterType.Gauss5_S1, 17), nj.FlowMapStage(ctx, 5, 0.0, 0.005),
                  nj.ErosionStage(ctx, 5)]
        pipe = nj.BasePipeline(stages, "g")

        def step():
            gd = nj.GeneratorData("g", t0_, res, 0, 0, write=t1_)
            pipe.Schedule(gd)
            pipe.pipelineRunning = False
            return gd

        for _ in range(5):
            gd = step()
        stream.synchronize()
        n = 500
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        stream.synchronize()
        eager = (time.perf_counter() - t0) / n * 1e3
        res_t = data if gd.data.ptr == t0_.ptr else wr
        want = res_t.clone()
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, stream=stream):
                step()
            for _ in range(5):
                g.replay()
            stream.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                g.replay()
            stream.synchronize()
            graph = (time.perf_counter() - t0) / n * 1e3
            same = bool(torch.equal(res_t, want))
            print("res %4d pair: eager %.4f ms per pipeline, graph replay %.4f ms, same result: %s" % (res, eager, graph, same))
        except Exception as e:  # noqa: BLE001
            print("res %4d: eager %.4f ms; capture failed: %s" % (res, eager, str(e).splitlines()[0]))
        ctx.close()
