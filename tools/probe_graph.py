#!/usr/bin/env python3
"""Probe: the metric pipeline of one small tile captured in a HIP graph (through torch's capture API on the
context's stream) against eager launches -- does a graph shorten the GPU-side gaps between dependent kernels?"""
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
        gd = nj.GeneratorData("g", ctx.wrap(data.data_ptr(), res * res), res, 0, 0)
        stages = [nj.NoiseStage(ctx, nj.FractalNoise.Simplex, 0.4, 1.0, 13, 2.0, 0.0, 1700),
                  nj.KernelFilterStage(ctx, nj.KernelFilterType.Gauss5_S1, 17), nj.FlowMapStage(ctx, 5, 0.0, 0.005),
                  nj.ErosionStage(ctx, 5)]
        pipe = nj.BasePipeline(stages, "g")

        def step():
            pipe.Schedule(gd)
            pipe.pipelineRunning = False

        for _ in range(5):
            step()
        stream.synchronize()
        n = 300
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        stream.synchronize()
        eager = (time.perf_counter() - t0) / n * 1e3
        want = data.clone()
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
            same = bool(torch.equal(data, want))
            print("res %4d: eager %.4f ms per pipeline, graph replay %.4f ms, same result: %s" % (res, eager, graph, same))
        except Exception as e:  # noqa: BLE001
            print("res %4d: eager %.4f ms; capture failed: %s" % (res, eager, str(e).splitlines()[0]))
        ctx.close()
