#!/usr/bin/env python3
"""How long does the host take to enqueue one pass of the metric pipeline, against the GPU time of
the pass?  (If the host is slower, the stream runs dry between launches.)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import noize_job_amd as nj  # noqa: E402

res = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
with nj.Context(0) as ctx:
    data = ctx.alloc(res * res)
    stages = [nj.NoiseStage(ctx, nj.FractalNoise.Simplex, 0.4, 1.0, 13, 2.0, 0.0, 1700),
              nj.KernelFilterStage(ctx, nj.KernelFilterType.Gauss5_S1, 17), nj.FlowMapStage(ctx, 5, 0.0, 0.005),
              nj.ErosionStage(ctx, 5)]
    pipe = nj.BasePipeline(stages, "metric")
    gd = nj.GeneratorData("bench", data, res, 0, 0)

    def step():
        pipe.Schedule(gd)
        pipe.pipelineRunning = False

    for _ in range(5):
        step()
    ctx.synchronize()
    for steps in (1, 5, 20, 100):
        h0 = ctx.record()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        t1 = time.perf_counter()
        h1 = ctx.record()
        h1.Complete()
        t2 = time.perf_counter()
        print("steps=%3d  host enqueue %.3f ms/step   wall %.3f ms/step   gpu(events) %.3f ms/step" % (
            steps, (t1 - t0) / steps * 1e3, (t2 - t0) / steps * 1e3, ctx.elapsed_ms(h0, h1) / steps))
