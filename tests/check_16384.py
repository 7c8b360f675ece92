#!/usr/bin/env python3
"""One-off: BASELINE config 5's grid (16384^2) as ONE tile on one GPU against the CPU oracle, bit for bit.
(~14 GiB of device memory, ~20 s of oracle time on 32 cores.)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import noize_job_amd as nj  # noqa: E402
import oracle as O  # noqa: E402

res = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
with nj.Context(0) as ctx:
    data = ctx.alloc(res * res)
    stages = [nj.NoiseStage(ctx, nj.FractalNoise.Simplex, 0.4, 1.0, 13, 2.0, 0.0, 1700),
              nj.KernelFilterStage(ctx, nj.KernelFilterType.Gauss5_S1, 17), nj.FlowMapStage(ctx, 5, 0.0, 0.005),
              nj.ErosionStage(ctx, 5)]
    pipe = nj.BasePipeline(stages, "big")
    t0 = time.perf_counter()
    pipe.Enqueue(nj.GeneratorData("big", data, res, 0, 0))
    pipe.RunToCompletion()
    t1 = time.perf_counter()
    got = data.ToArray((res, res))
    print("GPU %dx%d: %.1f ms incl. first-use allocations" % (res, res, (t1 - t0) * 1e3), flush=True)
    t0 = time.perf_counter()
    want = O.pipeline(res, res)
    print("oracle: %.1f s" % (time.perf_counter() - t0), flush=True)
    same = np.array_equal(got, want)
    print("bit-equal:", same, " max |diff| %.3g" % float(np.abs(got - want).max()))
    pipe.Destroy()
    sys.exit(0 if same else 1)
