#!/usr/bin/env python3
"""Probe: do compute-bound and memory-bound stages of two independent pipelines overlap on the GPU when
they are issued on two streams, staggered by one stage?  Prints ms per pipeline for (a) one stream,
(b) two streams in lockstep, (c) two streams staggered."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import noize_job_amd as nj  # noqa: E402

R, STEPS = 4096, 100  # long enough for the clocks to settle


def make(ctx):
    data = ctx.alloc(R * R)
    stages = [nj.NoiseStage(ctx, nj.FractalNoise.Simplex, 0.4, 1.0, 13, 2.0, 0.0, 1700),
              nj.KernelFilterStage(ctx, nj.KernelFilterType.Gauss5_S1, 17), nj.FlowMapStage(ctx, 5, 0.0, 0.005),
              nj.ErosionStage(ctx, 5)]
    gd = nj.GeneratorData("p", data, R, 0, 0, write=ctx.alloc(R * R))  # READ / WRITE pair
    wi = nj.PipelineWorkItem(gd)
    for s in stages:
        s.Schedule(wi, nj.JobHandle())
    ctx.synchronize()
    return stages, wi


def main():
    a, b = nj.Context(0), nj.Context(0)
    sa, wa = make(a)
    sb, wb = make(b)
    h0 = nj.JobHandle()

    def run_all(stages, wi):
        for s in stages:
            s.Schedule(wi, h0)

    # (a) one stream: 2*STEPS pipelines back to back
    t = time.perf_counter()
    for _ in range(2 * STEPS):
        run_all(sa, wa)
    a.synchronize()
    ta = (time.perf_counter() - t) / (2 * STEPS)
    # (b) two streams, same stage issued on both at the same time
    t = time.perf_counter()
    for _ in range(STEPS):
        for x, y in zip(sa, sb):
            x.Schedule(wa, h0)
            y.Schedule(wb, h0)
    a.synchronize(); b.synchronize()
    tb = (time.perf_counter() - t) / (2 * STEPS)
    # (c) two streams, B one stage behind A (host issue order only; streams are independent)
    t = time.perf_counter()
    sa[0].Schedule(wa, h0)
    for _ in range(STEPS):
        for i in range(4):
            sb[i].Schedule(wb, h0)
            sa[(i + 1) % 4].Schedule(wa, h0)
    a.synchronize(); b.synchronize()
    tc = (time.perf_counter() - t) / (2 * STEPS)
    print("ms per pipeline: one stream %.4f | two streams lockstep %.4f | two streams staggered %.4f" %
          (ta * 1e3, tb * 1e3, tc * 1e3))


if __name__ == "__main__":
    main()
