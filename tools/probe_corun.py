#!/usr/bin/env python3
"""Probe: how much do two DIFFERENT stages gain from sharing the chip?  For every pair of {noise, Gauss x17, flow x5}:
N launches of each alone, then both loops at once on two streams.  co-run factor = t(both) / (t(a) + t(b)): 1.0 = no
gain over running them one after the other, max(a, b) / (a + b) = perfect overlap."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import noize_job_amd as nj  # noqa: E402

R, N = 4096, 150


def make(ctx):
    gd = nj.GeneratorData("p", ctx.alloc(R * R), R, 0, 0, write=ctx.alloc(R * R))
    wi = nj.PipelineWorkItem(gd)
    st = {"noise": nj.NoiseStage(ctx, nj.FractalNoise.Simplex, 0.4, 1.0, 13, 2.0, 0.0, 1700),
          "gauss": nj.KernelFilterStage(ctx, nj.KernelFilterType.Gauss5_S1, 17),
          "flow": nj.FlowMapStage(ctx, 5, 0.0, 0.005)}
    for s in st.values():
        s.Schedule(wi, nj.JobHandle())
    ctx.synchronize()
    return st, wi


def main():
    a, b = nj.Context(0), nj.Context(0)
    sa, wa = make(a)
    sb, wb = make(b)
    h0 = nj.JobHandle()

    def loop(jobs):
        for _ in range(30):
            for s, w in jobs:
                s.Schedule(w, h0)
        a.synchronize(); b.synchronize()
        t = time.perf_counter()
        for _ in range(N):
            for s, w in jobs:
                s.Schedule(w, h0)
        a.synchronize(); b.synchronize()
        return (time.perf_counter() - t) / N * 1e3

    alone = {k: loop([(sa[k], wa)]) for k in sa}
    print("alone (ms): " + "  ".join("%s %.4f" % kv for kv in alone.items()))
    for x, y in (("noise", "gauss"), ("noise", "flow"), ("gauss", "flow"), ("noise", "noise"), ("gauss", "gauss"), ("flow", "flow")):
        t = loop([(sa[x], wa), (sb[y], wb)])
        print("%s || %s: %.4f ms per pair = %.3f of the sum (perfect overlap %.3f)" %
              (x, y, t, t / (alone[x] + alone[y]), max(alone[x], alone[y]) / (alone[x] + alone[y])))


if __name__ == "__main__":
    main()
