#!/usr/bin/env python3
"""Strict against tolerance mode (nz_ctx_set_float_mode), stage by stage on one stream: back-to-back launch time of the four
metric stages on a READ / WRITE plane pair, and of the whole step, in both modes, alternating (A B A B) on one box.
usage: bench_modes.py [--res 4096] [--reps 200] [--rounds 3]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import noize_job_amd as nj  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    res = a.res
    with nj.Context(0) as ctx:
        gd = nj.GeneratorData("b", ctx.alloc(res * res), res, 0, 0, write=ctx.alloc(res * res))
        stages = {"noise": nj.NoiseStage(ctx, nj.FractalNoise.Simplex, 0.4, 1.0, 13, 2.0, 0.0, 1700),
                  "gauss": nj.KernelFilterStage(ctx, nj.KernelFilterType.Gauss5_S1, 17),
                  "flow": nj.FlowMapStage(ctx, 5, 0.0, 0.005), "erosion": nj.ErosionStage(ctx, 5)}
        wi = nj.PipelineWorkItem(gd)

        def timed(fn):
            for _ in range(20):
                fn()
            h0 = ctx.record()
            for _ in range(a.reps):
                fn()
            h1 = ctx.record()
            h1.Complete()
            return ctx.elapsed_ms(h0, h1) / a.reps

        def step():
            for st in stages.values():
                st.Schedule(wi, nj.JobHandle())
        for rnd in range(a.rounds):
            for mode, name in ((0, "strict"), (1, "fast"), (2, "relaxed")):
                ctx.float_mode = mode
                stages["noise"].Schedule(wi, nj.JobHandle())
                row = {n: timed(lambda st=st: st.Schedule(wi, nj.JobHandle())) for n, st in stages.items()}
                row["step"] = timed(step)
                print("round %d %-7s " % (rnd, name) + "  ".join("%s %.4f" % (k, v) for k, v in row.items()) +
                      "  (%.0f Mcells/s)" % (res * res / row["step"] / 1e3), flush=True)


if __name__ == "__main__":
    main()
