#!/usr/bin/env python3
"""Fixed cost per launch: one stage enqueued `reps` times back to back (no host sync, no markers in
between) at several resolutions.  usage: probe_backtoback.py [stage] [--reps 20]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import noize_job_amd as nj  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("stage", nargs="?", default="noise")
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    with nj.Context(0) as ctx:
        for res in (1024, 2048, 4096, 8192):
            data = ctx.alloc(res * res)
            gd = nj.GeneratorData("b", data, res, 0, 0)
            st = {"noise": lambda: nj.NoiseStage(ctx, nj.FractalNoise.Simplex, 0.4, 1.0, 13, 2.0, 0.0, 1700),
                  "gauss": lambda: nj.KernelFilterStage(ctx, nj.KernelFilterType.Gauss5_S1, 18),
                  "flow": lambda: nj.FlowMapStage(ctx, 5, 0.0, 0.005),
                  "erosion": lambda: nj.ErosionStage(ctx, 4)}[a.stage]()
            wi = nj.PipelineWorkItem(gd)
            for _ in range(3):
                st.Schedule(wi, nj.JobHandle())
            ctx.synchronize()
            best = 1e9
            for _ in range(5):
                h0 = ctx.record()
                for _ in range(a.reps):
                    st.Schedule(wi, nj.JobHandle())
                h1 = ctx.record()
                h1.Complete()
                best = min(best, ctx.elapsed_ms(h0, h1) / a.reps)
            print("%-8s res=%d  %.4f ms per stage call, %.4f ms per 4096^2 cells" % (a.stage, res, best,
                                                                                best * 4096 * 4096 / (res * res)))
            st.OnDestroy() if hasattr(st, "OnDestroy") else None
            data.Dispose()


if __name__ == "__main__":
    main()
