#!/usr/bin/env python3
"""Times one stage of the metric pipeline in isolation (HIP events on the context's stream).
usage: bench_stage.py {noise|gauss|flow|erosion|mesh|all} [--res 4096] [--reps 20] [--pair]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import noize_job_amd as nj  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("stage")
    ap.add_argument("--res", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--basis", type=int, default=3)
    ap.add_argument("--octaves", type=int, default=13)
    ap.add_argument("--gauss", type=int, default=17)
    ap.add_argument("--flow", type=int, default=5)
    ap.add_argument("--erosion", type=int, default=5)
    ap.add_argument("--pair", action="store_true", help="a READ / WRITE plane pair, as the tile pipelines use (no flush copies, free launch counts)")
    a = ap.parse_args()
    res = a.res
    with nj.Context(0) as ctx:
        data = ctx.alloc(res * res)
        gd = nj.GeneratorData("b", data, res, 0, 0, write=ctx.alloc(res * res) if a.pair else None)
        stages = {"noise": nj.NoiseStage(ctx, nj.FractalNoise(a.basis), 0.4, 1.0, a.octaves, 2.0, 0.0, 1700),
                  "gauss": nj.KernelFilterStage(ctx, nj.KernelFilterType.Gauss5_S1, a.gauss),
                  "flow": nj.FlowMapStage(ctx, a.flow, 0.0, 0.005),
                  "erosion": nj.ErosionStage(ctx, a.erosion)}
        if a.stage == "mesh":  # BASELINE config 3: Overshoot mesh of resolution res-8 over the res^2 tile
            stages["mesh"] = nj.MeshTileStage(ctx, nj.MeshType.OvershootSquareGridHeightMap)
            mesh_data = nj.MeshStageData("b", data, res - 8, res, 4, 1000.0, 1000.0)
        names = [n for n in stages if n != "mesh"] if a.stage == "all" else [a.stage]
        stages["noise"].Schedule(nj.PipelineWorkItem(gd), nj.JobHandle())  # something sensible to filter
        for n in names:
            st = stages[n]
            wi = nj.PipelineWorkItem(mesh_data if n == "mesh" else gd)
            for _ in range(3):
                st.Schedule(wi, nj.JobHandle())
            ts = []
            for _ in range(a.reps):
                h0 = ctx.record()
                st.Schedule(wi, h0)
                h1 = st.jobHandle
                h1.Complete()
                ts.append(ctx.elapsed_ms(h0, h1))
            ts = np.array(ts)
            h0 = ctx.record()                       # back to back: the clocks stay up, no idle gaps
            for _ in range(a.reps):
                st.Schedule(wi, nj.JobHandle())
            h1 = ctx.record()
            h1.Complete()
            b2b = ctx.elapsed_ms(h0, h1) / a.reps
            print("%-8s res=%d median %.4f ms  min %.4f  back-to-back %.4f  (%.0f Mcells/s)" % (
                n, res, np.median(ts), ts.min(), b2b, res * res / b2b / 1e3))


if __name__ == "__main__":
    main()
