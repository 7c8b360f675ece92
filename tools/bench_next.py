#!/usr/bin/env python3
"""Times the stages outside the metric pipeline (SURVEY.md 8a rows not in the metric, 8f "next" rows)
on one device-resident tile: every noise basis, every KernelFilterType, wide Gaussian / box blurs,
value erosion, the element-wise stages, thermal erosion and the mesh.  Back-to-back launches, HIP events.
usage: bench_next.py [--res 4096] [--reps 10] [--json out.json] [--float-mode strict|fast|relaxed] [--only SUBSTRING]
(tools/collect_next.sh runs it under rocprofv3 for the per-kernel counters, tools/next_counters_table.py prints them)"""
import argparse
import json
import os
import gc
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import noize_job_amd as nj  # noqa: E402

gc.disable()  # a full collection pass of the host (tens of ms with a big heap) must not land in a timed loop

HBM = 8000.0  # GB/s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--json", default=None)
    ap.add_argument("--float-mode", choices=("strict", "fast", "relaxed"), default="strict")
    ap.add_argument("--only", default="", help="only the rows whose name contains this")
    a = ap.parse_args()
    res, cells = a.res, a.res * a.res
    rows = []
    with nj.Context(0) as ctx:
        ctx.float_mode = {"strict": 0, "fast": 1, "relaxed": 2}[a.float_mode]
        data, right = ctx.alloc(cells), ctx.alloc(cells)
        gd = nj.GeneratorData("b", data, res, 0, 0)
        rd = nj.ReduceData("b", data, right, res, 0, 0)
        md = nj.MeshStageData("b", data, res - 8, res, 4, 1000.0, 1000.0)
        seed = nj.NoiseStage(ctx, nj.FractalNoise.Simplex, 0.4, 1.0, 13, 2.0, 0.0, 1700)

        def run(name, stage, item, bytes_per_cell, note=""):
            if a.only and a.only not in name:
                stage.OnDestroy()
                return
            wi = nj.PipelineWorkItem(item)
            for _ in range(2):
                stage.Schedule(wi, nj.JobHandle())
            ctx.synchronize()
            best = 1e9
            for _ in range(3):
                h0 = ctx.record()
                for _ in range(a.reps):
                    stage.Schedule(wi, nj.JobHandle())
                h1 = ctx.record()
                h1.Complete()
                best = min(best, ctx.elapsed_ms(h0, h1) / a.reps)
            gbs = bytes_per_cell * cells / (best * 1e-3) / 1e9
            rows.append({"stage": name, "ms": round(best, 4), "Mcells/s": round(cells / best / 1e3),
                         "algorithmic_B_per_cell": bytes_per_cell, "algorithmic_GB/s": round(gbs, 1),
                         "frac_hbm": round(gbs / HBM, 3), "note": note})
            print("%-44s %8.4f ms %9.0f Mcells/s  %6.0f B/cell  %7.0f GB/s (%5.1f %% of 8 TB/s) %s" % (
                name, best, cells / best / 1e3, bytes_per_cell, gbs, 100 * gbs / HBM, note))
            stage.OnDestroy()

        seed.Schedule(nj.PipelineWorkItem(gd), nj.JobHandle())
        right.CopyFrom(data.ToArray())
        for basis in nj.FractalNoise:
            run("noise %s x13 oct" % basis.name, nj.NoiseStage(ctx, basis, 0.4, 1.0, 13, 2.0, 0.0, 1700), gd, 4,
                "fp32-VALU bound")
        seed.Schedule(nj.PipelineWorkItem(gd), nj.JobHandle())
        for f in nj.KernelFilterType:
            if f == nj.KernelFilterType.Sobel3_2D:
                continue  # unsupported in the reference too (README marks it broken)
            run("filter %s x6" % f.name, nj.KernelFilterStage(ctx, f, 6), gd, 8 * 6)
            seed.Schedule(nj.PipelineWorkItem(gd), nj.JobHandle())
        for f in (nj.KernelFilterType.Gauss5_S1, nj.KernelFilterType.Gauss9_S1, nj.KernelFilterType.Smooth3):
            run("filter %s x1 (the delegate's single application)" % f.name, nj.KernelFilterStage(ctx, f, 1), gd, 8)
        seed.Schedule(nj.PipelineWorkItem(gd), nj.JobHandle())
        for width in (5, 9, 13, 25):
            run("gaussian blur width %d sigma 2.0 x2" % width,
                nj.StageGaussianBlur(ctx, 2, nj.GaussSigma.s2d00, width), gd, 8 * 2)
        run("box blur width 25 x2", nj.StageSmoothBlur(ctx, 2, 25), gd, 8 * 2)
        seed.Schedule(nj.PipelineWorkItem(gd), nj.JobHandle())
        run("value erosion x1", nj.ErosionStage(ctx, 1), gd, 8)
        run("value erosion x8", nj.ErosionStage(ctx, 8), gd, 8 * 8)
        seed.Schedule(nj.PipelineWorkItem(gd), nj.JobHandle())
        run("flow map x1", nj.FlowMapStage(ctx, 1, 0.0, 0.005), gd, 24 + 20)
        seed.Schedule(nj.PipelineWorkItem(gd), nj.JobHandle())
        run("flow map x12", nj.FlowMapStage(ctx, 12, 0.0, 0.005), gd, 24 + 44 * 11 + 20)
        # the same stage bodies on a READ / WRITE plane pair (nz_*_rw: SWAP_RWTILE is a pointer swap); the pair is its
        # own two planes, since the stages leave the result in either
        seed.Schedule(nj.PipelineWorkItem(gd), nj.JobHandle())
        pd = nj.GeneratorData("pair", ctx.alloc(cells), res, 0, 0, write=ctx.alloc(cells))
        pd.data.CopyFrom(data.ToArray())
        pair = "READ/WRITE pair: no flush copy"
        run("pair: filter Gauss5_S1 x1", nj.KernelFilterStage(ctx, nj.KernelFilterType.Gauss5_S1, 1), pd, 8, pair)
        run("pair: filter Gauss5_S1 x17", nj.KernelFilterStage(ctx, nj.KernelFilterType.Gauss5_S1, 17), pd, 8 * 17, pair)
        run("pair: gaussian blur width 25 sigma 2.0 x1", nj.StageGaussianBlur(ctx, 1, nj.GaussSigma.s2d00, 25), pd, 8, pair)
        run("pair: value erosion x1", nj.ErosionStage(ctx, 1), pd, 8, pair)
        run("pair: value erosion x5", nj.ErosionStage(ctx, 5), pd, 8 * 5, pair)
        seed.Schedule(nj.PipelineWorkItem(pd), nj.JobHandle())
        run("pair: flow map x5", nj.FlowMapStage(ctx, 5, 0.0, 0.005), pd, 24 + 44 * 4 + 20, pair)
        pd.data.Dispose()
        pd.write.Dispose()
        seed.Schedule(nj.PipelineWorkItem(gd), nj.JobHandle())
        run("constant MULTIPLY", nj.ConstantStage(ctx, nj.ConstantOperationType.MULTIPLY, 0.999), gd, 8)
        run("constant BINARIZE", nj.ConstantStage(ctx, nj.ConstantOperationType.BINARIZE, 0.5), gd, 8)
        seed.Schedule(nj.PipelineWorkItem(gd), nj.JobHandle())
        for op in nj.ReductionType:
            run("reduce %s" % op.name, nj.ReduceStage(ctx, op), rd, 12)
            seed.Schedule(nj.PipelineWorkItem(gd), nj.JobHandle())
        run("curve LUT 256", nj.CurveStage(ctx, lambda t: 1.0 - t, 256), gd, 8)
        seed.Schedule(nj.PipelineWorkItem(gd), nj.JobHandle())
        run("thermal erosion x1 (4 phases)", nj.StageThermalErosion(ctx, 1), gd, 8 * 4,
            "in place; the four colour phases as two passes over the plane")
        seed.Schedule(nj.PipelineWorkItem(gd), nj.JobHandle())
        vb = 4 + (48.0 * (res - 7) ** 2 + 24.0 * (res - 8) ** 2) / cells
        run("mesh Overshoot %d^2" % (res - 8), nj.MeshTileStage(ctx, nj.MeshType.OvershootSquareGridHeightMap), md, vb,
            "write-only streams")
        run("mesh SquareGrid %d^2" % (res - 8), nj.MeshTileStage(ctx, nj.MeshType.SquareGridHeightMap), md, vb)
        # live erosion grid jobs (not PipelineStages: timed through the C ABI)
        import numpy as np
        rng = np.random.default_rng(0)
        pool = ctx.from_host(np.where(rng.random((res, res)) < 0.3, rng.random((res, res), dtype=np.float32) * 0.3, 0).astype(np.float32))
        flow, track = ctx.alloc(cells), ctx.alloc(cells)
        rng3 = ctx.alloc(3)
        for name, fn, bpc, note in (
                ("map range (min, max, range of a plane)", lambda: ctx.call("nz_get_map_range", data.ptr, cells, rng3.ptr, float("inf"), float("-inf")), 4,
                 "GetMapRangeJob: one read of the plane, result stays in device memory"),
                ("normalise with device args", lambda: ctx.call("nz_map_normalize_values_dev", data.ptr, data.ptr, rng3.ptr, res), 8, ""),
                ("live erosion: flow from track", lambda: ctx.call("nz_update_flow_from_track", pool.ptr, flow.ptr, track.ptr, 0.05, 0.1, 1000.0, res), 24, ""),
                ("live erosion: pool automaton x1 (4 colour passes)", lambda: ctx.call("nz_pool_automata", pool.ptr, data.ptr, 1, res), 4 * 8,
                 "parallel runs of acting steps; 30 % of the cells under water")):
            if a.only and a.only not in name:
                continue
            fn()
            ctx.synchronize()
            h0 = ctx.record()
            for _ in range(3):
                fn()
            h1 = ctx.record()
            h1.Complete()
            ms = ctx.elapsed_ms(h0, h1) / 3
            gbs = bpc * cells / (ms * 1e-3) / 1e9
            rows.append({"stage": name, "ms": round(ms, 4), "Mcells/s": round(cells / ms / 1e3), "algorithmic_B_per_cell": bpc,
                         "algorithmic_GB/s": round(gbs, 1), "frac_hbm": round(gbs / HBM, 3), "note": note})
            print("%-44s %8.4f ms %9.0f Mcells/s  %6.0f B/cell  %7.0f GB/s (%5.1f %% of 8 TB/s) %s" % (
                name, ms, cells / ms / 1e3, bpc, gbs, 100 * gbs / HBM, note))
    if a.json:
        json.dump({"res": res, "rows": rows}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
