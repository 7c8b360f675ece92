#!/usr/bin/env python3
"""Throughput of many independent tiles (the reference's use: one BasePipeline per tile request,
Scripts/MeshTileGenerator.cs:181-211): N tiles of res^2 cells, round-robin over S contexts (HIP streams).
With --batch B the tiles go through the batched stage bodies instead (nz_*_batch): B tiles per launch sequence.
usage: bench_tiles.py [--res 512] [--streams 1 2 4] [--tiles 256] [--batch 4 16 64]"""
import argparse
import os
import gc
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import noize_job_amd as nj  # noqa: E402

gc.disable()  # a full collection pass of the host (tens of ms with a big heap) must not land in a timed loop


def make(ctx, res):
    stages = [nj.NoiseStage(ctx, nj.FractalNoise.Simplex, 0.4, 1.0, 13, 2.0, 0.0, 1700),
              nj.KernelFilterStage(ctx, nj.KernelFilterType.Gauss5_S1, 17), nj.FlowMapStage(ctx, 5, 0.0, 0.005),
              nj.ErosionStage(ctx, 5)]
    pipe = nj.BasePipeline(stages, "tile")
    gd = nj.GeneratorData("t", ctx.alloc(res * res), res, 0, 0, write=ctx.alloc(res * res))  # READ / WRITE pair
    return pipe, gd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, nargs="+", default=[512, 1024, 2048])
    ap.add_argument("--streams", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--tiles", type=int, default=256)
    ap.add_argument("--batch", type=int, nargs="*", default=[4, 16, 64])
    a = ap.parse_args()
    for res in a.res:
        for B in a.batch:
            with nj.Context(0) as ctx:
                stages = [nj.NoiseStage(ctx, nj.FractalNoise.Simplex, 0.4, 1.0, 13, 2.0, 0.0, 1700),
                          nj.KernelFilterStage(ctx, nj.KernelFilterType.Gauss5_S1, 17),
                          nj.FlowMapStage(ctx, 5, 0.0, 0.005), nj.ErosionStage(ctx, 5)]
                pipe = nj.BasePipeline(stages, "batch")
                batch = nj.GeneratorDataBatch.create(ctx, "b", res, [(res * k, 0) for k in range(B)])
                batch.write = ctx.alloc(B * res * res)  # READ / WRITE pair: the stages swap instead of flushing
                for _ in range(3):
                    pipe.Schedule(batch)
                    pipe.pipelineRunning = False
                ctx.synchronize()
                n = max(1, a.tiles // B)
                t0 = time.perf_counter()
                for _ in range(n):
                    pipe.Schedule(batch)
                    pipe.pipelineRunning = False
                ctx.synchronize()
                dt = time.perf_counter() - t0
                print("res %5d  batch %3d  : %8.1f tiles/s  %9.0f Mcells/s  (%.3f ms per tile)" % (
                    res, B, n * B / dt, n * B * res * res / dt / 1e6, dt / (n * B) * 1e3))
                pipe.Destroy()
                batch.data.Dispose()
                batch.write.Dispose()
                batch.positions.Dispose()
        for S in a.streams:
            ctxs = [nj.Context(0) for _ in range(S)]
            pipes = [make(c, res) for c in ctxs]
            for _ in range(3):
                for i, (p, gd) in enumerate(pipes):
                    gd.xpos = 7 * i
                    p.Schedule(gd)
                    p.pipelineRunning = False
            for c in ctxs:
                c.synchronize()
            t0 = time.perf_counter()
            for k in range(a.tiles):
                p, gd = pipes[k % S]
                gd.xpos = res * k            # a different tile of the world each time
                p.Schedule(gd)
                p.pipelineRunning = False
            t_host = time.perf_counter() - t0
            for c in ctxs:
                c.synchronize()
            dt = time.perf_counter() - t0
            print("res %5d  streams %d: %8.1f tiles/s  %9.0f Mcells/s  (%.3f ms per tile, host enqueue %.3f ms per tile)" % (
                res, S, a.tiles / dt, a.tiles * res * res / dt / 1e6, dt / a.tiles * 1e3, t_host / a.tiles * 1e3))
            for p, gd in pipes:
                p.Destroy()
                gd.data.Dispose()
                gd.write.Dispose()
            for c in ctxs:
                c.close()


if __name__ == "__main__":
    main()
