#!/usr/bin/env python3
"""The README's quick start as a runnable script: one 4096^2 tile, then 64 tiles of 512^2 as one batch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import noize_job_amd as nj  # noqa: E402

ctx = nj.Context(0)
tile = nj.GeneratorData("t", ctx.alloc(4096 * 4096), 4096, xpos=0, zpos=0,
                        write=ctx.alloc(4096 * 4096))  # optional WRITE plane: stages swap the pair instead of flushing
pipe = nj.BasePipeline([nj.NoiseStage(ctx, nj.FractalNoise.Simplex, 0.4, 1.0, 13, 2.0, 0.0, 1700),
                        nj.KernelFilterStage(ctx, nj.KernelFilterType.Gauss5_S1, 17),
                        nj.FlowMapStage(ctx, 5, 0.0, 0.005), nj.ErosionStage(ctx, 5)])
pipe.Enqueue(tile, completeAction=lambda d: print("done", d.uuid))
pipe.RunToCompletion()
heights = tile.data.ToArray((4096, 4096))
print("tile", heights.shape, "mean %.6f" % float(heights.mean()))
batch = nj.GeneratorDataBatch.create(ctx, "b", 512, [(512 * k, 0) for k in range(64)])
pipe.Enqueue(batch, completeAction=lambda d: print("done", d.uuid, "x", d.count))
pipe.RunToCompletion()
print("batch tile 63 mean %.6f" % float(batch.tile(63).ToArray((512, 512)).mean()))
pipe.Destroy()
ctx.close()
