#!/bin/bash
# wide blurs: 32-row (256 threads) vs 64-row (512 threads) tiles per kernel width
cd "$(dirname "$0")/.."
for big in 99 11; do
  echo "== NZ_WIDE_BIG_FROM=$big"
  NZ_WIDE_BIG_FROM=$big python3 - <<'PY'
import os, sys
sys.path.insert(0, ".")
import noize_job_amd as nj
res = 4096
with nj.Context(0) as ctx:
    data = ctx.alloc(res * res)
    gd = nj.GeneratorData("b", data, res, 0, 0)
    nj.NoiseStage(ctx, nj.FractalNoise.Simplex, 0.4, 1.0, 13, 2.0, 0.0, 1700).Schedule(nj.PipelineWorkItem(gd), nj.JobHandle())
    for width in (11, 13, 15, 17, 19, 21, 23, 25):
        st = nj.StageGaussianBlur(ctx, 2, nj.GaussSigma.s2d00, width)
        wi = nj.PipelineWorkItem(gd)
        for _ in range(3):
            st.Schedule(wi, nj.JobHandle())
        ctx.synchronize()
        best = 1e9
        for _ in range(3):
            h0 = ctx.record()
            for _ in range(10):
                st.Schedule(wi, nj.JobHandle())
            h1 = ctx.record(); h1.Complete()
            best = min(best, ctx.elapsed_ms(h0, h1) / 10)
        print("width %2d x2: %.4f ms" % (width, best))
        st.OnDestroy()
PY
done
