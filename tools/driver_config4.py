"""BASELINE config 4's driver loop alone (for a kernel timeline: tools/trace_config4.sh): an 8192^2 cellular base, then
`--updates` Updates of three cycles each through LiveErosion.TriggerQueuedBeyerMT."""
import argparse
import os
import gc
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import noize_job_amd as nj  # noqa: E402

gc.disable()  # a full collection pass of the host (tens of ms with a big heap) must not land in a timed loop

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=8192)
ap.add_argument("--updates", type=int, default=20)
ap.add_argument("--particles", type=int, default=10000)
ap.add_argument("--one-handle-per-job", action="store_true")
a = ap.parse_args()
with nj.Context(0) as ctx:
    h = ctx.alloc(a.res * a.res)
    gd = nj.GeneratorData("c4", h, a.res, 0, 0)
    st = nj.NoiseStage(ctx, nj.FractalNoise.Cellular, 0.4, 1.0, 13, 2.0, 0.0, 1700)
    st.Schedule(nj.PipelineWorkItem(gd), nj.JobHandle())
    st.jobHandle.Complete()
    es = nj.ErosionSettings(PARTICLES_PER_CYCLE=a.particles, CYCLES=3, WATER_STEPS=10)
    tm = nj.tile_set_meta(a.res, height=1000, tile_size=a.res, tile_res=a.res - 16, margin=8)
    G = nj.LiveErosion(ctx, h, tm, es)
    G.fewHandles = not a.one_handle_per_job
    G.TriggerQueuedBeyerMT([1, 2, 3]).Complete()
    t0 = time.perf_counter()
    for u in range(a.updates):
        G.TriggerQueuedBeyerMT([10 * u + 1, 10 * u + 2, 10 * u + 3])
    G.jobHandle.Complete()
    print("%.4f ms per cycle" % ((time.perf_counter() - t0) / (3 * a.updates) * 1e3))
    G.OnDestroy()
