"""(round 4; tools/probe_piles.py is round 3's count of piles and increments)  How are the PileSolver events of a live-erosion cycle spread over the pile kernel's blocks?  Runs config 4's cycles up to
the event reduction and counts, per block of side 2 * (PILING_RADIUS + 1) and per colour, the cells whose deltaSediment makes
a pile.  GPU only."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import noize_job_amd as nj  # noqa: E402
from noize_job_amd import _native as N  # noqa: E402

res = 8192
with nj.Context(0) as ctx:
    h = ctx.alloc(res * res)
    gd = nj.GeneratorData("c4", h, res, 0, 0)
    st = nj.NoiseStage(ctx, nj.FractalNoise.Cellular, 0.4, 1.0, 13, 2.0, 0.0, 1700)
    st.Schedule(nj.PipelineWorkItem(gd), nj.JobHandle())
    st.jobHandle.Complete()
    es = nj.ErosionSettings(PARTICLES_PER_CYCLE=10000, CYCLES=1, WATER_STEPS=10)
    tm = nj.tile_set_meta(res, height=1000, tile_size=res, tile_res=res - 16, margin=8)
    G = nj.LiveErosion(ctx, h, tm, es)
    ep = es.AsParameters()
    epp, tmp_ = C.byref(ep), C.byref(tm)
    thr = np.float32(es.PILE_THRESHOLD) / np.float32(tm.HEIGHT)
    B = 2 * (es.PILING_RADIUS + 1)
    nb = (res + B - 1) // B
    for cyc in range(1, 61):
        # one cycle by hand up to the event reduction, a look at the sediment events, then the rest
        ctx.call("nz_thermal_erosion", h.ptr, float(es.TALUS), float(es.THERMAL_STEP), float(tm.TILE_SIZE[0] // tm.HEIGHT),
                 es.THERMAL_CYCLES, res)
        ctx.call("nz_fill_beyer_queue", G.particleQueue._h, epp, tmp_, cyc % 4, res, G.QUEUE_SIZE, cyc, 10)
        ctx.call("nz_queued_beyer_cycle", h.ptr, G.poolMap.ptr, G.streamMap.ptr, G.particleTrack.ptr, G.particleQueue._h,
                 G.events._h, epp, tmp_, G.EVENT_LIMIT, res)
        ctx.call("nz_process_beyer_erosive_events", h.ptr, G.poolMap.ptr, G.streamMap.ptr, G.particleTrack.ptr, G.events._h,
                 epp, tmp_, res).Complete()
        if cyc in (1, 5, 20, 60):
            sed = G.events.sediment()
            pile = (sed != 0) & ~((sed < 0) | (sed <= thr))
            per = pile.reshape(nb, B, nb, B).sum(axis=(1, 3))
            print("cycle %d: %d sediment events, %d piles in %d of %d blocks; piles per block max %d, p99 %d, mean of busy %.2f"
                  % (cyc, int((sed != 0).sum()), int(pile.sum()), int((per > 0).sum()), nb * nb, per.max(),
                     int(np.percentile(per[per > 0], 99)), per[per > 0].mean()))
            for c in range(4):
                sub = per[(c & 1)::2, (c >> 1)::2]
                print("   colour %d: %d busy blocks, max %d" % (c, int((sub > 0).sum()), sub.max()))
        G.particleQueue.Clear()
        ctx.call("nz_erode_height_maps", h.ptr, G.events._h, epp, tmp_, res)
        ctx.call("nz_update_flow_from_track", G.poolMap.ptr, G.streamMap.ptr, G.particleTrack.ptr, ep.FLOW_LOSS_RATE,
                 ep.SURFACE_EVAPORATION_RATE, float(tm.HEIGHT), res)
        ctx.call("nz_pool_automata_job", G.poolMap.ptr, h.ptr, G.particleQueue._h, epp, tmp_, es.WATER_STEPS, res, 1)
    ctx.synchronize()
    G.OnDestroy()
