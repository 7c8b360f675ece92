#!/usr/bin/env python3
"""How the PileSolver events of a BASELINE config 4 cycle fall into pile_kernel's blocks (around cycle 100): events,
dispersed / piled, blocks holding a pile per colour, piles in the fullest block -- a block's piles run one after the other."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import noize_job_amd as nj  # noqa: E402

res = 8192
with nj.Context(0) as ctx:
    h = ctx.alloc(res * res)
    gd = nj.GeneratorData("c4", h, res, 0, 0)
    nj.NoiseStage(ctx, nj.FractalNoise.Cellular, 0.4, 1.0, 13, 2.0, 0.0, 1700).Schedule(nj.PipelineWorkItem(gd), nj.JobHandle())
    es = nj.ErosionSettings(PARTICLES_PER_CYCLE=10000, CYCLES=1, WATER_STEPS=10)
    tm = nj.tile_set_meta(res, height=1000, tile_size=res, tile_res=res - 16, margin=8)
    G = nj.LiveErosion(ctx, h, tm, es)
    for c in range(100):
        G.TriggerQueuedBeyerMT([c + 1]).Complete()
    ep = es.AsParameters()
    epp, tmp_ = C.byref(ep), C.byref(tm)
    ctx.call("nz_fill_beyer_queue", G.particleQueue._h, epp, tmp_, 0, res, 10000, 777, 10)
    ctx.call("nz_queued_beyer_cycle", h.ptr, G.poolMap.ptr, G.streamMap.ptr, G.particleTrack.ptr, G.particleQueue._h, G.events._h,
             epp, tmp_, 1500, res)
    ctx.call("nz_process_beyer_erosive_events", h.ptr, G.poolMap.ptr, G.streamMap.ptr, G.particleTrack.ptr, G.events._h, epp, tmp_, res)
    ctx.synchronize()
    sed = G.events.sediment()
    thr = es.PILE_THRESHOLD / tm.HEIGHT
    ev = sed != 0
    disp = ev & ((sed < 0) | (sed <= thr))
    pile = ev & ~disp
    B = 2 * (es.PILING_RADIUS + 1)
    nb = (res + B - 1) // B
    per = pile.reshape(nb, B, nb, B).sum(axis=(1, 3))
    print("cells with a sediment event %d: dispersed %d, piled %d; pile amounts: median %.3g max %.3g (increment %.3g)" % (
        ev.sum(), disp.sum(), pile.sum(), np.median(sed[pile]) if pile.any() else 0, sed[pile].max() if pile.any() else 0,
        es.MIN_PILE_INCREMENT / tm.HEIGHT))
    inc = es.MIN_PILE_INCREMENT / tm.HEIGHT
    steps = (np.where(pile, np.ceil(sed / inc), 0)).reshape(nb, B, nb, B).sum(axis=(1, 3))
    print("increments to place: %d in all; per block: median %d, 99 %% %d, max %d (that block holds %d piles)" % (
        steps.sum(), np.median(steps[per > 0]), np.percentile(steps[per > 0], 99), steps.max(), per.flat[steps.argmax()]))
    for colour in range(4):
        cx, cz = colour & 1, colour >> 1
        sub = per[cx::2, cz::2]
        print("  colour %d: %5d of %5d blocks hold a pile, fullest block %d piles, mean %.2f" % (
            colour, (sub > 0).sum(), sub.size, sub.max(), sub[sub > 0].mean() if (sub > 0).any() else 0))
    G.OnDestroy()
