#!/usr/bin/env python3
"""Random live-erosion set-ups against the oracle, job by job, bit for bit: resolutions, particle counts, ages, pile
radii / thresholds / increments, tile heights, patch sizes, planes with standing water and flow, worker counts.
usage: soak_live.py [--minutes 3] [--seed 0]      (prints one line per case that differs; exit code 1 if any)"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # tests/ is where the oracle may be used
sys.path.insert(0, ROOT)
import noize_job_amd as nj  # noqa: E402
import oracle  # noqa: E402  (the checker: tools and tests only)

f32 = np.float32


def one_case(rng, nj, oracle, G, L, ep, epp, tmp_, res, particles, th, workers, shape, stats):
    why = None
    for cyc in range(2):
        seed = int(rng.integers(1, 2 ** 31 - 1))
        G.ctx.call("nz_fill_beyer_queue", G.particleQueue._h, epp, tmp_, cyc, res, particles, seed, workers)
        L.fill_queue(cyc, particles, seed, workers)
        G.ctx.call("nz_queued_beyer_cycle", G.heightMap.ptr, G.poolMap.ptr, G.streamMap.ptr, G.particleTrack.ptr,
                   G.particleQueue._h, G.events._h, epp, tmp_, 1500, res)
        n = L.descend()
        if G.events.Count != n:
            why = "event count"
            break
        G.ctx.call("nz_process_beyer_erosive_events", G.heightMap.ptr, G.poolMap.ptr, G.streamMap.ptr, G.particleTrack.ptr,
                   G.events._h, epp, tmp_, res)
        L.process_events()
        stats[0] += n
        thr = f32(ep.PILE_THRESHOLD) / f32(th)
        stats[1] += int((L.sediment > thr).sum())
        stats[2] += int(((L.sediment != 0) & ~(L.sediment > thr)).sum())
        if not (np.array_equal(G.events.sediment(), L.sediment) and np.array_equal(G.poolMap.ToArray(shape), L.pool)
                and np.array_equal(G.particleTrack.ToArray(shape), L.track)):
            why = "events"
            break
        G.particleQueue.Clear()
        one_call = bool(rng.integers(0, 2))  # the two siblings as one launch (nz_erode_height_maps_and_flow) or as two entries
        if one_call:
            G.ctx.call("nz_erode_height_maps_and_flow", G.heightMap.ptr, G.events._h, G.poolMap.ptr, G.streamMap.ptr,
                       G.particleTrack.ptr, epp, tmp_, res)
        else:
            G.ctx.call("nz_erode_height_maps", G.heightMap.ptr, G.events._h, epp, tmp_, res)
        L.erode_height_maps()
        if not np.array_equal(G.heightMap.ToArray(shape), L.height):
            why = "sediment (disperse / piles)" + (", one call" if one_call else "")
            break
        if not one_call:
            G.ctx.call("nz_update_flow_from_track", G.poolMap.ptr, G.streamMap.ptr, G.particleTrack.ptr, ep.FLOW_LOSS_RATE,
                       ep.SURFACE_EVAPORATION_RATE, float(th), res)
        L.update_flow_from_track()
        G.ctx.call("nz_pool_automata_job", G.poolMap.ptr, G.heightMap.ptr, G.particleQueue._h, epp, tmp_, 3, res, 1)
        L.pool_automata(3, drain=True)
        stats[3] += int(L.count.value)
        if not (np.array_equal(G.poolMap.ToArray(shape), L.pool) and np.array_equal(G.streamMap.ToArray(shape), L.flow)
                and np.array_equal(G.particleTrack.ToArray(shape), L.track)):
            why = "flow from track / automaton"
            break
    return why


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=3.0)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    oracle.lib()
    t_end = time.time() + 60 * a.minutes
    cases = bad = 0
    stats = [0, 0, 0, 0]   # particle steps, cells piled, dispersed, particles drained
    with nj.Context(0) as ctx:
        while time.time() < t_end:
            res = int(rng.integers(40, 420))
            particles = int(rng.integers(50, 6000))
            th = int(rng.choice([100, 500, 1000, 3000]))
            patch = float(rng.choice([0.5, 1.0, 2.5, 4.0]))
            workers = int(rng.choice([1, 3, 10, 64]))
            es = nj.ErosionSettings(
                PARTICLES_PER_CYCLE=particles, MAXAGE=int(rng.integers(3, 160)), PILING_RADIUS=int(rng.integers(1, 21)),
                PILE_THRESHOLD=float(rng.choice([0.05, 0.4, 2.0, 6.0])), MIN_PILE_INCREMENT=float(rng.choice([0.05, 0.25, 1.0, 3.0])),
                INERTIA=float(rng.uniform(0.0, 0.95)), GRAVITY=float(rng.uniform(0.5, 20.0)), FRICTION=float(rng.uniform(0.0, 3.0)),
                DRAG=float(rng.uniform(0.0005, 0.3)), EVAP=float(rng.uniform(0.0, 0.2)), CAPACITY=float(rng.uniform(0.05, 8.0)),
                EROSION=float(rng.uniform(0.0, 0.9)), DEPOSITION=float(rng.uniform(0.0, 0.9)),
                FLOW_HEIGHT_CONTRIBUTION=float(rng.uniform(0.0, 30.0)))
            octaves = int(rng.integers(1, 9))
            h = oracle.kernel_filter(oracle.fractal(int(rng.integers(0, 2)) * 3, res, res, 0.4, 1.0, 2.0, 0.0, octaves, 0, 0,
                                                    int(rng.integers(40, 600))), 2, int(rng.integers(0, 4)))
            if rng.random() < 0.3:  # terraces: plateaus and equal neighbours
                h = (np.round(h * f32(rng.integers(5, 60))) / f32(rng.integers(5, 60))).astype(f32)
            h = np.clip(h, 0, 1).astype(f32)
            tm = nj.tile_set_meta(res, height=th, tile_size=res, tile_res=res, patch_res=patch)
            G = nj.LiveErosion(ctx, ctx.from_host(h), tm, es, queueCapacity=1 << 17)
            ep = es.AsParameters()
            L = oracle.LiveErosionOracle(h, oracle.erosion_params(**{n: getattr(ep, n) for n, _ in ep._fields_}), tile_height=th,
                                         patch_res=patch, capacity=1 << 17)
            pool0 = np.where(rng.random((res, res)) < rng.uniform(0, 0.1), rng.random((res, res), dtype=f32) * f32(0.01), 0).astype(f32)
            flow0 = (rng.random((res, res), dtype=f32) * f32(rng.uniform(0, 0.9))).astype(f32)
            G.poolMap.CopyFrom(pool0); G.streamMap.CopyFrom(flow0)
            L.pool[:] = pool0; L.flow[:] = flow0
            epp, tmp_ = C.byref(ep), C.byref(tm)
            shape = (res, res)
            why = None
            try:
                why = one_case(rng, nj, oracle, G, L, ep, epp, tmp_, res, particles, th, workers, shape, stats)
            except nj.NoizeError as e:   # (an error out of the library: say which set-up, then let it end the run)
                print("ERROR %s: case %d res %d particles %d height %d patch %g workers %d %s" % (
                    e, cases, res, particles, th, patch, workers, {n: getattr(ep, n) for n, _ in ep._fields_}), flush=True)
                raise
            cases += 1
            if why:
                bad += 1
                print("DIFFERS (%s): res %d particles %d height %d patch %g workers %d %s" % (
                    why, res, particles, th, patch, workers, {n: getattr(ep, n) for n, _ in ep._fields_}), flush=True)
            if cases % 200 == 0:
                print("%d cases, %d differ (%d particle steps, %d cells piled, %d dispersed, %d particles drained from pools)" % (
                    cases, bad, *stats), flush=True)
            G.OnDestroy()
    print("%d cases, %d differ (%d particle steps, %d cells piled, %d dispersed, %d particles drained from pools)" % (
        cases, bad, *stats))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
