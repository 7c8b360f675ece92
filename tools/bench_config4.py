#!/usr/bin/env python3
"""BASELINE config 4 on one MI355X: 8192^2 cellular fBm (13 octaves) base + live particle erosion
(LiveErosion.TriggerQueuedBeyerMT: thermal -> spawn -> descent -> event reduce -> sediment -> flow from track ->
pool automaton), per-job GPU time from HIP events on the context's stream, whole cycles per second.
usage: bench_config4.py [--res 8192] [--particles 10000] [--cycles 20] [--water-steps 10] [--json out.json]
                        [--at 1,100,1000]   per-job times around these cycle counts of ONE long run (the long-run regime:
                                            relief cut down, a tenth of the cells holding water, pools draining into the
                                            queue), with the share of cells under water and acting in the automaton"""
import argparse
import ctypes as C
import json
import os
import gc
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import noize_job_amd as nj  # noqa: E402

gc.disable()  # a full collection pass of the host (tens of ms with a big heap) must not land in a timed loop


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=8192)
    ap.add_argument("--particles", type=int, default=10000)
    ap.add_argument("--cycles", type=int, default=20)
    ap.add_argument("--water-steps", type=int, default=10)
    ap.add_argument("--json", default=None)
    ap.add_argument("--at", default="", help="comma-separated cycle counts of one long run at which the per-job times are taken")
    ap.add_argument("--skip-fresh", action="store_true", help="only the --at run")
    ap.add_argument("--siblings", choices=("one-call", "two-calls"), default="one-call",
                    help="ErodeHeightMaps and UpdateFlowFromTrackJob: nz_erode_height_maps_and_flow, or the two entries one after the other")
    a = ap.parse_args()
    res = a.res
    out = {"config": "%dx%d cellular-13oct base + live particle erosion, %d particles per cycle, WATER_STEPS %d" %
                     (res, res, a.particles, a.water_steps)}
    with nj.Context(0) as ctx:
        h = ctx.alloc(res * res)
        gd = nj.GeneratorData("c4", h, res, 0, 0)
        st = nj.NoiseStage(ctx, nj.FractalNoise.Cellular, 0.4, 1.0, 13, 2.0, 0.0, 1700)
        for _ in range(2):
            st.Schedule(nj.PipelineWorkItem(gd), nj.JobHandle())
        a0 = ctx.record()
        st.Schedule(nj.PipelineWorkItem(gd), nj.JobHandle())
        a1 = ctx.record()
        a1.Complete()
        out["noise_base_ms"] = round(ctx.elapsed_ms(a0, a1), 4)
        es = nj.ErosionSettings(PARTICLES_PER_CYCLE=a.particles, CYCLES=1, WATER_STEPS=a.water_steps)
        tm = nj.tile_set_meta(res, height=1000, tile_size=res, tile_res=res - 16, margin=8)
        G = nj.LiveErosion(ctx, h, tm, es)
        ep = es.AsParameters()
        epp, tmp_ = C.byref(ep), C.byref(tm)
        jobs = [
            ("thermal", lambda: ctx.call("nz_thermal_erosion", h.ptr, float(es.TALUS), float(es.THERMAL_STEP),
                                         float(tm.TILE_SIZE[0] // tm.HEIGHT), es.THERMAL_CYCLES, res)),
            ("spawn", lambda: ctx.call("nz_fill_beyer_queue", G.particleQueue._h, epp, tmp_, 0, res, a.particles, 7, 10)),
            ("descent", lambda: ctx.call("nz_queued_beyer_cycle", h.ptr, G.poolMap.ptr, G.streamMap.ptr, G.particleTrack.ptr,
                                         G.particleQueue._h, G.events._h, epp, tmp_, 1500, res)),
            ("event_reduce", lambda: ctx.call("nz_process_beyer_erosive_events", h.ptr, G.poolMap.ptr, G.streamMap.ptr,
                                              G.particleTrack.ptr, G.events._h, epp, tmp_, res)),
            ("clear_queue", lambda: G.particleQueue.Clear()),
            ("erode_height_maps", lambda: ctx.call("nz_erode_height_maps", h.ptr, G.events._h, epp, tmp_, res)),
            ("flow_from_track", lambda: ctx.call("nz_update_flow_from_track", G.poolMap.ptr, G.streamMap.ptr,
                                                 G.particleTrack.ptr, ep.FLOW_LOSS_RATE, ep.SURFACE_EVAPORATION_RATE,
                                                 float(tm.HEIGHT), res)),
            ("pool_automata", lambda: ctx.call("nz_pool_automata_job", G.poolMap.ptr, h.ptr, G.particleQueue._h, epp, tmp_,
                                               a.water_steps, res, 1)),
        ]
        if a.siblings == "one-call":
            i = [n for n, _ in jobs].index("erode_height_maps")
            jobs[i:i + 2] = [("erode_height_maps+flow_from_track",
                              lambda: ctx.call("nz_erode_height_maps_and_flow", h.ptr, G.events._h, G.poolMap.ptr, G.streamMap.ptr,
                                               G.particleTrack.ptr, epp, tmp_, res))]
        G.fuseSiblings = a.siblings == "one-call"
        out["siblings"] = a.siblings
        if a.at:
            # one long run; around every checkpoint the jobs of 10 cycles are bracketed with stream markers
            import numpy as np
            checkpoints = sorted(int(x) for x in a.at.split(",") if x)
            out["long_run"] = {}
            done = 0
            seed = 1

            def one_cycle(timed):
                nonlocal seed
                seed += 1
                jobs[1] = ("spawn", lambda: ctx.call("nz_fill_beyer_queue", G.particleQueue._h, epp, tmp_, (seed // 30) % 4, res,
                                                     a.particles, seed, 10))
                marks = [ctx.record()] if timed else None
                for _, fn in jobs:
                    fn()
                    if timed:
                        marks.append(ctx.record())
                return marks
            for cp in checkpoints:
                while done < cp - 5:
                    one_cycle(False)
                    done += 1
                    if done % 64 == 0:
                        ctx.synchronize()
                acc_cp = {n: 0.0 for n, _ in jobs}
                ev = 0
                for _ in range(10):
                    marks = one_cycle(True)
                    marks[-1].Complete()
                    for i, (n, _) in enumerate(jobs):
                        acc_cp[n] += ctx.elapsed_ms(marks[i], marks[i + 1])
                    ev += G.events.Count
                    done += 1
                pool = G.poolMap.ToArray()
                out["long_run"]["cycle_%d" % cp] = {
                    "per_job_ms": {n: round(v / 10, 4) for n, v in acc_cp.items()}, "cycle_ms": round(sum(acc_cp.values()) / 10, 4),
                    "events_per_cycle": ev // 10, "cells_holding_water": round(float((pool > 0).mean()), 5),
                    "cells_acting_in_the_automaton": round(float((pool >= 1e-3).mean()), 6),
                    "queue_before_spawn": int(G.particleQueue.Count)}
            if a.skip_fresh:
                print(json.dumps(out, indent=1))
                if a.json:
                    with open(a.json, "w") as f:
                        json.dump(out, f, indent=1)
                G.OnDestroy()
                return
        acc = {n: 0.0 for n, _ in jobs}
        events = 0
        for cyc in range(a.cycles + 2):
            marks = [ctx.record()]
            for _, fn in jobs:
                fn()
                marks.append(ctx.record())
            marks[-1].Complete()
            if cyc >= 2:
                for i, (n, _) in enumerate(jobs):
                    acc[n] += ctx.elapsed_ms(marks[i], marks[i + 1])
                events += G.events.Count
        out["per_job_ms"] = {n: round(v / a.cycles, 4) for n, v in acc.items()}
        out["cycle_ms"] = round(sum(acc.values()) / a.cycles, 4)
        out["events_per_cycle"] = events // a.cycles
        # the automaton alone on a plane under water everywhere (every walk is one run as long as its row: the run form's
        # worst case) and on a plane with a wet cell in fifty
        import numpy as np
        for name, plane in (("pool_automata_all_wet_ms", np.full((res, res), 0.01, np.float32)),
                            ("pool_automata_2pct_wet_ms",
                             np.where(np.random.default_rng(5).random((res, res)) < 0.02, 0.01, 0).astype(np.float32))):
            ms = []
            for _ in range(3):
                wet = ctx.from_host(plane)
                m0 = ctx.record()
                ctx.call("nz_pool_automata", wet.ptr, h.ptr, a.water_steps, res)
                m1 = ctx.record()
                m1.Complete()
                ms.append(ctx.elapsed_ms(m0, m1))
            out[name] = round(min(ms), 4)
        out["pool_runs"] = os.environ.get("NZ_POOL_RUNS", "1")
        # whole Updates through the host driver, wall clock
        es.CYCLES = 3
        n_up = max(30, a.cycles // 3)

        def updates():
            G.TriggerQueuedBeyerMT([1, 2, 3]).Complete()
            t0 = time.perf_counter()
            for u in range(n_up):
                G.TriggerQueuedBeyerMT([10 * u + 1, 10 * u + 2, 10 * u + 3])
            G.jobHandle.Complete()
            return time.perf_counter() - t0
        G.fewHandles = False   # a JobHandle out of every job, as the reference schedules them
        out["driver_cycle_ms_one_handle_per_job"] = round(updates() / (3 * n_up) * 1e3, 4)
        G.fewHandles = True    # the default: only the handles somebody waits for
        dt = updates()
        out["driver_cycles_per_s"] = round(3 * n_up / dt, 2)
        out["driver_cycle_ms"] = round(dt / (3 * n_up) * 1e3, 4)
        out["driver_particle_steps_per_s"] = round(out["events_per_cycle"] * 3 * n_up / dt)
        G.OnDestroy()
    print(json.dumps(out, indent=1))
    if a.json:
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
