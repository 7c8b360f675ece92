"""Fast-forwards tests/soak_live.py --seed 707 to case 7457 (the draws only) and runs that case with timing."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # tests/repro/ -> the repository (under tests/: these scripts drive the oracle, the checker)
sys.path.insert(0, ROOT)
import noize_job_amd as nj
import oracle
f32 = np.float32
rng = np.random.default_rng(707)
target = int(sys.argv[1]) if len(sys.argv) > 1 else 7457
safe = len(sys.argv) > 2 and sys.argv[2] == "safe"
case = 0
while True:
    res = int(rng.integers(40, 420)); particles = int(rng.integers(50, 6000)); th = int(rng.choice([100, 500, 1000, 3000]))
    patch = float(rng.choice([0.5, 1.0, 2.5, 4.0])); workers = int(rng.choice([1, 3, 10, 64]))
    kw = dict(PARTICLES_PER_CYCLE=particles, MAXAGE=int(rng.integers(3, 160)), PILING_RADIUS=int(rng.integers(1, 21)),
              PILE_THRESHOLD=float(rng.choice([0.05, 0.4, 2.0, 6.0])), MIN_PILE_INCREMENT=float(rng.choice([0.05, 0.25, 1.0, 3.0])),
              INERTIA=float(rng.uniform(0.0, 0.95)), GRAVITY=float(rng.uniform(0.5, 20.0)), FRICTION=float(rng.uniform(0.0, 3.0)),
              DRAG=float(rng.uniform(0.0005, 0.3)), EVAP=float(rng.uniform(0.0, 0.2)), CAPACITY=float(rng.uniform(0.05, 8.0)),
              EROSION=float(rng.uniform(0.0, 0.9)), DEPOSITION=float(rng.uniform(0.0, 0.9)),
              FLOW_HEIGHT_CONTRIBUTION=float(rng.uniform(0.0, 30.0)))
    octaves = int(rng.integers(1, 9)); basis = int(rng.integers(0, 2)) * 3; size = int(rng.integers(40, 600)); fit = int(rng.integers(0, 4))
    terr = rng.random() < 0.3
    ta = tb = None
    if terr:
        ta, tb = rng.integers(5, 60), rng.integers(5, 60)
    p_frac_draw = rng.random((res, res)); p_u = rng.uniform(0, 0.1); p_vals = rng.random((res, res), dtype=f32)
    f_vals = rng.random((res, res), dtype=f32); f_u = rng.uniform(0, 0.9)
    seeds = []
    for cyc in range(2):
        seeds.append((int(rng.integers(1, 2 ** 31 - 1)), bool(rng.integers(0, 2))))
    if case == target:
        break
    case += 1
print("case", case, "res", res, "particles", particles, "th", th, "patch", patch, "workers", workers, kw, seeds, flush=True)
h = oracle.kernel_filter(oracle.fractal(basis, res, res, 0.4, 1.0, 2.0, 0.0, octaves, 0, 0, size), 2, fit)
if terr:
    h = (np.round(h * f32(ta)) / f32(tb)).astype(f32)
h = np.clip(h, 0, 1).astype(f32)
es = nj.ErosionSettings(**kw)
tm = nj.tile_set_meta(res, height=th, tile_size=res, tile_res=res, patch_res=patch)
pool0 = np.where(p_frac_draw < p_u, p_vals * f32(0.01), 0).astype(f32)
flow0 = (f_vals * f32(f_u)).astype(f32)
with nj.Context(0) as ctx:
    G = nj.LiveErosion(ctx, ctx.from_host(h), tm, es, queueCapacity=1 << 17)
    G.safe = safe
    ep = es.AsParameters()
    L = oracle.LiveErosionOracle(h, oracle.erosion_params(**{n: getattr(ep, n) for n, _ in ep._fields_}), tile_height=th, patch_res=patch, capacity=1 << 17)
    G.poolMap.CopyFrom(pool0); G.streamMap.CopyFrom(flow0); L.pool[:] = pool0; L.flow[:] = flow0
    epp, tmp_ = C.byref(ep), C.byref(tm)
    shape = (res, res)
    for cyc, (seed, one_call) in enumerate(seeds):
        G.ctx.call("nz_fill_beyer_queue", G.particleQueue._h, epp, tmp_, cyc, res, particles, seed, workers)
        L.fill_queue(cyc, particles, seed, workers)
        G.ctx.call("nz_queued_beyer_cycle", G.heightMap.ptr, G.poolMap.ptr, G.streamMap.ptr, G.particleTrack.ptr, G.particleQueue._h, G.events._h, epp, tmp_, 1500, res)
        n = L.descend()
        G.ctx.call("nz_process_beyer_erosive_events", G.heightMap.ptr, G.poolMap.ptr, G.streamMap.ptr, G.particleTrack.ptr, G.events._h, epp, tmp_, res)
        L.process_events()
        thr = f32(ep.PILE_THRESHOLD) / f32(th)
        print("cycle", cyc, "events", n, "piled cells", int((L.sediment > thr).sum()), "max sediment / increment", float(L.sediment.max() / (ep.MIN_PILE_INCREMENT / th)), flush=True)
        G.particleQueue.Clear()
        ctx.synchronize()
        t0 = time.time()
        if one_call:
            G.ctx.call("nz_erode_height_maps_and_flow", G.heightMap.ptr, G.events._h, G.poolMap.ptr, G.streamMap.ptr, G.particleTrack.ptr, epp, tmp_, res)
        else:
            G.ctx.call("nz_erode_height_maps", G.heightMap.ptr, G.events._h, epp, tmp_, res)
        try:
            ctx.synchronize()
            print("   GPU erode %.3f s (one_call %s, retries %d)" % (time.time() - t0, one_call, G.pileRetries), flush=True)
        except nj.NoizeError as e:
            print("   GPU erode FAILED after %.3f s: %s" % (time.time() - t0, str(e)[:120]), flush=True)
            sys.exit(1)
        t0 = time.time()
        L.erode_height_maps()
        print("   oracle erode %.3f s; equal: %s" % (time.time() - t0, bool(np.array_equal(G.heightMap.ToArray(shape), L.height))), flush=True)
        if not one_call:
            G.ctx.call("nz_update_flow_from_track", G.poolMap.ptr, G.streamMap.ptr, G.particleTrack.ptr, ep.FLOW_LOSS_RATE, ep.SURFACE_EVAPORATION_RATE, float(th), res)
        L.update_flow_from_track()
        G.ctx.call("nz_pool_automata_job", G.poolMap.ptr, G.heightMap.ptr, G.particleQueue._h, epp, tmp_, 3, res, 1)
        L.pool_automata(3, drain=True)
    G.OnDestroy()
