"""The particle half of live erosion (BASELINE config 4): the oracle's own known answers on the CPU, and HIP == oracle
bit for bit on the GPU (same seeds, same planes), plus the invariants that tie the restatement to the reference's
intent (Geologic/ParticleErosion/*)."""
import ctypes as C
import math
import os

import numpy as np
import pytest

f32 = np.float32


def terrain(oracle, res, basis=None, octaves=8, size=300):
    basis = oracle.SIMPLEX if basis is None else basis
    return oracle.kernel_filter(oracle.fractal(basis, res, res, 0.4, 1.0, 2.0, 0.0, octaves, 0, 0, size), 2, 3)


# ---- oracle known answers (CPU) ------------------------------------------------------------------------------------
def test_atan_sin_polynomials_are_accurate(oracle):
    for x in np.concatenate([np.linspace(0, 3, 3001), np.linspace(3, 4000, 2001)]).astype(f32):
        assert abs(oracle.live_atanf(float(x)) - math.atan(float(x))) < 2.5e-7
    for x in np.linspace(0, math.pi / 2, 4001).astype(f32):
        assert abs(oracle.live_sinf(float(x)) - math.sin(float(x))) < 1.5e-7
    assert oracle.live_atanf(0.0) == 0.0 and oracle.live_sinf(0.0) == 0.0 and math.isnan(oracle.live_atanf(float("nan")))


def test_spawn_follows_xorshift32_and_the_worker_layout(oracle):
    ep = oracle.erosion_params()
    L = oracle.LiveErosionOracle(np.zeros((64, 64), f32), ep, capacity=4096)
    assert L.fill_queue(1, 1000, 777, 10) == 1000 and L.count.value == 1000
    q = L.queued()
    assert q["px"].min() >= 0 and q["px"].max() < 64 and q["pz"].min() >= 0 and q["pz"].max() < 64
    assert np.all(q["water"] == 1.0)
    # worker 3's first particle: Random(seed + 3) -> state = seed + 3 after one xorshift step is DISCARDED (NextState
    # returns the old state), NextInt2 = two draws, x first
    s = (777 + 3) & 0xFFFFFFFF

    def nxt(s):
        t = s
        s ^= (s << 13) & 0xFFFFFFFF
        s ^= s >> 17
        s ^= (s << 5) & 0xFFFFFFFF
        return t, s
    _, s = nxt(s)
    a, s = nxt(s)
    b, s = nxt(s)
    assert (q["px"][300], q["pz"][300]) == ((a * 64) >> 32, (b * 64) >> 32)
    # pid: Convert.ToUInt16(generation * size) then `pid += (thread * count) + i`, cumulative, wrapping at 16 bits
    pid = (1 * 1000) & 0xFFFF
    for k in range(3):
        pid = (pid + 3 * 100 + k) & 0xFFFF
        assert q["pid"][300 + k] == pid
    # a second call tops the queue up: required = max(1, maxParticles - Count) = 1 -> COUNT = max(floor(1 / 10), 1) = 1
    assert L.fill_queue(2, 1000, 5, 10) == 10 and L.count.value == 1010


def test_a_particle_on_a_ramp_runs_downhill_and_its_sediment_is_conserved(oracle):
    res = 64
    ramp = np.tile((np.arange(res, dtype=f32) / f32(res))[:, None], (1, res))  # height rises with x: downhill = -x
    ep = oracle.erosion_params(MAXAGE=40)
    L = oracle.LiveErosionOracle(ramp, ep, tile_height=100, patch_res=1.0, capacity=16)
    L.queue[0] = (50, 32, 1.0, 7)
    L.count.value = 1
    n = L.descend()
    assert 2 <= n <= 42
    L.process_events()
    touched = np.argwhere(L.track > 0)
    assert len(touched) == n - 1 and touched[:, 0].max() == 50 and touched[:, 0].min() == 50 - (n - 2)  # one cell per step, x falling
    assert np.all(np.abs(touched[:, 1] - 32) <= n)
    # what the particle eroded it either deposited on the way or dropped where it died -- unless it left the tile
    left_tile = touched[:, 0].min() == 0
    total = float(L.sediment.sum(dtype=np.float64))
    assert (total < 0 and left_tile) or abs(total) < 1e-6
    assert L.track.max() == f32(80.0)  # the first step leaves water = 1 x TRACK_PLACEMENT_MULTIPLIER


def test_kernel_disperse_spreads_an_event_over_5x5_and_clamps_at_the_border(oracle):
    res = 32
    ep = oracle.erosion_params()
    L = oracle.LiveErosionOracle(np.full((res, res), 0.5, f32), ep)
    L.sediment[10, 12] = -0.001          # erosion: dispersed
    L.sediment[0, 0] = 0.0015            # deposit below PILE_THRESHOLD / HEIGHT = 0.002: dispersed, corner
    L.erode_height_maps()
    d = L.height.astype(np.float64) - 0.5
    k = np.array([0.12007838424321349, 0.23388075658535032, 0.29208171834287244, 0.23388075658535032, 0.12007838424321349])
    assert np.allclose(d[8:13, 10:15], -0.001 * np.outer(k, k), atol=1e-7)
    assert abs(d[8:13, 10:15].sum() + 0.001) < 1e-6 and abs(d[:3, :3].sum() - 0.0015) < 1e-6  # clamped taps pile up at the border
    assert d[0, 0] > d[1, 1] > d[2, 2] > 0 and np.count_nonzero(d) == 25 + 9


def test_pile_solver_raises_manhattan_rings(oracle):
    res = 48
    ep = oracle.erosion_params(PILING_RADIUS=6, MIN_PILE_INCREMENT=1.0, PILE_THRESHOLD=2.0)
    L = oracle.LiveErosionOracle(np.full((res, res), 0.25, f32), ep, tile_height=1000)
    L.sediment[20, 30] = 0.0105           # > 0.002 -> PileSolver, increments of 0.001
    L.erode_height_maps()
    d = L.height.astype(np.float64) - 0.25
    # ManhattanVertex lists every ring corner twice and the centre four times (GetOffset with i = 0 for each of the four
    # directions, :1196-1199): each copy is raised and COUNTED, one value is committed -- the reference's pile keeps
    # less than it was given.  Reproduced as is.
    assert 0.002 < d.sum() < 0.0105 and d.min() >= 0 and d[20, 30] > 0
    # round 1 raises the 8 vertices of dist 0 (4 x centre, 4 diagonals) by one increment each; round 2 spends the
    # remaining 2.5 increments on centre copy 0, the diagonal (1, -1) and centre copy 1; the LAST centre copy is what
    # CommitChanges leaves in the cell
    want = np.zeros((48, 48))
    want[20, 30] = 1
    want[21, 29], want[19, 29], want[19, 31], want[21, 31] = 2, 1, 1, 1
    assert np.allclose(d, want * 0.001, atol=2e-7)
    xx, zz = np.nonzero(d)
    assert np.all(np.abs(xx - 20) + np.abs(zz - 30) <= 6 + 1)   # i runs to dist + 1: one step beyond the radius


def test_drained_pools_become_particles(oracle):
    res = 16
    h = np.tile(np.linspace(0.2, 0.8, res, dtype=f32)[:, None], (1, res))
    ep = oracle.erosion_params()
    L = oracle.LiveErosionOracle(h, ep)
    L.pool[8, 8] = 0.01                   # the dry, lower neighbour (7, 8) takes all of it ... as a particle
    L.pool_automata(1, drain=True)
    q = L.queued()
    assert len(q) == 1 and (q["px"][0], q["pz"][0], q["pid"][0]) == (7, 8, 64000) and q["water"][0] == f32(0.01)
    assert L.pool.sum() == 0.0
    L2 = oracle.LiveErosionOracle(h, ep)
    L2.pool[8, 8] = 0.01
    L2.pool_automata(1, drain=False)      # drainParticles == false: the neighbour is wetted instead
    # (the later colour passes of the same iteration hand it on downhill)
    assert L2.pool.sum() == f32(0.01) and L2.pool[8, 8] == 0 and L2.count.value == 0


# ---- HIP == oracle (GPU) -----------------------------------------------------------------------------------------------
def _gpu_state(nj, ctx, height, settings, tile_height, patch_res, capacity=1 << 16):
    res = height.shape[0]
    tm = nj.tile_set_meta(res, height=tile_height, tile_size=res, tile_res=res, patch_res=patch_res)
    return nj.LiveErosion(ctx, ctx.from_host(height), tm, settings, queueCapacity=capacity)


def _params(oracle, es, tile_behaviour_all=True):
    ep = es.AsParameters()
    return oracle.erosion_params(**{n: getattr(ep, n) for n, _ in ep._fields_})


@pytest.mark.gpu
def test_spawn_matches_oracle(nj, ctx, oracle):
    es = nj.ErosionSettings(PARTICLES_PER_CYCLE=5000)
    G = _gpu_state(nj, ctx, np.zeros((200, 200), f32), es, 1000, 1.0)
    L = oracle.LiveErosionOracle(np.zeros((200, 200), f32), _params(oracle, es))
    ep, tm = es.AsParameters(), G.tileMeta
    for gen, seed in ((0, 1), (3, 2 ** 31 - 5), (2, 123456789)):
        G.ctx.call("nz_fill_beyer_queue", G.particleQueue._h, C.byref(ep), C.byref(tm), gen, 200, 5000, seed, 10)
        L.fill_queue(gen, 5000, seed, 10)
        assert np.array_equal(G.particleQueue.ToArray(), L.queued())
    G.OnDestroy()


@pytest.mark.gpu
@pytest.mark.parametrize("particles,workers", [(12345, 7), (200000, 1), (70000, 1024), (999, 64), (16, 3), (1, 10)])
def test_spawn_jumps_ahead_in_the_workers_streams(nj, ctx, oracle, particles, workers):
    """The kernel reaches a worker's k-th particle by jumping 32 * (k / 16) xorshift steps ahead (GF(2) matrices), the
    oracle walks there: worker counts that do not divide the particles, chunks that are not full, more chunks than
    threads, a second call on a part-filled queue, ids that wrap at 16 bits."""
    es = nj.ErosionSettings(PARTICLES_PER_CYCLE=particles)
    cap = 1 << 19
    G = _gpu_state(nj, ctx, np.zeros((300, 300), f32), es, 1000, 1.0, capacity=cap)
    L = oracle.LiveErosionOracle(np.zeros((300, 300), f32), _params(oracle, es), capacity=cap)
    ep, tm = es.AsParameters(), G.tileMeta
    for gen, seed, want in ((0, 99, particles), (3, 2 ** 31 - 7, particles), (1, 5, particles + particles // 3 + 1)):
        G.ctx.call("nz_fill_beyer_queue", G.particleQueue._h, C.byref(ep), C.byref(tm), gen, 300, want, seed, workers)
        L.fill_queue(gen, want, seed, workers)
        got, ref = G.particleQueue.ToArray(), L.queued()
        assert len(got) == len(ref) and np.array_equal(got, ref)
    G.OnDestroy()


@pytest.mark.gpu
@pytest.mark.parametrize("res,particles,tile_height,patch,radius",
                         [(256, 3000, 1000, 1.0, 7), (384, 6000, 500, 2.5, 7),
                          (40, 600, 1000, 1.0, 7),        # 40: a third of the cells in the frame
                          (256, 3000, 1000, 1.0, 30)])    # a pile solver with > 16 KB of LDS: the one-call form launches the
                                                          # flow update on its own again (nz_live.hip, pile_ticket_flow_kernel)
@pytest.mark.parametrize("siblings", ["two-calls", "one-call"])
def test_cycle_jobs_match_oracle_one_by_one(nj, ctx, oracle, res, particles, tile_height, patch, radius, siblings):
    """siblings = one-call: ErodeHeightMaps and UpdateFlowFromTrackJob through nz_erode_height_maps_and_flow (the pile
    solver's launch carries the flow update's workgroups) -- the same planes as the two entries one after the other."""
    h = terrain(oracle, res)
    es = nj.ErosionSettings(PARTICLES_PER_CYCLE=particles, PILE_THRESHOLD=0.4, PILING_RADIUS=radius, MIN_PILE_INCREMENT=0.25)
    G = _gpu_state(nj, ctx, h, es, tile_height, patch)
    L = oracle.LiveErosionOracle(h, _params(oracle, es), tile_height=tile_height, patch_res=patch)
    ep, tm = es.AsParameters(), G.tileMeta
    epp, tmp_ = C.byref(ep), C.byref(tm)
    rng = np.random.default_rng(5)
    pool0 = np.where(rng.random((res, res)) < 0.02, rng.random((res, res), dtype=f32) * f32(0.004), 0).astype(f32)
    flow0 = (rng.random((res, res), dtype=f32) * f32(0.3)).astype(f32)
    G.poolMap.CopyFrom(pool0); G.streamMap.CopyFrom(flow0)
    L.pool[:] = pool0; L.flow[:] = flow0
    shape = (res, res)
    for cyc in range(3):
        seed = 1000 + 17 * cyc
        G.ctx.call("nz_fill_beyer_queue", G.particleQueue._h, epp, tmp_, cyc % 4, res, particles, seed, 10)
        L.fill_queue(cyc % 4, particles, seed, 10)
        gq, lq = G.particleQueue.ToArray(), L.queued()
        assert np.array_equal(np.sort(gq, order=["px", "pz", "water", "pid"]), np.sort(lq, order=["px", "pz", "water", "pid"]))
        G.ctx.call("nz_queued_beyer_cycle", G.heightMap.ptr, G.poolMap.ptr, G.streamMap.ptr, G.particleTrack.ptr,
                   G.particleQueue._h, G.events._h, epp, tmp_, 1500, res)
        n = L.descend()
        assert G.events.Count == n and n > len(lq)
        G.ctx.call("nz_process_beyer_erosive_events", G.heightMap.ptr, G.poolMap.ptr, G.streamMap.ptr, G.particleTrack.ptr,
                   G.events._h, epp, tmp_, res)
        L.process_events()
        assert np.array_equal(G.events.sediment(), L.sediment), cyc
        assert np.array_equal(G.poolMap.ToArray(shape), L.pool) and np.array_equal(G.particleTrack.ToArray(shape), L.track)
        G.particleQueue.Clear()
        if siblings == "one-call":
            G.ctx.call("nz_erode_height_maps_and_flow", G.heightMap.ptr, G.events._h, G.poolMap.ptr, G.streamMap.ptr,
                       G.particleTrack.ptr, epp, tmp_, res)
        else:
            G.ctx.call("nz_erode_height_maps", G.heightMap.ptr, G.events._h, epp, tmp_, res)
        L.erode_height_maps()
        piles = int(((L.sediment > f32(ep.PILE_THRESHOLD) / f32(tile_height))).sum())
        assert np.array_equal(G.heightMap.ToArray(shape), L.height), (cyc, piles)
        if cyc == 0 and res >= 256:
            assert piles > 0, "the parameters were chosen so that the PileSolver path runs"
        if res < 256:  # events in the two-cell frame, where clamped taps of one source fold onto one target
            frame = np.ones(shape, bool); frame[2:-2, 2:-2] = False
            assert (L.sediment[frame] != 0).any()
        if siblings != "one-call":
            G.ctx.call("nz_update_flow_from_track", G.poolMap.ptr, G.streamMap.ptr, G.particleTrack.ptr, ep.FLOW_LOSS_RATE,
                       ep.SURFACE_EVAPORATION_RATE, float(tile_height), res)
        L.update_flow_from_track()
        assert np.array_equal(G.poolMap.ToArray(shape), L.pool) and np.array_equal(G.streamMap.ToArray(shape), L.flow)
        assert not G.particleTrack.ToArray(shape).any()
        G.ctx.call("nz_pool_automata_job", G.poolMap.ptr, G.heightMap.ptr, G.particleQueue._h, epp, tmp_, 4, res, 1)
        L.pool_automata(4, drain=True)
        assert np.array_equal(G.poolMap.ToArray(shape), L.pool) and np.array_equal(G.streamMap.ToArray(shape), L.flow)
    G.OnDestroy()


@pytest.mark.gpu
def test_descent_on_hostile_planes_matches_oracle(nj, ctx, oracle):
    """The descent step is written without branches (every way out a flag, every `if` a select): planes that push
    particles through all of them at once -- plateaus (slope 0: the 0 / 0 of the velocity model), spikes and pits of
    +-10^4, negative zero, denormals, negative pools, flow outside [0, 1] -- must still give the oracle's events."""
    res, th = 160, 1000
    rng = np.random.default_rng(11)
    h = terrain(oracle, res).copy()
    h[20:60, 30:90] = f32(0.37)                       # a plateau
    h[100:140, 100:140] = np.round(h[100:140, 100:140] * 50) / 50  # terraces
    r = rng.random((res, res))
    h[r < 0.01] = f32(1e4); h[(r >= 0.01) & (r < 0.02)] = f32(-1e4)
    h[(r >= 0.02) & (r < 0.03)] = f32(-0.0); h[(r >= 0.03) & (r < 0.04)] = f32(1e-40)
    pool0 = np.where(rng.random((res, res)) < 0.05, (rng.random((res, res), dtype=f32) - f32(0.3)) * f32(0.01), 0).astype(f32)
    flow0 = (rng.random((res, res), dtype=f32) * f32(2.0) - f32(0.5)).astype(f32)
    es = nj.ErosionSettings(PARTICLES_PER_CYCLE=4000)
    G = _gpu_state(nj, ctx, h, es, th, 1.0)
    L = oracle.LiveErosionOracle(h, _params(oracle, es), tile_height=th, patch_res=1.0)
    G.poolMap.CopyFrom(pool0); G.streamMap.CopyFrom(flow0)
    L.pool[:] = pool0; L.flow[:] = flow0
    ep, tm = es.AsParameters(), G.tileMeta
    epp, tmp_ = C.byref(ep), C.byref(tm)
    shape = (res, res)
    for cyc in range(3):
        G.ctx.call("nz_fill_beyer_queue", G.particleQueue._h, epp, tmp_, cyc, res, 4000, 31 + cyc, 10)
        L.fill_queue(cyc, 4000, 31 + cyc, 10)
        G.ctx.call("nz_queued_beyer_cycle", G.heightMap.ptr, G.poolMap.ptr, G.streamMap.ptr, G.particleTrack.ptr,
                   G.particleQueue._h, G.events._h, epp, tmp_, 1500, res)
        n = L.descend()
        assert G.events.Count == n
        G.ctx.call("nz_process_beyer_erosive_events", G.heightMap.ptr, G.poolMap.ptr, G.streamMap.ptr, G.particleTrack.ptr,
                   G.events._h, epp, tmp_, res)
        L.process_events()
        assert np.array_equal(G.events.sediment(), L.sediment, equal_nan=True), cyc
        assert np.array_equal(G.poolMap.ToArray(shape), L.pool, equal_nan=True)
        assert np.array_equal(G.particleTrack.ToArray(shape), L.track, equal_nan=True)
        G.particleQueue.Clear(); L.count.value = 0
    G.OnDestroy()


@pytest.mark.gpu
@pytest.mark.parametrize("res,particles,updates,cycles,water_steps", [(512, 10000, 3, 3, 5), (2048, 40000, 1, 2, 10),
                                                                      (8192, 10000, 1, 2, 10)])  # the last: config 4 itself
def test_config4_live_erosion_equals_oracle(nj, ctx, oracle, res, particles, updates, cycles, water_steps):
    # BASELINE config 4 at sizes the oracle finishes in seconds: cellular fBm 13 octaves -> LiveErosion, the driver loop
    # of TriggerQueuedBeyerMT (thermal -> spawn -> descent -> event reduce -> sediment -> flow from track -> pool
    # automaton with drains), control textures included.  2048^2: 32 mask words per automaton walk, the bits cleaned
    # nine times per cycle, WATER_STEPS at the config's 10.  8192^2, 10 000 particles per cycle, WATER_STEPS 10 IS
    # BASELINE config 4: every plane and the particle queue bit-equal to the oracle after two cycles.
    th = 1000
    h = oracle.fractal(oracle.CELLULAR, res, res, 0.4, 1.0, 2.0, 0.0, 13, 0, 0, 1700)
    es = nj.ErosionSettings(PARTICLES_PER_CYCLE=particles, CYCLES=cycles, WATER_STEPS=water_steps)
    tm = nj.tile_set_meta(res, height=th, tile_size=2000, tile_res=res - 16, margin=8)
    G = nj.LiveErosion(ctx, ctx.from_host(h), tm, es)
    G.EnableControlTextures()
    L = oracle.LiveErosionOracle(h, _params(oracle, es), tile_height=th, patch_res=float(tm.PATCH_RES[0]))
    shape = (res, res)
    gen = 0
    for update in range(updates):
        seeds = [40000 * update + 11 * c + 3 for c in range(es.CYCLES)]
        G.TriggerQueuedBeyerMT(seeds).Complete()
        for c in range(es.CYCLES):
            L.cycle(gen % 4, particles, seeds[c], water_steps=es.WATER_STEPS,
                    thermal=(es.TALUS, es.THERMAL_STEP, float(2000 // th), es.THERMAL_CYCLES))
        gen += 1
        for name, got, want in (("height", G.heightMap, L.height), ("pool", G.poolMap, L.pool), ("flow", G.streamMap, L.flow),
                                ("track", G.particleTrack, L.track)):
            assert np.array_equal(got.ToArray(shape), want), (update, name)
        assert np.array_equal(np.sort(G.particleQueue.ToArray(), order=["px", "pz", "water"]),
                              np.sort(L.queued(), order=["px", "pz", "water"]))
    # invariants of the model, whatever the seeds: pools never negative, the track is consumed, flow stays in [0, 1),
    # heights stay in [0, 1], erosion moved material but did not create it beyond what left the tile
    pool, flow, height = L.pool, L.flow, L.height
    assert pool.min() >= 0 and L.track.max() == 0 and 0 <= flow.min() and flow.max() < 1 and 0 <= height.min() and height.max() <= 1
    assert not np.array_equal(height, h) and float(np.abs(height - h).max()) < 0.2
    mres = tm.TILE_RES[0]
    tex = G.textureControl.ToArray().reshape(mres, mres, 4)
    wat = G.waterControl.ToArray().reshape(mres, mres, 4)
    assert np.array_equal(wat[..., 0], oracle.set_rgba32(L.pool, mres, 1000.0, 0)[..., 0])
    assert np.array_equal(wat[..., 2], oracle.set_rgba32(L.flow, mres, 2.0, 2)[..., 2])
    assert np.array_equal(tex[..., 3], oracle.set_rgba32(L.flow, mres, 1.0, 3)[..., 3])
    cur = oracle.curviture_map(L.height, mres, th, float(tm.PATCH_RES[0]), 1)[..., 1]
    assert np.abs(tex[..., 1].astype(int) - cur.astype(int)).max() <= 1   # powf / logf of the device: one byte step at most
    G.OnDestroy()


@pytest.mark.gpu
@pytest.mark.parametrize("res,particles,cycles", [(512, 10000, 300), (2048, 10000, 50)])
def test_live_erosion_soak_equals_oracle_at_ten_checkpoints(nj, ctx, oracle, res, particles, cycles):
    # The long-run regime of LiveErosion.TriggerQueuedBeyerMT (Component/LiveErosion.cs:378-436 loops
    # erosionSettings.CYCLES): hundreds of cycles on one tile -- the terrain is cut down by most of its relief, a tenth of
    # the cells hold water, pools fill, drain and re-enter the particle queue, piles accumulate -- compared with the
    # oracle at ten checkpoints: every plane and the particle queue, bit for bit.
    th, per = 1000, cycles // 10
    h = oracle.fractal(oracle.CELLULAR, res, res, 0.4, 1.0, 2.0, 0.0, 13, 0, 0, 1700)
    es = nj.ErosionSettings(PARTICLES_PER_CYCLE=particles, CYCLES=per, WATER_STEPS=10)
    tm = nj.tile_set_meta(res, height=th, tile_size=2000, tile_res=res - 16, margin=8)
    G = nj.LiveErosion(ctx, ctx.from_host(h), tm, es)
    L = oracle.LiveErosionOracle(h, _params(oracle, es), tile_height=th, patch_res=float(tm.PATCH_RES[0]))
    shape = (res, res)
    wet = []
    for cp in range(10):
        seeds = [7919 * cp + 13 * c + 1 for c in range(per)]
        G.TriggerQueuedBeyerMT(seeds).Complete()
        for c in range(per):
            L.cycle(cp % 4, particles, seeds[c], water_steps=es.WATER_STEPS,
                    thermal=(es.TALUS, es.THERMAL_STEP, float(2000 // th), es.THERMAL_CYCLES))
        for name, got, want in (("height", G.heightMap, L.height), ("pool", G.poolMap, L.pool), ("flow", G.streamMap, L.flow),
                                ("track", G.particleTrack, L.track)):
            assert np.array_equal(got.ToArray(shape), want), (cp, name)
        assert np.array_equal(np.sort(G.particleQueue.ToArray(), order=["px", "pz", "water"]),
                              np.sort(L.queued(), order=["px", "pz", "water"])), cp
        wet.append(float((L.pool > 0).mean()))
    assert float(np.abs(L.height - h).max()) > 0.02 and wet[-1] > 0.001   # the run did leave the fresh-terrain regime
    G.OnDestroy()


@pytest.mark.gpu
def test_config4_full_size_is_deterministic_and_keeps_its_invariants(nj, ctx, oracle):
    # BASELINE config 4 at its full size, 8192^2 cellular fBm 13 octaves + live erosion: too large to hand to the oracle
    # cycle by cycle in the suite's time, so the size-independent properties -- two runs from the same seeds agree bit
    # for bit (every atomic in the path is an integer add or an order-free append), a different seed does not, pools
    # and flow stay in range, the particle track is consumed, heights stay in [0, 1]
    res, particles = 8192, 10000
    es = nj.ErosionSettings(PARTICLES_PER_CYCLE=particles, CYCLES=2, WATER_STEPS=2)
    tm = nj.tile_set_meta(res, height=1000, tile_size=8000, tile_res=res - 16, margin=8)
    base = nj.GeneratorData("c4", ctx.alloc(res * res), res, 0, 0)
    st = nj.NoiseStage(ctx, nj.FractalNoise.Cellular, 0.4, 1.0, 13, 2.0, 0.0, 1700)
    st.ReceiveHandledInput(nj.PipelineWorkItem(base), nj.JobHandle())
    st.jobHandle.Complete()
    runs = []
    for seeds in ([5, 6], [5, 6], [5, 7]):
        h = ctx.alloc(res * res)
        ctx.call("nz_flush_write_slice", h.ptr, base.data.ptr, res * res).Complete()
        G = nj.LiveErosion(ctx, h, tm, es)
        G.TriggerQueuedBeyerMT(seeds).Complete()
        runs.append((h.ToArray(), G.poolMap.ToArray(), G.streamMap.ToArray(), G.particleTrack.ToArray(), G.events.Count,
                     np.sort(G.particleQueue.ToArray(), order=["px", "pz", "water"])))
        G.OnDestroy()
        h.Dispose()
    a, b, c = runs
    for i in range(4):
        assert np.array_equal(a[i], b[i])
    assert a[4] == b[4] and np.array_equal(a[5], b[5])
    assert not np.array_equal(a[0], c[0])
    height, pool, flow, track = a[:4]
    assert pool.min() >= 0 and track.max() == 0 and flow.min() >= 0 and flow.max() < 1 and 0 <= height.min() and height.max() <= 1
    assert a[4] > particles   # events of the last cycle: every particle leaves at least its death event
    base.data.Dispose()


@pytest.mark.gpu
def test_safe_mode_runs_a_timed_out_pile_launch_again_colour_by_colour(nj, oracle):
    """nz_ctx_set_pile_safe: with a poll limit of 1 every block of the ticket launch that has to wait for a neighbour at all gives
    up -- the height plane it leaves is invalid.  In safe mode ErodeHeightMaps notices, puts the plane back and runs itself again
    with a launch per colour: the caller sees the oracle's plane, the retry counter says the path ran, and the context stays
    on the colour launches.  Without safe mode the same time-out is an error at the next wait."""
    lib = nj._native.lib
    res, particles, th = 256, 4000, 1000
    h = terrain(oracle, res)
    es = nj.ErosionSettings(PARTICLES_PER_CYCLE=particles, PILE_THRESHOLD=0.2, PILING_RADIUS=7, MIN_PILE_INCREMENT=0.25)
    if os.environ.get("NZ_PILE_TICKET") == "0":
        pytest.skip("knob matrix: no ticket launch")

    def cycles(ctx, safe, n=3):
        G = _gpu_state(nj, ctx, h, es, th, 1.0)
        G.safe = safe
        L = oracle.LiveErosionOracle(h, _params(oracle, es), tile_height=th, patch_res=1.0)
        ep, tm = es.AsParameters(), G.tileMeta
        epp, tmp_ = C.byref(ep), C.byref(tm)
        for cyc in range(n):
            seed = 77 + cyc
            G.ctx.call("nz_fill_beyer_queue", G.particleQueue._h, epp, tmp_, cyc % 4, res, particles, seed, 10)
            L.fill_queue(cyc % 4, particles, seed, 10)
            G.ctx.call("nz_queued_beyer_cycle", G.heightMap.ptr, G.poolMap.ptr, G.streamMap.ptr, G.particleTrack.ptr,
                       G.particleQueue._h, G.events._h, epp, tmp_, 1500, res)
            L.descend()
            G.ctx.call("nz_process_beyer_erosive_events", G.heightMap.ptr, G.poolMap.ptr, G.streamMap.ptr, G.particleTrack.ptr,
                       G.events._h, epp, tmp_, res)
            L.process_events()
            G.particleQueue.Clear()
            G.ctx.call("nz_erode_height_maps_and_flow", G.heightMap.ptr, G.events._h, G.poolMap.ptr, G.streamMap.ptr,
                       G.particleTrack.ptr, epp, tmp_, res)
            L.erode_height_maps()
            L.update_flow_from_track()
            got = G.heightMap.ToArray((res, res))       # (waits for the context)
            assert np.array_equal(got, L.height), cyc
            assert np.array_equal(G.poolMap.ToArray((res, res)), L.pool) and np.array_equal(G.streamMap.ToArray((res, res)), L.flow)
        return G

    try:
        assert lib.nz_debug_pile_poll_limit(1) == 0
        with nj.Context(0) as c:
            G = cycles(c, True)
            assert G.pileRetries >= 1, "the forced time-out did not happen: the test has no power"
            first = G.pileRetries
            G.OnDestroy()
            G = cycles(c, True, n=1)                     # the context stays on the colour launches: nothing to retry any more
            assert G.pileRetries == first
            G.OnDestroy()
        with nj.Context(0) as c:                         # without safe mode: the error reaches the caller's wait
            with pytest.raises(nj.NoizeError) as e:
                cycles(c, False)
            assert e.value.status == nj._native.NZ_ERR_HIP
    finally:
        lib.nz_debug_pile_poll_limit(0)
    with nj.Context(0) as c:                             # and with the default limit nothing ever times out
        G = cycles(c, True)
        assert G.pileRetries == 0
        G.OnDestroy()


@pytest.mark.gpu
def test_a_block_that_works_for_seconds_is_not_a_time_out(nj, oracle):
    """Round 6's soak (tests/soak_live.py --seed 707, case 7457): 5 000 droplets on an 89^2 plane, increments of a 60 000th of the
    tile height -- one pile of 1.4 million increments keeps a block of the pile solver busy for seconds.  Its neighbours' bounded
    wait counts polls WITHOUT a sign of life from the block they wait for (the flag carries a heartbeat), so it does not expire:
    forced here with a bound of ~20 ms without progress, which the old bound on the waiting time itself could not survive."""
    lib = nj._native.lib
    if os.environ.get("NZ_PILE_TICKET") == "0":
        pytest.skip("knob matrix: no ticket launch")
    res, particles, th = 89, 5013, 3000
    es = nj.ErosionSettings(PARTICLES_PER_CYCLE=particles, MAXAGE=70, PILING_RADIUS=10, PILE_THRESHOLD=0.05, MIN_PILE_INCREMENT=0.05,
                            INERTIA=0.18391865, GRAVITY=18.4762, FRICTION=1.55434888, DRAG=0.00306747, EVAP=0.03214284,
                            CAPACITY=7.83815, EROSION=0.85846327, DEPOSITION=0.23305014, FLOW_HEIGHT_CONTRIBUTION=17.42407852)
    h = np.clip(terrain(oracle, res, octaves=5, size=200), 0, 1).astype(f32)
    try:
        assert lib.nz_debug_pile_poll_limit(20000) == 0
        with nj.Context(0) as c:
            G = _gpu_state(nj, c, h, es, th, 2.5, capacity=1 << 17)
            L = oracle.LiveErosionOracle(h, _params(oracle, es), tile_height=th, patch_res=2.5, capacity=1 << 17)
            ep, tm = es.AsParameters(), G.tileMeta
            epp, tmp_ = C.byref(ep), C.byref(tm)
            longest = 0.0
            for cyc, seed in enumerate((476132161, 1146990835)):
                G.ctx.call("nz_fill_beyer_queue", G.particleQueue._h, epp, tmp_, cyc, res, particles, seed, 64)
                L.fill_queue(cyc, particles, seed, 64)
                G.ctx.call("nz_queued_beyer_cycle", G.heightMap.ptr, G.poolMap.ptr, G.streamMap.ptr, G.particleTrack.ptr,
                           G.particleQueue._h, G.events._h, epp, tmp_, 1500, res)
                L.descend()
                G.ctx.call("nz_process_beyer_erosive_events", G.heightMap.ptr, G.poolMap.ptr, G.streamMap.ptr, G.particleTrack.ptr,
                           G.events._h, epp, tmp_, res)
                L.process_events()
                longest = max(longest, float(L.sediment.max() / (f32(ep.MIN_PILE_INCREMENT) / f32(th))))
                G.particleQueue.Clear()
                G.ctx.call("nz_erode_height_maps", G.heightMap.ptr, G.events._h, epp, tmp_, res)
                L.erode_height_maps()
                assert np.array_equal(G.heightMap.ToArray((res, res)), L.height), cyc      # (the wait: no NZ_ERR_HIP)
                G.ctx.call("nz_update_flow_from_track", G.poolMap.ptr, G.streamMap.ptr, G.particleTrack.ptr, ep.FLOW_LOSS_RATE,
                           ep.SURFACE_EVAPORATION_RATE, float(th), res)
                L.update_flow_from_track()
                G.ctx.call("nz_pool_automata_job", G.poolMap.ptr, G.heightMap.ptr, G.particleQueue._h, epp, tmp_, 3, res, 1)
                L.pool_automata(3, drain=True)
            assert longest > 2e5, "no pile long enough to outlast the bound: the test has no power (%g increments)" % longest
            assert G.pileRetries == 0
            G.OnDestroy()
    finally:
        lib.nz_debug_pile_poll_limit(0)
