"""The row-stripe schedule and the RCCL halo exchange BEHIND the C ABI (nz_comm.cpp: nz_comm_*, nz_halo_exchange*,
nz_sharded_*): what a C# / C++ host calls to run BASELINE config 5.

CPU half (no GPU): the launch plan the library compiles (rows per exchange, launch order, interior / border split) equals
what noize_job_amd.sharded.pipeline_steps does -- the Python schedule the gloo tests drive (tests/test_sharded_cpu.py) is
the specification.  GPU half: world 1, several lockstep stripes whose ghost rows travel through native ncclSend / ncclRecv
with the rank itself as peer, bit-equal to the monolithic tile and to the oracle; the >= 2-GPU cases are skipped where
one GPU is visible."""
import itertools
import multiprocessing as mp
import os

import numpy as np
import pytest

from conftest import ROOT


def _same_marks(got, want, OP_MARK=7):
    """The library marks all five stage boundaries, stages left out included (empty intervals, so that a host always
    finds five handles); the Python schedule marks the stages it runs.  Everything else must agree."""
    have = {r[1] for r in want if r[0] == OP_MARK}
    return [r for r in got if r[0] != OP_MARK or r[1] in have]


def _native_plan(sh, rank, world, grows, cols, p, overlap, stripes=None):
    g = sh.ShardedGrid(None, None, grows, cols, p, stripes=stripes if stripes is not None else world, overlap=overlap,
                       as_rank=(rank, world))
    try:
        return g.plan(), [g.stripe(i)[0] for i in range(g.local_stripes)]
    finally:
        g.close()


def _exchanges_planned(sh, grows, cols, p, stripes, as_rank=(0, 1)):
    """Exchange batches per step of the native plan: one per filter launch + the flow launch + the erosion launch under the
    default fusion depths (6 for the metric list), more or fewer under the NZ_FLOW_NMAX / NZ_EROSION_EMAX
    knobs of tools/run_knob_matrix.sh -- the rows sent do not depend on how the applications are grouped."""
    g = sh.ShardedGrid(None, None, grows, cols, p, stripes=stripes, overlap=0, as_rank=as_rank)
    try:
        return sum(1 for r in g.plan() if r[0] == 2)
    finally:
        g.close()


def _default_fusion():
    return not any(os.environ.get(k) for k in ("NZ_FLOW_NMAX", "NZ_EROSION_EMAX"))


PARAM_SETS = [
    dict(),                                                                   # the metric pipeline: Gauss5 x17, flow x5, erosion x5
    dict(filter=3, gaussIterations=7, flowIterations=3, erosionIterations=1),  # 3 taps
    dict(filter=0, gaussIterations=4, flowIterations=12, erosionIterations=11),  # 9 taps; flow in 3 launches, erosion in 2
    dict(filter=5, gaussIterations=1, flowIterations=0, erosionIterations=0),  # 7 taps, one application, nothing else
    dict(gaussIterations=0, flowIterations=6, erosionIterations=0),            # flow only, two launches: state planes travel
    dict(gaussIterations=0, flowIterations=0, erosionIterations=3),
]


@pytest.mark.parametrize("mode", ["exchange", "recompute", "exchange_once"])
@pytest.mark.parametrize("overlap", [1, 0])   # interior rows first (the Python schedule's overlapped form) / no split
def test_native_plan_equals_the_python_schedule(nj, mode, overlap):
    from noize_job_amd import sharded as sh
    for kw, (world, grows) in itertools.product(PARAM_SETS, [(1, 300), (2, 700), (4, 4096), (3, 1000), (8, 16384)]):
        p = sh.PipelineParams(haloMode=mode, **kw)
        for rank in sorted({0, world // 2, world - 1}):
            want, result_plane, plan = sh.python_plan(rank, world, grows, 512, p, overlap=bool(overlap))
            got, stripes = _native_plan(sh, rank, world, grows, 512, p, overlap)
            assert all(r[1] in (-1, 0) for r in got)
            assert _same_marks([(r[0],) + r[2:] for r in got], want) == want, (kw, world, rank)
            assert [r[2] for r in got if r[0] == sh.OP_MARK] == [0, 1, 2, 3, 4]
            st = stripes[0]
            assert (st.cols, st.rows, st.grow0, st.grows, st.own0, st.own1) == \
                   (plan.cols, plan.rows, plan.grow0, plan.grows, plan.own0, plan.own1)


def test_native_plan_with_two_stripes_per_rank(nj):
    # rank r holds stripes 2r and 2r + 1: each stripe's launches are those of the Python schedule for that stripe of
    # 2 * world, the exchanges are shared by the rank's stripes
    from noize_job_amd import sharded as sh
    p = sh.PipelineParams(haloMode="exchange")
    world, grows = 4, 4096
    for rank in range(world):
        got, stripes = _native_plan(sh, rank, world, grows, 256, p, 1, stripes=2 * world)
        assert len(stripes) == 2
        common = [r for r in got if r[1] == -1]
        for j in range(2):
            want, _, plan = sh.python_plan(2 * rank + j, 2 * world, grows, 256, p, overlap=True)
            mine = [(r[0],) + r[2:] for r in got if r[1] in (-1, j)]
            assert _same_marks(mine, want) == want
            assert (stripes[j].grow0, stripes[j].own0, stripes[j].own1) == (plan.grow0, plan.own0, plan.own1)
        assert [r[0] for r in common].count(sh.OP_XBEGIN) == 6 and [r[0] for r in common].count(sh.OP_XFINISH) == 6


def test_border_first_plan_covers_every_row_once(nj):
    # overlap 2 has no Python counterpart: a launch's rows come out as {rows sent up, rows sent down, the rest}, the
    # exchange for launch i + 1 is posted behind the first two, and every launch waits for its own exchange first
    from noize_job_amd import sharded as sh
    for kw in PARAM_SETS:
        p = sh.PipelineParams(haloMode="exchange", **kw)
        ref, _ = _native_plan(sh, 1, 4, 4096, 512, p, 0)
        got, _ = _native_plan(sh, 1, 4, 4096, 512, p, 2)
        launches = lambda plan: [r for r in plan if r[0] in (sh.OP_NOISE, sh.OP_FILTER, sh.OP_FLOW, sh.OP_EROSION)]
        # the same launches over the same rows: merge consecutive pieces of one launch
        merged = []
        for r in launches(got):
            key = (r[0], r[2], r[3], r[4], r[7])
            if merged and merged[-1][0] == key:
                merged[-1][1].append((r[5], r[6]))
            else:
                merged.append((key, [(r[5], r[6])]))
        want = [((r[0], r[2], r[3], r[4], r[7]), [(r[5], r[6])]) for r in launches(ref)]
        assert [k for k, _ in merged] == [k for k, _ in want]
        for (_, pieces), (_, whole) in zip(merged, want):
            rows = sorted(pieces)
            assert rows[0][0] == whole[0][0] and rows[-1][1] == whole[0][1]
            assert all(a[1] == b[0] for a, b in zip(rows, rows[1:]))   # no gap, no overlap
        # as many exchanges, with the same rows and planes, in the same order
        assert [r for r in got if r[0] == sh.OP_XBEGIN] == [r for r in ref if r[0] == sh.OP_XBEGIN]
        # an exchange is finished before the first launch piece that follows the NEXT begin's predecessor launch
        ops = [r[0] for r in got]
        assert ops.count(sh.OP_XFINISH) >= ops.count(sh.OP_XBEGIN)


@pytest.mark.parametrize("mode", ["exchange", "exchange_once"])
def test_every_send_meets_its_receive_on_every_pair_of_ranks(nj, mode):
    # No box offers two GPUs, so the lists are checked where they can be: the plans of ALL ranks of a job side by side.
    # RCCL matches the k-th send of rank a to rank b with the k-th receive rank b posts from rank a (inside one group per
    # exchange): the two sequences must have the same length and the same sizes, for every pair, in every exchange; a
    # rank never talks to anyone but its two neighbours; a transfer between two stripes of one rank posts both ends.
    from noize_job_amd import sharded as sh
    for kw, (world, per_rank), overlap in itertools.product(PARAM_SETS, [(2, 1), (3, 1), (4, 2), (8, 1), (8, 2)], (0, 1, 2)):
        p = sh.PipelineParams(haloMode=mode, **kw)
        grows = 4096
        lists = []
        for rank in range(world):
            g = sh.ShardedGrid(None, None, grows, 640, p, stripes=world * per_rank, overlap=overlap, as_rank=(rank, world))
            lists.append(g.transfers())
            exchanges = g.traffic()[0]
            g.close()
        assert all(len({t[0] for t in l}) <= exchanges for l in lists)
        for x in range(exchanges):
            for a in range(world):
                mine = [t for t in lists[a] if t[0] == x]
                assert all(abs(t[1] - t[2]) <= 1 and a in (t[1], t[2]) for t in mine), (kw, world, a)
                for b in range(world):
                    sends = [t[3] for t in mine if t[1] == a and t[2] == b]
                    recvs = [t[3] for t in lists[b] if t[0] == x and t[1] == a and t[2] == b]
                    assert sends == recvs, (kw, world, overlap, x, a, b)
        # and something does travel between every pair of neighbours
        for a in range(world - 1):
            assert sum(t[3] for t in lists[a] if t[1] == a and t[2] == a + 1) > 0, (kw, world, a)


def test_sharded_create_rejects_what_cannot_run(nj):
    from noize_job_amd import sharded as sh
    with pytest.raises(nj.NoizeError):   # stripes thinner than the ghost rows an exchange hands over
        _native_plan(sh, 0, 64, 512, 64, sh.PipelineParams(haloMode="exchange"), True)
    with pytest.raises(nj.NoizeError):   # stripes must be a multiple of the world size
        _native_plan(sh, 0, 4, 4096, 64, sh.PipelineParams(), True, stripes=6)
    with pytest.raises(nj.NoizeError):   # a filter without a fused stripe kernel (Sobel3_2D)
        _native_plan(sh, 0, 2, 4096, 64, sh.PipelineParams(filter=11), True)
    g = sh.ShardedGrid(None, None, 4096, 64, sh.PipelineParams(), as_rank=(0, 2))
    with pytest.raises(nj.NoizeError):   # a plan-only object cannot run
        nj._native.check(nj._native.lib.nz_sharded_pipeline(None, g._h, None, 0, None), "nz_sharded_pipeline")
    g.close()


# ---- GPU half ----------------------------------------------------------------------------------------------------------
def _mono(nj, ctx, rows_cols, p):
    """The monolithic run of the same kernels through the tile API (square tiles only)."""
    res = rows_cols
    data = ctx.alloc(res * res)
    stages = [nj.NoiseStage(ctx, p.noiseType, p.hurst, p.startingAmplitude, p.octaves, p.stepdown, p.detuneRate, p.noiseSize)]
    if p.gaussIterations:
        stages.append(nj.KernelFilterStage(ctx, p.filter, p.gaussIterations))
    if p.flowIterations:
        stages.append(nj.FlowMapStage(ctx, p.flowIterations, p.normMin, p.normMax))
    if p.erosionIterations:
        stages.append(nj.ErosionStage(ctx, p.erosionIterations))
    pipe = nj.BasePipeline(stages)
    pipe.Enqueue(nj.GeneratorData("mono", data, res, p.xpos, p.zpos))
    pipe.RunToCompletion()
    out = data.ToArray((res, res))
    pipe.Destroy()
    data.Dispose()
    return out


def _rccl_worker(kind, out_path, args):
    """Child process (hard time limit in the parent): everything that talks to RCCL."""
    import sys
    sys.path.insert(0, ROOT)
    import noize_job_amd as nj
    from noize_job_amd import sharded as sh
    ctx = nj.Context(0)
    comm = sh.NativeComm(ctx, sh.NativeComm.unique_id(), 0, 1)
    res = {"rccl": sh.rccl_version()}
    if kind == "grid":
        grows, cols, pkw, stripes, overlap = args
        p = sh.PipelineParams(**pkw)
        g = sh.ShardedGrid(ctx, comm, grows, cols, p, stripes=stripes, overlap=overlap)
        for _ in range(2):  # a second pass over the same planes gives the same grid
            g.run()
        parts = [g.owned_rows(i) for i in range(g.local_stripes)]
        assert [a for a, _ in parts] == sorted(a for a, _ in parts) and parts[0][0] == 0
        res["grid"] = np.concatenate([b for _, b in parts], axis=0)
        res["traffic"] = np.array(g.traffic(), np.int64)
        # the path's one collective: GetMapRangeJob -> ncclAllGather -> fold -> NormalizeMap with device args
        rng = ctx.alloc(3)
        g.map_range(rng.ptr)
        g.normalize(rng.ptr)
        res["range"] = rng.ToArray()
        res["norm"] = np.concatenate([g.owned_rows(i)[1] for i in range(g.local_stripes)], axis=0)
        g.close()
    elif kind == "grid_vs_file":
        # a grid too large to hand back through a file: the parent's oracle plane is memory-mapped here and every stripe's owned
        # rows are compared in place (two passes: the second runs on the planes the first one left behind)
        grows, cols, pkw, stripes, overlap, want_path = args
        want = np.load(want_path, mmap_mode="r")
        g = sh.ShardedGrid(ctx, comm, grows, cols, sh.PipelineParams(**pkw), stripes=stripes, overlap=overlap)
        same, rows = [], 0
        for _ in range(2):
            g.run()
        for i in range(g.local_stripes):
            g0, got = g.owned_rows(i)
            same.append(bool(np.array_equal(got, want[g0:g0 + got.shape[0]])))
            rows += got.shape[0]
        res["same"] = np.array(same)
        res["rows"] = np.array([rows])
        res["traffic"] = np.array(g.traffic(), np.int64)
        g.close()
    elif kind == "rehearsal":
        grows, cols, mode, stripes = args
        p = sh.PipelineParams(haloMode=mode)
        g = sh.ShardedGrid(ctx, comm, grows, cols, p, stripes=stripes, as_rank=(3, 8))
        g.set_timing(True)
        for _ in range(3):
            g.run()
        ctx.synchronize()
        res["exchange_ms"] = np.array([g.exchange_ms()])
        res["traffic"] = np.array(g.traffic(), np.int64)
        g.close()
    comm.close()
    ctx.close()
    np.savez(out_path, **res)


def _in_child(tmp_path, kind, args, limit=240):
    out = str(tmp_path / ("%s.npz" % kind))
    proc = mp.get_context("spawn").Process(target=_rccl_worker, args=(kind, out, args))
    proc.start()
    proc.join(limit)
    if proc.is_alive():
        proc.kill()
        proc.join()
        pytest.fail("the RCCL worker did not finish within %d s" % limit)
    assert proc.exitcode == 0
    return np.load(out)


@pytest.mark.gpu
@pytest.mark.timeout(400, method="thread")
@pytest.mark.parametrize("mode,overlap,stripes", [("exchange", 0, 3), ("exchange", 1, 3), ("exchange", 2, 3),
                                                  ("exchange_once", 0, 2), ("recompute", 0, 4)])
def test_lockstep_stripes_through_native_rccl_equal_the_oracle(nj, ctx, oracle, tmp_path, mode, overlap, stripes):
    # world 1: every ghost row of the three stripes travels through ncclSend / ncclRecv posted by the library itself
    grows = cols = 384
    pkw = dict(octaves=8, noiseSize=300, gaussIterations=17, flowIterations=5, erosionIterations=5, xpos=100, zpos=900,
               haloMode=mode)
    r = _in_child(tmp_path, "grid", (grows, cols, pkw, stripes, overlap))
    assert int(r["rccl"]) >= 20000
    from noize_job_amd import sharded as sh
    want = oracle.pipeline(grows, cols, octaves=8, noise_size=300, xpos=100, zpos=900)
    assert np.array_equal(r["grid"], want)
    assert np.array_equal(r["grid"], _mono(nj, ctx, grows, sh.PipelineParams(**pkw)))
    exchanges, sent = r["traffic"]
    if mode == "recompute":
        assert exchanges == 0 and sent == 0
    elif mode == "exchange":
        # 4 filter launches (5 + 4 + 4 + 4 applications x 2 rows, both ways), the flow launch's heights (10 rows both ways), the
        # erosion launch (5 rows downwards) -- over the stripes' 2 inner edges
        assert exchanges == _exchanges_planned(sh, grows, cols, sh.PipelineParams(**pkw), stripes)
        assert exchanges == 6 or not _default_fusion()
        if not os.environ.get("NZ_FLOW_NMAX"):   # a flow stage in several launches also sends its state planes
            assert sent == (stripes - 1) * cols * 4 * (2 * (10 + 8 + 8 + 8) + 2 * 10 + 5)
    rng_want = oracle.get_map_range(want)
    assert r["range"].view(np.uint32).tolist() == rng_want.view(np.uint32).tolist()
    assert np.array_equal(r["norm"], oracle.normalize_args(want, rng_want))


@pytest.mark.gpu
@pytest.mark.timeout(400, method="thread")
def test_flow_state_planes_travel_between_launches(nj, ctx, oracle, tmp_path):
    # 12 flow iterations = three launches: the five state planes are exchanged before the second and the third
    grows = cols = 256
    pkw = dict(octaves=6, noiseSize=200, gaussIterations=3, flowIterations=12, erosionIterations=9, haloMode="exchange")
    want = oracle.pipeline(grows, cols, octaves=6, noise_size=200, gauss_iterations=3, flow_iterations=12,
                           erosion_iterations=9)
    for overlap in (0, 2):
        r = _in_child(tmp_path, "grid", (grows, cols, pkw, 2, overlap))
        assert np.array_equal(r["grid"], want), overlap


@pytest.mark.gpu
def test_sharded_grid_without_rccl_uses_device_copies(nj, ctx, oracle):
    # comm None: one rank, the same plan, ghost rows by device copies on the context's stream (a host without RCCL)
    from noize_job_amd import sharded as sh
    grows, cols = 333, 200
    pkw = dict(octaves=8, noiseSize=300, gaussIterations=5, flowIterations=3, erosionIterations=7)
    want = oracle.pipeline(grows, cols, octaves=8, noise_size=300, gauss_iterations=5, flow_iterations=3, erosion_iterations=7)
    for mode, stripes in (("exchange", 3), ("exchange_once", 3), ("recompute", 5)):
        g = sh.ShardedGrid(ctx, None, grows, cols, sh.PipelineParams(haloMode=mode, **pkw), stripes=stripes)
        h, marks = g.run(marks=True)
        h.Complete()
        assert all(ctx.elapsed_ms(marks[i], marks[i + 1]) >= 0.0 for i in range(4))
        got = np.concatenate([g.owned_rows(i)[1] for i in range(stripes)], axis=0)
        g.close()
        assert np.array_equal(got, want), mode


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(36))
def test_random_stripe_geometry_equals_the_oracle(nj, ctx, oracle, seed):
    # seeded sweep over what the plan is built from: ragged grids (rows not a multiple of the stripes, columns not a multiple
    # of 4), the noise bases, 3...9-tap filters, iteration counts that need several launches, the three halo modes and the
    # three schedules (no split / interior first / border first) -- one rank, ghost rows by device copies
    from noize_job_amd import sharded as sh
    rng = np.random.default_rng(7700 + seed)
    grows, cols = int(rng.integers(150, 460)), int(rng.integers(33, 420))
    mode = ("exchange", "exchange_once", "recompute")[seed % 3]
    overlap = int(rng.integers(0, 3))
    filt = int(rng.choice([0, 1, 2, 3, 4, 5, 6, 7, 8]))
    taps = {0: 9, 1: 7, 2: 5, 3: 3, 4: 9, 5: 7, 6: 5, 7: 3, 8: 3}[filt]
    # every basis but Sin (0), which calls the device's sinf and is compared with a tolerance elsewhere (test_gpu_parity.py)
    kw = dict(noiseType=int(rng.integers(1, 8)), octaves=int(rng.integers(1, 7)), noiseSize=int(rng.integers(50, 900)),
              xpos=int(rng.integers(-3000, 3000)), zpos=int(rng.integers(-3000, 3000)), filter=filt,
              gaussIterations=int(rng.integers(0, 9)), flowIterations=int(rng.integers(0, 8)),
              erosionIterations=int(rng.integers(0, 8)))
    halo = kw["gaussIterations"] * (taps // 2) + 2 * kw["flowIterations"] + kw["erosionIterations"]
    stripes = int(rng.integers(2, 7))
    while stripes > 1 and mode != "recompute" and grows // stripes < halo:
        stripes -= 1  # an exchange takes ghost rows from the adjacent stripe only
    want = oracle.pipeline(grows, cols, noise_type=kw["noiseType"], octaves=kw["octaves"], noise_size=kw["noiseSize"],
                           xpos=kw["xpos"], zpos=kw["zpos"], filter_type=filt, gauss_iterations=kw["gaussIterations"],
                           flow_iterations=kw["flowIterations"], erosion_iterations=kw["erosionIterations"])
    g = sh.ShardedGrid(ctx, None, grows, cols, sh.PipelineParams(haloMode=mode, **kw), stripes=stripes, overlap=overlap)
    try:
        for _ in range(2):  # the second step runs on the planes the first one left behind
            g.run().Complete()
        got = np.concatenate([g.owned_rows(i)[1] for i in range(stripes)], axis=0)
    finally:
        g.close()
    assert np.array_equal(got, want, equal_nan=True), (grows, cols, stripes, mode, overlap, kw)


@pytest.mark.gpu
def test_external_source_plane_is_exchanged_not_recomputed(nj, ctx, oracle):
    # an uploaded height map: no noise stage, the ghost rows of the source travel (exchange / exchange_once)
    from noize_job_amd import sharded as sh
    import ctypes as C
    grows, cols, stripes = 300, 160, 3
    rng = np.random.default_rng(5)
    src = rng.random((grows, cols), dtype=np.float32)
    want = oracle.erosion_min(oracle.flowmap(oracle.kernel_filter(src, oracle.GAUSS5_S1, 6), 4, 0.0, 0.005), 3)
    for mode in ("exchange", "exchange_once"):
        p = sh.PipelineParams(gaussIterations=6, flowIterations=4, erosionIterations=3, haloMode=mode)
        g = sh.ShardedGrid(ctx, None, grows, cols, p, stripes=stripes, external_source=True)
        for i in range(stripes):
            st, source, _ = g.stripe(i)
            g0, n = st.grow0 + st.own0, st.own1 - st.own0
            rows = np.ascontiguousarray(src[g0:g0 + n])
            nj._native.check(nj._native.lib.nz_tile_upload(ctx._h, source + st.own0 * cols * 4, rows.ctypes.data, rows.size,
                                                           0, None), "upload")
            ctx.synchronize()
        g.run().Complete()
        got = np.concatenate([g.owned_rows(i)[1] for i in range(stripes)], axis=0)
        g.close()
        assert np.array_equal(got, want), mode
    with pytest.raises(nj.NoizeError):
        sh.ShardedGrid(ctx, None, grows, cols, sh.PipelineParams(haloMode="recompute"), stripes=stripes, external_source=True)


@pytest.mark.gpu
@pytest.mark.timeout(400, method="thread")
@pytest.mark.parametrize("mode", ["exchange", "recompute"])
def test_interior_rank_rehearsal_runs_and_counts_its_traffic(tmp_path, mode):
    # rank 3 of 8 of a 4096 x 1024 grid on one GPU, neighbours played by the rank itself: timing only
    r = _in_child(tmp_path, "rehearsal", (4096, 1024, mode, 8))
    exchanges, sent = r["traffic"]
    if mode == "recompute":
        assert exchanges == 0
    else:
        from noize_job_amd import sharded as sh
        assert exchanges == _exchanges_planned(sh, 4096, 1024, sh.PipelineParams(haloMode="exchange"), 8, as_rank=(3, 8))
        assert exchanges == 6 or not _default_fusion()
        if not os.environ.get("NZ_FLOW_NMAX"):
            assert sent == 1024 * 4 * (2 * (10 + 8 + 8 + 8) + 2 * 10 + 5)
        assert float(r["exchange_ms"][0]) >= 0.0


@pytest.mark.gpu
@pytest.mark.timeout(400, method="thread")
def test_cpp_host_mirror_runs_the_sharded_grid(oracle, tmp_path):
    # noize_pipeline.hpp: Comm + ShardedPipeline from a compiled host with no Python and no torch in the process: RCCL is
    # the system's (dlopen by libnoize_hip.so), three stripes on one rank, ghost rows through ncclSend / ncclRecv
    import subprocess
    exe = os.path.join(ROOT, "noize_job_amd", "host", "host_demo")
    assert os.path.exists(exe), "host_demo not built (run __graft_entry__.build())"
    out = str(tmp_path / "grid.f32")
    want = oracle.pipeline(384, 384, xpos=100, zpos=900)
    for stripes, mode, overlap in ((3, 1, 0), (3, 1, 2), (2, 2, 0), (4, 0, 0)):
        subprocess.run([exe, "384", out, "sharded", str(stripes), str(mode), str(overlap)], check=True, timeout=300)
        assert np.array_equal(np.fromfile(out, dtype=np.float32).reshape(384, 384), want), (stripes, mode, overlap)


@pytest.mark.gpu
@pytest.mark.timeout(400, method="thread")
def test_cpp_host_mirror_one_process_per_gpu(nj, oracle, tmp_path):
    import subprocess
    world = min(nj.Context.device_count(), 4)
    if world < 2:
        pytest.skip("needs at least two GPUs (RCCL refuses two ranks on one device)")
    exe = os.path.join(ROOT, "noize_job_amd", "host", "host_demo")
    res, out, idfile = 256 * world, str(tmp_path / "grid.f32"), str(tmp_path / "rccl.id")
    procs = [subprocess.Popen([exe, str(res), out, "sharded-rank", str(r), str(world), idfile]) for r in range(world)]
    assert all(pr.wait(300) == 0 for pr in procs)
    got = np.concatenate([np.fromfile("%s.%d" % (out, r), dtype=np.float32).reshape(-1, res) for r in range(world)], axis=0)
    assert np.array_equal(got, oracle.pipeline(res, res, xpos=100, zpos=900))


def _two_gpu_worker(rank, world, idfile, grows, cols, pkw, out_dir):
    import sys
    import time
    sys.path.insert(0, ROOT)
    import noize_job_amd as nj
    from noize_job_amd import sharded as sh
    ctx = nj.Context(rank)
    if rank == 0:
        uid = sh.NativeComm.unique_id()
        with open(idfile + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(idfile + ".tmp", idfile)
    else:
        while not os.path.exists(idfile):
            time.sleep(0.05)
        uid = open(idfile, "rb").read()
    comm = sh.NativeComm(ctx, uid, rank, world)
    g = sh.ShardedGrid(ctx, comm, grows, cols, sh.PipelineParams(**pkw))
    g.run()
    g0, rows = g.owned_rows(0)
    rng = ctx.alloc(3)
    g.map_range(rng.ptr)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), g0=g0, rows=rows, range=rng.ToArray())
    g.close()
    comm.close()
    ctx.close()


@pytest.mark.gpu
@pytest.mark.timeout(400, method="thread")
@pytest.mark.parametrize("mode", ["exchange", "exchange_once", "recompute"])
def test_native_sharded_on_two_or_more_gpus(nj, oracle, tmp_path, mode):
    world = min(nj.Context.device_count(), 4)
    if world < 2:
        pytest.skip("needs at least two GPUs (RCCL refuses two ranks on one device)")
    grows, cols = 128 * world, 200
    pkw = dict(octaves=8, noiseSize=300, xpos=100, zpos=900, haloMode=mode)
    idfile = str(tmp_path / "rccl.id")
    procs = [mp.get_context("spawn").Process(target=_two_gpu_worker, args=(r, world, idfile, grows, cols, pkw, str(tmp_path)))
             for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(300)
    alive = [pr for pr in procs if pr.is_alive()]
    for pr in alive:
        pr.kill()
    assert not alive and all(pr.exitcode == 0 for pr in procs)
    parts = [np.load(str(tmp_path / ("rank%d.npz" % r))) for r in range(world)]
    got = np.concatenate([pt["rows"] for pt in parts], axis=0)
    want = oracle.pipeline(grows, cols, octaves=8, noise_size=300, xpos=100, zpos=900)
    assert np.array_equal(got, want)
    for pt in parts:
        assert pt["range"].view(np.uint32).tolist() == oracle.get_map_range(want).view(np.uint32).tolist()
