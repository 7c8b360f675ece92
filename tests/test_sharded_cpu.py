"""N > 1 path on CPU: world_size-2 and -3 `gloo` process groups run the row-stripe schedule of
noize_job_amd.sharded (partition, ghost-row widths, neighbour exchange order) with the oracle as
compute back end; the gathered stripes must equal the monolithic oracle run bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawn(fn, world, make_args):
    """mp.spawn on a rendezvous port probed as free; the port can be gone by the time rank 0 binds it (another process of
    the box, a socket still closing): that one failure is retried on another port."""
    for attempt in range(3):
        try:
            mp.spawn(fn, args=make_args(_free_port()), nprocs=world, join=True)
            return
        except Exception as e:  # ProcessRaisedException carries the rank's traceback as text
            if attempt == 2 or not any(t in str(e) for t in ("EADDRINUSE", "address already in use", "Address already in use")):
                raise


def _worker(rank, world, port, grows, cols, pkw, out_path, overlap=True):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O
    from noize_job_amd import sharded as sh
    from oracle_stripe_ops import OracleStripeOps
    O.set_threads(2)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    ops = OracleStripeOps()
    p = sh.PipelineParams(**pkw)
    plan = sh.StripePlan(rank, world, grows, cols, sh.halo_rows_needed(ops, p),
                         neighbours_own_halo=p.haloMode != "recompute")
    bufs = (torch.full((plan.rows, cols), float("nan")), torch.full((plan.rows, cols), float("nan")),
            torch.full((5, plan.rows, cols), float("nan")), torch.full((5, plan.rows, cols), float("nan")))
    # "recompute" never communicates; overlap: the launch that needs the ghost rows is split into its interior rows
    # (enqueued while the halos travel) and its border rows (after finish())
    comm = sh.NoComm() if p.haloMode == "recompute" else sh.TorchComm(dist, overlap=overlap)
    res = sh.run_pipeline(ops, comm, plan, p, bufs)
    mine = res[plan.own0:plan.own1].contiguous()
    parts = [None] * world
    dist.all_gather_object(parts, (plan.g0, mine.numpy()))
    if rank == 0:
        full = np.concatenate([a for _, a in sorted(parts, key=lambda t: t[0])], axis=0)
        np.save(out_path, full)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,grows,cols,mode", [(2, 64, 48, "exchange"), (3, 70, 33, "exchange"),
                                                    (2, 64, 48, "exchange_blocking"), (3, 70, 33, "exchange_blocking"),
                                                    (2, 64, 48, "recompute"), (3, 70, 33, "recompute"),
                                                    (2, 64, 48, "exchange_once")])
def test_sharded_schedule_equals_monolithic(oracle, tmp_path, world, grows, cols, mode):
    overlap = mode != "exchange_blocking"
    pkw = dict(octaves=6, noiseSize=40, gaussIterations=7, flowIterations=3, erosionIterations=4, xpos=11, zpos=5,
               haloMode="exchange" if mode == "exchange_blocking" else mode)
    out = str(tmp_path / "sharded.npy")
    _spawn(_worker, world, lambda port: (world, port, grows, cols, pkw, out, overlap))
    got = np.load(out)
    want = oracle.pipeline(grows, cols, octaves=6, noise_size=40, gauss_iterations=7, flow_iterations=3,
                           erosion_iterations=4, xpos=11, zpos=5)
    assert got.shape == want.shape
    assert np.array_equal(got, want)


def _range_worker(rank, world, port, grows, cols, case, out_path):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from noize_job_amd import sharded as sh
    from oracle_stripe_ops import OracleStripeOps
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    ops = OracleStripeOps()
    plan = sh.StripePlan(rank, world, grows, cols, 2)
    full = torch.from_numpy(_range_case(case, grows, cols))
    plane = torch.full((plan.rows, cols), float("nan"))
    plane[plan.own0:plan.own1] = full[plan.g0:plan.g0 + plan.nown]
    res, work = torch.empty(3), torch.empty(sh.map_range_work_floats(world))
    sh.global_map_range(ops, dist, plane, plan, res, work, *_RANGE_LIMS[case])
    ops.normalize_args(plane, plan, res)
    parts = [None] * world
    dist.all_gather_object(parts, (plan.g0, res.numpy().copy(), plane[plan.own0:plan.own1].numpy().copy()))
    if rank == 0:
        parts.sort(key=lambda t: t[0])
        np.savez(out_path, res=np.stack([r for _, r, _ in parts]), full=np.concatenate([a for _, _, a in parts], axis=0))
    dist.barrier()
    dist.destroy_process_group()


_RANGE_LIMS = {"plain": (np.inf, -np.inf), "zeros": (np.inf, -np.inf), "nan_rank": (np.inf, -np.inf), "limits": (0.25, 0.5)}


def _range_case(case, grows, cols):
    rng = np.random.default_rng(len(case))
    a = rng.random((grows, cols), dtype=np.float32)
    if case == "zeros":       # the minimum is zero: the sign of the LAST zero of the whole grid stays
        a[3, 5] = -0.0; a[grows // 2 + 1, 2] = 0.0; a[grows - 2, 7] = -0.0
    if case == "nan_rank":    # one rank holds nothing but NaN
        a[:grows // 3] = np.nan
    return a


@pytest.mark.parametrize("world,case", [(2, "plain"), (3, "zeros"), (3, "nan_rank"), (2, "limits")])
def test_global_map_range_all_gather_equals_monolithic(oracle, tmp_path, world, case):
    # the one collective of the path: GetMapRangeJob per rank, all-gather, the same fold in rank order; every rank
    # ends with the monolithic {min, max, range} bit for bit and normalises its rows with it
    grows, cols = 30, 17
    out = str(tmp_path / "range.npz")
    _spawn(_range_worker, world, lambda port: (world, port, grows, cols, case, out))
    got = np.load(out)
    a = _range_case(case, grows, cols)
    want = oracle.get_map_range(a, *_RANGE_LIMS[case])
    for r in got["res"]:
        assert r.view(np.uint32).tolist() == want.view(np.uint32).tolist(), (r, want)
    assert np.array_equal(got["full"], oracle.normalize_args(a, want), equal_nan=True)


def test_overlapped_exchange_splits_a_launch_into_interior_and_border_rows():
    # with an asynchronous comm the launch that needs ghost rows runs rows [own0 + up, own1 - down) first -- they read
    # no ghost row -- and the border rows after finish(); a window thinner than its halos runs whole, after finish()
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from noize_job_amd import sharded as sh
    from oracle_stripe_ops import OracleStripeOps
    log = []

    class Rec(OracleStripeOps):
        def fractal(self, buf, pl, p):
            log.append(("noise", pl.own0, pl.own1))

        def kernel_filter(self, src, dst, pl, f, T):
            log.append(("gauss", pl.own0, pl.own1))

        def flow_fused(self, h, si, so, dst, pl, *a):
            log.append(("flow", pl.own0, pl.own1))

        def erosion(self, src, dst, pl, E):
            log.append(("erosion", pl.own0, pl.own1))

    class Comm:
        overlap = True

        def begin(self, planes, plan, up, down):
            log.append(("begin", up, down))
            return "req"

        def finish(self, reqs):
            assert reqs == "req"
            log.append(("finish",))

    p = sh.PipelineParams(gaussIterations=3, flowIterations=0, erosionIterations=2, haloMode="exchange")
    ops = Rec()
    plan = sh.StripePlan(1, 3, 90, 8, sh.halo_rows_needed(ops, p))
    sh.run_pipeline(ops, Comm(), plan, p, (0, 1, [0] * 5, [1] * 5))
    o0, o1 = plan.own0, plan.own1
    assert log == [("noise", o0, o1),
                   ("begin", 6, 6), ("gauss", o0 + 6, o1 - 6), ("finish",), ("gauss", o0, o0 + 6), ("gauss", o1 - 6, o1),
                   ("begin", 2, 0), ("erosion", o0 + 2, o1), ("finish",), ("erosion", o0, o0 + 2)]
    log.clear()
    thin = sh.StripePlan(1, 9, 90, 8, sh.halo_rows_needed(ops, p))     # 10 owned rows < 6 + 6
    sh.run_pipeline(ops, Comm(), thin, p, (0, 1, [0] * 5, [1] * 5))
    assert log[1:4] == [("begin", 6, 6), ("finish",), ("gauss", thin.own0, thin.own1)]


def test_stripe_plan_partitions_rows():
    from noize_job_amd.sharded import StripePlan, split_iterations
    for world, grows in ((1, 10), (2, 64), (3, 70), (8, 16384)):
        plans = [StripePlan(r, world, grows, 16, 4) for r in range(world)]
        assert plans[0].g0 == 0 and plans[-1].g0 + plans[-1].nown == grows
        for a, b in zip(plans, plans[1:]):
            assert a.g0 + a.nown == b.g0
        assert plans[0].up is None and plans[-1].down is None
        for pl in plans:
            st = pl.stripe()
            assert (st.rows, st.own0, st.own1, st.grow0) == (pl.nown + 8, 4, 4 + pl.nown, pl.g0 - 4)
    assert split_iterations(17, 3) == [3, 3, 3, 3, 3, 2] and split_iterations(5, 16) == [5]
    assert sum(split_iterations(17, 4)) == 17 and max(split_iterations(17, 4)) <= 4


def test_recompute_mode_needs_the_whole_pipeline_radius():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from noize_job_amd import sharded as sh
    from oracle_stripe_ops import OracleStripeOps
    ops = OracleStripeOps()
    p = sh.PipelineParams(haloMode="recompute")            # Gauss5 x17, flow x5, erosion x5
    assert sh.halo_rows_needed(ops, p) == 17 * 2 + 5 * 2 + 5
    assert sh.halo_rows_needed(ops, sh.PipelineParams()) == max(3 * 2, 2 * 3, 5)  # oracle ops: flow fused <= 3
    plan = sh.StripePlan(3, 8, 16384, 64, 49)
    win = plan.widened(49, 44)
    assert (win.own0, win.own1, win.g0, win.nown) == (0, 49 + 2048 + 44, 3 * 2048 - 49, 2048 + 93)
    top = sh.StripePlan(0, 8, 16384, 64, 49).widened(49, 44)   # clipped at the global border
    assert (top.own0, top.g0, top.nown) == (49, 0, 2048 + 44)
    # every launch of the schedule shrinks the window; the last one produces exactly the owned rows
    calls = []

    class Rec(OracleStripeOps):
        def fractal(self, buf, pl, p):
            calls.append((pl.own0, pl.own1))

        def kernel_filter(self, src, dst, pl, f, T):
            calls.append((pl.own0, pl.own1))

        def flow_fused(self, h, si, so, dst, pl, *a):
            calls.append((pl.own0, pl.own1))

        def erosion(self, src, dst, pl, E):
            calls.append((pl.own0, pl.own1))

    res = []
    steps = list(sh.pipeline_steps(Rec(), plan, p, (0, 1, [0] * 5, [1] * 5), res))
    assert steps == []                                         # nothing to exchange
    assert calls[0] == (0, 49 + 2048 + 44) and calls[-1] == (plan.own0, plan.own1)
    assert all(a[0] <= b[0] and a[1] >= b[1] for a, b in zip(calls, calls[1:]))


@pytest.mark.parametrize("mode", ["exchange", "recompute", "exchange_once"])
def test_lockstep_driver_equals_monolithic(oracle, mode):
    # the same schedule with all ranks in one process (the driver the single-GPU rehearsal uses)
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from noize_job_amd import sharded as sh
    from oracle_stripe_ops import OracleStripeOps
    world, grows, cols = 4, 96, 40
    p = sh.PipelineParams(octaves=5, noiseSize=30, gaussIterations=5, flowIterations=4, erosionIterations=3,
                          haloMode=mode)
    ops = OracleStripeOps()
    halo = sh.halo_rows_needed(ops, p)
    plans = [sh.StripePlan(r, world, grows, cols, halo, neighbours_own_halo=mode != "recompute")
             for r in range(world)]
    bufs = [(torch.full((pl.rows, cols), float("nan")), torch.full((pl.rows, cols), float("nan")),
             torch.full((5, pl.rows, cols), float("nan")), torch.full((5, pl.rows, cols), float("nan")))
            for pl in plans]

    def copy_rows(dst, d0, src, s0, n):
        dst[d0:d0 + n] = src[s0:s0 + n]

    res = sh.run_pipeline_lockstep([ops] * world, plans, p, bufs, copy_rows)
    got = np.concatenate([r[pl.own0:pl.own1].numpy() for r, pl in zip(res, plans)], axis=0)
    want = oracle.pipeline(grows, cols, octaves=5, noise_size=30, gauss_iterations=5, flow_iterations=4,
                           erosion_iterations=3)
    assert np.array_equal(got, want)
