"""Row-stripe sharding over real GPUs: one process per GPU, torch.distributed backend `nccl` (= RCCL over xGMI),
the C-ABI stripe entry points on every rank.  Needs at least two visible GPUs; skipped on a one-GPU box (the same
schedule is covered there by tests/test_sharded_cpu.py over `gloo` and tests/test_gpu_sharded.py in one process)."""
import os
import socket

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawn(fn, world, make_args):
    """mp.spawn on a rendezvous port probed as free; the port can be gone by the time rank 0 binds it (another process of
    the box, a socket still closing): that one failure is retried on another port."""
    import torch.multiprocessing as mp
    for attempt in range(3):
        try:
            mp.spawn(fn, args=make_args(_free_port()), nprocs=world, join=True)
            return
        except Exception as e:  # ProcessRaisedException carries the rank's traceback as text
            if attempt == 2 or not any(t in str(e) for t in ("EADDRINUSE", "address already in use", "Address already in use")):
                raise


def _worker(rank, world, port, grows, cols, pkw, out_path):
    import sys
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import noize_job_amd as nj
    from noize_job_amd import sharded as sh
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world,
                            device_id=torch.device("cuda", rank))
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx = nj.Context(rank, stream=stream.cuda_stream)
    ops = sh.HipStripeOps(ctx)
    p = sh.PipelineParams(**pkw)
    plan = sh.StripePlan(rank, world, grows, cols, sh.halo_rows_needed(ops, p), neighbours_own_halo=p.haloMode != "recompute")
    nan = float("nan")
    bufs = (torch.full((plan.rows, cols), nan, device="cuda"), torch.full((plan.rows, cols), nan, device="cuda"),
            torch.full((5, plan.rows, cols), nan, device="cuda"), torch.full((5, plan.rows, cols), nan, device="cuda"))
    comm = sh.NoComm() if p.haloMode == "recompute" else sh.TorchComm(dist)  # asynchronous: interior rows overlap the P2P batch
    res = sh.run_pipeline(ops, comm, plan, p, bufs)
    torch.cuda.synchronize()
    mine = res[plan.own0:plan.own1].contiguous()
    parts = [torch.empty_like(mine) for _ in range(world)] if plan.nown * world == grows else None
    assert parts is not None, "the test uses grids that split evenly"
    dist.all_gather(parts, mine)
    if rank == 0:
        np.save(out_path, torch.cat(parts, 0).cpu().numpy())
    # the path's one collective: per-rank GetMapRangeJob -> all-gather over RCCL -> fold in rank order -> normalise
    rng_res = torch.empty(3, device="cuda")
    work = torch.empty(sh.map_range_work_floats(world), device="cuda")
    sh.global_map_range(ops, dist, res, plan, rng_res, work)
    ops.normalize_args(res, plan, rng_res)
    torch.cuda.synchronize()
    dist.all_gather(parts, res[plan.own0:plan.own1].contiguous())
    if rank == 0:
        np.save(out_path + ".range.npy", rng_res.cpu().numpy())
        np.save(out_path + ".norm.npy", torch.cat(parts, 0).cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


@pytest.mark.parametrize("mode", ["exchange", "exchange_once", "recompute"])
def test_nccl_sharded_equals_monolithic(oracle, tmp_path, mode):
    import torch
    import torch.multiprocessing as mp
    world = min(torch.cuda.device_count(), 4)
    if world < 2:
        pytest.skip("needs at least two GPUs (RCCL refuses two ranks on one device)")
    grows, cols = 128 * world, 200
    pkw = dict(octaves=8, noiseSize=300, gaussIterations=17, flowIterations=5, erosionIterations=5, xpos=100, zpos=900,
               haloMode=mode)
    out = str(tmp_path / "sharded.npy")
    _spawn(_worker, world, lambda port: (world, port, grows, cols, pkw, out))
    want = oracle.pipeline(grows, cols, octaves=8, noise_size=300, xpos=100, zpos=900)
    assert np.array_equal(np.load(out), want)
    rng_want = oracle.get_map_range(want)
    assert np.load(out + ".range.npy").view(np.uint32).tolist() == rng_want.view(np.uint32).tolist()
    assert np.array_equal(np.load(out + ".norm.npy"), oracle.normalize_args(want, rng_want))


def _single_rank_worker(rank, port, grows, cols, pkw, out_path):
    """ONE process, ONE GPU, backend nccl: the process group, the stream it shares with the C ABI's context and
    batch_isend_irecv on row slices of device planes are all real; the peer of every transfer is the rank itself (RCCL
    runs a send and the matching receive of one group on the same device).  The grid is cut into `nstripes` stripes
    that live in this process and are driven in lockstep; their ghost rows travel through RCCL."""
    import sys
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import noize_job_amd as nj
    from noize_job_amd import sharded as sh
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx = nj.Context(0, stream=stream.cuda_stream)
    ops = sh.HipStripeOps(ctx)
    p = sh.PipelineParams(**pkw)
    nstripes = 3
    halo = sh.halo_rows_needed(ops, p)
    plans = [sh.StripePlan(r, nstripes, grows, cols, halo) for r in range(nstripes)]
    nan = float("nan")
    bufs = [(torch.full((pl.rows, cols), nan, device="cuda"), torch.full((pl.rows, cols), nan, device="cuda"),
             torch.full((5, pl.rows, cols), nan, device="cuda"), torch.full((5, pl.rows, cols), nan, device="cuda"))
            for pl in plans]
    moved = [0]

    def copy_rows(dst, d0, src, s0, n):  # ghost rows <- the neighbour stripe's owned rows, through RCCL
        reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend, src[s0:s0 + n], 0), dist.P2POp(dist.irecv, dst[d0:d0 + n], 0)])
        for r in reqs:
            r.wait()   # the current stream (the context's) waits; the host does not
        moved[0] += n * cols * 4

    res = sh.run_pipeline_lockstep([ops] * nstripes, plans, p, bufs, copy_rows)
    torch.cuda.synchronize()
    full = torch.cat([r[pl.own0:pl.own1] for r, pl in zip(res, plans)], 0)
    t = torch.tensor([float(moved[0])], device="cuda")
    dist.all_reduce(t)   # a collective on the same group for good measure
    np.save(out_path, full.cpu().numpy())
    assert t.item() == moved[0] and moved[0] > 0
    # the path's one collective: GetMapRangeJob -> all-gather over RCCL -> fold -> normalise with device args
    whole = sh.StripePlan(0, 1, grows, cols, 0)
    plane = full.clone()
    rng_res = torch.empty(3, device="cuda")
    work = torch.empty(sh.map_range_work_floats(1), device="cuda")
    sh.global_map_range(ops, dist, plane, whole, rng_res, work)
    ops.normalize_args(plane, whole, rng_res)
    torch.cuda.synchronize()
    np.save(out_path + ".range.npy", rng_res.cpu().numpy())
    np.save(out_path + ".norm.npy", plane.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


@pytest.mark.timeout(400, method="thread")
def test_nccl_single_rank_halo_exchange_through_rccl(oracle, tmp_path):
    # what a one-GPU box can exercise of the RCCL path: init_process_group("nccl"), the context on torch's stream,
    # batch_isend_irecv on row slices of device tensors between stripe kernels -- in a child process with a hard limit,
    # so that a transfer that never completes fails the test instead of hanging the run
    import multiprocessing as mp
    grows, cols = 384, 256
    pkw = dict(octaves=8, noiseSize=300, gaussIterations=17, flowIterations=5, erosionIterations=5, xpos=100, zpos=900,
               haloMode="exchange")
    out = str(tmp_path / "single.npy")
    proc = mp.get_context("spawn").Process(target=_single_rank_worker, args=(0, _free_port(), grows, cols, pkw, out))
    proc.start()
    proc.join(300)
    if proc.is_alive():
        proc.kill()
        proc.join()
        pytest.fail("the single-rank nccl worker did not finish within 300 s")
    assert proc.exitcode == 0
    want = oracle.pipeline(grows, cols, octaves=8, noise_size=300, xpos=100, zpos=900)
    assert np.array_equal(np.load(out), want)
    rng_want = oracle.get_map_range(want)
    assert np.load(out + ".range.npy").view(np.uint32).tolist() == rng_want.view(np.uint32).tolist()
    assert np.array_equal(np.load(out + ".norm.npy"), oracle.normalize_args(want, rng_want))
