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
    comm = sh.NoComm() if p.haloMode == "recompute" else sh.TorchComm(dist)
    res = sh.run_pipeline(ops, comm, plan, p, bufs)
    torch.cuda.synchronize()
    mine = res[plan.own0:plan.own1].contiguous()
    parts = [torch.empty_like(mine) for _ in range(world)] if plan.nown * world == grows else None
    assert parts is not None, "the test uses grids that split evenly"
    dist.all_gather(parts, mine)
    if rank == 0:
        np.save(out_path, torch.cat(parts, 0).cpu().numpy())
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
    mp.spawn(_worker, args=(world, _free_port(), grows, cols, pkw, out), nprocs=world, join=True)
    want = oracle.pipeline(grows, cols, octaves=8, noise_size=300, xpos=100, zpos=900)
    assert np.array_equal(np.load(out), want)
