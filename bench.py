#!/usr/bin/env python3
"""bench.py -- the BASELINE.json metric on MI355X.

A step = one pass of the metric pipeline (simplex fBm 13 oct -> Gauss5 x17 -> FlowMap x5 -> value
erosion x5) over one device-resident grid; nothing crosses PCIe inside the timed region.
  N = 1 : the 4096^2 tile BASELINE.json's metric is quoted on.
  N > 1 : one process per GPU (torch.distributed, backend nccl = RCCL); the grid is row-stripe
          sharded, 2048 x 16384 cells per rank (N = 8 is BASELINE config 5, 16384^2) -> "scaling":
          "weak".  The stripes are independent: the source is closed-form noise, so every rank
          recomputes the 49 ghost rows per side the stencils consume (--halo recompute, default; no
          data-path collective).  --halo exchange swaps ghost rows with the neighbour ranks over
          RCCL before every stencil launch instead (the form an uploaded height map would need).
Rank 0 prints ONE JSON line.  `roofline` describes the stage that takes the most GPU time, `stages`
every stage; both come from HIP events recorded on the kernels' stream inside the timed steps.
`cpu_baseline` is the CPU oracle (reference-shaped restatement of the Burst jobs) timed on this
box's host cores on one full 4096^2 pass.
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
VALU_PEAK_GOPS = 78643.2     # 256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz, non-FMA fp32 ops

# algorithmic bytes per cell: fused minimum, one read + one write per plane per application
G_IT, F_IT, E_IT = 17, 5, 5
BYTES = {"noise": 4.0, "gauss": 8.0 * G_IT, "flow": 24.0 + 44.0 * (F_IT - 1) + 20.0, "erosion": 8.0 * E_IT}
KERNEL_OF = {"noise": "fractal_simplex_tab_kernel<2>", "gauss": "conv_reg_kernel<5, true>",
             "flow": "flow_fused_kernel<true, true, 4>", "erosion": "erosion_reg_kernel<3>"}
NOISE_OPS_PER_OCTAVE_CELL = 83.0  # VALU instructions of the table-driven simplex octave (ISA count: 166 per 2 cells)


CPU_PASSES = 11         # ~10 s of host work on the GPU box's 32 cores (0.9-1.0 s per 4096^2 pass)
PREHEAT_MIN_STEPS = 50  # untimed passes before the timed region, warm-up included (clock settling)
MAX_MARKED_STEPS = 200  # per-stage markers are kept for the last steps only (the handle ring holds 4096)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # the GPU's clocks keep ramping for some tens of ms of continuous work (0.96 ms/step over 5 steps,
    # 0.89 over 20, 0.85 over 100 and beyond, tools/probe_host_enqueue.py): the defaults time steady state
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--res", type=int, default=4096, help="tile resolution at N=1")
    ap.add_argument("--stripe-rows", type=int, default=2048, help="rows per rank at N>1")
    ap.add_argument("--cols", type=int, default=16384, help="grid columns at N>1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sharded", action="store_true", help="run the row-stripe path even with one rank (rehearsal)")
    ap.add_argument("--halo", choices=("recompute", "exchange", "exchange_once"), default="recompute",
                    help="N>1: ghost rows recomputed from the closed-form noise (no data-path communication) or "
                         "exchanged with the neighbour ranks over RCCL before every launch")
    ap.add_argument("--as-rank", type=int, nargs=2, metavar=("R", "P"), default=None,
                    help="one-process rehearsal of rank R of a P-rank job (needs --halo recompute)")
    ap.add_argument("--flush", choices=("swap", "copy"), default="swap",
                    help="N=1: the tile is a READ / WRITE plane pair and TileHelpers.SWAP_RWTILE is a pointer swap "
                         "(nz_*_rw entries), or one plane with the in-place entries and their flush copies")
    ap.add_argument("--stripes", type=int, default=1,
                    help="N=1: run the tile as this many independent row stripes, each on its own HIP stream (ghost "
                         "rows recomputed from the closed-form noise; the fp32-bound kernels of one stripe overlap "
                         "the HBM-bound kernels of another); 1 = the stage pipeline on one stream")
    ap.add_argument("--cpu-res", type=int, default=4096)
    return ap.parse_args()


def pmc_traffic(kernel_key):
    """HBM bytes per launch from the newest committed PMC summary (profiles/*pmc*.json), or None."""
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc*.json"))):
        try:
            with open(path) as f:
                d = json.load(f)
            if kernel_key in d.get("kernels", {}):
                best = d["kernels"][kernel_key].get("hbm_bytes_per_launch")
        except (OSError, ValueError):
            pass
    return best


def two_tiles(nj, ctx, stages, gd, res, p, swap, steps=100):
    """Informational, outside the timed steps: two INDEPENDENT tiles, each with its own context (HIP stream), pipelines
    issued alternately.  The fp32-bound and the HBM-bound kernels of different tiles overlap; `value` above is the
    single tile BASELINE.json names."""
    ctx2 = nj.Context(ctx.device)
    cells = res * res
    stages2 = [nj.NoiseStage(ctx2, nj.FractalNoise.Simplex, p.hurst, p.startingAmplitude, p.octaves, p.stepdown,
                             p.detuneRate, p.noiseSize),
               nj.KernelFilterStage(ctx2, nj.KernelFilterType.Gauss5_S1, G_IT),
               nj.FlowMapStage(ctx2, F_IT, p.normMin, p.normMax), nj.ErosionStage(ctx2, E_IT)]
    gd2 = nj.GeneratorData("bench2", ctx2.alloc(cells), res, res, 0, write=ctx2.alloc(cells) if swap else None)
    h0 = nj.JobHandle()

    def both():
        for a, b in zip(stages, stages2):
            a.Schedule(nj.PipelineWorkItem(gd), h0)
            b.Schedule(nj.PipelineWorkItem(gd2), h0)
    for _ in range(20):
        both()
    ctx.synchronize(); ctx2.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        both()
    ctx.synchronize(); ctx2.synchronize()
    dt = (time.perf_counter() - t0) / (2 * steps)
    for st in stages2:
        st.OnDestroy()
    gd2.data.Dispose()
    if gd2.write is not None:
        gd2.write.Dispose()
    ctx2.close()
    return {"streams": 2, "ms_per_tile": round(dt * 1e3, 4), "Mcells/s": round(cells / dt / 1e6, 1),
            "note": "two independent 4096^2 tiles on two HIP streams, not the headline value"}


def two_stripes(nj, sh, torch, device, data, res, p, steps=100):
    """Informational, outside the timed steps: the SAME 4096^2 tile as two independent row stripes, each on its own
    HIP stream (ghost rows recomputed from the closed-form noise, the last launch storing into the tile's plane):
    what `--stripes 2` times as its headline."""
    ctxs = [nj.Context(device) for _ in range(2)]
    try:
        tile = sh.StripedTile(ctxs, data.data_ptr(), res, res, p,
                              lambda *shape: torch.empty(shape, dtype=torch.float32, device="cuda"))
        torch.cuda.synchronize()
        for _ in range(20):
            tile.run()
        tile.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            tile.run()
        tile.synchronize()
        dt = (time.perf_counter() - t0) / steps
    finally:
        for c in ctxs:
            c.close()
    return {"streams": 2, "ms_per_tile": round(dt * 1e3, 4), "Mcells/s": round(res * res / dt / 1e6, 1),
            "note": "one 4096^2 tile as two independent row stripes on two HIP streams (bench.py --stripes 2), "
                    "not the headline value"}


def cpu_baseline(res):
    import oracle as O
    O.lib()
    times = []
    for _ in range(CPU_PASSES):
        t0 = time.perf_counter()
        O.pipeline(res, res, O.SIMPLEX, 0.4, 1.0, 2.0, 0.0, 13, 0, 0, 1700, O.GAUSS5_S1, G_IT, F_IT, 0.0, 0.005, E_IT)
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]
    return {"value": round(res * res / dt / 1e6, 2), "unit": "Mcells/s", "cores": O.get_threads(), "kind": "port",
            "sample": "median of %d passes of the full metric pipeline on a %dx%d tile (%.2f s per pass), OpenMP "
                      "row-parallel passes with the reference's serial flush copies" % (CPU_PASSES, res, res, dt)}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    import noize_job_amd as nj
    from noize_job_amd import sharded as sh

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs one process per GPU: launch with python -m torch.distributed.run "
                             "--nproc-per-node %d bench.py --gpus %d ..." % (args.gpus, args.gpus, args.gpus))
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # stdout carries the ONE JSON line only: whatever libraries print while they initialise (RCCL's version
    # banner under NCCL_DEBUG=VERSION) is sent to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs an MI355X: no HIP device is visible")
    local_rank %= ndev  # a launcher that narrows HIP_VISIBLE_DEVICES per rank leaves one device, index 0
    torch.cuda.set_device(local_rank)
    sharded = world > 1 or args.sharded or args.as_rank is not None
    if args.as_rank is not None and (world != 1 or args.halo != "recompute"):
        raise SystemExit("--as-rank is a single-process rehearsal of the communication-free schedule")
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx = nj.Context(local_rank, stream=stream.cuda_stream)

    p = sh.PipelineParams(gaussIterations=G_IT, flowIterations=F_IT, erosionIterations=E_IT, haloMode=args.halo)
    marks = []  # per step: handles at stage boundaries

    if not sharded:
        res = args.res
        cells = res * res
        data = torch.empty(cells, dtype=torch.float32, device="cuda")
        tile = ctx.wrap(data.data_ptr(), cells)
        swap = args.flush == "swap"
        data_w = torch.empty(cells, dtype=torch.float32, device="cuda") if swap else None
        stages = [nj.NoiseStage(ctx, nj.FractalNoise.Simplex, p.hurst, p.startingAmplitude, p.octaves, p.stepdown,
                                p.detuneRate, p.noiseSize),
                  nj.KernelFilterStage(ctx, nj.KernelFilterType.Gauss5_S1, G_IT),
                  nj.FlowMapStage(ctx, F_IT, p.normMin, p.normMax),
                  nj.ErosionStage(ctx, E_IT)]
        pipe = nj.BasePipeline(stages, "metric")
        gd = nj.GeneratorData("bench", tile, res, 0, 0, write=ctx.wrap(data_w.data_ptr(), cells) if swap else None)

        def step(record):
            if record:
                hs = [ctx.record()]
                for st in stages:  # same chain BasePipeline.Schedule builds, with a marker between stages
                    st.Schedule(nj.PipelineWorkItem(gd), hs[-1])
                    hs.append(st.jobHandle)
                marks.append(hs)
            else:
                pipe.Schedule(gd)
                pipe.pipelineRunning = False

        workload = "%dx%d tile: simplex-13oct(h0.4,size1700) -> Gauss5_S1 x%d -> FlowMap x%d (norm 0/0.005) -> " \
                   "ValueErosion x%d" % (res, res, G_IT, F_IT, E_IT)
        parallelism = "single tile"
        flush_note = ("READ/WRITE plane pair, SWAP_RWTILE = pointer swap (nz_*_rw entries)" if swap else
                      "one plane, in-place entries with flush copies")
    else:
        swap, flush_note = False, "stripe entries (explicit src / dst planes)"
        ops = sh.HipStripeOps(ctx)
        halo = sh.halo_rows_needed(ops, p)
        prank, pworld = args.as_rank if args.as_rank is not None else (rank, world)
        plan = sh.StripePlan(prank, pworld, args.stripe_rows * pworld, args.cols, halo,
                             neighbours_own_halo=args.halo != "recompute")
        cells = plan.nown * world * plan.cols  # every rank owns stripe_rows rows
        bufs = (torch.zeros(plan.rows, plan.cols, dtype=torch.float32, device="cuda"),
                torch.zeros(plan.rows, plan.cols, dtype=torch.float32, device="cuda"),
                torch.zeros(sh.FLOW_PLANES, plan.rows, plan.cols, dtype=torch.float32, device="cuda"),
                torch.zeros(sh.FLOW_PLANES, plan.rows, plan.cols, dtype=torch.float32, device="cuda"))
        comm = sh.NoComm() if args.halo == "recompute" else sh.TorchComm(dist)

        def step(record):
            if record:  # stream markers where the stages begin (exchanges of a stage are charged to it)
                hs = {}
                sh.run_pipeline(ops, comm, plan, p, bufs, on_stage=lambda name: hs.__setitem__(name, ctx.record()))
                marks.append([hs[n] for n in ("noise", "gauss", "flow", "erosion", "end")])
            else:
                sh.run_pipeline(ops, comm, plan, p, bufs)

        how = {"exchange": "ghost rows exchanged over RCCL before every launch",
               "exchange_once": "%d ghost rows per side of the source plane exchanged once over RCCL" % halo,
               "recompute": "%d ghost rows per side recomputed from the closed-form noise, no data-path "
                            "communication" % halo}[args.halo]
        workload = "%dx%d grid as %d row stripes of %dx%d (%s): simplex-13oct -> Gauss5_S1 x%d " \
                   "-> FlowMap x%d -> ValueErosion x%d" % (plan.grows, plan.cols, pworld, args.stripe_rows, plan.cols,
                                                            how, G_IT, F_IT, E_IT)
        parallelism = "row-stripe dp%d" % world
        if args.as_rank is not None:
            parallelism = "rehearsal of rank %d of %d on one GPU" % (prank, pworld)

    striped = None
    if not sharded and args.stripes > 1:
        # the same tile, same kernels, as independent stripes on their own streams; no per-stage markers here
        sctx = [nj.Context(local_rank) for _ in range(args.stripes)]
        striped = sh.StripedTile(sctx, data.data_ptr(), res, res, p,
                                 lambda *shape: torch.empty(shape, dtype=torch.float32, device="cuda"))
        torch.cuda.synchronize()

        def step(record):  # noqa: F811
            if record:  # stream markers of stripe 0 (its kernels share the chip with the other stripes')
                hs = {}
                striped.run(on_stage=lambda name: hs.__setitem__(name, sctx[0].record()))
                marks.append([hs[n] for n in ("noise", "gauss", "flow", "erosion", "end")])
            else:
                striped.run()
        parallelism = "single tile as %d row stripes on %d HIP streams" % (args.stripes, args.stripes)
        flush_note = "stripe entries (explicit src / dst planes), last launch stores into the tile's plane"

    def fence():
        if sharded:
            dist.barrier()
        if striped is not None:
            striped.synchronize()
        torch.cuda.synchronize()

    # The chip's clocks need some tens of ms of continuous work to settle (see --steps above).  A caller that asks
    # for a short run still gets the steady-state rate: the GPU is kept busy with untimed passes first, so that
    # warm-up + preheat cover at least PREHEAT_MIN_STEPS passes.  The K timed steps are exactly the K asked for.
    preheat = max(0, PREHEAT_MIN_STEPS - args.warmup)
    for _ in range(preheat):
        step(False)
    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.steps - i <= MAX_MARKED_STEPS)
    fence()
    dt = time.perf_counter() - t0
    if sharded:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    out = None
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = cells / (dt / args.steps) / 1e6
        total_bytes = sum(BYTES.values())
        out = {"metric": "Mcells/s 4096^2 simplex13oct->Gauss5x17->FlowMap->Erosion; %HBM roofline @1/8GPU",
               "value": round(value, 1), "unit": "Mcells/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": workload, "cells": cells, "parallelism": parallelism, "preheat_steps": preheat,
                          "algorithmic_bytes_per_cell": total_bytes, "flush": flush_note},
               "pipeline_hbm": {"achieved": round(total_bytes * cells / (dt / args.steps) / 1e9 / world, 1),
                                "peak": HBM_PEAK_GBS, "unit": "GB/s per GPU",
                                "frac": round(total_bytes * cells / (dt / args.steps) / 1e9 / world / HBM_PEAK_GBS, 4)}}
        if marks:
            names = ["noise", "gauss", "flow", "erosion"]
            flow_cap = nj._native.lib.nz_flow_fused_max_iterations()
            flow_launches = len(sh.split_iterations(F_IT, flow_cap))
            # the tile API ends a one-launch flow stage with a copy back into the caller's plane; stripes ping-pong
            pingpong = sharded or swap or striped is not None  # explicit src / dst: no copy back, no even-launch-count rule
            mctx = sctx[0] if striped is not None else ctx    # the context whose stream carries the markers
            ero_cap = nj._native.lib.nz_erosion_max_fused_iterations()
            launches = {"noise": 1, "gauss": None,
                        "flow": flow_launches + (1 if flow_launches == 1 and not pingpong else 0),
                        "erosion": len(sh.split_iterations(E_IT, ero_cap)) if pingpong else 2}
            KERNEL_OF["erosion"] = "erosion_reg_kernel<%d>" % (sh.split_iterations(E_IT, ero_cap)[0] if pingpong else 3)
            rcells = cells // world  # rank 0's own cells: the stage figures are per GPU
            if striped is not None:
                rcells = striped.parts[0][1].nown * res  # stripe 0's own cells
            acc = {n: 0.0 for n in names}
            for hs in marks:
                for i, n in enumerate(names):
                    acc[n] += mctx.elapsed_ms(hs[i], hs[i + 1])
            stages_out = {}
            for n in names:
                ms = acc[n] / len(marks)
                gbs = BYTES[n] * rcells / (ms * 1e-3) / 1e9
                stages_out[n] = {"kernel": KERNEL_OF[n], "ms": round(ms, 4), "algorithmic_GB/s": round(gbs, 1),
                                 "frac_hbm": round(gbs / HBM_PEAK_GBS, 4)}
            stages_out["noise"]["valu_Gops/s"] = round(13 * NOISE_OPS_PER_OCTAVE_CELL * rcells /
                                                       (stages_out["noise"]["ms"] * 1e-3) / 1e9, 1)
            stages_out["noise"]["frac_valu"] = round(stages_out["noise"]["valu_Gops/s"] / VALU_PEAK_GOPS, 4)
            stages_out["gauss"]["launches"] = N_gauss = len(sh.split_iterations(G_IT, nj._native.lib.nz_kernel_filter_max_fused(2)))
            if N_gauss & 1 and not pingpong:  # the tile API keeps the launch count even (result back in `src`)
                stages_out["gauss"]["launches"] = N_gauss + 1
            out["stages"] = stages_out
            # The tile API's one-launch flow stage ends with a plane copy back into the caller's buffer: time
            # that copy on its own (outside the timed steps) so the flow KERNEL's launch time is known
            kernel_ms = {n: stages_out[n]["ms"] for n in names}
            if not pingpong and flow_launches == 1:
                scratch = ctx.alloc(rcells)
                hc0 = ctx.record()
                for _ in range(20):
                    ctx.call("nz_flush_write_slice", scratch.ptr, tile.ptr, rcells)
                hc1 = ctx.record()
                hc1.Complete()
                copy_ms = ctx.elapsed_ms(hc0, hc1) / 20
                scratch.Dispose()
                stages_out["flow"]["copy_back_ms"] = round(copy_ms, 4)
                kernel_ms["flow"] = stages_out["flow"]["ms"] - copy_ms
                launches["flow"] = 1
            # `roofline`: the HBM-bound kernel with the largest share of the step.  The fBm kernel is reported
            # beside it against the fp32 VALU rate that bounds it (4 B/cell written for ~1.2 k VALU slots/cell).
            dom = max(("gauss", "flow", "erosion"), key=lambda n: kernel_ms[n])
            n_launch = stages_out["gauss"]["launches"] if dom == "gauss" else launches[dom]
            s = stages_out[dom]
            launch_ms = kernel_ms[dom] / n_launch
            alg_bytes = BYTES[dom] * rcells / n_launch
            out["roofline"] = {"kernel": s["kernel"], "bound": "hbm",
                               "achieved": round(alg_bytes / (launch_ms * 1e-3) / 1e9, 1),
                               "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(alg_bytes / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                               "traffic": None if sharded else pmc_traffic(dom),
                               "launches_per_step": n_launch, "avg_launch_ms": round(launch_ms, 4),
                               "algorithmic_bytes_per_launch": round(alg_bytes),
                               "note": "HBM-bound kernel with the largest share of the step (kernel time / launches); "
                                       "its iterations are fused on chip, so algorithmic bytes per launch exceed the "
                                       "HBM bytes actually moved (traffic) and frac can exceed 1"}
            if out["roofline"]["traffic"]:  # HBM bytes actually moved per launch (PMC) over the launch time
                out["roofline"]["traffic_GB/s"] = round(out["roofline"]["traffic"] / (launch_ms * 1e-3) / 1e9, 1)
            out["valu_roofline"] = {"kernel": stages_out["noise"]["kernel"], "bound": "fp32-valu",
                                    "achieved": stages_out["noise"]["valu_Gops/s"], "peak": VALU_PEAK_GOPS,
                                    "unit": "Gop/s", "frac": stages_out["noise"]["frac_valu"],
                                    "avg_launch_ms": stages_out["noise"]["ms"],
                                    "traffic": None if sharded else pmc_traffic("noise"),
                                    "note": "fBm octave accumulation; %d VALU slots per octave-cell counted in the ISA"
                                            % int(NOISE_OPS_PER_OCTAVE_CELL)}
        if not args.no_cpu_baseline and not sharded:
            out["cpu_baseline"] = cpu_baseline(args.cpu_res)
        if not sharded and striped is None:
            # informational, outside the timed steps; never allowed to cost the JSON line: an exception is recorded, and
            # should one of them ever block, a watchdog prints the line without them and ends the process
            import threading

            def bail():
                out["extras"] = "skipped: an informational measurement did not return within 120 s"
                os.write(real_stdout, (json.dumps(out) + "\n").encode())
                os._exit(0)
            watchdog = threading.Timer(120.0, bail)
            watchdog.daemon = True
            watchdog.start()
            for key, fn in (("two_tiles_in_flight", lambda: two_tiles(nj, ctx, stages, gd, res, p, swap)),
                            ("tile_as_two_stripes", lambda: two_stripes(nj, sh, torch, local_rank, data, res, p))):
                try:
                    out[key] = fn()
                except Exception as e:  # noqa: BLE001
                    out[key] = {"error": "%s: %s" % (type(e).__name__, e)}
            watchdog.cancel()
    if sharded:
        dist.barrier()
        dist.destroy_process_group()
    if striped is not None:
        for c in sctx:
            c.close()
    ctx.close()
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    os.close(real_stdout)
    if out is not None:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
