#!/usr/bin/env python3
"""Prints the descent kernel's per-step clock split (library built by tools/probe_descent.sh) at BASELINE config 4."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import noize_job_amd as nj  # noqa: E402
from noize_job_amd import _native  # noqa: E402

res = 8192
lib = _native.lib
fn = lib.nz_debug_descent_probe
fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int32]
fn.restype = C.c_int32
with nj.Context(0) as ctx:
    h = ctx.alloc(res * res)
    gd = nj.GeneratorData("c4", h, res, 0, 0)
    nj.NoiseStage(ctx, nj.FractalNoise.Cellular, 0.4, 1.0, 13, 2.0, 0.0, 1700).Schedule(nj.PipelineWorkItem(gd), nj.JobHandle())
    es = nj.ErosionSettings(PARTICLES_PER_CYCLE=10000, CYCLES=1, WATER_STEPS=10)
    tm = nj.tile_set_meta(res, height=1000, tile_size=res, tile_res=res - 16, margin=8)
    G = nj.LiveErosion(ctx, h, tm, es)
    for c in range(30):
        G.TriggerQueuedBeyerMT([c + 1]).Complete()
    ctx.synchronize()
    fn(None, 1)
    n = 20
    for c in range(n):
        G.TriggerQueuedBeyerMT([100 + c]).Complete()
    ctx.synchronize()
    out = (C.c_ulonglong * 8)()
    fn(out, 0)
    steps = max(out[3], 1)
    print("wave steps %d over %d waves (%d cycles of the job), longest wave %d steps" % (out[3], out[4], n, out[5]))
    for k, name in enumerate(("until the step's loads are in", "the step's arithmetic", "flush + emit")):
        print("  %-32s %8.0f shader clocks per wave step" % (name, out[k] / steps))
    print("  wave steps with a lane going uphill %d, with a lane that cannot (second slope pass) %d" % (out[6], out[7]))
    G.OnDestroy()
