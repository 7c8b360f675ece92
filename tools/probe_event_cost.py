"""What do the per-stage JobHandles (one hipEventRecord per stage call) cost the metric step?  The four stage entries of
the 4096^2 metric pipeline on one stream, with a handle out of every call (what BasePipeline.Schedule does) against the
same calls with out = NULL (stream order only), alternating on one box.  GPU only: python tools/probe_event_cost.py"""
import ctypes as C
import os
import gc
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import noize_job_amd as nj  # noqa: E402
from noize_job_amd import _native as N  # noqa: E402

gc.disable()  # a full collection pass of the host (tens of ms with a big heap) must not land in a timed loop

res = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cells = res * res
ctx = nj.Context(0)
a, b, work = ctx.alloc(cells), ctx.alloc(cells), ctx.alloc(11 * cells)
lib = N.lib
t = N.RWTile(a.ptr, b.ptr, res, 1)


def step(handles):
    h = N.handle_t(0)
    out = C.byref(h) if handles else None
    N.check(lib.nz_fractal(ctx._h, int(nj.FractalNoise.Simplex), t.read, res, 0.4, 1.0, 2.0, 0.0, 13, 0, 0, 1700, h.value, out), "noise")
    N.check(lib.nz_kernel_filter_stage_rw(ctx._h, C.byref(t), int(nj.KernelFilterType.Gauss5_S1), 17, h.value, out), "filter")
    N.check(lib.nz_flowmap_stage_rw(ctx._h, C.byref(t), work.ptr, 5, 0.0, 0.005, h.value, out), "flow")
    N.check(lib.nz_erosion_stage_rw(ctx._h, C.byref(t), 5, h.value, out), "erosion")


def run(handles, n):
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step(handles)
    ctx.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


run(True, 400)
for rnd in range(4):
    print("round %d: with handles %.4f ms   without %.4f ms" % (rnd, run(True, 400), run(False, 400)), flush=True)
