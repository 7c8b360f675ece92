#!/usr/bin/env python3
"""Timeline of one conv_reg_kernel launch from the -DNZ_CONV_PROBE stamps (tools/probe_conv_phases.sh builds that
variant on the GPU box): when workgroups start and end, how long the load + first application, the later
applications and the store take, and how many workgroups a CU runs at a time."""
import collections
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import noize_job_amd as nj  # noqa: E402

res = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 17
lib = nj._native.lib
lib.nz_debug_set_conv_probe.argtypes = [C.c_void_p]
lib.nz_debug_set_conv_probe.restype = C.c_int32
with nj.Context(0) as ctx:
    gd = nj.GeneratorData("p", ctx.alloc(res * res), res, 0, 0, write=ctx.alloc(res * res))
    nj.NoiseStage(ctx, nj.FractalNoise.Simplex, 0.4, 1.0, 4, 2.0, 0.0, 1700).Schedule(nj.PipelineWorkItem(gd), nj.JobHandle())
    stage = nj.KernelFilterStage(ctx, nj.KernelFilterType.Gauss5_S1, iters)
    NWG = 8192
    probe = ctx.alloc(NWG * 16, dtype=np.uint64)
    probe.CopyFrom(np.zeros(NWG * 16, np.uint64))
    for _ in range(60):  # clocks settle
        stage.Schedule(nj.PipelineWorkItem(gd), nj.JobHandle())
    ctx.synchronize()
    assert lib.nz_debug_set_conv_probe(C.c_void_p(probe.ptr)) == 0
    T1 = int(os.environ.get("PROBE_T", "5"))  # one launch of T1 applications is stamped
    nj.KernelFilterStage(ctx, nj.KernelFilterType.Gauss5_S1, T1).Schedule(nj.PipelineWorkItem(gd), nj.JobHandle())
    ctx.synchronize()
    p = probe.ToArray((NWG, 16))
p = p[p[:, 0] > 0]
T = int((p[0, 1:12] > 0).sum())
t0 = p[:, 0].min()
start, end = (p[:, 0] - t0) / 100.0, (p[:, 12] - t0) / 100.0  # us (100 MHz)
app = [(p[:, 1] - p[:, 0]) / 100.0] + [(p[:, 1 + k] - p[:, k]) / 100.0 for k in range(1, T)]
store = (p[:, 12] - p[:, T]) / 100.0
print("workgroups %d, applications in the stamped launch %d, launch span %.1f us" % (len(p), T, end.max()))
print("start times: " + "  ".join("%2d%% by %.1f us" % (q, np.percentile(start, q)) for q in (25, 50, 75, 90, 100)))
print("workgroup lifetime: mean %.1f us (min %.1f, max %.1f)" % ((end - start).mean(), (end - start).min(), (end - start).max()))
print("load + application 1: %.2f us   later applications: %s us   store: %.2f us" % (
    app[0].mean(), " ".join("%.2f" % a.mean() for a in app[1:]), store.mean()))
cu = collections.defaultdict(list)
for i in range(len(p)):
    hw, xcc = int(p[i, 14]), int(p[i, 15]) & 0xf
    cu[(xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf)].append((start[i], end[i]))
conc, busy = [], []
for k, iv in cu.items():
    ev = sorted([(a, 1) for a, _ in iv] + [(b, -1) for _, b in iv])
    n, last, area, on = 0, 0.0, 0.0, 0.0
    for t, d in ev:
        area += n * (t - last)
        on += (t - last) if n > 0 else 0.0
        n, last = n + d, t
    conc.append(area / max(on, 1e-9))
    busy.append(on / end.max())
print("CUs seen %d; workgroups per CU %.2f; resident workgroups while busy: mean %.2f; CU busy fraction of the span: mean %.2f min %.2f" % (
    len(cu), len(p) / len(cu), np.mean(conc), np.mean(busy), np.min(busy)))
per_xcc = collections.Counter(k[0] for k in cu for _ in cu[k])
print("workgroups per XCC:", dict(sorted(per_xcc.items())))
