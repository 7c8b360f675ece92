#!/usr/bin/env python3
"""Timeline of one flow_stream_kernel launch from the -DNZ_FLOW_PROBE stamps (tools/probe_flow_stream.sh builds that
variant on the GPU box): when waves start and end, how long the pipeline fill and the steady rows take, the shader clock
they ran at, and how many waves a SIMD holds at a time."""
import collections
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import noize_job_amd as nj  # noqa: E402

res = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lib = nj._native.lib
lib.nz_debug_set_flow_probe.argtypes = [C.c_void_p]
lib.nz_debug_set_flow_probe.restype = C.c_int32
with nj.Context(0) as ctx:
    gd = nj.GeneratorData("p", ctx.alloc(res * res), res, 0, 0, write=ctx.alloc(res * res))
    nj.NoiseStage(ctx, nj.FractalNoise.Simplex, 0.4, 1.0, 13, 2.0, 0.0, 1700).Schedule(nj.PipelineWorkItem(gd), nj.JobHandle())
    nj.KernelFilterStage(ctx, nj.KernelFilterType.Gauss5_S1, 17).Schedule(nj.PipelineWorkItem(gd), nj.JobHandle())
    NW = 16384
    probe = ctx.alloc(NW * 8, dtype=np.uint64)
    probe.CopyFrom(np.zeros(NW * 8, np.uint64))
    for _ in range(200):  # clocks settle
        g2 = nj.GeneratorData("q", gd.data, res, 0, 0, write=gd.write)
        nj.FlowMapStage(ctx, 5, 0.0, 0.005).Schedule(nj.PipelineWorkItem(g2), nj.JobHandle())
    ctx.synchronize()
    assert lib.nz_debug_set_flow_probe(C.c_void_p(probe.ptr)) == 0
    g2 = nj.GeneratorData("q", gd.data, res, 0, 0, write=gd.write)
    nj.FlowMapStage(ctx, 5, 0.0, 0.005).Schedule(nj.PipelineWorkItem(g2), nj.JobHandle())
    ctx.synchronize()
    p = probe.ToArray((NW, 8))
blk = np.nonzero(p[:, 0] > 0)[0]
p = p[p[:, 0] > 0]
t0 = p[:, 0].min()
start, mid, end = (p[:, 0] - t0) / 100.0, (p[:, 2] - t0) / 100.0, (p[:, 4] - t0) / 100.0  # us (100 MHz)
clk = (p[:, 5] - p[:, 1]) / np.maximum(p[:, 4] - p[:, 0], 1) * 100.0  # MHz
inner = (p[:, 7] >> 32) & 1
print("waves %d, launch span %.1f us, shader clock median %.0f MHz" % (len(p), end.max(), np.median(clk)))
print("start times: " + "  ".join("%2d%% by %.1f us" % (q, np.percentile(start, q)) for q in (25, 50, 75, 90, 99, 100)))
print("end times:   " + "  ".join("%2d%% by %.1f us" % (q, np.percentile(end, q)) for q in (1, 10, 25, 50, 75, 90, 100)))
life = end - start
print("wave lifetime: mean %.1f us (min %.1f, median %.1f, max %.1f); fill %.1f us, rest %.1f us" % (
    life.mean(), life.min(), np.median(life), life.max(), (mid - start).mean(), (end - mid).mean()))
for name, m in (("inner strips", inner == 1), ("border strips", inner == 0)):
    if m.any():
        print("  %-14s %5d waves: lifetime mean %.1f us, max %.1f; end mean %.1f" % (name, m.sum(), life[m].mean(), life[m].max(), end[m].mean()))
simd = collections.defaultdict(list)
for i in range(len(p)):
    hw, xcc = int(p[i, 6]), int(p[i, 7]) & 0xf
    simd[(xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf, (hw >> 4) & 3)].append((start[i], end[i]))
conc, busy, cnt = [], [], []
for k, iv in simd.items():
    ev = sorted([(a, 1) for a, _ in iv] + [(b, -1) for _, b in iv])
    n, last, area, on = 0, 0.0, 0.0, 0.0
    for t, d in ev:
        area += n * (t - last)
        on += (t - last) if n > 0 else 0.0
        n, last = n + d, t
    conc.append(area / max(on, 1e-9))
    busy.append(on / end.max())
    cnt.append(len(iv))
print("SIMDs seen %d; waves per SIMD: mean %.2f min %d max %d; resident waves while busy: mean %.2f; SIMD busy fraction of "
      "the span: mean %.2f min %.2f" % (len(simd), np.mean(cnt), min(cnt), max(cnt), np.mean(conc), np.mean(busy), np.min(busy)))
hist = collections.Counter(cnt)
print("waves per SIMD histogram:", dict(sorted(hist.items())))
slot = p[:, 6] & 0xf
for k in sorted(set(slot.tolist())):
    m = slot == k
    print("  wave slot %d: %5d waves, start mean %.1f us, end mean %.1f us (min %.1f, max %.1f)" % (
        k, m.sum(), start[m].mean(), end[m].mean(), end[m].min(), end[m].max()))
# dispatch order (block index) against finishing time: waves of a SIMD are served oldest first
order = np.argsort(start, kind="stable")
for q in range(4):
    m = order[q * len(order) // 4:(q + 1) * len(order) // 4]
    print("  start-time quartile %d: end mean %.1f us" % (q, end[m].mean()))
# is the wave slot a function of the block index (breadth-first fill: block b -> slot b // 1024)?
for lo in range(0, int(blk.max()) + 1, 512):
    m = (blk >= lo) & (blk < lo + 512)
    if m.any():
        h = collections.Counter((p[m, 6] & 0xf).tolist())
        print("  blocks %4d..%4d: slots %s, end mean %.1f us" % (lo, lo + 511, dict(sorted(h.items())), end[m].mean()))
