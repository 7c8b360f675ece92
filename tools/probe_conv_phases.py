#!/usr/bin/env python3
"""Where a tile of the CHAINED filter grid spends its time (-DNZ_CONV_PROBE stamps, tools/probe_conv_phases.sh builds that
variant on the GPU box): per workgroup, s_memrealtime (100 MHz) at its start, when its ticket is back, when its producers'
flags are up, when wave 0's tile rows have landed, after the X pass / first barrier of application 1, after every
application, when wave 0's stores are acknowledged, when everybody's are -- plus HW_ID / XCC_ID.  Prints the mean of every
phase per launch of the chain, the shares of a workgroup's lifetime, and how many workgroups a CU holds at a time.
usage: probe_conv_phases.py [res] [iterations] [float mode]"""
import collections
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import noize_job_amd as nj  # noqa: E402

res = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 17
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
SL = 24
lib = nj._native.lib
lib.nz_debug_set_conv_probe.argtypes = [C.c_void_p]
lib.nz_debug_set_conv_probe.restype = C.c_int32
with nj.Context(0) as ctx:
    ctx.float_mode = mode
    gd = nj.GeneratorData("p", ctx.alloc(res * res), res, 0, 0, write=ctx.alloc(res * res))
    nj.NoiseStage(ctx, nj.FractalNoise.Simplex, 0.4, 1.0, 4, 2.0, 0.0, 1700).Schedule(nj.PipelineWorkItem(gd), nj.JobHandle())
    stage = nj.KernelFilterStage(ctx, nj.KernelFilterType.Gauss5_S1, iters)
    NWG = 16384
    probe = ctx.alloc(NWG * SL, dtype=np.uint64)
    probe.CopyFrom(np.zeros(NWG * SL, np.uint64))
    for _ in range(60):  # clocks settle
        stage.Schedule(nj.PipelineWorkItem(gd), nj.JobHandle())
    ctx.synchronize()
    assert lib.nz_debug_set_conv_probe(C.c_void_p(probe.ptr)) == 0
    h0 = ctx.record()
    stage.Schedule(nj.PipelineWorkItem(gd), nj.JobHandle())  # the stamped stage: one chained grid (or its separate launches)
    h1 = stage.jobHandle
    h1.Complete()
    stage_ms = ctx.elapsed_ms(h0, h1)
    p = probe.ToArray((NWG, SL))
p = p[p[:, 0] > 0].astype(np.int64)
t0 = p[:, 0].min() if not (p[:, 18] > 0).any() else p[:, 18].min()
us = lambda a: a / 100.0  # noqa: E731
chained = bool((p[:, 7] > 0).any())
launch = (p[:, 13] >> 32) if chained else np.zeros(len(p), np.int64)
print("float mode %d, %d^2, %d applications: %d workgroups stamped, stage %.1f us by HIP events (probe build), span of the stamps %.1f us, %s"
      % (mode, res, iters, len(p), stage_ms * 1e3, us(p[:, 12].max() - t0), "chained grid" if chained else "separate launches"))
rows = []
for l in sorted(set(launch.tolist())):
    q = p[launch == l]
    T = int((q[0, 1:6] > 0).sum())
    start = q[:, 7] if chained else q[:, 0]
    ph = collections.OrderedDict()
    if chained:
        ph["entry -> ticket back"] = us(q[:, 7] - q[:, 18])
        ph["flag wait"] = us(q[:, 8] - q[:, 7])
        ph["load (issue -> landed)"] = us(q[:, 9] - q[:, 8])
    else:
        ph["load (issue -> landed)"] = us(q[:, 9] - q[:, 0])
    ph["app 1: X pass + edges"] = us(q[:, 16] - q[:, 9])
    ph["app 1: barrier wait"] = us(q[:, 17] - q[:, 16])
    ph["app 1: Z pass"] = us(q[:, 1] - q[:, 17])
    for k in range(1, T):
        ph["app %d" % (k + 1)] = us(q[:, 1 + k] - q[:, k])
    ph["store issue"] = us(q[:, 12] - q[:, T])
    if chained:
        ph["store drain (wave 0)"] = us(q[:, 10] - q[:, 12])
        ph["store drain (barrier)"] = us(q[:, 11] - q[:, 10])
        life = us(q[:, 11] - q[:, 18])
    else:
        life = us(q[:, 12] - q[:, 0])
    print("launch %d: T = %d, %d tiles, first start %.1f us, last end %.1f us, workgroup lifetime mean %.2f us (p10 %.2f, p90 %.2f)" % (
        l, T, len(q), us(q[:, 0].min() - t0), us(q[:, 12].max() - t0), life.mean(), np.percentile(life, 10), np.percentile(life, 90)))
    for k, v in ph.items():
        print("    %-26s mean %6.2f us  p50 %6.2f  p90 %6.2f   %5.1f %% of the lifetime" % (k, v.mean(), np.percentile(v, 50), np.percentile(v, 90),
                                                                                       100.0 * v.mean() / life.mean()))
    rows.append((life.mean(), len(q)))
# CU occupancy over time
cu = collections.defaultdict(list)
end_col = 11 if chained else 12
for i in range(len(p)):
    hw, xcc = int(p[i, 14]), int(p[i, 15]) & 0xf
    cu[(xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf)].append((us(p[i, 18 if chained else 0] - t0), us(p[i, end_col] - t0)))
span = us(p[:, end_col].max() - t0)
conc, busy = [], []
for k, iv in cu.items():
    ev = sorted([(a, 1) for a, _ in iv] + [(b, -1) for _, b in iv])
    n, last, area, on = 0, 0.0, 0.0, 0.0
    for t, d in ev:
        area += n * (t - last)
        on += (t - last) if n > 0 else 0.0
        n, last = n + d, t
    conc.append(area / max(on, 1e-9))
    busy.append(on / span)
print("CUs seen %d; workgroups per CU %.1f; resident workgroups while busy: mean %.2f; CU busy fraction of the span: mean %.3f min %.3f" % (
    len(cu), len(p) / len(cu), np.mean(conc), np.mean(busy), np.min(busy)))
tot_life = sum(a * n for a, n in rows)
print("sum of workgroup lifetimes / (CUs x span) = %.2f resident workgroups on average" % (tot_life / (len(cu) * span)))
# how long a freed slot stays empty: per CU, the k-th end (in time order) against the (k + slots)-th entry
gaps, slots_seen = [], []
for k, iv in cu.items():
    st = sorted(a for a, _ in iv)
    en = sorted(b for _, b in iv)
    n0 = sum(1 for a in st if a < en[0])  # workgroups that entered before the first one left: the CU's slots
    slots_seen.append(n0)
    for j in range(len(st) - n0):
        gaps.append(st[j + n0] - en[j])
gaps = np.array(gaps)
print("slots per CU (entries before the CU's first exit): mean %.2f, min %d, max %d" % (np.mean(slots_seen), min(slots_seen), max(slots_seen)))
print("exit -> the entry that takes the slot: mean %.2f us, p10 %.2f, p50 %.2f, p90 %.2f, p99 %.2f  (%d hand-overs)" % (
    gaps.mean(), np.percentile(gaps, 10), np.percentile(gaps, 50), np.percentile(gaps, 90), np.percentile(gaps, 99), len(gaps)))
