#!/usr/bin/env python3
"""The pool automaton on a plane with a given share of wet cells, a few jobs in a row (for rocprofv3 --kernel-trace --stats):
tools/probe_pool_wet.py [--res 4096] [--wet 0.3] [--iterations 1] [--jobs 5]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import noize_job_amd as nj  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=4096)
ap.add_argument("--wet", type=float, default=0.3)
ap.add_argument("--iterations", type=int, default=1)
ap.add_argument("--jobs", type=int, default=5)
a = ap.parse_args()
rng = np.random.default_rng(5)
plane = np.where(rng.random((a.res, a.res)) < a.wet, 0.01, 0).astype(np.float32)
height = rng.random((a.res, a.res), dtype=np.float32)
with nj.Context(0) as ctx:
    h = ctx.from_host(height)
    for _ in range(a.jobs):
        wet = ctx.from_host(plane)
        m0 = ctx.record()
        ctx.call("nz_pool_automata", wet.ptr, h.ptr, a.iterations, a.res)
        m1 = ctx.record()
        m1.Complete()
        print("job %.4f ms" % ctx.elapsed_ms(m0, m1))
        wet.Dispose()
