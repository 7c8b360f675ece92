#!/usr/bin/env python3
"""Soak run of the random sweeps of tests/test_gpu_sweep.py with many more seeds (one-off hunt for rare
mismatches between the HIP path and the oracle; kept under tests/ because it drives the oracle, not collected by
pytest).  usage: python3 tests/soak.py [minutes]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))  # test_gpu_sweep, conftest
import noize_job_amd as nj  # noqa: E402
import oracle as O  # noqa: E402
import test_gpu_sweep as S  # noqa: E402

minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
deadline = time.time() + 60 * minutes
ctx = nj.Context(0)
f32 = np.float32
seed, fails, runs = int(os.environ.get("SOAK_SEED", "10000")), 0, 0
last_print = time.time()
orig_rng = np.random.default_rng
while time.time() < deadline:
    seed += 1
    # the sweep functions derive their generator from a small offset + seed: shift the whole family
    np.random.default_rng = lambda s=None, _k=seed: orig_rng(None if s is None else s * 7919 + _k)
    for name, fn, args in (("stencils", S.test_filters_erosion_flow_random_sizes, (nj, ctx, O, 1)),
                           ("pairs", S.test_rw_pair_random_stage_chains, (nj, ctx, O, 1)),
                           ("mesh", S.test_mesh_random_shapes, (nj, ctx, O, 1)),
                           ("stripes", S.test_stripe_entry_points_random_geometry_and_pitch, (nj, ctx, O, 1)),
                           ("noise", S.test_noise_random_parameters, (nj, ctx, O, 1 + seed % 7)),
                           ("nonfinite", S.test_non_finite_cells_propagate_like_the_oracle, (nj, ctx, O, 1))):
        runs += 1
        try:
            fn(*args)
        except AssertionError as e:
            fails += 1
            print("MISMATCH seed %d %s: %s" % (seed, name, str(e).splitlines()[0][:300]), flush=True)
        except nj.NoizeError as e:
            print("seed %d %s: rejected shape (%s)" % (seed, name, e), flush=True)
    if seed % 20 == 0 or time.time() - last_print > 30:  # the box kills a run that stays silent for minutes
        last_print = time.time()
        print("seed %d: %d runs, %d mismatches" % (seed, runs, fails), flush=True)
np.random.default_rng = orig_rng
print("done: %d runs, %d mismatches" % (runs, fails))
sys.exit(1 if fails else 0)
