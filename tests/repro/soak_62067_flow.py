"""Replays the one mismatch of round 6's first soak (tests/soak.py, seed 62067): flow map x12 on a 300^2 plane of sparse impulses -- the
oracle's ternary clamp passed a NaN on where Unity.Mathematics' clamp gives 1 (HISTORY.md).  40 repetitions per stage of the seed."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # tests/repro/ -> the repository (under tests/: these scripts drive the oracle, the checker)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import noize_job_amd as nj
import oracle as O
import test_gpu_sweep as S
seed = 62067
orig = np.random.default_rng
np.random.default_rng = lambda s=None, _k=seed: orig(None if s is None else s * 7919 + _k)
ctx = nj.Context(0)
rng = np.random.default_rng(1000 + 1)
f32 = np.float32
for loop in range(7):
    res = int(rng.choice(S.SIZES))
    t = S._tile(rng, res)
    ft = int(rng.choice([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13]))
    it = int(rng.integers(1, 12)) if ft != 11 else int(rng.integers(1, 3))
    width, sigma, it2 = int(rng.integers(1, 27)), int(rng.integers(0, 16)), int(rng.integers(1, 4))
    width, it3 = int(rng.integers(1, 26)), int(rng.integers(1, 4))
    it4 = int(rng.integers(1, 14))
    itf = int(rng.integers(1, 13))
    lo, hi = [(0.0, 0.005), (-0.1, 0.1), (0.0, 0.0)][int(rng.integers(0, 3))]
    itt = int(rng.integers(1, 4))
    want = O.flowmap(t, itf, lo, hi)
    bad_runs = 0
    for rep in range(40):
        got = S._run(nj, nj.FlowMapStage(ctx, itf, lo, hi), nj.GeneratorData("f", ctx.from_host(t), res))
        if not np.array_equal(got, want, equal_nan=True):
            bad_runs += 1
            d = np.argwhere(~((got == want) | (np.isnan(got) & np.isnan(want))))
            if bad_runs <= 2:
                print("   rep", rep, "cells differing", len(d), "first", d[:5].tolist(), "got", got[tuple(d[0])], "want", want[tuple(d[0])],
                      "rows", d[:, 0].min(), d[:, 0].max(), "cols", d[:, 1].min(), d[:, 1].max())
    print("loop", loop, "res", res, "flow it", itf, lo, hi, "tile kind?", "bad runs of 40:", bad_runs, flush=True)
