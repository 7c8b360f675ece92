"""How far ahead of the GPU may a host run before HIP makes it wait -- and for how long?  After a synchronisation, N launches of
a ~60 us kernel (nz_constant_job on a 4096^2 plane) back to back, each call timed on the host: without handles, with a
handle out of every call.  GPU only."""
import ctypes as C
import os
import gc
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import noize_job_amd as nj  # noqa: E402
from noize_job_amd import _native as N  # noqa: E402

gc.disable()  # a full collection pass of the host (tens of ms with a big heap) must not land in a timed loop

res = 4096
ctx = nj.Context(0)
a = ctx.alloc(res * res)
lib = N.lib


def run(n, handles):
    ctx.synchronize()
    t = [time.perf_counter()]
    h = N.handle_t(0)
    for _ in range(n):
        N.check(lib.nz_constant_job(ctx._h, 0, a.ptr, None, 1.0, res, 0, C.byref(h) if handles else None), "constant")
        t.append(time.perf_counter())
    ctx.synchronize()
    total = (time.perf_counter() - t[0]) * 1e3
    iv = [(t[k + 1] - t[k]) * 1e3 for k in range(n)]
    slow = [(k, round(iv[k], 2)) for k in range(n) if iv[k] > 0.5]
    print("%s: %d calls, median %.1f us per call, GPU time %.1f ms; calls that took > 0.5 ms: %s"
          % ("a handle per call" if handles else "no handles      ", n, sorted(iv)[n // 2] * 1e3, total, slow[:12]), flush=True)


run(200, False)
for handles in (False, True, False, True):
    run(3000, handles)
