"""Device context, device tiles and JobHandle objects on top of the C ABI.

`Context` is the analogue of the Unity job scheduler the reference schedules onto: stage calls are
enqueued on one HIP stream and return a `JobHandle` (Unity.Jobs.JobHandle) that can be polled
(`IsCompleted`) or waited on (`Complete()`), which is how BasePipeline drives completion
(Pipeline/Executable/Pipeline.cs:160-181).
"""
import ctypes as C

import numpy as np

from . import _native as N


class JobHandle:
    """Unity.Jobs.JobHandle analogue: a marker on the issuing context's stream.  `JobHandle()` ==
    default(JobHandle), which is already complete.  The id names its context (nz_handle_context_id), so a handle may
    be handed as a dependency to a stage of ANY context: the consumer's stream waits for it on the device."""
    __slots__ = ("ctx", "id")

    def __init__(self, ctx=None, hid=0):
        self.ctx = ctx
        self.id = int(hid)

    @property
    def IsCompleted(self):
        if self.id == 0 or self.ctx is None:
            return True
        done = C.c_int32(0)
        N.check(N.lib.nz_handle_query(self.ctx._h, self.id, C.byref(done)), "nz_handle_query")
        return bool(done.value)

    def Complete(self):
        if self.id and self.ctx is not None:
            N.check(N.lib.nz_handle_wait(self.ctx._h, self.id), "nz_handle_wait")

    @staticmethod
    def CombineDependencies(ctx, *handles):
        """JobHandle.CombineDependencies: a marker on `ctx`'s stream that completes after all `handles` (of any
        contexts)."""
        ids = [h.id for h in handles if h is not None and h.id]
        arr = (N.handle_t * max(1, len(ids)))(*ids)
        out = N.handle_t(0)
        N.check(N.lib.nz_handle_combine(ctx._h, arr, len(ids), C.byref(out)), "nz_handle_combine")
        return JobHandle(ctx, out.value)

    def __repr__(self):
        return "JobHandle(ctx %d, #%d)" % (self.id >> 40, self.id & ((1 << 40) - 1))


def _dep(dep):
    if dep is None:
        return 0
    return dep.id if isinstance(dep, JobHandle) else int(dep)


class DeviceTile:
    """NativeArray<float>/NativeSlice<float> analogue living in HBM (nz_tile_alloc).  Also wraps
    foreign device memory (e.g. a torch tensor's data_ptr) without owning it."""

    def __init__(self, ctx, length, ptr=None, dtype=np.float32):
        self.ctx = ctx
        self.Length = int(length)
        self.dtype = np.dtype(dtype)
        self._owned = ptr is None
        if ptr is None:
            p = N.dev_ptr()
            nfloats = (self.Length * self.dtype.itemsize + 3) // 4
            N.check(N.lib.nz_tile_alloc(ctx._h, nfloats, C.byref(p)), "nz_tile_alloc")
            self.ptr = p.value
        else:
            self.ptr = int(ptr)

    # NativeArray.CopyFrom / CopyTo
    def CopyFrom(self, host, dep=None):
        host = np.ascontiguousarray(host, dtype=self.dtype).reshape(-1)
        assert host.size == self.Length, "length mismatch"
        N.check(N.lib.nz_tile_upload(self.ctx._h, self.ptr, host.ctypes.data, host.nbytes // 4, _dep(dep), None),
                "nz_tile_upload")
        self.ctx.synchronize()  # the pageable host array may go away
        return self

    def ToArray(self, shape=None, dep=None):
        out = np.empty(self.Length, self.dtype)
        N.check(N.lib.nz_bytes_download(self.ctx._h, self.ptr, out.ctypes.data, out.nbytes, _dep(dep), None),
                "nz_bytes_download")
        self.ctx.synchronize()
        return out.reshape(shape) if shape is not None else out

    def Dispose(self):
        if self._owned and self.ptr:
            N.check(N.lib.nz_tile_free(self.ctx._h, self.ptr), "nz_tile_free")
        self.ptr = 0

    @property
    def IsCreated(self):
        return bool(self.ptr)

    def offset(self, n_elems, length):
        """NativeSlice(array, start, length) view."""
        return DeviceTile(self.ctx, length, ptr=self.ptr + n_elems * self.dtype.itemsize, dtype=self.dtype)


class Context:
    def __init__(self, device=0, stream=None):
        h = N.ctx_p()
        if stream is None:
            N.check(N.lib.nz_ctx_create(device, C.byref(h)), "nz_ctx_create")
        else:
            N.check(N.lib.nz_ctx_create_on_stream(device, C.c_void_p(int(stream)), C.byref(h)),
                    "nz_ctx_create_on_stream")
        self._h = h
        self.device = device

    @staticmethod
    def device_count():
        n = C.c_int32(0)
        rc = N.lib.nz_device_count(C.byref(n))
        return n.value if rc == N.NZ_OK else 0

    def close(self):
        if self._h:
            N.lib.nz_ctx_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def synchronize(self):
        N.check(N.lib.nz_ctx_synchronize(self._h), "nz_ctx_synchronize")

    # nz_ctx_set_float_mode: FloatMode.Strict / FloatMode.Fast of the [BurstCompile] attributes (Fractal.cs:19,
    # KernelJob.cs:17, FlowMapJob.cs:16) as a property of the context the jobs are scheduled on
    @property
    def float_mode(self):
        return N.lib.nz_ctx_float_mode(self._h)

    @float_mode.setter
    def float_mode(self, mode):
        N.check(N.lib.nz_ctx_set_float_mode(self._h, int(mode)), "nz_ctx_set_float_mode")

    def alloc(self, length, dtype=np.float32):
        return DeviceTile(self, length, dtype=dtype)

    def from_host(self, arr):
        arr = np.ascontiguousarray(arr)
        return DeviceTile(self, arr.size, dtype=arr.dtype).CopyFrom(arr)

    def wrap(self, ptr, length, dtype=np.float32):
        return DeviceTile(self, length, ptr=ptr, dtype=dtype)

    def record(self):
        out = N.handle_t(0)
        N.check(N.lib.nz_handle_record(self._h, C.byref(out)), "nz_handle_record")
        return JobHandle(self, out.value)

    def elapsed_ms(self, start, stop):
        ms = C.c_float(0)
        N.check(N.lib.nz_handle_elapsed_ms(self._h, start.id, stop.id, C.byref(ms)), "nz_handle_elapsed_ms")
        return ms.value

    # generic call helper: appends (dep, &out) and wraps the returned handle
    def call(self, name, *args, dep=None, handle=True):
        """handle=False: `out` is NULL -- no event is recorded (one costs the stream ~3 us); the work is ordered by this
        context's stream only and the JobHandle returned is the empty one.  For links of a chain that stays on this
        context; never hand such a handle to ANOTHER context as a dependency."""
        if not handle:
            N.check(getattr(N.lib, name)(self._h, *args, _dep(dep), None), name)
            return JobHandle()
        out = N.handle_t(0)
        N.check(getattr(N.lib, name)(self._h, *args, _dep(dep), C.byref(out)), name)
        return JobHandle(self, out.value)
