"""Device-resident context buffers shared between pipelines: the `NativeArray<float>` part of the
reference's PipelineStateManager (Pipeline/PipelineState/PipelineStateManager.cs:13-189,
PipelineState.cs:230-349), its job-fence locks (PipelineStateLock.cs:12-39) and the two stages that move
a tile into / out of a named context buffer (Pipeline/PipelineState/Stage/*.cs).

A context buffer is a DeviceTile owned by the manager and keyed by a string, so a tile produced by one
pipeline (WriteGeneratorContextStage) stays in HBM for the pipelines that read it
(ReadGeneratorContextStage); `SetSavePath` adds the reference's on-disk form (persistence.py).
"""
from .persistence import PipelineSerdeManager
from .pipeline import GeneratorData, MeshStageData, MeshTileStage, MeshType, PipelineStage


class HandleLock:  # PipelineStateLock.cs:12-27
    """Locked until the scheduled write has completed.  `spyHandle` is the LockJob marker scheduled after
    the write; the reference checks that the write is a dependency of it (always true by construction), so
    the lock state is the write's completion."""

    def __init__(self, handle, spy):
        self.jobHandle = handle
        self.spyHandle = spy

    def isLocked(self):
        return not self.jobHandle.IsCompleted


class PipelineStateManager:
    def __init__(self, ctx):
        self.ctx = ctx
        self.buffers = {}
        self.locks = {}
        self.notifier = {}
        self.savedState = None

    def SetSavePath(self, basePath, saveName, saveVersion):  # :18-20 (basePath = Application.persistentDataPath)
        self.savedState = PipelineSerdeManager(basePath, saveName, saveVersion)

    # ---- buffers (:39-75, PipelineState.cs:239-270)
    def GetBuffer(self, name, size=-1, ignoreSaved=False):
        if name not in self.buffers:
            if size < 0:
                raise KeyError("No allocated buffer named %s" % name)
            self.buffers[name] = self.ctx.alloc(size)
        buffer = self.buffers[name]
        if self.savedState is not None and not ignoreSaved:
            cacheSize = self.savedState.CachedSize(name)
            if cacheSize > 0:
                host = self.savedState.ReadData(name)
                if host is not None:
                    full = buffer.ToArray() if host.size < buffer.Length else None
                    if full is not None:
                        full[:host.size] = host
                        host = full
                    buffer.CopyFrom(host[:buffer.Length])
        return buffer

    def SaveBufferToDisk(self, name, size=-1):  # :98-113
        if self.savedState is None:
            raise ValueError("No serde manager is active")
        host = self.GetBufferNoLoad(name).ToArray()
        self.savedState.WriteData(host if size < 0 else host[:size], name)

    def GetBufferNoLoad(self, name):
        if name not in self.buffers:
            raise KeyError("No allocated buffer named %s" % name)
        return self.buffers[name]

    def BufferExists(self, name):  # :115-120
        return name in self.buffers

    def ReleaseBuffer(self, name):  # :122-127
        if name not in self.buffers:
            return False
        self.buffers.pop(name).Dispose()
        return True

    # ---- locks (:136-148, PipelineState.cs:311-329)
    def IsLocked(self, key):
        lock = self.locks.get(key)
        return lock.isLocked() if lock is not None else False

    def TrySetLock(self, key, handle, spyHandle):
        if self.IsLocked(key):  # no release needed: a completed handle is an open lock
            return False
        self.locks[key] = HandleLock(handle, spyHandle)
        return True

    # ---- callbacks (:159-181)
    def RegisterCallback(self, key, action):
        self.notifier.setdefault(key, []).append(action)
        return True

    def RemoveCallback(self, key, action):
        if key not in self.buffers:
            raise KeyError("missing buffer %s" % key)
        if action in self.notifier.get(key, []):
            self.notifier[key].remove(action)
        return True

    def TriggerUpdateCallbacks(self, key):
        for action in list(self.notifier.get(key, [])):
            action()
        return True

    def OnDestroy(self):  # :183-188
        for key in list(self.buffers):
            self.ReleaseBuffer(key)


def _buffer_name(d, contextAlias):  # getBufferName, ReadGeneratorContextStage.cs:18-20
    return "%d_%d__%d__%s" % (d.xpos, d.zpos, d.resolution, contextAlias)


class ReadGeneratorContextStage(PipelineStage):  # Pipeline/PipelineState/Stage/ReadGeneratorContextStage.cs:13-46
    """Fills the work item's tile from the context buffer `{xpos}_{zpos}__{resolution}__{contextAlias}`;
    schedulable once that buffer exists and no write to it is in flight."""

    def __init__(self, ctx, contextAlias=""):
        super().__init__(ctx)
        self.contextAlias = contextAlias

    def IsSchedulable(self, job):
        if job.stageManager is None:
            return False
        name = _buffer_name(job.data, self.contextAlias)
        if not job.stageManager.BufferExists(name):
            return False
        return not job.stageManager.IsLocked(name)

    def Schedule(self, requirements, dependency):
        self.CheckRequirements(GeneratorData, requirements)
        gd = requirements.data
        res = gd.resolution * gd.resolution
        buffer = requirements.stageManager.GetBuffer(_buffer_name(gd, self.contextAlias), res)
        self.jobHandle = self.ctx.call("nz_flush_write_slice", gd.data.ptr, buffer.ptr, res, dep=dependency)


class WriteGeneratorContextStage(PipelineStage):  # .../WriteGeneratorContextStage.cs:13-46
    """Copies the work item's tile into the context buffer and locks the buffer until the copy is done."""

    def __init__(self, ctx, contextAlias=""):
        super().__init__(ctx)
        self.contextAlias = contextAlias

    def IsSchedulable(self, job):
        if job.stageManager is None:
            return False
        return not job.stageManager.IsLocked(_buffer_name(job.data, self.contextAlias))

    def Schedule(self, requirements, dependency):
        self.CheckRequirements(GeneratorData, requirements)
        gd = requirements.data
        res = gd.resolution * gd.resolution
        name = _buffer_name(gd, self.contextAlias)
        buffer = requirements.stageManager.GetBuffer(name, res)
        h1 = self.ctx.call("nz_flush_write_slice", buffer.ptr, gd.data.ptr, res, dep=dependency)
        self.jobHandle = self.ctx.record()  # LockJob: a no-op marker scheduled after the copy
        requirements.stageManager.TrySetLock(name, h1, self.jobHandle)


class MeshTileReferenceDataStage(MeshTileStage):  # Mesh/Stage/MeshTileReferenceDataStage.cs:22-84
    """MeshTileStage whose heights come from the context buffer `{xpos}_{zpos}__{inputResolution}__{contextAlias}`
    instead of the work item's own data; schedulable once that buffer exists and is not being written."""

    def __init__(self, ctx, meshType=MeshType.OvershootSquareGridHeightMap, contextAlias=""):
        super().__init__(ctx, meshType)
        self.contextAlias = contextAlias

    def getBufferName(self, d):
        return "%d_%d__%d__%s" % (d.xpos, d.zpos, d.inputResolution, self.contextAlias)

    def IsSchedulable(self, job):
        if job.stageManager is None:
            return False
        name = self.getBufferName(job.data)
        if not job.stageManager.BufferExists(name):
            return False
        return not job.stageManager.IsLocked(name)

    def Schedule(self, requirements, dependency):
        d = requirements.data
        if not isinstance(d, MeshStageData):
            raise Exception("Unhandled stageio %s" % type(d).__name__)
        buffer = requirements.stageManager.GetBuffer(self.getBufferName(d), d.inputResolution * d.inputResolution)
        own, d.data = d.data, buffer          # the mesh job reads the context buffer (:62-64)
        try:
            super().Schedule(requirements, dependency)
        finally:
            d.data = own
