"""Host logic of the Pipeline/Stage mirror that needs no GPU (Pipeline/Stage/PipelineStage.cs,
Pipeline/Executable/Pipeline.cs semantics)."""
import pytest


class FakeTile:
    def __init__(self, n):
        self.Length = n
        self.ptr = 0


def make_stage(nj, log, name, schedulable=True):
    class S(nj.PipelineStage):
        def __init__(self):
            super().__init__(ctx=None)
            self.resized = []

        def ResizeNativeContainers(self, size):
            self.resized.append(size)

        def IsSchedulable(self, job):
            return schedulable() if callable(schedulable) else schedulable

        def Schedule(self, requirements, dependency):
            self.CheckRequirements(nj.GeneratorData, requirements)
            log.append((name, requirements.data.uuid, dependency.id))
            self.jobHandle = nj.JobHandle(None, dependency.id + 1)

        def OnStageComplete(self):
            log.append((name, "complete"))

    return S()


def test_stages_chain_through_on_stage_scheduled(nj):
    log = []
    a, b, c = (make_stage(nj, log, n) for n in "abc")
    pipe = nj.BasePipeline([a, b, c], alias="t")
    seen = []
    pipe.Schedule(nj.GeneratorData("u1", FakeTile(16), 4), scheduleAction=lambda d, h: seen.append((d.uuid, h.id)))
    # each stage receives the previous stage's handle (PipelineStage.cs:44-48, Pipeline.cs:140-149)
    assert log == [("a", "u1", 0), ("b", "u1", 1), ("c", "u1", 2)]
    assert seen == [("u1", 3)] and pipe.pipelineRunning and not pipe.pipelineBeingScheduled
    done = []
    pipe.activeItem.completeAction = lambda d: done.append(d.uuid)
    assert pipe.LateUpdate() is True  # JobHandle(None, ..) counts as completed
    assert done == ["u1"] and log[-3:] == [("a", "complete"), ("b", "complete"), ("c", "complete")]


def test_check_requirements_resizes_once_per_length_and_rejects_other_io(nj):
    log = []
    s = make_stage(nj, log, "s")
    s.ReceiveHandledInput(nj.PipelineWorkItem(nj.GeneratorData("x", FakeTile(64), 8)), nj.JobHandle())
    s.ReceiveHandledInput(nj.PipelineWorkItem(nj.GeneratorData("y", FakeTile(64), 8)), nj.JobHandle())
    s.ReceiveHandledInput(nj.PipelineWorkItem(nj.GeneratorData("z", FakeTile(256), 16)), nj.JobHandle())
    assert s.resized == [64, 256]
    with pytest.raises(Exception, match="Unhandled stageio MeshStageData"):
        s.ReceiveHandledInput(nj.PipelineWorkItem(nj.MeshStageData("m", FakeTile(4))), nj.JobHandle())


def test_empty_pipeline_throws(nj):
    with pytest.raises(Exception, match="No stages in pipeline"):
        nj.BasePipeline(None).Schedule(nj.GeneratorData("u", FakeTile(1), 1))


def test_queue_and_dependency_hell(nj):
    log = []
    ready = {"ok": False}
    s = make_stage(nj, log, "s", schedulable=lambda: ready["ok"])
    pipe = nj.BasePipeline([s])
    pipe.Enqueue(nj.GeneratorData("q1", FakeTile(4), 2))
    pipe.Enqueue(nj.GeneratorData("q2", FakeTile(4), 2))
    pipe.Update()
    assert log == [] and len(pipe.dependencyHell) == 2  # parked (Pipeline.cs:202-214)
    ready["ok"] = True
    pipe.RunToCompletion()
    assert [e[1] for e in log if e[1] != "complete"] == ["q1", "q2"]


def test_blur_helper_limit_width(nj):
    assert [nj.BlurHelper.limitWidth(w) for w in (1, 2, 3, 4, 24, 25, 26, 99)] == [3, 3, 3, 5, 25, 25, 25, 25]


def test_enum_values_match_the_reference(nj):
    assert int(nj.FractalNoise.Simplex) == 3 and int(nj.FractalNoise.DomainRotatedSimplex) == 7
    assert int(nj.KernelFilterType.Gauss5_S1) == 2 and int(nj.KernelFilterType.Prewitt3Vertical) == 13
    assert int(nj.GaussSigma.s8d00) == 15 and int(nj.MeshType.OvershootSquareGridHeightMap) == 1


def test_reduce_pipeline_requests_both_upstreams_then_reduces(nj):
    # ReducePipeline.cs:82-148 with stand-in stages: order of events and the ReduceData handed to the stages
    log = []

    class FakeCtx:
        def alloc(self, n):
            t = FakeTile(n)
            t.IsCreated = True
            t.Dispose = lambda: log.append(("dispose", n))
            return t

    class R(nj.PipelineStage):
        def __init__(self):
            super().__init__(ctx=None)

        def Schedule(self, requirements, dependency):
            self.CheckRequirements(nj.ReduceData, requirements)
            d = requirements.data
            log.append(("reduce", d.uuid, d.data.Length, d.rightData.Length, d.xpos, d.zpos))
            self.jobHandle = nj.JobHandle(None, 1)

    left = nj.BasePipeline([make_stage(nj, log, "L")], "left")
    right = nj.BasePipeline([make_stage(nj, log, "R")], "right")
    red = nj.ReducePipeline(FakeCtx(), [R()], left, right, "red")
    assert red.GetDependencies() == [left, right, red, left, right]
    done = []
    red.Enqueue(nj.GeneratorData("u1", FakeTile(16), 4, 5, 6), completeAction=lambda d: done.append(d.uuid))
    red.Enqueue(nj.GeneratorData("u2", FakeTile(16), 4, 7, 8), completeAction=lambda d: done.append(d.uuid))
    red.RunToCompletion()
    assert done == ["u1", "u2"]
    events = [e for e in log if e[0] in ("L", "R", "reduce") and e[1] != "complete"]
    assert events == [("L", "u1", 0), ("R", "u1", 0), ("reduce", "u1", 16, 16, 5, 6),
                      ("L", "u2", 0), ("R", "u2", 0), ("reduce", "u2", 16, 16, 7, 8)]
    red.Enqueue(nj.GeneratorData("u3", FakeTile(64), 8), completeAction=lambda d: done.append(d.uuid))
    red.RunToCompletion()                   # a new length reallocates the right-hand plane (:92-99)
    assert ("dispose", 16) in log and done[-1] == "u3"
    red.Destroy()
    assert ("dispose", 64) in log
    with pytest.raises(Exception, match="Unhandled stageio"):
        red.ScheduleUpstreams(nj.PipelineWorkItem(nj.MeshStageData("m", FakeTile(4))))


def test_context_stages_are_gated_by_the_state_manager(nj, tmp_path):
    # Pipeline/PipelineState/Stage/*.cs: a read is schedulable once the named buffer exists and is not locked,
    # a write while no earlier write is in flight; BasePipeline parks unschedulable items (dependencyHell)
    import numpy as np
    calls = []

    class HostTile:
        def __init__(self, n):
            self.Length, self.ptr, self.a, self.IsCreated = n, id(self), np.zeros(n, np.float32), True

        def ToArray(self, shape=None):
            return self.a.copy()

        def CopyFrom(self, host):
            self.a[:] = host
            return self

        def Dispose(self):
            self.IsCreated = False

    class Handle:
        def __init__(self, done=True):
            self.IsCompleted, self.id = done, 1

        def Complete(self):
            self.IsCompleted = True

    class FakeCtx:
        pending = None

        def alloc(self, n):
            return HostTile(n)

        def call(self, name, *args, dep=None):
            calls.append((name, args[2]))
            FakeCtx.pending = Handle(done=False)
            return FakeCtx.pending

        def record(self):
            return Handle()

    ctx = FakeCtx()
    mgr = nj.PipelineStateManager(ctx)
    rd, wr = nj.ReadGeneratorContextStage(ctx, "height"), nj.WriteGeneratorContextStage(ctx, "height")
    gd = nj.GeneratorData("t", HostTile(16), 4, 3, -7)
    assert not rd.IsSchedulable(nj.PipelineWorkItem(gd))                      # no stageManager
    wi = nj.PipelineWorkItem(gd, stageManager=mgr)
    assert not rd.IsSchedulable(wi) and wr.IsSchedulable(wi)                   # buffer does not exist yet
    wr.Schedule(wi, nj.JobHandle())
    assert mgr.BufferExists("3_-7__4__height") and calls == [("nz_flush_write_slice", 16)]
    assert mgr.IsLocked("3_-7__4__height") and not rd.IsSchedulable(wi) and not wr.IsSchedulable(wi)
    assert mgr.TrySetLock("3_-7__4__height", Handle(), Handle()) is False      # still locked
    FakeCtx.pending.Complete()
    assert not mgr.IsLocked("3_-7__4__height") and rd.IsSchedulable(wi) and wr.IsSchedulable(wi)
    # callbacks, release, persistence hooks
    hits = []
    mgr.RegisterCallback("3_-7__4__height", lambda: hits.append(1))
    mgr.TriggerUpdateCallbacks("3_-7__4__height")
    assert hits == [1]
    with pytest.raises(ValueError, match="No serde manager"):
        mgr.SaveBufferToDisk("3_-7__4__height")
    mgr.SetSavePath(str(tmp_path), "ctx", "v1")
    mgr.GetBufferNoLoad("3_-7__4__height").a[:] = np.arange(16, dtype=np.float32)
    mgr.SaveBufferToDisk("3_-7__4__height")
    mgr2 = nj.PipelineStateManager(ctx)
    mgr2.SetSavePath(str(tmp_path), "ctx", "v1")
    assert np.array_equal(mgr2.GetBuffer("3_-7__4__height", 16).a, np.arange(16, dtype=np.float32))  # loaded from disk
    assert np.array_equal(mgr2.GetBuffer("other", 4, ignoreSaved=True).a, np.zeros(4, np.float32))
    with pytest.raises(KeyError):
        mgr.GetBuffer("missing")
    assert mgr.ReleaseBuffer("3_-7__4__height") and not mgr.BufferExists("3_-7__4__height")
    mgr2.OnDestroy()
    assert not mgr2.BufferExists("other")
    # the pipeline hands its contextManager to work items; a read queued before the write waits in dependencyHell
    mgr3 = nj.PipelineStateManager(ctx)
    reader = nj.BasePipeline([nj.ReadGeneratorContextStage(ctx, "h")], "reader", contextManager=mgr3)
    reader.Enqueue(nj.GeneratorData("t", HostTile(16), 4, 0, 0))
    reader.Update()
    assert not reader.pipelineRunning and len(reader.dependencyHell) == 1
    writer = nj.BasePipeline([nj.WriteGeneratorContextStage(ctx, "h")], "writer", contextManager=mgr3)
    writer.Enqueue(nj.GeneratorData("t", HostTile(16), 4, 0, 0))
    writer.Update()
    reader.Update()
    assert not reader.pipelineRunning                                          # the write is still in flight
    FakeCtx.pending.Complete()
    reader.Update()
    assert reader.pipelineRunning and not reader.dependencyHell
