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
