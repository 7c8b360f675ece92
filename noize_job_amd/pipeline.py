"""Host-side mirror of the reference's Pipeline / PipelineStage operator API for the hot path.

Same names, field names, argument meaning and error behaviour as the C# classes, so that stage
graphs written against the reference (NoiseStage -> KernelFilterStage -> FlowMapStage ->
MeshTileStage ...) run unchanged; each `Schedule` body calls the C ABI where the reference calls a
Burst job delegate.  (The C# host in INTEGRATION.md is the same code in the reference's own
language; no .NET toolchain exists in this image, so the Python mirror is what the tests drive.)

Reference files: Pipeline/Stage/PipelineStage.cs, Pipeline/Stage/StageIO.cs,
Pipeline/Stage/StageIOTypes/*.cs, Pipeline/Stage/PipelineDefinition.cs,
Pipeline/Executable/Pipeline.cs, Noise/NoiseStage.cs, Filter/KernelFilterStage.cs,
Filter/Kernel/Blur/Stage{Gaussian,Smooth}Blur.cs, Geologic/Stage/FlowMapStage.cs,
Mesh/Stage/MeshTileStage.cs.
"""
import collections
import ctypes as C
import enum

import numpy as np

from . import _native as N
from .runtime import JobHandle


# ---- enums (values are the C# enum values) ----------------------------------------------------
class FractalNoise(enum.IntEnum):  # Noise/NoiseStage.cs:15-24
    Sin = 0
    Perlin = 1
    PeriodicPerlin = 2
    Simplex = 3
    RotatedSimplex = 4
    Cellular = 5
    DomainRotatedPerlin = 6
    DomainRotatedSimplex = 7


class KernelFilterType(enum.IntEnum):  # Filter/Kernel/KernelJob.cs:79-94
    Gauss9_S1 = 0
    Gauss7_S1 = 1
    Gauss5_S1 = 2
    Gauss3_S1 = 3
    Gauss9_S2 = 4
    Gauss7_S2 = 5
    Gauss5_S2 = 6
    Gauss3_S2 = 7
    Smooth3 = 8
    Sobel3Horizontal = 9
    Sobel3Vertical = 10
    Sobel3_2D = 11
    Prewitt3Horizontal = 12
    Prewitt3Vertical = 13


class GaussSigma(enum.IntEnum):  # Filter/Kernel/Blur/BlurKernels.cs:8-25; sigma = 0.5 * (value + 1)
    s0d50 = 0
    s1d00 = 1
    s1d50 = 2
    s2d00 = 3
    s2d50 = 4
    s3d00 = 5
    s3d50 = 6
    s4d00 = 7
    s4d50 = 8
    s5d00 = 9
    s5d50 = 10
    s6d00 = 11
    s6d50 = 12
    s7d00 = 13
    s7d50 = 14
    s8d00 = 15


class ConstantOperationType(enum.IntEnum):  # Filter/ConstantStage.cs:15-18
    MULTIPLY = 0
    BINARIZE = 1


class ReductionType(enum.IntEnum):  # Filter/Reduce/ReduceStage.cs:12-18
    SUBTRACT = 0
    MULTIPLY = 1
    ROOTSUMSQUARES = 2
    MAX = 3
    MIN = 4


class MeshType(enum.IntEnum):  # Mesh/Stage/MeshTileStage.cs:23-26
    SquareGridHeightMap = 0
    OvershootSquareGridHeightMap = 1


class BlurHelper:  # Filter/Kernel/Blur/BlurKernels.cs:27-37
    max_width = 25

    @staticmethod
    def limitWidth(width):
        if width % 2 == 0:
            width += 1
        width = min(width, BlurHelper.max_width)
        return max(3, width)


# ---- StageIO payloads ---------------------------------------------------------------------------
class StageIO:  # Pipeline/Stage/StageIO.cs:8-11
    def __init__(self, uuid="", data=None):
        self.uuid = uuid
        self.data = data  # DeviceTile (the NativeSlice<float>)


class GeneratorData(StageIO):  # StageIOTypes/GeneratorData.cs:9-15
    """`write` (optional, new-framework): a second plane of the same size, the WRITE slice of the tile's RWTileData
    pair (Pipeline/Tiles/TileData.cs:49-93).  With it the stencil stages (kernel filter, blurs, erosion, flow map)
    run their nz_*_rw forms: TileHelpers.SWAP_RWTILE is a swap of `data` and `write` instead of a flush copy, so
    after a stage `data` is whichever of the two planes holds the result."""

    def __init__(self, uuid="", data=None, resolution=512, xpos=0, zpos=0, write=None):
        super().__init__(uuid, data)
        self.resolution = resolution
        self.xpos = xpos
        self.zpos = zpos
        self.write = write


def _tiles_of(d):
    """The planes an element-wise / per-tile job has to visit: the payload's one plane, or every tile of a
    GeneratorDataBatch (those jobs take `resolution`, not a cell count, so a batch goes tile by tile)."""
    if isinstance(d, GeneratorDataBatch):
        return [d.tile(k) for k in range(d.count)]
    return [d.data]


def _call_rw(ctx, name, d, *args, dep=None):
    """Runs an nz_*_rw entry on the payload's READ / WRITE pair and adopts the pair as the call left it."""
    t = N.RWTile(d.data.ptr, d.write.ptr, d.resolution, getattr(d, "count", 1))
    handle = ctx.call(name, C.byref(t), *args, dep=dep)
    if t.read != d.data.ptr:
        d.data, d.write = d.write, d.data
    return handle


class GeneratorDataBatch(GeneratorData):
    """New-framework payload: `count` independent tiles of `resolution`^2 cells stored back to back in `data`
    (tile k at offset k * resolution^2), world positions in `positions` (device int32 array {xpos, zpos} per
    tile).  The noise / filter / blur / erosion / flow-map stages run such a batch through one launch
    sequence (nz_*_batch); every tile comes out exactly as it would alone."""

    def __init__(self, uuid="", data=None, resolution=512, positions=None, count=1, write=None):
        super().__init__(uuid, data, resolution, 0, 0, write)
        self.positions = positions
        self.count = count

    @classmethod
    def create(cls, ctx, uuid, resolution, positions):
        """positions: sequence of (xpos, zpos); allocates the stacked planes and uploads the positions."""
        pos = np.ascontiguousarray(positions, np.int32).reshape(-1, 2)
        return cls(uuid, ctx.alloc(len(pos) * resolution * resolution), resolution, ctx.from_host(pos), len(pos))

    def tile(self, k):
        n = self.resolution * self.resolution
        return self.data.offset(k * n, n)


class MeshStageData(StageIO):  # StageIOTypes/MeshStageData.cs:9-21
    def __init__(self, uuid="", data=None, resolution=512, inputResolution=512, marginPix=5, tileSize=512.0,
                 tileHeight=512.0, xpos=0, zpos=0, mesh=None, count=1):
        super().__init__(uuid, data)
        self.count = count  # new-framework: `count` height planes stored back to back -> `count` meshes
        self.resolution = resolution
        self.inputResolution = inputResolution
        self.marginPix = marginPix
        self.tileSize = tileSize
        self.tileHeight = tileHeight
        self.xpos = xpos
        self.zpos = zpos
        self.mesh = mesh  # MeshBuffers, stands in for UnityEngine.Mesh


class ReduceData(StageIO):  # StageIOTypes/ReduceData.cs:9-17
    def __init__(self, uuid="", data=None, rightData=None, resolution=512, xpos=0, zpos=0):
        super().__init__(uuid, data)
        self.rightData = rightData
        self.resolution = resolution
        self.xpos = xpos
        self.zpos = zpos


class DownsampleData(StageIO):  # StageIOTypes/DownsampleData.cs:9-17
    def __init__(self, uuid="", data=None, inputData=None, resolution=512, inputResolution=512):
        super().__init__(uuid, data)
        self.inputData = inputData
        self.resolution = resolution
        self.inputResolution = inputResolution


class MeshBuffers:
    """What Mesh.AllocateWritableMeshData + PositionStream32.Setup provide (Mesh/Streams/
    PositionStream.cs:90-123): one interleaved 48-byte vertex stream and a uint32 index buffer."""
    VERTEX_DTYPE = np.dtype([("position", np.float32, 3), ("normal", np.float32, 3), ("tangent", np.float32, 4),
                             ("texCoord0", np.float32, 2)])

    def __init__(self):
        self.vertices = None  # DeviceTile, float32 view of the 48-byte records
        self.indices = None   # DeviceTile, uint32
        self.vertexCount = 0
        self.indexCount = 0

    def vertex_array(self):
        return self.vertices.ToArray().view(self.VERTEX_DTYPE).reshape(-1)

    def index_array(self):
        return self.indices.ToArray()


class PipelineWorkItem:  # Pipeline/Stage/PipelineDefinition.cs:18-25
    def __init__(self, data, completeAction=None, scheduledAction=None, dependency=None, stageManager=None):
        self.data = data
        self.completeAction = completeAction
        self.scheduledAction = scheduledAction
        self.dependency = dependency if dependency is not None else JobHandle()
        self.stageManager = stageManager


# ---- PipelineStage ------------------------------------------------------------------------------
class PipelineStage:  # Pipeline/Stage/PipelineStage.cs:10-62
    def __init__(self, ctx):
        self.ctx = ctx
        self.jobHandle = JobHandle()
        self.OnStageScheduledAction = []  # multicast delegate
        self.arraysInitialized = False
        self.dataLength = 0

    def ResizeNativeContainers(self, size):
        pass

    def IsSchedulable(self, job):
        return True

    def CheckRequirements(self, io_type, requirements):
        if isinstance(requirements.data, io_type):
            d = requirements.data
            if d.data.Length != self.dataLength:
                self.dataLength = d.data.Length
                self.ResizeNativeContainers(d.data.Length)
        else:
            raise Exception("Unhandled stageio %s" % type(requirements.data).__name__)

    def Schedule(self, requirements, dependency):
        pass

    def ReceiveHandledInput(self, requirements, dependency):
        self.Schedule(requirements, dependency)
        self.TransformData(requirements)
        self.OnStageScheduled(requirements, self.jobHandle)

    def Destroy(self):
        self.OnDestroy()

    def TransformData(self, data):
        pass

    def OnStageScheduled(self, requirements, dependency):
        for action in self.OnStageScheduledAction:
            action(requirements, self.jobHandle)

    def OnStageComplete(self):
        pass

    def OnDestroy(self):
        pass

    # helper shared by the stages that own one `tmp` plane (KernelFilterStage.cs:22-29 etc.)
    def _resize_tmp(self):
        if getattr(self, "tmp", None) is not None and self.tmp.IsCreated:
            self.tmp.Dispose()
        self.tmp = self.ctx.alloc(self.dataLength)


class NoiseStage(PipelineStage):  # Noise/NoiseStage.cs:13-61
    def __init__(self, ctx, noiseType=FractalNoise.Sin, hurst=0.0, startingAmplitude=1.0, octaves=1, stepdown=2.0,
                 detuneRate=0.0, noiseSize=1000):
        super().__init__(ctx)
        self.noiseType = noiseType
        self.hurst = hurst
        self.startingAmplitude = startingAmplitude
        self.octaves = octaves
        self.stepdown = stepdown
        self.detuneRate = detuneRate
        self.noiseSize = noiseSize

    def Schedule(self, requirements, dependency):
        self.CheckRequirements(GeneratorData, requirements)
        d = requirements.data
        if isinstance(d, GeneratorDataBatch):
            self.jobHandle = self.ctx.call("nz_fractal_batch", int(self.noiseType), d.data.ptr, d.resolution, d.count,
                                           d.positions.ptr, self.hurst, self.startingAmplitude, self.stepdown,
                                           self.detuneRate, self.octaves, self.noiseSize, dep=dependency)
            return
        # jobs[(int)noiseType](d.data, d.resolution, hurst, startingAmplitude, stepdown, detuneRate, octaves, ...)
        self.jobHandle = self.ctx.call("nz_fractal", int(self.noiseType), d.data.ptr, d.resolution, self.hurst,
                                       self.startingAmplitude, self.stepdown, self.detuneRate, self.octaves, d.xpos,
                                       d.zpos, self.noiseSize, dep=dependency)


class KernelFilterStage(PipelineStage):  # Filter/KernelFilterStage.cs:13-51
    def __init__(self, ctx, filter=KernelFilterType.Gauss9_S1, iterations=1):
        super().__init__(ctx)
        self.filter = filter
        self.iterations = iterations
        self.tmp = None

    def ResizeNativeContainers(self, size):
        self._resize_tmp()

    def Schedule(self, requirements, dependency):
        self.CheckRequirements(GeneratorData, requirements)
        d = requirements.data
        if d.write is not None and self.filter != KernelFilterType.Sobel3_2D:
            self.jobHandle = _call_rw(self.ctx, "nz_kernel_filter_stage_rw", d, int(self.filter), self.iterations,
                                      dep=dependency)
            return
        if isinstance(d, GeneratorDataBatch):
            self.jobHandle = self.ctx.call("nz_kernel_filter_stage_batch", d.data.ptr, self.tmp.ptr, int(self.filter),
                                           self.iterations, d.resolution, d.count, dep=dependency)
            return
        # the reference chains `iterations` SeparableKernelFilter.Schedule calls (:35-41); the
        # library fuses the chain into as few launches as halo growth allows
        self.jobHandle = self.ctx.call("nz_kernel_filter_stage", d.data.ptr, self.tmp.ptr, int(self.filter),
                                       self.iterations, d.resolution, dep=dependency)

    def OnDestroy(self):
        if self.tmp is not None and self.tmp.IsCreated:
            self.tmp.Dispose()


class StageGaussianBlur(PipelineStage):  # Filter/Kernel/Blur/StageGaussianBlur.cs:14-53
    def __init__(self, ctx, iterations=1, sigma=GaussSigma.s0d50, width=3):
        super().__init__(ctx)
        self.iterations = iterations
        self.sigma = sigma
        self.width = width
        self.tmp = None

    def ResizeNativeContainers(self, size):
        self._resize_tmp()

    def Schedule(self, requirements, dependency):
        self.CheckRequirements(GeneratorData, requirements)
        d = requirements.data
        width_ = BlurHelper.limitWidth(self.width)
        if d.write is not None:
            self.jobHandle = _call_rw(self.ctx, "nz_gauss_blur_stage_rw", d, width_, int(self.sigma), self.iterations,
                                      dep=dependency)
            return
        if isinstance(d, GeneratorDataBatch):
            self.jobHandle = self.ctx.call("nz_gauss_blur_stage_batch", d.data.ptr, self.tmp.ptr, width_, int(self.sigma),
                                           self.iterations, d.resolution, d.count, dep=dependency)
            return
        self.jobHandle = self.ctx.call("nz_gauss_blur_stage", d.data.ptr, self.tmp.ptr, width_, int(self.sigma),
                                       self.iterations, d.resolution, dep=dependency)

    def OnDestroy(self):
        if self.tmp is not None and self.tmp.IsCreated:
            self.tmp.Dispose()


class StageSmoothBlur(PipelineStage):  # Filter/Kernel/Blur/StageSmoothBlur.cs:14-52
    def __init__(self, ctx, iterations=1, width=1):
        super().__init__(ctx)
        self.iterations = iterations
        self.width = width
        self.tmp = None

    def ResizeNativeContainers(self, size):
        self._resize_tmp()

    def Schedule(self, requirements, dependency):
        self.CheckRequirements(GeneratorData, requirements)
        d = requirements.data
        width_ = BlurHelper.limitWidth(self.width)
        if d.write is not None:  # a batch too: the _rw entries honour nz_rw_tile.count
            self.jobHandle = _call_rw(self.ctx, "nz_smooth_blur_stage_rw", d, width_, self.iterations, dep=dependency)
            return
        if isinstance(d, GeneratorDataBatch):
            self.jobHandle = self.ctx.call("nz_smooth_blur_stage_batch", d.data.ptr, self.tmp.ptr, width_,
                                           self.iterations, d.resolution, d.count, dep=dependency)
            return
        self.jobHandle = self.ctx.call("nz_smooth_blur_stage", d.data.ptr, self.tmp.ptr, width_, self.iterations,
                                       d.resolution, dep=dependency)

    def OnDestroy(self):
        if self.tmp is not None and self.tmp.IsCreated:
            self.tmp.Dispose()


class ErosionStage(PipelineStage):
    """ErosionKernelJob (Filter/Kernel/KernelJob.cs:317-350) has no PipelineStage wrapper in the
    reference; this stage applies its delegate `iterations` times in KernelFilterStage's shape."""

    def __init__(self, ctx, iterations=1):
        super().__init__(ctx)
        self.iterations = iterations
        self.tmp = None

    def ResizeNativeContainers(self, size):
        self._resize_tmp()

    def Schedule(self, requirements, dependency):
        self.CheckRequirements(GeneratorData, requirements)
        d = requirements.data
        if d.write is not None:
            self.jobHandle = _call_rw(self.ctx, "nz_erosion_stage_rw", d, self.iterations, dep=dependency)
            return
        if isinstance(d, GeneratorDataBatch):
            self.jobHandle = self.ctx.call("nz_erosion_stage_batch", d.data.ptr, self.tmp.ptr, self.iterations,
                                           d.resolution, d.count, dep=dependency)
            return
        self.jobHandle = self.ctx.call("nz_erosion_stage", d.data.ptr, self.tmp.ptr, self.iterations, d.resolution,
                                       dep=dependency)

    def OnDestroy(self):
        if self.tmp is not None and self.tmp.IsCreated:
            self.tmp.Dispose()


class ConstantStage(PipelineStage):  # Filter/ConstantStage.cs:13-56
    def __init__(self, ctx, operation=ConstantOperationType.MULTIPLY, value=0.5):
        super().__init__(ctx)
        self.operation = operation
        self.value = value
        self.tmp = None

    def ResizeNativeContainers(self, size):
        self._resize_tmp()

    def Schedule(self, requirements, dependency):
        self.CheckRequirements(GeneratorData, requirements)
        d = requirements.data
        for t in _tiles_of(d):
            self.jobHandle = dependency = self.ctx.call("nz_constant_job", int(self.operation), t.ptr, self.tmp.ptr,
                                                        self.value, d.resolution, dep=dependency)

    def OnDestroy(self):
        if self.tmp is not None and self.tmp.IsCreated:
            self.tmp.Dispose()


class ReduceStage(PipelineStage):  # Filter/Reduce/ReduceStage.cs:20-69
    def __init__(self, ctx, operation=ReductionType.SUBTRACT):
        super().__init__(ctx)
        self.operation = operation
        self.tmp = None

    def ResizeNativeContainers(self, size):
        self._resize_tmp()

    def Schedule(self, requirements, dependency):
        self.CheckRequirements(ReduceData, requirements)
        d = requirements.data
        self.jobHandle = self.ctx.call("nz_reduction_job", int(self.operation), d.data.ptr, d.rightData.ptr,
                                       self.tmp.ptr, d.resolution, dep=dependency)

    def TransformData(self, inputData):  # :53-62: downstream stages see a GeneratorData
        d = inputData.data
        inputData.data = GeneratorData(d.uuid, d.data, d.resolution, d.xpos, d.zpos)

    def OnDestroy(self):
        if self.tmp is not None and self.tmp.IsCreated:
            self.tmp.Dispose()


class CurveStage(PipelineStage):  # Filter/Curve/CurveStage.cs:13-71
    """`unityCurve` is any callable t in [0,1) -> value (stands in for UnityEngine.AnimationCurve.Evaluate)."""

    def __init__(self, ctx, unityCurve=None, samples=256):
        super().__init__(ctx)
        self.unityCurve = unityCurve if unityCurve is not None else (lambda t: t)
        self.samples = samples
        self.curve = None
        self.tmp = None

    def ExtractCurve(self):  # :32-40
        host = np.array([self.unityCurve(np.float32(i) / np.float32(self.samples)) for i in range(self.samples)],
                        np.float32)
        if self.curve is None or not self.curve.IsCreated:
            self.curve = self.ctx.alloc(self.samples)
        self.curve.CopyFrom(host)
        return host

    def ResizeNativeContainers(self, size):
        if self.curve is not None and self.curve.IsCreated:
            self.curve.Dispose()
        self.curve = None
        self.ExtractCurve()
        self._resize_tmp()

    def Schedule(self, requirements, dependency):
        self.CheckRequirements(GeneratorData, requirements)
        d = requirements.data
        for t in _tiles_of(d):
            self.jobHandle = dependency = self.ctx.call("nz_curve_job", t.ptr, self.tmp.ptr, self.curve.ptr,
                                                        self.samples, d.resolution, dep=dependency)

    def OnDestroy(self):
        for t in (self.curve, self.tmp):
            if t is not None and t.IsCreated:
                t.Dispose()
        self.curve = self.tmp = None


class CropStage(PipelineStage):  # Filter/Sample/CropStage.cs:11-19
    """"CenterCropResolution" in the reference's menu; its job never sets the offset, so the crop is the
    top-left corner (Filter/Sample/CropJob.cs:43-59) -- reproduced as is."""

    def Schedule(self, requirements, dependency):
        d = requirements.data  # (DownsampleData) cast
        if not isinstance(d, DownsampleData):
            raise Exception("Unhandled stageio %s" % type(d).__name__)
        self.jobHandle = self.ctx.call("nz_crop_job", d.inputData.ptr, d.inputResolution, d.data.ptr, d.resolution,
                                       dep=dependency)


class StageThermalErosion(PipelineStage):  # Filter/Kernel/Blur/StageThermalErosion.cs:12-29
    def __init__(self, ctx, iterations=1, talus=45, increment=0.5, meshHeightWidthRatio=0.75):
        super().__init__(ctx)
        self.iterations = iterations
        self.talus = talus
        self.increment = increment
        self.meshHeightWidthRatio = meshHeightWidthRatio

    def Schedule(self, requirements, dependency):
        self.CheckRequirements(GeneratorData, requirements)
        d = requirements.data
        for t in _tiles_of(d):
            self.jobHandle = dependency = self.ctx.call("nz_thermal_erosion", t.ptr, float(self.talus), self.increment,
                                                        self.meshHeightWidthRatio, self.iterations, d.resolution,
                                                        dep=dependency)


class FlowMapStage(PipelineStage):  # Geologic/Stage/FlowMapStage.cs:16-220
    def __init__(self, ctx, iterations=5, normMin=-0.1, normMax=0.1):
        super().__init__(ctx)
        self.iterations = iterations
        self.normMin = normMin
        self.normMax = normMax
        self.resolution = 0
        self.work = None  # the stage's water/flux READ+WRITE planes (:52-62)

    def DisposeArrays(self):
        if self.work is not None and self.work.IsCreated:
            self.work.Dispose()
        self.work = None

    def ResizeNativeContainers(self, size):
        self.DisposeArrays()
        # 11 planes per tile (nz_flowmap_stage_work_floats); a batch stacks its tiles inside every plane
        self.work = self.ctx.alloc(11 * self.dataLength)

    def Schedule(self, requirements, dependency):
        d = requirements.data
        if not isinstance(d, GeneratorData):
            raise Exception("Unhandled stageio %s" % type(d).__name__)
        if self.resolution != d.resolution:
            self.resolution = d.resolution
        self.CheckRequirements(GeneratorData, requirements)
        if d.write is not None:
            self.jobHandle = _call_rw(self.ctx, "nz_flowmap_stage_rw", d, self.work.ptr, self.iterations, self.normMin,
                                      self.normMax, dep=dependency)
            return
        if isinstance(d, GeneratorDataBatch):
            self.jobHandle = self.ctx.call("nz_flowmap_stage_batch", d.data.ptr, self.work.ptr, self.iterations,
                                           self.normMin, self.normMax, d.resolution, d.count, dep=dependency)
            return
        self.jobHandle = self.ctx.call("nz_flowmap_stage", d.data.ptr, self.work.ptr, self.iterations, self.normMin,
                                       self.normMax, d.resolution, dep=dependency)

    def OnDestroy(self):
        self.DisposeArrays()


class MeshTileStage(PipelineStage):  # Mesh/Stage/MeshTileStage.cs:28-61
    def __init__(self, ctx, meshType=MeshType.SquareGridHeightMap):
        super().__init__(ctx)
        self.meshType = meshType
        self.currentMesh = None

    def Schedule(self, requirements, dependency):
        d = requirements.data  # (MeshStageData) cast
        if not isinstance(d, MeshStageData):
            raise Exception("Unhandled stageio %s" % type(d).__name__)
        self.currentMesh = d.mesh if d.mesh is not None else MeshBuffers()
        d.mesh = self.currentMesh
        m = self.currentMesh
        count = getattr(d, "count", 1)
        nv, ni = N.lib.nz_mesh_vertex_count(d.resolution) * count, N.lib.nz_mesh_index_count(d.resolution) * count
        if m.vertexCount != nv or m.vertices is None:  # Mesh.AllocateWritableMeshData(1)
            m.vertices = self.ctx.alloc(nv * 12)
            m.indices = self.ctx.alloc(ni, dtype=np.uint32)
            m.vertexCount, m.indexCount = nv, ni
        if count > 1:
            self.jobHandle = self.ctx.call("nz_heightmap_mesh_batch", int(self.meshType), m.vertices.ptr, m.indices.ptr,
                                           d.resolution, d.inputResolution, d.marginPix, d.tileHeight, d.tileSize,
                                           d.data.ptr, count, dep=dependency)
            return
        self.jobHandle = self.ctx.call("nz_heightmap_mesh", int(self.meshType), m.vertices.ptr, m.indices.ptr,
                                       d.resolution, d.inputResolution, d.marginPix, d.tileHeight, d.tileSize,
                                       d.data.ptr, dep=dependency)


# ---- BasePipeline -------------------------------------------------------------------------------
class BasePipeline:  # Pipeline/Executable/Pipeline.cs:19-287
    def __init__(self, stages, alias="Unnamed Pipeline", contextManager=None):
        self.alias = alias
        self.contextManager = contextManager  # PipelineStateManager handed to every work item (:76-104)
        self.queue = collections.deque()  # ConcurrentQueue: deque.append is thread-safe
        self.dependencyHell = []
        self.activeItem = None
        self.pipelineHandle = JobHandle()
        self.pipelineBeingScheduled = False
        self.pipelineRunning = False
        self.stage_instances = list(stages) if stages is not None else None
        self.Setup()
        self.pipeLineReady = True

    def Setup(self):  # :130-152
        if not self.stage_instances:
            return
        previous = None
        for stage in self.stage_instances:
            if previous is not None:
                previous.OnStageScheduledAction.append(stage.ReceiveHandledInput)
            previous = stage
        self.stage_instances[-1].OnStageScheduledAction.append(self.OnPipelineFullyScheduled)

    def Enqueue(self, input, scheduleAction=None, completeAction=None, dependency=None):  # :76-90
        self.queue.append(PipelineWorkItem(input, completeAction, scheduleAction, dependency, self.contextManager))

    def Schedule(self, input=None, scheduleAction=None, completeAction=None, dependency=None):  # :91-120
        if isinstance(input, PipelineWorkItem):
            self.activeItem = input
        elif input is not None:
            self.activeItem = PipelineWorkItem(input, completeAction, scheduleAction, dependency, self.contextManager)
        if not self.stage_instances:
            raise Exception("No stages in pipeline")
        self.pipelineBeingScheduled = True
        self.stage_instances[0].ReceiveHandledInput(self.activeItem, self.activeItem.dependency)

    def OnPipelineFullyScheduled(self, res, handle):  # :122-128
        self.pipelineHandle = handle
        self.pipelineRunning = True
        self.pipelineBeingScheduled = False
        if self.activeItem.scheduledAction:
            self.activeItem.scheduledAction(res.data, handle)

    def WorkIsSchedulable(self, item):  # :256-265
        ready = True
        for stage in self.stage_instances:
            ready = stage.IsSchedulable(item) and ready
        return ready

    def GetNextJob(self):  # :183-214
        for i, job in enumerate(list(self.dependencyHell)):
            if self.WorkIsSchedulable(job):
                self.dependencyHell.remove(job)
                return job
        while self.queue:
            wi = self.queue.popleft()
            if self.WorkIsSchedulable(wi):
                return wi
            self.dependencyHell.append(wi)
        return None

    def Update(self):  # :154-158,224-230
        if not self.pipelineRunning and not self.pipelineBeingScheduled:
            job = self.GetNextJob()
            if job is not None:
                self.Schedule(job)

    def _retry_is_local(self):
        """Nobody outside this pipeline has been handed a handle of the pass that is running: the work item carries no
        scheduledAction, and no stage has a scheduled-action hook beyond the hand-over to the next stage (a joint, a
        downstream pipeline's OnScheduledUpstream).  Work scheduled on such a handle has consumed the failed pass's planes
        and cannot be recalled from here."""
        own = {s.ReceiveHandledInput for s in self.stage_instances} | {self.OnPipelineFullyScheduled}
        return self.activeItem.scheduledAction is None and all(a in own for s in self.stage_instances
                                                               for a in s.OnStageScheduledAction)

    def _complete(self):
        """pipelineHandle.Complete().  NZ_ERR_RETRY -- a chained kernel-filter launch timed out, the planes computed since
        are invalid and the context has switched to separate launches -- is answered once by running the work item again,
        when (a) the pipeline regenerates its tile from scratch (its first stage is the NoiseStage) and (b) the failed pass
        is this pipeline's own business (_retry_is_local).  The failed pass is wound up first (the stages' OnStageComplete,
        as after any pass), then the item is scheduled as a retry: its dependency was satisfied by the first pass and is
        not applied again.  Any other pipeline's input is gone with the stage that failed, or its handles are in other
        hands: the error goes to the caller."""
        try:
            self.pipelineHandle.Complete()
        except N.NoizeError as e:
            if e.status != N.NZ_ERR_RETRY or not isinstance(self.stage_instances[0], NoiseStage) or not self._retry_is_local():
                raise
            self.CleanUp()
            self.pipelineRunning = False
            item = self.activeItem
            item.dependency, item.retries = JobHandle(), getattr(item, "retries", 0) + 1
            self.Schedule(item)
            self.pipelineHandle.Complete()  # (a second failure is the caller's)

    def LateUpdate(self):  # :160-181
        if self.pipelineRunning and self.pipelineHandle.IsCompleted:
            self._complete()
            self.CleanUp()
            if self.activeItem.completeAction:
                self.activeItem.completeAction(self.activeItem.data)
            self.pipelineRunning = False
            return True
        return False

    def CleanUp(self):  # :232-242
        for stage in self.stage_instances:
            stage.OnStageComplete()

    def RunToCompletion(self):
        """Drive Update/LateUpdate (Unity's frame loop) until the queue is drained."""
        while self.queue or self.dependencyHell or self.pipelineRunning:
            self.Update()
            if self.pipelineRunning:
                self._complete()
                self.LateUpdate()

    def GetDependencies(self):  # :63-65
        return [self]

    def Destroy(self):  # :244-254,267-274
        for stage in self.stage_instances:
            stage.OnDestroy()


class Upstream(enum.IntEnum):  # Pipeline/Executable/ReducePipeline.cs:27-30
    LEFT = 0
    RIGHT = 1


class PipelineJoint:  # ReducePipeline.cs:18-25
    def __init__(self, stages, action):
        self.stages = stages
        self.status = {Upstream.LEFT: False, Upstream.RIGHT: False}
        self.action = action

    @property
    def ready(self):
        return self.status[Upstream.LEFT] and self.status[Upstream.RIGHT]


class ReducePipeline(BasePipeline):  # Pipeline/Executable/ReducePipeline.cs:31-166
    """Takes one work item, requests the same tile from both upstream pipelines (the right one into a
    plane this pipeline owns) and, once both have completed, runs its own stages on a ReduceData of the
    two planes.  `ctx` allocates the right-hand plane (the reference's Persistent NativeArray)."""

    def __init__(self, ctx, stages, upstreamPipelineLeft, upstreamPipelineRight, alias="Unnamed Pipeline",
                 contextManager=None, deviceJoin=False):
        super().__init__(stages, alias, contextManager)
        self.ctx = ctx
        # deviceJoin (new-framework): the upstream pipelines may run on other contexts (HIP streams).  Instead of
        # waiting on the host for both to COMPLETE (the reference polls JobHandle.IsCompleted, :123-149), this
        # pipeline's stages are scheduled as soon as both upstreams are SCHEDULED, with
        # JobHandle.CombineDependencies(left, right) as their dependency: the join happens on the device
        # (hipStreamWaitEvent), the host never blocks between the three pipelines.
        self.deviceJoin = deviceJoin
        self._scheduled = {}
        self.upstreamPipelineLeft = upstreamPipelineLeft
        self.upstreamPipelineRight = upstreamPipelineRight
        self.upstreamsRunning = False
        self.currentWorkItem = None
        self.currentDataLength = 0
        self.rightData = None

    def GetDependencies(self):  # :52-62
        return ([self.upstreamPipelineLeft, self.upstreamPipelineRight, self] +
                self.upstreamPipelineLeft.GetDependencies() + self.upstreamPipelineRight.GetDependencies())

    def Update(self):  # OnUpdate :64-80
        if not self.pipelineRunning and not self.pipelineBeingScheduled and not self.upstreamsRunning:
            if self.queue:
                wi = self.queue.popleft()
                self.upstreamsRunning = True
                self.ScheduleUpstreams(wi)

    def ScheduleUpstreams(self, wi):  # :82-121
        leftData = wi.data
        if not isinstance(leftData, GeneratorData):
            raise Exception("Unhandled stageio %s" % type(leftData).__name__)
        if leftData.data.Length != self.currentDataLength:
            self.currentDataLength = leftData.data.Length
            if self.rightData is not None and self.rightData.IsCreated:
                self.rightData.Dispose()
            self.rightData = self.ctx.alloc(self.currentDataLength)
        if self.rightData is None or not self.rightData.IsCreated:
            self.rightData = self.ctx.alloc(self.currentDataLength)
        self.currentWorkItem = PipelineJoint(
            {Upstream.LEFT: leftData,
             Upstream.RIGHT: GeneratorData(leftData.uuid, self.rightData, leftData.resolution, leftData.xpos,
                                           leftData.zpos)},
            wi.completeAction)
        if self.deviceJoin:
            self._scheduled = {}
            self.upstreamPipelineLeft.Enqueue(self.currentWorkItem.stages[Upstream.LEFT],
                                              scheduleAction=lambda res, h: self.OnScheduledUpstream(res, h, Upstream.LEFT))
            self.upstreamPipelineRight.Enqueue(self.currentWorkItem.stages[Upstream.RIGHT],
                                               scheduleAction=lambda res, h: self.OnScheduledUpstream(res, h, Upstream.RIGHT))
            return
        self.upstreamPipelineLeft.Enqueue(self.currentWorkItem.stages[Upstream.LEFT], completeAction=self.OnCompleteLeft)
        self.upstreamPipelineRight.Enqueue(self.currentWorkItem.stages[Upstream.RIGHT],
                                           completeAction=self.OnCompleteRight)

    def OnScheduledUpstream(self, res, handle, side):
        """deviceJoin: an upstream has enqueued all its work; `handle` is its pipelineHandle."""
        self._scheduled[side] = handle
        self.currentWorkItem.status[side] = True
        self.currentWorkItem.stages[side] = res
        if self.currentWorkItem.ready:
            self.upstreamsRunning = False
            j = self.currentWorkItem
            dep = JobHandle.CombineDependencies(self.stage_instances[0].ctx, self._scheduled[Upstream.LEFT],
                                                self._scheduled[Upstream.RIGHT])
            self.Schedule(ReduceData(res.uuid, j.stages[Upstream.LEFT].data, j.stages[Upstream.RIGHT].data,
                                     res.resolution, res.xpos, res.zpos), completeAction=j.action, dependency=dep)

    def OnCompleteUpstream(self, res, side):  # :123-149
        self.currentWorkItem.status[side] = True
        self.currentWorkItem.stages[side] = res
        if self.currentWorkItem.ready:
            self.upstreamsRunning = False
            j = self.currentWorkItem
            self.Schedule(ReduceData(res.uuid, j.stages[Upstream.LEFT].data, j.stages[Upstream.RIGHT].data,
                                     res.resolution, res.xpos, res.zpos), completeAction=j.action)

    def OnCompleteLeft(self, res):
        self.OnCompleteUpstream(res, Upstream.LEFT)

    def OnCompleteRight(self, res):
        self.OnCompleteUpstream(res, Upstream.RIGHT)

    def RunToCompletion(self):
        """Unity's frame loop over this pipeline and everything upstream of it."""
        pipes = []
        for pl in self.GetDependencies():
            if pl not in pipes:
                pipes.append(pl)
        busy = True
        while busy:
            for pl in pipes:
                pl.Update()
            for pl in pipes:
                if pl.pipelineRunning:
                    pl.pipelineHandle.Complete()
                    pl.LateUpdate()
            busy = any(pl.queue or pl.dependencyHell or pl.pipelineRunning or pl.pipelineBeingScheduled or
                       getattr(pl, "upstreamsRunning", False) for pl in pipes)

    def Destroy(self):  # :157-163
        if self.rightData is not None and self.rightData.IsCreated:
            self.rightData.Dispose()
        super().Destroy()
