"""noize_job_amd -- MI355X-native (gfx950, hand-written HIP) engine for noize-job's per-cell
terrain path: fractal noise -> separable kernel filters -> flow map -> value erosion -> mesh.

Importing the package loads libnoize_hip.so through the C ABI of include/noize_hip.h and raises if
it is missing: there is no CPU or PyTorch fallback in the product path.
"""
from . import _native
from ._native import NoizeError, Stripe
from .runtime import Context, DeviceTile, JobHandle
from .pipeline import (BasePipeline, BlurHelper, ConstantOperationType, ConstantStage, CropStage, CurveStage, DownsampleData, ErosionStage, FlowMapStage, FractalNoise, GaussSigma,
                       GeneratorData, GeneratorDataBatch, KernelFilterStage, KernelFilterType, MeshBuffers, MeshStageData,
                       MeshTileStage, MeshType, NoiseStage, PipelineJoint, PipelineStage, PipelineWorkItem, ReduceData,
                       ReducePipeline, ReduceStage, Upstream,
                       ReductionType, StageGaussianBlur, StageThermalErosion,
                       StageIO, StageSmoothBlur)

from .pipeline_state import (HandleLock, MeshTileReferenceDataStage, PipelineStateManager, ReadGeneratorContextStage,
                             WriteGeneratorContextStage)

from .live_erosion import (ColorChannelByte, ErosionMode, ErosionSettings, ErosiveEvents, LiveErosion, ParticleQueue,
                           tile_set_meta)

__all__ = [n for n in dir() if not n.startswith("_")]
