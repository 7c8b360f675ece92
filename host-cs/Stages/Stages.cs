// Stages.cs -- the stage classes of the hot path with the reference's fields and Schedule bodies; where the reference
// calls a static Burst-job delegate, these call the C ABI (one extern "C" entry per delegate, include/noize_hip.h).
//   NoiseStage            Noise/NoiseStage.cs:13-61
//   KernelFilterStage     Filter/KernelFilterStage.cs:13-51
//   StageGaussianBlur     Filter/Kernel/Blur/StageGaussianBlur.cs:14-53
//   StageSmoothBlur       Filter/Kernel/Blur/StageSmoothBlur.cs:14-52
//   ErosionStage          ErosionKernelJob (Filter/Kernel/KernelJob.cs:317-350) in KernelFilterStage's shape
//   FlowMapStage          Geologic/Stage/FlowMapStage.cs:16-220
//   MeshTileStage         Mesh/Stage/MeshTileStage.cs:28-61
//   ConstantStage / ReduceStage / CurveStage   Filter/ConstantStage.cs, Filter/Reduce/ReduceStage.cs, Filter/Curve/CurveStage.cs
//   CropStage             Filter/Sample/CropStage.cs:11-19
//   StageThermalErosion   Filter/Kernel/Blur/StageThermalErosion.cs:12-29
// Source only (no .NET toolchain in the build image).
using System;

namespace xshazwar.noize.hip {

    public enum FractalNoise { Sin, Perlin, PeriodicPerlin, Simplex, RotatedSimplex, Cellular, DomainRotatedPerlin, DomainRotatedSimplex }  // NoiseStage.cs:15-24
    public enum KernelFilterType { Gauss9_S1, Gauss7_S1, Gauss5_S1, Gauss3_S1, Gauss9_S2, Gauss7_S2, Gauss5_S2, Gauss3_S2, Smooth3,
                                   Sobel3Horizontal, Sobel3Vertical, Sobel3_2D, Prewitt3Horizontal, Prewitt3Vertical }                  // KernelJob.cs:79-94
    public enum GaussSigma { s0d50, s1d00, s1d50, s2d00, s2d50, s3d00, s3d50, s4d00, s4d50, s5d00, s5d50, s6d00, s6d50, s7d00, s7d50, s8d00 } // BlurKernels.cs:8-25
    public enum MeshType { SquareGridHeightMap, OvershootSquareGridHeightMap }                                                           // MeshTileStage.cs:23-26
    public enum ConstantOperationType { MULTIPLY, BINARIZE }                                                                             // ConstantStage.cs:15-18
    public enum ReductionType { SUBTRACT, MULTIPLY, ROOTSUMSQUARES, MAX, MIN }                                                           // ReduceStage.cs:12-18

    public static class BlurHelper {             // BlurKernels.cs:27-37
        public const int max_width = 25;
        public static int limitWidth(int width) {
            if (width % 2 == 0) width += 1;
            width = Math.Min(width, max_width);
            return Math.Max(3, width);
        }
    }

    public abstract class TmpStage : PipelineStage {   // the `tmp` NativeArray the filter stages own (KernelFilterStage.cs:22-29)
        protected DeviceTile tmp;
        protected static int BatchCount(GeneratorData d) => d is GeneratorDataBatch b ? b.count : 1;
        protected TmpStage(GpuContext ctx) : base(ctx) {}
        public override void ResizeNativeContainers(int size) { tmp?.Dispose(); tmp = ctx.Alloc(size); }
        public override void OnDestroy() { tmp?.Dispose(); tmp = null; }
    }

    public class NoiseStage : PipelineStage {
        public FractalNoise noiseType = FractalNoise.Sin;
        public float hurst = 0f, startingAmplitude = 1f, stepdown = 2f, detuneRate = 0f;
        public int octaves = 1, noiseSize = 1000;
        public NoiseStage(GpuContext ctx) : base(ctx) {}
        public override void Schedule(PipelineWorkItem requirements, GpuJobHandle dependency) {     // :55-60
            CheckRequirements<GeneratorData>(requirements);
            GeneratorData d = (GeneratorData) requirements.data;
            if (d is GeneratorDataBatch b) {      // `count` tiles, one launch
                Native.Check(Native.nz_fractal_batch(ctx.Handle, (int) noiseType, b.data.Ptr, b.resolution, b.count, b.positions.Ptr, hurst,
                                                     startingAmplitude, stepdown, detuneRate, octaves, noiseSize, dependency.id, out ulong hb), "nz_fractal_batch");
                jobHandle = Done(hb);
                return;
            }
            // jobs[(int) noiseType](d.data, d.resolution, hurst, startingAmplitude, stepdown, detuneRate, octaves, d.xpos, d.zpos, noiseSize, dependency)
            Native.Check(Native.nz_fractal(ctx.Handle, (int) noiseType, d.data.Ptr, d.resolution, hurst, startingAmplitude, stepdown,
                                           detuneRate, octaves, d.xpos, d.zpos, noiseSize, dependency.id, out ulong h), "nz_fractal");
            jobHandle = Done(h);
        }
    }

    public class KernelFilterStage : TmpStage {
        public KernelFilterType filter = KernelFilterType.Gauss9_S1;
        public int iterations = 1;
        public KernelFilterStage(GpuContext ctx) : base(ctx) {}
        public override void Schedule(PipelineWorkItem requirements, GpuJobHandle dependency) {     // :31-43
            CheckRequirements<GeneratorData>(requirements);
            GeneratorData d = (GeneratorData) requirements.data;
            ulong h;
            if (d.write != null && filter != KernelFilterType.Sobel3_2D) {
                NzRwTile t = new NzRwTile { read = d.data.Ptr, write = d.write.Ptr, resolution = d.resolution, count = BatchCount(d) };
                Native.Check(Native.nz_kernel_filter_stage_rw(ctx.Handle, ref t, (int) filter, iterations, dependency.id, out h), "nz_kernel_filter_stage_rw");
                Adopt(d, t);
            } else if (d is GeneratorDataBatch b) {
                Native.Check(Native.nz_kernel_filter_stage_batch(ctx.Handle, b.data.Ptr, tmp.Ptr, (int) filter, iterations, b.resolution, b.count, dependency.id, out h), "nz_kernel_filter_stage_batch");
            } else {
                // the reference chains `iterations` SeparableKernelFilter.Schedule calls (:35-41); the library fuses the chain
                Native.Check(Native.nz_kernel_filter_stage(ctx.Handle, d.data.Ptr, tmp.Ptr, (int) filter, iterations, d.resolution, dependency.id, out h), "nz_kernel_filter_stage");
            }
            jobHandle = Done(h);
        }
    }

    public class StageGaussianBlur : TmpStage {
        public int iterations = 1, width = 3;
        public GaussSigma sigma = GaussSigma.s0d50;
        public StageGaussianBlur(GpuContext ctx) : base(ctx) {}
        public override void Schedule(PipelineWorkItem requirements, GpuJobHandle dependency) {
            CheckRequirements<GeneratorData>(requirements);
            GeneratorData d = (GeneratorData) requirements.data;
            int width_ = BlurHelper.limitWidth(width);
            ulong h;
            if (d.write != null) {
                NzRwTile t = new NzRwTile { read = d.data.Ptr, write = d.write.Ptr, resolution = d.resolution, count = BatchCount(d) };
                Native.Check(Native.nz_gauss_blur_stage_rw(ctx.Handle, ref t, width_, (int) sigma, iterations, dependency.id, out h), "nz_gauss_blur_stage_rw");
                Adopt(d, t);
            } else if (d is GeneratorDataBatch b) {
                Native.Check(Native.nz_gauss_blur_stage_batch(ctx.Handle, b.data.Ptr, tmp.Ptr, width_, (int) sigma, iterations, b.resolution, b.count, dependency.id, out h), "nz_gauss_blur_stage_batch");
            } else {
                Native.Check(Native.nz_gauss_blur_stage(ctx.Handle, d.data.Ptr, tmp.Ptr, width_, (int) sigma, iterations, d.resolution, dependency.id, out h), "nz_gauss_blur_stage");
            }
            jobHandle = Done(h);
        }
    }

    public class StageSmoothBlur : TmpStage {
        public int iterations = 1, width = 1;
        public StageSmoothBlur(GpuContext ctx) : base(ctx) {}
        public override void Schedule(PipelineWorkItem requirements, GpuJobHandle dependency) {
            CheckRequirements<GeneratorData>(requirements);
            GeneratorData d = (GeneratorData) requirements.data;
            int width_ = BlurHelper.limitWidth(width);
            ulong h;
            if (d.write != null) {
                NzRwTile t = new NzRwTile { read = d.data.Ptr, write = d.write.Ptr, resolution = d.resolution, count = BatchCount(d) };
                Native.Check(Native.nz_smooth_blur_stage_rw(ctx.Handle, ref t, width_, iterations, dependency.id, out h), "nz_smooth_blur_stage_rw");
                Adopt(d, t);
            } else if (d is GeneratorDataBatch b) {
                Native.Check(Native.nz_smooth_blur_stage_batch(ctx.Handle, b.data.Ptr, tmp.Ptr, width_, iterations, b.resolution, b.count, dependency.id, out h), "nz_smooth_blur_stage_batch");
            } else {
                Native.Check(Native.nz_smooth_blur_stage(ctx.Handle, d.data.Ptr, tmp.Ptr, width_, iterations, d.resolution, dependency.id, out h), "nz_smooth_blur_stage");
            }
            jobHandle = Done(h);
        }
    }

    public class ErosionStage : TmpStage {
        public int iterations = 1;
        public ErosionStage(GpuContext ctx) : base(ctx) {}
        public override void Schedule(PipelineWorkItem requirements, GpuJobHandle dependency) {
            CheckRequirements<GeneratorData>(requirements);
            GeneratorData d = (GeneratorData) requirements.data;
            ulong h;
            if (d.write != null) {
                NzRwTile t = new NzRwTile { read = d.data.Ptr, write = d.write.Ptr, resolution = d.resolution, count = BatchCount(d) };
                Native.Check(Native.nz_erosion_stage_rw(ctx.Handle, ref t, iterations, dependency.id, out h), "nz_erosion_stage_rw");
                Adopt(d, t);
            } else if (d is GeneratorDataBatch b) {
                Native.Check(Native.nz_erosion_stage_batch(ctx.Handle, b.data.Ptr, tmp.Ptr, iterations, b.resolution, b.count, dependency.id, out h), "nz_erosion_stage_batch");
            } else {
                Native.Check(Native.nz_erosion_stage(ctx.Handle, d.data.Ptr, tmp.Ptr, iterations, d.resolution, dependency.id, out h), "nz_erosion_stage");
            }
            jobHandle = Done(h);
        }
    }

    public class CropStage : PipelineStage {      // "CenterCropResolution": the job never sets its offset, so the crop is top-left (CropJob.cs:43-59)
        public CropStage(GpuContext ctx) : base(ctx) {}
        public override void Schedule(PipelineWorkItem requirements, GpuJobHandle dependency) {
            if (!(requirements.data is DownsampleData d)) throw new Exception($"Unhandled stageio {requirements.data.GetType()}");
            Native.Check(Native.nz_crop_job(ctx.Handle, d.inputData.Ptr, d.inputResolution, d.data.Ptr, d.resolution, dependency.id, out ulong h), "nz_crop_job");
            jobHandle = Done(h);
        }
    }

    public class StageThermalErosion : PipelineStage {
        public int iterations = 1;
        public float talus = 45f, increment = 0.5f, meshHeightWidthRatio = 0.75f;
        public StageThermalErosion(GpuContext ctx) : base(ctx) {}
        public override void Schedule(PipelineWorkItem requirements, GpuJobHandle dependency) {     // :20-28
            CheckRequirements<GeneratorData>(requirements);
            GeneratorData d = (GeneratorData) requirements.data;
            Native.Check(Native.nz_thermal_erosion(ctx.Handle, d.data.Ptr, talus, increment, meshHeightWidthRatio, iterations, d.resolution,
                                                   dependency.id, out ulong h), "nz_thermal_erosion");
            jobHandle = Done(h);
        }
    }

    public class FlowMapStage : PipelineStage {
        public int iterations = 5;
        public float normMin = -0.1f, normMax = 0.1f;
        DeviceTile work;                         // the stage's water / flux READ + WRITE planes (:52-62)
        public FlowMapStage(GpuContext ctx) : base(ctx) {}
        public override void ResizeNativeContainers(int size) {                                     // :197-205
            work?.Dispose();
            work = ctx.Alloc(11 * size);         // the stage's 11 planes (nz_flowmap_stage_work_floats); a batch stacks its tiles inside every plane
        }
        public override void Schedule(PipelineWorkItem requirements, GpuJobHandle dependency) {     // :207-214 -> ScheduleAll :124-195
            CheckRequirements<GeneratorData>(requirements);
            GeneratorData d = (GeneratorData) requirements.data;
            ulong h;
            if (d.write != null) {
                NzRwTile t = new NzRwTile { read = d.data.Ptr, write = d.write.Ptr, resolution = d.resolution, count = d is GeneratorDataBatch bb ? bb.count : 1 };
                Native.Check(Native.nz_flowmap_stage_rw(ctx.Handle, ref t, work.Ptr, iterations, normMin, normMax, dependency.id, out h), "nz_flowmap_stage_rw");
                Adopt(d, t);
            } else if (d is GeneratorDataBatch b) {
                Native.Check(Native.nz_flowmap_stage_batch(ctx.Handle, b.data.Ptr, work.Ptr, iterations, normMin, normMax, b.resolution, b.count, dependency.id, out h), "nz_flowmap_stage_batch");
            } else {
                Native.Check(Native.nz_flowmap_stage(ctx.Handle, d.data.Ptr, work.Ptr, iterations, normMin, normMax, d.resolution, dependency.id, out h), "nz_flowmap_stage");
            }
            jobHandle = Done(h);
        }
        public override void OnDestroy() { work?.Dispose(); work = null; }                          // :216-219
    }

    public class MeshTileStage : PipelineStage {
        public MeshType meshType = MeshType.SquareGridHeightMap;
        public MeshTileStage(GpuContext ctx) : base(ctx) {}
        public override void Schedule(PipelineWorkItem requirements, GpuJobHandle dependency) {     // :40-46
            MeshStageData d = (MeshStageData) requirements.data;
            int count = Math.Max(1, d.count);
            int nv = (int) (ulong) Native.nz_mesh_vertex_count(d.resolution) * count, ni = (int) (ulong) Native.nz_mesh_index_count(d.resolution) * count;
            if (d.vertices == null || d.vertices.Length != nv * 12) {   // Mesh.AllocateWritableMeshData(1)
                d.vertices?.Dispose(); d.indices?.Dispose();
                d.vertices = ctx.Alloc(nv * 12);                         // 48-byte records {pos3, normal3, tangent4, uv2}
                d.indices = ctx.Alloc(ni);                               // uint32
            }
            ulong h;
            if (count > 1)
                Native.Check(Native.nz_heightmap_mesh_batch(ctx.Handle, (int) meshType, d.vertices.Ptr, d.indices.Ptr, d.resolution, d.inputResolution,
                                                            d.marginPix, d.tileHeight, d.tileSize, d.data.Ptr, count, dependency.id, out h), "nz_heightmap_mesh_batch");
            else
                Native.Check(Native.nz_heightmap_mesh(ctx.Handle, (int) meshType, d.vertices.Ptr, d.indices.Ptr, d.resolution, d.inputResolution,
                                                      d.marginPix, d.tileHeight, d.tileSize, d.data.Ptr, dependency.id, out h), "nz_heightmap_mesh");
            jobHandle = Done(h);
        }
        // OnStageComplete (:48-57): Mesh.ApplyAndDisposeWritableMeshData -> copy d.vertices / d.indices into the engine's mesh
    }

    public class ConstantStage : TmpStage {
        public ConstantOperationType operation = ConstantOperationType.MULTIPLY;
        public float value = 0.5f;
        public ConstantStage(GpuContext ctx) : base(ctx) {}
        public override void Schedule(PipelineWorkItem requirements, GpuJobHandle dependency) {
            CheckRequirements<GeneratorData>(requirements);
            GeneratorData d = (GeneratorData) requirements.data;
            Native.Check(Native.nz_constant_job(ctx.Handle, (int) operation, d.data.Ptr, tmp.Ptr, value, d.resolution, dependency.id, out ulong h), "nz_constant_job");
            jobHandle = Done(h);
        }
    }

    public class ReduceStage : TmpStage {
        public ReductionType operation = ReductionType.SUBTRACT;
        public ReduceStage(GpuContext ctx) : base(ctx) {}
        public override void Schedule(PipelineWorkItem requirements, GpuJobHandle dependency) {
            CheckRequirements<ReduceData>(requirements);
            ReduceData d = (ReduceData) requirements.data;
            Native.Check(Native.nz_reduction_job(ctx.Handle, (int) operation, d.data.Ptr, d.rightData.Ptr, tmp.Ptr, d.resolution, dependency.id, out ulong h), "nz_reduction_job");
            jobHandle = Done(h);
        }
        public override void TransformData(PipelineWorkItem inputData) {                            // ReduceStage.cs:53-62
            ReduceData d = (ReduceData) inputData.data;
            inputData.data = new GeneratorData { uuid = d.uuid, data = d.data, resolution = d.resolution, xpos = d.xpos, zpos = d.zpos };
        }
    }

    public class CurveStage : TmpStage {
        public Func<float, float> unityCurve = t => t;   // stands in for UnityEngine.AnimationCurve.Evaluate
        public int samples = 256;
        DeviceTile curve;
        public CurveStage(GpuContext ctx) : base(ctx) {}
        public override void ResizeNativeContainers(int size) {                                      // CurveStage.cs:26-40
            base.ResizeNativeContainers(size);
            curve?.Dispose();
            curve = ctx.Alloc(samples);
            float[] host = new float[samples];
            for (int i = 0; i < samples; i++) host[i] = unityCurve((float) i / (float) samples);
            curve.CopyFrom(host);
        }
        public override void Schedule(PipelineWorkItem requirements, GpuJobHandle dependency) {
            CheckRequirements<GeneratorData>(requirements);
            GeneratorData d = (GeneratorData) requirements.data;
            Native.Check(Native.nz_curve_job(ctx.Handle, d.data.Ptr, tmp.Ptr, curve.Ptr, samples, d.resolution, dependency.id, out ulong h), "nz_curve_job");
            jobHandle = Done(h);
        }
        public override void OnDestroy() { base.OnDestroy(); curve?.Dispose(); curve = null; }
    }
}
