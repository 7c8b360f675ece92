// ContextStages.cs -- the stages that move a tile into / out of a named context buffer of the PipelineStateManager, and
// the mesh stage that reads its heights from one:
//   ReadGeneratorContextStage     Pipeline/PipelineState/Stage/ReadGeneratorContextStage.cs:13-46
//   WriteGeneratorContextStage    Pipeline/PipelineState/Stage/WriteGeneratorContextStage.cs:13-46
//   MeshTileReferenceDataStage    Mesh/Stage/MeshTileReferenceDataStage.cs:22-84
// FlushWriteSlice.Schedule (Pipeline/Tiles/TileData.cs:15-42) is nz_flush_write_slice: a device-to-device copy job; LockJob
// (PipelineStateLock.cs:29-39) is a marker recorded behind it (nz_handle_record).  Same behaviour as
// noize_job_amd/pipeline_state.py.  Source only (no .NET toolchain in the build image).
using System;

namespace xshazwar.noize.hip {

    public class ReadGeneratorContextStage : PipelineStage {
        public string contextAlias = "";
        public ReadGeneratorContextStage(GpuContext ctx) : base(ctx) {}
        string getBufferName(GeneratorData d) => $"{d.xpos}_{d.zpos}__{d.resolution}__{contextAlias}";      // :18-20

        // schedulable once the buffer exists and no write to it is in flight (:21-32)
        public override bool IsSchedulable(PipelineWorkItem job) {
            if (job.stageManager == null) return false;
            string name = getBufferName((GeneratorData) job.data);
            if (!job.stageManager.BufferExists(name)) return false;
            return !job.stageManager.IsLocked(name);
        }

        public override void Schedule(PipelineWorkItem requirements, GpuJobHandle dependency) {              // :33-45
            CheckRequirements<GeneratorData>(requirements);
            GeneratorData gd = (GeneratorData) requirements.data;
            int res = gd.resolution * gd.resolution;
            DeviceTile buffer = requirements.stageManager.GetBuffer(getBufferName(gd), res);
            // job(requirements.data.data, contextTarget, dependency): write_ = the work item's tile, read_ = the context buffer
            Native.Check(Native.nz_flush_write_slice(ctx.Handle, gd.data.Ptr, buffer.Ptr, (UIntPtr)(uint) res, dependency.id, out ulong h),
                         "nz_flush_write_slice");
            jobHandle = Done(h);
        }
    }

    public class WriteGeneratorContextStage : PipelineStage {
        public string contextAlias = "";
        public WriteGeneratorContextStage(GpuContext ctx) : base(ctx) {}
        string getBufferName(GeneratorData d) => $"{d.xpos}_{d.zpos}__{d.resolution}__{contextAlias}";      // :19-21

        public override bool IsSchedulable(PipelineWorkItem job) {                                            // :22-30
            if (job.stageManager == null) return false;
            return !job.stageManager.IsLocked(getBufferName((GeneratorData) job.data));
        }

        public override void Schedule(PipelineWorkItem requirements, GpuJobHandle dependency) {              // :31-45
            CheckRequirements<GeneratorData>(requirements);
            GeneratorData gd = (GeneratorData) requirements.data;
            int res = gd.resolution * gd.resolution;
            string bufferName = getBufferName(gd);
            DeviceTile buffer = requirements.stageManager.GetBuffer(bufferName, res);
            Native.Check(Native.nz_flush_write_slice(ctx.Handle, buffer.Ptr, gd.data.Ptr, (UIntPtr)(uint) res, dependency.id, out ulong h1),
                         "nz_flush_write_slice");
            Native.Check(Native.nz_handle_record(ctx.Handle, out ulong spy), "nz_handle_record");   // LockJob: a no-op behind the copy
            jobHandle = Done(spy);
            requirements.stageManager.TrySetLock(bufferName, Done(h1), jobHandle);
        }
    }

    // MeshTileStage whose heights come from the context buffer {xpos}_{zpos}__{inputResolution}__{contextAlias} instead of
    // the work item's own data; schedulable once that buffer exists and is not being written
    public class MeshTileReferenceDataStage : MeshTileStage {
        public string contextAlias = "";
        public MeshTileReferenceDataStage(GpuContext ctx) : base(ctx) {}
        string getBufferName(MeshStageData d) => $"{d.xpos}_{d.zpos}__{d.inputResolution}__{contextAlias}"; // :37-39

        public override bool IsSchedulable(PipelineWorkItem job) {                                            // :41-53
            if (job.stageManager == null) return false;
            string name = getBufferName((MeshStageData) job.data);
            if (!job.stageManager.BufferExists(name)) return false;
            return !job.stageManager.IsLocked(name);
        }

        public override void Schedule(PipelineWorkItem requirements, GpuJobHandle dependency) {              // :55-72
            if (!(requirements.data is MeshStageData d)) throw new Exception($"Unhandled stageio {requirements.data.GetType()}");
            DeviceTile buffer = requirements.stageManager.GetBuffer(getBufferName(d), d.inputResolution * d.inputResolution);
            DeviceTile own = d.data;
            d.data = buffer;                     // the mesh job reads the context buffer (:62-64)
            try { base.Schedule(requirements, dependency); }
            finally { d.data = own; }
        }
    }
}
