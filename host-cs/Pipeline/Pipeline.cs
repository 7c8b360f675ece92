// Pipeline.cs -- the reference's operator API, Unity-free: StageIO payloads, PipelineWorkItem, PipelineStage and
// BasePipeline with the same members and call order as
//   Pipeline/Stage/StageIO.cs:8-11, Pipeline/Stage/StageIOTypes/*.cs, Pipeline/Stage/PipelineDefinition.cs:18-25,
//   Pipeline/Stage/PipelineStage.cs:10-62, Pipeline/Executable/Pipeline.cs:19-287,
// so stage graphs written against the reference (NoiseStage -> KernelFilterStage -> FlowMapStage -> MeshTileStage)
// keep their shape; NativeSlice<float> is a DeviceTile, JobHandle a GpuJobHandle.  Source only (no .NET toolchain in
// the build image).
using System;
using System.Collections.Concurrent;
using System.Collections.Generic;

namespace xshazwar.noize.hip {

    public class StageIO {                       // StageIO.cs:8-11
        public string uuid;
        public DeviceTile data;
    }

    public class GeneratorData : StageIO {       // StageIOTypes/GeneratorData.cs:9-15
        public int resolution, xpos, zpos;
        // new: the WRITE slice of the tile's RWTileData pair (Pipeline/Tiles/TileData.cs:49-93).  When set, the stencil
        // stages run their nz_*_rw forms and TileHelpers.SWAP_RWTILE is a swap of `data` and `write`.
        public DeviceTile write;
    }

    // New-framework payload: `count` independent tiles of resolution^2 cells stored back to back in `data` (tile k at offset
    // k * resolution^2), world positions in `positions` (device int32 pairs {xpos, zpos} per tile).  The noise / filter / blur /
    // erosion / flow-map stages run such a batch through ONE launch sequence (nz_*_batch); every tile comes out exactly as it
    // would alone.  (The reference runs one BasePipeline per tile request, Scripts/MeshTileGenerator.cs:181-211: 512^2 tiles
    // one at a time cannot fill 256 CUs.)
    public class GeneratorDataBatch : GeneratorData {
        public DeviceTile positions;
        public int count = 1;

        public static GeneratorDataBatch Create(GpuContext ctx, string uuid, int resolution, int[] positionsXZ) {
            int n = positionsXZ.Length / 2;
            GeneratorDataBatch b = new GeneratorDataBatch { uuid = uuid, resolution = resolution, count = n,
                                                            data = ctx.Alloc(n * resolution * resolution), positions = ctx.Alloc(2 * n) };
            b.positions.CopyFrom(positionsXZ);
            return b;
        }
        public DeviceTile tile(int k) => data.Offset(k * resolution * resolution, resolution * resolution);
    }

    public class MeshStageData : StageIO {       // StageIOTypes/MeshStageData.cs:9-21
        public int resolution, inputResolution, marginPix, xpos, zpos;
        public float tileSize, tileHeight;
        public DeviceTile vertices, indices;     // stand in for UnityEngine.Mesh: 48-byte Stream0 records, uint indices
        public int count = 1;                    // new: `count` height planes stored back to back -> `count` meshes, one launch
    }

    public class ReduceData : StageIO {          // StageIOTypes/ReduceData.cs:9-17
        public DeviceTile rightData;
        public int resolution, xpos, zpos;
    }

    public class DownsampleData : StageIO {      // StageIOTypes/DownsampleData.cs:9-17
        public DeviceTile inputData;
        public int resolution, inputResolution;
    }

    public class PipelineWorkItem {              // PipelineDefinition.cs:18-25
        public StageIO data;
        public Action<StageIO> completeAction;
        public Action<StageIO, GpuJobHandle> scheduledAction;
        public GpuJobHandle dependency;
        public PipelineStateManager stageManager;    // the context buffers this item's stages may read / write (:24)
    }

    public abstract class PipelineStage {        // PipelineStage.cs:10-62
        protected readonly GpuContext ctx;
        public GpuContext Context => ctx;
        public GpuJobHandle jobHandle;
        public Action<PipelineWorkItem, GpuJobHandle> OnStageScheduledAction;
        protected int dataLength = 0;
        protected PipelineStage(GpuContext ctx) { this.ctx = ctx; }

        public virtual void ResizeNativeContainers(int size) {}
        public virtual bool IsSchedulable(PipelineWorkItem job) => true;

        protected void CheckRequirements<T>(PipelineWorkItem requirements) where T : StageIO {   // :29-39
            if (requirements.data is T d) {
                if (d.data.Length != dataLength) { dataLength = d.data.Length; ResizeNativeContainers(d.data.Length); }
            } else {
                throw new Exception($"Unhandled stageio {requirements.data.GetType()}");
            }
        }

        public abstract void Schedule(PipelineWorkItem requirements, GpuJobHandle dependency);
        public virtual void TransformData(PipelineWorkItem data) {}

        public void ReceiveHandledInput(PipelineWorkItem requirements, GpuJobHandle dependency) {  // :41-48
            Schedule(requirements, dependency);
            TransformData(requirements);
            OnStageScheduledAction?.Invoke(requirements, jobHandle);
        }

        public virtual void OnStageComplete() {}
        public virtual void OnDestroy() {}
        protected GpuJobHandle Done(ulong h) => ctx.Wrap(h);

        // adopt the READ / WRITE pair as an nz_*_rw entry left it
        protected static void Adopt(GeneratorData d, NzRwTile t) {
            if (t.read != d.data.Ptr) { DeviceTile s = d.data; d.data = d.write; d.write = s; }
        }
    }

    public class BasePipeline {                  // Pipeline.cs:19-287
        public string alias = "Unnamed Pipeline";
        protected readonly List<PipelineStage> stage_instances;
        protected readonly ConcurrentQueue<PipelineWorkItem> queue = new ConcurrentQueue<PipelineWorkItem>();
        protected readonly List<PipelineWorkItem> dependencyHell = new List<PipelineWorkItem>();
        protected PipelineWorkItem activeItem;
        public GpuJobHandle pipelineHandle;
        public bool pipelineRunning, pipelineBeingScheduled;

        public PipelineStateManager contextManager;  // handed to every work item (Pipeline.cs:76-104)

        public BasePipeline(IEnumerable<PipelineStage> stages, string alias = "Unnamed Pipeline", PipelineStateManager contextManager = null) {
            this.alias = alias;
            this.contextManager = contextManager;
            stage_instances = new List<PipelineStage>(stages);
            Setup();
        }

        void Setup() {                           // :130-152
            PipelineStage previous = null;
            foreach (PipelineStage stage in stage_instances) {
                if (previous != null) previous.OnStageScheduledAction += stage.ReceiveHandledInput;
                previous = stage;
            }
            if (previous != null) previous.OnStageScheduledAction += OnPipelineFullyScheduled;
        }

        public void Enqueue(StageIO input, Action<StageIO, GpuJobHandle> scheduleAction = null, Action<StageIO> completeAction = null,
                            GpuJobHandle dependency = default) {                                  // :76-90
            queue.Enqueue(new PipelineWorkItem { data = input, completeAction = completeAction, scheduledAction = scheduleAction,
                                                 dependency = dependency, stageManager = contextManager });
        }

        public void Schedule(PipelineWorkItem item) {                                               // :104-120
            activeItem = item;
            if (stage_instances.Count == 0) throw new Exception("No stages in pipeline");
            pipelineBeingScheduled = true;
            stage_instances[0].ReceiveHandledInput(activeItem, activeItem.dependency);
        }

        // The stock stage list NoiseStage -> [KernelFilterStage] -> [FlowMapStage] -> [ErosionStage] (README.md:23-32), all on
        // one context, as nz_terrain_params (what ShardedPipeline hands to nz_sharded_create); false if the list is anything else.
        public static bool StockListParams(IList<PipelineStage> stages, out NzTerrainParams tp, out NoiseStage n) {
            tp = default(NzTerrainParams); n = null;
            if (stages.Count == 0 || stages[0].GetType() != typeof(NoiseStage)) return false;
            n = (NoiseStage) stages[0];
            KernelFilterStage f = null; FlowMapStage w = null; ErosionStage e = null;
            int k = 0;                           // the optional stages in this order, each at most once
            for (int i = 1; i < stages.Count; i++) {
                PipelineStage s = stages[i];
                if (!ReferenceEquals(s.Context, n.Context)) return false;
                if (k < 1 && s.GetType() == typeof(KernelFilterStage)) { f = (KernelFilterStage) s; k = 1; }
                else if (k < 2 && s.GetType() == typeof(FlowMapStage)) { w = (FlowMapStage) s; k = 2; }
                else if (k < 3 && s.GetType() == typeof(ErosionStage)) { e = (ErosionStage) s; k = 3; }
                else return false;
            }
            if (f != null && f.filter == KernelFilterType.Sobel3_2D) return false;
            tp = new NzTerrainParams {
                noiseType = (int) n.noiseType, hurst = n.hurst, startingAmplitude = n.startingAmplitude, stepdown = n.stepdown,
                detuneRate = n.detuneRate, octaves = n.octaves, noiseSize = n.noiseSize,
                filter = f != null ? (int) f.filter : 0, filterIterations = f != null ? f.iterations : 0,
                flowIterations = w != null ? w.iterations : 0, normMin = w != null ? w.normMin : 0f, normMax = w != null ? w.normMax : 0f,
                erosionIterations = e != null ? e.iterations : 0 };
            return true;
        }

        void OnPipelineFullyScheduled(PipelineWorkItem res, GpuJobHandle handle) {                  // :122-128
            pipelineHandle = handle;
            pipelineRunning = true;
            pipelineBeingScheduled = false;
            activeItem.scheduledAction?.Invoke(res.data, handle);
        }

        bool WorkIsSchedulable(PipelineWorkItem item) {                                             // :256-265
            bool ready = true;
            foreach (PipelineStage stage in stage_instances) ready = stage.IsSchedulable(item) && ready;
            return ready;
        }

        PipelineWorkItem GetNextJob() {                                                             // :183-214
            for (int i = 0; i < dependencyHell.Count; i++)
                if (WorkIsSchedulable(dependencyHell[i])) { PipelineWorkItem j = dependencyHell[i]; dependencyHell.RemoveAt(i); return j; }
            while (queue.TryDequeue(out PipelineWorkItem wi)) {
                if (WorkIsSchedulable(wi)) return wi;
                dependencyHell.Add(wi);
            }
            return null;
        }

        public virtual void Update() {                                                                      // :154-158,224-230
            if (!pipelineRunning && !pipelineBeingScheduled) {
                PipelineWorkItem job = GetNextJob();
                if (job != null) Schedule(job);
            }
        }

        // Nobody outside this pipeline has been handed a handle of the pass that is running: the work item carries no
        // scheduledAction and no stage has a scheduled-action hook beyond its hand-over to the next stage (a joint, a
        // downstream pipeline).  Work scheduled on such a handle has consumed the failed pass's planes and cannot be recalled.
        bool RetryIsLocal() {
            if (activeItem.scheduledAction != null) return false;
            foreach (PipelineStage s in stage_instances)
                if (s.OnStageScheduledAction == null || s.OnStageScheduledAction.GetInvocationList().Length != 1) return false;
            return true;
        }

        // pipelineHandle.Complete().  NZ_ERR_RETRY -- a chained kernel-filter launch timed out, the planes computed since are
        // invalid and the context has switched to separate launches -- is answered once by running the work item again, when
        // the pipeline regenerates its tile from scratch (its first stage is the NoiseStage) and the failed pass is this
        // pipeline's own business (RetryIsLocal).  The failed pass is wound up first (OnStageComplete, as after any pass); the
        // item's dependency was satisfied by the first pass and is not applied again.  Otherwise the error goes to the caller.
        void CompleteActive() {
            try {
                pipelineHandle.Complete();
            } catch (NoizeException e) when (e.status == Native.NZ_ERR_RETRY && stage_instances[0].GetType() == typeof(NoiseStage)
                                             && RetryIsLocal()) {
                foreach (PipelineStage stage in stage_instances) stage.OnStageComplete();
                pipelineRunning = false;
                activeItem.dependency = default(GpuJobHandle);
                Schedule(activeItem);
                pipelineHandle.Complete();  // (a second failure is the caller's)
            }
        }

        public bool LateUpdate() {                                                                  // :160-181
            if (pipelineRunning && pipelineHandle.IsCompleted) {
                CompleteActive();
                foreach (PipelineStage stage in stage_instances) stage.OnStageComplete();
                activeItem.completeAction?.Invoke(activeItem.data);
                pipelineRunning = false;
                return true;
            }
            return false;
        }

        public virtual void Destroy() { foreach (PipelineStage stage in stage_instances) stage.OnDestroy(); }  // :244-254
    }
}
