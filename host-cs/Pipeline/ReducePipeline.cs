// ReducePipeline.cs -- Pipeline/Executable/ReducePipeline.cs:18-166, Unity-free: one work item is requested from BOTH
// upstream pipelines (the right one into a plane this pipeline owns) and the pipeline's own stages run on a ReduceData
// of the two results.
//
// The reference joins on the HOST: each upstream's completeAction flips a flag, and the stages are scheduled once both
// have completed (:123-149).  With `deviceJoin` the join happens on the DEVICE instead: the upstreams may run on
// other GpuContexts (HIP streams); as soon as both are SCHEDULED their pipeline handles are combined
// (JobHandle.CombineDependencies -> nz_handle_combine) and passed as the dependency of this pipeline's first stage, so
// the consumer's stream waits with hipStreamWaitEvent and the host never blocks between the three pipelines.
// Source only (no .NET toolchain in the build image).
using System;
using System.Collections.Generic;

namespace xshazwar.noize.hip {

    public enum Upstream { LEFT, RIGHT }                                                           // :27-30

    public class PipelineJoint {                                                                   // :18-25
        public readonly Dictionary<Upstream, GeneratorData> stages = new Dictionary<Upstream, GeneratorData>();
        public readonly Dictionary<Upstream, bool> status = new Dictionary<Upstream, bool> { { Upstream.LEFT, false }, { Upstream.RIGHT, false } };
        public readonly Dictionary<Upstream, GpuJobHandle> scheduled = new Dictionary<Upstream, GpuJobHandle>();
        public Action<StageIO> action;
        public bool ready => status[Upstream.LEFT] && status[Upstream.RIGHT];
    }

    public class ReducePipeline : BasePipeline {
        readonly GpuContext ctx;
        public readonly BasePipeline upstreamPipelineLeft, upstreamPipelineRight;
        public readonly bool deviceJoin;
        public bool upstreamsRunning;
        PipelineJoint currentWorkItem;
        int currentDataLength;
        DeviceTile rightData;

        public ReducePipeline(GpuContext ctx, IEnumerable<PipelineStage> stages, BasePipeline left, BasePipeline right,
                              string alias = "Unnamed Pipeline", bool deviceJoin = false) : base(stages, alias) {
            this.ctx = ctx; upstreamPipelineLeft = left; upstreamPipelineRight = right; this.deviceJoin = deviceJoin;
        }

        public override void Update() {                                                            // OnUpdate :64-80
            if (!pipelineRunning && !pipelineBeingScheduled && !upstreamsRunning && queue.TryDequeue(out PipelineWorkItem wi)) {
                upstreamsRunning = true;
                ScheduleUpstreams(wi);
            }
        }

        void ScheduleUpstreams(PipelineWorkItem wi) {                                              // :82-121
            if (!(wi.data is GeneratorData leftData)) throw new Exception($"Unhandled stageio {wi.data.GetType()}");
            if (leftData.data.Length != currentDataLength || rightData == null || !rightData.IsCreated) {
                currentDataLength = leftData.data.Length;
                rightData?.Dispose();
                rightData = ctx.Alloc(currentDataLength);
            }
            currentWorkItem = new PipelineJoint { action = wi.completeAction };
            currentWorkItem.stages[Upstream.LEFT] = leftData;
            currentWorkItem.stages[Upstream.RIGHT] = new GeneratorData { uuid = leftData.uuid, data = rightData, resolution = leftData.resolution,
                                                                         xpos = leftData.xpos, zpos = leftData.zpos };
            if (deviceJoin) {
                upstreamPipelineLeft.Enqueue(currentWorkItem.stages[Upstream.LEFT], scheduleAction: (res, h) => OnScheduledUpstream(res, h, Upstream.LEFT));
                upstreamPipelineRight.Enqueue(currentWorkItem.stages[Upstream.RIGHT], scheduleAction: (res, h) => OnScheduledUpstream(res, h, Upstream.RIGHT));
            } else {
                upstreamPipelineLeft.Enqueue(currentWorkItem.stages[Upstream.LEFT], completeAction: res => OnCompleteUpstream(res, Upstream.LEFT));
                upstreamPipelineRight.Enqueue(currentWorkItem.stages[Upstream.RIGHT], completeAction: res => OnCompleteUpstream(res, Upstream.RIGHT));
            }
        }

        void OnScheduledUpstream(StageIO res, GpuJobHandle handle, Upstream side) {                // deviceJoin
            currentWorkItem.scheduled[side] = handle;
            Joined(res, side, () => GpuJobHandle.CombineDependencies(ctx, currentWorkItem.scheduled[Upstream.LEFT], currentWorkItem.scheduled[Upstream.RIGHT]));
        }

        void OnCompleteUpstream(StageIO res, Upstream side) => Joined(res, side, () => default);   // :123-149

        void Joined(StageIO res, Upstream side, Func<GpuJobHandle> dependency) {
            currentWorkItem.status[side] = true;
            currentWorkItem.stages[side] = (GeneratorData) res;
            if (!currentWorkItem.ready) return;
            upstreamsRunning = false;
            GeneratorData l = currentWorkItem.stages[Upstream.LEFT], r = currentWorkItem.stages[Upstream.RIGHT];
            Schedule(new PipelineWorkItem {
                data = new ReduceData { uuid = l.uuid, data = l.data, rightData = r.data, resolution = l.resolution, xpos = l.xpos, zpos = l.zpos },
                completeAction = currentWorkItem.action, dependency = dependency() });
        }

        public override void Destroy() {                                                           // :157-163
            rightData?.Dispose();
            rightData = null;
            base.Destroy();
        }
    }
}
