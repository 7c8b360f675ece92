// ShardedPipeline.cs -- one large grid over the GPUs of a node (new-framework feature; include/noize_hip.h, nz_comm.cpp).
// The reference's host asks for one tile at a time (BasePipeline.Schedule, Pipeline/Executable/Pipeline.cs:104-128;
// Scripts/MeshTileGenerator.cs:166-192); a grid larger than a tile is cut into row stripes, one process per GPU, and the
// stock stage list NoiseStage -> [KernelFilterStage] -> [FlowMapStage] -> [ErosionStage] runs on all of them, ghost rows
// exchanged with the neighbour ranks over RCCL (ncclSend / ncclRecv inside libnoize_hip.so).  Source only (no .NET
// toolchain in the build image); noize_job_amd/host/noize_pipeline.hpp holds the same two classes in C++, and
// `host_demo sharded` / `host_demo sharded-rank` run them.
using System;
using System.Collections.Generic;

namespace xshazwar.noize.hip {

    public enum HaloMode { Recompute = 0, Exchange = 1, ExchangeOnce = 2 }   // enum nz_halo_mode

    // nz_comm: ncclCommInitRank on the context's device + the exchanges' own stream
    public sealed class GpuComm : IDisposable {
        public const int IdBytes = 128;           // NZ_COMM_ID_BYTES = sizeof(ncclUniqueId)
        public IntPtr Handle { get; private set; }
        public readonly GpuContext ctx;

        // by ONE rank; the bytes travel to the other ranks out of band (a file, a socket, the launcher's store)
        public static byte[] UniqueId() {
            byte[] id = new byte[IdBytes];
            Native.Check(Native.nz_comm_unique_id(id), "nz_comm_unique_id");
            return id;
        }

        public GpuComm(GpuContext ctx, byte[] id, int rank, int world) {
            if (id == null || id.Length != IdBytes) throw new ArgumentException("an ncclUniqueId is 128 bytes");
            this.ctx = ctx;
            Native.Check(Native.nz_comm_init(ctx.Handle, id, rank, world, out IntPtr h), "nz_comm_init");   // blocks until all ranks joined
            Handle = h;
        }
        public int Rank => Native.nz_comm_rank(Handle);
        public int World => Native.nz_comm_world(Handle);
        // The library refuses while ShardedPipelines still hold the communicator (dispose them first): the status is thrown
        // and the handle KEPT, so that a later Dispose() still reaches ncclCommDestroy.
        public void Dispose() {
            if (Handle == IntPtr.Zero) return;
            Native.Check(Native.nz_comm_destroy(Handle), "nz_comm_destroy");
            Handle = IntPtr.Zero;
        }
    }

    public sealed class ShardedPipeline : IDisposable {
        public readonly GpuContext ctx;
        public IntPtr Handle { get; private set; }

        // `stages`: the stock list (BasePipeline.StockListParams); comm may be null on one rank (device copies instead of
        // RCCL); stripes = 0: one per rank; overlap: 0 transfers on the compute stream between the launches (fastest on
        // MI355X), 1 interior rows first, 2 border rows first
        public ShardedPipeline(GpuContext ctx, GpuComm comm, IList<PipelineStage> stages, int grows, int cols, int stripes = 0,
                               HaloMode haloMode = HaloMode.Exchange, int overlap = 0, int xpos = 0, int zpos = 0) {
            this.ctx = ctx;
            if (!BasePipeline.StockListParams(stages, out NzTerrainParams tp, out NoiseStage _))
                throw new Exception("ShardedPipeline: not the stock stage list");
            NzShardedDesc d = new NzShardedDesc {
                grows = grows, cols = cols, stripes = stripes > 0 ? stripes : (comm != null ? comm.World : 1),
                haloMode = (int) haloMode, overlap = overlap, xpos = xpos, zpos = zpos };
            Native.Check(Native.nz_sharded_create(ctx.Handle, comm != null ? comm.Handle : IntPtr.Zero, ref d, ref tp, out IntPtr h),
                         "nz_sharded_create");
            Handle = h;
        }

        public int LocalStripes => Native.nz_sharded_local_stripes(Handle);

        // one pass over every local stripe (enqueue only): what BasePipeline.Schedule is for a tile
        public GpuJobHandle Schedule(GpuJobHandle dependency = default(GpuJobHandle)) {
            Native.Check(Native.nz_sharded_pipeline(ctx.Handle, Handle, null, dependency.id, out ulong h), "nz_sharded_pipeline");
            return ctx.Wrap(h);
        }

        // local stripe i: its geometry and a view of the plane whose OWNED rows hold the result
        public DeviceTile ResultRows(int i, out NzStripe stripe) {
            stripe = default(NzStripe);
            Native.Check(Native.nz_sharded_stripe(Handle, i, ref stripe, out IntPtr _, out IntPtr result), "nz_sharded_stripe");
            int cols = stripe.cols, rows = stripe.own1 - stripe.own0;
            return new DeviceTile(ctx, IntPtr.Add(result, stripe.own0 * cols * sizeof(float)), rows * cols);
        }

        // GetMapRangeJob + MapNormalizeValues over the WHOLE grid: per-rank folds, one ncclAllGather, the same fold in rank
        // order (Filter/NormalizeJob.cs:17-92); `args` = device {min, max, range}
        public GpuJobHandle NormalizeToGlobalRange(DeviceTile args, float limMin = float.PositiveInfinity,
                                                   float limMax = float.NegativeInfinity) {
            Native.Check(Native.nz_sharded_map_range(ctx.Handle, Handle, args.Ptr, limMin, limMax, 0, out ulong _), "nz_sharded_map_range");
            Native.Check(Native.nz_sharded_normalize(ctx.Handle, Handle, args.Ptr, 0, out ulong h), "nz_sharded_normalize");
            return ctx.Wrap(h);
        }

        public void Dispose() {
            if (Handle != IntPtr.Zero) { Native.nz_sharded_destroy(Handle); Handle = IntPtr.Zero; }
        }
    }
}
