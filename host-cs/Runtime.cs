// Runtime.cs -- device context, device tiles and job handles for the C# host of libnoize_hip.so.
// Stands where Unity.Jobs / Unity.Collections stand in the reference: NativeSlice<float> -> DeviceTile (a plane in
// HBM), JobHandle -> GpuJobHandle (a marker on the context's HIP stream).  No Unity engine dependency.
// Source only: the build image of this repository has no .NET toolchain; the same calls are exercised from Python
// (noize_job_amd/pipeline.py) and C++ (noize_job_amd/host/noize_pipeline.hpp) by the test suite, and
// tests/test_host_cs.py checks every P/Invoke signature against include/noize_hip.h.
using System;
using System.Runtime.InteropServices;

namespace xshazwar.noize.hip {

    [StructLayout(LayoutKind.Sequential)]
    public struct NzStripe { public int cols, rows, grow0, grows, own0, own1, pitch; }            // nz_stripe

    [StructLayout(LayoutKind.Sequential)]
    public struct NzRwTile { public IntPtr read, write; public int resolution, count; }           // nz_rw_tile

    // the stock stage list as a parameter block (nz_sharded_create)
    [StructLayout(LayoutKind.Sequential)]
    public struct NzTerrainParams {                                                                  // nz_terrain_params
        public int noiseType;
        public float hurst, startingAmplitude, stepdown, detuneRate;
        public int octaves, noiseSize;
        public int filter;
        public int filterIterations;
        public int flowIterations;
        public float normMin, normMax;
        public int erosionIterations;
    }

    // one grid over the GPUs of a node (nz_sharded_create): haloMode 0 recompute, 1 exchange, 2 exchange_once
    [StructLayout(LayoutKind.Sequential)]
    public struct NzShardedDesc {                                                                    // nz_sharded_desc
        public int grows, cols;
        public int stripes;
        public int haloMode;
        public int overlap;
        public int xpos, zpos;
        public int externalSource;
        public int asRank, asWorld;
    }

    // ErosionParameters, Geologic/ParticleErosion/LiveErosionDataTypes.cs:78-100 (field order kept)
    [StructLayout(LayoutKind.Sequential)]
    public struct NzErosionParams {
        public float INERTIA, GRAVITY, DRAG, FRICTION, EVAP, EROSION, DEPOSITION, FLOW_HEIGHT_CONTRIBUTION;
        public float SLOW_CULL_ANGLE, SLOW_CULL_SPEED, CAPACITY;
        public int MAXAGE;
        public float TERMINAL_VELOCITY;
        public float SURFACE_EVAPORATION_RATE, POOL_PLACEMENT_MULTIPLIER, TRACK_PLACEMENT_MULTIPLIER, FLOW_LOSS_RATE;
        public int PILING_RADIUS;
        public float MIN_PILE_INCREMENT, PILE_THRESHOLD;
    }

    // TileSetMeta, Pipeline/Tiles/TileTypes.cs:15-27 (int2 / float2 flattened)
    [StructLayout(LayoutKind.Sequential)]
    public struct NzTileSetMeta {
        public int TILE_RES_x, TILE_RES_y, TILE_SIZE_x, TILE_SIZE_y, GENERATOR_RES_x, GENERATOR_RES_y;
        public float PATCH_RES_x, PATCH_RES_y;
        public int HEIGHT;
        public float HEIGHT_F;
        public int MARGIN;
    }

    public class NoizeException : Exception {       // negative nz_status -> exception (SURVEY.md 8b, error convention)
        public readonly int status;
        public NoizeException(int status, string message) : base(message) { this.status = status; }
    }

    public static partial class Native {
        public const int NZ_ERR_RETRY = -7;      // a chained kernel-filter launch timed out: schedule the work item again
        public static void Check(int status, string where) {
            if (status != 0) throw new NoizeException(status, $"{where} failed ({status}): {Marshal.PtrToStringAnsi(nz_last_error())}");
        }
    }

    // Unity.Jobs.JobHandle on this path.  The id names the context that issued it, so a handle may be passed as the
    // dependency of a stage on ANY context: the consumer's stream waits for it on the device.
    public struct GpuJobHandle {
        public IntPtr ctx;
        public ulong id;
        public bool IsCompleted {
            get {
                if (id == 0 || ctx == IntPtr.Zero) return true;   // default(JobHandle)
                Native.Check(Native.nz_handle_query(ctx, id, out int done), "nz_handle_query");
                return done != 0;
            }
        }
        public void Complete() { if (id != 0 && ctx != IntPtr.Zero) Native.Check(Native.nz_handle_wait(ctx, id), "nz_handle_wait"); }
        public static GpuJobHandle CombineDependencies(GpuContext on, params GpuJobHandle[] handles) {
            ulong[] ids = Array.ConvertAll(handles, h => h.id);
            Native.Check(Native.nz_handle_combine(on.Handle, ids, ids.Length, out ulong h2), "nz_handle_combine");
            return new GpuJobHandle { ctx = on.Handle, id = h2 };
        }
    }

    // One HIP stream; all stage calls of a pipeline instance are enqueued on it from one thread
    // (the reference schedules everything from the Unity main thread, Pipeline/Executable/Pipeline.cs:29-30).
    public sealed class GpuContext : IDisposable {
        public IntPtr Handle { get; private set; }
        public int Device => Native.nz_ctx_device(Handle);
        public GpuContext(int device = 0) {
            Native.Check(Native.nz_ctx_create(device, out IntPtr h), "nz_ctx_create");
            Handle = h;
        }
        public DeviceTile Alloc(int length) => new DeviceTile(this, length);
        // FloatMode of the [BurstCompile] attributes (Fractal.cs:19, KernelJob.cs:17, FlowMapJob.cs:16) as a property of the
        // context the jobs are scheduled on: 0 strict (default), 1 fast, 2 relaxed (include/noize_hip.h, nz_float_mode)
        public int FloatMode {
            get => Native.nz_ctx_float_mode(Handle);
            set => Native.Check(Native.nz_ctx_set_float_mode(Handle, value), "nz_ctx_set_float_mode");
        }
        public GpuJobHandle Record() {            // a marker behind everything enqueued so far
            Native.Check(Native.nz_handle_record(Handle, out ulong h), "nz_handle_record");
            return Wrap(h);
        }
        public void Synchronize() => Native.Check(Native.nz_ctx_synchronize(Handle), "nz_ctx_synchronize");
        public GpuJobHandle Wrap(ulong id) => new GpuJobHandle { ctx = Handle, id = id };
        public void Dispose() {
            if (Handle != IntPtr.Zero) { Native.nz_ctx_destroy(Handle); Handle = IntPtr.Zero; }
        }
    }

    // NativeArray<float>(n, Allocator.Persistent, UninitializedMemory) / NativeSlice<float>, living in HBM
    public sealed class DeviceTile : IDisposable {
        public readonly GpuContext ctx;
        public IntPtr Ptr { get; private set; }
        public int Length { get; }
        readonly bool owned;
        public DeviceTile(GpuContext ctx, int length) {
            this.ctx = ctx; Length = length; owned = true;
            Native.Check(Native.nz_tile_alloc(ctx.Handle, (UIntPtr)(uint)length, out IntPtr p), "nz_tile_alloc");
            Ptr = p;
        }
        public DeviceTile(GpuContext ctx, IntPtr devicePointer, int length) { this.ctx = ctx; Ptr = devicePointer; Length = length; owned = false; }
        public bool IsCreated => Ptr != IntPtr.Zero;
        public unsafe void CopyFrom(float[] host) {       // NativeArray.CopyFrom
            fixed (float* p = host) {
                Native.Check(Native.nz_tile_upload(ctx.Handle, Ptr, (IntPtr)p, (UIntPtr)(uint)host.Length, 0, out ulong h), "nz_tile_upload");
                Native.Check(Native.nz_handle_wait(ctx.Handle, h), "nz_handle_wait");
            }
        }
        public unsafe void CopyFrom(int[] host) {         // an int32 plane (a batch's positions): the same four-byte elements
            fixed (int* p = host) {
                Native.Check(Native.nz_tile_upload(ctx.Handle, Ptr, (IntPtr)p, (UIntPtr)(uint)host.Length, 0, out ulong h), "nz_tile_upload");
                Native.Check(Native.nz_handle_wait(ctx.Handle, h), "nz_handle_wait");
            }
        }
        // NativeSlice(array, start, length): a view, not an owner
        public DeviceTile Offset(int start, int length) => new DeviceTile(ctx, IntPtr.Add(Ptr, start * sizeof(float)), length);
        public unsafe float[] ToArray() {                  // NativeArray.ToArray
            float[] host = new float[Length];
            fixed (float* p = host) {
                Native.Check(Native.nz_tile_download(ctx.Handle, Ptr, (IntPtr)p, (UIntPtr)(uint)Length, 0, out ulong h), "nz_tile_download");
                Native.Check(Native.nz_handle_wait(ctx.Handle, h), "nz_handle_wait");
            }
            return host;
        }
        public void Dispose() {
            if (owned && Ptr != IntPtr.Zero) Native.Check(Native.nz_tile_free(ctx.Handle, Ptr), "nz_tile_free");
            Ptr = IntPtr.Zero;
        }
    }
}
