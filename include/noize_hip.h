/*
 * noize_hip.h -- C ABI of libnoize_hip.so: the MI355X (gfx950) replacement for the Burst job
 * structs on noize-job's per-cell terrain path.
 *
 * Every entry point replaces one reference *static delegate* (the seam the C# stages call,
 * SURVEY.md 8b) or one `PipelineStage.Schedule` body.  Conventions:
 *   - `NativeSlice<float>` / `NativeArray<float>`  ->  `float*` DEVICE pointer (nz_tile_alloc, or any
 *     hipMalloc'd / torch allocation on the ctx's device); planes are row-major, index z*res + x
 *     (Pipeline/Tiles/TileData.cs:72-77).
 *   - `JobHandle dependency` -> `nz_handle dep` (0 = default(JobHandle)); the returned JobHandle ->
 *     `nz_handle* out` (may be NULL: no handle, stream order only).  Work is enqueued asynchronously on the ctx's HIP stream;
 *     where an entry ends in a kernel launch its handle is that launch's completion event (no event record of its own).
 *   - exceptions -> negative `int32` status; `nz_last_error()` gives the message.
 *   - scalar argument order is the delegate's.
 * One nz_ctx is driven by one host thread at a time (the reference schedules everything from the
 * Unity main thread, Pipeline/Executable/Pipeline.cs:29-30,154-181).
 * All citations are relative to /root/reference.
 */
#ifndef NOIZE_HIP_H
#define NOIZE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NZ_VERSION 100

typedef struct nz_ctx nz_ctx;
typedef uint64_t nz_handle;

enum nz_status {
    NZ_OK = 0,
    NZ_ERR_INVALID = -1,     /* bad argument (reference: undefined behaviour or C# exception) */
    NZ_ERR_UNSUPPORTED = -2, /* combination without an implementation (e.g. Sobel3_2D in a batched launch) */
    NZ_ERR_HIP = -3,         /* HIP runtime error */
    NZ_ERR_NOMEM = -4,
    NZ_ERR_NO_DEVICE = -5,   /* no gfx950 device / HIP runtime unavailable */
    NZ_ERR_COMM = -6,        /* RCCL error, or librccl.so.1 could not be opened */
    NZ_ERR_RETRY = -7        /* a chained kernel-filter launch timed out (nz_handle_wait / nz_ctx_synchronize report it): the
                                planes computed since are invalid, the context has switched to separate launches -- schedule
                                the work item again (the hosts' BasePipeline does, once) */
};

/* NoiseStage.FractalNoise, Noise/NoiseStage.cs:15-24 */
enum nz_noise_type {
    NZ_NOISE_SIN = 0, NZ_NOISE_PERLIN, NZ_NOISE_PERIODIC_PERLIN, NZ_NOISE_SIMPLEX,
    NZ_NOISE_ROTATED_SIMPLEX, NZ_NOISE_CELLULAR, NZ_NOISE_DOMAIN_ROTATED_PERLIN,
    NZ_NOISE_DOMAIN_ROTATED_SIMPLEX
};

/* KernelFilterType, Filter/Kernel/KernelJob.cs:79-94 */
enum nz_kernel_filter_type {
    NZ_GAUSS9_S1 = 0, NZ_GAUSS7_S1, NZ_GAUSS5_S1, NZ_GAUSS3_S1,
    NZ_GAUSS9_S2, NZ_GAUSS7_S2, NZ_GAUSS5_S2, NZ_GAUSS3_S2,
    NZ_SMOOTH3, NZ_SOBEL3_HORIZONTAL, NZ_SOBEL3_VERTICAL, NZ_SOBEL3_2D,
    NZ_PREWITT3_HORIZONTAL, NZ_PREWITT3_VERTICAL
};

/* MeshType, Mesh/Stage/MeshTileStage.cs:23-26 */
enum nz_mesh_type { NZ_MESH_SQUARE_GRID = 0, NZ_MESH_OVERSHOOT_SQUARE_GRID = 1 };

/* Row-stripe view of a (grows x cols) global grid held by one rank (new-framework feature,
 * SURVEY.md 8e).  The buffer holds `rows` rows of `cols` floats; buffer row b is global row
 * b + grow0.  Stencil reads clamp to the global border only; rows [own0, own1) are produced, the
 * others are ghost rows the caller fills by halo exchange.  A single reference tile is
 * {res, res, 0, res, 0, res, 0}. */
typedef struct nz_stripe {
    int32_t cols;  /* cells per row */
    int32_t rows;  /* rows held in the buffer */
    int32_t grow0; /* global row of buffer row 0 (negative when ghost rows hang over the top border) */
    int32_t grows; /* rows of the global grid */
    int32_t own0;  /* owned rows [own0, own1) in buffer coordinates */
    int32_t own1;
    int32_t pitch; /* floats between consecutive rows; 0 = cols */
} nz_stripe;

/* The READ / WRITE slice pair of one tile, device resident: the reference's RWTileData
 * (Pipeline/Tiles/TileData.cs:49-93: `src` is read with a clamp, `dst` is written) together with
 * TileHelpers.SWAP_RWTILE (TileData.cs:42-45), which the reference implements as a copy job WRITE -> READ after
 * every job.  Here the swap is a swap: an `_rw` stage entry reads `read`, may use `write` as its ping-pong plane,
 * and RETURNS WITH `read` POINTING AT THE PLANE THAT HOLDS THE RESULT (the two pointers exchanged, or not) -- no
 * flush copy, and no constraint on the number of launches.  The struct is updated when the call returns (enqueue
 * time); both planes belong to the caller and stay valid until the returned handle has completed.
 * `count` tiles of resolution^2 floats stored back to back form a batch (1 = a single tile). */
typedef struct nz_rw_tile {
    float *read;
    float *write;
    int32_t resolution;
    int32_t count;
} nz_rw_tile;

/* ---- runtime ---------------------------------------------------------------------------- */
int32_t nz_version(void);
const char *nz_last_error(void);
int32_t nz_device_count(int32_t *count);

/* own stream */
int32_t nz_ctx_create(int32_t device, nz_ctx **out);
/* borrow an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream; NULL = default) */
int32_t nz_ctx_create_on_stream(int32_t device, void *hip_stream, nz_ctx **out);
int32_t nz_ctx_destroy(nz_ctx *ctx);
int32_t nz_ctx_synchronize(nz_ctx *ctx);
void *nz_ctx_stream(nz_ctx *ctx);
int32_t nz_ctx_device(nz_ctx *ctx); /* the HIP device the context was created on (-1 for NULL) */

/* Floating-point mode of a context's kernels.  The reference compiles every hot job with
 * [BurstCompile(FloatPrecision.Standard/High, FloatMode.Fast)] (Noise/Fractal/Fractal.cs:19, Filter/Kernel/KernelJob.cs:17,
 * Geologic/FlowMap/FlowMapJob.cs:16): Burst may contract and re-associate, so the reference's own results are only defined to
 * a tolerance (1e-5 relative, 1e-6 absolute is the contract of this path).
 *   NZ_FLOAT_STRICT (default): every kernel reproduces the operation sequence of the C# source, IEEE binary32, no
 *     contraction -- results are a pure function of the inputs, equal across every launch shape;
 *   NZ_FLOAT_FAST: the smooth tail of the simplex fBm octave (corner falloffs, gradient dots, octave accumulation) and the
 *     convolution tap sums are FMA-contracted; every discrete decision and every cancellation (skew / unskew, floors, lattice
 *     hashes, selects, clamps, the min filter) stays exact.  Every stage stays within 1e-5 relative / 1e-6 absolute of the strict
 *     result for the same input plane (measured at 4096^2: fBm 9e-7, Gauss5 x17 4.5e-7);
 *   NZ_FLOAT_RELAXED: FAST, and the flow map's iterations use a reciprocal (v_rcp_f32) for the outflow scale's division, an FMA
 *     in the water update, v_sqrt_f32 and a reciprocal multiply in the velocity / normalise epilogue.  The flow map itself
 *     amplifies a change of one ulp: total = water + height rounds the water (~1e-4) to the height's ulp (6e-8), so a 1e-11
 *     change of a cell's water moves its outflows by 6e-8 wherever it crosses a rounding boundary -- ~1e-4 of the cells of the
 *     metric tile leave the 1e-5 band (largest deviation 2e-5 of the normalised range), with ANY arithmetic that is not
 *     bit-identical, Burst's own FloatMode.Fast builds included.
 * Within a mode sharded == monolithic and all launch shapes of a stage still agree bit for bit.  Kernels without a tolerance
 * form run their strict one.  Applies to work enqueued after the call. */
enum nz_float_mode { NZ_FLOAT_STRICT = 0, NZ_FLOAT_FAST = 1, NZ_FLOAT_RELAXED = 2 };
int32_t nz_ctx_set_float_mode(nz_ctx *ctx, int32_t mode);
int32_t nz_ctx_float_mode(nz_ctx *ctx); /* -1 for NULL */

/* NativeArray<float>(n, Allocator.Persistent, UninitializedMemory) / Dispose() */
int32_t nz_tile_alloc(nz_ctx *ctx, size_t n_floats, float **out_dev);
int32_t nz_tile_free(nz_ctx *ctx, float *dev);
/* host NativeArray interop (async on the ctx stream; host memory must stay valid until `out` completes) */
int32_t nz_tile_upload(nz_ctx *ctx, float *dev, const float *host, size_t n_floats, nz_handle dep, nz_handle *out);
int32_t nz_tile_download(nz_ctx *ctx, const float *dev, float *host, size_t n_floats, nz_handle dep, nz_handle *out);
int32_t nz_bytes_download(nz_ctx *ctx, const void *dev, void *host, size_t n_bytes, nz_handle dep, nz_handle *out);

/* FlushWriteSliceDelegate(write_, read_, deps), Pipeline/Tiles/TileData.cs:15-42: write_.CopyFrom(read_) as a
 * job -- device to device here.  What Read/WriteGeneratorContextStage schedule
 * (Pipeline/PipelineState/Stage/ReadGeneratorContextStage.cs:36-44, WriteGeneratorContextStage.cs:30-44). */
int32_t nz_flush_write_slice(nz_ctx *ctx, float *write_, const float *read_, size_t n_floats, nz_handle dep,
                             nz_handle *out);

/* JobHandle: marker recorded on the stream after the last kernel of a call.  A handle value names the context that
 * issued it (context id in the bits above bit 40), so it may be passed as `dep` to ANY context of the process:
 * a dependency on another context's handle makes this context's stream wait for that marker on the device
 * (hipStreamWaitEvent) without blocking the host -- the reference's JobHandle dependencies between pipelines
 * (Pipeline/Executable/ReducePipeline.cs:82-148) and the job-fence locks of
 * Pipeline/PipelineState/PipelineStateLock.cs:12-39.  Handles of a destroyed context read as completed. */
int32_t nz_handle_record(nz_ctx *ctx, nz_handle *out);
/* `ctx` may be any live context: the handle names its owner */
int32_t nz_handle_query(nz_ctx *ctx, nz_handle h, int32_t *is_completed); /* JobHandle.IsCompleted */
int32_t nz_handle_wait(nz_ctx *ctx, nz_handle h);                          /* JobHandle.Complete() */
/* JobHandle.CombineDependencies(h0, h1, ...): a marker on ctx's stream that completes after all of them (handles of
 * any context; count may be 0) */
int32_t nz_handle_combine(nz_ctx *ctx, const nz_handle *handles, int32_t count, nz_handle *out);
/* the id (>= 1) of a context, and of the context a handle was issued by (0 for default(JobHandle)) */
int32_t nz_ctx_id(nz_ctx *ctx);
int32_t nz_handle_context_id(nz_handle h);
/* GPU time between two handles of `ctx` in ms (hipEventElapsedTime); both must have completed */
int32_t nz_handle_elapsed_ms(nz_ctx *ctx, nz_handle start, nz_handle stop, float *ms);

/* ---- noise: FractalJobDelegate, Noise/Fractal/Fractal.cs:76-88 ------------------------------ */
int32_t nz_fractal(nz_ctx *ctx, int32_t noiseType, float *src, int32_t resolution, float hurst,
                   float startingAmplitude, float stepdown, float detuneRate, int32_t octaves,
                   int32_t xpos, int32_t zpos, int32_t noiseSize, nz_handle dep, nz_handle *out);
/* same on the owned rows of a stripe: cell (x, b) is world cell (x + xpos, b + grow0 + zpos) */
int32_t nz_fractal_stripe(nz_ctx *ctx, int32_t noiseType, float *buf, const nz_stripe *st, float hurst,
                          float startingAmplitude, float stepdown, float detuneRate, int32_t octaves,
                          int32_t xpos, int32_t zpos, int32_t noiseSize, nz_handle dep, nz_handle *out);

/* ---- separable kernel filters ------------------------------------------------------------- */
/* SeperableKernelFilterDelegate, Filter/Kernel/KernelJob.cs:308-314 (one X+Z application) */
int32_t nz_kernel_filter(nz_ctx *ctx, float *src, float *tmp, int32_t filter, int32_t resolution,
                         nz_handle dep, nz_handle *out);
/* Edge1DFilterDelegate(src, tmp, EdgeAlgorithm algo, EdgeDirection dir, resolution, dep) and
 * Edge2DFilterDelegate(src, tmp, algo, resolution, dep), Filter/Kernel/Edge/EdgeJob.cs:22-43.
 * algo: 0 SOBEL, 1 PREWITT; dir: 0 HORIZONTAL, 1 VERTICAL (EdgeDetection.cs:13-21).  The 2-D form is
 * ScheduleReduce<RootSumSquaresTiles>: sqrt(horizontal^2 + vertical^2). */
int32_t nz_edge_1d_filter(nz_ctx *ctx, float *src, float *tmp, int32_t algo, int32_t dir, int32_t resolution,
                          nz_handle dep, nz_handle *out);
int32_t nz_edge_2d_filter(nz_ctx *ctx, float *src, float *tmp, int32_t algo, int32_t resolution, nz_handle dep,
                          nz_handle *out);
/* GaussFilter.GaussFilterDelegate, Filter/Kernel/Blur/BlurJob.cs:23-30 (sigma = GaussSigma enum 0..15) */
int32_t nz_gauss_filter(nz_ctx *ctx, float *src, float *tmp, int32_t width, int32_t sigma,
                        int32_t resolution, nz_handle dep, nz_handle *out);
/* SmoothFilter.SmoothFilterDelegate, BlurJob.cs:46-52 */
int32_t nz_smooth_filter(nz_ctx *ctx, float *src, float *tmp, int32_t width, int32_t resolution,
                         nz_handle dep, nz_handle *out);
/* SeparableKernelFilter.ScheduleSeries, KernelJob.cs:165-185; kernelX/kernelZ are HOST arrays of
 * kernelSize floats (the NativeArray<float> kernel bodies) */
int32_t nz_separable_series(nz_ctx *ctx, float *src, float *tmp, int32_t resolution, int32_t kernelSize,
                            const float *kernelX, const float *kernelZ, float kernelFactor,
                            nz_handle dep, nz_handle *out);
/* ErosionKernelJobDelegate, KernelJob.cs:350 (min-X then min-Z, window {-1,0}) */
int32_t nz_erosion_kernel(nz_ctx *ctx, float *src, int32_t resolution, nz_handle dep, nz_handle *out);

/* Stage bodies: the `iterations` loops of KernelFilterStage.Schedule (Filter/KernelFilterStage.cs:31-43),
 * StageGaussianBlur.Schedule / StageSmoothBlur.Schedule (Filter/Kernel/Blur/Stage*.cs) fused into
 * as few launches as halo growth allows.  Result lands in `src`; `tmp` is stage scratch. */
int32_t nz_kernel_filter_stage(nz_ctx *ctx, float *src, float *tmp, int32_t filter, int32_t iterations,
                               int32_t resolution, nz_handle dep, nz_handle *out);
int32_t nz_gauss_blur_stage(nz_ctx *ctx, float *src, float *tmp, int32_t width, int32_t sigma,
                            int32_t iterations, int32_t resolution, nz_handle dep, nz_handle *out);
int32_t nz_smooth_blur_stage(nz_ctx *ctx, float *src, float *tmp, int32_t width, int32_t iterations,
                             int32_t resolution, nz_handle dep, nz_handle *out);
/* ErosionKernelJob applied `iterations` times (the reference has no stage wrapper for it) */
int32_t nz_erosion_stage(nz_ctx *ctx, float *src, float *tmp, int32_t iterations, int32_t resolution,
                         nz_handle dep, nz_handle *out);

/* The same stage bodies on a READ / WRITE pair (nz_rw_tile above): the result is in tile->read when the call
 * returns; the flush copies of the in-place forms are gone and the launch count is free (five erosion iterations
 * are one launch, a single filter application is one launch without a copy). */
int32_t nz_kernel_filter_stage_rw(nz_ctx *ctx, nz_rw_tile *tile, int32_t filter, int32_t iterations, nz_handle dep,
                                  nz_handle *out);
int32_t nz_gauss_blur_stage_rw(nz_ctx *ctx, nz_rw_tile *tile, int32_t width, int32_t sigma, int32_t iterations,
                               nz_handle dep, nz_handle *out);
int32_t nz_smooth_blur_stage_rw(nz_ctx *ctx, nz_rw_tile *tile, int32_t width, int32_t iterations, nz_handle dep,
                                nz_handle *out);
int32_t nz_erosion_stage_rw(nz_ctx *ctx, nz_rw_tile *tile, int32_t iterations, nz_handle dep, nz_handle *out);

/* stripe forms: one launch that advances `iterations` applications on rows [own0, own1); the filter needs
 * nz_kernel_filter_halo_rows(...) valid ghost rows on each side, the min erosion `iterations` rows ABOVE only
 * (its window is {-1, 0}); rows beyond the global border are never needed.  Reads `src`, writes `dst`
 * (owned rows only). */
int32_t nz_kernel_filter_halo_rows(int32_t filter, int32_t iterations);
int32_t nz_kernel_filter_max_fused(int32_t filter);
int32_t nz_erosion_max_fused_iterations(void);
int32_t nz_kernel_filter_stripe(nz_ctx *ctx, const float *src, float *dst, const nz_stripe *st,
                                int32_t filter, int32_t iterations, nz_handle dep, nz_handle *out);
int32_t nz_erosion_stripe(nz_ctx *ctx, const float *src, float *dst, const nz_stripe *st,
                          int32_t iterations, nz_handle dep, nz_handle *out);

/* ---- flow map ------------------------------------------------------------------------------ */
/* FillArrayJobDelegate, Geologic/FlowMap/FlowMapComponents.cs:204 */
int32_t nz_fill_array(nz_ctx *ctx, float *data, int32_t resolution, float value, nz_handle dep, nz_handle *out);
/* FlowMapStepComputeFlowDelegate, Geologic/FlowMap/FlowMapJob.cs:82-98 */
int32_t nz_flowmap_compute_flow(nz_ctx *ctx, const float *src, const float *waterMap, float *flowMapN,
                                float *flowMapN__buff, float *flowMapS, float *flowMapS__buff,
                                float *flowMapE, float *flowMapE__buff, float *flowMapW,
                                float *flowMapW__buff, int32_t resolution, nz_handle dep, nz_handle *out);
/* FlowMapStepUpdateWaterDelegate, FlowMapJob.cs:154-165 */
int32_t nz_flowmap_update_water(nz_ctx *ctx, float *waterMap, float *waterMap__buff, const float *flowMapN,
                                const float *flowMapS, const float *flowMapE, const float *flowMapW,
                                int32_t resolution, nz_handle dep, nz_handle *out);
/* FlowMapWriteValuesDelegate, FlowMapJob.cs:220-228 */
int32_t nz_flowmap_write_values(nz_ctx *ctx, float *src, const float *flowMapN, const float *flowMapS,
                                const float *flowMapE, const float *flowMapW, int32_t resolution,
                                nz_handle dep, nz_handle *out);
/* MapNormalizeValuesDelegate, Filter/NormalizeJob.cs:94-100; args = HOST {min, max, range} */
int32_t nz_map_normalize_values(nz_ctx *ctx, float *src, float *tmp, const float *args,
                                int32_t resolution, nz_handle dep, nz_handle *out);
/* GetMapRangeJob.Schedule, Filter/NormalizeJob.cs:17-55: res = DEVICE {min, max, max - min} of the plane, folded from
 * lim_min / lim_max (the reference's defaults: +infinity / -infinity) with math.min / math.max: NaN cells are skipped,
 * and where the extreme is zero its sign is that of the last zero cell, as the sequential fold leaves it */
int32_t nz_get_map_range(nz_ctx *ctx, const float *map, size_t n_floats, float *res, float lim_min, float lim_max,
                         nz_handle dep, nz_handle *out);
/* MapNormalizeValuesDelegate with args = DEVICE {min, max, range} (what nz_get_map_range leaves; on a sharded grid,
 * after the ranks have all-reduced min and max) */
int32_t nz_map_normalize_values_dev(nz_ctx *ctx, float *src, float *tmp, const float *args, int32_t resolution,
                                    nz_handle dep, nz_handle *out);
/* ... on any contiguous run of cells (the owned rows of a stripe) */
int32_t nz_normalize_cells_dev(nz_ctx *ctx, float *data, size_t n_floats, const float *args, nz_handle dep,
                               nz_handle *out);
/* FlowMapStage.Schedule, Geologic/Stage/FlowMapStage.cs:124-214: fill -> iterations x (flow, water)
 * -> velocity -> normalise, result in `src`.  `work` = stage-owned scratch of
 * nz_flowmap_stage_work_floats(resolution) floats (the stage's 11 planes; flux is defined as zero
 * at the start of every run). */
size_t nz_flowmap_stage_work_floats(int32_t resolution);
int32_t nz_flowmap_stage(nz_ctx *ctx, float *src, float *work, int32_t iterations, float normMin,
                         float normMax, int32_t resolution, nz_handle dep, nz_handle *out);

/* FlowMapStage.Schedule on a READ / WRITE pair: heights are read from tile->read, the normalised flow map is
 * written to tile->write and the pair is swapped; `work` = nz_flowmap_stage_rw_work_floats(resolution, count) floats
 * (the two sets of five state planes; no private copy of the heights is needed). */
size_t nz_flowmap_stage_rw_work_floats(int32_t resolution, int32_t count);
int32_t nz_flowmap_stage_rw(nz_ctx *ctx, nz_rw_tile *tile, float *work, int32_t iterations, float normMin,
                            float normMax, nz_handle dep, nz_handle *out);

/* stripe forms for sharded runs: state = {water, fN, fS, fE, fW} planes of the stripe's shape. */
/* `iterations` (<= nz_flow_fused_max_iterations()) whole iterations in one launch on an on-chip tile; needs
 * 2*iterations valid ghost rows of height (and of every state_in plane unless `first`).  first != 0: the
 * initial state (water 1e-4, flux 0) is implied and state_in is not read.  last != 0: the launch ends in
 * velocity + normalise and writes `dst` only; otherwise it writes the five state_out planes. */
int32_t nz_flow_fused_max_iterations(void);
int32_t nz_flow_fused_stripe(nz_ctx *ctx, const float *height, const float *const *state_in, float *const *state_out,
                             float *dst, const nz_stripe *st, int32_t iterations, int32_t first, int32_t last,
                             float normMin, float normMax, nz_handle dep, nz_handle *out);

/* ---- batched stage bodies (new-framework feature) ------------------------------------------------
 * `count` independent tiles of resolution^2 cells stored back to back (tile k at data + k * resolution^2) go
 * through one launch sequence: the reference runs one BasePipeline per tile request
 * (Scripts/MeshTileGenerator.cs:181-211, default resolution 512), and a 512^2 tile alone cannot fill 256 CUs.
 * Every tile is clamped at its own border exactly as in the single-tile entry points; results are identical.
 * `positions` = DEVICE array of 2 * count int32 {xpos, zpos} (GeneratorData.xpos/zpos per tile).
 * Flow-map work buffer: count * nz_flowmap_stage_work_floats(resolution) floats, plane-major. */
int32_t nz_fractal_batch(nz_ctx *ctx, int32_t noiseType, float *data, int32_t resolution, int32_t count,
                         const int32_t *positions, float hurst, float startingAmplitude, float stepdown,
                         float detuneRate, int32_t octaves, int32_t noiseSize, nz_handle dep, nz_handle *out);
int32_t nz_kernel_filter_stage_batch(nz_ctx *ctx, float *src, float *tmp, int32_t filter, int32_t iterations,
                                     int32_t resolution, int32_t count, nz_handle dep, nz_handle *out);
int32_t nz_gauss_blur_stage_batch(nz_ctx *ctx, float *src, float *tmp, int32_t width, int32_t sigma,
                                  int32_t iterations, int32_t resolution, int32_t count, nz_handle dep,
                                  nz_handle *out);
int32_t nz_smooth_blur_stage_batch(nz_ctx *ctx, float *src, float *tmp, int32_t width, int32_t iterations,
                                   int32_t resolution, int32_t count, nz_handle dep, nz_handle *out);
int32_t nz_erosion_stage_batch(nz_ctx *ctx, float *src, float *tmp, int32_t iterations, int32_t resolution,
                               int32_t count, nz_handle dep, nz_handle *out);
int32_t nz_flowmap_stage_batch(nz_ctx *ctx, float *src, float *work, int32_t iterations, float normMin,
                               float normMax, int32_t resolution, int32_t count, nz_handle dep, nz_handle *out);
/* `count` meshes from height planes stored back to back (inputResolution^2 floats each); mesh k is written at
 * vertices + k * nz_mesh_vertex_count(resolution) * 48 bytes and indices + k * nz_mesh_index_count(resolution) */
int32_t nz_heightmap_mesh_batch(nz_ctx *ctx, int32_t meshType, void *vertices, uint32_t *indices,
                                int32_t resolution, int32_t inputResolution, int32_t marginPix, float tileHeight,
                                float tileSize, const float *heights, int32_t count, nz_handle dep, nz_handle *out);

/* ---- mesh: HeightMapMeshJobScheduleDelegate, Mesh/Job/HeightMapMeshJob.cs:55-65 -------------- */
/* (Mesh, MeshData) -> device vertex stream of (resolution+1)^2 records
 * {float3 position; float3 normal; float4 tangent; float2 texCoord0} = 48 B
 * (PositionStream32.Stream0, Mesh/Streams/PositionStream.cs:77-82) and device index buffer of
 * 6*resolution^2 uint32 (TriangleUInt32, Mesh/Streams/Triangle.cs:19-27). */
size_t nz_mesh_vertex_count(int32_t resolution);
size_t nz_mesh_index_count(int32_t resolution);
int32_t nz_heightmap_mesh(nz_ctx *ctx, int32_t meshType, void *vertices, uint32_t *indices,
                          int32_t resolution, int32_t inputResolution, int32_t marginPix,
                          float tileHeight, float tileSize, const float *heights, nz_handle dep,
                          nz_handle *out);

/* HeightMapMeshJob<G, PositionStream16> (Mesh/Streams/PositionStream.cs:11-74, TriangleUInt16 Triangle.cs:7-17): the
 * same 48-byte vertex records, indices truncated to 16 bits as `(ushort)` does -- "only valid for square mesh up to
 * resolution 256*256" (:12).  `indices`: 6 * resolution^2 uint16. */
int32_t nz_heightmap_mesh16(nz_ctx *ctx, int32_t meshType, void *vertices, uint16_t *indices, int32_t resolution,
                            int32_t inputResolution, int32_t marginPix, float tileHeight, float tileSize,
                            const float *heights, nz_handle dep, nz_handle *out);

/* ---- element-wise stages either side of the path ------------------------------ */
/* ConstantJobScheduleDelegate, Filter/ConstantJob.cs:47-53; operation = ConstantStage.ConstantOperationType
 * {MULTIPLY = 0, BINARIZE = 1} (Filter/ConstantStage.cs:15-18) */
int32_t nz_constant_job(nz_ctx *ctx, int32_t operation, float *srcL, float *tmp, float constantValue,
                        int32_t resolution, nz_handle dep, nz_handle *out);
/* ReductionJobScheduleDelegate, Filter/ReductionJob.cs:54-60; operation = ReductionType {SUBTRACT, MULTIPLY,
 * ROOTSUMSQUARES, MAX, MIN} (Filter/Reduce/ReduceStage.cs:12-18); the result lands in srcL */
int32_t nz_reduction_job(nz_ctx *ctx, int32_t operation, float *srcL, const float *srcR, float *tmp,
                         int32_t resolution, nz_handle dep, nz_handle *out);
/* CurveJobScheduleDelegate, Filter/Curve/CurveJob.cs:91-97; `curve` = DEVICE array of curveSize samples */
int32_t nz_curve_job(nz_ctx *ctx, float *src, float *tmp, const float *curve, int32_t curveSize,
                     int32_t resolution, nz_handle dep, nz_handle *out);

/* ---- live erosion: the deterministic grid jobs (planes indexed x * resolution + z, WorldTile.getIdx,
 * Geologic/ParticleErosion/LiveErosionDataTypes.cs:608-610) ------------------------------------------------
 * UpdateFlowFromTrackJob.Schedule(pool, flow, track, ep, tm, res, deps), MultiThreadErosionJob.cs:240-261:
 * ep.FLOW_LOSS_RATE, ep.SURFACE_EVAPORATION_RATE and tm.HEIGHT are passed as scalars. */
int32_t nz_update_flow_from_track(nz_ctx *ctx, float *pool, float *flow, float *track, float flowLossRate,
                                  float surfaceEvaporationRate, float tileHeight, int32_t resolution,
                                  nz_handle dep, nz_handle *out);
/* PoolAutomataJob.Schedule(pool, height, particleQueue, ep, tm, iterations, res, drainParticles, deps),
 * MultiThreadErosionJob.cs:289-325, with drainParticles == false: `iterations` x four colour passes of
 * WorldTile.SpreadPool (LiveErosionDataTypes.cs:938-1010).  The neighbour order comes from NativeArray.Sort() of
 * com.unity.collections 1.4.0 (not in the reference tree; restated: insertion sort for 4 elements). */
int32_t nz_pool_automata(nz_ctx *ctx, float *pool, const float *height, int32_t iterations, int32_t resolution,
                         nz_handle dep, nz_handle *out);

/* ---- live erosion: the particle half (BASELINE config 4).  LiveErosion.TriggerQueuedBeyerMT
 * (Geologic/ParticleErosion/Component/LiveErosion.cs:378-436) chains these jobs per cycle.  The reference is
 * deterministic per particle but not per run; this library fixes the three free choices (see nz_live.hip): the random
 * seed is an argument, per-cell event sums are 2^-40 fixed point (order-independent), and ErodeHeightMaps applies the
 * per-cell sediment events in the order one worker would have produced them (all KernelDisperse events, then the
 * PileSolver events).  Same seed, same planes -> same result, bit for bit, on every run. ------------------------- */
/* ErosionParameters, Geologic/ParticleErosion/LiveErosionDataTypes.cs:78-100 (field order kept) */
typedef struct nz_erosion_params {
    float INERTIA, GRAVITY, DRAG, FRICTION, EVAP, EROSION, DEPOSITION, FLOW_HEIGHT_CONTRIBUTION;
    float SLOW_CULL_ANGLE, SLOW_CULL_SPEED, CAPACITY;
    int32_t MAXAGE;
    float TERMINAL_VELOCITY;
    float SURFACE_EVAPORATION_RATE, POOL_PLACEMENT_MULTIPLIER, TRACK_PLACEMENT_MULTIPLIER, FLOW_LOSS_RATE;
    int32_t PILING_RADIUS;
    float MIN_PILE_INCREMENT, PILE_THRESHOLD;
} nz_erosion_params;
/* TileSetMeta, Pipeline/Tiles/TileTypes.cs:15-27 (field order kept); the jobs read GENERATOR_RES, PATCH_RES.x, HEIGHT */
typedef struct nz_tile_set_meta {
    int32_t TILE_RES[2], TILE_SIZE[2], GENERATOR_RES[2];
    float PATCH_RES[2];
    int32_t HEIGHT;
    float HEIGHT_F;
    int32_t MARGIN;
} nz_tile_set_meta;
/* a queued BeyerParticle: what its constructors set that is not a constant (LiveErosionDataTypes.cs:221-237) */
typedef struct nz_particle {
    int32_t px, pz; /* pos */
    float water;
    uint32_t pid;
} nz_particle;
/* NativeQueue<BeyerParticle> (+ the NativeList it is copied to, CopyBeyerQueueJob) in device memory */
typedef struct nz_particle_queue nz_particle_queue;
/* NativeParallelMultiHashMap<int, ErosiveEvent> events + NativeQueue<ErosiveEvent> erosions: per-cell sums, the cells
 * that received an event, and the per-cell sediment event as a dense plane */
typedef struct nz_erosive_events nz_erosive_events;

int32_t nz_particle_queue_create(nz_ctx *ctx, int32_t capacity, nz_particle_queue **out);
int32_t nz_particle_queue_destroy(nz_ctx *ctx, nz_particle_queue *queue);
/* NativeQueue.Count / ToArray (the host waits for the ctx's stream); NZ_ERR_NOMEM if a job found the queue too small */
int32_t nz_particle_queue_count(nz_ctx *ctx, nz_particle_queue *queue, int32_t *count);
int32_t nz_particle_queue_download(nz_ctx *ctx, nz_particle_queue *queue, nz_particle *host, int32_t max_count,
                                   int32_t *count);
int32_t nz_particle_queue_upload(nz_ctx *ctx, nz_particle_queue *queue, const nz_particle *host, int32_t count);
/* ClearQueueJob<BeyerParticle>.ScheduleRun(queue, deps), MultiThreadErosionJob.cs:133-153 */
int32_t nz_clear_particle_queue(nz_ctx *ctx, nz_particle_queue *queue, nz_handle dep, nz_handle *out);
int32_t nz_erosive_events_create(nz_ctx *ctx, int32_t resolution, nz_erosive_events **out);
int32_t nz_erosive_events_destroy(nz_ctx *ctx, nz_erosive_events *events);
/* device plane, x * res + z: the per-cell sediment events nz_process_beyer_erosive_events left.  READ ONLY for hosts:
 * nz_erode_height_maps finds the events through the cycle's cell list, not by scanning this plane */
float *nz_erosive_events_sediment(nz_erosive_events *events);
int32_t nz_erosive_events_count(nz_ctx *ctx, nz_erosive_events *events, int32_t *count); /* events of the last descent */

/* FillBeyerQueueJob.ScheduleParallel(particles, ep, tm, generationRound, res, maxParticles, deps, concurrency),
 * MultiThreadErosionJob.cs:37-71; `seed` replaces UnityEngine.Random.Range(0, Int32.MaxValue) (:50).  The number of
 * particles already queued is read when the job RUNS (the reference reads particles.Count when it schedules). */
int32_t nz_fill_beyer_queue(nz_ctx *ctx, nz_particle_queue *particles, const nz_erosion_params *ep,
                            const nz_tile_set_meta *tm, int32_t generationRound, int32_t res, int32_t maxParticles,
                            int32_t seed, int32_t concurrency, nz_handle dep, nz_handle *out);
/* QueuedBeyerCycleMultiThreadJob.ScheduleParallel(height, pool, flow, track, particles, events, ep, tm, eventLimit, res,
 * deps), :196-223: every queued particle descends until it is dead; the planes are read only */
int32_t nz_queued_beyer_cycle(nz_ctx *ctx, const float *height, const float *pool, const float *flow, const float *track,
                              nz_particle_queue *particles, nz_erosive_events *events, const nz_erosion_params *ep,
                              const nz_tile_set_meta *tm, int32_t eventLimit, int32_t res, nz_handle dep, nz_handle *out);
/* ProcessBeyerErosiveEventsJob.ScheduleRun(height, pool, flow, track, erosions, events, ep, tm, res, deps), :356-384 */
int32_t nz_process_beyer_erosive_events(nz_ctx *ctx, float *height, float *pool, float *flow, float *track,
                                        nz_erosive_events *events, const nz_erosion_params *ep,
                                        const nz_tile_set_meta *tm, int32_t res, nz_handle dep, nz_handle *out);
/* ErodeHeightMaps.ScheduleRun(height, erosions, ep, tm, res, deps), :459-479: applies the events of the LAST
 * nz_process_beyer_erosive_events on `events` (in place on `height`).  The PileSolver events of all four block colours run as
 * ONE launch whose busy blocks wait for their lower-coloured busy neighbours (NZ_PILE_TICKET=0: a launch per colour); that
 * wait is bounded.  A block that gives up (never, while resident waves hold the lower tickets) makes the context's next wait /
 * synchronisation return NZ_ERR_HIP -- the height plane is then invalid, and the context runs a launch per colour from then on.
 * SAFE MODE, nz_ctx_set_pile_safe(ctx, 1): the job keeps a copy of `height` as it found it (one plane copy per call), waits for
 * its ticket launch, and, should a block have given up, puts the plane back and runs itself again colour by colour: the caller
 * sees a job that succeeded (nz_ctx_pile_retries counts them).  The three hosts expose it as LiveErosion's `safe` option. */
int32_t nz_ctx_set_pile_safe(nz_ctx *ctx, int32_t on);
int32_t nz_ctx_pile_retries(nz_ctx *ctx);
/* test hook: polls after which a block of the ticket launch gives up; <= 0: the default (2^22, seconds) */
int32_t nz_debug_pile_poll_limit(int32_t polls);
int32_t nz_erode_height_maps(nz_ctx *ctx, float *height, nz_erosive_events *events, const nz_erosion_params *ep,
                             const nz_tile_set_meta *tm, int32_t res, nz_handle dep, nz_handle *out);
/* ErodeHeightMaps and UpdateFlowFromTrackJob as ONE call.  The reference schedules the two on the same dependency and
 * combines their handles (Component/LiveErosion.cs:408-412): siblings in its job graph, run side by side by the worker
 * threads.  Here the pile solver's launch -- a few thousand waves that wait for memory and for each other on an otherwise idle
 * chip -- carries the flow update's workgroups behind its own.  Results: exactly those of nz_erode_height_maps followed by
 * nz_update_flow_from_track(pool, flow, track, ep->FLOW_LOSS_RATE, ep->SURFACE_EVAPORATION_RATE, tm->HEIGHT) -- the two jobs
 * share no plane (pool / flow / track must not be `height`).  (A pile solver with more than 16 KB of
 * LDS -- PILING_RADIUS beyond ~18 -- runs the two launches one after the other.) */
int32_t nz_erode_height_maps_and_flow(nz_ctx *ctx, float *height, nz_erosive_events *events, float *pool, float *flow,
                                      float *track, const nz_erosion_params *ep, const nz_tile_set_meta *tm, int32_t res,
                                      nz_handle dep, nz_handle *out);
/* PoolAutomataJob.Schedule(pool, height, particleQueue, ep, tm, iterations, res, drainParticles, deps), :289-325:
 * with drainParticles != 0 a pool that finds a dry, lower neighbour leaves as one particle (pid 64000) in the queue */
int32_t nz_pool_automata_job(nz_ctx *ctx, float *pool, const float *height, nz_particle_queue *particleQueue,
                             const nz_erosion_params *ep, const nz_tile_set_meta *tm, int32_t iterations, int32_t res,
                             int32_t drainParticles, nz_handle dep, nz_handle *out);
/* CurvitureMapJob.ScheduleRun(texture, height, tm, target, res, deps), :413-435: `texture` = device RGBA32 pixels of a
 * meshRes^2 texture (4 bytes per pixel), target = ColorChannelByte {R, G, B, A} */
int32_t nz_curviture_map(nz_ctx *ctx, uint8_t *texture, const float *height, const nz_tile_set_meta *tm, int32_t target,
                         int32_t res, int32_t meshRes, nz_handle dep, nz_handle *out);
/* SetRGBA32Job.ScheduleRun(src, texture, target, deps, scale), :506-528 (dataRes = sqrt(src.Length)) */
int32_t nz_set_rgba32(nz_ctx *ctx, const float *src, uint8_t *texture, int32_t target, int32_t dataRes, int32_t meshRes,
                      float scale, nz_handle dep, nz_handle *out);

/* MeshJobScheduleDelegate with G = SharedSquareGridPosition (Mesh/Job/MeshJob.cs:37-60,
 * Mesh/Generators/SharedSquareGridPosition.cs:20-50; MeshHelper.makeSquarePlanarMesh): the flat unit-square grid,
 * same vertex / index layout and counts as nz_heightmap_mesh.  TileSize / Height of the delegate only set
 * mesh.bounds and are not needed here. */
int32_t nz_square_grid_mesh(nz_ctx *ctx, void *vertices, uint32_t *indices, int32_t resolution, nz_handle dep,
                            nz_handle *out);

/* CropJobDelegate(input, inputResolution, output, outputResolution, dep), Filter/Sample/CropJob.cs:62-68.
 * As in the reference, Offset stays 0 (ScheduleParallel :43-59 never sets it): the top-left
 * outputResolution^2 corner, reads clamped to the input plane. */
int32_t nz_crop_job(nz_ctx *ctx, const float *input, int32_t inputResolution, float *output,
                    int32_t outputResolution, nz_handle dep, nz_handle *out);

/* ThermalErosionFilterDelegate, Filter/Kernel/Blur/ThermalErosionFilter.cs:149-157: `iterations` x 4 phases of
 * in-place talus relaxation on disjoint 2x2 blocks (talus in degrees) */
int32_t nz_thermal_erosion(nz_ctx *ctx, float *src, float talus, float incrementRatio, float meshHeightWidthRatio,
                           int32_t iterations, int32_t resolution, nz_handle dep, nz_handle *out);

/* Test hook for the chained filter launches (a stage of three or more fused launches runs as ONE grid whose tiles wait
 * for the tiles of the previous launch they depend on): the workgroup that takes work item `item` (items count through
 * the chain, launch 0's tiles first) sleeps `sleeps` x ~3.4 us before it loads its tile -- a straggler that reads a plane
 * later launches overwrite.  Results must not change.  item < 0: off. */
int32_t nz_debug_chain_delay(int32_t item, int32_t sleeps);
/* Test hook: polls (~2 us each) after which a tile of a chained launch gives up waiting for a producer; <= 0 restores the
 * default (2^21: seconds).  With a small limit and a long nz_debug_chain_delay the time-out path can be exercised. */
int32_t nz_debug_chain_poll_limit(int32_t polls);

/* ---- the stock stage list as a parameter block --------------------------------------------------------------------
 * NoiseStage -> [KernelFilterStage] -> [FlowMapStage] -> [ErosionKernelJob x n] (README.md:23-32, the metric pipeline) as
 * nz_sharded_create takes it; an iteration count of 0 leaves a stage out.  (Rounds 3 and 4 also offered the list as ONE call
 * on a single tile, nz_terrain_pipeline: two row stripes on two streams of the context.  With the round-3 kernels the
 * overlap bought nothing -- 0.6595 against 0.6413 ms per 4096^2 step stage by stage -- and it was removed in round 5.) */
typedef struct nz_terrain_params {
    int32_t noiseType;            /* NoiseStage.FractalNoise, Noise/NoiseStage.cs:15-24 */
    float hurst, startingAmplitude, stepdown, detuneRate;
    int32_t octaves, noiseSize;
    int32_t filter;               /* KernelFilterType */
    int32_t filterIterations;     /* KernelFilterStage.iterations, 0 = no filter stage */
    int32_t flowIterations;       /* FlowMapStage.iterations, 0 = no flow stage */
    float normMin, normMax;
    int32_t erosionIterations;    /* ErosionKernelJob applications, 0 = none */
} nz_terrain_params;

/* ---- one large grid over the GPUs of a node: row stripes + RCCL neighbour halo exchange (new-framework feature,
 * SURVEY.md 8e; the reference only has independent clamped tiles, Scripts/MeshTileGenerator.cs:166-192, which a host
 * requests one by one through BasePipeline.Schedule, Pipeline/Executable/Pipeline.cs:104-128).  One process per GPU;
 * every process creates a context on its device and joins ONE communicator.  RCCL (librccl.so.1) is opened when the
 * first communicator entry is called: a host that never shards never loads it.
 *
 * nz_comm_unique_id  = ncclGetUniqueId: called by ONE rank, the 128 bytes travel to the others out of band (a file, a
 *                      socket, the launcher's store);
 * nz_comm_init       = ncclCommInitRank on ctx's device (blocks until all `world` ranks have called it) plus the
 *                      communicator's own HIP stream: exchanges run there, ordered against ctx's stream by events, so that
 *                      kernels which read no ghost row overlap them. */
typedef struct nz_comm nz_comm;
#define NZ_COMM_ID_BYTES 128
int32_t nz_comm_unique_id(uint8_t *id_out /* NZ_COMM_ID_BYTES */);
int32_t nz_comm_init(nz_ctx *ctx, const uint8_t *id, int32_t rank, int32_t world, nz_comm **out);
int32_t nz_comm_destroy(nz_comm *comm);
int32_t nz_comm_rank(const nz_comm *comm);
int32_t nz_comm_world(const nz_comm *comm);
int32_t nz_comm_rccl_version(int32_t *version); /* ncclGetVersion of the library actually loaded, e.g. 22606 */

/* Neighbour halo exchange of the stripe `st` that rank comm->rank holds of a grid split into comm->world row stripes
 * (rank r above rank r + 1): for each of the n_planes planes (all of the stripe's shape) the `up_rows` ghost rows above
 * the owned rows are received from rank - 1, which sends the last up_rows rows it owns, and the `down_rows` ghost rows
 * below from rank + 1 (its first down_rows owned rows) -- ncclGroupStart; ncclSend / ncclRecv with rank +- 1;
 * ncclGroupEnd on the communicator's stream, behind everything enqueued on ctx's stream so far.  Ranks at the global
 * border have no neighbour on that side (clamp-to-edge applies there).  Every rank must make the same call.
 *   nz_halo_exchange_begin  : posts the batch and returns; kernels enqueued on ctx afterwards run concurrently with it
 *                             (they must not touch the ghost rows);
 *   nz_halo_exchange_finish : ctx's stream waits for the batch; `out` completes after it;
 *   nz_halo_exchange        : begin + finish. */
int32_t nz_halo_exchange_begin(nz_ctx *ctx, nz_comm *comm, float *const *planes, int32_t n_planes, const nz_stripe *st,
                               int32_t up_rows, int32_t down_rows, nz_handle dep);
int32_t nz_halo_exchange_finish(nz_ctx *ctx, nz_comm *comm, nz_handle *out);
int32_t nz_halo_exchange(nz_ctx *ctx, nz_comm *comm, float *const *planes, int32_t n_planes, const nz_stripe *st,
                         int32_t up_rows, int32_t down_rows, nz_handle dep, nz_handle *out);

/* GetMapRangeJob (Filter/NormalizeJob.cs:17-55) of a grid whose rows are spread over the ranks: `res` = DEVICE {min,
 * max, max - min} of the WHOLE grid on every rank.  Each rank folds its `n_floats` cells (nz_get_map_range), one
 * ncclAllGather carries the per-rank triples, and the same fold runs over the gathered minima and maxima in rank order
 * -- the order the monolithic job walks the grid in, so the result is the monolithic one down to the sign of a zero
 * extreme.  The path's one collective.  comm == NULL: one rank. */
int32_t nz_comm_allgather_range(nz_ctx *ctx, nz_comm *comm, const float *map, size_t n_floats, float *res, float lim_min,
                                float lim_max, nz_handle dep, nz_handle *out);

/* The stock stage list (nz_terrain_params) on a grows x cols grid cut into `stripes` row stripes over all ranks; rank r
 * holds stripes [r * S, (r + 1) * S), S = stripes / world.  The object owns the stripes' planes and the launch plan,
 * which is compiled once: nz_sharded_pipeline replays it.  haloMode:
 *   NZ_HALO_RECOMPUTE     every stripe evaluates the noise on its rows plus the stencil radius of everything downstream
 *                         and each launch produces a window that shrinks by the radius it consumed: no data-path
 *                         communication (closed-form source only);
 *   NZ_HALO_EXCHANGE      before each launch the stripes exchange exactly the ghost rows it consumes.  `overlap`:
 *                         0  the transfers are enqueued on the compute stream itself, between the launch that produced the
 *                            rows and the launch that reads them (no second stream, no event hand-off) -- the fastest form
 *                            on MI355X: a transfer kernel takes ~14 us, and one running beside a stencil launch that fills
 *                            the chip does not finish before that launch does (measured);
 *                         1  on the communicator's stream, while the launch's interior rows, which read no ghost row, run;
 *                            its border rows follow after the wait;
 *                         2  a launch produces the rows its neighbours need first, the exchange for the NEXT launch travels
 *                            while its interior rows run;
 *   NZ_HALO_EXCHANGE_ONCE the source plane's ghost rows for the whole pipeline are exchanged once, then as RECOMPUTE.
 * Transfers between stripes of different ranks AND between two stripes of one rank go through ncclSend / ncclRecv (RCCL
 * runs a send and its matching receive on one device), so that a one-GPU box executes the very code path of a node;
 * comm == NULL (one rank, no RCCL): device copies on the context's stream.
 * externalSource != 0: no noise stage; the caller fills the owned rows of every stripe's source plane (an uploaded
 * height map) before nz_sharded_pipeline.  asRank / asWorld (asWorld > 0): rehearsal on one rank of the geometry rank
 * asRank of asWorld would have; neighbours beyond the process are played by its own stripes (timing only).
 * Same kernels, same operation order as the single-tile entries: the sharded result equals the monolithic grid bit for
 * bit (the only clamps are at the global border). */
typedef struct nz_sharded nz_sharded;
enum nz_halo_mode { NZ_HALO_RECOMPUTE = 0, NZ_HALO_EXCHANGE = 1, NZ_HALO_EXCHANGE_ONCE = 2 };
typedef struct nz_sharded_desc {
    int32_t grows, cols;    /* the global grid */
    int32_t stripes;        /* over all ranks; a multiple of the world size */
    int32_t haloMode;       /* enum nz_halo_mode */
    int32_t overlap;        /* NZ_HALO_EXCHANGE: 0 inline on the compute stream, 1 interior rows first, 2 border rows first */
    int32_t xpos, zpos;     /* GeneratorData.xpos / zpos of the grid's first cell */
    int32_t externalSource;
    int32_t asRank, asWorld;
} nz_sharded_desc;
int32_t nz_sharded_create(nz_ctx *ctx, nz_comm *comm, const nz_sharded_desc *desc, const nz_terrain_params *params,
                          nz_sharded **out);
int32_t nz_sharded_destroy(nz_sharded *sh);
int32_t nz_sharded_local_stripes(const nz_sharded *sh);
/* geometry of local stripe i, its source plane (what the noise stage fills, or the caller) and the plane whose owned
 * rows hold the result after nz_sharded_pipeline; any pointer may be NULL */
int32_t nz_sharded_stripe(const nz_sharded *sh, int32_t i, nz_stripe *st, float **source, float **result);
/* The compiled plan as records of 8 int32 {op, local stripe (-1: all), n, a, b, own0, own1, planes}:
 *   op 1 noise                       rows [own0, own1) of plane `planes`
 *   op 2 exchange begin              n = planes per stripe, a = up_rows, b = down_rows, planes = first plane id
 *   op 3 exchange finish
 *   op 4 kernel filter   n = fused applications, rows [own0, own1), planes = src | dst << 8
 *   op 5 flow map        n = fused iterations, a = first, b = last, planes = src | dst << 8 | state_in << 16 | state_out << 24
 *   op 6 value erosion   n = fused applications
 *   op 7 stage marker    n = 0 noise, 1 filter, 2 flow, 3 erosion, 4 end
 * `records` may be NULL to query the count. */
int32_t nz_sharded_plan(const nz_sharded *sh, int32_t *records, int32_t max_records, int32_t *count);
/* The transfers of the plan's exchanges in the order this rank posts them, as records of 4 int32 {exchange, source rank,
 * destination rank, floats}: a rank posts ncclSend for the records it is the source of and ncclRecv for those it is the
 * destination of, in this order inside one group per exchange.  On a plan-only object (ctx == NULL at creation, asRank of
 * asWorld) these are the lists of that rank of the real job: laid side by side for all ranks, the k-th send of rank a to
 * rank b must be the k-th receive rank b posts from rank a, with the same size (tests/test_sharded_native.py). */
int32_t nz_sharded_transfers(const nz_sharded *sh, int32_t *records, int32_t max_records, int32_t *count);
/* one pass of the pipeline on every local stripe (enqueue only); `marks` (nullable, 5 handles): markers on the context's
 * stream where the noise, filter, flow and erosion launches begin, and at the end */
int32_t nz_sharded_pipeline(nz_ctx *ctx, nz_sharded *sh, nz_handle *marks, nz_handle dep, nz_handle *out);
/* exchanges and payload bytes this rank sends per pass */
int32_t nz_sharded_traffic(const nz_sharded *sh, int32_t *exchanges, size_t *bytes_sent);
/* with timing on, every wait of the compute stream for an exchange (overlap 0: every exchange itself) is bracketed by
 * events; nz_sharded_exchange_ms sums and forgets the brackets recorded so far (host blocks until they have completed; at
 * most 1024 are kept) */
int32_t nz_sharded_set_timing(nz_sharded *sh, int32_t on);
int32_t nz_sharded_exchange_ms(nz_sharded *sh, float *ms);
/* GetMapRangeJob + MapNormalizeValues over the result planes of the whole grid (nz_comm_allgather_range over all
 * stripes of all ranks, then NormalizeMap on the owned rows with the device args) */
int32_t nz_sharded_map_range(nz_ctx *ctx, nz_sharded *sh, float *res, float lim_min, float lim_max, nz_handle dep,
                             nz_handle *out);
int32_t nz_sharded_normalize(nz_ctx *ctx, nz_sharded *sh, const float *args, nz_handle dep, nz_handle *out);

#ifdef __cplusplus
}
#endif
#endif /* NOIZE_HIP_H */
