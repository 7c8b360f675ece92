/*
 * noize_oracle.h -- CPU restatement of noize-job's per-cell terrain hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the parity checker and the timed
 * "CPU restatement of the Burst path" baseline.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; nothing under noize_job_amd/ does.
 *
 * PARITY PINNED AT IMAGE LEVEL ONLY (round 6); NUMERICALLY UNPINNED.  The reference (C# / Unity Burst)
 * ships no tests, golden vectors or fixtures for this path and cannot be compiled or run here (no
 * .NET / Unity toolchain).  The noise bases live in com.unity.mathematics 1.2.1 (package.json:17), which
 * is not in the reference tree; they are restated from the published webgl-noise algorithm (SURVEY.md
 * Appendix A).  What this oracle IS checked against:
 *   - the reference's README screenshots (README.md:23-40, docs~/0-6.jpg: its own output with every
 *     parameter readable beside it): simplex and cellular fBm, Gauss5 x17, the flow map -- Pearson r
 *     0.94 ... 0.9999 with negative controls, tests/test_reference_screenshots.py.  An image pin (JPEG,
 *     8 bit, a display mapping), not the 1e-5 bar;
 *   - the reference's Gaussian coefficient literals (Filter/Kernel/KernelJob.cs:97-105, Filter/Kernel/
 *     Blur/BlurKernels.cs:59-318), bit for bit: tests/golden/gauss_tables.json.
 * Without any reference-held output: cnoise, psrnoise, the 3-D bases, Sin, the min filter, the mesh, the
 * live erosion.
 *
 * Floating-point model: strict IEEE-754 binary32, operations in the order the C# source
 * writes them, no contraction (build with -ffp-contract=off, no -ffast-math); libm
 * sinf/cosf/exp2f/sqrtf.
 *
 * Every function takes a (rows, cols) rectangle; the reference's square tile is
 * rows == cols == resolution.  Index = z*cols + x (Pipeline/Tiles/TileData.cs:76).
 * All citations are relative to /root/reference.
 */
#ifndef NOIZE_ORACLE_H
#define NOIZE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Noise/NoiseStage.cs:15-24 */
enum nzo_noise_type {
    NZO_NOISE_SIN = 0,
    NZO_NOISE_PERLIN = 1,
    NZO_NOISE_PERIODIC_PERLIN = 2,
    NZO_NOISE_SIMPLEX = 3,
    NZO_NOISE_ROTATED_SIMPLEX = 4,
    NZO_NOISE_CELLULAR = 5,
    NZO_NOISE_DOMAIN_ROTATED_PERLIN = 6,
    NZO_NOISE_DOMAIN_ROTATED_SIMPLEX = 7
};

/* Filter/Kernel/KernelJob.cs:79-94 */
enum nzo_kernel_filter_type {
    NZO_GAUSS9_S1 = 0, NZO_GAUSS7_S1, NZO_GAUSS5_S1, NZO_GAUSS3_S1,
    NZO_GAUSS9_S2, NZO_GAUSS7_S2, NZO_GAUSS5_S2, NZO_GAUSS3_S2,
    NZO_SMOOTH3, NZO_SOBEL3_HORIZONTAL, NZO_SOBEL3_VERTICAL, NZO_SOBEL3_2D,
    NZO_PREWITT3_HORIZONTAL, NZO_PREWITT3_VERTICAL
};

/* Mesh/Stage/MeshTileStage.cs:23-26 */
enum nzo_mesh_type { NZO_MESH_SQUARE = 0, NZO_MESH_OVERSHOOT = 1 };

void nzo_set_threads(int n);
int  nzo_get_threads(void);

/* ---- Unity.Mathematics.noise restatements ---- */
float nzo_cnoise2(float x, float y);
float nzo_snoise2(float x, float y);
float nzo_psrnoise2(float x, float y, float perx, float pery, float rot);
void  nzo_cellular2(float x, float y, float *f1, float *f2);
float nzo_cnoise3(float x, float y, float z);
float nzo_snoise3(float x, float y, float z);
/* hash value fed to the gradient table of psrnoise (for range tests) */
float nzo_psr_hash(float px, float py);

/* IMakeNoise getters, Noise/Fractal/Fractal.cs:141-278 */
float nzo_noise_value(int noiseType, float x, float z);

/* Fractal.cs:31-40 */
float nzo_fractal_norm(float hurst, int octaves, float startingAmplitude);
/* Fractal.cs:114-131, one cell */
float nzo_fractal_cell(int noiseType, int x, int z, float hurst, float startingAmplitude,
                       float stepdown, float detuneRate, int octaves, int xpos, int zpos,
                       int noiseSize);
/* FractalJob.ScheduleParallel, Fractal.cs:42-73 */
int nzo_fractal(int noiseType, float *dst, int rows, int cols, float hurst,
                float startingAmplitude, float stepdown, float detuneRate, int octaves,
                int xpos, int zpos, int noiseSize);

/* ---- separable kernel filters ---- */
/* one pass (parallel rows) followed by the serial flush tmp->src,
 * GenericKernelJob.ScheduleParallel KernelJob.cs:31-53 */
void nzo_pass_sample_x(float *src, float *tmp, int rows, int cols, int ksize,
                       const float *kernel, float factor);
void nzo_pass_sample_z(float *src, float *tmp, int rows, int cols, int ksize,
                       const float *kernel, float factor);
void nzo_pass_min_x(float *src, float *tmp, int rows, int cols, int ksize);
void nzo_pass_min_z(float *src, float *tmp, int rows, int cols, int ksize);
/* SeparableKernelFilter.ScheduleSeries KernelJob.cs:165-185 */
void nzo_separable(float *src, float *tmp, int rows, int cols, int ksize, const float *kx,
                   const float *kz, float factor);
/* SeparableKernelFilter.Schedule KernelJob.cs:217-306; returns <0 for Sobel3_2D (needs reduce) */
int nzo_kernel_filter(float *src, float *tmp, int filterType, int rows, int cols);
/* table lookup used by nzo_kernel_filter: writes kx,kz (<=9 floats), factor, size */
int nzo_kernel_filter_table(int filterType, float *kx, float *kz, float *factor, int *ksize);
/* BlurHelper.limitWidth BlurKernels.cs:29-36 */
int nzo_limit_width(int width);
/* GaussianKernel.GetKernel BlurKernels.cs:48-56: writes limitWidth(width) floats, returns that count */
int nzo_gauss_kernel(int sigmaEnum, int width, float *out);
/* GaussFilter.Schedule BlurJob.cs:11-21 / SmoothFilter.Schedule :34-44 */
int nzo_gauss(float *src, float *tmp, int width, int sigmaEnum, int rows, int cols);
int nzo_smooth(float *src, float *tmp, int width, int rows, int cols);
/* ErosionKernelJob.Schedule KernelJob.cs:317-347 */
int nzo_erosion_min(float *src, int rows, int cols);

/* ---- flow map ---- */
void nzo_fill(float *data, int rows, int cols, float value); /* FlowMapComponents.cs:175-202 */
/* FlowMapStepComputeFlow.ScheduleParallel FlowMapJob.cs:37-79 (+4 flush copies) */
void nzo_flow_step(const float *height, const float *water, float *fN, float *fN_buf, float *fS,
                   float *fS_buf, float *fE, float *fE_buf, float *fW, float *fW_buf, int rows,
                   int cols);
/* FlowMapStepUpdateWater.ScheduleParallel FlowMapJob.cs:121-151 */
void nzo_water_step(float *water, float *water_buf, const float *fN, const float *fS,
                    const float *fE, const float *fW, int rows, int cols);
/* FlowMapWriteValues.ScheduleParallel FlowMapJob.cs:189-217 */
void nzo_velocity(float *dst, const float *fN, const float *fS, const float *fE, const float *fW,
                  int rows, int cols);
/* MapNormalizeValues.ScheduleParallel NormalizeJob.cs:70-90; args = {min,max,range} */
void nzo_normalize(float *src, float *tmp, const float *args, int rows, int cols);
void nzo_get_map_range(const float *map, size_t n, float lim_min, float lim_max, float *res);
/* FlowMapStage.ScheduleAll FlowMapStage.cs:124-195; flux planes zero-initialised per run */
int nzo_flowmap(float *src, int rows, int cols, int iterations, float normMin, float normMax);

/* ---- mesh ---- */
/* HeightMapMeshJob.ScheduleParallel HeightMapMeshJob.cs:24-52.
 * vtx: (res+1)^2 records of 12 floats {pos3, normal3, tangent4, uv2}; idx: 6*res*res uint32. */
int nzo_mesh_square_grid(int resolution, float *vtx, uint32_t *idx); /* Mesh/Generators/SharedSquareGridPosition.cs */
int nzo_mesh_heightmap(int meshType, const float *heights, int resolution, int inputResolution,
                       int marginPix, float tileHeight, float tileSize, float *vtx,
                       uint32_t *idx);

/* ---- element-wise stages ---- */
int nzo_constant(float *src, float *tmp, int op, float value, int rows, int cols);      /* Filter/ConstantJob.cs */
int nzo_reduce(float *srcL, const float *srcR, float *tmp, int op, int rows, int cols); /* Filter/ReductionJob.cs */
int nzo_update_flow_from_track(float *pool, float *flow, float *track, int res, float flowLossRate,
                               float surfaceEvaporationRate, float tileHeight); /* LiveErosionDataTypes.cs:869-886 */
int nzo_pool_automata(float *pool, const float *height, int res, int iterations); /* MultiThreadErosionJob.cs:264-327 */
int nzo_crop(const float *input, int inputResolution, float *output, int outputResolution); /* Filter/Sample/CropJob.cs */
int nzo_curve(float *src, float *tmp, const float *curve, int curveSize, int rows, int cols); /* Filter/Curve/CurveJob.cs */

int nzo_thermal_erosion(float *src, int resolution, float talus, float incrementRatio,
                        float meshHeightWidthRatio, int iterations); /* Filter/Kernel/Blur/ThermalErosionFilter.cs */

/* ---- live erosion, particle half (noize_oracle_live.c; BASELINE config 4) ---- */
/* ErosionParameters, Geologic/ParticleErosion/LiveErosionDataTypes.cs:78-100 (field order kept) */
typedef struct nzo_erosion_params {
    float INERTIA, GRAVITY, DRAG, FRICTION, EVAP, EROSION, DEPOSITION, FLOW_HEIGHT_CONTRIBUTION;
    float SLOW_CULL_ANGLE, SLOW_CULL_SPEED, CAPACITY;
    int32_t MAXAGE;
    float TERMINAL_VELOCITY;
    float SURFACE_EVAPORATION_RATE, POOL_PLACEMENT_MULTIPLIER, TRACK_PLACEMENT_MULTIPLIER, FLOW_LOSS_RATE;
    int32_t PILING_RADIUS;
    float MIN_PILE_INCREMENT, PILE_THRESHOLD;
} nzo_erosion_params;
/* a queued BeyerParticle: what the constructors set that is not a constant (:221-237) */
typedef struct nzo_particle { int32_t px, pz; float water; uint32_t pid; } nzo_particle;
float nzo_live_atanf(float x);
float nzo_live_sinf(float x);
float nzo_live_from_fix(long long a);
int nzo_fill_beyer_queue(nzo_particle *queue, int *count, int capacity, int generationRound, int res, int maxParticles,
                         int seed, int concurrency);
int nzo_beyer_descent(const float *height, const float *pool, const float *flow, int res, const nzo_particle *particles,
                      int n, const nzo_erosion_params *ep, int tileHeight, float patchRes, long long *accPool,
                      long long *accTrack, long long *accSed, int *touched);
int nzo_process_beyer_events(float *pool, float *track, float *sediment, int res, const nzo_erosion_params *ep,
                             long long *accPool, long long *accTrack, long long *accSed, int *touched);
int nzo_erode_height_maps(float *height, const float *sediment, int res, const nzo_erosion_params *ep, int tileHeight);
int nzo_pool_automata_drain(float *pool, const float *height, int res, int iterations, nzo_particle *queue, int *count,
                            int capacity);
int nzo_curviture_map(unsigned char *texture, int channel, const float *height, int res, int meshRes, int tileHeight,
                      float patchRes);
int nzo_set_rgba32(unsigned char *texture, int channel, const float *src, int dataRes, int meshRes, float scale);

/* reference-shaped metric pipeline on one tile (bench cpu_baseline): fractal -> kernel filter x G
 * -> flowmap(F) -> erosion x E.  tmp must hold rows*cols floats. */
int nzo_pipeline(float *data, float *tmp, int rows, int cols, int noiseType, float hurst,
                 float startingAmplitude, float stepdown, float detuneRate, int octaves, int xpos,
                 int zpos, int noiseSize, int filterType, int gaussIterations, int flowIterations,
                 float normMin, float normMax, int erosionIterations);

#ifdef __cplusplus
}
#endif
#endif
