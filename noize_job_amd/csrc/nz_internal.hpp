// nz_internal.hpp -- shared declarations of libnoize_hip (host side + launcher prototypes).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <vector>

#include "../../include/noize_hip.h"

// ---- error plumbing -------------------------------------------------------------------------
void nz_set_error(const char *fmt, ...);

#define NZ_HIP(expr)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            nz_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                         __LINE__);                                                       \
            return NZ_ERR_HIP;                                                            \
        }                                                                                 \
    } while (0)

#define NZ_REQUIRE(cond, ...)        \
    do {                             \
        if (!(cond)) {               \
            nz_set_error(__VA_ARGS__); \
            return NZ_ERR_INVALID;   \
        }                            \
    } while (0)

// ---- context --------------------------------------------------------------------------------
struct nz_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    // JobHandle ring: the handle with sequence number q lives in events[q % size] while q > last_seq - size.
    // A handle VALUE is (ctx id << NZ_HANDLE_SEQ_BITS) | q, so that a handle names its context: a dependency on
    // another context's handle becomes a hipStreamWaitEvent (JobHandle dependencies across pipelines,
    // Pipeline/Executable/ReducePipeline.cs:82-148).  `hmx` guards the ring against a foreign context's lookup.
    std::vector<hipEvent_t> events;
    uint64_t last_seq = 0;
    uint32_t id = 0;
    std::mutex hmx;
    // psrnoise gradient tables (rot 0 and rot 0.62), built with the host libm
    float *d_rgrad = nullptr;
    // snoise lattice tables: int T1[292] (16*permute(i)) followed by float4 T2[580] (gradient of permute(j))
    void *d_simplex = nullptr;
    // stage scratch owned by the ctx (grown on demand)
    float *scratch = nullptr;
    size_t scratch_floats = 0;
    // chained launches (nz_launch_conv_chain): tile flags (grown on demand, never cleared: they carry an epoch)
    int *chain_flags = nullptr;
    size_t chain_flags_n = 0;
    // the error words in mapped host memory ([0]: a chained filter launch gave up waiting, [1]: the pile solver's ticket launch
    // did -- one word per kind, so neither overwrites the other), and their device address
    unsigned *chain_err = nullptr, *chain_err_dev = nullptr;
    unsigned chain_epoch = 0;
    bool chain_off = false;  // a chained launch once timed out on this context: separate launches from then on
    // Which work a chained launch's time-out belongs to: a failing tile lowers *chain_err_epoch (device memory, atomicMin) to
    // its launch's epoch before it raises the flag; the host remembers the handle sequence number each of its last chained
    // launches was issued at.  Once seen, the failure is reported (NZ_ERR_RETRY) by every wait on a handle in [retry_lo,
    // retry_hi] -- issued at or after the failing launch and before the failure was first REPORTED (retry_open: the window's
    // upper end follows last_seq until then) -- and by the next nz_ctx_synchronize; a wait on an older handle (another
    // pipeline's, a fence recorded before the stage) completes clean.
    unsigned *chain_err_epoch = nullptr;
    struct chain_mark { unsigned epoch; uint64_t seq; };
    chain_mark chain_marks[64] = {};
    uint64_t retry_lo = 0, retry_hi = 0;
    bool retry_sync_pending = false, retry_open = false;
    // live erosion, the pile solver's ticket launch (nz_live.hip): safe mode (nz_ctx_set_pile_safe) waits for it and runs the job
    // again colour by colour should it ever give up -- for good on this context (pile_ticket_off)
    bool pile_safe = false, pile_ticket_off = false;
    int pile_retries = 0;
    bool handle_rides = false;    // this entry's handle may ride on its last kernel launch (nz_ctx_handle_rides)
    uint64_t armed_seq = 0;       // ... and this is the sequence number reserved for it (nz_ctx_arm_last_launch)
    // pool automaton, sparse form (nz_pool_job in nz_stages.cpp): {entries, done, -} in device memory, and in mapped host
    // memory what the last job that ran reported: job number << 32 | non-empty mask words it found
    int *pool_ctl = nullptr;
    unsigned long long *pool_hint = nullptr, *pool_hint_dev = nullptr;
    unsigned long long pool_seq = 0;
    int float_mode = NZ_FLOAT_STRICT;  // nz_ctx_set_float_mode
};
int32_t nz_ctx_pool_state(nz_ctx *ctx);  // allocates the three on first use

#define NZ_TRY_(expr)             \
    do {                          \
        int32_t rc_t_ = (expr);    \
        if (rc_t_) return rc_t_;    \
    } while (0)

constexpr int NZ_HANDLE_SEQ_BITS = 40;
constexpr uint64_t NZ_HANDLE_SEQ_MASK = (1ull << NZ_HANDLE_SEQ_BITS) - 1;
int32_t nz_ctx_begin(nz_ctx *ctx, nz_handle dep);           // set device, order the stream after `dep`
int32_t nz_ctx_finish(nz_ctx *ctx, nz_handle *out);         // record the JobHandle marker
int32_t nz_ctx_scratch(nz_ctx *ctx, size_t floats, float **out);

// ---- kernel parameter blocks ----------------------------------------------------------------
constexpr int NZ_MAX_KSIZE = 25;
// psrnoise tables.  The hash arguments are iu = xw + 0.5 yw and yw with xw = fmod(., 1010), yw = fmod(., 102):
// integers in (-1061, 1061) and (-102, 102), negative for negative lattice coordinates (C fmod keeps the sign).
//   T1[i] = 8 * (permute(i - O1) + O2): the first, un-reduced permute, argument in [-O1, T1 - O1)
//   T2[rot][j] = (cos u, sin u) of the gradient hashed from a = j - O2 (second permute + rgrad2), a = permute + yw
constexpr int NZ_PSR_O1 = 1064, NZ_PSR_T1 = NZ_PSR_O1 + 1064;
constexpr int NZ_PSR_O2 = 104, NZ_PSR_T2 = NZ_PSR_O2 + 392;

struct nz_kernel_taps {
    float kx[NZ_MAX_KSIZE];
    float kz[NZ_MAX_KSIZE];
    float factor;
    int ksize;
};

struct nz_fractal_params {
    float posx, posz;     // (float)xpos, (float)(zpos + first row)
    float noise_size;     // (float)NoiseSize
    float G;              // exp2f(-hurst), host libm
    float amp;            // StartingAmplitude
    float stepdown, detune_rate;
    float norm;           // CalcFractalNormValue
    float fmax;           // max over the octaves of |frequency| (same fp32 recurrence as the kernels; NaN if it overflows)
    int octaves;
    // batched launch (one grid per blockIdx.y): grids `bstride` floats apart, {xpos, zpos} of grid b at positions[2b]
    const int32_t *positions = nullptr;
    size_t bstride = 0;
    int rows_per_wg = 8;  // rows one workgroup walks through (8 amortises the table staging; fewer for small grids)
};

// plane geometry handed to every stencil kernel: clamp rows are the global border seen from the
// buffer, intersected with the buffer itself.
struct nz_geom {
    int cols;      // row length
    int pitch;     // floats between rows
    int rows;      // rows in the buffer
    int zc0, zc1;  // inclusive clamp range for row reads (buffer coordinates)
    int or0, or1;  // rows to produce [or0, or1)
    // batched launches: `count` independent grids of this geometry, `bstride` floats apart in every plane;
    // the kernels take the grid index from blockIdx.y
    int count = 1;
    size_t bstride = 0;
};

inline nz_geom nz_geom_batch(int res, int count) {
    nz_geom g{res, res, res, 0, res - 1, 0, res};
    g.count = count;
    g.bstride = (size_t)res * res;
    return g;
}
// floats covered by rows [or0, or1) of every grid of the batch when the grids are stored back to back
inline size_t nz_geom_span(const nz_geom &g) {
    return g.count > 1 ? (size_t)g.count * g.bstride : (size_t)(g.or1 - g.or0) * g.pitch;
}

inline nz_geom nz_geom_from_stripe(const nz_stripe &s) {
    nz_geom g;
    g.cols = s.cols;
    g.pitch = s.pitch > 0 ? s.pitch : s.cols;
    g.rows = s.rows;
    int lo = -s.grow0, hi = s.grows - 1 - s.grow0;
    g.zc0 = lo > 0 ? lo : 0;
    g.zc1 = hi < s.rows - 1 ? hi : s.rows - 1;
    g.or0 = s.own0;
    g.or1 = s.own1;
    return g;
}

inline nz_geom nz_geom_tile(int res) { return nz_geom{res, res, res, 0, res - 1, 0, res}; }

// ---- handles that ride on a launch --------------------------------------------------------------------------------------
// A JobHandle used to be a hipEventRecord behind the entry's work: a barrier packet of its own, ~2.8 us of the stream.  A
// kernel launch can carry the event instead (hipExtLaunchKernelGGL's stop event: the dispatch's own completion signal,
// tools/probe_write_value.hip measures no cost at all).  An entry whose LAST enqueued operation is the last launch of a
// helper says so (nz_ctx_handle_rides); the helper arms the launch it knows to be its last (nz_ctx_arm_last_launch); that
// launch -- NZ_LAUNCH instead of hipLaunchKernelGGL -- takes the event; nz_ctx_finish hands out the handle, or records an
// event the old way when nothing took it.  nz_ctx_begin disarms whatever an entry that failed left behind.
extern thread_local hipEvent_t nz_tls_stop_event;
// the float mode of the context whose entry this thread is running (set by nz_ctx_begin: every launcher runs behind one)
extern thread_local int nz_tls_float_mode;
#define NZ_LAUNCH(kernel, grid, block, lds, stream, ...)                                                  \
    do {                                                                                                   \
        if (nz_tls_stop_event) {                                                                           \
            hipEvent_t nz_stop_ = nz_tls_stop_event;                                                       \
            nz_tls_stop_event = nullptr;                                                                   \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, nullptr, nz_stop_, 0, __VA_ARGS__);    \
        } else {                                                                                           \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                             \
        }                                                                                                  \
    } while (0)
void nz_ctx_handle_rides(nz_ctx *ctx, bool wanted);
void nz_ctx_arm_last_launch(nz_ctx *ctx);

// every stage launch goes through here: one place that knows which stream a context's work runs on
template <class F>
inline int32_t launch_on_ctx(nz_ctx *ctx, const nz_geom &g, F launch) {
    return launch(ctx->stream, g);
}

int32_t nz_check_stripe(const nz_stripe *st, int halo, int halo_below = -1);  // rows needed above / below the owned ones

// nz_stages.cpp, for the sharded plan (nz_comm.cpp)
int32_t nz_filter_taps(int32_t filter, nz_kernel_taps *t);  // KernelFilterType -> taps
int nz_conv_tcap(int ksize);                                // applications fused per launch (0: no fused kernel)
int32_t nz_fractal_rows(nz_ctx *ctx, hipStream_t stream, int noiseType, float *dst, int rows, int cols, int pitch, float hurst,
                        float amp, float stepdown, float detune, int octaves, int xpos, int zpos_first_row, int noiseSize);

// ---- launchers (defined in the .hip files) ---------------------------------------------------
// `positions` (nullable, device): {xpos, zpos} per grid of a batched launch of `count` grids `bstride` floats apart
int32_t nz_launch_fractal(hipStream_t s, int noiseType, float *dst, int rows, int cols, int pitch,
                          const nz_fractal_params &p, const float *d_rgrad, const void *d_simplex, int count = 1,
                          size_t bstride = 0, const int32_t *positions = nullptr);

int nz_conv_max_fused(int ksize);
// T fused applications of (X pass, Z pass) src -> dst on rows [or0, or1)
int32_t nz_launch_conv_fused(hipStream_t s, const float *src, float *dst, const nz_geom &g,
                             const nz_kernel_taps &k, int T);
// the same as one row-streaming launch (nz_conv_stream.hip; 3 and 5 taps)
int nz_conv_stream_max(int ksize);
bool nz_conv_stream_wanted(const nz_geom &g, int ksize, int T);
int32_t nz_launch_conv_stream(hipStream_t s, const float *src, float *dst, const nz_geom &g, const nz_kernel_taps &k, int T);
// L launches as one grid with tile-level dependencies; see nz_filter.hip
int nz_conv_chain_items(int ksize, const nz_geom &g, const int *Ts, int L);
int nz_cu_count();  // compute units of the current device (256 on MI355X): a launch of at most this many workgroups is one round
bool nz_conv_small_grid(int ksize, const nz_geom &g);  // 64-row tiles
bool nz_conv_tiny_grid(int ksize, const nz_geom &g);
int32_t nz_launch_conv_chain(hipStream_t s, float *plane0, float *plane1, const nz_geom &g, const nz_kernel_taps &k,
                             const int *Ts, int L, int *flags, unsigned epoch, unsigned *err_host, unsigned *err_epoch);
int32_t nz_ctx_chain_state(nz_ctx *ctx, size_t items, int **flags, unsigned *epoch, unsigned **err_host, unsigned **err_epoch);
int32_t nz_ctx_error_word(nz_ctx *ctx, unsigned **err_host);  // mapped host memory, device address: [0] chained filter, [1] pile solver
// one whole application of a wide odd kernel (11..25 taps), src -> dst
bool nz_conv_has_wide(int ksize);
int32_t nz_launch_conv_wide(hipStream_t s, const float *src, float *dst, const nz_geom &g, const nz_kernel_taps &k);
// single unfused passes (src -> dst): any kernelSize <= 25, and single applications
int32_t nz_launch_conv_pass_x(hipStream_t s, const float *src, float *dst, const nz_geom &g,
                              const nz_kernel_taps &k);
int32_t nz_launch_conv_pass_z(hipStream_t s, const float *src, float *dst, const nz_geom &g,
                              const nz_kernel_taps &k);
// E fused applications of the {-1,0} min window: src -> dst
int nz_erosion_max_fused();
int32_t nz_launch_erosion_fused(hipStream_t s, const float *src, float *dst, const nz_geom &g, int E);
// one reference min pass, window k in [-k_off, k_off), along x (along_z == 0) or z

int32_t nz_launch_fill(hipStream_t s, float *data, size_t n, float value);
int32_t nz_launch_copy(hipStream_t s, float *dst, const float *src, size_t n);
// delegate-level flow kernels (single tile semantics, global-memory stencils)
int32_t nz_launch_flow_step(hipStream_t s, const float *h, const float *w, const float *fN, const float *fS,
                            const float *fE, const float *fW, float *oN, float *oS, float *oE, float *oW,
                            const nz_geom &g);
int32_t nz_launch_water_step(hipStream_t s, const float *w, float *w_out, const float *fN, const float *fS,
                             const float *fE, const float *fW, const nz_geom &g);
int32_t nz_launch_velocity(hipStream_t s, float *dst, const float *fN, const float *fS, const float *fE,
                           const float *fW, const nz_geom &g, int normalize, float nmin, float nrange);
int32_t nz_launch_normalize(hipStream_t s, const float *src, float *dst, size_t n, float nmin, float nrange);
// n <= nz_flow_fused_max() iterations per launch on an on-chip tile; state planes are {water,fN,fS,fE,fW}
int nz_flow_fused_max();
// h_out (nullable): the launch also stores its interior height cells there (a private copy, so that a
// later launch may overwrite the caller's plane)
int32_t nz_launch_flow_fused(hipStream_t s, const float *h, const float *const in[5], float *const out[5], float *dst,
                             float *h_out, const nz_geom &g, int n, int first, int last, float nmin, float nrange);

// the whole stage (first && last) as one row-streaming launch (nz_flow_stream.hip)
bool nz_flow_stream_wanted(const nz_geom &g, int n);
int32_t nz_launch_flow_stream(hipStream_t s, const float *h, float *dst, const nz_geom &g, int n, float nmin, float nrange);

int32_t nz_launch_mesh_planar(hipStream_t s, void *vertices, uint32_t *indices, int res);
int32_t nz_launch_mesh(hipStream_t s, int meshType, void *vertices, uint32_t *indices, int res, int in_res,
                       float tile_height, float tile_size, const float *heights, int count = 1, int index16 = 0);

// element-wise stages (nz_elementwise.hip)
int32_t nz_launch_constant(hipStream_t s, int op, float *data, size_t n, float c);
int32_t nz_launch_reduce(hipStream_t s, int op, float *l, const float *r, size_t n);
int32_t nz_launch_flow_from_track(hipStream_t s, float *pool, float *flow, float *track, size_t n, float flowLossRate,
                                  float evaporation);
// drain_hdr / drain_data (nullable): the particle queue a drained pool leaves through (PoolAutomataJob, drainParticles)
size_t nz_map_range_scratch_floats();
int32_t nz_launch_map_range(hipStream_t s, const float *map, size_t n, float lim_min, float lim_max, float *res, void *scratch);
int32_t nz_launch_normalize_args(hipStream_t s, float *data, size_t n, const float *args);
// gathered {min, max, range} triples -> mins[n], maxs[n]; {lo[0], hi[1]} -> res = {min, max, max - min}
int32_t nz_launch_range_split(hipStream_t s, const float *triples, int n, float *mins, float *maxs);
int32_t nz_launch_range_compose(hipStream_t s, const float *lo, const float *hi, float *res);
size_t nz_pool_automata_mask_words(int res);
int32_t nz_launch_pool_automata_masks(hipStream_t s, const float *pool, int res, unsigned *mask, int *ctl, int with_list);
int32_t nz_launch_pool_automata_clean(hipStream_t s, const float *pool, int res, unsigned *mask, int *ctl);
int32_t nz_launch_pool_automata_pass(hipStream_t s, float *pool, const float *height, int res, int xoff, int zoff,
                                     int32_t *drain_hdr = nullptr, nz_particle *drain_data = nullptr,
                                     unsigned *mask = nullptr, int *ctl = nullptr);
// the whole job as one launch of one workgroup, from the list of non-empty mask words (nz_elementwise.hip)
int32_t nz_launch_pool_automata_sparse(hipStream_t s, float *pool, const float *height, int res, int iterations, int limit,
                                       int dense_follows, int32_t *drain_hdr, nz_particle *drain_data, unsigned *mask, int *ctl,
                                       unsigned long long *hint_dev, unsigned long long seq);
int32_t *nz_particle_queue_hdr(nz_particle_queue *q);
nz_particle *nz_particle_queue_data(nz_particle_queue *q);
int32_t nz_launch_crop(hipStream_t s, const float *in, int in_res, float *out, int out_res);
int32_t nz_launch_curve(hipStream_t s, float *data, size_t n, const float *curve, int curveSize);
bool nz_thermal_pair_fits(int resolution);
int32_t nz_launch_thermal_pair(hipStream_t s, float *data, int resolution, int zodd, float maxDiff, float increment);
int32_t nz_launch_thermal_phase(hipStream_t s, float *data, int resolution, int flip, float maxDiff, float increment);
