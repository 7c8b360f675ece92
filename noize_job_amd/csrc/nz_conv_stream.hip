// nz_conv_stream.hip -- T applications of a 3- or 5-tap separable filter as ONE row-streaming launch (gfx950).
//
// Replaces the same jobs as conv_reg_kernel (nz_filter.hip): GenericKernelJob<KernelTileMutation<KernelSample{X,Z}Operator>,
// RWTileData> x 2 + two flush copies per application (Filter/Kernel/KernelJob.cs:31-53,165-185, operators
// Filter/Kernel/KernelOperators.cs:18-67), `iterations` times (Filter/KernelFilterStage.cs).
//
// The tile kernel holds a 128 x 128 tile in the registers of a 512-thread workgroup and steps all of it through the
// applications in lock-step: boundary rows of every 8-row block cross LDS behind two workgroup barriers per application,
// and the halo of T applications is cut off all four sides.  Here ONE WAVE owns a 128-column strip (two columns per lane)
// and walks down its rows with the applications pipelined behind each other, application j + 1 O = (K - 1) / 2 rows
// behind application j.  A lane keeps, per application, the X-pass results of the last K - 1 rows of its two columns:
// the Z pass of row c runs when the X pass of row c + O arrives, its result is row c of the next application's input in
// the same step.  X-neighbours are the adjacent lanes' registers (wave-shift DPP), Z-neighbours the lane's own.  No LDS,
// no barrier, no flag; HBM traffic is one read and one write of the plane per launch.  Redundant work: the X halo (O T
// columns per side of 128) and the pipeline fill of a row segment (application j starts O (T - 1 - j) rows early).
//
// Arithmetic order is the reference's: X pass sums taps k ascending, Z pass k descending (KernelOperators.cs:34-40,59-65),
// product then add (no FMA contraction), then * factor.  Clamp-to-edge (Pipeline/Tiles/TileData.cs:72-77) is applied per
// pass: a tap beyond the grid takes the border cell's value of THAT pass's input -- the lane holding the grid's first /
// last column substitutes its own values for the missing neighbours, a stage that starts on the grid's first row fills
// its window with that row's X-pass result, and one that ends on the last row repeats it.
#include <cstdlib>

#include "nz_internal.hpp"

namespace {

constexpr int CS_TW = 128;  // columns per strip, halo included

__device__ __forceinline__ float cs_prev(float v) {  // lane i <- lane i-1 (lane 0: 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float cs_next(float v) {  // lane i <- lane i+1 (lane 63: 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

typedef unsigned cs_v2u __attribute__((ext_vector_type(2)));

// Buffer accesses: the row goes in the scalar offset, the lane's column in the vector offset, and a lane that must not
// store carries a vector offset beyond num_records -- the hardware drops the access, no branch splits the step (behind a
// conditional store the compiler would wait for every outstanding load, i.e. for the rows prefetched last).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t cs_rsrc(const float *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, (int)bytes, 0x00020000);
}

template <int T>
struct cs_bounds {
    int in_lo[T], in_hi[T];  // input rows stage j takes in (in_hi: rows repeated below the grid's last row included)
    int lo[T], hi[T];        // rows it puts out
};

// One step: every stage takes in one row and puts out the row O above it.
//   COND : stages outside their row range are skipped and the grid's first / last rows get their clamped windows
//          (pipeline fill and drain); the steps in between run without tests;
//   XEDGE: the strip touches the grid's first / last column.
// FAST (NZ_FLOAT_FAST): the tap sums as FMAs, same tap order -- the very sums of conv_tile's tolerance form (nz_filter.hip),
// so both forms still agree bit for bit within the mode
template <bool FAST>
__device__ __forceinline__ float cs_acc(float total, float v, float k) {
    return FAST ? __builtin_fmaf(v, k, total) : total + v * k;
}

template <int KS, int T, bool UNIT, bool COND, bool XEDGE, bool FAST>
__device__ __forceinline__ void cs_step(float (&X)[T][KS - 1][2], const int t, const float2 vin, const cs_bounds<T> &b,
                                        const nz_kernel_taps &taps, const nz_geom &g, __amdgpu_buffer_rsrc_t rdst,
                                        const unsigned vo_st0, const unsigned vo_st1, const bool lane_x0,
                                        const bool lane_x1, const bool lane_x1o) {
    constexpr int O = (KS - 1) / 2, HW = KS - 1;
    float v[2] = {vin.x, vin.y};
#pragma unroll
    for (int j = 0; j < T; j++) {
        const int r = t - O * j;
        const bool act = !COND || (r >= b.in_lo[j] && r < b.in_hi[j]);
        float out[2] = {0.0f, 0.0f};
        bool em = false;
        if (act) {
            float xn[2];
            if (!COND || r <= g.zc1) {
                // ---- X pass of row r (KernelSampleXOperator: taps k ascending)
                float w[2 + 2 * O];
                if (XEDGE && lane_x1o) v[1] = v[0];  // odd row length: the lane's second column is beyond the grid
#pragma unroll
                for (int o = 0; o < O; o++) {
                    // columns x0 - O + o (from the lane on the left) and x0 + 2 + o (from the lane on the right)
                    float l = cs_prev(v[(2 - O + o) & 1]), rr = cs_next(v[o & 1]);
                    if (O == 2) {
                        l = cs_prev(v[o]);
                        rr = cs_next(v[o]);
                    }
                    if (XEDGE) {
                        l = lane_x0 ? v[0] : l;
                        rr = (lane_x1 || lane_x1o) ? v[1] : rr;
                    }
                    w[o] = l;
                    w[2 + O + o] = rr;
                }
                w[O] = v[0];
                w[O + 1] = v[1];
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    float total = w[e] * taps.kx[0];  // 0 + a*b == a*b
#pragma unroll
                    for (int kk = 1; kk < KS; kk++) total = cs_acc<FAST>(total, w[e + kk], taps.kx[kk]);
                    xn[e] = UNIT ? total : total * taps.factor;
                }
            } else {  // below the grid's last row: the clamped tap repeats that row's X-pass result
                xn[0] = X[j][HW - 1][0];
                xn[1] = X[j][HW - 1][1];
            }
            if (COND && r == b.in_lo[j]) {  // first row of this stage: rows above it are itself (grid's first row) or unused
#pragma unroll
                for (int i = 0; i < HW; i++) {
                    X[j][i][0] = xn[0];
                    X[j][i][1] = xn[1];
                }
            }
            // ---- Z pass of row c = r - O (KernelSampleZOperator: k descending, Kernel[k_off - k])
            const int c = r - O;
            em = !COND || (c >= b.lo[j] && c < b.hi[j]);
            if (em) {
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    float total = xn[e] * taps.kz[0];
#pragma unroll
                    for (int kk = 1; kk < KS; kk++) total = cs_acc<FAST>(total, X[j][HW - kk][e], taps.kz[kk]);
                    out[e] = UNIT ? total : total * taps.factor;
                }
            }
#pragma unroll
            for (int i = 0; i + 1 < HW; i++) {
                X[j][i][0] = X[j][i + 1][0];
                X[j][i][1] = X[j][i + 1][1];
            }
            X[j][HW - 1][0] = xn[0];
            X[j][HW - 1][1] = xn[1];
        }
        if (j == T - 1) {
            if (em) {
                const unsigned so = (unsigned)(r - O) * (unsigned)g.pitch * 4u;
                if (!XEDGE) {
                    cs_v2u d = {__float_as_uint(out[0]), __float_as_uint(out[1])};
                    __builtin_amdgcn_raw_buffer_store_b64(d, rdst, (int)vo_st0, (int)so, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(out[0]), rdst, (int)vo_st0, (int)so, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(out[1]), rdst, (int)vo_st1, (int)so, 0);
                }
            }
        } else {
            v[0] = out[0];
            v[1] = out[1];
        }
    }
}

template <bool XEDGE>
__device__ __forceinline__ float2 cs_load_row(__amdgpu_buffer_rsrc_t rsrc, const nz_geom &g, int row, unsigned vo0,
                                              unsigned vo1) {
    const unsigned so = (unsigned)row * (unsigned)g.pitch * 4u;
    if (!XEDGE) {
        cs_v2u d = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)vo0, (int)so, 0);
        return make_float2(__uint_as_float(d.x), __uint_as_float(d.y));
    }
    return make_float2(__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)vo0, (int)so, 0)),
                       __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)vo1, (int)so, 0)));
}

template <int KS, int T, bool UNIT, bool XEDGE, bool FAST>
__device__ __forceinline__ void conv_stream_body(const float *__restrict__ src, float *__restrict__ dst, const nz_geom &g,
                                                 const nz_kernel_taps &taps, const int lx0, const int s0, const int s1,
                                                 const unsigned plane_bytes) {
    constexpr int O = (KS - 1) / 2, HW = KS - 1, U = KS;  // U steps per trip: a window is KS registers deep while a step runs
    constexpr int HX = (O * T + 1) & ~1;
    const int lane = threadIdx.x;
    const int gx = lx0 + 2 * lane;
    const bool lane_x0 = gx == 0, lane_x1 = gx + 1 == g.cols - 1, lane_x1o = gx == g.cols - 1;
    const bool st_lane = 2 * lane >= HX && 2 * lane < CS_TW - HX;
    const __amdgpu_buffer_rsrc_t rsrc = cs_rsrc(src, plane_bytes), rdst = cs_rsrc(dst, plane_bytes);
    // vector offsets: loads clamp the column into the grid, stores go beyond num_records where nothing may be stored
    const unsigned vo_ld0 = (unsigned)min(max(gx, 0), g.cols - 1) * 4u, vo_ld1 = (unsigned)min(max(gx + 1, 0), g.cols - 1) * 4u;
    const unsigned vo_st0 = st_lane && gx >= 0 && gx < g.cols ? (unsigned)gx * 4u : plane_bytes;
    const unsigned vo_st1 = st_lane && gx + 1 >= 0 && gx + 1 < g.cols ? (unsigned)(gx + 1) * 4u : plane_bytes;
    cs_bounds<T> b;
#pragma unroll
    for (int j = 0; j < T; j++) {
        const int m = O * (T - 1 - j);
        b.lo[j] = max(g.zc0, s0 - m);
        b.hi[j] = min(g.zc1 + 1, s1 + m);
        b.in_lo[j] = max(g.zc0, b.lo[j] - O);
        // rows below the grid's last row are taken in as repeats of it while outputs still need them
        b.in_hi[j] = b.hi[j] + O;
    }
    float X[T][HW][2];
#pragma unroll
    for (int j = 0; j < T; j++)
#pragma unroll
        for (int i = 0; i < HW; i++) X[j][i][0] = X[j][i][1] = 0.0f;

    // rows are taken in from t = in_lo[0]; the last stage puts out row s1 - 1 at t = s1 - 1 + O T
    const int t0 = b.in_lo[0], t1 = s1 + O * T;
    float2 P[U];  // rows t .. t + U - 1, prefetched
#pragma unroll
    for (int u = 0; u < U; u++) P[u] = cs_load_row<XEDGE>(rsrc, g, min(t0 + u, g.zc1), vo_ld0, vo_ld1);
    int t = t0;
#define NZ_CS_TRIP(C)                                                                                                 \
    _Pragma("unroll") for (int u = 0; u < U; u++) {                                                                   \
        const float2 vin = P[u];                                                                                      \
        cs_step<KS, T, UNIT, C, XEDGE, FAST>(X, t + u, vin, b, taps, g, rdst, vo_st0, vo_st1, lane_x0, lane_x1, lane_x1o);  \
        /* the row U steps ahead, asked for once this step's row is dead: it lands in the same registers */           \
        P[u] = cs_load_row<XEDGE>(rsrc, g, min(t + u + U, g.zc1), vo_ld0, vo_ld1);                                    \
        /* nothing moves across steps: left alone, the scheduler gathers the trip's loads at its end and their uses  */ \
        /* at its start, and a row is then waited for a few hundred cycles after it was asked for, not U steps later */ \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
    }
    // pipeline fill: every stage has put out its first row once t >= s0 + O T
    const int tfill = min(t1, s0 + O * T);
    for (; t < tfill; t += U) { NZ_CS_TRIP(true) }
    // steady state: every stage inside its row range, every row taken in a real one
    const int tsteady = min(t1, g.zc1 + 1);
    for (; t + U <= tsteady; t += U) { NZ_CS_TRIP(false) }
    // the last steps; the drain of a segment that ends on the grid's last row (steps past t1 do nothing)
    for (; t < t1; t += U) { NZ_CS_TRIP(true) }
#undef NZ_CS_TRIP
}

template <int KS, int T, bool UNIT, bool FAST>
__global__ __launch_bounds__(64) void conv_stream_kernel(const float *__restrict__ src, float *__restrict__ dst, nz_geom g,
                                                        nz_kernel_taps taps, int S, int nstrips, int aligned) {
    constexpr int O = (KS - 1) / 2, HX = (O * T + 1) & ~1, OW = CS_TW - 2 * HX;
    const int strip = blockIdx.x % nstrips, seg = blockIdx.x / nstrips;
    const int lx0 = strip * OW - HX;
    const int s0 = g.or0 + seg * S, s1 = min(s0 + S, g.or1);
    const size_t off = blockIdx.y * g.bstride;  // batched launch: one independent grid per blockIdx.y
    const unsigned plane_bytes = (unsigned)g.rows * (unsigned)g.pitch * 4u;
    const bool inner = aligned && lx0 > 0 && lx0 + CS_TW < g.cols;
    if (inner) conv_stream_body<KS, T, UNIT, false, FAST>(src + off, dst + off, g, taps, lx0, s0, s1, plane_bytes);
    else conv_stream_body<KS, T, UNIT, true, FAST>(src + off, dst + off, g, taps, lx0, s0, s1, plane_bytes);
}

template <int KS, int T>
int32_t launch_stream(hipStream_t s, const float *src, float *dst, const nz_geom &g, const nz_kernel_taps &k, int waves) {
    constexpr int O = (KS - 1) / 2, HX = (O * T + 1) & ~1, OW = CS_TW - 2 * HX;
    const int nstrips = (g.cols + OW - 1) / OW, rows = g.or1 - g.or0;
    const long long per = (long long)nstrips * g.count;
    int nseg = (int)(waves / per > 0 ? waves / per : 1);
    int S = (rows + nseg - 1) / nseg;
    if (S < 16) S = 16;
    nseg = (rows + S - 1) / S;
    const uintptr_t bits = reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | (uintptr_t)(g.pitch * 4) |
                           (uintptr_t)(g.bstride * 4);
    const int aligned = (bits & 7) == 0;
    const dim3 grid((unsigned)(nstrips * nseg), g.count);
    const bool fast = nz_tls_float_mode >= NZ_FLOAT_FAST;
#define NZ_CSL(U, F) NZ_LAUNCH((conv_stream_kernel<KS, T, U, F>), grid, dim3(64), 0, s, src, dst, g, k, S, nstrips, aligned)
    if (k.factor == 1.0f) {
        if (fast) NZ_CSL(true, true); else NZ_CSL(true, false);
    } else {
        if (fast) NZ_CSL(false, true); else NZ_CSL(false, false);
    }
#undef NZ_CSL
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

}  // namespace

// largest T the streaming form holds (0: this tap count has none)
int nz_conv_stream_max(int ksize) { return ksize == 3 || ksize == 5 ? 6 : 0; }

// NZ_CONV_STREAM=1: the streaming form for grids of 40 M cells and more; 2: every size (test matrix); default 0: never.
// Measured against the tile kernels (chained at 4096^2, separate launches elsewhere), Gauss5 x17: 4096^2 0.228 against
// 0.206 ms (a segment of ~50 rows spends a fifth of its steps filling the pipeline), 8192^2 0.743 against 0.767,
// 16384^2 2.90 against 2.52, a 2048 x 16384 stripe 0.432 against 0.391: per output cell it executes 7 % fewer
// instructions, but its per-wave streams (512 B per row, one row stride apart) use HBM worse than tile loads, and four
// launches of ~3000 long waves each end in a tail of lone waves.  Kept as the barrier-free reference form; buffer offsets
// are 32 bits with room for the out-of-range marker, so planes must stay below 2 GiB.
bool nz_conv_stream_wanted(const nz_geom &g, int ksize, int T) {
    static const int mode = getenv("NZ_CONV_STREAM") ? atoi(getenv("NZ_CONV_STREAM")) : 1;
    if (mode == 0 || T < 1 || T > nz_conv_stream_max(ksize)) return false;
    if ((size_t)g.rows * g.pitch * 4 >= ((size_t)1 << 31)) return false;
    if (mode == 2) return true;
    // Where it has measured faster than the (chained) tile kernel, Gauss5 x17: 6656^2 0.514 against 0.531 ms, 8192^2 0.741
    // against 0.759, 11000^2 1.27 against 1.36 (round 4).  Below ~40 M cells a segment is mostly pipeline fill, at 16384^2 it
    // loses (2.90 against 2.52 ms), and a wide, short stripe (2048 x 16384: 0.432 against 0.391) has too few rows per wave.
    const long long cells = (long long)g.cols * (g.or1 - g.or0) * g.count;
    return cells >= 40ll * 1024 * 1024 && cells < 200ll * 1024 * 1024 && g.cols <= 2 * (g.or1 - g.or0);
}

int32_t nz_launch_conv_stream(hipStream_t s, const float *src, float *dst, const nz_geom &g, const nz_kernel_taps &k, int T) {
    constexpr int waves = 4096;  // (3072: -2 % in FAST mode only; fewer lose)
    if (g.or1 <= g.or0) return NZ_OK;
#define NZ_CS(KS_)                                                   \
    switch (T) {                                                     \
        case 1: return launch_stream<KS_, 1>(s, src, dst, g, k, waves); \
        case 2: return launch_stream<KS_, 2>(s, src, dst, g, k, waves); \
        case 3: return launch_stream<KS_, 3>(s, src, dst, g, k, waves); \
        case 4: return launch_stream<KS_, 4>(s, src, dst, g, k, waves); \
        case 5: return launch_stream<KS_, 5>(s, src, dst, g, k, waves); \
        case 6: return launch_stream<KS_, 6>(s, src, dst, g, k, waves); \
    }
    if (k.ksize == 3) { NZ_CS(3) }
    if (k.ksize == 5) { NZ_CS(5) }
#undef NZ_CS
    nz_set_error("conv_stream: kernelSize %d x T=%d unsupported", k.ksize, T);
    return NZ_ERR_INVALID;
}
