// nz_flow.hip -- pipe-model flow map (gfx950).
//
// Replaces FillArrayJob, FlowMapStepComputeFlow<ComputeFlowStep,...>, FlowMapStepUpdateWater<UpdateWaterStep,...>,
// FlowMapWriteValues<CreateVelocityField,...> and MapNormalizeValues<NormalizeMap,...>
// (Geologic/FlowMap/FlowMapComponents.cs:16-202, Geologic/FlowMap/FlowMapJob.cs:16-228,
// Filter/NormalizeJob.cs:57-92) as bound in Geologic/Stage/FlowMapStage.cs:25-29.
//
// Two tiers:
//  * delegate-level kernels (flow_step / water_step / velocity / normalize): one launch per reference
//    job, same inputs and outputs, for drop-in at delegate level;
//  * flow_fused_kernel: up to five whole iterations (outflow + water update) per launch for the stage
//    body, the tile's state held in registers (see the kernel's comment).  Between launches the state
//    ping-pongs between two plane sets, the reference's READ/WRITE pairs (FlowMapStage.cs:52-62),
//    minus the serial flush copies; the default 5-iteration stage is a single launch.
//
// Arithmetic follows the C# order exactly (W,E,S,N; csum = ((x+y)+z)+w; true divisions), no FMA
// contraction, so results are bit-identical to the CPU restatement.
#include <cstdlib>

#include "nz_internal.hpp"
#include "nz_flow_common.hpp"

namespace {

constexpr int CT = 256;

// ---- delegate-level kernels ------------------------------------------------------------------

__global__ __launch_bounds__(CT) void fill_kernel(float *__restrict__ data, size_t n, float value) {
    size_t i = ((size_t)blockIdx.x * CT + threadIdx.x) * 4;
    if (i + 4 <= n && ((reinterpret_cast<uintptr_t>(data) & 15) == 0)) {
        *reinterpret_cast<float4 *>(data + i) = make_float4(value, value, value, value);
    } else {
        for (size_t k = i; k < n && k < i + 4; k++) data[k] = value;
    }
}

__global__ __launch_bounds__(CT) void copy_kernel(float *__restrict__ dst, const float *__restrict__ src, size_t n) {
    size_t i = ((size_t)blockIdx.x * CT + threadIdx.x) * 4;
    bool al = ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0;
    if (i + 4 <= n && al) {
        *reinterpret_cast<float4 *>(dst + i) = *reinterpret_cast<const float4 *>(src + i);
    } else {
        for (size_t k = i; k < n && k < i + 4; k++) dst[k] = src[k];
    }
}

// the flux planes are read at the thread's own cell only, so oX may alias fX (in-place update)
__global__ __launch_bounds__(CT) void flow_step_kernel(const float *__restrict__ h, const float *__restrict__ w,
                                                      const float *fN, const float *fS, const float *fE,
                                                      const float *fW, float *oN, float *oS, float *oE, float *oW,
                                                      nz_geom g) {
    int x = blockIdx.x * CT + threadIdx.x;
    int z = g.or0 + blockIdx.y;
    if (x >= g.cols || z >= g.or1) return;
    size_t c = (size_t)z * g.pitch + x;
    size_t iW = (size_t)z * g.pitch + clampi(x - 1, 0, g.cols - 1);
    size_t iE = (size_t)z * g.pitch + clampi(x + 1, 0, g.cols - 1);
    size_t iS = (size_t)clampi(z - 1, g.zc0, g.zc1) * g.pitch + x;
    size_t iN = (size_t)clampi(z + 1, g.zc0, g.zc1) * g.pitch + x;
    float water_0 = w[c], height_0 = h[c];
    float totalHt = water_0 + height_0;
    flux4 old = {fW[c], fE[c], fS[c], fN[c]};
    flux4 f = compute_flow(totalHt, water_0, w[iW] + h[iW], w[iE] + h[iE], w[iS] + h[iS], w[iN] + h[iN], old);
    oW[c] = f.w; oE[c] = f.e; oS[c] = f.s; oN[c] = f.n;
}

// water is read at the thread's own cell only, so w_out may alias w
__global__ __launch_bounds__(CT) void water_step_kernel(const float *w, float *w_out,
                                                       const float *__restrict__ fN, const float *__restrict__ fS,
                                                       const float *__restrict__ fE, const float *__restrict__ fW,
                                                       nz_geom g) {
    int x = blockIdx.x * CT + threadIdx.x;
    int z = g.or0 + blockIdx.y;
    if (x >= g.cols || z >= g.or1) return;
    size_t c = (size_t)z * g.pitch + x;
    flux4 own = {fW[c], fE[c], fS[c], fN[c]};
    float inE = fE[(size_t)z * g.pitch + clampi(x - 1, 0, g.cols - 1)];
    float inW = fW[(size_t)z * g.pitch + clampi(x + 1, 0, g.cols - 1)];
    float inN = fN[(size_t)clampi(z - 1, g.zc0, g.zc1) * g.pitch + x];
    float inS = fS[(size_t)clampi(z + 1, g.zc0, g.zc1) * g.pitch + x];
    w_out[c] = update_water(w[c], own, inE, inW, inN, inS);
}

// CreateVelocityField.CalculateCell (FlowMapComponents.cs:120-139), optionally followed by
// NormalizeMap.CalculateCell (:157-165) in the same thread
__global__ __launch_bounds__(CT) void velocity_kernel(float *__restrict__ dst, const float *__restrict__ fN,
                                                     const float *__restrict__ fS, const float *__restrict__ fE,
                                                     const float *__restrict__ fW, nz_geom g, int normalize,
                                                     float nmin, float nrange) {
    int x = blockIdx.x * CT + threadIdx.x;
    int z = g.or0 + blockIdx.y;
    if (x >= g.cols || z >= g.or1) return;
    size_t row = (size_t)z * g.pitch;
    int xm = clampi(x - 1, 0, g.cols - 1), xp = clampi(x + 1, 0, g.cols - 1);
    size_t rm = (size_t)clampi(z - 1, g.zc0, g.zc1) * g.pitch, rp = (size_t)clampi(z + 1, g.zc0, g.zc1) * g.pitch;
    float dl = fE[row + xm] - fW[row + x];
    float dr = fE[row + x] - fW[row + xp];
    float dt = fS[rp + x] - fN[row + x];
    float db = fS[row + x] - fN[rm + x];
    float vx = (dl + dr) * 0.5f;
    float vy = (dt + db) * 0.5f;
    float v = sqrtf(vx * vx + vy * vy);
    if (normalize) {
        if (nrange < 1e-12f) v = 0.0f;
        v = (v - nmin) / nrange;
    }
    dst[row + x] = v;
}

__global__ __launch_bounds__(CT) void normalize_kernel(const float *src, float *dst, size_t n, float nmin,
                                                      float nrange) {
    size_t i = (size_t)blockIdx.x * CT + threadIdx.x;
    if (i >= n) return;
    float v = src[i];
    if (nrange < 1e-12f) v = 0.0f;
    dst[i] = (v - nmin) / nrange;
}

// ---- multi-iteration fused kernel -----------------------------------------------------------------
// n iterations (outflow + water update) on an LDS/register-resident tile, optionally starting from
// the implied initial state (FIRST: water 1e-4, flux 0) and optionally ending in the velocity +
// normalise epilogue (LAST; the last water update is dead and skipped).  HBM traffic per launch: one
// read of height (+ five state planes unless FIRST) and one write of the five state planes (or of the
// single output plane when LAST), for n iterations.
//
// A workgroup of 512 threads holds a 48 x 128 tile (halo 2n included); every thread owns three groups
// of 4 consecutive cells whose height, water and four flux values stay in registers for the whole
// launch.  West/east neighbours come from the adjacent lane with a wave-shift DPP move (no LDS);
// south/north neighbours (rows z-1 / z+1) go through three LDS planes (total height, fN, fS) with
// 16-byte accesses; pitch 132 floats keeps those conflict free.  Cells outside the grid never feed a
// cell inside it: border cells substitute their own value for a clamped neighbour, exactly what
// clamp-to-edge reads return (ReadTileData.GetData, Pipeline/Tiles/TileData.cs:106-116).
#ifndef NZ_FT_NT
#define NZ_FT_NT 512
#endif
#ifndef NZ_FT_OCC
#define NZ_FT_OCC 4
#endif
constexpr int FT_TH = 48, FT_TW = 128, FT_NT = NZ_FT_NT, FT_LP = FT_TW + 4;
constexpr int FT_TH_TINY = 32, FT_TH_MID = 64;  // the tiles of grids that fit one round of the CUs: 1024 threads (nz_launch_flow_fused)
// FT_MAX_N = 5 (nz_flow_common.hpp): 2n halo rows, n = 5 leaves a 28 x 104 interior

// (bound_ctrl: the lane without a source reads 0, and no v_mov 0 has to initialise the destination first)
__device__ __forceinline__ float wave_from_prev_lane(float v) {  // lane i <- lane i-1
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_from_next_lane(float v) {  // lane i <- lane i+1
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

struct f4 {
    float v[4];
};

__device__ __forceinline__ f4 lds_load4(const float *p) {
    float4 t = *reinterpret_cast<const float4 *>(p);
    return f4{{t.x, t.y, t.z, t.w}};
}
__device__ __forceinline__ void lds_store4(float *p, const float v[4]) {
    *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
}

__device__ __forceinline__ void load_group(const float *__restrict__ p, const nz_geom &g, int gx0, int gz, bool fast,
                                           float out[4]) {
    if (fast) {
        float4 t = *reinterpret_cast<const float4 *>(p + (size_t)gz * g.pitch + gx0);
        out[0] = t.x; out[1] = t.y; out[2] = t.z; out[3] = t.w;
        // after the access: keeps the two forms from being sunk into one tail of 4-byte accesses with selected addresses
        asm volatile("; 16-byte load" ::: "memory");
    } else {
        size_t row = (size_t)clampi(gz, g.zc0, g.zc1) * g.pitch;
#pragma unroll
        for (int e = 0; e < 4; e++) out[e] = p[row + clampi(gx0 + e, 0, g.cols - 1)];
    }
}

__device__ __forceinline__ void store_group(float *__restrict__ p, const nz_geom &g, int gx0, int gz, bool fast,
                                            const float v[4]) {
    if (fast) {
        *reinterpret_cast<float4 *>(p + (size_t)gz * g.pitch + gx0) = make_float4(v[0], v[1], v[2], v[3]);
        asm volatile("; 16-byte store" ::: "memory");
    } else {
#pragma unroll
        for (int e = 0; e < 4; e++)
            if (gx0 + e < g.cols) p[(size_t)gz * g.pitch + gx0 + e] = v[e];
    }
}

template <bool FIRST, bool LAST, bool EDGE, bool FAST, int FTH, int FNT>
__device__ __forceinline__ void flow_fused_body(float *s_tot, float *s_fn, float *s_fs, int tile, const float *__restrict__ h, const float *__restrict__ w_in,
                                                          const float *__restrict__ fN_in, const float *__restrict__ fS_in,
                                                          const float *__restrict__ fE_in, const float *__restrict__ fW_in,
                                                          float *__restrict__ w_out, float *__restrict__ fN_out,
                                                          float *__restrict__ fS_out, float *__restrict__ fE_out,
                                                          float *__restrict__ fW_out, float *__restrict__ dst,
                                                          float *__restrict__ h_out, nz_geom g, int n, float nmin,
                                                          float nrange, int aligned) {
    constexpr int FG = FTH * FT_TW / 4 / FNT;  // groups of four cells per thread
    static_assert(FG * FNT * 4 == FTH * FT_TW, "the threads' groups tile the register tile");
    const int tid = threadIdx.x;
    const int H = 2 * n, HX = (H + 3) & ~3;
    const int OW = FT_TW - 2 * HX, OH = FTH - 2 * H;
    const int tiles_x = (g.cols + OW - 1) / OW;
    const int by = tile / tiles_x, bx = tile - by * tiles_x;
    const int ox0 = bx * OW, oz0 = g.or0 + by * OH;
    const int lx0 = ox0 - HX, lz0 = oz0 - H;
    // tile strictly inside the grid: no cell is a border cell, every load is in range
    constexpr bool inner = !EDGE;
    const bool fast = inner && aligned;

    float hh[FG][4], ww[FG][4], fW[FG][4], fE[FG][4], fS[FG][4], fN[FG][4];
    int grow[FG], gcol[FG];
#pragma unroll
    for (int j = 0; j < FG; j++) {
        grow[j] = (tid >> 5) + j * (FNT / 32);
        gcol[j] = (tid & 31) * 4;
        int gx0 = lx0 + gcol[j], gz = lz0 + grow[j];
        load_group(h, g, gx0, gz, fast, hh[j]);
        if (FIRST) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                ww[j][e] = 0.0001f;  // fillStage, FlowMapStage.cs:129
                fW[j][e] = 0.0f; fE[j][e] = 0.0f; fS[j][e] = 0.0f; fN[j][e] = 0.0f;
            }
        } else {
            load_group(w_in, g, gx0, gz, fast, ww[j]);
            load_group(fW_in, g, gx0, gz, fast, fW[j]);
            load_group(fE_in, g, gx0, gz, fast, fE[j]);
            load_group(fS_in, g, gx0, gz, fast, fS[j]);
            load_group(fN_in, g, gx0, gz, fast, fN[j]);
        }
    }

    for (int it = 0; it < n; it++) {
        // Rows within 2*it of the tile's top / bottom edge already hold values no interior cell depends on (each
        // iteration's clamped-neighbour error creeps 2 rows inwards), so interior tiles skip them: a wave owns the
        // row pairs (2w, 2w+1) + 16j and the bounds are even, which keeps the skip wave-uniform.
        const int dead = EDGE ? 0 : 2 * it;
        // ---- 1. publish water + height
#pragma unroll
        for (int j = 0; j < FG; j++) {
            if (grow[j] < dead || grow[j] >= FTH - dead) continue;
            float tot[4];
#pragma unroll
            for (int e = 0; e < 4; e++) tot[e] = ww[j][e] + hh[j][e];
            lds_store4(&s_tot[grow[j] * FT_LP + gcol[j]], tot);
        }
        __syncthreads();
        // ---- 2. outflow (ComputeFlowStep), publish fN / fS
#pragma unroll
        for (int j = 0; j < FG; j++) {
            int r = grow[j];
            if (r < dead || r >= FTH - dead) continue;
            f4 tS = lds_load4(&s_tot[(r > 0 ? r - 1 : r) * FT_LP + gcol[j]]);
            f4 tN = lds_load4(&s_tot[(r < FTH - 1 ? r + 1 : r) * FT_LP + gcol[j]]);
            float tot[4];  // recomputed rather than kept live across the barrier (same value)
#pragma unroll
            for (int e = 0; e < 4; e++) tot[e] = ww[j][e] + hh[j][e];
            float left = wave_from_prev_lane(tot[3]), right = wave_from_next_lane(tot[0]);
            int gx0 = lx0 + gcol[j], gz = lz0 + r;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float self = tot[e];
                float tW = e > 0 ? tot[e - 1] : left;
                float tE = e < 3 ? tot[e + 1] : right;
                float ts = tS.v[e], tn = tN.v[e];
                if (!inner) {
                    if (gx0 + e <= 0) tW = self;
                    if (gx0 + e >= g.cols - 1) tE = self;
                    if (gz <= g.zc0) ts = self;
                    if (gz >= g.zc1) tn = self;
                }
                flux4 f = compute_flow_m<FAST>(self, ww[j][e], tW, tE, ts, tn, flux4{fW[j][e], fE[j][e], fS[j][e], fN[j][e]}, false);
                fW[j][e] = f.w; fE[j][e] = f.e; fS[j][e] = f.s; fN[j][e] = f.n;
            }
            lds_store4(&s_fn[r * FT_LP + gcol[j]], fN[j]);
            lds_store4(&s_fs[r * FT_LP + gcol[j]], fS[j]);
        }
        __syncthreads();
        if (LAST && it == n - 1) break;
        // ---- 3. water update (UpdateWaterStep)
#pragma unroll
        for (int j = 0; j < FG; j++) {
            int r = grow[j];
            if (!EDGE && (r < dead + 2 || r >= FTH - dead - 2)) continue;  // the water of the next iteration's dead rows
            f4 nS = lds_load4(&s_fn[(r > 0 ? r - 1 : r) * FT_LP + gcol[j]]);            // fN of row z-1
            f4 sN = lds_load4(&s_fs[(r < FTH - 1 ? r + 1 : r) * FT_LP + gcol[j]]);    // fS of row z+1
            float eW = wave_from_prev_lane(fE[j][3]), wE = wave_from_next_lane(fW[j][0]);
            int gx0 = lx0 + gcol[j], gz = lz0 + r;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float inE = e > 0 ? fE[j][e - 1] : eW;
                float inW = e < 3 ? fW[j][e + 1] : wE;
                float inN = nS.v[e], inS = sN.v[e];
                if (!inner) {
                    if (gx0 + e <= 0) inE = fE[j][e];
                    if (gx0 + e >= g.cols - 1) inW = fW[j][e];
                    if (gz <= g.zc0) inN = fN[j][e];
                    if (gz >= g.zc1) inS = fS[j][e];
                }
                ww[j][e] = update_water_m<FAST>(ww[j][e], flux4{fW[j][e], fE[j][e], fS[j][e], fN[j][e]}, inE, inW, inN, inS);
            }
        }
    }

    // ---- epilogue: interior groups only
    const float inv_range = FAST && LAST ? 1.0f / nrange : 0.0f;
#pragma unroll
    for (int j = 0; j < FG; j++) {
        int r = grow[j];
        int gx0 = lx0 + gcol[j], gz = lz0 + r;
        bool interior = r >= H && r < H + OH && gcol[j] >= HX && gcol[j] < HX + OW && gz < g.or1 && gx0 < g.cols;
        float out[4];
        if (LAST) {
            // CreateVelocityField + NormalizeMap, FlowMapComponents.cs:120-139,157-165
            f4 nS = lds_load4(&s_fn[(r > 0 ? r - 1 : r) * FT_LP + gcol[j]]);
            f4 sN = lds_load4(&s_fs[(r < FTH - 1 ? r + 1 : r) * FT_LP + gcol[j]]);
            float eW = wave_from_prev_lane(fE[j][3]), wE = wave_from_next_lane(fW[j][0]);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float fE_w = e > 0 ? fE[j][e - 1] : eW;
                float fW_e = e < 3 ? fW[j][e + 1] : wE;
                float fN_s = nS.v[e], fS_n = sN.v[e];
                if (!inner) {
                    if (gx0 + e <= 0) fE_w = fE[j][e];
                    if (gx0 + e >= g.cols - 1) fW_e = fW[j][e];
                    if (gz <= g.zc0) fN_s = fN[j][e];
                    if (gz >= g.zc1) fS_n = fS[j][e];
                }
                float dl = fE_w - fW[j][e];
                float dr = fE[j][e] - fW_e;
                float dt = fS_n - fN[j][e];
                float db = fS[j][e] - fN_s;
                out[e] = velocity_norm_m<FAST>(dl, dr, dt, db, nmin, nrange, inv_range);
            }
        }
        if (interior) {
            bool vec = aligned && gx0 + 4 <= g.cols;
            if (h_out) store_group(h_out, g, gx0, gz, vec, hh[j]);  // private copy of the height plane
            if (LAST) {
                store_group(dst, g, gx0, gz, vec, out);
            } else {
                store_group(w_out, g, gx0, gz, vec, ww[j]);
                store_group(fW_out, g, gx0, gz, vec, fW[j]);
                store_group(fE_out, g, gx0, gz, vec, fE[j]);
                store_group(fS_out, g, gx0, gz, vec, fS[j]);
                store_group(fN_out, g, gx0, gz, vec, fN[j]);
            }
        }
    }
}

// The interior instantiation (tile strictly inside the grid) carries no border selects; the choice is
// uniform per workgroup.
template <bool FIRST, bool LAST, int OCC, bool FAST, int FTH, int FNT>
__global__ __launch_bounds__(FNT, OCC) void flow_fused_kernel(const float *__restrict__ h, const float *__restrict__ w_in,
                                                               const float *__restrict__ fN_in, const float *__restrict__ fS_in,
                                                               const float *__restrict__ fE_in, const float *__restrict__ fW_in,
                                                               float *__restrict__ w_out, float *__restrict__ fN_out,
                                                               float *__restrict__ fS_out, float *__restrict__ fE_out,
                                                               float *__restrict__ fW_out, float *__restrict__ dst,
                                                               float *__restrict__ h_out, nz_geom g, int n, float nmin,
                                                               float nrange, int aligned) {
    __shared__ __attribute__((aligned(16))) float s_tot[FTH * FT_LP];
    __shared__ __attribute__((aligned(16))) float s_fn[FTH * FT_LP];
    __shared__ __attribute__((aligned(16))) float s_fs[FTH * FT_LP];
    const int H = 2 * n, HX = (H + 3) & ~3;
    const int OW = FT_TW - 2 * HX, OH = FTH - 2 * H;
    const int tiles_x = (g.cols + OW - 1) / OW;
    // XCD-aware tile order (workgroups go round-robin to the 8 XCDs): XCD x takes the x-th contiguous eighth of
    // the tiles, so tiles that share halo rows / columns meet in one L2
    int tile = blockIdx.x;
    {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, x = tile & 7, k = tile >> 3;
        tile = x * q + (x < r ? x : r) + k;
    }
    const int by = tile / tiles_x, bx = tile - by * tiles_x;
    const int lx0 = bx * OW - HX, lz0 = g.or0 + by * OH - H;
    const bool inner = lx0 > 0 && lx0 + FT_TW < g.cols && lz0 > g.zc0 && lz0 + FTH - 1 < g.zc1;
    // batched launch: one independent grid per blockIdx.y, every plane shifted by the same stride
    const size_t off = blockIdx.y * g.bstride;
#define NZ_SH(p) ((p) ? (p) + off : (p))
    if (inner)
        flow_fused_body<FIRST, LAST, false, FAST, FTH, FNT>(s_tot, s_fn, s_fs, tile, h + off, NZ_SH(w_in), NZ_SH(fN_in), NZ_SH(fS_in),
                                            NZ_SH(fE_in), NZ_SH(fW_in), NZ_SH(w_out), NZ_SH(fN_out), NZ_SH(fS_out),
                                            NZ_SH(fE_out), NZ_SH(fW_out), NZ_SH(dst), NZ_SH(h_out), g, n, nmin, nrange,
                                            aligned);
    else
        flow_fused_body<FIRST, LAST, true, FAST, FTH, FNT>(s_tot, s_fn, s_fs, tile, h + off, NZ_SH(w_in), NZ_SH(fN_in), NZ_SH(fS_in),
                                           NZ_SH(fE_in), NZ_SH(fW_in), NZ_SH(w_out), NZ_SH(fN_out), NZ_SH(fS_out),
                                           NZ_SH(fE_out), NZ_SH(fW_out), NZ_SH(dst), NZ_SH(h_out), g, n, nmin, nrange,
                                           aligned);
#undef NZ_SH
}


}  // namespace

int nz_flow_fused_max() {
    static const int cap = getenv("NZ_FLOW_NMAX") ? atoi(getenv("NZ_FLOW_NMAX")) : 5;
    return cap < 1 ? 1 : (cap > FT_MAX_N ? FT_MAX_N : cap);
}

// n iterations; `first`: implied initial state, inputs unread; `last`: velocity+normalise into dst
int32_t nz_launch_flow_fused(hipStream_t s, const float *h, const float *const in[5], float *const out[5], float *dst,
                             float *h_out, const nz_geom &g, int n, int first, int last, float nmin, float nrange) {
    if (n < 1 || n > FT_MAX_N) {
        nz_set_error("flow_fused: n=%d unsupported", n);
        return NZ_ERR_INVALID;
    }
    if (g.or1 <= g.or0) return NZ_OK;
    if (first && last && nz_flow_stream_wanted(g, n)) return nz_launch_flow_stream(s, h, dst, g, n, nmin, nrange);
    int H = 2 * n, HX = (H + 3) & ~3;
    int OW = FT_TW - 2 * HX, OH = FT_TH - 2 * H;
    long long blocks = (long long)((g.cols + OW - 1) / OW) * ((g.or1 - g.or0 + OH - 1) / OH);
    // A TINY grid (a tile of the reference's own sizes, 256^2 ... 512^2): at most a workgroup per CU whatever the tile, and what
    // the launch waits for is the latency of one workgroup's dependent iterations.  32-row tiles of 1024 threads -- four cells
    // per thread, four waves per SIMD -- run an iteration in ~0.6 of the time of the 48-row tile's twelve cells per thread (more
    // workgroups, each with a third less work per SIMD): while they still fit one round of the CUs, they are used.
    static const int tiny = getenv("NZ_FLOW_TINY") ? atoi(getenv("NZ_FLOW_TINY")) : 1;  // 0: never; 2 / 3: the 32- / 64-row tile at every size (test matrix)
    const int OH_t = FT_TH_TINY - 2 * H;
    const long long blocks_t = OH_t > 0 ? (long long)((g.cols + OW - 1) / OW) * ((g.or1 - g.or0 + OH_t - 1) / OH_t) : 0;
    const bool use_tiny = OH_t >= 4 && (tiny == 2 || (tiny == 1 && blocks_t * g.count <= nz_cu_count()));
    // ... and the next size up (768^2 ... 1024^2): 64-row tiles of 1024 threads -- eight cells per thread, four waves per SIMD,
    // 0.69 of the tile is interior at n = 5 instead of 0.58 -- while THEY fit one round (the 48-row tile's 370 workgroups at
    // 1024^2 are one and a half per CU: the launch lasts as long as the CUs that hold two).  Flow x5: 512^2 20.3 -> 15.3 us
    // (32-row tiles), 1024^2 30.6 -> 26.1 us (64-row tiles); 768^2 keeps the 48-row tile (224 workgroups: 21.4 against 24.5 us)
    const int OH_m = FT_TH_MID - 2 * H;
    const long long blocks_m = (long long)((g.cols + OW - 1) / OW) * ((g.or1 - g.or0 + OH_m - 1) / OH_m);
    const bool use_mid = !use_tiny && (tiny == 3 || (tiny == 1 && blocks * g.count > nz_cu_count() && blocks_m * g.count <= nz_cu_count()));
    if (use_tiny) blocks = blocks_t;
    if (use_mid) blocks = blocks_m;
    uintptr_t bits = reinterpret_cast<uintptr_t>(h) | (uintptr_t)(g.pitch * 4);
    if (!first)
        for (int i = 0; i < 5; i++) bits |= reinterpret_cast<uintptr_t>(in[i]);
    if (h_out) bits |= reinterpret_cast<uintptr_t>(h_out);
    if (last)
        bits |= reinterpret_cast<uintptr_t>(dst);
    else
        for (int i = 0; i < 5; i++) bits |= reinterpret_cast<uintptr_t>(out[i]);
    int aligned = (bits & 15) == 0;
    const float *w_in = first ? nullptr : in[0], *fN_in = first ? nullptr : in[1], *fS_in = first ? nullptr : in[2];
    const float *fE_in = first ? nullptr : in[3], *fW_in = first ? nullptr : in[4];
    float *w_out = last ? nullptr : out[0], *fN_out = last ? nullptr : out[1], *fS_out = last ? nullptr : out[2];
    float *fE_out = last ? nullptr : out[3], *fW_out = last ? nullptr : out[4];
    // (the kernel is register-allocated for four waves per SIMD = two 512-thread workgroups per CU: one workgroup per CU with
    // the whole register file lost twice, 0.342 against 0.241 ms)
    const bool fast = nz_tls_float_mode >= NZ_FLOAT_RELAXED;
#define NZ_FFA h, w_in, fN_in, fS_in, fE_in, fW_in, w_out, fN_out, fS_out, fE_out, fW_out, dst, h_out, g, n, nmin, nrange, aligned
#define NZ_FFL(F, L, M)                                                                                                          \
    do {                                                                                                                         \
        if (use_tiny)                                                                                                            \
            NZ_LAUNCH((flow_fused_kernel<F, L, 4, M, FT_TH_TINY, 1024>), dim3((unsigned)blocks, g.count), dim3(1024), 0, s, NZ_FFA); \
        else if (use_mid)                                                                                                        \
            NZ_LAUNCH((flow_fused_kernel<F, L, 4, M, FT_TH_MID, 1024>), dim3((unsigned)blocks, g.count), dim3(1024), 0, s, NZ_FFA); \
        else                                                                                                                     \
            NZ_LAUNCH((flow_fused_kernel<F, L, NZ_FT_OCC, M, FT_TH, FT_NT>), dim3((unsigned)blocks, g.count), dim3(FT_NT), 0, s, NZ_FFA); \
    } while (0)
#define NZ_FF(F, L)                    \
    do {                               \
        if (fast) NZ_FFL(F, L, true);  \
        else NZ_FFL(F, L, false);      \
    } while (0)
    if (first && last) NZ_FF(true, true);
    else if (first) NZ_FF(true, false);
    else if (last) NZ_FF(false, true);
    else NZ_FF(false, false);
#undef NZ_FF
#undef NZ_FFL
#undef NZ_FFA
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

int32_t nz_launch_fill(hipStream_t s, float *data, size_t n, float value) {
    if (n == 0) return NZ_OK;
    size_t blocks = (n + (size_t)CT * 4 - 1) / ((size_t)CT * 4);
    NZ_LAUNCH(fill_kernel, dim3((unsigned)blocks), dim3(CT), 0, s, data, n, value);
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

int32_t nz_launch_copy(hipStream_t s, float *dst, const float *src, size_t n) {
    if (n == 0) return NZ_OK;
    size_t blocks = (n + (size_t)CT * 4 - 1) / ((size_t)CT * 4);
    NZ_LAUNCH(copy_kernel, dim3((unsigned)blocks), dim3(CT), 0, s, dst, src, n);
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

int32_t nz_launch_flow_step(hipStream_t s, const float *h, const float *w, const float *fN, const float *fS,
                            const float *fE, const float *fW, float *oN, float *oS, float *oE, float *oW,
                            const nz_geom &g) {
    if (g.or1 <= g.or0) return NZ_OK;
    dim3 grid((g.cols + CT - 1) / CT, g.or1 - g.or0);
    hipLaunchKernelGGL(flow_step_kernel, grid, dim3(CT), 0, s, h, w, fN, fS, fE, fW, oN, oS, oE, oW, g);
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

int32_t nz_launch_water_step(hipStream_t s, const float *w, float *w_out, const float *fN, const float *fS,
                             const float *fE, const float *fW, const nz_geom &g) {
    if (g.or1 <= g.or0) return NZ_OK;
    dim3 grid((g.cols + CT - 1) / CT, g.or1 - g.or0);
    hipLaunchKernelGGL(water_step_kernel, grid, dim3(CT), 0, s, w, w_out, fN, fS, fE, fW, g);
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

int32_t nz_launch_velocity(hipStream_t s, float *dst, const float *fN, const float *fS, const float *fE,
                           const float *fW, const nz_geom &g, int normalize, float nmin, float nrange) {
    if (g.or1 <= g.or0) return NZ_OK;
    dim3 grid((g.cols + CT - 1) / CT, g.or1 - g.or0);
    hipLaunchKernelGGL(velocity_kernel, grid, dim3(CT), 0, s, dst, fN, fS, fE, fW, g, normalize, nmin, nrange);
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

int32_t nz_launch_normalize(hipStream_t s, const float *src, float *dst, size_t n, float nmin, float nrange) {
    if (n == 0) return NZ_OK;
    size_t blocks = (n + CT - 1) / CT;
    NZ_LAUNCH(normalize_kernel, dim3((unsigned)blocks), dim3(CT), 0, s, src, dst, n, nmin, nrange);
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}
