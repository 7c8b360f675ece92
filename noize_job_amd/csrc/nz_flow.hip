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

namespace {

constexpr int CT = 256;
constexpr float TIMESTEP = 0.2f;  // FlowMapComponents.cs:19,79

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

struct flux4 {
    float w, e, s, n;
};

// ComputeFlowStep.CalculateCell, FlowMapComponents.cs:20-65
__device__ __forceinline__ flux4 compute_flow(float totalHt, float water_0, float tW, float tE, float tS, float tN,
                                              flux4 old) {
    float dW = totalHt - tW, dE = totalHt - tE, dS = totalHt - tS, dN = totalHt - tN;
    flux4 f;
    f.w = fmaxf(0.0f, old.w + dW);
    f.e = fmaxf(0.0f, old.e + dE);
    f.s = fmaxf(0.0f, old.s + dS);
    f.n = fmaxf(0.0f, old.n + dN);
    float sum_ = (f.w + f.e) + (f.s + f.n);  // math.csum(float4) = (x.x + x.y) + (x.z + x.w)
    if (sum_ > 0.0f) {
        float K = water_0 / (sum_ * TIMESTEP);
        K = fmaxf(0.0f, fminf(1.0f, K));
        f.w *= K; f.e *= K; f.s *= K; f.n *= K;
    } else {
        f.w = 0.0f; f.e = 0.0f; f.s = 0.0f; f.n = 0.0f;
    }
    return f;
}

// The same without the branch, bit for bit.  Every f is fmaxf(0, .), so f is in [0, +inf] (never NaN: fmaxf returns its
// other operand) and sum_ is in [0, +inf]: `sum_ > 0` fails only when all four are +0.  Then water_0 / 0 is +-inf or NaN,
// which the clamp turns into a K of 0 or 1 (fminf(1, NaN) = 1), and +0 * K = +0 -- the zeros the else branch stores.
__device__ __forceinline__ flux4 compute_flow_nb(float totalHt, float water_0, float tW, float tE, float tS, float tN,
                                                 flux4 old) {
    float dW = totalHt - tW, dE = totalHt - tE, dS = totalHt - tS, dN = totalHt - tN;
    flux4 f;
    f.w = fmaxf(0.0f, old.w + dW);
    f.e = fmaxf(0.0f, old.e + dE);
    f.s = fmaxf(0.0f, old.s + dS);
    f.n = fmaxf(0.0f, old.n + dN);
    float sum_ = (f.w + f.e) + (f.s + f.n);
    float K = water_0 / (sum_ * TIMESTEP);
    K = fmaxf(0.0f, fminf(1.0f, K));
    f.w *= K; f.e *= K; f.s *= K; f.n *= K;
    return f;
}

// UpdateWaterStep.CalculateCell, FlowMapComponents.cs:81-104
__device__ __forceinline__ float update_water(float water, flux4 own, float fE_west, float fW_east, float fN_south,
                                              float fS_north) {
    float flowOUT = own.w + own.e + own.s + own.n;
    float flowIN = 0.0f;
    flowIN += fE_west;
    flowIN += fW_east;
    flowIN += fN_south;
    flowIN += fS_north;
    float ht = water + ((flowIN - flowOUT) * TIMESTEP);
    return fmaxf(0.0f, ht);
}

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
constexpr int FT_G = FT_TH * FT_TW / 4 / FT_NT;  // groups per thread = 3
constexpr int FT_MAX_N = 5;  // 2n halo rows: n = 5 leaves a 28 x 104 interior

__device__ __forceinline__ float wave_from_prev_lane(float v) {  // lane i <- lane i-1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_from_next_lane(float v) {  // lane i <- lane i+1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

// the same with 0 for the lane that has no neighbour and no zeroed destination to pay for (streaming kernel)
__device__ __forceinline__ float wave_prev0(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_next0(float v) {
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

template <bool FIRST, bool LAST, bool EDGE>
__device__ __forceinline__ void flow_fused_body(float *s_tot, float *s_fn, float *s_fs, int tile, const float *__restrict__ h, const float *__restrict__ w_in,
                                                          const float *__restrict__ fN_in, const float *__restrict__ fS_in,
                                                          const float *__restrict__ fE_in, const float *__restrict__ fW_in,
                                                          float *__restrict__ w_out, float *__restrict__ fN_out,
                                                          float *__restrict__ fS_out, float *__restrict__ fE_out,
                                                          float *__restrict__ fW_out, float *__restrict__ dst,
                                                          float *__restrict__ h_out, nz_geom g, int n, float nmin,
                                                          float nrange, int aligned) {
    const int tid = threadIdx.x;
    const int H = 2 * n, HX = (H + 3) & ~3;
    const int OW = FT_TW - 2 * HX, OH = FT_TH - 2 * H;
    const int tiles_x = (g.cols + OW - 1) / OW;
    const int by = tile / tiles_x, bx = tile - by * tiles_x;
    const int ox0 = bx * OW, oz0 = g.or0 + by * OH;
    const int lx0 = ox0 - HX, lz0 = oz0 - H;
    // tile strictly inside the grid: no cell is a border cell, every load is in range
    constexpr bool inner = !EDGE;
    const bool fast = inner && aligned;

    float hh[FT_G][4], ww[FT_G][4], fW[FT_G][4], fE[FT_G][4], fS[FT_G][4], fN[FT_G][4];
    int grow[FT_G], gcol[FT_G];
#pragma unroll
    for (int j = 0; j < FT_G; j++) {
        grow[j] = (tid >> 5) + j * (FT_NT / 32);
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
        for (int j = 0; j < FT_G; j++) {
            if (grow[j] < dead || grow[j] >= FT_TH - dead) continue;
            float tot[4];
#pragma unroll
            for (int e = 0; e < 4; e++) tot[e] = ww[j][e] + hh[j][e];
            lds_store4(&s_tot[grow[j] * FT_LP + gcol[j]], tot);
        }
        __syncthreads();
        // ---- 2. outflow (ComputeFlowStep), publish fN / fS
#pragma unroll
        for (int j = 0; j < FT_G; j++) {
            int r = grow[j];
            if (r < dead || r >= FT_TH - dead) continue;
            f4 tS = lds_load4(&s_tot[(r > 0 ? r - 1 : r) * FT_LP + gcol[j]]);
            f4 tN = lds_load4(&s_tot[(r < FT_TH - 1 ? r + 1 : r) * FT_LP + gcol[j]]);
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
                flux4 f = compute_flow(self, ww[j][e], tW, tE, ts, tn, flux4{fW[j][e], fE[j][e], fS[j][e], fN[j][e]});
                fW[j][e] = f.w; fE[j][e] = f.e; fS[j][e] = f.s; fN[j][e] = f.n;
            }
            lds_store4(&s_fn[r * FT_LP + gcol[j]], fN[j]);
            lds_store4(&s_fs[r * FT_LP + gcol[j]], fS[j]);
        }
        __syncthreads();
        if (LAST && it == n - 1) break;
        // ---- 3. water update (UpdateWaterStep)
#pragma unroll
        for (int j = 0; j < FT_G; j++) {
            int r = grow[j];
            if (!EDGE && (r < dead + 2 || r >= FT_TH - dead - 2)) continue;  // the water of the next iteration's dead rows
            f4 nS = lds_load4(&s_fn[(r > 0 ? r - 1 : r) * FT_LP + gcol[j]]);            // fN of row z-1
            f4 sN = lds_load4(&s_fs[(r < FT_TH - 1 ? r + 1 : r) * FT_LP + gcol[j]]);    // fS of row z+1
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
                ww[j][e] = update_water(ww[j][e], flux4{fW[j][e], fE[j][e], fS[j][e], fN[j][e]}, inE, inW, inN, inS);
            }
        }
    }

    // ---- epilogue: interior groups only
#pragma unroll
    for (int j = 0; j < FT_G; j++) {
        int r = grow[j];
        int gx0 = lx0 + gcol[j], gz = lz0 + r;
        bool interior = r >= H && r < H + OH && gcol[j] >= HX && gcol[j] < HX + OW && gz < g.or1 && gx0 < g.cols;
        float out[4];
        if (LAST) {
            // CreateVelocityField + NormalizeMap, FlowMapComponents.cs:120-139,157-165
            f4 nS = lds_load4(&s_fn[(r > 0 ? r - 1 : r) * FT_LP + gcol[j]]);
            f4 sN = lds_load4(&s_fs[(r < FT_TH - 1 ? r + 1 : r) * FT_LP + gcol[j]]);
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
                float vx = (dl + dr) * 0.5f;
                float vy = (dt + db) * 0.5f;
                float v = sqrtf(vx * vx + vy * vy);
                if (nrange < 1e-12f) v = 0.0f;
                out[e] = (v - nmin) / nrange;
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
template <bool FIRST, bool LAST, int OCC>
__global__ __launch_bounds__(FT_NT, OCC) void flow_fused_kernel(const float *__restrict__ h, const float *__restrict__ w_in,
                                                               const float *__restrict__ fN_in, const float *__restrict__ fS_in,
                                                               const float *__restrict__ fE_in, const float *__restrict__ fW_in,
                                                               float *__restrict__ w_out, float *__restrict__ fN_out,
                                                               float *__restrict__ fS_out, float *__restrict__ fE_out,
                                                               float *__restrict__ fW_out, float *__restrict__ dst,
                                                               float *__restrict__ h_out, nz_geom g, int n, float nmin,
                                                               float nrange, int aligned) {
    __shared__ __attribute__((aligned(16))) float s_tot[FT_TH * FT_LP];
    __shared__ __attribute__((aligned(16))) float s_fn[FT_TH * FT_LP];
    __shared__ __attribute__((aligned(16))) float s_fs[FT_TH * FT_LP];
    const int H = 2 * n, HX = (H + 3) & ~3;
    const int OW = FT_TW - 2 * HX, OH = FT_TH - 2 * H;
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
    const bool inner = lx0 > 0 && lx0 + FT_TW < g.cols && lz0 > g.zc0 && lz0 + FT_TH - 1 < g.zc1;
    // batched launch: one independent grid per blockIdx.y, every plane shifted by the same stride
    const size_t off = blockIdx.y * g.bstride;
#define NZ_SH(p) ((p) ? (p) + off : (p))
    if (inner)
        flow_fused_body<FIRST, LAST, false>(s_tot, s_fn, s_fs, tile, h + off, NZ_SH(w_in), NZ_SH(fN_in), NZ_SH(fS_in),
                                            NZ_SH(fE_in), NZ_SH(fW_in), NZ_SH(w_out), NZ_SH(fN_out), NZ_SH(fS_out),
                                            NZ_SH(fE_out), NZ_SH(fW_out), NZ_SH(dst), NZ_SH(h_out), g, n, nmin, nrange,
                                            aligned);
    else
        flow_fused_body<FIRST, LAST, true>(s_tot, s_fn, s_fs, tile, h + off, NZ_SH(w_in), NZ_SH(fN_in), NZ_SH(fS_in),
                                           NZ_SH(fE_in), NZ_SH(fW_in), NZ_SH(w_out), NZ_SH(fN_out), NZ_SH(fS_out),
                                           NZ_SH(fE_out), NZ_SH(fW_out), NZ_SH(dst), NZ_SH(h_out), g, n, nmin, nrange,
                                           aligned);
#undef NZ_SH
}


// ---- row-streaming form of the whole stage (first && last) ------------------------------------------------------------
// The tile kernel above keeps six planes of a 48 x 128 tile on chip and steps all of it through the iterations in
// lock-step: the 2-cell halo per iteration leaves a 28 x 104 interior at n = 5, i.e. 1.8x the arithmetic of the cells it
// stores, and every iteration crosses three workgroup barriers.  Here ONE WAVE owns a 128-column strip (two columns per
// lane) and walks down its rows with the iterations pipelined behind each other, iteration i two rows behind iteration
// i - 1 (time skewing).  Per iteration a lane keeps only what the rows still in flight need: two rows of total height
// and water of the previous iteration and two rows of its own four outflows, 12 registers per column and iteration;
// the z-neighbours of a cell are those registers, the x-neighbours the adjacent lanes' (wave-shift DPP).  No LDS traffic
// except the wave's private ring of height rows (tot = water + height needs the height again four times, two rows
// later each time), no barrier, no flag.  Redundant work is the x halo (2n columns per side of 128) and the pipeline
// fill of a row segment (iteration i starts 2(n - i) + 1 rows above the segment): 108 / 128 x S / (S + 2n) of the
// executed cell-iterations are stored ones, 0.70 for the 51-row segments that give every SIMD of the chip three waves
// at 4096^2 (0.47 for the tile kernel).  Same per-cell functions, same operand order: bit-identical results.
// -DNZ_FLOW_PROBE: lane 0 of every wave stamps s_memrealtime (100 MHz) and s_memtime (shader clock) at its start, after
// the pipeline fill and at its end, plus HW_ID / XCC_ID, into a caller-supplied buffer (tools/probe_flow_stream.py).
// Never built by the Makefile.
#ifdef NZ_FLOW_PROBE
__device__ unsigned long long *nz_flow_probe_buf = nullptr;  // [wave][8]
#define NZ_FPROBE(slot, val)                                                                                        \
    do {                                                                                                            \
        if (threadIdx.x == 0 && nz_flow_probe_buf)                                                                  \
            nz_flow_probe_buf[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (slot)] = (val);                  \
    } while (0)
#else
#define NZ_FPROBE(slot, val)
#endif
#ifndef NZ_FS_PRIO
#define NZ_FS_PRIO 1
#endif
constexpr int FS_RING = 16;   // rows of height kept per wave (needs 2n - 1 <= 9)
constexpr int FS_TW = 128;    // columns per strip, halo included

template <int NST>
struct fs_state {
    // stage i (0-based) is about to compute the outflow of row r = t - 2i from the state of iteration i - 1:
    float Tm[NST][2], T0[NST][2];   // total height (water + height) of rows r - 1, r after iteration i - 1
    float Wm[NST][2], W0[NST][2];   // water of rows r - 1, r after iteration i - 1
    float FA[NST][2][4], FB[NST][2][4];  // {W, E, S, N} outflow of rows r - 2, r - 1 of iteration i
};

struct fs_bounds {
    int loF[FT_MAX_N], hiF[FT_MAX_N];  // rows whose outflow stage i computes
    int loW[FT_MAX_N], hiW[FT_MAX_N];  // rows whose water it updates (last stage: rows whose velocity it stores)
};

// One step: every stage advances one row.
//   NACT : only stages 0 .. NACT-1 compute (pipeline fill of a segment away from the grid's first row: stage i joins four
//          steps after stage i - 1); stage NACT's windows are kept moving so that it finds its rows when it joins.  A
//          stage that has just joined updates water / stores velocity two rows early: values nobody reads, stores
//          masked by the row test;
//   COND : stages outside their row range are skipped (fill and drain of the segments on the grid's first / last rows),
//          and a cell in the grid's first / last row takes its own value for the clamped z-neighbour;
//   XEDGE: the strip touches the grid's first / last column: the lane that holds it takes its own value for the clamped
//          x-neighbour (column 0 is the first of its lane's two columns; the last column the second, or the first when the row length is odd).
// Cells outside the grid are computed like any other and never read by a cell inside it.
template <int NST, int NACT, bool COND, bool XEDGE, bool VEC>
__device__ __forceinline__ void fs_step(fs_state<NST> &st, const int t, const float2 hp, const float2 hn, float2 *ring,
                                        const fs_bounds &b, const int gx, const bool lane_x0, const bool lane_x1, const bool lane_x1o,
                                        const nz_geom &g, const float nmin, const float nrange,
                                        float *__restrict__ dst, const bool store_lane) {
    float Tp[2] = {0.0001f + hp.x, 0.0001f + hp.y};  // fillStage (FlowMapStage.cs:129): water 1e-4 everywhere
    float Wp[2] = {0.0001f, 0.0001f};
    float FCprev[2][4] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
#pragma unroll
    for (int i = 0; i < NST; i++) {
        if (i > NACT) continue;
        if (i == NACT) {  // not computing yet: its windows follow what stage NACT - 1 hands on
#pragma unroll
            for (int e = 0; e < 2; e++) {
                st.Tm[i][e] = st.T0[i][e]; st.T0[i][e] = Tp[e];
                st.Wm[i][e] = st.W0[i][e]; st.W0[i][e] = Wp[e];
            }
            continue;
        }
        const int r = t - 2 * i;
        float FC[2][4];
        // height of the row whose water this stage updates (read from the ring ahead of the outflow arithmetic)
        float2 hh = make_float2(0.0f, 0.0f);
        if (i < NST - 1) hh = ring[((r - 1) & (FS_RING - 1)) * 64];
        // The prefetched row h(t + 2) goes into the ring here, before the last stage's stores: waiting for that load behind
        // a (conditional) store would mean waiting for the store as well -- vmcnt counts both, in order.
        if (i == NACT - 1) ring[((t + 2) & (FS_RING - 1)) * 64] = hn;
        // ---- outflow of row r (ComputeFlowStep)
        const bool fa = !COND || (r >= b.loF[i] && r < b.hiF[i]);
        if (fa) {
            float left = wave_prev0(st.T0[i][1]), right = wave_next0(st.T0[i][0]);
            if (XEDGE) {
                left = lane_x0 ? st.T0[i][0] : left;
                right = lane_x1 ? st.T0[i][1] : right;
            }
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const float self = st.T0[i][e];
                const float tW = e == 0 ? left : st.T0[i][0];
                float tE = e == 1 ? right : st.T0[i][1];
                if (XEDGE && e == 0 && lane_x1o) tE = self;
                float tS = st.Tm[i][e], tN = Tp[e];
                if (COND) {
                    if (r <= g.zc0) tS = self;
                    if (r >= g.zc1) tN = self;
                }
                flux4 old;
                if (i == 0) old = flux4{0.0f, 0.0f, 0.0f, 0.0f};
                else old = flux4{st.FA[i > 0 ? i - 1 : 0][e][0], st.FA[i > 0 ? i - 1 : 0][e][1],
                                 st.FA[i > 0 ? i - 1 : 0][e][2], st.FA[i > 0 ? i - 1 : 0][e][3]};
                const flux4 f = compute_flow_nb(self, st.W0[i][e], tW, tE, tS, tN, old);
                FC[e][0] = f.w; FC[e][1] = f.e; FC[e][2] = f.s; FC[e][3] = f.n;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 2; e++)
#pragma unroll
                for (int k = 0; k < 4; k++) FC[e][k] = 0.0f;
        }
        // the previous stage's row r (its FA) has now been consumed: its window moves on
        if (i > 0) {
#pragma unroll
            for (int e = 0; e < 2; e++)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    st.FA[i > 0 ? i - 1 : 0][e][k] = st.FB[i > 0 ? i - 1 : 0][e][k];
                    st.FB[i > 0 ? i - 1 : 0][e][k] = FCprev[e][k];
                }
        }
        // ---- row r - 1: water update (UpdateWaterStep), or for the last stage velocity + normalise
        const int rw = r - 1;
        const bool wa = !COND || (rw >= b.loW[i] && rw < b.hiW[i]);
        float Wn[2] = {0.0f, 0.0f}, Tn[2] = {0.0f, 0.0f};
        if (wa) {
            // own row rw = FB, row rw - 1 = FA (its fN flows in), row rw + 1 = FC (its fS flows in)
            float eW = wave_prev0(st.FB[i][1][1]), wE = wave_next0(st.FB[i][0][0]);
            if (XEDGE) {
                eW = lane_x0 ? st.FB[i][0][1] : eW;
                wE = lane_x1 ? st.FB[i][1][0] : wE;
            }
            float nS[2], sN[2];  // fN of row rw - 1, fS of row rw + 1
#pragma unroll
            for (int e = 0; e < 2; e++) {
                nS[e] = st.FA[i][e][3];
                sN[e] = FC[e][2];
                if (COND) {
                    if (rw <= g.zc0) nS[e] = st.FB[i][e][3];
                    if (rw >= g.zc1) sN[e] = st.FB[i][e][2];
                }
            }
            if (i < NST - 1) {
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const float inE = e == 0 ? eW : st.FB[i][0][1];
                    float inW = e == 1 ? wE : st.FB[i][1][0];
                    if (XEDGE && e == 0 && lane_x1o) inW = st.FB[i][0][0];
                    Wn[e] = update_water(st.Wm[i][e], flux4{st.FB[i][e][0], st.FB[i][e][1], st.FB[i][e][2], st.FB[i][e][3]},
                                         inE, inW, nS[e], sN[e]);
                    Tn[e] = Wn[e] + (e == 0 ? hh.x : hh.y);
                }
            } else {
                // CreateVelocityField + NormalizeMap, FlowMapComponents.cs:120-139,157-165
                float out[2];
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const float fE_w = e == 0 ? eW : st.FB[i][0][1];
                    float fW_e = e == 1 ? wE : st.FB[i][1][0];
                    if (XEDGE && e == 0 && lane_x1o) fW_e = st.FB[i][0][0];
                    const float dl = fE_w - st.FB[i][e][0];
                    const float dr = st.FB[i][e][1] - fW_e;
                    const float dt = sN[e] - st.FB[i][e][3];
                    const float db = st.FB[i][e][2] - nS[e];
                    const float vx = (dl + dr) * 0.5f;
                    const float vy = (dt + db) * 0.5f;
                    float v = sqrtf(vx * vx + vy * vy);
                    if (nrange < 1e-12f) v = 0.0f;
                    out[e] = (v - nmin) / nrange;
                }
                if (store_lane && (COND || rw >= b.loW[NST - 1])) {
                    float *p = dst + (size_t)rw * g.pitch + gx;
                    if (VEC) {
                        *reinterpret_cast<float2 *>(p) = make_float2(out[0], out[1]);
                    } else {
                        if (gx >= 0 && gx < g.cols) p[0] = out[0];
                        if (gx + 1 >= 0 && gx + 1 < g.cols) p[1] = out[1];
                    }
                }
            }
        }
        // the windows of stage i move on; what it produced feeds stage i + 1 (row r - 2) in this same step
#pragma unroll
        for (int e = 0; e < 2; e++) {
            st.Tm[i][e] = st.T0[i][e]; st.T0[i][e] = Tp[e];
            st.Wm[i][e] = st.W0[i][e]; st.W0[i][e] = Wp[e];
            Tp[e] = Tn[e]; Wp[e] = Wn[e];
#pragma unroll
            for (int k = 0; k < 4; k++) FCprev[e][k] = FC[e][k];
        }
    }
#pragma unroll
    for (int e = 0; e < 2; e++)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            st.FA[NACT - 1][e][k] = st.FB[NACT - 1][e][k];
            st.FB[NACT - 1][e][k] = FCprev[e][k];
        }
}

// Waves of a SIMD are served oldest first: left alone, the three waves of a SIMD finish one after the other and the last
// one runs alone (at most one VALU instruction per four cycles) for a quarter of the launch.  Priority by progress keeps
// them together: a wave that is a step ahead of another (modulo 4) yields to it.
__device__ __forceinline__ void fs_prio(int step) {
    switch (step & 3) {
        case 0: __builtin_amdgcn_s_setprio(3); break;
        case 1: __builtin_amdgcn_s_setprio(2); break;
        case 2: __builtin_amdgcn_s_setprio(1); break;
        default: __builtin_amdgcn_s_setprio(0); break;
    }
}

template <bool VEC>
__device__ __forceinline__ float2 fs_load_row(const float *__restrict__ h, const nz_geom &g, int row, int gx) {
    if (VEC) return *reinterpret_cast<const float2 *>(h + (size_t)row * g.pitch + gx);
    const size_t base = (size_t)row * g.pitch;
    return make_float2(h[base + clampi(gx, 0, g.cols - 1)], h[base + clampi(gx + 1, 0, g.cols - 1)]);
}

template <int NST, bool XEDGE>
__device__ __forceinline__ void flow_stream_body(float2 *ring, const float *__restrict__ h, float *__restrict__ dst,
                                                 const nz_geom &g, const int lx0, const int s0, const int s1,
                                                 const float nmin, const float nrange) {
    constexpr bool VEC = !XEDGE;
    constexpr int H = 2 * NST;
    const int lane = threadIdx.x;
    const int gx = lx0 + 2 * lane;
    // the grid's last column is the second of its lane's two columns when the row length is even, the first when it is odd
    const bool lane_x0 = gx == 0, lane_x1 = gx + 1 == g.cols - 1, lane_x1o = gx == g.cols - 1;
    const bool store_lane = 2 * lane >= H && 2 * lane < FS_TW - H;
    fs_bounds b;
#pragma unroll
    for (int i = 0; i < NST; i++) {
        const int m = 2 * (NST - 1 - i) + 1;
        b.loF[i] = max(g.zc0, s0 - m);
        b.hiF[i] = min(g.zc1 + 1, s1 + m);
        b.loW[i] = max(g.zc0, s0 - m + 1);
        b.hiW[i] = min(g.zc1 + 1, s1 + m - 1);
    }
    fs_state<NST> st;
#pragma unroll
    for (int i = 0; i < NST; i++)
#pragma unroll
        for (int e = 0; e < 2; e++) {
            st.Tm[i][e] = 0.0f; st.T0[i][e] = 0.0f; st.Wm[i][e] = 0.0001f; st.W0[i][e] = 0.0001f;
#pragma unroll
            for (int k = 0; k < 4; k++) { st.FA[i][e][k] = 0.0f; st.FB[i][e][k] = 0.0f; }
        }
    // rows read: [t0 - 1, t1], all clamped into the grid's rows (a clamped row is only ever a cell's z-neighbour beyond
    // the grid, which the COND steps replace)
    const int t0 = b.loF[0], t1 = s1 + H - 1;
    {
        const float2 hm = fs_load_row<VEC>(h, g, max(t0 - 1, g.zc0), gx), h0 = fs_load_row<VEC>(h, g, t0, gx);
        ring[(t0 & (FS_RING - 1)) * 64] = h0;
        st.Tm[0][0] = 0.0001f + hm.x; st.Tm[0][1] = 0.0001f + hm.y;
        st.T0[0][0] = 0.0001f + h0.x; st.T0[0][1] = 0.0001f + h0.y;
    }
    float2 hp = fs_load_row<VEC>(h, g, min(t0 + 1, g.zc1), gx);
    ring[((t0 + 1) & (FS_RING - 1)) * 64] = hp;
    int t = t0;
#define NZ_FS_STEP(NA, C, T, HP, HN)                                                                              \
    do {                                                                                                          \
        if (NZ_FS_PRIO) fs_prio((T) - t0);                                                                        \
        fs_step<NST, NA, C, XEDGE, VEC>(st, T, HP, HN, ring, b, gx, lane_x0, lane_x1, lane_x1o, g, nmin, nrange, dst, \
                                        store_lane);                                                              \
    } while (0)
#define NZ_FS_PHASE(K)                                                            \
    if (NST > K) {                                                                \
        for (int q = 0; q < 4 && t < t1; q++, t++) {                              \
            const float2 hn = fs_load_row<VEC>(h, g, min(t + 2, g.zc1), gx);      \
            NZ_FS_STEP((K < NST ? K : NST), false, t, hp, hn);                    \
            hp = hn;                                                              \
        }                                                                         \
    }
    if (s0 - (H - 1) > g.zc0 && s1 + H - 1 <= g.zc1) {
        // pipeline fill away from the grid's first and last rows: stage i joins at row s0 - m_i, four steps after stage i - 1
        NZ_FS_PHASE(1)
        NZ_FS_PHASE(2)
        NZ_FS_PHASE(3)
        NZ_FS_PHASE(4)
    } else {
        // on the grid's first rows stage i joins at row zc0 (two steps apart) and border cells need their clamped
        // z-neighbours: the last stage's row zc0 + 1, the first whose z-neighbours are both real, is reached at t = zc0 + H - 1
        const int tfill = min(t1, max(s0, g.zc0 + 1) + H - 1);
        for (; t < tfill; t++) {
            const float2 hn = fs_load_row<VEC>(h, g, min(t + 2, g.zc1), gx);
            NZ_FS_STEP(NST, true, t, hp, hn);
            hp = hn;
        }
    }
    NZ_FPROBE(2, __builtin_amdgcn_s_memrealtime());
    NZ_FPROBE(3, __builtin_amdgcn_s_memtime());
    // steady state: every stage is inside its row range and no stage is on the grid's first or last row (the first stage
    // reaches the last row, zc1, at t = zc1).  Three steps per trip: a row window is three registers deep while a step runs
    // (rows r - 1, r and the incoming r + 1), so after three steps every value is back in the register it started in and
    // the windows rotate by renaming, not by moves.
    const int tsteady = min(t1, g.zc1);
    for (; t + 2 < tsteady; t += 3) {
        const float2 hn = fs_load_row<VEC>(h, g, t + 2, gx);
        NZ_FS_STEP(NST, false, t, hp, hn);
        const float2 hn2 = fs_load_row<VEC>(h, g, min(t + 3, g.zc1), gx);
        NZ_FS_STEP(NST, false, t + 1, hn, hn2);
        const float2 hn3 = fs_load_row<VEC>(h, g, min(t + 4, g.zc1), gx);
        NZ_FS_STEP(NST, false, t + 2, hn2, hn3);
        hp = hn3;
    }
    // the last one or two steps of an inner segment; the drain of a segment that ends on the grid's last row
    for (; t < t1; t++) {
        const float2 hn = fs_load_row<VEC>(h, g, min(t + 2, g.zc1), gx);
        NZ_FS_STEP(NST, true, t, hp, hn);
        hp = hn;
    }
#undef NZ_FS_PHASE
#undef NZ_FS_STEP
}

template <int NST>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NST >= 4 ? 3 : 4))) void flow_stream_kernel(
    const float *__restrict__ h, float *__restrict__ dst, nz_geom g, int S, int nstrips, float nmin, float nrange,
    int aligned) {
    __shared__ float2 s_ring[FS_RING * 64];
    constexpr int H = 2 * NST, OW = FS_TW - 2 * H;
    const int strip = blockIdx.x % nstrips, seg = blockIdx.x / nstrips;
    const int lx0 = strip * OW - H;
    const int s0 = g.or0 + seg * S, s1 = min(s0 + S, g.or1);
    const size_t off = blockIdx.y * g.bstride;  // batched launch: one independent grid per blockIdx.y
    const bool inner = aligned && lx0 > 0 && lx0 + FS_TW < g.cols;
    float2 *ring = s_ring + threadIdx.x;
    NZ_FPROBE(0, __builtin_amdgcn_s_memrealtime());
    NZ_FPROBE(1, __builtin_amdgcn_s_memtime());
    NZ_FPROBE(6, (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)));   // HW_ID
    NZ_FPROBE(7, (unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) | ((unsigned long long)inner << 32));  // XCC_ID
    if (inner) flow_stream_body<NST, false>(ring, h + off, dst + off, g, lx0, s0, s1, nmin, nrange);
    else flow_stream_body<NST, true>(ring, h + off, dst + off, g, lx0, s0, s1, nmin, nrange);
    NZ_FPROBE(4, __builtin_amdgcn_s_memrealtime());
    NZ_FPROBE(5, __builtin_amdgcn_s_memtime());
}

}  // namespace

int nz_flow_fused_max() {
    static const int cap = getenv("NZ_FLOW_NMAX") ? atoi(getenv("NZ_FLOW_NMAX")) : 5;
    return cap < 1 ? 1 : (cap > FT_MAX_N ? FT_MAX_N : cap);
}

// The whole stage (first && last) in its row-streaming form.  A segment of S rows per wave: S is chosen so that the grid
// is about one round of waves for the chip (NZ_FLOW_STREAM_WAVES resident waves; longer segments waste less on the pipeline
// fill, but a second, partial round of waves would cost more), never below 16 rows.
static bool nz_flow_stream_wanted(const nz_geom &g, int n) {
    static const int mode = getenv("NZ_FLOW_STREAM") ? atoi(getenv("NZ_FLOW_STREAM")) : 1;
    if (mode == 0 || n < 1 || n > FT_MAX_N) return false;
    if (mode == 2) return true;  // test matrix: every size
    return (long long)g.cols * (g.or1 - g.or0) * g.count >= 1024ll * 1024;
}

static int32_t nz_launch_flow_stream(hipStream_t s, const float *h, float *dst, const nz_geom &g, int n, float nmin,
                                     float nrange) {
    static const int waves = getenv("NZ_FLOW_STREAM_WAVES") ? atoi(getenv("NZ_FLOW_STREAM_WAVES")) : 3072;
    static const int s_env = getenv("NZ_FLOW_STREAM_S") ? atoi(getenv("NZ_FLOW_STREAM_S")) : 0;
    const int H = 2 * n, OW = FS_TW - 2 * H;
    const int nstrips = (g.cols + OW - 1) / OW, rows = g.or1 - g.or0;
    long long per = (long long)nstrips * g.count;
    int nseg = (int)(waves / per > 0 ? waves / per : 1);
    int S = (rows + nseg - 1) / nseg;
    if (S < 16) S = 16;
    if (s_env > 0) S = s_env;
    nseg = (rows + S - 1) / S;
    uintptr_t bits = reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(dst) | (uintptr_t)(g.pitch * 4) |
                     (uintptr_t)(g.bstride * 4);
    const int aligned = (bits & 7) == 0;
    const dim3 grid((unsigned)(nstrips * nseg), g.count);
#define NZ_FS(N) hipLaunchKernelGGL((flow_stream_kernel<N>), grid, dim3(64), 0, s, h, dst, g, S, nstrips, nmin, nrange, aligned)
    switch (n) {
        case 1: NZ_FS(1); break;
        case 2: NZ_FS(2); break;
        case 3: NZ_FS(3); break;
        case 4: NZ_FS(4); break;
        default: NZ_FS(5); break;
    }
#undef NZ_FS
    NZ_HIP(hipGetLastError());
    return NZ_OK;
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
    // OCC = waves per SIMD the register allocator must leave room for: 4 -> two 512-thread workgroups per CU
    static const int occ = getenv("NZ_FLOW_OCC") ? atoi(getenv("NZ_FLOW_OCC")) : 4;
#define NZ_FF(F, L)                                                                                                  \
    do {                                                                                                             \
        if (occ >= 4)                                                                                                \
            hipLaunchKernelGGL((flow_fused_kernel<F, L, NZ_FT_OCC>), dim3((unsigned)blocks, g.count), dim3(FT_NT), 0, s, h, w_in, \
                               fN_in, fS_in, fE_in, fW_in, w_out, fN_out, fS_out, fE_out, fW_out, dst, h_out, g, n,   \
                               nmin, nrange, aligned);                                                                     \
        else                                                                                                         \
            hipLaunchKernelGGL((flow_fused_kernel<F, L, 2>), dim3((unsigned)blocks, g.count), dim3(FT_NT), 0, s, h, w_in, \
                               fN_in, fS_in, fE_in, fW_in, w_out, fN_out, fS_out, fE_out, fW_out, dst, h_out, g, n,   \
                               nmin, nrange, aligned);                                                                     \
    } while (0)
    if (first && last) NZ_FF(true, true);
    else if (first) NZ_FF(true, false);
    else if (last) NZ_FF(false, true);
    else NZ_FF(false, false);
#undef NZ_FF
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

int32_t nz_launch_fill(hipStream_t s, float *data, size_t n, float value) {
    if (n == 0) return NZ_OK;
    size_t blocks = (n + (size_t)CT * 4 - 1) / ((size_t)CT * 4);
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)blocks), dim3(CT), 0, s, data, n, value);
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

int32_t nz_launch_copy(hipStream_t s, float *dst, const float *src, size_t n) {
    if (n == 0) return NZ_OK;
    size_t blocks = (n + (size_t)CT * 4 - 1) / ((size_t)CT * 4);
    hipLaunchKernelGGL(copy_kernel, dim3((unsigned)blocks), dim3(CT), 0, s, dst, src, n);
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
    hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)blocks), dim3(CT), 0, s, src, dst, n, nmin, nrange);
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

#ifdef NZ_FLOW_PROBE
extern "C" int32_t nz_debug_set_flow_probe(void *buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(nz_flow_probe_buf), &buf, sizeof buf) == hipSuccess ? 0 : -3;
}
#endif
