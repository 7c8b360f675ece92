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
//  * flow_iter_kernel: one whole iteration (outflow + water update) per launch for the stage body.
//    A workgroup stages water+height of a 16 x 64 tile (+2 halo) and the old flux (+1 halo) once,
//    recomputes the neighbours' new outflow in LDS instead of re-reading it from HBM, and writes the
//    five state planes once: 44 B/cell instead of the 104 B/cell the reference moves per iteration.
//    State ping-pongs between two plane sets, the reference's READ/WRITE pairs
//    (FlowMapStage.cs:52-62), minus the four serial flush copies.
//
// Arithmetic follows the C# order exactly (W,E,S,N; csum = ((x+y)+z)+w; true divisions), no FMA
// contraction, so results are bit-identical to the CPU restatement.
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
    float sum_ = f.w + f.e + f.s + f.n;
    if (sum_ > 0.0f) {
        float K = water_0 / (sum_ * TIMESTEP);
        K = fmaxf(0.0f, fminf(1.0f, K));
        f.w *= K; f.e *= K; f.s *= K; f.n *= K;
    } else {
        f.w = 0.0f; f.e = 0.0f; f.s = 0.0f; f.n = 0.0f;
    }
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

// ---- fused iteration -------------------------------------------------------------------------
constexpr int FOW = 64, FOH = 16;           // output tile
constexpr int TW2 = FOW + 4, TH2 = FOH + 4; // water/total tile (halo 2)
constexpr int TW1 = FOW + 2, TH1 = FOH + 2; // flux tile (halo 1)

template <bool FIRST>
__global__ __launch_bounds__(CT) void flow_iter_kernel(const float *__restrict__ h, const float *__restrict__ w_in,
                                                      const float *__restrict__ fN_in, const float *__restrict__ fS_in,
                                                      const float *__restrict__ fE_in, const float *__restrict__ fW_in,
                                                      float *__restrict__ w_out, float *__restrict__ fN_out,
                                                      float *__restrict__ fS_out, float *__restrict__ fE_out,
                                                      float *__restrict__ fW_out, nz_geom g) {
    __shared__ float s_tot[TH2 * TW2];  // water + height, halo 2
    __shared__ float s_wat[TH2 * TW2];  // water, halo 2 (only halo 1 is used)
    __shared__ float s_fw[TH1 * TW1], s_fe[TH1 * TW1], s_fs[TH1 * TW1], s_fn[TH1 * TW1];  // new flux, halo 1

    const int tid = threadIdx.x;
    const int tiles_x = (g.cols + FOW - 1) / FOW;
    const int by = blockIdx.x / tiles_x, bx = blockIdx.x - by * tiles_x;
    const int ox0 = bx * FOW, oz0 = g.or0 + by * FOH;

    // stage water and water+height with clamp-to-edge reads (ReadTileData.GetData)
    for (int idx = tid; idx < TH2 * TW2; idx += CT) {
        int r = idx / TW2, c = idx - r * TW2;
        int gx = clampi(ox0 - 2 + c, 0, g.cols - 1);
        int gz = clampi(oz0 - 2 + r, g.zc0, g.zc1);
        size_t gi = (size_t)gz * g.pitch + gx;
        float wv = FIRST ? 0.0001f : w_in[gi];
        s_wat[idx] = wv;
        s_tot[idx] = wv + h[gi];
    }
    __syncthreads();

    // new outflow of every cell of the halo-1 region
    for (int idx = tid; idx < TH1 * TW1; idx += CT) {
        int r = idx / TW1, c = idx - r * TW1;
        int gx = ox0 - 1 + c, gz = oz0 - 1 + r;
        int cx = clampi(gx, 0, g.cols - 1), cz = clampi(gz, g.zc0, g.zc1);
        // LDS coordinates (halo-2 frame) of the clamped cell and its clamped neighbours
        int lr = cz - (oz0 - 2), lc = cx - (ox0 - 2);
        int lcW = clampi(cx - 1, 0, g.cols - 1) - (ox0 - 2), lcE = clampi(cx + 1, 0, g.cols - 1) - (ox0 - 2);
        int lrS = clampi(cz - 1, g.zc0, g.zc1) - (oz0 - 2), lrN = clampi(cz + 1, g.zc0, g.zc1) - (oz0 - 2);
        lr = clampi(lr, 0, TH2 - 1); lc = clampi(lc, 0, TW2 - 1);
        lcW = clampi(lcW, 0, TW2 - 1); lcE = clampi(lcE, 0, TW2 - 1);
        lrS = clampi(lrS, 0, TH2 - 1); lrN = clampi(lrN, 0, TH2 - 1);
        flux4 old = {0.0f, 0.0f, 0.0f, 0.0f};
        if (!FIRST) {
            size_t gi = (size_t)cz * g.pitch + cx;
            old.w = fW_in[gi]; old.e = fE_in[gi]; old.s = fS_in[gi]; old.n = fN_in[gi];
        }
        flux4 f = compute_flow(s_tot[lr * TW2 + lc], s_wat[lr * TW2 + lc], s_tot[lr * TW2 + lcW],
                               s_tot[lr * TW2 + lcE], s_tot[lrS * TW2 + lc], s_tot[lrN * TW2 + lc], old);
        s_fw[idx] = f.w; s_fe[idx] = f.e; s_fs[idx] = f.s; s_fn[idx] = f.n;
    }
    __syncthreads();

    // water update + stores: 4 consecutive cells per thread
    {
        int r = tid >> 4, c0 = (tid & 15) * 4;
        int gz = oz0 + r;
        if (gz < g.or1) {
            float ow[4], oe[4], os[4], on[4], wat[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                int c = c0 + e;
                int gx = ox0 + c;
                int cx = gx < g.cols ? gx : g.cols - 1;
                // flux-frame coordinates: cell (r,c) sits at (r+1, c+1); a clamped neighbour is the cell itself
                int fr = r + 1, fc = cx - ox0 + 1;
                int fcW = cx > 0 ? fc - 1 : fc, fcE = cx < g.cols - 1 ? fc + 1 : fc;
                int frS = gz > g.zc0 ? fr - 1 : fr, frN = gz < g.zc1 ? fr + 1 : fr;
                fc = clampi(fc, 0, TW1 - 1); fcE = clampi(fcE, 0, TW1 - 1);
                flux4 own = {s_fw[fr * TW1 + fc], s_fe[fr * TW1 + fc], s_fs[fr * TW1 + fc], s_fn[fr * TW1 + fc]};
                float wv = s_wat[(r + 2) * TW2 + clampi(cx - ox0 + 2, 0, TW2 - 1)];
                wat[e] = update_water(wv, own, s_fe[fr * TW1 + fcW], s_fw[fr * TW1 + fcE], s_fn[frS * TW1 + fc],
                                      s_fs[frN * TW1 + fc]);
                ow[e] = own.w; oe[e] = own.e; os[e] = own.s; on[e] = own.n;
            }
            size_t gi = (size_t)gz * g.pitch + ox0 + c0;
            bool vec_ok = (ox0 + c0 + 4 <= g.cols) && ((g.pitch & 3) == 0) &&
                          (((reinterpret_cast<uintptr_t>(w_out) | reinterpret_cast<uintptr_t>(fN_out) |
                             reinterpret_cast<uintptr_t>(fS_out) | reinterpret_cast<uintptr_t>(fE_out) |
                             reinterpret_cast<uintptr_t>(fW_out)) & 15) == 0);
            if (vec_ok) {
                *reinterpret_cast<float4 *>(fW_out + gi) = make_float4(ow[0], ow[1], ow[2], ow[3]);
                *reinterpret_cast<float4 *>(fE_out + gi) = make_float4(oe[0], oe[1], oe[2], oe[3]);
                *reinterpret_cast<float4 *>(fS_out + gi) = make_float4(os[0], os[1], os[2], os[3]);
                *reinterpret_cast<float4 *>(fN_out + gi) = make_float4(on[0], on[1], on[2], on[3]);
                *reinterpret_cast<float4 *>(w_out + gi) = make_float4(wat[0], wat[1], wat[2], wat[3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    if (ox0 + c0 + e < g.cols) {
                        fW_out[gi + e] = ow[e]; fE_out[gi + e] = oe[e]; fS_out[gi + e] = os[e];
                        fN_out[gi + e] = on[e]; w_out[gi + e] = wat[e];
                    }
                }
            }
        }
    }
}

}  // namespace

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

int32_t nz_launch_flow_iter(hipStream_t s, const float *h, const float *w_in, const float *fN_in,
                            const float *fS_in, const float *fE_in, const float *fW_in, float *w_out,
                            float *fN_out, float *fS_out, float *fE_out, float *fW_out, const nz_geom &g,
                            int first) {
    if (g.or1 <= g.or0) return NZ_OK;
    long long blocks = (long long)((g.cols + FOW - 1) / FOW) * ((g.or1 - g.or0 + FOH - 1) / FOH);
    if (first) {
        hipLaunchKernelGGL((flow_iter_kernel<true>), dim3((unsigned)blocks), dim3(CT), 0, s, h, w_in, fN_in, fS_in,
                           fE_in, fW_in, w_out, fN_out, fS_out, fE_out, fW_out, g);
    } else {
        hipLaunchKernelGGL((flow_iter_kernel<false>), dim3((unsigned)blocks), dim3(CT), 0, s, h, w_in, fN_in, fS_in,
                           fE_in, fW_in, w_out, fN_out, fS_out, fE_out, fW_out, g);
    }
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}
