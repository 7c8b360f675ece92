// nz_flow_common.hpp -- per-cell functions of the pipe-model flow map shared by nz_flow.hip and nz_flow_stream.hip.
#pragma once

#include "nz_internal.hpp"

namespace {

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


// ---- relaxed mode (NZ_FLOAT_RELAXED) -----------------------------------------------------------------------------------
// The reference compiles the flow jobs with FloatMode.Fast (Geologic/FlowMap/FlowMapJob.cs:16).  The relaxed forms keep
// every max / clamp / comparison and the operand order of the sums, and replace: the IEEE division of the outflow scale
// by water * rcp(sum * dt) (v_rcp_f32, 1 ulp: ten VALU instructions less per cell and iteration), the water update's
// multiply-add by one FMA, the velocity's sqrt by v_sqrt_f32 (1 ulp) behind an FMA, and the normalisation's division by the
// wave-uniform range by a multiplication with its reciprocal.  Used by BOTH fused kernels (tile and row-streaming), which
// therefore still agree bit for bit within the mode.  The map amplifies one ulp (total = water + height rounds the water to
// the height's ulp), so these forms leave the strict result's 1e-5 band in ~1e-4 of the cells: a mode of its own
// (include/noize_hip.h), NZ_FLOAT_FAST runs the strict forms.
template <bool FAST>
__device__ __forceinline__ flux4 compute_flow_m(float totalHt, float water_0, float tW, float tE, float tS, float tN, flux4 old,
                                                bool branch_free) {
    if (!FAST) return branch_free ? compute_flow_nb(totalHt, water_0, tW, tE, tS, tN, old) : compute_flow(totalHt, water_0, tW, tE, tS, tN, old);
    float dW = totalHt - tW, dE = totalHt - tE, dS = totalHt - tS, dN = totalHt - tN;
    flux4 f;
    f.w = fmaxf(0.0f, old.w + dW);
    f.e = fmaxf(0.0f, old.e + dE);
    f.s = fmaxf(0.0f, old.s + dS);
    f.n = fmaxf(0.0f, old.n + dN);
    float sum_ = (f.w + f.e) + (f.s + f.n);
    // sum_ == 0: rcp = +inf, K = +inf or NaN, clamped to 1 -- and the four +0 stay +0 (see compute_flow_nb)
    float K = water_0 * __builtin_amdgcn_rcpf(sum_ * TIMESTEP);
    K = fmaxf(0.0f, fminf(1.0f, K));
    f.w *= K; f.e *= K; f.s *= K; f.n *= K;
    return f;
}

template <bool FAST>
__device__ __forceinline__ float update_water_m(float water, flux4 own, float fE_west, float fW_east, float fN_south,
                                                float fS_north) {
    if (!FAST) return update_water(water, own, fE_west, fW_east, fN_south, fS_north);
    float flowOUT = own.w + own.e + own.s + own.n;
    float flowIN = ((fE_west + fW_east) + fN_south) + fS_north;  // (0 + a is a)
    return fmaxf(0.0f, __builtin_fmaf(flowIN - flowOUT, TIMESTEP, water));
}

// CreateVelocityField + NormalizeMap (FlowMapComponents.cs:120-139,157-165) of one cell from its four flux differences;
// inv_range = 1 / nrange (tolerance mode only)
template <bool FAST>
__device__ __forceinline__ float velocity_norm_m(float dl, float dr, float dt, float db, float nmin, float nrange, float inv_range) {
    const float vx = (dl + dr) * 0.5f;
    const float vy = (dt + db) * 0.5f;
    if (!FAST) {
        float v = sqrtf(vx * vx + vy * vy);
        if (nrange < 1e-12f) v = 0.0f;
        return (v - nmin) / nrange;
    }
    float v = __builtin_amdgcn_sqrtf(__builtin_fmaf(vx, vx, vy * vy));
    if (nrange < 1e-12f) v = 0.0f;
    return (v - nmin) * inv_range;
}

constexpr int FT_MAX_N = 5;  // iterations one fused launch can hold

}  // namespace
