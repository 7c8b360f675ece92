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


constexpr int FT_MAX_N = 5;  // iterations one fused launch can hold

}  // namespace
