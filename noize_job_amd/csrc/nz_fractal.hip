// nz_fractal.hip -- fBm octave accumulation over the reference's noise bases (gfx950).
//
// Replaces FractalJob<FractalGenerator<N>, WriteTileData> (Noise/Fractal/Fractal.cs:19-74) for the
// eight N of Noise/NoiseStage.cs:26-35.  The arithmetic below is the oracle's operation sequence
// (oracle/noize_oracle.c), compiled with -ffp-contract=off: the top octaves
// multiply coordinates by up to 2^12, so the skew/unskew steps cancel catastrophically and any
// re-association or FMA contraction moves the result by more than the 1e-5 tolerance.  Every
// operation is an IEEE-754 binary32 add/mul/div/floor, which makes the result bit-identical to the
// CPU restatement.  psrnoise's cos/sin come from a table built by the host libm (its hash is an exact
// small integer), so no device trig is involved except for the Sin basis.  The lattice hashes of the
// other bases are exact small-integer arithmetic as well: the default kernels read them (and the decoded
// corner gradients) from host-built tables in LDS and fall back to the direct evaluation below wherever a
// coordinate leaves the range the tables are exact for.
//
// VALU-bound (~1.1 k fp32 ops per cell for 13 simplex octaves against 4 B written): one thread
// produces VEC consecutive cells of a row so the octave chains of different cells interleave, and
// stores them with one 8/16-byte coalesced write.  Batched launches take the tile from blockIdx.y.
#include <cstdlib>

#include "nz_internal.hpp"

namespace {

__device__ __forceinline__ float fracf_(float x) { return x - floorf(x); }
__device__ __forceinline__ float lerpf_(float a, float b, float s) { return a + s * (b - a); }
__device__ __forceinline__ float stepf_(float y, float x) { return x >= y ? 1.0f : 0.0f; }
__device__ __forceinline__ float mod289f(float x) { return x - floorf(x * (1.0f / 289.0f)) * 289.0f; }
__device__ __forceinline__ float mod7f(float x) { return x - floorf(x * (1.0f / 7.0f)) * 7.0f; }
__device__ __forceinline__ float permutef(float x) { return mod289f((34.0f * x + 1.0f) * x); }

// Exact-product forms.  fmaf(a, b, c) == round(round(a*b) + c) whenever a*b is exactly representable,
// so where both factors are small integers (every hash argument of snoise / cnoise / cellular is an
// integer below 2^12, every product below 2^24) the fused instruction returns the very same bits as
// the reference's separate multiply and add, one VALU slot cheaper.  NOT valid for psrnoise's first
// permute, whose un-reduced argument overflows 2^24.
#ifndef NZ_EXACT_FMA
#define NZ_EXACT_FMA 1
#endif
__device__ __forceinline__ float mod289i(float x) {  // x: integer-valued, |x| < 2^24
#if NZ_EXACT_FMA
    return __builtin_fmaf(-floorf(x * (1.0f / 289.0f)), 289.0f, x);
#else
    return mod289f(x);
#endif
}
__device__ __forceinline__ float permutei(float x) {  // x: integer-valued, 0 <= x <= 700
#if NZ_EXACT_FMA
    return mod289i(__builtin_fmaf(34.0f, x, 1.0f) * x);
#else
    return permutef(x);
#endif
}
__device__ __forceinline__ float twice_minus_one(float f) {  // 2*f is exact
#if NZ_EXACT_FMA
    return __builtin_fmaf(2.0f, f, -1.0f);
#else
    return 2.0f * f - 1.0f;
#endif
}
__device__ __forceinline__ float taylor_inv_sqrt(float r) { return 1.79284291400159f - 0.85373472095314f * r; }
__device__ __forceinline__ float fadef(float t) { return t * t * t * (t * (t * 6.0f - 15.0f) + 10.0f); }
__device__ __forceinline__ float rectify(float v) { return (1.0f + v) / 2.0f * 1.0f; }

// noise.cnoise(float2), Appendix A.2
__device__ __forceinline__ float cnoise2(float Px, float Py) {
    float flx = floorf(Px), fly = floorf(Py);
    float frx = Px - flx, fry = Py - fly;
    float Pi0 = mod289f(flx + 0.0f), Pi1 = mod289f(fly + 0.0f);
    float Pi2 = mod289f(flx + 1.0f), Pi3 = mod289f(fly + 1.0f);
    float Pf0 = frx - 0.0f, Pf1 = fry - 0.0f, Pf2 = frx - 1.0f, Pf3 = fry - 1.0f;
    float ix[4] = {Pi0, Pi2, Pi0, Pi2};
    float iy[4] = {Pi1, Pi1, Pi3, Pi3};
    float gx[4], gy[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float i = permutef(permutef(ix[k]) + iy[k]);
        float g = fracf_(i * (1.0f / 41.0f)) * 2.0f - 1.0f;
        gy[k] = fabsf(g) - 0.5f;
        float tx = floorf(g + 0.5f);
        gx[k] = g - tx;
    }
    float g00x = gx[0], g00y = gy[0], g10x = gx[1], g10y = gy[1];
    float g01x = gx[2], g01y = gy[2], g11x = gx[3], g11y = gy[3];
    float n0 = taylor_inv_sqrt(g00x * g00x + g00y * g00y);
    float n1 = taylor_inv_sqrt(g01x * g01x + g01y * g01y);
    float n2 = taylor_inv_sqrt(g10x * g10x + g10y * g10y);
    float n3 = taylor_inv_sqrt(g11x * g11x + g11y * g11y);
    g00x *= n0; g00y *= n0;
    g01x *= n1; g01y *= n1;
    g10x *= n2; g10y *= n2;
    g11x *= n3; g11y *= n3;
    float n00 = g00x * Pf0 + g00y * Pf1;
    float n10 = g10x * Pf2 + g10y * Pf1;
    float n01 = g01x * Pf0 + g01y * Pf3;
    float n11 = g11x * Pf2 + g11y * Pf3;
    float fdx = fadef(Pf0), fdy = fadef(Pf1);
    float nx0 = lerpf_(n00, n10, fdx);
    float nx1 = lerpf_(n01, n11, fdx);
    return 2.3f * lerpf_(nx0, nx1, fdy);
}

// noise.snoise(float2), Appendix A.3
__device__ __forceinline__ float snoise2(float vx, float vy) {
    const float Cx = 0.211324865405187f, Cy = 0.366025403784439f;
    const float Cz = -0.577350269189626f, Cw = 0.024390243902439f;
    float s = vx * Cy + vy * Cy;
    float ix = floorf(vx + s), iy = floorf(vy + s);
    float t = ix * Cx + iy * Cx;
    float x0x = vx - ix + t, x0y = vy - iy + t;
    bool gt = x0x > x0y;
    float i1x = gt ? 1.0f : 0.0f, i1y = gt ? 0.0f : 1.0f;
    float x12x = x0x + Cx, x12y = x0y + Cx, x12z = x0x + Cz, x12w = x0y + Cz;
    x12x -= i1x;
    x12y -= i1y;
    ix = mod289f(ix);  // plain form: the lattice coordinate itself may exceed 2^24 (the exact-FMA form may not)
    iy = mod289f(iy);
    // (adding the literal 0 of the first corner is the identity on these non-negative integers)
    float p0 = permutei(permutei(iy) + ix);
    float p1 = permutei(permutei(iy + i1y) + ix + i1x);
    float p2 = permutei(permutei(iy + 1.0f) + ix + 1.0f);
    float m0 = fmaxf(0.5f - (x0x * x0x + x0y * x0y), 0.0f);
    float m1 = fmaxf(0.5f - (x12x * x12x + x12y * x12y), 0.0f);
    float m2 = fmaxf(0.5f - (x12z * x12z + x12w * x12w), 0.0f);
    m0 = m0 * m0; m1 = m1 * m1; m2 = m2 * m2;
    m0 = m0 * m0; m1 = m1 * m1; m2 = m2 * m2;
    float xa = twice_minus_one(fracf_(p0 * Cw));
    float xb = twice_minus_one(fracf_(p1 * Cw));
    float xc = twice_minus_one(fracf_(p2 * Cw));
    float h0 = fabsf(xa) - 0.5f, h1 = fabsf(xb) - 0.5f, h2 = fabsf(xc) - 0.5f;
    float a00 = xa - floorf(xa + 0.5f);
    float a01 = xb - floorf(xb + 0.5f);
    float a02 = xc - floorf(xc + 0.5f);
    m0 *= 1.79284291400159f - 0.85373472095314f * (a00 * a00 + h0 * h0);
    m1 *= 1.79284291400159f - 0.85373472095314f * (a01 * a01 + h1 * h1);
    m2 *= 1.79284291400159f - 0.85373472095314f * (a02 * a02 + h2 * h2);
    float g0 = a00 * x0x + h0 * x0y;
    float g1 = a01 * x12x + h1 * x12y;
    float g2 = a02 * x12z + h2 * x12w;
    return 130.0f * (m0 * g0 + m1 * g1 + m2 * g2);
}

// rgrad2 through the host-built tables: both hash arguments are integer-valued, so permute(p.x) is one
// table read (byte offset of the second table's row) and permute(. + p.y) + cos/sin a second one.
struct psr_tables {
    const int *t1;
    const float2 *t2;
};
__device__ __forceinline__ float2 rgrad2_tab(float px, float py, const psr_tables &tab) {
    int i1 = min(max((int)px + NZ_PSR_O1, 0), NZ_PSR_T1 - 1);
    int off = tab.t1[i1] + 8 * (int)py;
    off = min(max(off, 0), (NZ_PSR_T2 - 1) * 8);
    return *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(tab.t2) + off);
}
// Beyond |pos| ~ 2^21 the lattice coordinates no longer hold their halves, xw + 0.5 yw can come out as a
// half-integer and T1 (indexed by the integer) does not apply: the first permute is evaluated directly
// (IEEE mul/add/floor, the same value the host computes), the second one still comes from T2.
__device__ __forceinline__ float2 rgrad2_direct(float px, float py, const psr_tables &tab) {
    int a = (int)(permutef(px) + py) + NZ_PSR_O2;
    a = min(max(a, 0), NZ_PSR_T2 - 1);
    return tab.t2[a];
}

// C fmod for the arguments psrnoise feeds it: x is a multiple of 0.5 (lattice coordinates), per a small
// integer.  fmod is exact by definition (x - trunc(x/per)*per, no rounding).  With inv_lo a little BELOW
// 1/per, q = floor(|x| * inv_lo) is the true quotient or one less for |x|/per < 2^21, q*per and |x| - q*per
// are exact for |x| < 2^22, so r lands in [0, 2 per) and one conditional subtraction gives the library's
// value, sign (and signed zero) copied from x -- 7 VALU slots, no data-dependent loop.
#ifndef NZ_PSR_FMOD
#define NZ_PSR_FMOD 1
#endif
constexpr float PSR_FAST_LIMIT = 2097152.0f;  // |pos| below this keeps every lattice coordinate under 2^22
__device__ __forceinline__ float fmod_lattice(float x, float per, float inv_lo) {
    float a = fabsf(x);
    float q = floorf(a * inv_lo);
    float r = __builtin_fmaf(-q, per, a);
    float r2 = r - per;
    r = r >= per ? r2 : r;
    return copysignf(r, x);
}

// noise.psrnoise(float2 pos, float2 per = (1010,102), rot), Appendix A.4.
// WX / WY = false: the caller has shown that no lattice coordinate of this sample reaches the period in x / y
// (|p.x| <= |pos.x| + |pos.y| + 2.5 and |p.y| <= |pos.y| + 2 for the three corners), where fmod(x, per) == x bit for bit and
// the wrap is left out: the first six octaves of a 4096^2 tile at the default scale wrap nowhere, the next three only in y.
template <bool WX = true, bool WY = true>
__device__ __forceinline__ float psrnoise2(float posx, float posy, const psr_tables &tab) {
    const float perx = 1010.0f, pery = 102.0f;
    posy += 0.001f;
    float uvx = posx + posy * 0.5f, uvy = posy;
    float i0x = floorf(uvx), i0y = floorf(uvy);
    float f0x = uvx - i0x, f0y = uvy - i0y;
    bool gt = f0x > f0y;
    float i1x = gt ? 1.0f : 0.0f, i1y = gt ? 0.0f : 1.0f;
    float p0x = i0x - i0y * 0.5f, p0y = i0y;
    float p1x = p0x + i1x - i1y * 0.5f, p1y = p0y + i1y;
    float p2x = p0x + 0.5f, p2y = p0y + 1.0f;
    float d0x = posx - p0x, d0y = posy - p0y;
    float d1x = posx - p1x, d1y = posy - p1y;
    float d2x = posx - p2x, d2y = posy - p2y;
    constexpr float ipx = (1.0f / 1010.0f) * (1.0f - 0x1p-22f), ipy = (1.0f / 102.0f) * (1.0f - 0x1p-22f);
    float2 g0, g1, g2;
    if (!WX || !WY || (NZ_PSR_FMOD && fabsf(posx) < PSR_FAST_LIMIT && fabsf(posy) < PSR_FAST_LIMIT)) {
        float xw0 = p0x, xw1 = p1x, xw2 = p2x, yw0 = p0y, yw1 = p1y, yw2 = p2y;
        if (WX) { xw0 = fmod_lattice(p0x, perx, ipx); xw1 = fmod_lattice(p1x, perx, ipx); xw2 = fmod_lattice(p2x, perx, ipx); }
        if (WY) { yw0 = fmod_lattice(p0y, pery, ipy); yw1 = fmod_lattice(p1y, pery, ipy); yw2 = fmod_lattice(p2y, pery, ipy); }
        g0 = rgrad2_tab(xw0 + 0.5f * yw0, yw0, tab);
        g1 = rgrad2_tab(xw1 + 0.5f * yw1, yw1, tab);
        g2 = rgrad2_tab(xw2 + 0.5f * yw2, yw2, tab);
    } else {
        float xw0 = fmodf(p0x, perx), xw1 = fmodf(p1x, perx), xw2 = fmodf(p2x, perx);
        float yw0 = fmodf(p0y, pery), yw1 = fmodf(p1y, pery), yw2 = fmodf(p2y, pery);
        g0 = rgrad2_direct(xw0 + 0.5f * yw0, yw0, tab);
        g1 = rgrad2_direct(xw1 + 0.5f * yw1, yw1, tab);
        g2 = rgrad2_direct(xw2 + 0.5f * yw2, yw2, tab);
    }
    float w0 = g0.x * d0x + g0.y * d0y;
    float w1 = g1.x * d1x + g1.y * d1y;
    float w2 = g2.x * d2x + g2.y * d2y;
    float t0 = 0.8f - (d0x * d0x + d0y * d0y);
    float t1 = 0.8f - (d1x * d1x + d1y * d1y);
    float t2 = 0.8f - (d2x * d2x + d2y * d2y);
    t0 = fmaxf(t0, 0.0f); t1 = fmaxf(t1, 0.0f); t2 = fmaxf(t2, 0.0f);
    float t20 = t0 * t0, t21 = t1 * t1, t22 = t2 * t2;
    float t40 = t20 * t20, t41 = t21 * t21, t42 = t22 * t22;
    float n = t40 * w0 + t41 * w1 + t42 * w2;
    return 11.0f * n;
}

// one column of the 3x3 cellular search, Appendix A.5
__device__ __forceinline__ void cell_column(float pxc, float Piy, float Pfx_off, float Pfy, float d[3]) {
    const float K = 0.142857142857f, Ko = 0.428571428571f;
    const float oi[3] = {-1.0f, 0.0f, 1.0f};
    const float of[3] = {-0.5f, 0.5f, 1.5f};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float p = permutef(pxc + Piy + oi[k]);
        float ox = fracf_(p * K) - Ko;
        float oy = mod7f(floorf(p * K)) * K - Ko;
        float dx = Pfx_off + 1.0f * ox;
        float dy = Pfy - of[k] + 1.0f * oy;
        d[k] = dx * dx + dy * dy;
    }
}

__device__ __forceinline__ float cellular_rect(float Px, float Py) {
    float Pix = mod289f(floorf(Px)), Piy = mod289f(floorf(Py));
    float Pfx = fracf_(Px), Pfy = fracf_(Py);
    float px0 = permutef(Pix + -1.0f), px1 = permutef(Pix + 0.0f), px2 = permutef(Pix + 1.0f);
    float d1[3], d2[3], d3[3], d1a[3];
    cell_column(px0, Piy, Pfx + 0.5f, Pfy, d1);
    cell_column(px1, Piy, Pfx - 0.5f, Pfy, d2);
    cell_column(px2, Piy, Pfx - 1.5f, Pfy, d3);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        d1a[k] = fminf(d1[k], d2[k]);
        d2[k] = fmaxf(d1[k], d2[k]);
        d2[k] = fminf(d2[k], d3[k]);
        d1[k] = fminf(d1a[k], d2[k]);
        d2[k] = fmaxf(d1a[k], d2[k]);
    }
    if (!(d1[0] < d1[1])) { float s = d1[0]; d1[0] = d1[1]; d1[1] = s; }
    if (!(d1[0] < d1[2])) { float s = d1[0]; d1[0] = d1[2]; d1[2] = s; }
    d1[1] = fminf(d1[1], d2[1]);
    d1[2] = fminf(d1[2], d2[2]);
    d1[1] = fminf(d1[1], d1[2]);
    d1[1] = fminf(d1[1], d2[0]);
    float F1 = sqrtf(d1[0]), F2 = sqrtf(d1[1]);
    return rectify(F1) * rectify(F2);  // CellularGetter.Rectify Fractal.cs:274-277
}

// noise.cnoise(float3), Appendix A.6
__device__ __forceinline__ float cnoise3(float Px, float Py, float Pz) {
    float Pi0[3] = {floorf(Px), floorf(Py), floorf(Pz)};
    float Pf0[3] = {fracf_(Px), fracf_(Py), fracf_(Pz)};
    float Pi1[3], Pf1[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        Pi1[k] = Pi0[k] + 1.0f;
        Pi0[k] = mod289f(Pi0[k]);
        Pi1[k] = mod289f(Pi1[k]);
        Pf1[k] = Pf0[k] - 1.0f;
    }
    float ix[4] = {Pi0[0], Pi1[0], Pi0[0], Pi1[0]};
    float iy[4] = {Pi0[1], Pi0[1], Pi1[1], Pi1[1]};
    float n[2][4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float ixy = permutef(permutef(ix[k]) + iy[k]);
#pragma unroll
        for (int s = 0; s < 2; s++) {
            float ixyz = permutef(ixy + (s ? Pi1[2] : Pi0[2]));
            float gxx = ixyz * (1.0f / 7.0f);
            float gyy = fracf_(floorf(gxx) * (1.0f / 7.0f)) - 0.5f;
            gxx = fracf_(gxx);
            float gzz = 0.5f - fabsf(gxx) - fabsf(gyy);
            float sz = stepf_(gzz, 0.0f);
            gxx -= sz * (stepf_(0.0f, gxx) - 0.5f);
            gyy -= sz * (stepf_(0.0f, gyy) - 0.5f);
            float nr = taylor_inv_sqrt(gxx * gxx + gyy * gyy + gzz * gzz);
            float ax = gxx * nr, ay = gyy * nr, az = gzz * nr;
            float fx = (k & 1) ? Pf1[0] : Pf0[0];
            float fy = (k & 2) ? Pf1[1] : Pf0[1];
            float fz = s ? Pf1[2] : Pf0[2];
            n[s][k] = ax * fx + ay * fy + az * fz;
        }
    }
    float fdx = fadef(Pf0[0]), fdy = fadef(Pf0[1]), fdz = fadef(Pf0[2]);
    float nz0 = lerpf_(n[0][0], n[1][0], fdz);
    float nz1 = lerpf_(n[0][1], n[1][1], fdz);
    float nz2 = lerpf_(n[0][2], n[1][2], fdz);
    float nz3 = lerpf_(n[0][3], n[1][3], fdz);
    float nyz0 = lerpf_(nz0, nz2, fdy);
    float nyz1 = lerpf_(nz1, nz3, fdy);
    return 2.2f * lerpf_(nyz0, nyz1, fdx);
}

// noise.snoise(float3), Appendix A.6
__device__ __forceinline__ float snoise3(float vx, float vy, float vz) {
    const float Cx = 1.0f / 6.0f, Cy = 1.0f / 3.0f;
    float v[3] = {vx, vy, vz};
    float s = vx * Cy + vy * Cy + vz * Cy;
    float i[3], x0[3];
#pragma unroll
    for (int k = 0; k < 3; k++) i[k] = floorf(v[k] + s);
    float t = i[0] * Cx + i[1] * Cx + i[2] * Cx;
#pragma unroll
    for (int k = 0; k < 3; k++) x0[k] = v[k] - i[k] + t;
    float g[3] = {stepf_(x0[1], x0[0]), stepf_(x0[2], x0[1]), stepf_(x0[0], x0[2])};
    float l[3] = {1.0f - g[0], 1.0f - g[1], 1.0f - g[2]};
    float lz[3] = {l[2], l[0], l[1]};
    float i1[3], i2[3], x1[3], x2[3], x3[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        i1[k] = fminf(g[k], lz[k]);
        i2[k] = fmaxf(g[k], lz[k]);
        x1[k] = x0[k] - i1[k] + Cx;
        x2[k] = x0[k] - i2[k] + Cy;
        x3[k] = x0[k] - 0.5f;
        i[k] = mod289f(i[k]);
    }
    float oz[4] = {0.0f, i1[2], i2[2], 1.0f};
    float oy[4] = {0.0f, i1[1], i2[1], 1.0f};
    float ox[4] = {0.0f, i1[0], i2[0], 1.0f};
    const float n_ = 0.142857142857f;
    const float nsx = n_ * 2.0f - 0.0f, nsy = n_ * 0.5f - 1.0f, nsz = n_ * 1.0f - 0.0f;
    float X[4], Y[4], H[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float p = permutef(permutef(permutef(i[2] + oz[k]) + i[1] + oy[k]) + i[0] + ox[k]);
        float j = p - 49.0f * floorf(p * nsz * nsz);
        float x_ = floorf(j * nsz);
        float y_ = floorf(j - 7.0f * x_);
        X[k] = x_ * nsx + nsy;
        Y[k] = y_ * nsx + nsy;
        H[k] = 1.0f - fabsf(X[k]) - fabsf(Y[k]);
    }
    float b0[4] = {X[0], X[1], Y[0], Y[1]}, b1[4] = {X[2], X[3], Y[2], Y[3]};
    float s0[4], s1[4], sh[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        s0[k] = floorf(b0[k]) * 2.0f + 1.0f;
        s1[k] = floorf(b1[k]) * 2.0f + 1.0f;
        sh[k] = -stepf_(H[k], 0.0f);
    }
    float a0[4] = {b0[0] + s0[0] * sh[0], b0[2] + s0[2] * sh[0], b0[1] + s0[1] * sh[1],
                   b0[3] + s0[3] * sh[1]};
    float a1[4] = {b1[0] + s1[0] * sh[2], b1[2] + s1[2] * sh[2], b1[1] + s1[1] * sh[3],
                   b1[3] + s1[3] * sh[3]};
    float P[4][3] = {{a0[0], a0[1], H[0]}, {a0[2], a0[3], H[1]}, {a1[0], a1[1], H[2]},
                     {a1[2], a1[3], H[3]}};
    float xs[4][3] = {{x0[0], x0[1], x0[2]}, {x1[0], x1[1], x1[2]}, {x2[0], x2[1], x2[2]},
                      {x3[0], x3[1], x3[2]}};
    float m[4], pd[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float nr = taylor_inv_sqrt(P[k][0] * P[k][0] + P[k][1] * P[k][1] + P[k][2] * P[k][2]);
        float px = P[k][0] * nr, py = P[k][1] * nr, pz = P[k][2] * nr;
        m[k] = fmaxf(0.6f - (xs[k][0] * xs[k][0] + xs[k][1] * xs[k][1] + xs[k][2] * xs[k][2]), 0.0f);
        m[k] = m[k] * m[k];
        pd[k] = px * xs[k][0] + py * xs[k][1] + pz * xs[k][2];
    }
    return 42.0f * ((m[0] * m[0]) * pd[0] + (m[1] * m[1]) * pd[1] + (m[2] * m[2]) * pd[2] +
                    (m[3] * m[3]) * pd[3]);
}

__device__ __forceinline__ void domain_rotate(float x, float z, float &xr, float &zr, float &yr) {
    float xz = x + z;
    float s2 = xz * -0.211324865405187f;
    xr = x + s2;
    zr = z + s2;
    yr = xz * -0.577350269189626f;
}

// IMakeNoise.NoiseValue of the eight getters, Fractal.cs:141-278
template <int BASIS>
__device__ __forceinline__ float noise_value(float x, float z, const psr_tables &tab) {
    if constexpr (BASIS == NZ_NOISE_SIN) {
        float vx = 0.5f + (0.5f * sinf(x));
        float vy = 0.5f + (0.5f * sinf(z));
        return vx * vy;
    } else if constexpr (BASIS == NZ_NOISE_PERLIN) {
        return rectify(cnoise2(x, z));
    } else if constexpr (BASIS == NZ_NOISE_PERIODIC_PERLIN || BASIS == NZ_NOISE_ROTATED_SIMPLEX) {
        return rectify(psrnoise2(x, z, tab));
    } else if constexpr (BASIS == NZ_NOISE_SIMPLEX) {
        return rectify(snoise2(x, z));
    } else if constexpr (BASIS == NZ_NOISE_CELLULAR) {
        return cellular_rect(x, z);
    } else if constexpr (BASIS == NZ_NOISE_DOMAIN_ROTATED_PERLIN) {
        float xr, zr, yr;
        domain_rotate(x, z, xr, zr, yr);
        return rectify(cnoise3(xr, zr, yr));
    } else {
        float xr, zr, yr;
        domain_rotate(x, z, xr, zr, yr);
        return rectify(snoise3(xr, zr, yr));
    }
}

// ---- simplex fBm with tabulated lattice hashes -------------------------------------------------------
// In snoise the gradient of a lattice corner is a pure function of two small integers: the inner
// permute argument iy+{0,i1.y,1} in [0,289] and the outer one permute(..)+ix+{0,i1.x,1} in [0,577]
// (fp32 mod289 of an |integer| < 2.3e6 lies in [0,289] -- 289 for some negative multiples of 289 --
// and every product is < 2^24, so all of this is exact integer arithmetic in fp32).  Two tables built by the host with the reference's own operation
// sequence therefore return bit-identical gradients:
//   T1[i]  = 16 * permute(i)                                   (a byte offset into T2)
//   T2[j]  = {a0, h, 1.79284291400159 - 0.85373472095314*(a0*a0 + h*h), 0} of p = permute(j)
// and a corner costs one ds_read_b32, one ds_read_b128 and three integer adds instead of two
// permutes and the gradient decode (26 fp32 ops + 4 floors).  About 87 VALU slots per octave-cell
// instead of 151.
constexpr int NZ_T1_N = 292, NZ_T2_N = 580;
// The tables hold hashes of lattice coordinates reduced to [0, 289]; fp32 mod289 only guarantees that range
// for |integer| < 2.3e6 (and the skew of snoise stretches a coordinate by up to 1.73).  Samples beyond this
// limit (NaN included) take the direct evaluation, which follows the reference's arithmetic wherever it leads.
constexpr float NZ_TAB_LIMIT = 1048576.0f;

// (returns snoise / 2: see the end)
__device__ __forceinline__ float snoise2_tab(float vx, float vy, const int *s_t1, const float4 *s_t2) {
    const float Cx = 0.211324865405187f, Cy = 0.366025403784439f, Cz = -0.577350269189626f;
    float s = vx * Cy + vy * Cy;
    float fx = floorf(vx + s), fy = floorf(vy + s);
    float t = fx * Cx + fy * Cx;
    float x0x = vx - fx + t, x0y = vy - fy + t;
    bool gt = x0x > x0y;
    float i1x = gt ? 1.0f : 0.0f, i1y = gt ? 0.0f : 1.0f;
    float x12x = x0x + Cx, x12y = x0y + Cx, x12z = x0x + Cz, x12w = x0y + Cz;
    x12x -= i1x;
    x12y -= i1y;
    int ixi = (int)mod289i(fx), iyi = (int)mod289i(fy);
    // fp32 mod289 returns 289 (not 0) for some negative multiples of 289 (-8959, -17629, ...): the tables carry
    // that index.  Callers guarantee |v| < NZ_TAB_LIMIT (or NaN, which converts to index 0 and poisons the result
    // through x0 anyway), so both indices are in [0, 289] without a clamp
    int ix16 = ixi << 4;
    // corner 1 is (ix + 1, iy) or (ix, iy + 1): its T1 entry is one of the two already read, selected, not re-read
    int k0 = s_t1[iyi], k2 = s_t1[iyi + 1];
    const char *t2 = reinterpret_cast<const char *>(s_t2);
    int a0 = k0 + ix16, a2 = k2 + ix16;
    int a1 = gt ? a0 + 16 : a2;
    float4 g0 = *reinterpret_cast<const float4 *>(t2 + a0);
    float4 g1 = *reinterpret_cast<const float4 *>(t2 + a1);
    float4 g2 = *reinterpret_cast<const float4 *>(t2 + (a2 + 16));
    // keep the padding lane live: a 16-byte LDS read takes 4 LDS cycles per wave, the 12-byte form 8
    asm volatile("" ::"v"(g0.w), "v"(g1.w), "v"(g2.w));
    float m0 = fmaxf(0.5f - (x0x * x0x + x0y * x0y), 0.0f);
    float m1 = fmaxf(0.5f - (x12x * x12x + x12y * x12y), 0.0f);
    float m2 = fmaxf(0.5f - (x12z * x12z + x12w * x12w), 0.0f);
    m0 = m0 * m0; m1 = m1 * m1; m2 = m2 * m2;
    m0 = m0 * m0; m1 = m1 * m1; m2 = m2 * m2;
    m0 *= g0.z;
    m1 *= g1.z;
    m2 *= g2.z;
    float q0 = g0.x * x0x + g0.y * x0y;
    float q1 = g1.x * x12x + g1.y * x12y;
    float q2 = g2.x * x12z + g2.y * x12w;
    // rectify(130 n) = (1 + 130 n) / 2 = 0.5 + 65 n BIT FOR BIT: scaling by two commutes with rounding, so RN(130 n) / 2 ==
    // RN(65 n) and RN(1 + v) / 2 == RN(0.5 + v / 2) (where v is small enough for a halving to lose a denormal bit, both
    // forms return 0.5).  The caller adds the 0.5: one multiply and one add instead of two multiplies and an add.
    return 65.0f * (m0 * q0 + m1 * q1 + m2 * q2);
}
__device__ __forceinline__ float rectify_half(float half_v) { return 0.5f + half_v; }  // rectify(v), given v / 2

// ---- tolerance mode (NZ_FLOAT_FAST, nz_ctx_set_float_mode) ---------------------------------------------------------
// The reference compiles FractalJob with FloatMode.Fast (Noise/Fractal/Fractal.cs:19): its own results move with the
// compiler's contractions.  This form keeps every operation a DISCRETE decision or a cancellation depends on exactly as
// above -- the skew, both floors, the unskew (v - i + t cancels ~1e4 down to [0, 1)), the i1 select, mod289 and the table
// indices -- and contracts only the smooth polynomial tail: the corner falloffs m = 0.5 - x.x as two FMAs, the gradient
// dots and the corner sum as FMAs, the corner normalisation folded into the staged gradient table (g * norm, one rounding
// more per component), 130 * n -> rectify -> a * r folded into one FMA per octave-cell (t += (65 a) * n; the 0.5 a terms are
// wave-uniform and added once per cell).  Per octave-cell 63 VALU instructions instead of 83; every difference from the strict
// form is a rounding of relative size 2^-24 in a term of the octave, i.e. <= ~2e-7 relative on the fBm sum.
__device__ __forceinline__ float snoise2_tab_fast(float vx, float vy, const int *s_t1, const float2 *s_t2) {
    const float Cx = 0.211324865405187f, Cy = 0.366025403784439f, Cz = -0.577350269189626f;
    float s = vx * Cy + vy * Cy;
    float fx = floorf(vx + s), fy = floorf(vy + s);
    float t = fx * Cx + fy * Cx;
    float x0x = vx - fx + t, x0y = vy - fy + t;
    bool gt = x0x > x0y;
    // x12.xy = x0 + C.xx - i1: the constant pair selected instead of subtracted
    float x12x = x0x + (gt ? Cx - 1.0f : Cx), x12y = x0y + (gt ? Cx : Cx - 1.0f);
    float x12z = x0x + Cz, x12w = x0y + Cz;
    int ixi = (int)mod289i(fx), iyi = (int)mod289i(fy);
    int ix8 = ixi << 3;
    int k0 = s_t1[iyi], k2 = s_t1[iyi + 1];  // 8 * permute (staged halved)
    const char *t2 = reinterpret_cast<const char *>(s_t2);
    int a0 = k0 + ix8, a2 = k2 + ix8;
    int a1 = gt ? a0 + 8 : a2;
    float2 g0 = *reinterpret_cast<const float2 *>(t2 + a0);
    float2 g1 = *reinterpret_cast<const float2 *>(t2 + a1);
    float2 g2 = *reinterpret_cast<const float2 *>(t2 + (a2 + 8));
    float m0 = fmaxf(__builtin_fmaf(-x0y, x0y, __builtin_fmaf(-x0x, x0x, 0.5f)), 0.0f);
    float m1 = fmaxf(__builtin_fmaf(-x12y, x12y, __builtin_fmaf(-x12x, x12x, 0.5f)), 0.0f);
    float m2 = fmaxf(__builtin_fmaf(-x12w, x12w, __builtin_fmaf(-x12z, x12z, 0.5f)), 0.0f);
    m0 = m0 * m0; m1 = m1 * m1; m2 = m2 * m2;
    m0 = m0 * m0; m1 = m1 * m1; m2 = m2 * m2;
    float q0 = __builtin_fmaf(g0.x, x0x, g0.y * x0y);
    float q1 = __builtin_fmaf(g1.x, x12x, g1.y * x12y);
    float q2 = __builtin_fmaf(g2.x, x12z, g2.y * x12w);
    return __builtin_fmaf(m2, q2, __builtin_fmaf(m1, q1, m0 * q0));  // snoise / 130
}

// batched launch: grid blockIdx.y of the batch has its own world position and output plane
__device__ __forceinline__ void fractal_batch_enter(nz_fractal_params &p, float *__restrict__ &dst) {
    if (p.positions) {
        p.posx = (float)p.positions[2 * blockIdx.y];
        p.posz = (float)p.positions[2 * blockIdx.y + 1];
    }
    dst += blockIdx.y * p.bstride;
}


template <int VEC, bool FAST>
__global__ __launch_bounds__(256) void fractal_simplex_tab_kernel(float *__restrict__ dst, int rows, int cols, int pitch,
                                                                 int blocks_per_row, nz_fractal_params p,
                                                                 const int *__restrict__ t1g,
                                                                 const float4 *__restrict__ t2g) {
    __shared__ int s_t1[NZ_T1_N];
    __shared__ float4 s_t2[FAST ? 1 : NZ_T2_N];
    __shared__ float2 s_t2f[FAST ? NZ_T2_N : 1];  // tolerance mode: {a0, h} * norm, 8-byte entries (T1 staged as 8 * permute)
    if constexpr (FAST) {
        for (int i = threadIdx.x; i < NZ_T1_N; i += 256) s_t1[i] = t1g[i] >> 1;
        for (int i = threadIdx.x; i < NZ_T2_N; i += 256) {
            const float4 g = t2g[i];
            s_t2f[i] = make_float2(g.x * g.z, g.y * g.z);
        }
    } else {
        for (int i = threadIdx.x; i < NZ_T1_N; i += 256) s_t1[i] = t1g[i];
        for (int i = threadIdx.x; i < NZ_T2_N; i += 256) s_t2[i] = t2g[i];
    }
    __syncthreads();
    fractal_batch_enter(p, dst);
    int by = blockIdx.x / blocks_per_row;
    int bx = blockIdx.x - by * blocks_per_row;
    int x0 = (bx * 256 + threadIdx.x) * VEC;
    if (x0 >= cols) return;
    float xi[VEC];
#pragma unroll
    for (int c = 0; c < VEC; c++) xi[c] = ((float)(x0 + c) + p.posx) / p.noise_size;
    int zend = min(rows, (by + 1) * p.rows_per_wg);
    for (int z = by * p.rows_per_wg; z < zend; z++) {
        float zi = ((float)z + p.posz) / p.noise_size;
        float t[VEC];
#pragma unroll
        for (int c = 0; c < VEC; c++) t[c] = 0.0f;
        float detune = 0.0f, f = 1.0f, a = p.amp;
        float reach = fabsf(zi);
#pragma unroll
        for (int c = 0; c < VEC; c++) reach = fmaxf(reach, fabsf(xi[c]));
        if (FAST && p.fmax * reach < NZ_TAB_LIMIT) {
            float bias = 0.0f;  // the octaves' 0.5 a (wave-uniform)
            for (int i = 0; i < p.octaves; i++) {
                const float zV = f * zi, a65 = 65.0f * a;
#pragma unroll
                for (int c = 0; c < VEC; c++) t[c] = __builtin_fmaf(a65, snoise2_tab_fast(f * xi[c], zV, s_t1, s_t2f), t[c]);
                bias = __builtin_fmaf(0.5f, a, bias);
                detune += p.detune_rate;
                f *= (p.stepdown - detune);
                a *= p.G;
            }
#pragma unroll
            for (int c = 0; c < VEC; c++) t[c] += bias;
        } else if (p.fmax * reach < NZ_TAB_LIMIT) {  // every octave of this row stays inside the tables' range
            for (int i = 0; i < p.octaves; i++) {
                float zV = f * zi;
#pragma unroll
                for (int c = 0; c < VEC; c++) {
                    float xV = f * xi[c];
                    t[c] += a * rectify_half(snoise2_tab(xV, zV, s_t1, s_t2));
                }
                detune += p.detune_rate;
                f *= (p.stepdown - detune);
                a *= p.G;
            }
        } else {
            asm volatile("; guarded octave loop" ::: "memory");  // a real branch, never if-converted
            for (int i = 0; i < p.octaves; i++) {
                float zV = f * zi;
                float xV[VEC], big = fabsf(zV);
#pragma unroll
                for (int c = 0; c < VEC; c++) {
                    xV[c] = f * xi[c];
                    big = fmaxf(big, fabsf(xV[c]));
                }
                if (!FAST && big < NZ_TAB_LIMIT) {  // (tolerance mode: beyond the tables' range the strict direct form)
#pragma unroll
                    for (int c = 0; c < VEC; c++) t[c] += a * rectify_half(snoise2_tab(xV[c], zV, s_t1, s_t2));
                } else {
                    asm volatile("; direct evaluation" ::: "memory");
#pragma unroll
                    for (int c = 0; c < VEC; c++) t[c] += a * rectify(snoise2(xV[c], zV));
                }
                detune += p.detune_rate;
                f *= (p.stepdown - detune);
                a *= p.G;
            }
        }
        float *row = dst + (size_t)z * pitch;
        float o[VEC];
#pragma unroll
        for (int c = 0; c < VEC; c++) o[c] = t[c] / p.norm;
        bool full = x0 + VEC <= cols && ((reinterpret_cast<uintptr_t>(row + x0) & (VEC * 4 - 1)) == 0);
        if (full && VEC == 4) {
            *reinterpret_cast<float4 *>(row + x0) = make_float4(o[0], o[1 % VEC], o[2 % VEC], o[3 % VEC]);
        } else if (full && VEC == 2) {
            *reinterpret_cast<float2 *>(row + x0) = make_float2(o[0], o[VEC - 1]);
        } else {
#pragma unroll
            for (int c = 0; c < VEC; c++)
                if (x0 + c < cols) row[x0 + c] = o[c];
        }
    }
}

// ---- Perlin and cellular fBm with the same table technique ---------------------------------------------
// cnoise: corner gradient (gx, gy) * taylorInvSqrt(gx^2 + gy^2) is a function of j = permute(ix) + iy;
// cellular: the feature-point offset (ox, oy) is a function of a = permute(Pi.x + oi) + Pi.y + oi.
// Tables (host-built with the reference's operation sequence, so bit-identical):
//   P1[i] = 8*permute(i), i in [0,289];          P2[j] = {gx*norm, gy*norm} of permute(j), j in [0,577]
//   C1[i+1] = 8*permute(i), i in [-1,289];        C2[a+1] = {ox, oy} of permute(a), a in [-1,578]
constexpr int NZ_TB1_N = 292, NZ_TB2_N = 584;

__device__ __forceinline__ float cnoise2_tab(float Px, float Py, const int *s_t1, const float2 *s_t2) {
    float flx = floorf(Px), fly = floorf(Py);
    float frx = Px - flx, fry = Py - fly;
    int ix0 = (int)mod289i(flx + 0.0f), iy0 = (int)mod289i(fly + 0.0f);
    int ix1 = (int)mod289i(flx + 1.0f), iy1 = (int)mod289i(fly + 1.0f);
    ix0 = min(max(ix0, 0), 289); iy0 = min(max(iy0, 0), 289);  // 289 is a legal value of fp32 mod289 (negative cells)
    ix1 = min(max(ix1, 0), 289); iy1 = min(max(iy1, 0), 289);
    float Pf0 = frx, Pf1 = fry, Pf2 = frx - 1.0f, Pf3 = fry - 1.0f;
    int a0 = s_t1[ix0], a1 = s_t1[ix1];
    const char *t2 = reinterpret_cast<const char *>(s_t2);
    float2 g00 = *reinterpret_cast<const float2 *>(t2 + (a0 + 8 * iy0));
    float2 g10 = *reinterpret_cast<const float2 *>(t2 + (a1 + 8 * iy0));
    float2 g01 = *reinterpret_cast<const float2 *>(t2 + (a0 + 8 * iy1));
    float2 g11 = *reinterpret_cast<const float2 *>(t2 + (a1 + 8 * iy1));
    float n00 = g00.x * Pf0 + g00.y * Pf1;
    float n10 = g10.x * Pf2 + g10.y * Pf1;
    float n01 = g01.x * Pf0 + g01.y * Pf3;
    float n11 = g11.x * Pf2 + g11.y * Pf3;
    float fdx = fadef(Pf0), fdy = fadef(Pf1);
    float nx0 = lerpf_(n00, n10, fdx);
    float nx1 = lerpf_(n01, n11, fdx);
    return 2.3f * lerpf_(nx0, nx1, fdy);
}

__device__ __forceinline__ float cellular_rect_tab(float Px, float Py, const int *s_t1, const float2 *s_t2) {
    float fx = floorf(Px), fy = floorf(Py);
    int Pix = (int)mod289i(fx), Piy = (int)mod289i(fy);
    Pix = min(max(Pix, 0), 289);  // 289 is a legal value of fp32 mod289 (negative cells)
    Piy = min(max(Piy, 0), 289);
    float Pfx = Px - fx, Pfy = Py - fy;
    const char *t2 = reinterpret_cast<const char *>(s_t2);
    const float xoff[3] = {0.5f, -0.5f, -1.5f};
    const float of[3] = {-0.5f, 0.5f, 1.5f};
    float d[3][3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        int px8 = s_t1[Pix + c];  // 8 * permute(Pi.x + oi[c]), table index shifted by one
        float bx = Pfx + xoff[c];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float2 o = *reinterpret_cast<const float2 *>(t2 + (px8 + 8 * (Piy + k)));
            float dx = bx + o.x;
            float dy = Pfy - of[k] + o.y;
            d[c][k] = dx * dx + dy * dy;
        }
    }
    float d1[3], d2[3], d1a[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        d1a[k] = fminf(d[0][k], d[1][k]);
        d2[k] = fmaxf(d[0][k], d[1][k]);
        d2[k] = fminf(d2[k], d[2][k]);
        d1[k] = fminf(d1a[k], d2[k]);
        d2[k] = fmaxf(d1a[k], d2[k]);
    }
    if (!(d1[0] < d1[1])) { float t = d1[0]; d1[0] = d1[1]; d1[1] = t; }
    if (!(d1[0] < d1[2])) { float t = d1[0]; d1[0] = d1[2]; d1[2] = t; }
    d1[1] = fminf(d1[1], d2[1]);
    d1[2] = fminf(d1[2], d2[2]);
    d1[1] = fminf(d1[1], d1[2]);
    d1[1] = fminf(d1[1], d2[0]);
    float F1 = sqrtf(d1[0]), F2 = sqrtf(d1[1]);
    return rectify(F1) * rectify(F2);
}

template <int BASIS, int VEC>
__global__ __launch_bounds__(256) void fractal_tab2_kernel(float *__restrict__ dst, int rows, int cols, int pitch,
                                                          int blocks_per_row, nz_fractal_params p,
                                                          const int *__restrict__ t1g, const float2 *__restrict__ t2g) {
    __shared__ int s_t1[NZ_TB1_N];
    __shared__ float2 s_t2[NZ_TB2_N];
    for (int i = threadIdx.x; i < NZ_TB1_N; i += 256) s_t1[i] = t1g[i];
    for (int i = threadIdx.x; i < NZ_TB2_N; i += 256) s_t2[i] = t2g[i];
    __syncthreads();
    fractal_batch_enter(p, dst);
    int by = blockIdx.x / blocks_per_row;
    int bx = blockIdx.x - by * blocks_per_row;
    int x0 = (bx * 256 + threadIdx.x) * VEC;
    if (x0 >= cols) return;
    float xi[VEC];
#pragma unroll
    for (int c = 0; c < VEC; c++) xi[c] = ((float)(x0 + c) + p.posx) / p.noise_size;
    int zend = min(rows, (by + 1) * p.rows_per_wg);
    for (int z = by * p.rows_per_wg; z < zend; z++) {
        float zi = ((float)z + p.posz) / p.noise_size;
        float t[VEC];
#pragma unroll
        for (int c = 0; c < VEC; c++) t[c] = 0.0f;
        float detune = 0.0f, f = 1.0f, a = p.amp;
        float reach = fabsf(zi);
#pragma unroll
        for (int c = 0; c < VEC; c++) reach = fmaxf(reach, fabsf(xi[c]));
        if (p.fmax * reach < NZ_TAB_LIMIT) {  // every octave of this row stays inside the tables' range
            for (int i = 0; i < p.octaves; i++) {
                float zV = f * zi;
#pragma unroll
                for (int c = 0; c < VEC; c++) {
                    float xV = f * xi[c];
                    t[c] += a * (BASIS == NZ_NOISE_PERLIN ? rectify(cnoise2_tab(xV, zV, s_t1, s_t2))
                                                          : cellular_rect_tab(xV, zV, s_t1, s_t2));
                }
                detune += p.detune_rate;
                f *= (p.stepdown - detune);
                a *= p.G;
            }
        } else {
            asm volatile("; guarded octave loop" ::: "memory");  // a real branch, never if-converted
            for (int i = 0; i < p.octaves; i++) {
                float zV = f * zi;
#pragma unroll
                for (int c = 0; c < VEC; c++) {
                    float xV = f * xi[c];
                    float nv;
                    if (fmaxf(fabsf(xV), fabsf(zV)) < NZ_TAB_LIMIT) {
                        nv = BASIS == NZ_NOISE_PERLIN ? rectify(cnoise2_tab(xV, zV, s_t1, s_t2))
                                                      : cellular_rect_tab(xV, zV, s_t1, s_t2);
                    } else {
                        asm volatile("; direct evaluation" ::: "memory");
                        nv = BASIS == NZ_NOISE_PERLIN ? rectify(cnoise2(xV, zV)) : cellular_rect(xV, zV);
                    }
                    t[c] += a * nv;
                }
                detune += p.detune_rate;
                f *= (p.stepdown - detune);
                a *= p.G;
            }
        }
        float *row = dst + (size_t)z * pitch;
#pragma unroll
        for (int c = 0; c < VEC; c++)
            if (x0 + c < cols) row[x0 + c] = t[c] / p.norm;
    }
}

// ---- domain-rotated 3-D bases with tabulated hashes and gradients -------------------------------------
// noise.cnoise(float3) / noise.snoise(float3) hash a corner with three nested permutes of small integers
// and decode a gradient from the result: P3[j] = permute(j), j in [0,579] (permute only depends on j mod 289
// and every product stays below 2^24), G3[h] = the normalised gradient of hash h, both built by the host
// with the reference's operation sequence.  A corner costs three 4-byte and one 16-byte LDS read.
constexpr int NZ_P3_N = 580, NZ_G3_N = 292;

// mod289 of an integer-valued float as a table index.  In fp32 the result is 289, not 0, for some negative
// multiples of 289 (-8959, -17629, ...); the tables are built over that range.
__device__ __forceinline__ int lattice289(float f) { return min(max((int)mod289i(f), 0), 289); }

__device__ __forceinline__ float cnoise3_tab(float Px, float Py, float Pz, const int *s_p, const float4 *s_g) {
    float fl[3] = {floorf(Px), floorf(Py), floorf(Pz)};
    float Pf0[3] = {Px - fl[0], Py - fl[1], Pz - fl[2]};
    float Pf1[3] = {Pf0[0] - 1.0f, Pf0[1] - 1.0f, Pf0[2] - 1.0f};
    int i0[3], i1[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        i0[k] = lattice289(fl[k]);
        i1[k] = lattice289(fl[k] + 1.0f);
    }
    int a0 = s_p[i0[0]], a1 = s_p[i1[0]];
    int b[4] = {s_p[a0 + i0[1]], s_p[a1 + i0[1]], s_p[a0 + i1[1]], s_p[a1 + i1[1]]};
    float n[2][4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
#pragma unroll
        for (int z = 0; z < 2; z++) {
            float4 g = s_g[s_p[b[k] + (z ? i1[2] : i0[2])]];
            asm volatile("" ::"v"(g.w));  // keep the 16-byte read (a 12-byte one costs twice the LDS cycles)
            float fx = (k & 1) ? Pf1[0] : Pf0[0];
            float fy = (k & 2) ? Pf1[1] : Pf0[1];
            float fz = z ? Pf1[2] : Pf0[2];
            n[z][k] = g.x * fx + g.y * fy + g.z * fz;
        }
    }
    float fdx = fadef(Pf0[0]), fdy = fadef(Pf0[1]), fdz = fadef(Pf0[2]);
    float nz0 = lerpf_(n[0][0], n[1][0], fdz);
    float nz1 = lerpf_(n[0][1], n[1][1], fdz);
    float nz2 = lerpf_(n[0][2], n[1][2], fdz);
    float nz3 = lerpf_(n[0][3], n[1][3], fdz);
    float nyz0 = lerpf_(nz0, nz2, fdy);
    float nyz1 = lerpf_(nz1, nz3, fdy);
    return 2.2f * lerpf_(nyz0, nyz1, fdx);
}

__device__ __forceinline__ float snoise3_tab(float vx, float vy, float vz, const int *s_p, const float4 *s_g) {
    const float Cx = 1.0f / 6.0f, Cy = 1.0f / 3.0f;
    float v[3] = {vx, vy, vz};
    float s = vx * Cy + vy * Cy + vz * Cy;
    float i[3], x0[3];
#pragma unroll
    for (int k = 0; k < 3; k++) i[k] = floorf(v[k] + s);
    float t = i[0] * Cx + i[1] * Cx + i[2] * Cx;
#pragma unroll
    for (int k = 0; k < 3; k++) x0[k] = v[k] - i[k] + t;
    float g[3] = {stepf_(x0[1], x0[0]), stepf_(x0[2], x0[1]), stepf_(x0[0], x0[2])};
    float l[3] = {1.0f - g[0], 1.0f - g[1], 1.0f - g[2]};
    float lz[3] = {l[2], l[0], l[1]};
    float i1[3], i2[3], xs[4][3];
    int ii[3], o1[3], o2[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        i1[k] = fminf(g[k], lz[k]);
        i2[k] = fmaxf(g[k], lz[k]);
        xs[0][k] = x0[k];
        xs[1][k] = x0[k] - i1[k] + Cx;
        xs[2][k] = x0[k] - i2[k] + Cy;
        xs[3][k] = x0[k] - 0.5f;
        ii[k] = lattice289(i[k]);
        o1[k] = (int)i1[k];
        o2[k] = (int)i2[k];
    }
    const int oz[4] = {0, o1[2], o2[2], 1}, oy[4] = {0, o1[1], o2[1], 1}, ox[4] = {0, o1[0], o2[0], 1};
    float acc[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int h = s_p[s_p[s_p[ii[2] + oz[k]] + ii[1] + oy[k]] + ii[0] + ox[k]];
        float4 p = s_g[h];
        asm volatile("" ::"v"(p.w));
        float m = fmaxf(0.6f - (xs[k][0] * xs[k][0] + xs[k][1] * xs[k][1] + xs[k][2] * xs[k][2]), 0.0f);
        m = m * m;
        float pd = p.x * xs[k][0] + p.y * xs[k][1] + p.z * xs[k][2];
        acc[k] = (m * m) * pd;
    }
    return 42.0f * (acc[0] + acc[1] + acc[2] + acc[3]);
}

template <int BASIS, int VEC>
__global__ __launch_bounds__(256) void fractal_tab3_kernel(float *__restrict__ dst, int rows, int cols, int pitch,
                                                          int blocks_per_row, nz_fractal_params p,
                                                          const int *__restrict__ p3g, const float4 *__restrict__ g3g) {
    __shared__ int s_p[NZ_P3_N];
    __shared__ float4 s_g[NZ_G3_N];
    for (int i = threadIdx.x; i < NZ_P3_N; i += 256) s_p[i] = p3g[i];
    for (int i = threadIdx.x; i < NZ_G3_N; i += 256) s_g[i] = g3g[i];
    __syncthreads();
    fractal_batch_enter(p, dst);
    int by = blockIdx.x / blocks_per_row;
    int bx = blockIdx.x - by * blocks_per_row;
    int x0 = (bx * 256 + threadIdx.x) * VEC;  // VEC consecutive cells per thread: independent chains to interleave
    if (x0 >= cols) return;
    float xi[VEC];
#pragma unroll
    for (int c = 0; c < VEC; c++) xi[c] = ((float)(x0 + c) + p.posx) / p.noise_size;
    int zend = min(rows, (by + 1) * p.rows_per_wg);
    for (int z = by * p.rows_per_wg; z < zend; z++) {
        float zi = ((float)z + p.posz) / p.noise_size;
        float t[VEC], detune = 0.0f, f = 1.0f, a = p.amp;
        float reach = fabsf(zi);
#pragma unroll
        for (int c = 0; c < VEC; c++) {
            t[c] = 0.0f;
            reach = fmaxf(reach, fabsf(xi[c]));
        }
        // the rotation shrinks |x|, |z| (factor <= 1.16 on x + z), so the 2-D limit keeps the lattice in range
        const bool small_row = p.fmax * reach < NZ_TAB_LIMIT;
        for (int i = 0; i < p.octaves; i++) {
            float zV = f * zi;
#pragma unroll
            for (int c = 0; c < VEC; c++) {
                float xV = f * xi[c];
                float xr, zr, yr;
                domain_rotate(xV, zV, xr, zr, yr);
                float nv;
                if (small_row || fmaxf(fabsf(xV), fabsf(zV)) < NZ_TAB_LIMIT) {
                    nv = BASIS == NZ_NOISE_DOMAIN_ROTATED_PERLIN ? cnoise3_tab(xr, zr, yr, s_p, s_g)
                                                                 : snoise3_tab(xr, zr, yr, s_p, s_g);
                } else {
                    asm volatile("; direct evaluation" ::: "memory");  // a real branch, never if-converted
                    nv = BASIS == NZ_NOISE_DOMAIN_ROTATED_PERLIN ? cnoise3(xr, zr, yr) : snoise3(xr, zr, yr);
                }
                t[c] += a * rectify(nv);
            }
            detune += p.detune_rate;
            f *= (p.stepdown - detune);
            a *= p.G;
        }
#pragma unroll
        for (int c = 0; c < VEC; c++)
            if (x0 + c < cols) dst[(size_t)z * pitch + x0 + c] = t[c] / p.norm;
    }
}

constexpr int FR_THREADS = 256;

// FractalGenerator.NoiseValue (Fractal.cs:114-131) for VEC consecutive cells of one row.
template <int BASIS, int VEC>
__global__ __launch_bounds__(FR_THREADS) void fractal_kernel(float *__restrict__ dst, int rows, int cols,
                                                            int pitch, int blocks_per_row,
                                                            nz_fractal_params p,
                                                            const float2 *__restrict__ rgrad) {
    constexpr bool USES_TAB = BASIS == NZ_NOISE_PERIODIC_PERLIN || BASIS == NZ_NOISE_ROTATED_SIMPLEX;
    __shared__ float2 s_tab[USES_TAB ? NZ_PSR_T2 : 1];
    __shared__ int s_t1[USES_TAB ? NZ_PSR_T1 : 1];
    if constexpr (USES_TAB) {
        const int *t1g = reinterpret_cast<const int *>(rgrad);
        const float2 *src = reinterpret_cast<const float2 *>(t1g + NZ_PSR_T1) +
                            (BASIS == NZ_NOISE_ROTATED_SIMPLEX ? NZ_PSR_T2 : 0);
        for (int i = threadIdx.x; i < NZ_PSR_T1; i += FR_THREADS) s_t1[i] = t1g[i];
        for (int i = threadIdx.x; i < NZ_PSR_T2; i += FR_THREADS) s_tab[i] = src[i];
        __syncthreads();
    }
    const psr_tables tabs{s_t1, s_tab};
    fractal_batch_enter(p, dst);
    int by = blockIdx.x / blocks_per_row;
    int bx = blockIdx.x - by * blocks_per_row;
    int x0 = (bx * FR_THREADS + threadIdx.x) * VEC;
    if (x0 >= cols) return;
    float xi[VEC];
#pragma unroll
    for (int c = 0; c < VEC; c++) xi[c] = ((float)(x0 + c) + p.posx) / p.noise_size;
    // |xi| over the workgroup's columns (xi is monotonic in the column: the ends bound it)
    const float xi_first = ((float)(bx * FR_THREADS * VEC) + p.posx) / p.noise_size;
    const float xi_last = ((float)(min((bx + 1) * FR_THREADS * VEC, cols) - 1) + p.posx) / p.noise_size;
    const float xreach = fmaxf(fabsf(xi_first), fabsf(xi_last));
    // p.rows_per_wg rows per workgroup: 8 amortise the table fill of the periodic bases on big grids
    int zend = min(rows, (by + 1) * p.rows_per_wg);
    for (int z = by * p.rows_per_wg; z < zend; z++) {
        float zi = ((float)z + p.posz) / p.noise_size;
        float t[VEC];
#pragma unroll
        for (int c = 0; c < VEC; c++) t[c] = 0.0f;
        float detune = 0.0f, f = 1.0f, a = p.amp;
        for (int i = 0; i < p.octaves; i++) {
            float zV = f * zi;
            if constexpr (USES_TAB) {
                // which coordinates of this octave can reach psrnoise's periods anywhere in the workgroup's cells (uniform:
                // the bounds come from the workgroup's first and last column and the row)
                const float zr = fabsf(zV) + 0.001f, xr = fabsf(f) * xreach;
                const int nowrap = __builtin_amdgcn_readfirstlane((xr + zr + 3.5f < 1000.0f ? 1 : 0) | (zr + 3.0f < 100.0f ? 2 : 0));
                if (nowrap == 3) {
#pragma unroll
                    for (int c = 0; c < VEC; c++) t[c] += a * rectify(psrnoise2<false, false>(f * xi[c], zV, tabs));
                } else if (nowrap == 1) {
#pragma unroll
                    for (int c = 0; c < VEC; c++) t[c] += a * rectify(psrnoise2<false, true>(f * xi[c], zV, tabs));
                } else {
#pragma unroll
                    for (int c = 0; c < VEC; c++) t[c] += a * noise_value<BASIS>(f * xi[c], zV, tabs);
                }
            } else {
#pragma unroll
                for (int c = 0; c < VEC; c++) {
                    float xV = f * xi[c];
                    t[c] += a * noise_value<BASIS>(xV, zV, tabs);
                }
            }
            detune += p.detune_rate;
            f *= (p.stepdown - detune);
            a *= p.G;
        }
        float *row = dst + (size_t)z * pitch;
        float o[VEC];
#pragma unroll
        for (int c = 0; c < VEC; c++) o[c] = t[c] / p.norm;
        bool full = x0 + VEC <= cols && ((reinterpret_cast<uintptr_t>(row + x0) & (VEC * 4 - 1)) == 0);
        if (full) {
            if constexpr (VEC == 4) {
                *reinterpret_cast<float4 *>(row + x0) = make_float4(o[0], o[1], o[2], o[3]);
            } else if constexpr (VEC == 2) {
                *reinterpret_cast<float2 *>(row + x0) = make_float2(o[0], o[1]);
            } else {
                row[x0] = o[0];
            }
        } else {
#pragma unroll
            for (int c = 0; c < VEC; c++)
                if (x0 + c < cols) row[x0 + c] = o[c];
        }
    }
}

template <int BASIS, int VEC>
int32_t launch_basis(hipStream_t s, float *dst, int rows, int cols, int pitch, const nz_fractal_params &p,
                     const float *d_rgrad, int count) {
    int per_block = FR_THREADS * VEC;
    int bpr = (cols + per_block - 1) / per_block;
    long long blocks = (long long)bpr * ((rows + p.rows_per_wg - 1) / p.rows_per_wg);
    if (blocks > 0x7fffffffLL) {
        nz_set_error("fractal grid too large");
        return NZ_ERR_INVALID;
    }
    NZ_LAUNCH((fractal_kernel<BASIS, VEC>), dim3((unsigned)blocks, count), dim3(FR_THREADS), 0, s, dst, rows,
                       cols, pitch, bpr, p, reinterpret_cast<const float2 *>(d_rgrad));
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

}  // namespace

int32_t nz_launch_fractal(hipStream_t s, int noiseType, float *dst, int rows, int cols, int pitch,
                          const nz_fractal_params &p_in, const float *d_rgrad, const void *d_simplex, int count,
                          size_t bstride, const int32_t *positions) {
    if (count < 1) return NZ_OK;
    nz_fractal_params p = p_in;
    p.positions = positions;
    p.bstride = count > 1 || positions ? bstride : 0;
    // 4 rows per workgroup amortise the table staging on big grids (1 / 2 / 4 / 8 / 16 rows: 0.285 / 0.278 / 0.277 /
    // 0.281 / 0.292 ms for the metric's 4096^2 simplex plane); a small grid gets more, shorter workgroups (the launch
    // lasts as long as one workgroup's rows)
    p.rows_per_wg = 4;
    {
        long long wg_per_row = (cols + 511) / 512;
        while (p.rows_per_wg > 1 && wg_per_row * ((rows + p.rows_per_wg - 1) / p.rows_per_wg) * count < 2048) p.rows_per_wg >>= 1;
    }
    constexpr int use_tab = 1;  // (the direct kernels below serve Sin / psrnoise and every basis beyond its tables' range)
    if (noiseType == NZ_NOISE_SIMPLEX && use_tab && d_simplex) {
#ifndef NZ_FT_VEC
#define NZ_FT_VEC 2
#endif
        constexpr int VEC = NZ_FT_VEC;
        int bpr = (cols + 256 * VEC - 1) / (256 * VEC);
        long long blocks = (long long)bpr * ((rows + p.rows_per_wg - 1) / p.rows_per_wg);
        const int *t1 = reinterpret_cast<const int *>(d_simplex);
        const float4 *t2 = reinterpret_cast<const float4 *>(t1 + NZ_T1_N);
        if (nz_tls_float_mode >= NZ_FLOAT_FAST)
            NZ_LAUNCH((fractal_simplex_tab_kernel<VEC, true>), dim3((unsigned)blocks, count), dim3(256), 0, s, dst, rows, cols,
                      pitch, bpr, p, t1, t2);
        else
            NZ_LAUNCH((fractal_simplex_tab_kernel<VEC, false>), dim3((unsigned)blocks, count), dim3(256), 0, s, dst, rows, cols,
                      pitch, bpr, p, t1, t2);
        NZ_HIP(hipGetLastError());
        return NZ_OK;
    }
    if ((noiseType == NZ_NOISE_PERLIN || noiseType == NZ_NOISE_CELLULAR) && use_tab && d_simplex) {
        // table block layout: see build_simplex_tables() in nz_runtime.cpp
        const char *base = reinterpret_cast<const char *>(d_simplex) + (NZ_T1_N * 4 + NZ_T2_N * 16);
        if (noiseType == NZ_NOISE_CELLULAR) base += NZ_TB1_N * 4 + NZ_TB2_N * 8;
        const int *t1 = reinterpret_cast<const int *>(base);
        const float2 *t2 = reinterpret_cast<const float2 *>(base + NZ_TB1_N * 4);
#ifndef NZ_TAB2_VEC_PERLIN
#define NZ_TAB2_VEC_PERLIN 4  // cells per thread: Perlin 0.276 / 0.234 / 0.223 ms with 1 / 2 / 4 (4096^2, 13 octaves); cellular is indifferent
#endif
#ifndef NZ_TAB2_VEC_CELLULAR
#define NZ_TAB2_VEC_CELLULAR 4  // 0.510 -> 0.494 ms
#endif
        if (noiseType == NZ_NOISE_PERLIN) {
            constexpr int V = NZ_TAB2_VEC_PERLIN;
            int bpr = (cols + 256 * V - 1) / (256 * V);
            long long blocks = (long long)bpr * ((rows + p.rows_per_wg - 1) / p.rows_per_wg);
            NZ_LAUNCH((fractal_tab2_kernel<NZ_NOISE_PERLIN, V>), dim3((unsigned)blocks, count), dim3(256), 0, s, dst, rows,
                               cols, pitch, bpr, p, t1, t2);
        } else {
            constexpr int V = NZ_TAB2_VEC_CELLULAR;
            int bpr = (cols + 256 * V - 1) / (256 * V);
            long long blocks = (long long)bpr * ((rows + p.rows_per_wg - 1) / p.rows_per_wg);
            NZ_LAUNCH((fractal_tab2_kernel<NZ_NOISE_CELLULAR, V>), dim3((unsigned)blocks, count), dim3(256), 0, s, dst,
                               rows, cols, pitch, bpr, p, t1, t2);
        }
        NZ_HIP(hipGetLastError());
        return NZ_OK;
    }
    if ((noiseType == NZ_NOISE_DOMAIN_ROTATED_PERLIN || noiseType == NZ_NOISE_DOMAIN_ROTATED_SIMPLEX) && use_tab &&
        d_simplex) {
        const char *base = reinterpret_cast<const char *>(d_simplex) + (NZ_T1_N * 4 + NZ_T2_N * 16) +
                           2 * (NZ_TB1_N * 4 + NZ_TB2_N * 8);
        const int *p3 = reinterpret_cast<const int *>(base);
        const float4 *g3 = reinterpret_cast<const float4 *>(base + NZ_P3_N * 4);
#ifndef NZ_TAB3_VEC
#define NZ_TAB3_VEC 1
#endif
        constexpr int V3 = NZ_TAB3_VEC;  // cells per thread
        int bpr = (cols + 256 * V3 - 1) / (256 * V3);
        long long blocks = (long long)bpr * ((rows + p.rows_per_wg - 1) / p.rows_per_wg);
        if (noiseType == NZ_NOISE_DOMAIN_ROTATED_PERLIN)
            NZ_LAUNCH((fractal_tab3_kernel<NZ_NOISE_DOMAIN_ROTATED_PERLIN, V3>), dim3((unsigned)blocks, count), dim3(256), 0,
                               s, dst, rows, cols, pitch, bpr, p, p3, g3);
        else
            NZ_LAUNCH((fractal_tab3_kernel<NZ_NOISE_DOMAIN_ROTATED_SIMPLEX, V3>), dim3((unsigned)blocks, count), dim3(256), 0,
                               s, dst, rows, cols, pitch, bpr, p, p3, g3 + NZ_G3_N);
        NZ_HIP(hipGetLastError());
        return NZ_OK;
    }
    switch (noiseType) {
        case NZ_NOISE_SIN: return launch_basis<NZ_NOISE_SIN, 4>(s, dst, rows, cols, pitch, p, d_rgrad, count);
        case NZ_NOISE_PERLIN: return launch_basis<NZ_NOISE_PERLIN, 2>(s, dst, rows, cols, pitch, p, d_rgrad, count);
#ifndef NZ_PSR_VEC
#define NZ_PSR_VEC 4  // cells per thread: 0.492 / 0.518 / 0.464 ms with 1 / 2 / 4 (4096^2, 13 octaves)
#endif
        case NZ_NOISE_PERIODIC_PERLIN:
            return launch_basis<NZ_NOISE_PERIODIC_PERLIN, NZ_PSR_VEC>(s, dst, rows, cols, pitch, p, d_rgrad, count);
#ifndef NZ_FR_VEC
#define NZ_FR_VEC 2
#endif
        case NZ_NOISE_SIMPLEX: return launch_basis<NZ_NOISE_SIMPLEX, NZ_FR_VEC>(s, dst, rows, cols, pitch, p, d_rgrad, count);
        case NZ_NOISE_ROTATED_SIMPLEX:
            return launch_basis<NZ_NOISE_ROTATED_SIMPLEX, NZ_PSR_VEC>(s, dst, rows, cols, pitch, p, d_rgrad, count);
        case NZ_NOISE_CELLULAR: return launch_basis<NZ_NOISE_CELLULAR, 2>(s, dst, rows, cols, pitch, p, d_rgrad, count);
        case NZ_NOISE_DOMAIN_ROTATED_PERLIN:
            return launch_basis<NZ_NOISE_DOMAIN_ROTATED_PERLIN, 1>(s, dst, rows, cols, pitch, p, d_rgrad, count);
        case NZ_NOISE_DOMAIN_ROTATED_SIMPLEX:
            return launch_basis<NZ_NOISE_DOMAIN_ROTATED_SIMPLEX, 1>(s, dst, rows, cols, pitch, p, d_rgrad, count);
    }
    nz_set_error("unknown noise type %d", noiseType);
    return NZ_ERR_INVALID;
}
