/*
 * noize_oracle.c -- CPU restatement of noize-job's per-cell terrain hot path.
 * TEST INFRASTRUCTURE ONLY; parity pinned at image level only (the reference's screenshots), numerically unpinned -- see
 * noize_oracle.h for the full statement.
 *
 * Each function cites the reference file:line it follows (paths relative to
 * /root/reference).  Pass structure is the reference's: row-parallel pass into `tmp`, then
 * the single-threaded FlushWriteSlice copy tmp -> src (Pipeline/Tiles/TileData.cs:15-40).
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math -fopenmp (see Makefile).
 */
#include "noize_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 0;

void nzo_set_threads(int n) {
    g_threads = n;
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#endif
}

int nzo_get_threads(void) {
#ifdef _OPENMP
    return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------
 * L0 tile contract: Pipeline/Tiles/TileData.cs
 * ---------------------------------------------------------------------------------------- */

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* RWTileData.getIdx / ReadTileData.getIdx, TileData.cs:72-77,106-111: clamp both axes */
static inline int tile_idx(int x, int z, int rows, int cols) {
    x = clampi(x, 0, cols - 1);
    z = clampi(z, 0, rows - 1);
    return z * cols + x;
}

/* FlushWriteSlice.Execute, TileData.cs:37-39: write.CopyFrom(read), one thread */
static void flush_write_slice(float *write, const float *read, size_t n) {
    memcpy(write, read, n * sizeof(float));
}

/* ------------------------------------------------------------------------------------------
 * Unity.Mathematics helpers (math.cs), component-wise fp32
 * ---------------------------------------------------------------------------------------- */

static inline float fracf_(float x) { return x - floorf(x); }              /* math.frac */
static inline float lerpf_(float a, float b, float s) { return a + s * (b - a); } /* math.lerp */
static inline float stepf_(float y, float x) { return x >= y ? 1.0f : 0.0f; }     /* math.step(y,x) */
/* Unity.Mathematics math.min / math.max (1.2.x): `float.IsNaN(y) || x < y ? x : y` -- a NaN operand loses,
 * like C fminf / fmaxf and the GPU's v_min_f32 / v_max_f32 */
static inline float minf_(float a, float b) { return (b != b) || a < b ? a : b; }
static inline float maxf_(float a, float b) { return (b != b) || a > b ? a : b; }

/* noise/common.cs */
static inline float mod289f(float x) { return x - floorf(x * (1.0f / 289.0f)) * 289.0f; }
static inline float mod7f(float x) { return x - floorf(x * (1.0f / 7.0f)) * 7.0f; }
static inline float permutef(float x) { return mod289f((34.0f * x + 1.0f) * x); }
static inline float taylor_inv_sqrt(float r) { return 1.79284291400159f - 0.85373472095314f * r; }
static inline float fadef(float t) { return t * t * t * (t * (t * 6.0f - 15.0f) + 10.0f); }

/* ------------------------------------------------------------------------------------------
 * A.2 noise.cnoise(float2) -- classic Perlin
 * ---------------------------------------------------------------------------------------- */
float nzo_cnoise2(float Px, float Py) {
    /* Pi = floor(P.xyxy) + (0,0,1,1); Pf = frac(P.xyxy) - (0,0,1,1) */
    float Pi[4] = {floorf(Px) + 0.0f, floorf(Py) + 0.0f, floorf(Px) + 1.0f, floorf(Py) + 1.0f};
    float Pf[4] = {fracf_(Px) - 0.0f, fracf_(Py) - 0.0f, fracf_(Px) - 1.0f, fracf_(Py) - 1.0f};
    for (int k = 0; k < 4; k++) Pi[k] = mod289f(Pi[k]);
    float ix[4] = {Pi[0], Pi[2], Pi[0], Pi[2]}; /* Pi.xzxz */
    float iy[4] = {Pi[1], Pi[1], Pi[3], Pi[3]}; /* Pi.yyww */
    float fx[4] = {Pf[0], Pf[2], Pf[0], Pf[2]};
    float fy[4] = {Pf[1], Pf[1], Pf[3], Pf[3]};
    float gx[4], gy[4];
    for (int k = 0; k < 4; k++) {
        float i = permutef(permutef(ix[k]) + iy[k]);
        float g = fracf_(i * (1.0f / 41.0f)) * 2.0f - 1.0f;
        gy[k] = fabsf(g) - 0.5f;
        float tx = floorf(g + 0.5f);
        gx[k] = g - tx;
    }
    /* g00=(gx.x,gy.x) g10=(gx.y,gy.y) g01=(gx.z,gy.z) g11=(gx.w,gy.w) */
    float g00x = gx[0], g00y = gy[0], g10x = gx[1], g10y = gy[1];
    float g01x = gx[2], g01y = gy[2], g11x = gx[3], g11y = gy[3];
    /* norm = taylorInvSqrt(dot(g00,g00), dot(g01,g01), dot(g10,g10), dot(g11,g11)) */
    float n0 = taylor_inv_sqrt(g00x * g00x + g00y * g00y);
    float n1 = taylor_inv_sqrt(g01x * g01x + g01y * g01y);
    float n2 = taylor_inv_sqrt(g10x * g10x + g10y * g10y);
    float n3 = taylor_inv_sqrt(g11x * g11x + g11y * g11y);
    g00x *= n0; g00y *= n0;
    g01x *= n1; g01y *= n1;
    g10x *= n2; g10y *= n2;
    g11x *= n3; g11y *= n3;
    float n00 = g00x * fx[0] + g00y * fy[0];
    float n10 = g10x * fx[1] + g10y * fy[1];
    float n01 = g01x * fx[2] + g01y * fy[2];
    float n11 = g11x * fx[3] + g11y * fy[3];
    float fdx = fadef(Pf[0]), fdy = fadef(Pf[1]);
    /* n_x = lerp((n00,n01),(n10,n11), fade.x) */
    float nx0 = lerpf_(n00, n10, fdx);
    float nx1 = lerpf_(n01, n11, fdx);
    float nxy = lerpf_(nx0, nx1, fdy);
    return 2.3f * nxy;
}

/* ------------------------------------------------------------------------------------------
 * A.3 noise.snoise(float2) -- simplex
 * ---------------------------------------------------------------------------------------- */
float nzo_snoise2(float vx, float vy) {
    const float Cx = 0.211324865405187f, Cy = 0.366025403784439f;
    const float Cz = -0.577350269189626f, Cw = 0.024390243902439f;
    /* i = floor(v + dot(v, C.yy)) */
    float s = vx * Cy + vy * Cy;
    float ix = floorf(vx + s), iy = floorf(vy + s);
    /* x0 = v - i + dot(i, C.xx) */
    float t = ix * Cx + iy * Cx;
    float x0x = vx - ix + t, x0y = vy - iy + t;
    float i1x, i1y;
    if (x0x > x0y) { i1x = 1.0f; i1y = 0.0f; } else { i1x = 0.0f; i1y = 1.0f; }
    /* x12 = x0.xyxy + C.xxzz; x12.xy -= i1 */
    float x12x = x0x + Cx, x12y = x0y + Cx, x12z = x0x + Cz, x12w = x0y + Cz;
    x12x -= i1x;
    x12y -= i1y;
    ix = mod289f(ix);
    iy = mod289f(iy);
    /* p = permute(permute(i.y + (0,i1.y,1)) + i.x + (0,i1.x,1)) */
    float p0 = permutef(permutef(iy + 0.0f) + ix + 0.0f);
    float p1 = permutef(permutef(iy + i1y) + ix + i1x);
    float p2 = permutef(permutef(iy + 1.0f) + ix + 1.0f);
    /* m = max(0.5 - (dot(x0,x0), dot(x12.xy,x12.xy), dot(x12.zw,x12.zw)), 0) */
    float m0 = maxf_(0.5f - (x0x * x0x + x0y * x0y), 0.0f);
    float m1 = maxf_(0.5f - (x12x * x12x + x12y * x12y), 0.0f);
    float m2 = maxf_(0.5f - (x12z * x12z + x12w * x12w), 0.0f);
    m0 = m0 * m0; m1 = m1 * m1; m2 = m2 * m2;
    m0 = m0 * m0; m1 = m1 * m1; m2 = m2 * m2;
    /* x = 2*frac(p*C.www) - 1; h = |x| - 0.5; ox = floor(x+0.5); a0 = x - ox */
    float xa = 2.0f * fracf_(p0 * Cw) - 1.0f;
    float xb = 2.0f * fracf_(p1 * Cw) - 1.0f;
    float xc = 2.0f * fracf_(p2 * Cw) - 1.0f;
    float h0 = fabsf(xa) - 0.5f, h1 = fabsf(xb) - 0.5f, h2 = fabsf(xc) - 0.5f;
    float a00 = xa - floorf(xa + 0.5f);
    float a01 = xb - floorf(xb + 0.5f);
    float a02 = xc - floorf(xc + 0.5f);
    /* m *= 1.79284291400159 - 0.85373472095314 * (a0*a0 + h*h) */
    m0 *= 1.79284291400159f - 0.85373472095314f * (a00 * a00 + h0 * h0);
    m1 *= 1.79284291400159f - 0.85373472095314f * (a01 * a01 + h1 * h1);
    m2 *= 1.79284291400159f - 0.85373472095314f * (a02 * a02 + h2 * h2);
    /* g.x = a0.x*x0.x + h.x*x0.y; g.yz = a0.yz*x12.xz + h.yz*x12.yw */
    float g0 = a00 * x0x + h0 * x0y;
    float g1 = a01 * x12x + h1 * x12y;
    float g2 = a02 * x12z + h2 * x12w;
    return 130.0f * (m0 * g0 + m1 * g1 + m2 * g2);
}

/* ------------------------------------------------------------------------------------------
 * A.4 noise.psrnoise(float2 pos, float2 per, float rot)
 * ---------------------------------------------------------------------------------------- */
float nzo_psr_hash(float px, float py) { return permutef(permutef(px) + py); }

static inline void rgrad2(float px, float py, float rot, float *gx, float *gy) {
    float u = nzo_psr_hash(px, py) * 0.0243902439f + rot;
    u = fracf_(u) * 6.28318530718f;
    *gx = cosf(u);
    *gy = sinf(u);
}

float nzo_psrnoise2(float posx, float posy, float perx, float pery, float rot) {
    posy += 0.001f;
    float uvx = posx + posy * 0.5f, uvy = posy;
    float i0x = floorf(uvx), i0y = floorf(uvy);
    float f0x = fracf_(uvx), f0y = fracf_(uvy);
    float i1x, i1y;
    if (f0x > f0y) { i1x = 1.0f; i1y = 0.0f; } else { i1x = 0.0f; i1y = 1.0f; }
    float p0x = i0x - i0y * 0.5f, p0y = i0y;
    float p1x = p0x + i1x - i1y * 0.5f, p1y = p0y + i1y;
    float p2x = p0x + 0.5f, p2y = p0y + 1.0f;
    float d0x = posx - p0x, d0y = posy - p0y;
    float d1x = posx - p1x, d1y = posy - p1y;
    float d2x = posx - p2x, d2y = posy - p2y;
    /* xw = fmod((p0.x,p1.x,p2.x), per.x); yw likewise -- C# % is the truncated remainder */
    float xw0 = fmodf(p0x, perx), xw1 = fmodf(p1x, perx), xw2 = fmodf(p2x, perx);
    float yw0 = fmodf(p0y, pery), yw1 = fmodf(p1y, pery), yw2 = fmodf(p2y, pery);
    float iu0 = xw0 + 0.5f * yw0, iu1 = xw1 + 0.5f * yw1, iu2 = xw2 + 0.5f * yw2;
    float g0x, g0y, g1x, g1y, g2x, g2y;
    rgrad2(iu0, yw0, rot, &g0x, &g0y);
    rgrad2(iu1, yw1, rot, &g1x, &g1y);
    rgrad2(iu2, yw2, rot, &g2x, &g2y);
    float w0 = g0x * d0x + g0y * d0y;
    float w1 = g1x * d1x + g1y * d1y;
    float w2 = g2x * d2x + g2y * d2y;
    float t0 = 0.8f - (d0x * d0x + d0y * d0y);
    float t1 = 0.8f - (d1x * d1x + d1y * d1y);
    float t2 = 0.8f - (d2x * d2x + d2y * d2y);
    t0 = maxf_(t0, 0.0f); t1 = maxf_(t1, 0.0f); t2 = maxf_(t2, 0.0f);
    float t20 = t0 * t0, t21 = t1 * t1, t22 = t2 * t2;
    float t40 = t20 * t20, t41 = t21 * t21, t42 = t22 * t22;
    float n = t40 * w0 + t41 * w1 + t42 * w2;
    return 11.0f * n;
}

/* ------------------------------------------------------------------------------------------
 * A.5 noise.cellular(float2) -> (F1, F2)
 * ---------------------------------------------------------------------------------------- */
static inline void cell_column(float pxc, float Piy, float Pfx_off, float Pfy, float d[3]) {
    const float K = 0.142857142857f, Ko = 0.428571428571f, jitter = 1.0f;
    const float oi[3] = {-1.0f, 0.0f, 1.0f};
    const float of[3] = {-0.5f, 0.5f, 1.5f};
    for (int k = 0; k < 3; k++) {
        float p = permutef(pxc + Piy + oi[k]);
        float ox = fracf_(p * K) - Ko;
        float oy = mod7f(floorf(p * K)) * K - Ko;
        float dx = Pfx_off + jitter * ox;
        float dy = Pfy - of[k] + jitter * oy;
        d[k] = dx * dx + dy * dy;
    }
}

void nzo_cellular2(float Px, float Py, float *F1, float *F2) {
    float Pix = mod289f(floorf(Px)), Piy = mod289f(floorf(Py));
    float Pfx = fracf_(Px), Pfy = fracf_(Py);
    float px0 = permutef(Pix + -1.0f), px1 = permutef(Pix + 0.0f), px2 = permutef(Pix + 1.0f);
    float d1[3], d2[3], d3[3];
    cell_column(px0, Piy, Pfx + 0.5f, Pfy, d1);
    cell_column(px1, Piy, Pfx - 0.5f, Pfy, d2);
    cell_column(px2, Piy, Pfx - 1.5f, Pfy, d3);
    float d1a[3];
    for (int k = 0; k < 3; k++) {
        d1a[k] = minf_(d1[k], d2[k]);
        d2[k] = maxf_(d1[k], d2[k]);
        d2[k] = minf_(d2[k], d3[k]);
        d1[k] = minf_(d1a[k], d2[k]);
        d2[k] = maxf_(d1a[k], d2[k]);
    }
    /* d1.xy = (d1.x < d1.y) ? d1.xy : d1.yx */
    if (!(d1[0] < d1[1])) { float s = d1[0]; d1[0] = d1[1]; d1[1] = s; }
    /* d1.xz = (d1.x < d1.z) ? d1.xz : d1.zx */
    if (!(d1[0] < d1[2])) { float s = d1[0]; d1[0] = d1[2]; d1[2] = s; }
    d1[1] = minf_(d1[1], d2[1]);
    d1[2] = minf_(d1[2], d2[2]);
    d1[1] = minf_(d1[1], d1[2]);
    d1[1] = minf_(d1[1], d2[0]);
    *F1 = sqrtf(d1[0]);
    *F2 = sqrtf(d1[1]);
}

/* ------------------------------------------------------------------------------------------
 * A.6 noise.cnoise(float3) / noise.snoise(float3) -- only used by the DomainRotated getters
 * ---------------------------------------------------------------------------------------- */
float nzo_cnoise3(float Px, float Py, float Pz) {
    float Pi0[3] = {floorf(Px), floorf(Py), floorf(Pz)};
    float Pi1[3], Pf0[3] = {fracf_(Px), fracf_(Py), fracf_(Pz)}, Pf1[3];
    for (int k = 0; k < 3; k++) {
        Pi1[k] = Pi0[k] + 1.0f;
        Pi0[k] = mod289f(Pi0[k]);
        Pi1[k] = mod289f(Pi1[k]);
        Pf1[k] = Pf0[k] - 1.0f;
    }
    float ix[4] = {Pi0[0], Pi1[0], Pi0[0], Pi1[0]};
    float iy[4] = {Pi0[1], Pi0[1], Pi1[1], Pi1[1]};
    float gx[2][4], gy[2][4], gz[2][4];
    for (int k = 0; k < 4; k++) {
        float ixy = permutef(permutef(ix[k]) + iy[k]);
        for (int s = 0; s < 2; s++) {
            float ixyz = permutef(ixy + (s ? Pi1[2] : Pi0[2]));
            float gxx = ixyz * (1.0f / 7.0f);
            float gyy = fracf_(floorf(gxx) * (1.0f / 7.0f)) - 0.5f;
            gxx = fracf_(gxx);
            float gzz = 0.5f - fabsf(gxx) - fabsf(gyy);
            float sz = stepf_(gzz, 0.0f);
            gxx -= sz * (stepf_(0.0f, gxx) - 0.5f);
            gyy -= sz * (stepf_(0.0f, gyy) - 0.5f);
            gx[s][k] = gxx; gy[s][k] = gyy; gz[s][k] = gzz;
        }
    }
    /* lanes: k=0 -> 000/001, k=1 -> 100/101, k=2 -> 010/011, k=3 -> 110/111 */
    float n[2][4];
    for (int s = 0; s < 2; s++) {
        for (int k = 0; k < 4; k++) {
            float nr = taylor_inv_sqrt(gx[s][k] * gx[s][k] + gy[s][k] * gy[s][k] + gz[s][k] * gz[s][k]);
            float ax = gx[s][k] * nr, ay = gy[s][k] * nr, az = gz[s][k] * nr;
            float fx = (k & 1) ? Pf1[0] : Pf0[0];
            float fy = (k & 2) ? Pf1[1] : Pf0[1];
            float fz = s ? Pf1[2] : Pf0[2];
            n[s][k] = ax * fx + ay * fy + az * fz;
        }
    }
    float fdx = fadef(Pf0[0]), fdy = fadef(Pf0[1]), fdz = fadef(Pf0[2]);
    /* n_z = lerp((n000,n100,n010,n110),(n001,n101,n011,n111), fade.z) */
    float nz0 = lerpf_(n[0][0], n[1][0], fdz);
    float nz1 = lerpf_(n[0][1], n[1][1], fdz);
    float nz2 = lerpf_(n[0][2], n[1][2], fdz);
    float nz3 = lerpf_(n[0][3], n[1][3], fdz);
    /* n_yz = lerp(n_z.xy, n_z.zw, fade.y) */
    float nyz0 = lerpf_(nz0, nz2, fdy);
    float nyz1 = lerpf_(nz1, nz3, fdy);
    float nxyz = lerpf_(nyz0, nyz1, fdx);
    return 2.2f * nxyz;
}

float nzo_snoise3(float vx, float vy, float vz) {
    const float Cx = 1.0f / 6.0f, Cy = 1.0f / 3.0f;
    float v[3] = {vx, vy, vz};
    float s = vx * Cy + vy * Cy + vz * Cy;
    float i[3], x0[3];
    for (int k = 0; k < 3; k++) i[k] = floorf(v[k] + s);
    float t = i[0] * Cx + i[1] * Cx + i[2] * Cx;
    for (int k = 0; k < 3; k++) x0[k] = v[k] - i[k] + t;
    /* g = step(x0.yzx, x0.xyz); l = 1 - g; i1 = min(g.xyz, l.zxy); i2 = max(g.xyz, l.zxy) */
    float g[3] = {stepf_(x0[1], x0[0]), stepf_(x0[2], x0[1]), stepf_(x0[0], x0[2])};
    float l[3] = {1.0f - g[0], 1.0f - g[1], 1.0f - g[2]};
    float lz[3] = {l[2], l[0], l[1]};
    float i1[3], i2[3], x1[3], x2[3], x3[3];
    for (int k = 0; k < 3; k++) {
        i1[k] = minf_(g[k], lz[k]);
        i2[k] = maxf_(g[k], lz[k]);
        x1[k] = x0[k] - i1[k] + Cx;
        x2[k] = x0[k] - i2[k] + Cy;
        x3[k] = x0[k] - 0.5f;
        i[k] = mod289f(i[k]);
    }
    float oz[4] = {0.0f, i1[2], i2[2], 1.0f};
    float oy[4] = {0.0f, i1[1], i2[1], 1.0f};
    float ox[4] = {0.0f, i1[0], i2[0], 1.0f};
    const float n_ = 0.142857142857f;
    /* ns = n_ * D.wyz - D.xzx, D = (0, 0.5, 1, 2) */
    const float nsx = n_ * 2.0f - 0.0f, nsy = n_ * 0.5f - 1.0f, nsz = n_ * 1.0f - 0.0f;
    float X[4], Y[4], H[4];
    for (int k = 0; k < 4; k++) {
        float p = permutef(permutef(permutef(i[2] + oz[k]) + i[1] + oy[k]) + i[0] + ox[k]);
        float j = p - 49.0f * floorf(p * nsz * nsz);
        float x_ = floorf(j * nsz);
        float y_ = floorf(j - 7.0f * x_);
        X[k] = x_ * nsx + nsy;
        Y[k] = y_ * nsx + nsy;
        H[k] = 1.0f - fabsf(X[k]) - fabsf(Y[k]);
    }
    /* b0 = (x.xy, y.xy); b1 = (x.zw, y.zw); s = floor(b)*2+1; sh = -step(h, 0) */
    float b0[4] = {X[0], X[1], Y[0], Y[1]}, b1[4] = {X[2], X[3], Y[2], Y[3]};
    float s0[4], s1[4], sh[4];
    for (int k = 0; k < 4; k++) {
        s0[k] = floorf(b0[k]) * 2.0f + 1.0f;
        s1[k] = floorf(b1[k]) * 2.0f + 1.0f;
        sh[k] = -stepf_(H[k], 0.0f);
    }
    /* a0 = b0.xzyw + s0.xzyw * sh.xxyy; a1 = b1.xzyw + s1.xzyw * sh.zzww */
    float a0[4] = {b0[0] + s0[0] * sh[0], b0[2] + s0[2] * sh[0], b0[1] + s0[1] * sh[1],
                   b0[3] + s0[3] * sh[1]};
    float a1[4] = {b1[0] + s1[0] * sh[2], b1[2] + s1[2] * sh[2], b1[1] + s1[1] * sh[3],
                   b1[3] + s1[3] * sh[3]};
    float P[4][3] = {{a0[0], a0[1], H[0]}, {a0[2], a0[3], H[1]}, {a1[0], a1[1], H[2]},
                     {a1[2], a1[3], H[3]}};
    float *xs[4] = {x0, x1, x2, x3};
    float m[4], pd[4];
    for (int k = 0; k < 4; k++) {
        float nr = taylor_inv_sqrt(P[k][0] * P[k][0] + P[k][1] * P[k][1] + P[k][2] * P[k][2]);
        P[k][0] *= nr; P[k][1] *= nr; P[k][2] *= nr;
        float *x = xs[k];
        m[k] = maxf_(0.6f - (x[0] * x[0] + x[1] * x[1] + x[2] * x[2]), 0.0f);
        m[k] = m[k] * m[k];
        pd[k] = P[k][0] * x[0] + P[k][1] * x[1] + P[k][2] * x[2];
    }
    /* 42 * dot(m*m, (dot(p0,x0),...)) ; float4 dot = ((a+b)+c)+d */
    return 42.0f * ((m[0] * m[0]) * pd[0] + (m[1] * m[1]) * pd[1] + (m[2] * m[2]) * pd[2] +
                    (m[3] * m[3]) * pd[3]);
}

/* ------------------------------------------------------------------------------------------
 * IMakeNoise getters: Noise/Fractal/Fractal.cs:141-278
 * ---------------------------------------------------------------------------------------- */
static inline float rectify(float v) { return (1.0f + v) / 2.0f * 1.0f; } /* (RV+value)/2*RV */

static inline void domain_rotate(float x, float z, float *xr, float *zr, float *yr) {
    /* Fractal.cs:161-166,248-253 */
    float xz = x + z;
    float s2 = xz * -0.211324865405187f;
    *xr = x + s2;
    *zr = z + s2;
    *yr = xz * -0.577350269189626f;
}

float nzo_noise_value(int noiseType, float x, float z) {
    switch (noiseType) {
        case NZO_NOISE_SIN: { /* SinGetter Fractal.cs:210-225 */
            float vx = 0.5f + (0.5f * sinf(x));
            float vy = 0.5f + (0.5f * sinf(z));
            return vx * vy;
        }
        case NZO_NOISE_PERLIN: /* PerlinGetter :141-154 */
            return rectify(nzo_cnoise2(x, z));
        case NZO_NOISE_PERIODIC_PERLIN: /* PeriodicPerlinGetter :176-191, 2-arg overload = rot 0 */
            return rectify(nzo_psrnoise2(x, z, 1010.0f, 102.0f, 0.0f));
        case NZO_NOISE_SIMPLEX: /* SimplexGetter :227-241 */
            return rectify(nzo_snoise2(x, z));
        case NZO_NOISE_ROTATED_SIMPLEX: /* RotatedSimplexGetter :193-208 */
            return rectify(nzo_psrnoise2(x, z, 1010.0f, 102.0f, 0.62f));
        case NZO_NOISE_CELLULAR: { /* CellularGetter :263-278 */
            float f1, f2;
            nzo_cellular2(x, z, &f1, &f2);
            return rectify(f1) * rectify(f2);
        }
        case NZO_NOISE_DOMAIN_ROTATED_PERLIN: { /* :156-174 */
            float xr, zr, yr;
            domain_rotate(x, z, &xr, &zr, &yr);
            return rectify(nzo_cnoise3(xr, zr, yr));
        }
        case NZO_NOISE_DOMAIN_ROTATED_SIMPLEX: { /* :243-261 */
            float xr, zr, yr;
            domain_rotate(x, z, &xr, &zr, &yr);
            return rectify(nzo_snoise3(xr, zr, yr));
        }
    }
    return 0.0f;
}

/* FractalJob.CalcFractalNormValue Fractal.cs:31-40 (ignores startingAmplitude) */
float nzo_fractal_norm(float hurst, int octaves, float startingAmplitude) {
    (void)startingAmplitude;
    float G = exp2f(-hurst);
    float a = 1.0f, t = 0.0f;
    for (int i = 0; i < octaves; i++) {
        t += a * 1.0f;
        a *= G;
    }
    return t;
}

/* FractalGenerator.NoiseValue Fractal.cs:114-131 */
static inline float fractal_value(int noiseType, int x, int z, float posx, float posz,
                                  int noiseSize, float G, float startingAmplitude, float stepdown,
                                  float detuneRate, int octaves, float norm) {
    float xi = ((float)x + posx) / (float)noiseSize;
    float zi = ((float)z + posz) / (float)noiseSize;
    float detune = 0.0f, f = 1.0f, a = startingAmplitude, t = 0.0f;
    for (int i = 0; i < octaves; i++) {
        float xV = f * xi;
        float zV = f * zi;
        t += a * nzo_noise_value(noiseType, xV, zV);
        detune += detuneRate;
        f *= (stepdown - detune);
        a *= G;
    }
    return t / norm;
}

float nzo_fractal_cell(int noiseType, int x, int z, float hurst, float startingAmplitude,
                       float stepdown, float detuneRate, int octaves, int xpos, int zpos,
                       int noiseSize) {
    float norm = nzo_fractal_norm(hurst, octaves, startingAmplitude);
    return fractal_value(noiseType, x, z, (float)xpos, (float)zpos, noiseSize, exp2f(-hurst),
                         startingAmplitude, stepdown, detuneRate, octaves, norm);
}

/* FractalJob.ScheduleParallel Fractal.cs:42-73; rows of FractalGenerator.Execute :134-138;
 * WriteTileData.SetValue is unclamped (TileData.cs:135-143) */
int nzo_fractal(int noiseType, float *dst, int rows, int cols, float hurst,
                float startingAmplitude, float stepdown, float detuneRate, int octaves, int xpos,
                int zpos, int noiseSize) {
    if (noiseType < 0 || noiseType > 7 || rows <= 0 || cols <= 0) return -1;
    float norm = nzo_fractal_norm(hurst, octaves, startingAmplitude);
    float G = exp2f(-hurst);
    float posx = (float)xpos, posz = (float)zpos; /* SetPosition Fractal.cs:109-112 */
#pragma omp parallel for schedule(dynamic, 1)
    for (int z = 0; z < rows; z++) {
        for (int x = 0; x < cols; x++) {
            dst[(size_t)z * cols + x] = fractal_value(noiseType, x, z, posx, posz, noiseSize, G,
                                                      startingAmplitude, stepdown, detuneRate,
                                                      octaves, norm);
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Kernel operators: Filter/Kernel/KernelOperators.cs
 * ---------------------------------------------------------------------------------------- */

/* KernelSampleXOperator.ApplyKernel :32-41 (k ascending) + flush (KernelJob.cs:48-51) */
void nzo_pass_sample_x(float *src, float *tmp, int rows, int cols, int ksize, const float *kernel,
                       float factor) {
    int k_off = (ksize - 1) / 2;
#pragma omp parallel for schedule(static)
    for (int z = 0; z < rows; z++) {
        for (int x = 0; x < cols; x++) {
            float total = 0.0f;
            for (int k = -k_off; k <= k_off; k++) {
                total += src[tile_idx(x + k, z, rows, cols)] * kernel[k_off + k];
            }
            tmp[tile_idx(x, z, rows, cols)] = total * factor;
        }
    }
    flush_write_slice(src, tmp, (size_t)rows * cols);
}

/* KernelSampleZOperator.ApplyKernel :58-66 (k descending) */
void nzo_pass_sample_z(float *src, float *tmp, int rows, int cols, int ksize, const float *kernel,
                       float factor) {
    int k_off = (ksize - 1) / 2;
#pragma omp parallel for schedule(static)
    for (int z = 0; z < rows; z++) {
        for (int x = 0; x < cols; x++) {
            float total = 0.0f;
            for (int k = k_off; k >= -k_off; k--) {
                total += src[tile_idx(x, z + k, rows, cols)] * kernel[k_off - k];
            }
            tmp[tile_idx(x, z, rows, cols)] = total * factor;
        }
    }
    flush_write_slice(src, tmp, (size_t)rows * cols);
}

/* KernelMinXOperator.ApplyKernel :83-91: k in [-k_off, k_off) */
void nzo_pass_min_x(float *src, float *tmp, int rows, int cols, int ksize) {
    int k_off = (ksize - 1) / 2;
#pragma omp parallel for schedule(static)
    for (int z = 0; z < rows; z++) {
        for (int x = 0; x < cols; x++) {
            float m = 3.40282347e+38f; /* Single.MaxValue */
            for (int k = -k_off; k < k_off; k++) m = minf_(m, src[tile_idx(x + k, z, rows, cols)]);
            tmp[tile_idx(x, z, rows, cols)] = m;
        }
    }
    flush_write_slice(src, tmp, (size_t)rows * cols);
}

/* KernelMinZOperator.ApplyKernel :108-117 */
void nzo_pass_min_z(float *src, float *tmp, int rows, int cols, int ksize) {
    int k_off = (ksize - 1) / 2;
#pragma omp parallel for schedule(static)
    for (int z = 0; z < rows; z++) {
        for (int x = 0; x < cols; x++) {
            float m = 3.40282347e+38f;
            for (int k = -k_off; k < k_off; k++) m = minf_(m, src[tile_idx(x, z + k, rows, cols)]);
            tmp[tile_idx(x, z, rows, cols)] = m;
        }
    }
    flush_write_slice(src, tmp, (size_t)rows * cols);
}

/* SeparableKernelFilter.ScheduleSeries KernelJob.cs:165-185: X pass then Z pass on its output */
void nzo_separable(float *src, float *tmp, int rows, int cols, int ksize, const float *kx,
                   const float *kz, float factor) {
    nzo_pass_sample_x(src, tmp, rows, cols, ksize, kx, factor);
    nzo_pass_sample_z(src, tmp, rows, cols, ksize, kz, factor);
}

/* Normalised Gaussian, exp(-i^2/(2 sigma^2))/sum evaluated in double and rounded to fp32.
 * This regenerates the literals of KernelJob.cs:97-105 and BlurKernels.cs:59-318
 * (checked value-by-value in tests/test_oracle_tables.py against tests/golden/gauss_tables.json). */
static void gauss_coeffs(double sigma, int width, float *out) {
    int o = (width - 1) / 2;
    double w[25], sum = 0.0;
    for (int i = 0; i < width; i++) {
        double d = (double)(i - o);
        w[i] = exp(-(d * d) / (2.0 * sigma * sigma));
        sum += w[i];
    }
    for (int i = 0; i < width; i++) out[i] = (float)(w[i] / sum);
}

/* SeparableKernelFilter tables + switch, KernelJob.cs:97-136,217-294 */
int nzo_kernel_filter_table(int filterType, float *kx, float *kz, float *factor, int *ksize) {
    static const float sobel3_HX[3] = {-1.0f, 0.0f, 1.0f}, sobel3_HZ[3] = {1.0f, 2.0f, 1.0f};
    static const float sobel3_VX[3] = {1.0f, 2.0f, 1.0f}, sobel3_VZ[3] = {1.0f, 0.0f, -1.0f};
    static const float prewitt3_HX[3] = {1.0f, 0.0f, -1.0f}, prewitt3_HZ[3] = {1.0f, 1.0f, 1.0f};
    static const float prewitt3_VX[3] = {1.0f, 1.0f, 1.0f}, prewitt3_VZ[3] = {-1.0f, 0.0f, 1.0f};
    switch (filterType) {
        case NZO_GAUSS9_S1: case NZO_GAUSS7_S1: case NZO_GAUSS5_S1: case NZO_GAUSS3_S1:
        case NZO_GAUSS9_S2: case NZO_GAUSS7_S2: case NZO_GAUSS5_S2: case NZO_GAUSS3_S2: {
            static const int sizes[4] = {9, 7, 5, 3};
            int w = sizes[filterType & 3];
            double sigma = filterType >= NZO_GAUSS9_S2 ? 2.0 : 1.0;
            gauss_coeffs(sigma, w, kx);
            gauss_coeffs(sigma, w, kz);
            *factor = 1.0f;
            *ksize = w;
            return 0;
        }
        case NZO_SMOOTH3: /* smooth3 {1,1,1}, factor 1f/3f per pass, KernelJob.cs:107-108 */
            for (int i = 0; i < 3; i++) kx[i] = kz[i] = 1.0f;
            *factor = 1.0f / 3.0f;
            *ksize = 3;
            return 0;
        case NZO_SOBEL3_HORIZONTAL:
            memcpy(kx, sobel3_HX, sizeof sobel3_HX); memcpy(kz, sobel3_HZ, sizeof sobel3_HZ);
            *factor = 1.0f; *ksize = 3;
            return 0;
        case NZO_SOBEL3_VERTICAL:
            memcpy(kx, sobel3_VX, sizeof sobel3_VX); memcpy(kz, sobel3_VZ, sizeof sobel3_VZ);
            *factor = 1.0f; *ksize = 3;
            return 0;
        case NZO_PREWITT3_HORIZONTAL:
            memcpy(kx, prewitt3_HX, sizeof prewitt3_HX); memcpy(kz, prewitt3_HZ, sizeof prewitt3_HZ);
            *factor = 1.0f; *ksize = 3;
            return 0;
        case NZO_PREWITT3_VERTICAL:
            memcpy(kx, prewitt3_VX, sizeof prewitt3_VX); memcpy(kz, prewitt3_VZ, sizeof prewitt3_VZ);
            *factor = 1.0f; *ksize = 3;
            return 0;
        default:
            return -1; /* Sobel3_2D goes through ScheduleReduce (KernelJob.cs:187-215): out of scope */
    }
}

int nzo_kernel_filter(float *src, float *tmp, int filterType, int rows, int cols) {
    float kx[9], kz[9], factor;
    int ksize;
    if (nzo_kernel_filter_table(filterType, kx, kz, &factor, &ksize) != 0) return -1;
    nzo_separable(src, tmp, rows, cols, ksize, kx, kz, factor);
    return 0;
}

/* BlurHelper.limitWidth BlurKernels.cs:29-36 */
int nzo_limit_width(int width) {
    if (width % 2 == 0) width += 1;
    if (width > 25) width = 25;
    return width < 3 ? 3 : width;
}

/* GaussianKernel.GetKernel BlurKernels.cs:44-56: idx = floor(w/2)-1 into a 12-entry list per sigma;
 * sigma enum s0d50..s8d00 = 0.5 * (enum + 1) */
int nzo_gauss_kernel(int sigmaEnum, int width, float *out) {
    if (sigmaEnum < 0 || sigmaEnum > 15) return -1;
    int w = nzo_limit_width(width);
    gauss_coeffs(0.5 * (double)(sigmaEnum + 1), w, out);
    return w;
}

/* GaussFilter.Schedule BlurJob.cs:11-21.  The kernel body comes from limitWidth(width) but the
 * pass is scheduled with kernelSize = width as given (the stage passes an already-limited width,
 * StageGaussianBlur.cs:38). */
int nzo_gauss(float *src, float *tmp, int width, int sigmaEnum, int rows, int cols) {
    float k[25];
    int w = nzo_gauss_kernel(sigmaEnum, width, k);
    if (w < 0 || width < 1 || width > w) return -1;
    nzo_separable(src, tmp, rows, cols, width, k, k, 1.0f);
    return 0;
}

/* SmoothFilter.Schedule BlurJob.cs:34-44; SmoothBlur.GetKernel BlurKernels.cs:39-43 (1f/width) */
int nzo_smooth(float *src, float *tmp, int width, int rows, int cols) {
    float k[64];
    if (width < 1 || width > 64) return -1;
    for (int i = 0; i < width; i++) k[i] = 1.0f / (float)width;
    nzo_separable(src, tmp, rows, cols, width, k, k, 1.0f);
    return 0;
}

/* ErosionKernelJob.Schedule KernelJob.cs:317-347: size 3, min-X then min-Z, each flushed */
int nzo_erosion_min(float *src, int rows, int cols) {
    float *tmp = (float *)malloc((size_t)rows * cols * sizeof(float)); /* Allocator.TempJob :327 */
    if (!tmp) return -1;
    nzo_pass_min_x(src, tmp, rows, cols, 3);
    nzo_pass_min_z(src, tmp, rows, cols, 3);
    free(tmp);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Flow map: Geologic/FlowMap/FlowMapComponents.cs, FlowMapJob.cs, Geologic/Stage/FlowMapStage.cs
 * ---------------------------------------------------------------------------------------- */

/* FillArrayJob FlowMapComponents.cs:175-202 */
void nzo_fill(float *data, int rows, int cols, float value) {
#pragma omp parallel for schedule(static)
    for (int z = 0; z < rows; z++)
        for (int x = 0; x < cols; x++) data[(size_t)z * cols + x] = value;
}

/* ComputeFlowStep.CalculateCell FlowMapComponents.cs:20-65 */
void nzo_flow_step(const float *height, const float *water, float *fN, float *fN_buf, float *fS,
                   float *fS_buf, float *fE, float *fE_buf, float *fW, float *fW_buf, int rows,
                   int cols) {
    const float TIMESTEP = 0.2f;
#pragma omp parallel for schedule(static)
    for (int z = 0; z < rows; z++) {
        for (int x = 0; x < cols; x++) {
            int c = tile_idx(x, z, rows, cols);
            float height_0 = height[c], water_0 = water[c];
            int iW = tile_idx(x - 1, z, rows, cols), iE = tile_idx(x + 1, z, rows, cols);
            int iS = tile_idx(x, z - 1, rows, cols), iN = tile_idx(x, z + 1, rows, cols);
            float totalHt = water_0 + height_0;
            float dW = totalHt - (water[iW] + height[iW]);
            float dE = totalHt - (water[iE] + height[iE]);
            float dS = totalHt - (water[iS] + height[iS]);
            float dN = totalHt - (water[iN] + height[iN]);
            float flW = maxf_(0.0f, fW[c] + dW);
            float flE = maxf_(0.0f, fE[c] + dE);
            float flS = maxf_(0.0f, fS[c] + dS);
            float flN = maxf_(0.0f, fN[c] + dN);
            float sum_ = (flW + flE) + (flS + flN); /* math.csum(float4) = (x.x + x.y) + (x.z + x.w), Unity.Mathematics 1.2.1 */
            if (sum_ > 0.0f) {
                float K = water_0 / (sum_ * TIMESTEP);
                /* math.clamp(K, 0, 1) = max(0, min(1, K)) with Unity.Mathematics' min / max (a NaN SECOND operand is skipped):
                 * K is NaN when water_0 is 0 and sum_ * TIMESTEP underflows to 0 (sum_ a denormal): min(1, NaN) = 1.  (A ternary
                 * clamp, which rounds 1-5 carried here, passes the NaN on: found by round 6's soak, seed 62067.) */
                K = maxf_(0.0f, minf_(1.0f, K));
                fW_buf[c] = flW * K;
                fE_buf[c] = flE * K;
                fS_buf[c] = flS * K;
                fN_buf[c] = flN * K;
            } else {
                fW_buf[c] = 0.0f;
                fE_buf[c] = 0.0f;
                fS_buf[c] = 0.0f;
                fN_buf[c] = 0.0f;
            }
        }
    }
    /* FlowMapJob.cs:74-77: N, S, E, W flushed in that order, serially */
    size_t n = (size_t)rows * cols;
    flush_write_slice(fN, fN_buf, n);
    flush_write_slice(fS, fS_buf, n);
    flush_write_slice(fE, fE_buf, n);
    flush_write_slice(fW, fW_buf, n);
}

/* UpdateWaterStep.CalculateCell FlowMapComponents.cs:81-104 */
void nzo_water_step(float *water, float *water_buf, const float *fN, const float *fS,
                    const float *fE, const float *fW, int rows, int cols) {
    const float TIMESTEP = 0.2f;
#pragma omp parallel for schedule(static)
    for (int z = 0; z < rows; z++) {
        for (int x = 0; x < cols; x++) {
            int c = tile_idx(x, z, rows, cols);
            float flowOUT = fW[c] + fE[c] + fS[c] + fN[c];
            float flowIN = 0.0f;
            flowIN += fE[tile_idx(x - 1, z, rows, cols)];
            flowIN += fW[tile_idx(x + 1, z, rows, cols)];
            flowIN += fN[tile_idx(x, z - 1, rows, cols)];
            flowIN += fS[tile_idx(x, z + 1, rows, cols)];
            float ht = water[c] + ((flowIN - flowOUT) * TIMESTEP);
            ht = maxf_(0.0f, ht);
            water_buf[c] = ht;
        }
    }
    flush_write_slice(water, water_buf, (size_t)rows * cols);
}

/* CreateVelocityField.CalculateCell FlowMapComponents.cs:120-139; write-only (unclamped) target */
void nzo_velocity(float *dst, const float *fN, const float *fS, const float *fE, const float *fW,
                  int rows, int cols) {
#pragma omp parallel for schedule(static)
    for (int z = 0; z < rows; z++) {
        for (int x = 0; x < cols; x++) {
            float dl = fE[tile_idx(x - 1, z, rows, cols)] - fW[tile_idx(x, z, rows, cols)];
            float dr = fE[tile_idx(x, z, rows, cols)] - fW[tile_idx(x + 1, z, rows, cols)];
            float dt = fS[tile_idx(x, z + 1, rows, cols)] - fN[tile_idx(x, z, rows, cols)];
            float db = fS[tile_idx(x, z, rows, cols)] - fN[tile_idx(x, z - 1, rows, cols)];
            float vx = (dl + dr) * 0.5f;
            float vy = (dt + db) * 0.5f;
            dst[(size_t)z * cols + x] = sqrtf(vx * vx + vy * vy); /* sqrt(lengthsq(v)) */
        }
    }
}

/* NormalizeMap.CalculateCell FlowMapComponents.cs:157-165 + flush NormalizeJob.cs:89 */
void nzo_normalize(float *src, float *tmp, const float *args, int rows, int cols) {
#pragma omp parallel for schedule(static)
    for (int z = 0; z < rows; z++) {
        for (int x = 0; x < cols; x++) {
            int c = tile_idx(x, z, rows, cols);
            float v = src[c];
            if (args[2] < 1e-12f) v = 0.0f;
            tmp[c] = (v - args[0]) / args[2];
        }
    }
    flush_write_slice(src, tmp, (size_t)rows * cols);
}

/* GetMapRangeJob.Execute Filter/NormalizeJob.cs:33-43: one thread, in index order; math.min / math.max of
 * Unity.Mathematics 1.2.1 (`IsNaN(y) || x < y ? x : y`): a NaN cell is skipped, on a tie the later operand stays */
void nzo_get_map_range(const float *map, size_t n, float lim_min, float lim_max, float *res) {
    float min_ = lim_min, max_ = lim_max;
    for (size_t i = 0; i < n; i++) {
        float y = map[i];
        min_ = (y != y) || min_ < y ? min_ : y;
        max_ = (y != y) || max_ > y ? max_ : y;
    }
    res[0] = min_;
    res[1] = max_;
    res[2] = max_ - min_;
}

/* FlowMapStage.ScheduleAll FlowMapStage.cs:124-195.  The reference leaves the flux planes
 * uninitialised (:55-62); this build defines them as zero at the start of every run. */
int nzo_flowmap(float *src, int rows, int cols, int iterations, float normMin, float normMax) {
    if (iterations < 1) return -1;
    size_t n = (size_t)rows * cols;
    float *planes = (float *)calloc(n * 11, sizeof(float));
    if (!planes) return -1;
    float *tmp = planes, *water = planes + n, *water_b = planes + 2 * n;
    float *fN = planes + 3 * n, *fN_b = planes + 4 * n, *fS = planes + 5 * n, *fS_b = planes + 6 * n;
    float *fE = planes + 7 * n, *fE_b = planes + 8 * n, *fW = planes + 9 * n, *fW_b = planes + 10 * n;
    float args[3] = {normMin, normMax, normMax - normMin}; /* FlowMapStage.cs:48-51 */
    nzo_fill(water, rows, cols, 0.0001f);                  /* :129 */
    for (int i = 0; i < iterations; i++) {
        nzo_flow_step(src, water, fN, fN_b, fS, fS_b, fE, fE_b, fW, fW_b, rows, cols);
        nzo_water_step(water, water_b, fN, fS, fE, fW, rows, cols);
    }
    nzo_velocity(src, fN, fS, fE, fW, rows, cols); /* :179-186 writes over the height data */
    nzo_normalize(src, tmp, args, rows, cols);     /* :188-194 */
    free(planes);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Mesh: Mesh/Generators/{Overshoot,}SquareGridHeightMap.cs, Mesh/Job/HeightMapMeshJob.cs
 * ---------------------------------------------------------------------------------------- */

typedef struct {
    int res, in_res, off, type;
    float height, tile_size, normal_strength;
    const float *heights;
} mesh_gen;

/* OvershootSquareGridHeightMap.getIdx :54-59 / SquareGridHeightMap.getIdx :59-64 */
static inline float mesh_h(const mesh_gen *g, int x, int z) {
    if (g->type == NZO_MESH_OVERSHOOT) {
        x = clampi(x, 0 - g->off, g->res + g->off);
        z = clampi(z, 0 - g->off, g->res + g->off);
    } else {
        x = clampi(x, 0, g->res + 1);
        z = clampi(z, 0, g->res + 1);
    }
    return g->heights[((z + g->off) * g->in_res) + x + g->off];
}

static inline float interpolate_edge(float a, float b) { return a - (b - a); } /* Square :35-38 */

/* SetVertexValues: Overshoot :62-75, Square :67-82.  v = 12 floats {pos3,normal3,tangent4,uv2};
 * position.x/.z are set by the caller, tangent.w stays 0 (`new Vertex()`). */
static void mesh_vertex_values(const mesh_gen *g, float *v, int x, int z) {
    float t = mesh_h(g, x, z);
    v[1] = t * g->height;
    float l, r, u, d;
    if (g->type == NZO_MESH_OVERSHOOT) {
        l = mesh_h(g, x - 1, z);
        r = mesh_h(g, x + 1, z);
        u = mesh_h(g, x, z - 1);
        d = mesh_h(g, x, z + 1);
    } else {
        l = x > 0 ? mesh_h(g, x - 1, z) : interpolate_edge(t, mesh_h(g, x + 1, z));
        r = x < g->res - 1 ? mesh_h(g, x + 1, z) : interpolate_edge(t, mesh_h(g, x - 1, z));
        u = z > 0 ? mesh_h(g, x, z - 1) : interpolate_edge(mesh_h(g, x, z + 1), t);
        d = z < g->res - 1 ? mesh_h(g, x, z + 1) : interpolate_edge(mesh_h(g, x, z - 1), t);
    }
    float t1x = 4.0f, t1y = (r - l) / 2.0f, t1z = 0.0f;
    float t2x = 0.0f, t2y = (u - d) / 2.0f, t2z = 4.0f;
    /* math.cross(t2, t1) = (a.y b.z - a.z b.y, a.z b.x - a.x b.z, a.x b.y - a.y b.x) */
    v[6] = t2y * t1z - t2z * t1y;
    v[7] = t2z * t1x - t2x * t1z;
    v[8] = t2x * t1y - t2y * t1x;
    v[9] = 0.0f;
    /* normalize(x) = rsqrt(dot(x,x)) * x, rsqrt = 1/sqrt */
    float nx = (l - r) / 2.0f * g->normal_strength;
    float ny = 2.0f / g->height;
    float nz = (u - d) / 2.0f * g->normal_strength;
    float rs = 1.0f / sqrtf(nx * nx + ny * ny + nz * nz);
    v[3] = rs * nx;
    v[4] = rs * ny;
    v[5] = rs * nz;
    if (g->type == NZO_MESH_OVERSHOOT) {
        v[10] = ((float)x) / (((float)g->res) - 0.5f);
        v[11] = ((float)z) / (((float)g->res) - 0.5f);
    } else {
        v[10] = ((float)x) / ((float)g->res + 1.0f);
        v[11] = ((float)z) / ((float)g->res + 1.0f);
    }
}

/* MeshJob<SharedSquareGridPosition, PositionStream32>, Mesh/Generators/SharedSquareGridPosition.cs:20-50: the flat
 * unit-square grid of MeshHelper.makeSquarePlanarMesh (Mesh/Helpers/Helper.cs:47-58).  The vertex is reused along
 * a row, so texCoord0.x of column 0 is the default 0 and position.x of column 0 is the literal -0.5. */
int nzo_mesh_square_grid(int resolution, float *vtx, uint32_t *idx) {
    if (resolution < 1) return -1;
    int R = resolution;
#pragma omp parallel for schedule(static)
    for (int z = 0; z <= R; z++) {
        int vi = (R + 1) * z, ti = 2 * R * (z - 1);
        float v[12] = {0};
        v[5] = -1.0f;              /* normal.z */
        v[6] = 1.0f; v[9] = -1.0f; /* tangent.xw */
        v[0] = -0.5f;
        v[2] = (float)z / (float)R - 0.5f;
        v[1] = 0.0f;
        v[11] = ((float)z) / ((float)R + 1.0f);
        memcpy(vtx + (size_t)vi * 12, v, sizeof v);
        vi += 1;
        for (int x = 1; x <= R; x++, vi++, ti += 2) {
            v[0] = (float)x / (float)R - 0.5f;
            v[10] = ((float)x) / ((float)R + 1.0f);
            memcpy(vtx + (size_t)vi * 12, v, sizeof v);
            if (z > 0) {
                uint32_t *t0 = idx + (size_t)(ti + 0) * 3, *t1 = idx + (size_t)(ti + 1) * 3;
                t0[0] = (uint32_t)(vi - R - 2); t0[1] = (uint32_t)(vi - 1); t0[2] = (uint32_t)(vi - R - 1);
                t1[0] = (uint32_t)(vi - R - 1); t1[1] = (uint32_t)(vi - 1); t1[2] = (uint32_t)vi;
            }
        }
    }
    return 0;
}

int nzo_mesh_heightmap(int meshType, const float *heights, int resolution, int inputResolution,
                       int marginPix, float tileHeight, float tileSize, float *vtx,
                       uint32_t *idx) {
    (void)marginPix; /* MarginScale is commented out of the vertex path (Overshoot :64) */
    mesh_gen g;
    g.res = resolution;
    g.in_res = inputResolution;
    g.off = (inputResolution - resolution) / 2; /* PixOffset :33 */
    g.type = meshType;
    g.height = tileHeight;
    g.tile_size = tileSize;
    g.normal_strength = 8.0f; /* HeightMapMeshJob.cs:41 */
    g.heights = heights;
    if (resolution < 1 || inputResolution < resolution) return -1;
    /* The reference reads out of bounds (safety checks off) when the margin is too small
     * (SURVEY B17); the oracle and the HIP path reject those shapes instead. */
    if (meshType == NZO_MESH_OVERSHOOT) {
        int hi = resolution + 1 < resolution + g.off ? resolution + 1 : resolution + g.off;
        if (hi + g.off > inputResolution - 1) return -1;
    } else if (meshType == NZO_MESH_SQUARE) {
        if (resolution + g.off > inputResolution - 1) return -1;
    } else {
        return -1;
    }
    int R = resolution;
    /* Execute(z) Overshoot :77-102 / Square :84-105, one row per job, z in [0, R] */
#pragma omp parallel for schedule(static)
    for (int z = 0; z <= R; z++) {
        int vi = (R + 1) * z, ti = 2 * R * (z - 1);
        float v[12] = {0};
        v[0] = -(0.5f * tileSize / (float)R);
        v[2] = (float)z * tileSize / (float)R - 0.5f;
        mesh_vertex_values(&g, v, 0, z);
        memcpy(vtx + (size_t)vi * 12, v, sizeof v);
        vi += 1;
        for (int x = 1; x <= R; x++, vi++, ti += 2) {
            v[0] = (float)x * tileSize / (float)R - 0.5f;
            mesh_vertex_values(&g, v, x, z);
            memcpy(vtx + (size_t)vi * 12, v, sizeof v);
            if (z > 0) {
                uint32_t *t0 = idx + (size_t)(ti + 0) * 3, *t1 = idx + (size_t)(ti + 1) * 3;
                t0[0] = (uint32_t)(vi - R - 2); t0[1] = (uint32_t)(vi - 1); t0[2] = (uint32_t)(vi - R - 1);
                t1[0] = (uint32_t)(vi - R - 1); t1[1] = (uint32_t)(vi - 1); t1[2] = (uint32_t)(vi);
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Element-wise stages: Filter/Operators/SimpleMutation.cs, Filter/ConstantJob.cs,
 * Filter/ReductionJob.cs, Filter/Curve/CurveJob.cs -- parallel rows into tmp, then the serial flush
 * ---------------------------------------------------------------------------------------- */

/* ConstantMultiply / ConstantBinarize, SimpleMutation.cs:16-55; op = ConstantStage.ConstantOperationType */
int nzo_constant(float *src, float *tmp, int op, float value, int rows, int cols) {
    if (op < 0 || op > 1) return -1;
#pragma omp parallel for schedule(static)
    for (int z = 0; z < rows; z++) {
        for (int x = 0; x < cols; x++) {
            int c = tile_idx(x, z, rows, cols);
            tmp[c] = op == 0 ? src[c] * value : (src[c] >= value ? 1.0f : 0.0f);
        }
    }
    flush_write_slice(src, tmp, (size_t)rows * cols);
    return 0;
}

/* Subtract / Multiply / RootSumSquares / Max / Min Tiles, SimpleMutation.cs:57-171; op = ReductionType */
int nzo_reduce(float *srcL, const float *srcR, float *tmp, int op, int rows, int cols) {
    if (op < 0 || op > 4) return -1;
#pragma omp parallel for schedule(static)
    for (int z = 0; z < rows; z++) {
        for (int x = 0; x < cols; x++) {
            int c = tile_idx(x, z, rows, cols);
            float a = srcL[c], b = srcR[c], v;
            switch (op) {
                case 0: v = a - b; break;
                case 1: v = a * b; break;
                case 2: v = sqrtf((a * a) + (b * b)); break;
                case 3: v = maxf_(a, b); break;
                default: v = minf_(a, b); break;
            }
            tmp[c] = v;
        }
    }
    flush_write_slice(srcL, tmp, (size_t)rows * cols);
    return 0;
}

/* ---- live erosion: the deterministic grid jobs -----------------------------------------
 * WorldTile planes are indexed x * res + z (LiveErosionDataTypes.cs:608-610), not z * res + x. */

/* WorldTile.UpdateFlowMapFromTrack, LiveErosionDataTypes.cs:869-886 (UpdateFlowFromTrackJob, MultiThreadErosionJob.cs:
 * 226-262): flow decays, gains from the particle track where no pool stands, the track is cleared, pools evaporate */
int nzo_update_flow_from_track(float *pool, float *flow, float *track, int res, float flowLossRate,
                               float surfaceEvaporationRate, float tileHeight) {
    const float MINFLOWPOOL = .00005f; /* :440 */
#pragma omp parallel for schedule(static)
    for (int z = 0; z < res; z++) {
        for (int x = 0; x < res; x++) {
            size_t i = (size_t)x * res + z;
            float pv = flow[i], tv = track[i], poolV = pool[i];
            if (poolV > MINFLOWPOOL) {
                flow[i] = ((1.0f - 0.1f * flowLossRate) * pv);
            } else if (tv > 0.0f) {
                flow[i] = ((1.0f - flowLossRate) * pv) + (flowLossRate * 50.0f * tv) / (1.0f + 50.0f * tv);
            } else {
                flow[i] = (1.0f - flowLossRate) * pv;
            }
            track[i] = 0.0f;
            pool[i] = maxf_(poolV - (surfaceEvaporationRate / tileHeight), 0.0f);
        }
    }
    return 0;
}

/* FloodedNeighbor (LiveErosionDataTypes.cs:1013-1050): ordered by the HASH of height + water, i.e. by the bit
 * pattern of the float as a signed int (float.GetHashCode: the bits, +-0 -> 0); same idx compares equal. */
typedef struct { int idx; float height, water; } nzo_flooded;
static inline int nzo_float_hash(float f) {
    if (f == 0.0f) return 0;
    int v;
    memcpy(&v, &f, sizeof v);
    return v;
}
static inline int nzo_flooded_cmp(const nzo_flooded *a, const nzo_flooded *b) { /* a.CompareTo(b) */
    if (a->idx == b->idx) return 0;
    return nzo_float_hash(a->height + a->water) > nzo_float_hash(b->height + b->water) ? 1 : -1;
}
/* NativeArray<T>.Sort() of com.unity.collections 1.4.0 (package.json:17; not in the reference tree, restated from
 * the published source): partitions of <= 16 elements are insertion-sorted, element i+1 moving left while it
 * compares < 0 -- for 4 elements that is the whole sort. */
static inline void nzo_flooded_sort4(nzo_flooded *a) {
    for (int i = 0; i < 3; i++) {
        int j = i;
        nzo_flooded t = a[i + 1];
        while (j >= 0 && nzo_flooded_cmp(&t, &a[j]) < 0) {
            a[j + 1] = a[j];
            j--;
        }
        a[j + 1] = t;
    }
}

/* WorldTile.SpreadPool with drainParticles == false, LiveErosionDataTypes.cs:938-1010 */
static void nzo_spread_pool(float *pool, const float *height, int res, int x, int z) {
    size_t idx = (size_t)x * res + z;
    float hLand = height[idx], hWater = pool[idx];
    if (hWater <= 0.0f) return;
    float tHeight = hLand + hWater;
    /* up (0,1), right (1,0), down (0,-1), left (-1,0); SafeIdx clamps at the border (:585-589), so a border cell can
     * be its own neighbour */
    const int dx[4] = {0, 1, 0, -1}, dz[4] = {1, 0, -1, 0};
    nzo_flooded b[4];
    for (int e = 0; e < 4; e++) {
        int nx = x + dx[e], nz = z + dz[e];
        nx = nx < 0 ? 0 : (nx > res - 1 ? res - 1 : nx);
        nz = nz < 0 ? 0 : (nz > res - 1 ? res - 1 : nz);
        b[e].idx = nx * res + nz;
        b[e].height = height[b[e].idx];
        b[e].water = pool[b[e].idx];
    }
    nzo_flooded_sort4(b);
    for (int e = 0; e < 4; e++) {
        float fill = 0.0f;
        float diffV = tHeight - (b[e].height + b[e].water);
        if (hWater < 1E-3f) continue;
        if (b[e].water <= 0.0f && hLand >= b[e].height) { /* found a drain: all of the water goes there */
            pool[b[e].idx] = b[e].water + hWater;
            hWater = 0.0f;
            tHeight = hLand;
        } else if (diffV > 0.0f) {
            if (hWater <= 0.0f) continue;
            fill = minf_(0.25f * hWater, 0.25f * diffV);
            hWater -= fill;
            tHeight = hLand + hWater;
            pool[b[e].idx] = b[e].water + fill;
        } else if (diffV < 0.0f) {
            if (b[e].water <= 0.0f) continue;
            fill = minf_(0.25f * b[e].water, -0.25f * diffV);
            hWater += fill;
            tHeight = hLand + hWater;
            pool[b[e].idx] = b[e].water + (-1.0f * fill);
        }
    }
    pool[idx] = hWater;
}

/* PoolAutomataJob.Schedule / Execute, MultiThreadErosionJob.cs:264-327, drainParticles == false: per iteration four
 * colour passes (xoff, zoff); job k of a pass walks row z = 2k + zoff over x = xoff (+1 for odd k), step 2.  Rows of
 * one pass never touch a common cell, so the pass is deterministic however its jobs are scheduled. */
int nzo_pool_automata(float *pool, const float *height, int res, int iterations) {
    if (res < 2) return -1;
    for (int it = 0; it < iterations; it++)
        for (int xoff = 0; xoff < 2; xoff++)
            for (int zoff = 0; zoff < 2; zoff++) {
#pragma omp parallel for schedule(static)
                for (int k = 0; k < res / 2; k++) {
                    int offset = xoff + ((k % 2 != 0) ? 1 : 0);
                    int z = 2 * k + zoff;
                    for (int x = offset; x < res; x += 2)
                        if (pool[(size_t)x * res + z] > 0.0f) nzo_spread_pool(pool, height, res, x, z);
                }
            }
    return 0;
}

/* CropJob<ReadTileData,WriteTileData>.Execute, Filter/Sample/CropJob.cs:34-41.  ScheduleParallel (:43-59)
 * never assigns `Offset`, so the "centre crop" takes the top-left corner: out(x,z) = in(x+0, z+0) with the
 * read tile's clamp-to-edge (an output larger than the input repeats the last row / column). */
int nzo_crop(const float *input, int inputResolution, float *output, int outputResolution) {
    const int Offset = 0;
#pragma omp parallel for schedule(static)
    for (int z = 0; z < outputResolution; z++) {
        int zr = z + Offset;
        for (int x = 0; x < outputResolution; x++)
            output[(size_t)z * outputResolution + x] = input[tile_idx(x + Offset, zr, inputResolution, inputResolution)];
    }
    return 0;
}

/* CurveOperator.Apply, CurveJob.cs:69-80 */
int nzo_curve(float *src, float *tmp, const float *curve, int curveSize, int rows, int cols) {
    if (curveSize < 2) return -1;
#pragma omp parallel for schedule(static)
    for (int z = 0; z < rows; z++) {
        for (int x = 0; x < cols; x++) {
            int c = tile_idx(x, z, rows, cols);
            float v = src[c];
            float cl = maxf_(0.0f, minf_(1.0f, v)); /* math.clamp(v, 0, 1) = max(0, min(1, v)): a NaN cell reads as 1 */
            float rect = cl * (float)curveSize;
            float lowerIdx = minf_(floorf(rect), (float)(curveSize - 2));
            float left = curve[(int)lowerIdx], right = curve[(int)lowerIdx + 1];
            float value = lerpf_(left, right, (rect - lowerIdx));
            value = maxf_(0.0f, value);
            value = minf_(1.0f, value);
            tmp[c] = value;
        }
    }
    flush_write_slice(src, tmp, (size_t)rows * cols);
    return 0;
}

/* ThermalErosionFilter, Filter/Kernel/Blur/ThermalErosionFilter.cs:21-147 */
static inline void thermal_rectify(float *a, float *b, float maxDiff, float increment) { /* :84-99 */
    float diff = fabsf(*a - *b);
    if (diff > maxDiff) {
        float excess = diff - maxDiff;
        if (*a > *b) {
            *b += increment * excess;
            *a -= increment * excess;
        } else {
            *a += increment * excess;
            *b -= increment * excess;
        }
    }
}

int nzo_thermal_erosion(float *src, int resolution, float talus, float incrementRatio,
                        float meshHeightWidthRatio, int iterations) {
    float t = (talus / 90.0f) * 3.14159f / 2.0f;                           /* :131 */
    float maxDiff = (tanf(t) * meshHeightWidthRatio) / (float)resolution; /* :132 */
    int jobs = resolution / 2 - 1;                                        /* :137 */
    for (int i = 0; i < iterations; i++) {
        for (int flip = 0; flip < 4; flip++) {
#pragma omp parallel for schedule(static)
            for (int job = 0; job < jobs; job++) { /* Execute :102-120 */
                int offset = 1, z = job + 1;
                if (flip % 2 != 0) offset += 1;
                z *= 2;
                if (flip > 1) z -= 1;
                for (int x = offset; x < resolution - 1; x += 2) {
                    float *p0 = src + (size_t)z * resolution + x, *p2 = src + (size_t)(z + 1) * resolution + x;
                    float vx = p0[0], vy = p0[1], vz = p2[0], vw = p2[1];
                    thermal_rectify(&vx, &vy, maxDiff, incrementRatio); /* rectifyNeighborhood :74-81 */
                    thermal_rectify(&vx, &vz, maxDiff, incrementRatio);
                    thermal_rectify(&vx, &vw, maxDiff, incrementRatio);
                    thermal_rectify(&vy, &vz, maxDiff, incrementRatio);
                    thermal_rectify(&vy, &vw, maxDiff, incrementRatio);
                    thermal_rectify(&vz, &vw, maxDiff, incrementRatio);
                    p0[0] = vx; p0[1] = vy; p2[0] = vz; p2[1] = vw;
                }
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Metric pipeline (README.md:23-32): noise -> KernelFilterStage(iterations) -> FlowMapStage ->
 * ErosionKernelJob x E, every stage reference-shaped.
 * ---------------------------------------------------------------------------------------- */
int nzo_pipeline(float *data, float *tmp, int rows, int cols, int noiseType, float hurst,
                 float startingAmplitude, float stepdown, float detuneRate, int octaves, int xpos,
                 int zpos, int noiseSize, int filterType, int gaussIterations, int flowIterations,
                 float normMin, float normMax, int erosionIterations) {
    int rc = nzo_fractal(noiseType, data, rows, cols, hurst, startingAmplitude, stepdown,
                         detuneRate, octaves, xpos, zpos, noiseSize);
    if (rc) return rc;
    for (int i = 0; i < gaussIterations; i++) { /* KernelFilterStage.cs:35-41 */
        rc = nzo_kernel_filter(data, tmp, filterType, rows, cols);
        if (rc) return rc;
    }
    if (flowIterations > 0) {
        rc = nzo_flowmap(data, rows, cols, flowIterations, normMin, normMax);
        if (rc) return rc;
    }
    for (int i = 0; i < erosionIterations; i++) {
        rc = nzo_erosion_min(data, rows, cols);
        if (rc) return rc;
    }
    return 0;
}
