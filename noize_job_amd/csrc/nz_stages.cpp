// nz_stages.cpp -- the extern "C" stage entry points of libnoize_hip.so (one per reference job delegate or
// PipelineStage.Schedule body, include/noize_hip.h) and the launch planners behind them.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "nz_internal.hpp"

// ---------------------------------------------------------------------------------------------
// helpers shared by the stage entry points
// ---------------------------------------------------------------------------------------------
int32_t nz_check_stripe(const nz_stripe *st, int halo, int halo_below) {
    if (halo_below < 0) halo_below = halo;  // symmetric stencil
    NZ_REQUIRE(st, "stripe is NULL");
    NZ_REQUIRE(st->cols > 0 && st->rows > 0 && st->grows > 0, "stripe: non-positive extent");
    NZ_REQUIRE(st->pitch == 0 || st->pitch >= st->cols, "stripe: pitch < cols");
    NZ_REQUIRE(st->own0 >= 0 && st->own0 <= st->own1 && st->own1 <= st->rows, "stripe: owned rows outside buffer");
    NZ_REQUIRE(st->own0 + st->grow0 >= 0 && st->own1 + st->grow0 <= st->grows,
               "stripe: owned rows outside the global grid");
    // every row within `halo` of the owned rows must be in the buffer unless it is beyond the border
    int need_lo = st->own0 - halo, need_hi = st->own1 - 1 + halo_below;
    int dom_lo = -st->grow0, dom_hi = st->grows - 1 - st->grow0;
    if (need_lo < dom_lo) need_lo = dom_lo;
    if (need_hi > dom_hi) need_hi = dom_hi;
    NZ_REQUIRE(need_lo >= 0 && need_hi <= st->rows - 1, "stripe: %d ghost rows required above, %d below", halo,
               halo_below);
    return NZ_OK;
}

static int32_t check_res(int32_t resolution) {
    NZ_REQUIRE(resolution >= 1 && resolution <= 46340, "resolution %d out of range", resolution);
    return NZ_OK;
}

// FractalJob.CalcFractalNormValue, Noise/Fractal/Fractal.cs:31-40 (startingAmplitude is ignored)
static float calc_fractal_norm(float hurst, int octaves) {
    float G = exp2f(-hurst);
    float a = 1.0f, t = 0.0f;
    for (int i = 0; i < octaves; i++) {
        t += a * 1.0f;
        a *= G;
    }
    return t;
}

static int32_t fractal_impl(nz_ctx *ctx, hipStream_t stream, int noiseType, float *dst, int rows, int cols, int pitch,
                            float hurst,
                            float amp, float stepdown, float detune, int octaves, int xpos, int zpos_first_row,
                            int noiseSize, int count = 1, size_t bstride = 0, const int32_t *positions = nullptr) {
    NZ_REQUIRE(dst, "src is NULL");
    NZ_REQUIRE(noiseType >= 0 && noiseType <= NZ_NOISE_DOMAIN_ROTATED_SIMPLEX, "unknown noise type %d", noiseType);
    NZ_REQUIRE(octaves >= 0, "octaves < 0");
    NZ_REQUIRE(noiseSize != 0, "noiseSize == 0");
    nz_fractal_params p;
    p.posx = (float)xpos;  // SetPosition, Fractal.cs:109-112
    p.posz = (float)zpos_first_row;
    p.noise_size = (float)noiseSize;
    p.G = exp2f(-hurst);
    p.amp = amp;
    p.stepdown = stepdown;
    p.detune_rate = detune;
    p.norm = calc_fractal_norm(hurst, octaves);
    p.octaves = octaves;
    // largest |f| of FractalGenerator.NoiseValue's recurrence (Fractal.cs:121-127): lets a kernel decide once
    // per row whether every octave stays inside the range its lattice tables cover
    p.fmax = 0.0f;
    float f = 1.0f, det = 0.0f;
    for (int i = 0; i < octaves; i++) {
        if (!(fabsf(f) <= p.fmax)) p.fmax = fabsf(f);
        det += detune;
        f *= (stepdown - det);
    }
    return nz_launch_fractal(stream, noiseType, dst, rows, cols, pitch, p, ctx->d_rgrad, ctx->d_simplex, count, bstride,
                             positions);
}

// SeparableKernelFilter tables, Filter/Kernel/KernelJob.cs:97-136.  Gaussian bodies are
// exp(-i^2/2s^2)/sum in double rounded to fp32, which reproduces the reference literals
// (tests/golden/gauss_tables.json).
static void gauss_coeffs(double sigma, int width, float *out) {
    int o = (width - 1) / 2;
    double w[NZ_MAX_KSIZE], sum = 0.0;
    for (int i = 0; i < width; i++) {
        double d = (double)(i - o);
        w[i] = exp(-(d * d) / (2.0 * sigma * sigma));
        sum += w[i];
    }
    for (int i = 0; i < width; i++) out[i] = (float)(w[i] / sum);
}

static int32_t filter_taps(int32_t filter, nz_kernel_taps *t) {
    memset(t, 0, sizeof *t);
    auto set3 = [&](float a0, float a1, float a2, float b0, float b1, float b2, float f) {
        t->kx[0] = a0; t->kx[1] = a1; t->kx[2] = a2;
        t->kz[0] = b0; t->kz[1] = b1; t->kz[2] = b2;
        t->factor = f;
        t->ksize = 3;
    };
    switch (filter) {
        case NZ_GAUSS9_S1: case NZ_GAUSS7_S1: case NZ_GAUSS5_S1: case NZ_GAUSS3_S1:
        case NZ_GAUSS9_S2: case NZ_GAUSS7_S2: case NZ_GAUSS5_S2: case NZ_GAUSS3_S2: {
            static const int sizes[4] = {9, 7, 5, 3};
            int w = sizes[filter & 3];
            double sigma = filter >= NZ_GAUSS9_S2 ? 2.0 : 1.0;
            gauss_coeffs(sigma, w, t->kx);
            gauss_coeffs(sigma, w, t->kz);
            t->factor = 1.0f;
            t->ksize = w;
            return NZ_OK;
        }
        case NZ_SMOOTH3: set3(1, 1, 1, 1, 1, 1, 1.0f / 3.0f); return NZ_OK;          // KernelJob.cs:107-108
        case NZ_SOBEL3_HORIZONTAL: set3(-1, 0, 1, 1, 2, 1, 1.0f); return NZ_OK;       // :110-116
        case NZ_SOBEL3_VERTICAL: set3(1, 2, 1, 1, 0, -1, 1.0f); return NZ_OK;         // :117-122
        case NZ_PREWITT3_HORIZONTAL: set3(1, 0, -1, 1, 1, 1, 1.0f); return NZ_OK;     // :124-130
        case NZ_PREWITT3_VERTICAL: set3(1, 1, 1, -1, 0, 1, 1.0f); return NZ_OK;       // :131-136
        case NZ_SOBEL3_2D:
            nz_set_error("Sobel3_2D is not a single separable pass (ScheduleReduce, KernelJob.cs:187-215)");
            return NZ_ERR_UNSUPPORTED;
    }
    nz_set_error("unknown KernelFilterType %d", filter);
    return NZ_ERR_INVALID;
}

// BlurHelper.limitWidth, Filter/Kernel/Blur/BlurKernels.cs:29-36
static int limit_width(int width) {
    if (width % 2 == 0) width += 1;
    if (width > 25) width = 25;
    return width < 3 ? 3 : width;
}

// GaussFilter.Schedule, Filter/Kernel/Blur/BlurJob.cs:11-21: body of limitWidth(width) taps, pass run
// with kernelSize = width as given
static int32_t gauss_taps(int32_t width, int32_t sigma, nz_kernel_taps *t) {
    memset(t, 0, sizeof *t);
    NZ_REQUIRE(sigma >= 0 && sigma <= 15, "GaussSigma %d out of range", sigma);
    int w = limit_width(width);
    NZ_REQUIRE(width >= 1 && width <= w, "gauss width %d indexes outside its %d-tap kernel", width, w);
    gauss_coeffs(0.5 * (double)(sigma + 1), w, t->kx);
    memcpy(t->kz, t->kx, sizeof t->kx);
    t->factor = 1.0f;
    t->ksize = width;
    return NZ_OK;
}

// SmoothFilter.Schedule BlurJob.cs:34-44; SmoothBlur.GetKernel BlurKernels.cs:39-43
static int32_t smooth_taps(int32_t width, nz_kernel_taps *t) {
    memset(t, 0, sizeof *t);
    NZ_REQUIRE(width >= 1 && width <= NZ_MAX_KSIZE, "smooth width %d out of range [1,25]", width);
    for (int i = 0; i < width; i++) t->kx[i] = t->kz[i] = 1.0f / (float)width;
    t->factor = 1.0f;
    t->ksize = width;
    return NZ_OK;
}

static int conv_tcap(int ksize) {
    int hw = nz_conv_max_fused(ksize);
    if (hw == 0) return 0;
    int cap;
    switch (ksize) {  // fusion depth per launch (tuned on MI355X; other depths lost every measurement, HISTORY.md)
        case 3: cap = 6; break;
        case 5: cap = 5; break;
        case 7: cap = 3; break;
        default: cap = 3; break;
    }
    return cap < hw ? cap : hw;
}

// `iterations` applications of (X pass, Z pass).  swapped == nullptr: the result must be back in `src`, so the
// applications are grouped into an even number of fused launches that ping-pong src <-> tmp (or an odd number and a
// copy).  Otherwise (READ / WRITE pair, nz_rw_tile) the launch count is free and *swapped tells whether the result
// is in `tmp`.
static int32_t conv_iterations(nz_ctx *ctx, float *src, float *tmp, const nz_geom &g, const nz_kernel_taps &t,
                               int iterations, bool *swapped = nullptr) {
    NZ_REQUIRE(src && tmp && src != tmp, "src/tmp must be two distinct planes");
    NZ_REQUIRE(iterations >= 1, "iterations < 1");
    if (swapped) *swapped = false;
    int cap = (t.ksize & 1) ? conv_tcap(t.ksize) : 0;
    // a small grid is served by one round of workgroups whatever the depth: one launch less beats the deeper halo
    // (512^2 tiles, READ / WRITE pair, 17 applications as 6 + 6 + 5 instead of 5 + 4 + 4 + 4: 11 100 -> 11 700 tiles/s)
    // ... and a grid of the reference's own tile sizes (256^2 .. 512^2: a hundred 64-row tiles whatever the depth, all resident
    // at once) waits for the LATENCY of its dependent applications, not for throughput: nine applications per launch, 17 = 9 + 8
    // in two launches instead of three (round 5: Gauss5 x17 at 256^2 / 512^2 52 -> 42 / 43 us, 13 100 -> 13 900 tiles/s one at a
    // time; from 1024^2 on the deeper halo costs more than the launch it saves: 55 -> 59 us)
    if (t.ksize == 5 && cap == 5 && nz_conv_small_grid(t.ksize, g))
        cap = nz_conv_tiny_grid(t.ksize, g) ? 9 : 6;
    if (nz_conv_has_wide(t.ksize)) {  // one launch per application, ping-pong, copy back after an odd count
        float *cur = src, *other = tmp;
        for (int i = 0; i < iterations; i++) {
            // (a READ / WRITE pair ends with this launch -- odd count or not, nothing is copied back: see below)
            if (swapped && i == iterations - 1) nz_ctx_arm_last_launch(ctx);
            NZ_TRY_(launch_on_ctx(ctx, g, [&](hipStream_t st, const nz_geom &gb) {
                return nz_launch_conv_wide(st, cur, other, gb, t);
            }));
            float *s = cur; cur = other; other = s;
        }
        if (cur != src && swapped) {
            *swapped = true;
        } else if (cur != src) {
            nz_ctx_arm_last_launch(ctx);  // the copy back is the last operation
            NZ_TRY_(launch_on_ctx(ctx, g, [&](hipStream_t st, const nz_geom &gb) {
                size_t off = (size_t)gb.or0 * gb.pitch;
                return nz_launch_copy(st, src + off, tmp + off, nz_geom_span(gb));
            }));
        }
        return NZ_OK;
    }
    if (cap > 0 && iterations == 1) {  // the delegate's single application: one launch into tmp, copy back
        if (swapped) nz_ctx_arm_last_launch(ctx);
        NZ_TRY_(launch_on_ctx(ctx, g, [&](hipStream_t st, const nz_geom &gb) {
            return nz_launch_conv_fused(st, src, tmp, gb, t, 1);
        }));
        if (swapped) {
            *swapped = true;
            return NZ_OK;
        }
        nz_ctx_arm_last_launch(ctx);  // the copy back is the last operation
        return launch_on_ctx(ctx, g, [&](hipStream_t st, const nz_geom &gb) {
            size_t off = (size_t)gb.or0 * gb.pitch;
            return nz_launch_copy(st, src + off, tmp + off, nz_geom_span(gb));
        });
    }
    if (cap == 0) {  // even or out-of-table sizes: the two passes as launched by the reference
        if (g.count > 1) {  // no batched form of the generic passes: grid by grid
            nz_geom one = g;
            one.count = 1;
            one.bstride = 0;
            for (int b = 0; b < g.count; b++)
                NZ_TRY_(conv_iterations(ctx, src + b * g.bstride, tmp + b * g.bstride, one, t, iterations, nullptr));
            return NZ_OK;
        }
        for (int i = 0; i < iterations; i++) {
            int32_t rc = launch_on_ctx(ctx, g, [&](hipStream_t st, const nz_geom &gb) {
                return nz_launch_conv_pass_x(st, src, tmp, gb, t);
            });
            if (rc) return rc;
            rc = launch_on_ctx(ctx, g, [&](hipStream_t st, const nz_geom &gb) {
                return nz_launch_conv_pass_z(st, tmp, src, gb, t);
            });
            if (rc) return rc;
        }
        return NZ_OK;
    }
    int L = (iterations + cap - 1) / cap;
    if (!swapped && (L & 1) && L + 1 <= iterations) L += 1;  // an even count leaves the result in src
    int base = iterations / L, rem = iterations % L;
    // three or more launches of a 5..9-tap kernel run as ONE grid with tile-level dependencies (nz_filter.hip,
    // conv_chain_kernel): Gauss5 x17 0.212 -> 0.198 ms at 4096^2.  Two launches gain nothing (Gauss9 x6 -3 %, Gauss5 x6
    // +2.5 %, two single applications +15 %: a tile's poll and its sc1 accesses cost what the missing launch boundary
    // saves), and the 64-row tiles of the 3-tap kernels lose (x6: 0.131 vs 0.066 ms).  NZ_CONV_CHAIN=0: never;
    // NZ_CONV_CHAIN=2: whenever there are two launches or more (the test suite runs the parity tests under it).
    static const int chain_mode = getenv("NZ_CONV_CHAIN") ? atoi(getenv("NZ_CONV_CHAIN")) : 1;
    // (where the row-streaming form of a launch applies -- big grids, 3 / 5 taps -- plain launches of it are faster still)
    const bool streamed = nz_conv_stream_wanted(g, t.ksize, base + (rem ? 1 : 0)) && nz_conv_stream_wanted(g, t.ksize, base);
    // (A small grid -- fewer than ~7 M cells, 64-row tiles -- chains from TWO launches on since round 5: with the ticket gone and
    // four rows per thread the chained grid wins there too.  Gauss5 x17 2048^2 87.2 -> 67.9 us, 2560^2 110.9 -> 95.0; 1024^2 by
    // itself 45.6 -> 47.3 us, in a tile's pipeline 10 550 -> 11 020 tiles/s (one launch to enqueue and to start instead of three);
    // 512^2 18 000 -> 18 100.  Rounds 3 - 4 kept separate launches here: 512^2 55.7 against 51.7 us then.)
    const bool chain_on = !streamed && !ctx->chain_off &&
                          (chain_mode == 2 ? L >= 2 : (chain_mode == 1 && t.ksize >= 5 && L >= (nz_conv_small_grid(t.ksize, g) ? 2 : 3)));
    if (chain_on && L <= 8 && g.count == 1 && (size_t)g.rows * g.pitch * 4 < ((size_t)1 << 32) &&
        (swapped || !(L & 1))) {
        // T must not decrease along the chain: a tile of launch l + 1 waits for the launch-l tiles whose INTERIOR meets
        // its input window (read after write); its own stores land in the plane launch l reads, and the launch-l tiles
        // that read that region are all among the awaited ones only while H(l) <= H(l + 1) (write after read)
        int Ts[8];
        for (int i = 0; i < L; i++) Ts[i] = base + (i >= L - rem ? 1 : 0);
        int *flags = nullptr;
        unsigned epoch = 0, *err_host = nullptr, *err_epoch = nullptr;
        NZ_TRY_(nz_ctx_chain_state(ctx, (size_t)nz_conv_chain_items(t.ksize, g, Ts, L), &flags, &epoch, &err_host, &err_epoch));
        nz_ctx_arm_last_launch(ctx);  // (one launch; no copy follows in either form: an even count or a pair)
        NZ_TRY_(nz_launch_conv_chain(ctx->stream, src, tmp, g, t, Ts, L, flags, epoch, err_host, err_epoch));
        if (swapped) *swapped = (L & 1) != 0;
        return NZ_OK;
    }
    float *cur = src, *other = tmp;
    for (int i = 0; i < L; i++) {
        int T = base + (i < rem ? 1 : 0);
        // the last launch is the stage's last operation unless a copy back follows (single plane, odd count)
        if (i == L - 1 && (swapped || !(L & 1))) nz_ctx_arm_last_launch(ctx);
        int32_t rc = launch_on_ctx(ctx, g, [&](hipStream_t st, const nz_geom &gb) {
            return nz_launch_conv_fused(st, cur, other, gb, t, T);
        });
        if (rc) return rc;
        float *s = cur; cur = other; other = s;
    }
    if (cur != src && swapped) {
        *swapped = true;
        return NZ_OK;
    }
    if (cur != src) {  // odd count (only when cap == 1): copy back
        nz_ctx_arm_last_launch(ctx);
        return launch_on_ctx(ctx, g, [&](hipStream_t st, const nz_geom &gb) {
            size_t off = (size_t)gb.or0 * gb.pitch;
            return nz_launch_copy(st, src + off, tmp + off, nz_geom_span(gb));
        });
    }
    return NZ_OK;
}

static int32_t erosion_iterations(nz_ctx *ctx, float *src, float *tmp, const nz_geom &g, int iterations,
                                  bool *swapped = nullptr) {
    NZ_REQUIRE(src && tmp && src != tmp, "src/tmp must be two distinct planes");
    NZ_REQUIRE(iterations >= 1, "iterations < 1");
    if (swapped) *swapped = false;
    if (iterations == 1 && !swapped) {
        // ErosionKernelJob.ScheduleSeries KernelJob.cs:318-335: min-X then min-Z (size 3) = the min over
        // {x-1,x} x {z-1,z}; one launch into tmp, then the copy back that stands for the flush
        NZ_TRY_(launch_on_ctx(ctx, g, [&](hipStream_t st, const nz_geom &gb) {
            return nz_launch_erosion_fused(st, src, tmp, gb, 1);
        }));
        nz_ctx_arm_last_launch(ctx);  // the copy back is the last operation
        return launch_on_ctx(ctx, g, [&](hipStream_t st, const nz_geom &gb) {
            size_t off = (size_t)gb.or0 * gb.pitch;
            return nz_launch_copy(st, src + off, tmp + off, nz_geom_span(gb));
        });
    }
    // an even number of launches ends in src; when every launch already holds its one iteration and the
    // count is odd, the last result is copied back instead
    int cap = nz_erosion_max_fused();
    int L = (iterations + cap - 1) / cap;
    if (!swapped) {
        if (L < 2) L = 2;
        if ((L & 1) && L + 1 <= iterations) L += 1;
    }
    int base = iterations / L, rem = iterations % L;
    float *cur = src, *other = tmp;
    for (int i = 0; i < L; i++) {
        int E = base + (i < rem ? 1 : 0);
        if (i == L - 1 && (swapped || !(L & 1))) nz_ctx_arm_last_launch(ctx);  // no copy back follows
        int32_t rc = launch_on_ctx(ctx, g, [&](hipStream_t st, const nz_geom &gb) {
            return nz_launch_erosion_fused(st, cur, other, gb, E);
        });
        if (rc) return rc;
        float *s = cur; cur = other; other = s;
    }
    if (cur != src && swapped) {
        *swapped = true;
        return NZ_OK;
    }
    if (cur != src) {
        nz_ctx_arm_last_launch(ctx);  // the copy back is the last operation
        return launch_on_ctx(ctx, g, [&](hipStream_t st, const nz_geom &gb) {
            size_t off = (size_t)gb.or0 * gb.pitch;
            return nz_launch_copy(st, src + off, tmp + off, nz_geom_span(gb));
        });
    }
    return NZ_OK;
}

// what nz_comm.cpp (the sharded plan) needs of the above
int32_t nz_filter_taps(int32_t filter, nz_kernel_taps *t) { return filter_taps(filter, t); }
int nz_conv_tcap(int ksize) { return conv_tcap(ksize); }
int32_t nz_fractal_rows(nz_ctx *ctx, hipStream_t stream, int noiseType, float *dst, int rows, int cols, int pitch, float hurst,
                        float amp, float stepdown, float detune, int octaves, int xpos, int zpos_first_row, int noiseSize) {
    return fractal_impl(ctx, stream, noiseType, dst, rows, cols, pitch, hurst, amp, stepdown, detune, octaves, xpos,
                        zpos_first_row, noiseSize);
}

#define NZ_BEGIN(ctx, dep)                   \
    do {                                     \
        int32_t rc_ = nz_ctx_begin(ctx, dep); \
        if (rc_) return rc_;                 \
    } while (0)

#define NZ_TRY(expr)              \
    do {                          \
        int32_t rc_ = (expr);     \
        if (rc_) return rc_;      \
    } while (0)

// ---------------------------------------------------------------------------------------------
// noise
// ---------------------------------------------------------------------------------------------
extern "C" int32_t nz_fractal(nz_ctx *ctx, int32_t noiseType, float *src, int32_t resolution, float hurst,
                              float startingAmplitude, float stepdown, float detuneRate, int32_t octaves,
                              int32_t xpos, int32_t zpos, int32_t noiseSize, nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_TRY(check_res(resolution));
    nz_ctx_handle_rides(ctx, out != nullptr);  // one launch, and nothing behind it: the handle rides on it
    nz_ctx_arm_last_launch(ctx);
    NZ_TRY(fractal_impl(ctx, ctx->stream, noiseType, src, resolution, resolution, resolution, hurst, startingAmplitude,
                        stepdown, detuneRate, octaves, xpos, zpos, noiseSize));
    return nz_ctx_finish(ctx, out);
}

extern "C" int32_t nz_fractal_stripe(nz_ctx *ctx, int32_t noiseType, float *buf, const nz_stripe *st, float hurst,
                                     float startingAmplitude, float stepdown, float detuneRate, int32_t octaves,
                                     int32_t xpos, int32_t zpos, int32_t noiseSize, nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_TRY(nz_check_stripe(st, 0));
    NZ_REQUIRE(buf, "buf is NULL");
    int pitch = st->pitch > 0 ? st->pitch : st->cols;
    int rows = st->own1 - st->own0;
    if (rows > 0) {
        NZ_TRY(fractal_impl(ctx, ctx->stream, noiseType, buf + (size_t)st->own0 * pitch, rows, st->cols, pitch, hurst,
                            startingAmplitude, stepdown, detuneRate, octaves, xpos, zpos + st->grow0 + st->own0,
                            noiseSize));
    }
    return nz_ctx_finish(ctx, out);
}

// ---------------------------------------------------------------------------------------------
// separable filters
// ---------------------------------------------------------------------------------------------
// SeparableKernelFilter.ScheduleReduce<RootSumSquaresTiles> (KernelJob.cs:187-215) for Sobel3_2D: the horizontal
// filter on src, the vertical filter on a copy of the ORIGINAL plane, then src = sqrt(src^2 + copy^2); both with
// kernelFactor 1.  (The reference takes its copy on the host at schedule time, i.e. before `dependency` has run --
// the README lists the filter as broken; here the copy is ordered after `dep` like every other job.)
static int32_t edge_2d(nz_ctx *ctx, float *src, float *tmp, int resolution, int iterations, int filterH, int filterV) {
    NZ_REQUIRE(src && tmp && src != tmp, "src/tmp must be two distinct planes");
    NZ_REQUIRE(iterations >= 1, "iterations < 1");
    size_t n = (size_t)resolution * resolution;
    float *original = nullptr;
    NZ_TRY(nz_ctx_scratch(ctx, n, &original));
    nz_kernel_taps th, tv;
    NZ_TRY(filter_taps(filterH, &th));
    NZ_TRY(filter_taps(filterV, &tv));
    th.factor = tv.factor = 1.0f;
    nz_geom g = nz_geom_tile(resolution);
    for (int i = 0; i < iterations; i++) {
        NZ_TRY(nz_launch_copy(ctx->stream, original, src, n));
        NZ_TRY(conv_iterations(ctx, src, tmp, g, th, 1));
        NZ_TRY(conv_iterations(ctx, original, tmp, g, tv, 1));
        NZ_TRY(nz_launch_reduce(ctx->stream, 2 /* ROOTSUMSQUARES */, src, original, n));
    }
    return NZ_OK;
}

extern "C" int32_t nz_kernel_filter_stage(nz_ctx *ctx, float *src, float *tmp, int32_t filter, int32_t iterations,
                                          int32_t resolution, nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_TRY(check_res(resolution));
    if (filter == NZ_SOBEL3_2D) {
        NZ_TRY(edge_2d(ctx, src, tmp, resolution, iterations, NZ_SOBEL3_HORIZONTAL, NZ_SOBEL3_VERTICAL));
        return nz_ctx_finish(ctx, out);
    }
    nz_kernel_taps t;
    NZ_TRY(filter_taps(filter, &t));
    nz_ctx_handle_rides(ctx, out != nullptr);  // (conv_iterations arms its last launch unless a copy back follows it)
    NZ_TRY(conv_iterations(ctx, src, tmp, nz_geom_tile(resolution), t, iterations));
    return nz_ctx_finish(ctx, out);
}

// Edge1DFilter.Schedule / Edge2DFilter.Schedule, Filter/Kernel/Edge/EdgeJob.cs:11-44 (kernels: EdgeDetection.cs:23-84,
// the same numbers as the Sobel / Prewitt entries of KernelFilterType), kernelFactor 1
extern "C" int32_t nz_edge_1d_filter(nz_ctx *ctx, float *src, float *tmp, int32_t algo, int32_t dir, int32_t resolution,
                                     nz_handle dep, nz_handle *out) {
    NZ_REQUIRE((algo == 0 || algo == 1) && (dir == 0 || dir == 1), "EdgeAlgorithm %d / EdgeDirection %d out of range",
               algo, dir);
    int filter = algo == 0 ? (dir == 0 ? NZ_SOBEL3_HORIZONTAL : NZ_SOBEL3_VERTICAL)
                           : (dir == 0 ? NZ_PREWITT3_HORIZONTAL : NZ_PREWITT3_VERTICAL);
    return nz_kernel_filter_stage(ctx, src, tmp, filter, 1, resolution, dep, out);
}

extern "C" int32_t nz_edge_2d_filter(nz_ctx *ctx, float *src, float *tmp, int32_t algo, int32_t resolution,
                                     nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_TRY(check_res(resolution));
    NZ_REQUIRE(algo == 0 || algo == 1, "EdgeAlgorithm %d out of range", algo);
    NZ_TRY(edge_2d(ctx, src, tmp, resolution, 1, algo == 0 ? NZ_SOBEL3_HORIZONTAL : NZ_PREWITT3_HORIZONTAL,
                   algo == 0 ? NZ_SOBEL3_VERTICAL : NZ_PREWITT3_VERTICAL));
    return nz_ctx_finish(ctx, out);
}

extern "C" int32_t nz_kernel_filter(nz_ctx *ctx, float *src, float *tmp, int32_t filter, int32_t resolution,
                                    nz_handle dep, nz_handle *out) {
    return nz_kernel_filter_stage(ctx, src, tmp, filter, 1, resolution, dep, out);
}

extern "C" int32_t nz_gauss_blur_stage(nz_ctx *ctx, float *src, float *tmp, int32_t width, int32_t sigma,
                                       int32_t iterations, int32_t resolution, nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_TRY(check_res(resolution));
    nz_kernel_taps t;
    NZ_TRY(gauss_taps(width, sigma, &t));
    nz_ctx_handle_rides(ctx, out != nullptr);  // (conv_iterations arms its last launch unless a copy back follows it)
    NZ_TRY(conv_iterations(ctx, src, tmp, nz_geom_tile(resolution), t, iterations));
    return nz_ctx_finish(ctx, out);
}

extern "C" int32_t nz_gauss_filter(nz_ctx *ctx, float *src, float *tmp, int32_t width, int32_t sigma,
                                   int32_t resolution, nz_handle dep, nz_handle *out) {
    return nz_gauss_blur_stage(ctx, src, tmp, width, sigma, 1, resolution, dep, out);
}

extern "C" int32_t nz_smooth_blur_stage(nz_ctx *ctx, float *src, float *tmp, int32_t width, int32_t iterations,
                                        int32_t resolution, nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_TRY(check_res(resolution));
    nz_kernel_taps t;
    NZ_TRY(smooth_taps(width, &t));
    nz_ctx_handle_rides(ctx, out != nullptr);  // (conv_iterations arms its last launch unless a copy back follows it)
    NZ_TRY(conv_iterations(ctx, src, tmp, nz_geom_tile(resolution), t, iterations));
    return nz_ctx_finish(ctx, out);
}

extern "C" int32_t nz_smooth_filter(nz_ctx *ctx, float *src, float *tmp, int32_t width, int32_t resolution,
                                    nz_handle dep, nz_handle *out) {
    return nz_smooth_blur_stage(ctx, src, tmp, width, 1, resolution, dep, out);
}

extern "C" int32_t nz_separable_series(nz_ctx *ctx, float *src, float *tmp, int32_t resolution, int32_t kernelSize,
                                       const float *kernelX, const float *kernelZ, float kernelFactor,
                                       nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_TRY(check_res(resolution));
    NZ_REQUIRE(kernelX && kernelZ, "kernel body is NULL");
    NZ_REQUIRE(kernelSize >= 1 && kernelSize <= NZ_MAX_KSIZE, "kernelSize %d out of range [1,25]", kernelSize);
    nz_kernel_taps t;
    memset(&t, 0, sizeof t);
    int used = 2 * ((kernelSize - 1) / 2) + 1;  // taps an odd or even kernelSize actually touches
    memcpy(t.kx, kernelX, used * sizeof(float));
    memcpy(t.kz, kernelZ, used * sizeof(float));
    t.factor = kernelFactor;
    t.ksize = kernelSize;
    NZ_TRY(conv_iterations(ctx, src, tmp, nz_geom_tile(resolution), t, 1));
    return nz_ctx_finish(ctx, out);
}

extern "C" int32_t nz_erosion_stage(nz_ctx *ctx, float *src, float *tmp, int32_t iterations, int32_t resolution,
                                    nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_TRY(check_res(resolution));
    nz_ctx_handle_rides(ctx, out != nullptr);  // (as conv_iterations)
    NZ_TRY(erosion_iterations(ctx, src, tmp, nz_geom_tile(resolution), iterations));
    return nz_ctx_finish(ctx, out);
}

extern "C" int32_t nz_erosion_kernel(nz_ctx *ctx, float *src, int32_t resolution, nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_TRY(check_res(resolution));
    // the reference allocates its own TempJob plane (KernelJob.cs:327); here it is ctx scratch
    float *tmp = nullptr;
    NZ_TRY(nz_ctx_scratch(ctx, (size_t)resolution * resolution, &tmp));
    nz_ctx_handle_rides(ctx, out != nullptr);
    NZ_TRY(erosion_iterations(ctx, src, tmp, nz_geom_tile(resolution), 1));
    return nz_ctx_finish(ctx, out);
}

extern "C" int32_t nz_erosion_max_fused_iterations(void) { return nz_erosion_max_fused(); }

extern "C" int32_t nz_kernel_filter_max_fused(int32_t filter) {
    nz_kernel_taps t;
    if (filter_taps(filter, &t) != NZ_OK) return 0;
    int cap = conv_tcap(t.ksize);
    return cap < 1 ? 1 : cap;
}

extern "C" int32_t nz_kernel_filter_halo_rows(int32_t filter, int32_t iterations) {
    nz_kernel_taps t;
    if (filter_taps(filter, &t) != NZ_OK) return -1;
    return iterations * ((t.ksize - 1) / 2);
}

extern "C" int32_t nz_kernel_filter_stripe(nz_ctx *ctx, const float *src, float *dst, const nz_stripe *st,
                                           int32_t filter, int32_t iterations, nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    nz_kernel_taps t;
    NZ_TRY(filter_taps(filter, &t));
    NZ_REQUIRE(src && dst && src != dst, "src/dst must be two distinct planes");
    NZ_REQUIRE(iterations >= 1 && iterations <= nz_conv_max_fused(t.ksize), "iterations %d cannot be fused",
               iterations);
    NZ_TRY(nz_check_stripe(st, iterations * ((t.ksize - 1) / 2)));
    NZ_TRY(nz_launch_conv_fused(ctx->stream, src, dst, nz_geom_from_stripe(*st), t, iterations));
    return nz_ctx_finish(ctx, out);
}

extern "C" int32_t nz_erosion_stripe(nz_ctx *ctx, const float *src, float *dst, const nz_stripe *st,
                                     int32_t iterations, nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_REQUIRE(src && dst && src != dst, "src/dst must be two distinct planes");
    NZ_REQUIRE(iterations >= 1 && iterations <= nz_erosion_max_fused(), "iterations %d cannot be fused", iterations);
    NZ_TRY(nz_check_stripe(st, iterations, 0));  // the min window reaches upwards only
    NZ_TRY(nz_launch_erosion_fused(ctx->stream, src, dst, nz_geom_from_stripe(*st), iterations));
    return nz_ctx_finish(ctx, out);
}

// ---------------------------------------------------------------------------------------------
// flow map
// ---------------------------------------------------------------------------------------------
extern "C" int32_t nz_flush_write_slice(nz_ctx *ctx, float *write_, const float *read_, size_t n_floats, nz_handle dep,
                                        nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_REQUIRE(write_ && read_, "write/read is NULL");
    NZ_REQUIRE(write_ != read_, "write and read are the same slice");
    nz_ctx_handle_rides(ctx, out != nullptr);
    nz_ctx_arm_last_launch(ctx);
    if (n_floats) NZ_TRY(nz_launch_copy(ctx->stream, write_, read_, n_floats));
    return nz_ctx_finish(ctx, out);
}

extern "C" int32_t nz_fill_array(nz_ctx *ctx, float *data, int32_t resolution, float value, nz_handle dep,
                                 nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_TRY(check_res(resolution));
    NZ_REQUIRE(data, "data is NULL");
    nz_ctx_handle_rides(ctx, out != nullptr);
    nz_ctx_arm_last_launch(ctx);
    NZ_TRY(nz_launch_fill(ctx->stream, data, (size_t)resolution * resolution, value));
    return nz_ctx_finish(ctx, out);
}

extern "C" int32_t nz_flowmap_compute_flow(nz_ctx *ctx, const float *src, const float *waterMap, float *flowMapN,
                                           float *flowMapN__buff, float *flowMapS, float *flowMapS__buff,
                                           float *flowMapE, float *flowMapE__buff, float *flowMapW,
                                           float *flowMapW__buff, int32_t resolution, nz_handle dep,
                                           nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_TRY(check_res(resolution));
    NZ_REQUIRE(src && waterMap && flowMapN && flowMapS && flowMapE && flowMapW, "plane is NULL");
    // The reference writes the __buff planes and flushes them back (FlowMapJob.cs:70-77).  Each
    // cell reads only its own flux, so the update is done in place and the buffers stay untouched.
    (void)flowMapN__buff; (void)flowMapS__buff; (void)flowMapE__buff; (void)flowMapW__buff;
    NZ_TRY(nz_launch_flow_step(ctx->stream, src, waterMap, flowMapN, flowMapS, flowMapE, flowMapW, flowMapN, flowMapS,
                               flowMapE, flowMapW, nz_geom_tile(resolution)));
    return nz_ctx_finish(ctx, out);
}

extern "C" int32_t nz_flowmap_update_water(nz_ctx *ctx, float *waterMap, float *waterMap__buff,
                                           const float *flowMapN, const float *flowMapS, const float *flowMapE,
                                           const float *flowMapW, int32_t resolution, nz_handle dep,
                                           nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_TRY(check_res(resolution));
    NZ_REQUIRE(waterMap && flowMapN && flowMapS && flowMapE && flowMapW, "plane is NULL");
    (void)waterMap__buff;
    NZ_TRY(nz_launch_water_step(ctx->stream, waterMap, waterMap, flowMapN, flowMapS, flowMapE, flowMapW,
                                nz_geom_tile(resolution)));
    return nz_ctx_finish(ctx, out);
}

extern "C" int32_t nz_flowmap_write_values(nz_ctx *ctx, float *src, const float *flowMapN, const float *flowMapS,
                                           const float *flowMapE, const float *flowMapW, int32_t resolution,
                                           nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_TRY(check_res(resolution));
    NZ_REQUIRE(src && flowMapN && flowMapS && flowMapE && flowMapW, "plane is NULL");
    NZ_TRY(nz_launch_velocity(ctx->stream, src, flowMapN, flowMapS, flowMapE, flowMapW, nz_geom_tile(resolution), 0,
                              0.0f, 1.0f));
    return nz_ctx_finish(ctx, out);
}

extern "C" int32_t nz_map_normalize_values(nz_ctx *ctx, float *src, float *tmp, const float *args,
                                           int32_t resolution, nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_TRY(check_res(resolution));
    NZ_REQUIRE(src && args, "src/args is NULL");
    (void)tmp;  // element-wise: done in place, no flush copy
    nz_ctx_handle_rides(ctx, out != nullptr);  // one launch, nothing behind it
    nz_ctx_arm_last_launch(ctx);
    NZ_TRY(nz_launch_normalize(ctx->stream, src, src, (size_t)resolution * resolution, args[0], args[2]));
    return nz_ctx_finish(ctx, out);
}

// GetMapRangeJob.Schedule, Filter/NormalizeJob.cs:45-53: res = DEVICE {min, max, max - min}
extern "C" int32_t nz_get_map_range(nz_ctx *ctx, const float *map, size_t n_floats, float *res, float lim_min, float lim_max,
                                    nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_REQUIRE(map && res, "map/res is NULL");
    NZ_REQUIRE(n_floats >= 1, "empty map");
    float *scratch = nullptr;
    NZ_TRY(nz_ctx_scratch(ctx, nz_map_range_scratch_floats(), &scratch));
    NZ_TRY(nz_launch_map_range(ctx->stream, map, n_floats, lim_min, lim_max, res, scratch));
    return nz_ctx_finish(ctx, out);
}

// MapNormalizeValuesDelegate with `args` left in device memory by nz_get_map_range (the reference's NativeSlice<float> args)
extern "C" int32_t nz_map_normalize_values_dev(nz_ctx *ctx, float *src, float *tmp, const float *args, int32_t resolution,
                                               nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_TRY(check_res(resolution));
    NZ_REQUIRE(src && args, "src/args is NULL");
    (void)tmp;  // element-wise: done in place, no flush copy
    nz_ctx_handle_rides(ctx, out != nullptr);  // one launch, nothing behind it
    nz_ctx_arm_last_launch(ctx);
    NZ_TRY(nz_launch_normalize_args(ctx->stream, src, (size_t)resolution * resolution, args));
    return nz_ctx_finish(ctx, out);
}

// The same on any contiguous run of cells (a stripe's owned rows)
extern "C" int32_t nz_normalize_cells_dev(nz_ctx *ctx, float *data, size_t n_floats, const float *args, nz_handle dep,
                                          nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_REQUIRE(data && args, "data/args is NULL");
    nz_ctx_handle_rides(ctx, out != nullptr);  // one launch, nothing behind it
    nz_ctx_arm_last_launch(ctx);
    NZ_TRY(nz_launch_normalize_args(ctx->stream, data, n_floats, args));
    return nz_ctx_finish(ctx, out);
}

extern "C" size_t nz_flowmap_stage_work_floats(int32_t resolution) {
    return resolution > 0 ? (size_t)11 * resolution * resolution : 0;  // the reference stage's 11 planes
}

static int32_t flowmap_stage_impl(nz_ctx *ctx, float *src, float *work, int32_t iterations, float normMin,
                                  float normMax, int32_t resolution, int32_t count, nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_TRY(check_res(resolution));
    NZ_REQUIRE(src && work, "src/work is NULL");
    NZ_REQUIRE(iterations >= 1, "iterations < 1");
    NZ_REQUIRE(count >= 1 && count <= 65535, "count %d out of range [1,65535]", count);
    // a batch keeps plane p of all its tiles together: plane p = work + p * count * res^2
    size_t n = (size_t)resolution * resolution * count;
    nz_geom g = count > 1 ? nz_geom_batch(resolution, count) : nz_geom_tile(resolution);
    float *A[5], *B[5];  // {water, fN, fS, fE, fW} x {READ, WRITE}, FlowMapStage.cs:52-62
    for (int i = 0; i < 5; i++) {
        A[i] = work + (size_t)i * n;
        B[i] = work + (size_t)(5 + i) * n;
    }
    // `iterations` split into launches of <= nz_flow_fused_max() iterations that keep the tile on chip.
    // The first launch implies water == 0.0001 (fillStage, FlowMapStage.cs:129) and flux == 0 (defined);
    // the last one ends in writeStage + normStage (FlowMapStage.cs:179-194), args = {normMin, normMax,
    // normMax - normMin} (:48-51).
    int cap = nz_flow_fused_max();
    int launches = (iterations + cap - 1) / cap;
    int base = iterations / launches, rem = iterations % launches;
    float **cur = A, **nxt = B;
    // The result overwrites the height plane, which the last launch still reads with a halo: the
    // first launch keeps a private copy of it (the stage's 11th plane) for the later ones.
    float *hcopy = work + (size_t)10 * n;
    for (int i = 0; i < launches; i++) {
        int nit = base + (i < rem ? 1 : 0);
        int first = i == 0, last = i == launches - 1;
        const float *hsrc = first ? src : hcopy;
        float *dst = !last ? nullptr : (launches == 1 ? hcopy : src);
        if (last && launches > 1) {  // the stage's last operation (a single launch is followed by the copy back below)
            nz_ctx_handle_rides(ctx, out != nullptr);
            nz_ctx_arm_last_launch(ctx);
        }
        NZ_TRY(launch_on_ctx(ctx, g, [&](hipStream_t st, const nz_geom &gb) {
            return nz_launch_flow_fused(st, hsrc, first ? nullptr : cur, last ? nullptr : nxt, dst,
                                        (first && !last) ? hcopy : nullptr, gb, nit, first, last, normMin,
                                        normMax - normMin);
        }));
        float **s = cur; cur = nxt; nxt = s;
    }
    if (launches == 1) {
        nz_ctx_handle_rides(ctx, out != nullptr);
        nz_ctx_arm_last_launch(ctx);
        NZ_TRY(launch_on_ctx(ctx, g, [&](hipStream_t st, const nz_geom &gb) {
            size_t off = (size_t)gb.or0 * gb.pitch;
            return nz_launch_copy(st, src + off, hcopy + off, nz_geom_span(gb));
        }));
    }
    return nz_ctx_finish(ctx, out);
}

extern "C" int32_t nz_flowmap_stage(nz_ctx *ctx, float *src, float *work, int32_t iterations, float normMin,
                                    float normMax, int32_t resolution, nz_handle dep, nz_handle *out) {
    return flowmap_stage_impl(ctx, src, work, iterations, normMin, normMax, resolution, 1, dep, out);
}

extern "C" int32_t nz_flow_fused_max_iterations(void) { return nz_flow_fused_max(); }

// ---------------------------------------------------------------------------------------------
// batched stage bodies: `count` independent tiles of resolution^2 cells stored back to back, one launch
// sequence for all of them (new-framework feature: the reference runs one BasePipeline per tile request,
// Scripts/MeshTileGenerator.cs:181-211; small tiles cannot fill 256 CUs one at a time)
// ---------------------------------------------------------------------------------------------
static int32_t check_batch(int32_t resolution, int32_t count) {
    NZ_TRY(check_res(resolution));
    NZ_REQUIRE(count >= 1 && count <= 65535, "count %d out of range [1,65535]", count);
    return NZ_OK;
}

extern "C" int32_t nz_fractal_batch(nz_ctx *ctx, int32_t noiseType, float *data, int32_t resolution, int32_t count,
                                    const int32_t *positions, float hurst, float startingAmplitude, float stepdown,
                                    float detuneRate, int32_t octaves, int32_t noiseSize, nz_handle dep,
                                    nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_TRY(check_batch(resolution, count));
    NZ_REQUIRE(positions, "positions is NULL");
    nz_ctx_handle_rides(ctx, out != nullptr);
    nz_ctx_arm_last_launch(ctx);
    NZ_TRY(fractal_impl(ctx, ctx->stream, noiseType, data, resolution, resolution, resolution, hurst, startingAmplitude,
                        stepdown, detuneRate, octaves, 0, 0, noiseSize, count, (size_t)resolution * resolution, positions));
    return nz_ctx_finish(ctx, out);
}

extern "C" int32_t nz_kernel_filter_stage_batch(nz_ctx *ctx, float *src, float *tmp, int32_t filter, int32_t iterations,
                                                int32_t resolution, int32_t count, nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_TRY(check_batch(resolution, count));
    nz_kernel_taps t;
    NZ_TRY(filter_taps(filter, &t));
    nz_ctx_handle_rides(ctx, out != nullptr);  // (conv_iterations arms its last launch unless a copy back follows it)
    NZ_TRY(conv_iterations(ctx, src, tmp, nz_geom_batch(resolution, count), t, iterations));
    return nz_ctx_finish(ctx, out);
}

extern "C" int32_t nz_gauss_blur_stage_batch(nz_ctx *ctx, float *src, float *tmp, int32_t width, int32_t sigma,
                                             int32_t iterations, int32_t resolution, int32_t count, nz_handle dep,
                                             nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_TRY(check_batch(resolution, count));
    nz_kernel_taps t;
    NZ_TRY(gauss_taps(width, sigma, &t));
    nz_ctx_handle_rides(ctx, out != nullptr);  // (conv_iterations arms its last launch unless a copy back follows it)
    NZ_TRY(conv_iterations(ctx, src, tmp, nz_geom_batch(resolution, count), t, iterations));
    return nz_ctx_finish(ctx, out);
}

extern "C" int32_t nz_smooth_blur_stage_batch(nz_ctx *ctx, float *src, float *tmp, int32_t width, int32_t iterations,
                                              int32_t resolution, int32_t count, nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_TRY(check_batch(resolution, count));
    nz_kernel_taps t;
    NZ_TRY(smooth_taps(width, &t));
    nz_ctx_handle_rides(ctx, out != nullptr);  // (conv_iterations arms its last launch unless a copy back follows it)
    NZ_TRY(conv_iterations(ctx, src, tmp, nz_geom_batch(resolution, count), t, iterations));
    return nz_ctx_finish(ctx, out);
}

extern "C" int32_t nz_erosion_stage_batch(nz_ctx *ctx, float *src, float *tmp, int32_t iterations, int32_t resolution,
                                          int32_t count, nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_TRY(check_batch(resolution, count));
    nz_ctx_handle_rides(ctx, out != nullptr);  // (as conv_iterations)
    NZ_TRY(erosion_iterations(ctx, src, tmp, nz_geom_batch(resolution, count), iterations));
    return nz_ctx_finish(ctx, out);
}

extern "C" int32_t nz_flowmap_stage_batch(nz_ctx *ctx, float *src, float *work, int32_t iterations, float normMin,
                                          float normMax, int32_t resolution, int32_t count, nz_handle dep,
                                          nz_handle *out) {
    return flowmap_stage_impl(ctx, src, work, iterations, normMin, normMax, resolution, count, dep, out);
}

// ---------------------------------------------------------------------------------------------
// READ / WRITE pair forms (nz_rw_tile): TileHelpers.SWAP_RWTILE (Pipeline/Tiles/TileData.cs:42-45) as a swap of the
// two pointers instead of a copy job
// ---------------------------------------------------------------------------------------------
static int32_t check_rw(const nz_rw_tile *t) {
    NZ_REQUIRE(t, "tile is NULL");
    NZ_TRY(check_batch(t->resolution, t->count));
    NZ_REQUIRE(t->read && t->write && t->read != t->write, "read/write must be two distinct planes");
    return NZ_OK;
}
static nz_geom rw_geom(const nz_rw_tile *t) {
    return t->count > 1 ? nz_geom_batch(t->resolution, t->count) : nz_geom_tile(t->resolution);
}
static void rw_swap(nz_rw_tile *t, bool swapped) {
    if (swapped) {
        float *r = t->read;
        t->read = t->write;
        t->write = r;
    }
}

static int32_t conv_rw(nz_ctx *ctx, nz_rw_tile *tile, const nz_kernel_taps &t, int32_t iterations, nz_handle *out) {
    bool swapped = false;
    nz_ctx_handle_rides(ctx, out != nullptr);  // conv_iterations' last launch is this entry's last operation
    NZ_TRY(conv_iterations(ctx, tile->read, tile->write, rw_geom(tile), t, iterations, &swapped));
    rw_swap(tile, swapped);
    return nz_ctx_finish(ctx, out);
}

extern "C" int32_t nz_kernel_filter_stage_rw(nz_ctx *ctx, nz_rw_tile *tile, int32_t filter, int32_t iterations,
                                             nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_TRY(check_rw(tile));
    NZ_REQUIRE(filter != NZ_SOBEL3_2D, "Sobel3_2D keeps a third plane: use nz_kernel_filter_stage");
    nz_kernel_taps t;
    NZ_TRY(filter_taps(filter, &t));
    return conv_rw(ctx, tile, t, iterations, out);
}

extern "C" int32_t nz_gauss_blur_stage_rw(nz_ctx *ctx, nz_rw_tile *tile, int32_t width, int32_t sigma,
                                          int32_t iterations, nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_TRY(check_rw(tile));
    nz_kernel_taps t;
    NZ_TRY(gauss_taps(width, sigma, &t));
    return conv_rw(ctx, tile, t, iterations, out);
}

extern "C" int32_t nz_smooth_blur_stage_rw(nz_ctx *ctx, nz_rw_tile *tile, int32_t width, int32_t iterations,
                                           nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_TRY(check_rw(tile));
    nz_kernel_taps t;
    NZ_TRY(smooth_taps(width, &t));
    return conv_rw(ctx, tile, t, iterations, out);
}

extern "C" int32_t nz_erosion_stage_rw(nz_ctx *ctx, nz_rw_tile *tile, int32_t iterations, nz_handle dep,
                                       nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_TRY(check_rw(tile));
    bool swapped = false;
    nz_ctx_handle_rides(ctx, out != nullptr);  // erosion_iterations' last launch is this entry's last operation
    NZ_TRY(erosion_iterations(ctx, tile->read, tile->write, rw_geom(tile), iterations, &swapped));
    rw_swap(tile, swapped);
    return nz_ctx_finish(ctx, out);
}

extern "C" size_t nz_flowmap_stage_rw_work_floats(int32_t resolution, int32_t count) {
    return resolution > 0 && count > 0 ? (size_t)10 * resolution * resolution * count : 0;
}

extern "C" int32_t nz_flowmap_stage_rw(nz_ctx *ctx, nz_rw_tile *tile, float *work, int32_t iterations, float normMin,
                                       float normMax, nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_TRY(check_rw(tile));
    NZ_REQUIRE(work, "work is NULL");
    NZ_REQUIRE(iterations >= 1, "iterations < 1");
    size_t n = (size_t)tile->resolution * tile->resolution * tile->count;
    nz_geom g = rw_geom(tile);
    float *A[5], *B[5];  // {water, fN, fS, fE, fW} x {READ, WRITE}, FlowMapStage.cs:52-62
    for (int i = 0; i < 5; i++) {
        A[i] = work + (size_t)i * n;
        B[i] = work + (size_t)(5 + i) * n;
    }
    int cap = nz_flow_fused_max();
    int launches = (iterations + cap - 1) / cap;
    int base = iterations / launches, rem = iterations % launches;
    float **cur = A, **nxt = B;
    // every launch reads the heights from the READ plane, which nothing overwrites; the last one writes the WRITE plane
    for (int i = 0; i < launches; i++) {
        int nit = base + (i < rem ? 1 : 0);
        int first = i == 0, last = i == launches - 1;
        if (last) {  // the stage's last operation: its handle rides on this launch
            nz_ctx_handle_rides(ctx, out != nullptr);
            nz_ctx_arm_last_launch(ctx);
        }
        NZ_TRY(launch_on_ctx(ctx, g, [&](hipStream_t st, const nz_geom &gb) {
            return nz_launch_flow_fused(st, tile->read, first ? nullptr : cur, last ? nullptr : nxt,
                                        last ? tile->write : nullptr, nullptr, gb, nit, first, last, normMin,
                                        normMax - normMin);
        }));
        float **s = cur; cur = nxt; nxt = s;
    }
    rw_swap(tile, true);
    return nz_ctx_finish(ctx, out);
}

extern "C" int32_t nz_flow_fused_stripe(nz_ctx *ctx, const float *height, const float *const *state_in,
                                        float *const *state_out, float *dst, const nz_stripe *st, int32_t iterations,
                                        int32_t first, int32_t last, float normMin, float normMax, nz_handle dep,
                                        nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_REQUIRE(iterations >= 1 && iterations <= nz_flow_fused_max(), "iterations %d cannot be fused", iterations);
    NZ_TRY(nz_check_stripe(st, 2 * iterations));
    NZ_REQUIRE(height, "height is NULL");
    NZ_REQUIRE(first || state_in, "state_in is NULL");
    NZ_REQUIRE(last ? dst != nullptr : state_out != nullptr, "output plane is NULL");
    for (int i = 0; i < 5; i++) {
        NZ_REQUIRE(first || state_in[i], "state_in[%d] is NULL", i);
        NZ_REQUIRE(last || state_out[i], "state_out[%d] is NULL", i);
    }
    NZ_REQUIRE(!last || dst != height, "dst must not alias height");
    NZ_TRY(nz_launch_flow_fused(ctx->stream, height, first ? nullptr : state_in, last ? nullptr : state_out,
                                last ? dst : nullptr, nullptr, nz_geom_from_stripe(*st), iterations, first, last, normMin,
                                normMax - normMin));
    return nz_ctx_finish(ctx, out);
}

// ---------------------------------------------------------------------------------------------
// mesh
// ---------------------------------------------------------------------------------------------
extern "C" size_t nz_mesh_vertex_count(int32_t resolution) {  // VertexCount, Overshoot :26
    return resolution > 0 ? (size_t)(resolution + 1) * (resolution + 1) : 0;
}

extern "C" size_t nz_mesh_index_count(int32_t resolution) {  // IndexCount, Overshoot :28
    return resolution > 0 ? (size_t)6 * resolution * resolution : 0;
}

static int32_t heightmap_mesh_impl(nz_ctx *ctx, int32_t meshType, void *vertices, uint32_t *indices,
                                   int32_t resolution, int32_t inputResolution, int32_t marginPix, float tileHeight,
                                   float tileSize, const float *heights, int32_t count, nz_handle dep, nz_handle *out,
                                   int index16 = 0) {
    NZ_BEGIN(ctx, dep);
    (void)marginPix;  // MarginScale is commented out of the vertex path (Overshoot :64)
    NZ_REQUIRE(vertices && indices && heights, "buffer is NULL");
    NZ_REQUIRE(resolution >= 1 && resolution <= 26754, "resolution %d out of range", resolution);
    NZ_REQUIRE(inputResolution >= resolution && inputResolution <= 46340, "inputResolution %d out of range",
               inputResolution);
    int off = (inputResolution - resolution) / 2;  // PixOffset, Overshoot :33
    // shapes for which the reference would index outside the height plane are rejected
    if (meshType == NZ_MESH_OVERSHOOT_SQUARE_GRID) {
        int hi = resolution + 1 < resolution + off ? resolution + 1 : resolution + off;
        NZ_REQUIRE(hi + off <= inputResolution - 1, "overshoot mesh: margin too small for resolution %d / input %d",
                   resolution, inputResolution);
    } else if (meshType == NZ_MESH_SQUARE_GRID) {
        NZ_REQUIRE(resolution + off <= inputResolution - 1, "square mesh: inputResolution must exceed resolution");
    } else {
        nz_set_error("unknown MeshType %d", meshType);
        return NZ_ERR_INVALID;
    }
    nz_ctx_handle_rides(ctx, out != nullptr);  // vertex launch, then the index launch: the handle rides on that one
    nz_ctx_arm_last_launch(ctx);
    NZ_TRY(nz_launch_mesh(ctx->stream, meshType, vertices, indices, resolution, inputResolution, tileHeight, tileSize,
                          heights, count, index16));
    return nz_ctx_finish(ctx, out);
}

// HeightMapMeshJob<G, PositionStream16>: the same vertex stream, TriangleUInt16 indices (the reference's own caveat:
// only valid while the vertex count fits 16 bits, Mesh/Streams/PositionStream.cs:12)
extern "C" int32_t nz_heightmap_mesh16(nz_ctx *ctx, int32_t meshType, void *vertices, uint16_t *indices,
                                       int32_t resolution, int32_t inputResolution, int32_t marginPix,
                                       float tileHeight, float tileSize, const float *heights, nz_handle dep,
                                       nz_handle *out) {
    return heightmap_mesh_impl(ctx, meshType, vertices, reinterpret_cast<uint32_t *>(indices), resolution, inputResolution,
                               marginPix, tileHeight, tileSize, heights, 1, dep, out, 1);
}

extern "C" int32_t nz_heightmap_mesh(nz_ctx *ctx, int32_t meshType, void *vertices, uint32_t *indices,
                                     int32_t resolution, int32_t inputResolution, int32_t marginPix,
                                     float tileHeight, float tileSize, const float *heights, nz_handle dep,
                                     nz_handle *out) {
    return heightmap_mesh_impl(ctx, meshType, vertices, indices, resolution, inputResolution, marginPix, tileHeight,
                               tileSize, heights, 1, dep, out);
}

// MeshJobScheduleDelegate(Mesh, MeshData, resolution, dep, TileSize, Height), Mesh/Job/MeshJob.cs:37-60, with
// G = SharedSquareGridPosition: TileSize and Height only set mesh.bounds, the vertices span the unit square
extern "C" int32_t nz_square_grid_mesh(nz_ctx *ctx, void *vertices, uint32_t *indices, int32_t resolution, nz_handle dep,
                                       nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_REQUIRE(vertices && indices, "buffer is NULL");
    NZ_REQUIRE(resolution >= 1 && resolution <= 26754, "resolution %d out of range", resolution);
    nz_ctx_handle_rides(ctx, out != nullptr);  // vertex launch, then the index launch: the handle rides on that one
    nz_ctx_arm_last_launch(ctx);
    NZ_TRY(nz_launch_mesh_planar(ctx->stream, vertices, indices, resolution));
    return nz_ctx_finish(ctx, out);
}

// `count` meshes from `count` height planes stored back to back; mesh k at vertices + k * vertex_count * 48 bytes
// and indices + k * index_count
extern "C" int32_t nz_heightmap_mesh_batch(nz_ctx *ctx, int32_t meshType, void *vertices, uint32_t *indices,
                                           int32_t resolution, int32_t inputResolution, int32_t marginPix,
                                           float tileHeight, float tileSize, const float *heights, int32_t count,
                                           nz_handle dep, nz_handle *out) {
    NZ_REQUIRE(count >= 1 && count <= 65535, "count %d out of range [1,65535]", count);
    return heightmap_mesh_impl(ctx, meshType, vertices, indices, resolution, inputResolution, marginPix, tileHeight,
                               tileSize, heights, count, dep, out);
}

// ---------------------------------------------------------------------------------------------
// element-wise stages
// ---------------------------------------------------------------------------------------------
extern "C" int32_t nz_constant_job(nz_ctx *ctx, int32_t operation, float *srcL, float *tmp, float constantValue,
                                   int32_t resolution, nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_TRY(check_res(resolution));
    NZ_REQUIRE(srcL, "srcL is NULL");
    (void)tmp;  // element-wise: updated in place, no flush copy
    nz_ctx_handle_rides(ctx, out != nullptr);  // one launch, nothing behind it
    nz_ctx_arm_last_launch(ctx);
    NZ_TRY(nz_launch_constant(ctx->stream, operation, srcL, (size_t)resolution * resolution, constantValue));
    return nz_ctx_finish(ctx, out);
}

extern "C" int32_t nz_reduction_job(nz_ctx *ctx, int32_t operation, float *srcL, const float *srcR, float *tmp,
                                    int32_t resolution, nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_TRY(check_res(resolution));
    NZ_REQUIRE(srcL && srcR, "srcL/srcR is NULL");
    (void)tmp;
    nz_ctx_handle_rides(ctx, out != nullptr);  // one launch, nothing behind it
    nz_ctx_arm_last_launch(ctx);
    NZ_TRY(nz_launch_reduce(ctx->stream, operation, srcL, srcR, (size_t)resolution * resolution));
    return nz_ctx_finish(ctx, out);
}

extern "C" int32_t nz_curve_job(nz_ctx *ctx, float *src, float *tmp, const float *curve, int32_t curveSize,
                                int32_t resolution, nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_TRY(check_res(resolution));
    NZ_REQUIRE(src && curve, "src/curve is NULL");
    NZ_REQUIRE(curveSize >= 2 && curveSize <= 16384, "curve length %d out of range [2,16384]", curveSize);
    (void)tmp;
    nz_ctx_handle_rides(ctx, out != nullptr);  // one launch, nothing behind it
    nz_ctx_arm_last_launch(ctx);
    NZ_TRY(nz_launch_curve(ctx->stream, src, (size_t)resolution * resolution, curve, curveSize));
    return nz_ctx_finish(ctx, out);
}

// ---- live erosion: the deterministic grid jobs ----------------------------------------
// UpdateFlowFromTrackJob.Schedule, Geologic/ParticleErosion/MultiThreadErosionJob.cs:240-261
extern "C" int32_t nz_update_flow_from_track(nz_ctx *ctx, float *pool, float *flow, float *track, float flowLossRate,
                                             float surfaceEvaporationRate, float tileHeight, int32_t resolution,
                                             nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_TRY(check_res(resolution));
    NZ_REQUIRE(pool && flow && track, "pool/flow/track is NULL");
    NZ_TRY(nz_launch_flow_from_track(ctx->stream, pool, flow, track, (size_t)resolution * resolution, flowLossRate,
                                     surfaceEvaporationRate / tileHeight));
    return nz_ctx_finish(ctx, out);
}

// PoolAutomataJob (MultiThreadErosionJob.cs:264-327): `iterations` x four colour passes of WorldTile.SpreadPool.
//   NZ_POOL_RUNS=0   one lane per row, as the reference walks it (no mask);
//   otherwise        parallel runs of acting steps (nz_elementwise.hip): a masks launch, then
//       * the sparse form -- ONE launch of one workgroup that runs the whole job from the list of non-empty mask words --
//         when the last job that reported (at most 8 jobs ago) found few of them: 2 launches instead of ~50;
//       * the dense form otherwise: the sparse launch first (it still takes a job with few words and then turns the dense
//         launches into no-ops; and it reports), then 4 launches per iteration and a clean between iterations.
//   NZ_POOL_SPARSE=0: never the sparse launch; =2: always, and nothing follows it (the test matrix).
static int32_t pool_job(nz_ctx *ctx, float *pool, const float *height, int res, int iterations, int32_t *hdr, nz_particle *data) {
    static const int runs = [] { const char *e = getenv("NZ_POOL_RUNS"); return e ? atoi(e) : 1; }();
    static const int sparse = [] { const char *e = getenv("NZ_POOL_SPARSE"); return e ? atoi(e) : 1; }();
    if (!runs) {
        for (int i = 0; i < iterations; i++)
            for (int xoff = 0; xoff < 2; xoff++)
                for (int zoff = 0; zoff < 2; zoff++)
                    NZ_TRY(nz_launch_pool_automata_pass(ctx->stream, pool, height, res, xoff, zoff, hdr, data, nullptr, nullptr));
        return NZ_OK;
    }
    if (iterations == 0) return NZ_OK;
    float *p = nullptr;
    NZ_TRY(nz_ctx_scratch(ctx, nz_pool_automata_mask_words(res), &p));
    unsigned *mask = reinterpret_cast<unsigned *>(p);
    int *ctl = nullptr;
    bool dense = true;
    if (sparse) {
        NZ_TRY(nz_ctx_pool_state(ctx));
        ctl = ctx->pool_ctl;
        constexpr int LIMIT = 1024;  // non-empty words the one workgroup takes on: one per thread and pass
        const unsigned long long seq = ++ctx->pool_seq;
        const unsigned long long hint = *reinterpret_cast<volatile unsigned long long *>(ctx->pool_hint);
        const unsigned long long hint_seq = hint >> 32;
        const bool fresh = hint_seq != 0 && seq - hint_seq <= 8 && seq > hint_seq;
        dense = sparse == 2 ? false : !(fresh && (unsigned)hint <= LIMIT / 2);
        NZ_TRY(nz_launch_pool_automata_masks(ctx->stream, pool, res, mask, ctl, 1));
        NZ_TRY(nz_launch_pool_automata_sparse(ctx->stream, pool, height, res, iterations, LIMIT, dense ? 1 : 0, hdr, data, mask, ctl,
                                              ctx->pool_hint_dev, seq & 0xffffffffull));
    } else {
        NZ_TRY(nz_launch_pool_automata_masks(ctx->stream, pool, res, mask, nullptr, 0));
    }
    if (!dense) return NZ_OK;
    for (int i = 0; i < iterations; i++) {
        if (i > 0) NZ_TRY(nz_launch_pool_automata_clean(ctx->stream, pool, res, mask, ctl));
        for (int xoff = 0; xoff < 2; xoff++)
            for (int zoff = 0; zoff < 2; zoff++)
                NZ_TRY(nz_launch_pool_automata_pass(ctx->stream, pool, height, res, xoff, zoff, hdr, data, mask, ctl));
    }
    return NZ_OK;
}

// PoolAutomataJob.Schedule, MultiThreadErosionJob.cs:289-325, with drainParticles == false (the other setting feeds
// the particle queue, which is outside the deterministic part)
extern "C" int32_t nz_pool_automata(nz_ctx *ctx, float *pool, const float *height, int32_t iterations, int32_t resolution,
                                    nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_TRY(check_res(resolution));
    NZ_REQUIRE(pool && height && pool != height, "pool/height must be two distinct planes");
    NZ_REQUIRE(resolution >= 2 && iterations >= 0, "resolution < 2 or iterations < 0");
    NZ_TRY(pool_job(ctx, pool, height, resolution, iterations, nullptr, nullptr));
    return nz_ctx_finish(ctx, out);
}

// PoolAutomataJob.Schedule with its whole argument list (:289-325); drainParticles feeds the particle queue
extern "C" int32_t nz_pool_automata_job(nz_ctx *ctx, float *pool, const float *height, nz_particle_queue *particleQueue,
                                        const nz_erosion_params *ep, const nz_tile_set_meta *tm, int32_t iterations,
                                        int32_t res, int32_t drainParticles, nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    (void)ep;
    (void)tm;
    NZ_TRY(check_res(res));
    NZ_REQUIRE(pool && height && pool != height, "pool/height must be two distinct planes");
    NZ_REQUIRE(res >= 2 && iterations >= 0, "resolution < 2 or iterations < 0");
    NZ_REQUIRE(!drainParticles || particleQueue, "drainParticles needs a particle queue");
    int32_t *hdr = drainParticles ? nz_particle_queue_hdr(particleQueue) : nullptr;
    nz_particle *data = drainParticles ? nz_particle_queue_data(particleQueue) : nullptr;
    NZ_TRY(pool_job(ctx, pool, height, res, iterations, hdr, data));
    return nz_ctx_finish(ctx, out);
}

// CropJobDelegate, Filter/Sample/CropJob.cs:62-68
extern "C" int32_t nz_crop_job(nz_ctx *ctx, const float *input, int32_t inputResolution, float *output,
                               int32_t outputResolution, nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_TRY(check_res(inputResolution));
    NZ_TRY(check_res(outputResolution));
    NZ_REQUIRE(input && output && input != output, "input/output must be two distinct planes");
    nz_ctx_handle_rides(ctx, out != nullptr);  // one launch, nothing behind it
    nz_ctx_arm_last_launch(ctx);
    NZ_TRY(nz_launch_crop(ctx->stream, input, inputResolution, output, outputResolution));
    return nz_ctx_finish(ctx, out);
}

// ThermalErosionFilterDelegate, Filter/Kernel/Blur/ThermalErosionFilter.cs:149-157 (Schedule :117-144)
extern "C" int32_t nz_thermal_erosion(nz_ctx *ctx, float *src, float talus, float incrementRatio,
                                      float meshHeightWidthRatio, int32_t iterations, int32_t resolution, nz_handle dep,
                                      nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_TRY(check_res(resolution));
    NZ_REQUIRE(src, "src is NULL");
    NZ_REQUIRE(iterations >= 0, "iterations < 0");
    float t = (talus / 90.0f) * 3.14159f / 2.0f;                             // :131
    float maxDiff = (tanf(t) * meshHeightWidthRatio) / (float)resolution;   // :132
    nz_ctx_handle_rides(ctx, out != nullptr);  // the handle rides on the last phase's launch
    for (int i = 0; i < iterations; i++) {
        const bool last = i == iterations - 1;
        if (nz_thermal_pair_fits(resolution)) {  // two phases per pass over the plane (a row beyond the LDS strip: one launch per phase)
            NZ_TRY(nz_launch_thermal_pair(ctx->stream, src, resolution, 0, maxDiff, incrementRatio));
            if (last) nz_ctx_arm_last_launch(ctx);
            NZ_TRY(nz_launch_thermal_pair(ctx->stream, src, resolution, 1, maxDiff, incrementRatio));
        } else {
            for (int flip = 0; flip < 4; flip++) {
                if (last && flip == 3) nz_ctx_arm_last_launch(ctx);
                NZ_TRY(nz_launch_thermal_phase(ctx->stream, src, resolution, flip, maxDiff, incrementRatio));
            }
        }
    }
    return nz_ctx_finish(ctx, out);
}
