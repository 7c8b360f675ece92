// nz_filter.hip -- separable convolutions and the min "value erosion" filter (gfx950).
//
// Replaces GenericKernelJob<KernelTileMutation<KernelSample{X,Z}Operator>, RWTileData>
// (Filter/Kernel/KernelJob.cs:18-54,165-185; operators Filter/Kernel/KernelOperators.cs:18-67) and
// the KernelMin{X,Z}Operator pair of ErosionKernelJob (KernelJob.cs:317-347, KernelOperators.cs:69-118).
//
// conv_fused_kernel: one workgroup stages a 64 x 128 tile (rows x cols, halo included) in LDS with
// coalesced 16-byte loads, runs T applications of (X pass, Z pass) on it without touching HBM and
// stores the (64-2H) x (128-2HX) interior, H = T*(K-1)/2.  HBM traffic per launch is one read and
// one write of the plane for T filter applications (the reference moves 32 B/cell per application:
// two passes + two serial flush copies).  The FlushWriteSlice copies (Pipeline/Tiles/TileData.cs:15-40)
// do not exist here: passes ping-pong between `src` and `tmp`.
//
// Arithmetic order is the reference's: X pass sums taps k ascending, Z pass sums k descending
// (KernelOperators.cs:34-40,59-65), product then add (no FMA contraction), then * factor.
// Clamp-to-edge (TileData.cs:72-77) is applied per pass: cells outside the global grid are
// re-replicated from the border after every Z pass.
//
// LDS layout: pitch 132 floats (= 4 * 33): in the X pass the 64 lanes of a wave read the same
// column run of 64 different rows with ds_read_b128, and 33 being odd spreads every 16-lane group
// over all 16 four-bank slots; in the Z pass lanes read consecutive float4 columns of one row.
#include <cstdlib>

#include "nz_internal.hpp"

namespace {

constexpr int CT = 256;      // threads per workgroup
constexpr int TW = 128;      // LDS tile width, cells
constexpr int TH = 64;       // LDS tile height, rows
constexpr int LP = TW + 4;   // LDS pitch, floats

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ void tile_origin(const nz_geom &g, int OW, int OH, int &ox0, int &oz0) {
    int tiles_x = (g.cols + OW - 1) / OW;
    int by = blockIdx.x / tiles_x;
    int bx = blockIdx.x - by * tiles_x;
    ox0 = bx * OW;
    oz0 = g.or0 + by * OH;
}

// stage the TH x TW tile whose (0,0) is global (lz0, lx0) into LDS, clamping reads to the grid
__device__ __forceinline__ void load_tile(float *A, const float *__restrict__ src, const nz_geom &g, int lx0,
                                          int lz0, bool inside) {
    const int tid = threadIdx.x;
    bool vec_ok = inside && ((g.pitch & 3) == 0) && ((lx0 & 3) == 0) &&
                  ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
    if (vec_ok) {
#pragma unroll
        for (int i = 0; i < TH * TW / 4 / CT; i++) {
            int idx = tid + i * CT;
            int r = idx / (TW / 4), c4 = idx % (TW / 4);
            float4 v = *reinterpret_cast<const float4 *>(src + (size_t)(lz0 + r) * g.pitch + lx0 + c4 * 4);
            *reinterpret_cast<float4 *>(&A[r * LP + c4 * 4]) = v;
        }
    } else {
        for (int i = 0; i < TH * TW / CT; i++) {
            int idx = tid + i * CT;
            int r = idx / TW, c = idx % TW;
            int gx = clampi(lx0 + c, 0, g.cols - 1);
            int gz = clampi(lz0 + r, g.zc0, g.zc1);
            A[r * LP + c] = src[(size_t)gz * g.pitch + gx];
        }
    }
}

// write the [r0, r0+OH) x [c0, c0+OW) interior of the LDS tile to dst
__device__ __forceinline__ void store_tile(const float *A, float *__restrict__ dst, const nz_geom &g, int lx0,
                                           int lz0, int c0, int OW, int r0, int OH) {
    const int tid = threadIdx.x;
    int ox0 = lx0 + c0, oz0 = lz0 + r0;
    bool vec_ok = ((g.pitch & 3) == 0) && ((ox0 & 3) == 0) && ((OW & 3) == 0) && (ox0 + OW <= g.cols) &&
                  ((reinterpret_cast<uintptr_t>(dst) & 15) == 0);
    if (vec_ok) {
        int w4 = OW / 4;
        for (int idx = tid; idx < OH * w4; idx += CT) {
            int r = idx / w4, c4 = idx - r * w4;
            int gz = oz0 + r;
            if (gz < g.or1) {
                float4 v = *reinterpret_cast<const float4 *>(&A[(r0 + r) * LP + c0 + c4 * 4]);
                *reinterpret_cast<float4 *>(dst + (size_t)gz * g.pitch + ox0 + c4 * 4) = v;
            }
        }
    } else {
        for (int idx = tid; idx < OH * OW; idx += CT) {
            int r = idx / OW, c = idx - r * OW;
            int gz = oz0 + r, gx = ox0 + c;
            if (gz < g.or1 && gx < g.cols) dst[(size_t)gz * g.pitch + gx] = A[(r0 + r) * LP + c0 + c];
        }
    }
}

// cells of the tile that lie outside the grid take the value of the nearest grid cell (per-pass
// clamp-to-edge of RWTileData.GetData)
__device__ __forceinline__ void replicate_border(float *A, const nz_geom &g, int lx0, int lz0) {
    for (int idx = threadIdx.x; idx < TH * TW; idx += CT) {
        int r = idx / TW, c = idx % TW;
        int gx = lx0 + c, gz = lz0 + r;
        int cx = clampi(gx, 0, g.cols - 1), cz = clampi(gz, g.zc0, g.zc1);
        if (cx != gx || cz != gz) {
            int sr = clampi(cz - lz0, 0, TH - 1), sc = clampi(cx - lx0, 0, TW - 1);
            A[r * LP + c] = A[sr * LP + sc];
        }
    }
}

template <int KS>
__global__ __launch_bounds__(CT) void conv_fused_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                       nz_geom g, nz_kernel_taps taps, int T) {
    constexpr int O = (KS - 1) / 2;
    constexpr int OX4 = (O + 3) & ~3;  // aligned read margin of the X pass
    __shared__ __attribute__((aligned(16))) float A[TH * LP];

    const int H = T * O, HX = (H + 3) & ~3;
    const int OW = TW - 2 * HX, OH = TH - 2 * H;
    int ox0, oz0;
    tile_origin(g, OW, OH, ox0, oz0);
    const int lx0 = ox0 - HX, lz0 = oz0 - H;
    const bool inside = lx0 >= 0 && lx0 + TW <= g.cols && lz0 >= g.zc0 && lz0 + TH - 1 <= g.zc1;
    const int tid = threadIdx.x;

    load_tile(A, src, g, lx0, lz0, inside);
    __syncthreads();

    for (int t = 0; t < T; t++) {
        // ---- X pass: 4 row-runs of 8 cells per thread; lanes of a wave take 64 different rows
        float acc[4][8];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int item = tid + j * CT;
            int r = item & (TH - 1), run = item >> 6;
            float in[8 + 2 * OX4];
#pragma unroll
            for (int q = 0; q < (8 + 2 * OX4) / 4; q++) {
                int cc = clampi(run * 2 - OX4 / 4 + q, 0, TW / 4 - 1);
                float4 v = *reinterpret_cast<const float4 *>(&A[r * LP + cc * 4]);
                in[4 * q + 0] = v.x; in[4 * q + 1] = v.y; in[4 * q + 2] = v.z; in[4 * q + 3] = v.w;
            }
#pragma unroll
            for (int e = 0; e < 8; e++) {
                float total = 0.0f;
#pragma unroll
                for (int kk = 0; kk < KS; kk++) total += in[OX4 - O + e + kk] * taps.kx[kk];
                acc[j][e] = total * taps.factor;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int item = tid + j * CT;
            int r = item & (TH - 1), run = item >> 6;
            float *p = &A[r * LP + run * 8];
            *reinterpret_cast<float4 *>(p) = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
            *reinterpret_cast<float4 *>(p + 4) = make_float4(acc[j][4], acc[j][5], acc[j][6], acc[j][7]);
        }
        __syncthreads();

        // ---- Z pass: one 4-column x 8-row block per thread; lanes take consecutive float4 columns
        {
            int cg = tid & 31, rr = tid >> 5;
            float4 in[8 + 2 * O];
#pragma unroll
            for (int q = 0; q < 8 + 2 * O; q++) {
                int r = clampi(rr * 8 - O + q, 0, TH - 1);
                in[q] = *reinterpret_cast<const float4 *>(&A[r * LP + cg * 4]);
            }
            float4 outv[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                float4 total = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
                for (int kk = 0; kk < KS; kk++) {  // k = O - kk descending, Kernel[k_off - k] = kz[kk]
                    float4 v = in[e + 2 * O - kk];
                    float w = taps.kz[kk];
                    total.x += v.x * w; total.y += v.y * w; total.z += v.z * w; total.w += v.w * w;
                }
                outv[e] = make_float4(total.x * taps.factor, total.y * taps.factor, total.z * taps.factor,
                                      total.w * taps.factor);
            }
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 8; e++)
                *reinterpret_cast<float4 *>(&A[(rr * 8 + e) * LP + cg * 4]) = outv[e];
        }
        __syncthreads();
        if (!inside && t + 1 < T) {
            replicate_border(A, g, lx0, lz0);
            __syncthreads();
        }
    }
    store_tile(A, dst, g, lx0, lz0, HX, OW, H, OH);
}

// ---- register-resident fused kernel ------------------------------------------------------------------
// Same contract as conv_fused_kernel (T applications of X pass + Z pass on a 64 x 128 tile, halo
// included), but the tile never sits in LDS: each of the 256 threads keeps a 4-column x 8-row block in
// registers for the whole launch.  The X pass takes its (K-1)/2 west / east neighbours from the
// adjacent lanes with wave-shift DPP moves; the Z pass needs (K-1)/2 rows from the thread above and
// below, and only those boundary rows travel through LDS (16-byte accesses, double buffered: one
// barrier per application).  Global loads and stores go register <-> HBM directly, 16 B per lane.
// Clamp-to-edge is applied when a window is assembled: a tap beyond the grid takes the border cell's
// current value (RWTileData.GetData, Pipeline/Tiles/TileData.cs:72-82), so no fix-up pass is needed.
constexpr int RB = 8;  // rows per thread

__device__ __forceinline__ float dpp_prev(float v) {  // lane i <- lane i-1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_next(float v) {  // lane i <- lane i+1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

template <int KS, bool UNIT, int NT>
__global__ __launch_bounds__(NT) void conv_reg_kernel(const float *__restrict__ src, float *__restrict__ dst, nz_geom g,
                                                     nz_kernel_taps taps, int T, int aligned) {
    constexpr int O = (KS - 1) / 2;
    constexpr int WN = 4 + 2 * O;   // X window
    constexpr int ZN = RB + 2 * O;  // Z window
    constexpr int TH = NT / 32 * RB;  // tile rows: one 8-row block per 32 threads (shadows the file-level TH)
    // boundary rows of every 8-row block: [parity][block][top|bottom][o][column group]
    __shared__ float4 s_edge[2][TH / RB][2][O][TW / 4];

    const int tid = threadIdx.x, cg = tid & 31, rb = tid >> 5;
    const int H = T * O, HX = (H + 3) & ~3;
    const int OW = TW - 2 * HX, OH = TH - 2 * H;
    int ox0, oz0;
    tile_origin(g, OW, OH, ox0, oz0);
    const int lx0 = ox0 - HX, lz0 = oz0 - H;
    const int gx0 = lx0 + cg * 4, gzb = lz0 + rb * RB;
    const bool inside = lx0 >= 0 && lx0 + TW <= g.cols && lz0 >= g.zc0 && lz0 + TH - 1 <= g.zc1;
    const bool fast = inside && aligned;

    float v[RB][4];
#pragma unroll
    for (int r = 0; r < RB; r++) {
        if (fast) {
            float4 t = *reinterpret_cast<const float4 *>(src + (size_t)(gzb + r) * g.pitch + gx0);
            v[r][0] = t.x; v[r][1] = t.y; v[r][2] = t.z; v[r][3] = t.w;
        } else {
            size_t row = (size_t)clampi(gzb + r, g.zc0, g.zc1) * g.pitch;
#pragma unroll
            for (int e = 0; e < 4; e++) v[r][e] = src[row + clampi(gx0 + e, 0, g.cols - 1)];
        }
    }
    // window indices of the last grid column / first and last grid rows, for the clamps of edge tiles
    const int icx = g.cols - 1 - gx0 + O;
    const int iz0 = g.zc0 - gzb + O, iz1 = g.zc1 - gzb + O;

    for (int t = 0; t < T; t++) {
        // ---- X pass (KernelSampleXOperator: taps k ascending), in place row by row
#pragma unroll
        for (int r = 0; r < RB; r++) {
            float w[WN];
#pragma unroll
            for (int o = 0; o < O; o++) {
                w[o] = dpp_prev(v[r][4 - O + o]);
                w[4 + O + o] = dpp_next(v[r][o]);
            }
#pragma unroll
            for (int e = 0; e < 4; e++) w[O + e] = v[r][e];
            if (!inside) {
                if (gx0 == 0) {
#pragma unroll
                    for (int o = 0; o < O; o++) w[o] = w[O];
                }
                if (icx >= O && icx < WN - 1) {
                    float wc = w[O];
#pragma unroll
                    for (int i = O + 1; i < WN; i++) wc = (i == icx) ? w[i] : wc;
#pragma unroll
                    for (int i = O + 1; i < WN; i++) w[i] = (i > icx) ? wc : w[i];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float total = w[e] * taps.kx[0];  // 0 + a*b == a*b
#pragma unroll
                for (int kk = 1; kk < KS; kk++) total += w[e + kk] * taps.kx[kk];
                v[r][e] = UNIT ? total : total * taps.factor;
            }
        }
        // ---- exchange the block's boundary rows
        const int par = t & 1;
#pragma unroll
        for (int o = 0; o < O; o++) {
            s_edge[par][rb][0][o][cg] = make_float4(v[o][0], v[o][1], v[o][2], v[o][3]);
            s_edge[par][rb][1][o][cg] = make_float4(v[RB - O + o][0], v[RB - O + o][1], v[RB - O + o][2], v[RB - O + o][3]);
        }
        __syncthreads();
        float z[ZN][4];
#pragma unroll
        for (int o = 0; o < O; o++) {
            // rows gzb-O+o (bottom rows of the block above) and gzb+RB+o (top rows of the block below)
            float4 a = rb > 0 ? s_edge[par][rb - 1][1][o][cg] : make_float4(v[0][0], v[0][1], v[0][2], v[0][3]);
            float4 b = rb < TH / RB - 1 ? s_edge[par][rb + 1][0][o][cg]
                                        : make_float4(v[RB - 1][0], v[RB - 1][1], v[RB - 1][2], v[RB - 1][3]);
            z[o][0] = a.x; z[o][1] = a.y; z[o][2] = a.z; z[o][3] = a.w;
            z[RB + O + o][0] = b.x; z[RB + O + o][1] = b.y; z[RB + O + o][2] = b.z; z[RB + O + o][3] = b.w;
        }
#pragma unroll
        for (int r = 0; r < RB; r++)
#pragma unroll
            for (int e = 0; e < 4; e++) z[O + r][e] = v[r][e];
        if (!inside) {
            if (iz0 > 0 && iz0 < ZN) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float c = z[0][e];
#pragma unroll
                    for (int i = 1; i < ZN; i++) c = (i == iz0) ? z[i][e] : c;
#pragma unroll
                    for (int i = 0; i < ZN - 1; i++) z[i][e] = (i < iz0) ? c : z[i][e];
                }
            }
            if (iz1 >= 0 && iz1 < ZN - 1) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float c = z[0][e];
#pragma unroll
                    for (int i = 1; i < ZN; i++) c = (i == iz1) ? z[i][e] : c;
#pragma unroll
                    for (int i = 1; i < ZN; i++) z[i][e] = (i > iz1) ? c : z[i][e];
                }
            }
        }
        // ---- Z pass (KernelSampleZOperator: k descending, Kernel[k_off - k])
#pragma unroll
        for (int r = 0; r < RB; r++) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float total = z[r + 2 * O][e] * taps.kz[0];
#pragma unroll
                for (int kk = 1; kk < KS; kk++) total += z[r + 2 * O - kk][e] * taps.kz[kk];
                v[r][e] = UNIT ? total : total * taps.factor;
            }
        }
    }

    // ---- store the interior
#pragma unroll
    for (int r = 0; r < RB; r++) {
        int lr = rb * RB + r, gz = gzb + r;
        bool in = lr >= H && lr < H + OH && cg * 4 >= HX && cg * 4 < HX + OW && gz < g.or1 && gx0 < g.cols;
        if (in) {
            if (aligned && gx0 + 4 <= g.cols) {
                *reinterpret_cast<float4 *>(dst + (size_t)gz * g.pitch + gx0) = make_float4(v[r][0], v[r][1], v[r][2], v[r][3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (gx0 + e < g.cols) dst[(size_t)gz * g.pitch + gx0 + e] = v[r][e];
            }
        }
    }
}

// E applications of the {-1,0} min window fused as one window min over [x-E,x] x [z-E,z], register
// resident like conv_reg_kernel: the X pass reaches E <= 4 cells into the lane on the left (DPP), the Z
// pass E rows into the block above (LDS).  Cells outside the grid are loaded as +FLT_MAX: the clamped
// tap they stand for duplicates a cell that is already inside the window, so they must not win a min.
template <int E>
__global__ __launch_bounds__(CT) void erosion_reg_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                        nz_geom g, int aligned) {
    __shared__ float4 s_edge[TH / RB][E][TW / 4];
    constexpr float BIG = 3.40282347e+38f;
    constexpr int HX = 4;
    const int tid = threadIdx.x, cg = tid & 31, rb = tid >> 5;
    const int OW = TW - HX, OH = TH - E;
    int ox0, oz0;
    tile_origin(g, OW, OH, ox0, oz0);
    const int lx0 = ox0 - HX, lz0 = oz0 - E;
    const int gx0 = lx0 + cg * 4, gzb = lz0 + rb * RB;
    const bool inside = lx0 >= 0 && lx0 + TW <= g.cols && lz0 >= g.zc0 && lz0 + TH - 1 <= g.zc1;
    const bool fast = inside && aligned;
    float v[RB][4];
#pragma unroll
    for (int r = 0; r < RB; r++) {
        int gz = gzb + r;
        if (fast) {
            float4 t = *reinterpret_cast<const float4 *>(src + (size_t)gz * g.pitch + gx0);
            v[r][0] = t.x; v[r][1] = t.y; v[r][2] = t.z; v[r][3] = t.w;
        } else {
            bool zin = gz >= g.zc0 && gz <= g.zc1;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                int gx = gx0 + e;
                v[r][e] = (zin && gx >= 0 && gx < g.cols) ? src[(size_t)gz * g.pitch + gx] : BIG;
            }
        }
    }
    // X: min over [x-E, x]
#pragma unroll
    for (int r = 0; r < RB; r++) {
        float w[4 + E];
#pragma unroll
        for (int o = 0; o < E; o++) w[o] = dpp_prev(v[r][4 - E + o]);
        if (cg == 0) {  // no lane to the left inside this tile row: halo garbage, keep it neutral
#pragma unroll
            for (int o = 0; o < E; o++) w[o] = BIG;
        }
#pragma unroll
        for (int e = 0; e < 4; e++) w[E + e] = v[r][e];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            float m = w[e];
#pragma unroll
            for (int k = 1; k <= E; k++) m = fminf(m, w[e + k]);
            v[r][e] = m;
        }
    }
    // Z: min over [z-E, z]; the E rows above come from the block above
#pragma unroll
    for (int o = 0; o < E; o++)
        s_edge[rb][o][cg] = make_float4(v[RB - E + o][0], v[RB - E + o][1], v[RB - E + o][2], v[RB - E + o][3]);
    __syncthreads();
    float z[RB + E][4];
#pragma unroll
    for (int o = 0; o < E; o++) {
        float4 a = rb > 0 ? s_edge[rb - 1][o][cg] : make_float4(BIG, BIG, BIG, BIG);
        z[o][0] = a.x; z[o][1] = a.y; z[o][2] = a.z; z[o][3] = a.w;
    }
#pragma unroll
    for (int r = 0; r < RB; r++)
#pragma unroll
        for (int e = 0; e < 4; e++) z[E + r][e] = v[r][e];
#pragma unroll
    for (int r = 0; r < RB; r++) {
        int lr = rb * RB + r, gz = gzb + r;
        float out[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            float m = z[r][e];
#pragma unroll
            for (int k = 1; k <= E; k++) m = fminf(m, z[r + k][e]);
            out[e] = m;
        }
        bool in = lr >= E && cg * 4 >= HX && gz < g.or1 && gx0 < g.cols;
        if (in) {
            if (aligned && gx0 + 4 <= g.cols) {
                *reinterpret_cast<float4 *>(dst + (size_t)gz * g.pitch + gx0) = make_float4(out[0], out[1], out[2], out[3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (gx0 + e < g.cols) dst[(size_t)gz * g.pitch + gx0 + e] = out[e];
            }
        }
    }
}

// generic single passes straight from global memory (any odd/even kernelSize <= 25); neighbour
// reuse is served by L1/L2.  One thread per cell.
__global__ __launch_bounds__(CT) void conv_pass_x_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                        nz_geom g, nz_kernel_taps taps) {
    int x = blockIdx.x * CT + threadIdx.x;
    int z = g.or0 + blockIdx.y;
    if (x >= g.cols || z >= g.or1) return;
    const int k_off = (taps.ksize - 1) / 2;
    const float *row = src + (size_t)z * g.pitch;
    float total = 0.0f;
    for (int k = -k_off; k <= k_off; k++) total += row[clampi(x + k, 0, g.cols - 1)] * taps.kx[k_off + k];
    dst[(size_t)z * g.pitch + x] = total * taps.factor;
}

__global__ __launch_bounds__(CT) void conv_pass_z_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                        nz_geom g, nz_kernel_taps taps) {
    int x = blockIdx.x * CT + threadIdx.x;
    int z = g.or0 + blockIdx.y;
    if (x >= g.cols || z >= g.or1) return;
    const int k_off = (taps.ksize - 1) / 2;
    float total = 0.0f;
    for (int k = k_off; k >= -k_off; k--)
        total += src[(size_t)clampi(z + k, g.zc0, g.zc1) * g.pitch + x] * taps.kz[k_off - k];
    dst[(size_t)z * g.pitch + x] = total * taps.factor;
}

// single min passes with the reference's window k in [-k_off, k_off) (KernelOperators.cs:84-90,109-116)
__global__ __launch_bounds__(CT) void min_pass_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                     nz_geom g, int k_off, int along_z) {
    int x = blockIdx.x * CT + threadIdx.x;
    int z = g.or0 + blockIdx.y;
    if (x >= g.cols || z >= g.or1) return;
    float v = 3.40282347e+38f;
    for (int k = -k_off; k < k_off; k++) {
        int xx = along_z ? x : clampi(x + k, 0, g.cols - 1);
        int zz = along_z ? clampi(z + k, g.zc0, g.zc1) : z;
        v = fminf(v, src[(size_t)zz * g.pitch + xx]);
    }
    dst[(size_t)z * g.pitch + x] = v;
}

#ifndef NZ_CONV_NT
#define NZ_CONV_NT 256
#endif

template <int KS>
int32_t launch_fused(hipStream_t s, const float *src, float *dst, const nz_geom &g, const nz_kernel_taps &k, int T) {
    constexpr int O = (KS - 1) / 2;
    int H = T * O, HX = (H + 3) & ~3;
    static const int use_lds = getenv("NZ_CONV_LDS") ? atoi(getenv("NZ_CONV_LDS")) : 0;
    if (use_lds) {
        int OW = TW - 2 * HX, OH = TH - 2 * H;
        long long blocks = (long long)((g.cols + OW - 1) / OW) * ((g.or1 - g.or0 + OH - 1) / OH);
        hipLaunchKernelGGL((conv_fused_kernel<KS>), dim3((unsigned)blocks), dim3(CT), 0, s, src, dst, g, k, T);
    } else {
        constexpr int NT = NZ_CONV_NT, RTH = NT / 32 * RB;  // register tile: RTH rows x 128 columns
        int OW = TW - 2 * HX, OH = RTH - 2 * H;
        long long blocks = (long long)((g.cols + OW - 1) / OW) * ((g.or1 - g.or0 + OH - 1) / OH);
        int aligned = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | (uintptr_t)(g.pitch * 4)) & 15) == 0;
        if (k.factor == 1.0f)
            hipLaunchKernelGGL((conv_reg_kernel<KS, true, NT>), dim3((unsigned)blocks), dim3(NT), 0, s, src, dst, g, k, T, aligned);
        else
            hipLaunchKernelGGL((conv_reg_kernel<KS, false, NT>), dim3((unsigned)blocks), dim3(NT), 0, s, src, dst, g, k, T, aligned);
    }
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

}  // namespace

// largest T with a useful interior left (OH >= 32, OW >= 96); 0 = this size has no fused kernel
int nz_conv_max_fused(int ksize) {
    switch (ksize) {
        case 3: return 8;
        case 5: return 8;
        case 7: return 5;
        case 9: return 4;
        default: return 0;
    }
}

int32_t nz_launch_conv_fused(hipStream_t s, const float *src, float *dst, const nz_geom &g,
                             const nz_kernel_taps &k, int T) {
    if (T < 1 || T > nz_conv_max_fused(k.ksize)) {
        nz_set_error("conv_fused: T=%d unsupported for kernelSize %d", T, k.ksize);
        return NZ_ERR_INVALID;
    }
    if (g.or1 <= g.or0) return NZ_OK;
    switch (k.ksize) {
        case 3: return launch_fused<3>(s, src, dst, g, k, T);
        case 5: return launch_fused<5>(s, src, dst, g, k, T);
        case 7: return launch_fused<7>(s, src, dst, g, k, T);
        case 9: return launch_fused<9>(s, src, dst, g, k, T);
    }
    return NZ_ERR_INVALID;
}

int32_t nz_launch_conv_pass_x(hipStream_t s, const float *src, float *dst, const nz_geom &g,
                              const nz_kernel_taps &k) {
    if (g.or1 <= g.or0) return NZ_OK;
    dim3 grid((g.cols + CT - 1) / CT, g.or1 - g.or0);
    hipLaunchKernelGGL(conv_pass_x_kernel, grid, dim3(CT), 0, s, src, dst, g, k);
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

int32_t nz_launch_conv_pass_z(hipStream_t s, const float *src, float *dst, const nz_geom &g,
                              const nz_kernel_taps &k) {
    if (g.or1 <= g.or0) return NZ_OK;
    dim3 grid((g.cols + CT - 1) / CT, g.or1 - g.or0);
    hipLaunchKernelGGL(conv_pass_z_kernel, grid, dim3(CT), 0, s, src, dst, g, k);
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

int nz_erosion_max_fused() { return 4; }

int32_t nz_launch_erosion_fused(hipStream_t s, const float *src, float *dst, const nz_geom &g, int E) {
    if (E < 1 || E > nz_erosion_max_fused()) {
        nz_set_error("erosion_fused: E=%d unsupported", E);
        return NZ_ERR_INVALID;
    }
    if (g.or1 <= g.or0) return NZ_OK;
    int OW = TW - 4, OH = TH - E;
    long long blocks = (long long)((g.cols + OW - 1) / OW) * ((g.or1 - g.or0 + OH - 1) / OH);
    int aligned = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | (uintptr_t)(g.pitch * 4)) & 15) == 0;
    switch (E) {
        case 1: hipLaunchKernelGGL((erosion_reg_kernel<1>), dim3((unsigned)blocks), dim3(CT), 0, s, src, dst, g, aligned); break;
        case 2: hipLaunchKernelGGL((erosion_reg_kernel<2>), dim3((unsigned)blocks), dim3(CT), 0, s, src, dst, g, aligned); break;
        case 3: hipLaunchKernelGGL((erosion_reg_kernel<3>), dim3((unsigned)blocks), dim3(CT), 0, s, src, dst, g, aligned); break;
        default: hipLaunchKernelGGL((erosion_reg_kernel<4>), dim3((unsigned)blocks), dim3(CT), 0, s, src, dst, g, aligned); break;
    }
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

int32_t nz_launch_min_pass(hipStream_t s, const float *src, float *dst, const nz_geom &g, int ksize, int along_z) {
    if (g.or1 <= g.or0) return NZ_OK;
    dim3 grid((g.cols + CT - 1) / CT, g.or1 - g.or0);
    hipLaunchKernelGGL(min_pass_kernel, grid, dim3(CT), 0, s, src, dst, g, (ksize - 1) / 2, along_z);
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}
