// nz_filter.hip -- separable convolutions and the min "value erosion" filter (gfx950).
//
// Replaces GenericKernelJob<KernelTileMutation<KernelSample{X,Z}Operator>, RWTileData>
// (Filter/Kernel/KernelJob.cs:18-54,165-185; operators Filter/Kernel/KernelOperators.cs:18-67) and
// the KernelMin{X,Z}Operator pair of ErosionKernelJob (KernelJob.cs:317-347, KernelOperators.cs:69-118).
//
// conv_reg_kernel / erosion_reg_kernel: one workgroup owns a TH x 128 tile (rows x cols, halo included;
// TH = 128 rows with 512 threads for 5..9 taps, 64 rows with 256 threads for 3 taps and the min filter)
// held entirely in registers, runs T applications of (X pass, Z pass) on it without touching HBM and
// stores the (TH-2H) x (128-2HX) interior, H = T*(K-1)/2.  conv_wide_kernel (11..25 taps) runs one
// application per launch through two LDS planes.  HBM traffic per launch is one read and one
// write of the plane for T filter applications (the reference moves 32 B/cell per application: two
// passes + two serial flush copies).  The FlushWriteSlice copies (Pipeline/Tiles/TileData.cs:15-40) do
// not exist here: launches ping-pong between `src` and `tmp`.
//
// Arithmetic order is the reference's: X pass sums taps k ascending, Z pass sums k descending
// (KernelOperators.cs:34-40,59-65), product then add (no FMA contraction), then * factor.
// Clamp-to-edge (TileData.cs:72-77) is applied per pass when a window is assembled.
#include <cstdlib>

#include "nz_internal.hpp"

namespace {

constexpr int CT = 256;      // threads per workgroup
constexpr int TW = 128;      // LDS tile width, cells
// columns of x halo a fused tile carries on each side for H = T * O applications' worth of reach: a multiple of four (16-byte
// loads).  (Round 6 tried 16 wherever that is enough, so that the interior a tile STORES is 96 columns = whole 128-byte lines --
// 104 columns are 416 bytes at offsets that are multiples of 416, every row of a tile's store ends in partial lines, which is
// where WRITE_SIZE = 5.18 planes for four launches comes from -- and lost: 8 % more tiles, Gauss5 x17 0.189 -> 0.207 ms.)
__host__ __device__ constexpr int conv_hx(int H) { return (H + 3) & ~3; }
constexpr int TH = 64;       // LDS tile height, rows

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Workgroups are dealt round-robin to the 8 XCDs (blockIdx.x % 8), each with its own L2.  Giving XCD x the x-th
// contiguous eighth of the tiles keeps tiles that share halo rows / columns on one L2 (NZ_XCD_REMAP=0: identity).
#ifndef NZ_XCD_REMAP
#define NZ_XCD_REMAP 1
#endif
__device__ __forceinline__ int xcd_tile_index() {
    int bid = blockIdx.x;
#if NZ_XCD_REMAP
    int nb = gridDim.x, q = nb >> 3, r = nb & 7;
    int x = bid & 7, k = bid >> 3;
    bid = x * q + (x < r ? x : r) + k;
#endif
    return bid;
}

__device__ __forceinline__ void tile_origin(const nz_geom &g, int OW, int OH, int &ox0, int &oz0) {
    int tiles_x = (g.cols + OW - 1) / OW;
    int bid = xcd_tile_index();
    int by = bid / tiles_x;
    int bx = bid - by * tiles_x;
    ox0 = bx * OW;
    oz0 = g.or0 + by * OH;
}

// ---- register-resident fused kernel ------------------------------------------------------------------
// T applications of X pass + Z pass on a (NT/4) x 128 tile (halo included; 64 rows for 3 taps, 128 rows otherwise).
// The tile never sits in LDS: every thread keeps a 4-column x 8-row block in registers for the whole launch.  The
// X pass takes its (K-1)/2 west / east neighbours from the adjacent lanes with wave-shift DPP moves; the Z pass needs
// (K-1)/2 rows from the thread above and below, and only those boundary rows travel through LDS (16-byte accesses;
// double buffered with one barrier per application for 3 taps, one buffer and two barriers otherwise).  Global loads
// and stores go register <-> HBM directly, 16 B per lane.  Clamp-to-edge is applied when a window is assembled: a tap
// beyond the grid takes the border cell's current value (RWTileData.GetData, Pipeline/Tiles/TileData.cs:72-82), so no
// fix-up pass is needed.
constexpr int RB = 8;  // rows per thread

// bound_ctrl: the lane without a source reads 0 -- the value update_dpp(0, ...) left there, without the v_mov 0 that
// initialised its destination (a VALU slot per shifted value: 4 of ~85 per row of the 5-tap X pass)
#ifndef NZ_DPP_OLD
#define NZ_DPP_OLD 0
#endif
__device__ __forceinline__ float dpp_prev(float v) {  // lane i <- lane i-1
    if (NZ_DPP_OLD) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_next(float v) {  // lane i <- lane i+1
    if (NZ_DPP_OLD) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

// -DNZ_CONV_PROBE: thread 0 of every workgroup stamps s_memrealtime (100 MHz) at the start, after each application and
// at the end, plus its HW_ID / XCC_ID, into a caller-supplied buffer (tools/probe_conv_phases.py).  Never built by the Makefile.
#ifdef NZ_CONV_PROBE
__device__ unsigned long long *nz_probe_buf = nullptr;  // [workgroup][24]
// the buffer's address is read ONCE per workgroup (a __device__ variable re-read at every stamp would put a ~1 us load in front
// of each of them); conv_tile receives it as an extra argument
#define NZ_PB_PARAM , unsigned long long *nz_pb
#define NZ_PB_ARG , nz_pb
#define NZ_PB_INIT unsigned long long *nz_pb = threadIdx.x == 0 ? nz_probe_buf : nullptr
#define NZ_PROBE(slot, val)                                              \
    do {                                                                 \
        if (nz_pb) nz_pb[(size_t)blockIdx.x * 24 + (slot)] = (val);      \
    } while (0)
#define NZ_PROBE_T(slot) NZ_PROBE(slot, __builtin_amdgcn_s_memrealtime())
#else
#define NZ_PB_PARAM
#define NZ_PB_ARG
#define NZ_PB_INIT
#define NZ_PROBE(slot, val)
#define NZ_PROBE_T(slot)
#endif

// Register budget per tap count (waves per SIMD the allocator is asked to fit): the 5-tap kernel runs three
// 512-thread workgroups per CU (6 waves per SIMD, 80 VGPRs, one edge buffer of 32 KB each) so one workgroup's
// load/store phase overlaps the others' VALU phases: 5 % faster than two workgroups at 121 VGPRs; the 7- and
// 9-tap windows do not fit 80 registers without spilling and stay at two.
#ifndef NZ_CONV5_NBUF
#define NZ_CONV5_NBUF 1  // 2: double-buffered edge rows for the 5-tap kernel (one barrier per application, 64 KB of LDS: two workgroups per CU)
#endif
#ifndef NZ_CONV5_WAVES
#define NZ_CONV5_WAVES 6
#endif
// edge-row buffers: two (one barrier per application) for the 3-tap kernel, one (two barriers) for the others -- see conv_tile.
// (The small grids' shapes with two: Gauss5 x17 512^2 30.1 -> 29.5 us, 1024^2 47.8 -> 48.8, 2048^2 67.9 -> 76.9: one it stays.)
constexpr int conv_nbuf(int o) { return (o == 2 && NZ_CONV5_NBUF == 2) ? 2 : (o >= 2 ? 1 : 2); }
constexpr int conv_waves(int ks, int nt, int rbt) { return rbt < RB ? 4 : ks == 5 ? NZ_CONV5_WAVES : (nt >= 512 || ks >= 7) ? 4 : 6; }

// 16-byte accesses that other XCDs can see / that see other XCDs' stores while the kernel runs: `sc1` buffer loads and
// stores (they bypass the CU's L1; the stores write through and leave the XCD's L2), 4-byte ones as agent-scope relaxed
// atomics (global_load/store_dword sc1).  Used by the chained kernel below, where a tile of launch l + 1 reads what
// workgroups of launch l on other XCDs stored microseconds earlier (MI355X_MICROARCH.md, inter-workgroup visibility).
typedef unsigned nz_v4u __attribute__((ext_vector_type(4)));
constexpr int NZ_AUX_SC1 = 16;
__device__ __forceinline__ float4 load16_sc1(const float *base, size_t float_off) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, -1, 0x00020000);
    nz_v4u t = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(float_off * 4), 0, NZ_AUX_SC1);
    // (__builtin_bit_cast on the vector's ELEMENTS is miscompiled by ROCm 7.2's clang: the load is narrowed to one
    // dword and x is broadcast; __uint_as_float is not)
    return make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
}
__device__ __forceinline__ void store16_sc1(float *base, size_t float_off, float4 v) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, -1, 0x00020000);
    nz_v4u t = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(t, r, (int)(float_off * 4), 0, NZ_AUX_SC1);
}
// 4-byte forms for edge tiles: buffer accesses again (agent-scope atomic loads would each be waited for on their own)
__device__ __forceinline__ float load4_sc1(const float *base, size_t float_off) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, -1, 0x00020000);
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)(float_off * 4), 0, NZ_AUX_SC1));
}
__device__ __forceinline__ void store4_sc1(float *base, size_t float_off, float v) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, -1, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)(float_off * 4), 0, NZ_AUX_SC1);
}

// One workgroup's tile: T applications on the (NT/4) x 128 register tile whose interior starts at (ox0, oz0).
// SC1: every global access is an sc1 access (see above); the plane must then be smaller than 4 GiB (32-bit offsets).
// FAST (NZ_FLOAT_FAST): the tap sums as one multiply and KS - 1 FMAs, same tap order (the reference compiles its kernel jobs
// with FloatMode.Fast, Filter/Kernel/KernelJob.cs:17: Burst is free to contract exactly these sums)
template <bool FAST>
__device__ __forceinline__ float tap_acc(float total, float v, float k) {
    return FAST ? __builtin_fmaf(v, k, total) : total + v * k;
}

template <int KS, bool UNIT, int NT, bool SC1, bool FAST, int RBT>
__device__ __forceinline__ void conv_tile(const float *__restrict__ src, float *__restrict__ dst, const nz_geom &g,
                                          const nz_kernel_taps &taps, int T, int aligned, int ox0, int oz0,
                                          float4 *s_edge_raw NZ_PB_PARAM) {
    constexpr int O = (KS - 1) / 2;
    constexpr int WN = 4 + 2 * O;   // X window
    constexpr int ZN = RBT + 2 * O;  // Z window
    constexpr int TH = NT / 32 * RBT;  // tile rows: one RBT-row block per 32 threads (shadows the file-level TH)
    static_assert(O <= RBT, "a block's window reaches into the neighbouring blocks only");
    // boundary rows of every 8-row block: [parity][block][top|bottom][o][column group].  Double buffered
    // (one barrier per application) for the 3-tap kernel; the others keep one buffer and pay a second barrier
    // instead of giving up a resident workgroup (5 taps: 3 x 32 KB; 7/9 taps: 2 x 48/64 KB).
    constexpr int NBUF = conv_nbuf(O);
    float4 (*s_edge)[TH / RBT][2][O][TW / 4] = reinterpret_cast<float4 (*)[TH / RBT][2][O][TW / 4]>(s_edge_raw);

    const int tid = threadIdx.x, cg = tid & 31, rb = tid >> 5;
    const int H = T * O, HX = conv_hx(H);
    const int OW = TW - 2 * HX, OH = TH - 2 * H;
    const int lx0 = ox0 - HX, lz0 = oz0 - H;
    const int gx0 = lx0 + cg * 4, gzb = lz0 + rb * RBT;
    const bool inside = lx0 >= 0 && lx0 + TW <= g.cols && lz0 >= g.zc0 && lz0 + TH - 1 <= g.zc1;
    const bool fast = inside && aligned;
    NZ_PROBE_T(0);
    NZ_PROBE(14, (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)));   // HW_ID
    NZ_PROBE(15, (unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)));  // XCC_ID

    float v[RBT][4];
    if (fast) {  // a real branch (the asm keeps the two forms from being merged into 4-byte accesses with selected addresses)
#pragma unroll
        for (int r = 0; r < RBT; r++) {
            float4 t = SC1 ? load16_sc1(src, (size_t)(gzb + r) * g.pitch + gx0)
                           : *reinterpret_cast<const float4 *>(src + (size_t)(gzb + r) * g.pitch + gx0);
            v[r][0] = t.x; v[r][1] = t.y; v[r][2] = t.z; v[r][3] = t.w;
        }
        asm volatile("; 16-byte loads issued" ::: "memory");  // after the loads: they cannot be sunk into a common tail
    } else {
        asm volatile("; clamped 4-byte loads of an edge tile" ::: "memory");
#pragma unroll
        for (int r = 0; r < RBT; r++) {
            size_t row = (size_t)clampi(gzb + r, g.zc0, g.zc1) * g.pitch;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const size_t off = row + clampi(gx0 + e, 0, g.cols - 1);
                v[r][e] = SC1 ? load4_sc1(src, off) : src[off];
            }
        }
    }
#ifdef NZ_CONV_PROBE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    NZ_PROBE_T(9);  // wave 0's tile rows have landed (the wait exists in this build only)
#endif
    // window indices of the last grid column / first and last grid rows, for the clamps of edge tiles
    const int icx = g.cols - 1 - gx0 + O;
    const int iz0 = g.zc0 - gzb + O, iz1 = g.zc1 - gzb + O;

    for (int t = 0; t < T; t++) {
        // ---- X pass (KernelSampleXOperator: taps k ascending), in place row by row
#pragma unroll
        for (int r = 0; r < RBT; r++) {
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
                for (int kk = 1; kk < KS; kk++) total = tap_acc<FAST>(total, w[e + kk], taps.kx[kk]);
                v[r][e] = UNIT ? total : total * taps.factor;
            }
        }
        // ---- exchange the block's boundary rows
        const int par = NBUF == 2 ? (t & 1) : 0;
#pragma unroll
        for (int o = 0; o < O; o++) {
            s_edge[par][rb][0][o][cg] = make_float4(v[o][0], v[o][1], v[o][2], v[o][3]);
            s_edge[par][rb][1][o][cg] = make_float4(v[RBT - O + o][0], v[RBT - O + o][1], v[RBT - O + o][2], v[RBT - O + o][3]);
        }
        if (t == 0) NZ_PROBE_T(16);  // application 1: X pass done, edge rows written
        __syncthreads();
        if (t == 0) NZ_PROBE_T(17);  // ... and everybody has arrived
        float z[ZN][4];
#pragma unroll
        for (int o = 0; o < O; o++) {
            // rows gzb-O+o (bottom rows of the block above) and gzb+RBT+o (top rows of the block below)
            float4 a = rb > 0 ? s_edge[par][rb - 1][1][o][cg] : make_float4(v[0][0], v[0][1], v[0][2], v[0][3]);
            float4 b = rb < TH / RBT - 1 ? s_edge[par][rb + 1][0][o][cg]
                                        : make_float4(v[RBT - 1][0], v[RBT - 1][1], v[RBT - 1][2], v[RBT - 1][3]);
            z[o][0] = a.x; z[o][1] = a.y; z[o][2] = a.z; z[o][3] = a.w;
            z[RBT + O + o][0] = b.x; z[RBT + O + o][1] = b.y; z[RBT + O + o][2] = b.z; z[RBT + O + o][3] = b.w;
        }
        if (NBUF == 1 && t + 1 < T) __syncthreads();  // everyone has read the edges before they are rewritten
#pragma unroll
        for (int r = 0; r < RBT; r++)
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
        for (int r = 0; r < RBT; r++) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float total = z[r + 2 * O][e] * taps.kz[0];
#pragma unroll
                for (int kk = 1; kk < KS; kk++) total = tap_acc<FAST>(total, z[r + 2 * O - kk][e], taps.kz[kk]);
                v[r][e] = UNIT ? total : total * taps.factor;
            }
        }
        NZ_PROBE_T(1 + t);
    }

    // ---- store the interior
    if (fast) {  // whole tile inside the grid, 16-byte aligned rows: one 16-byte store per row
#pragma unroll
        for (int r = 0; r < RBT; r++) {
            int lr = rb * RBT + r, gz = gzb + r;
            bool in = lr >= H && lr < H + OH && cg * 4 >= HX && cg * 4 < HX + OW && gz < g.or1;
            if (in) {
                if (SC1) store16_sc1(dst, (size_t)gz * g.pitch + gx0, make_float4(v[r][0], v[r][1], v[r][2], v[r][3]));
                else *reinterpret_cast<float4 *>(dst + (size_t)gz * g.pitch + gx0) = make_float4(v[r][0], v[r][1], v[r][2], v[r][3]);
            }
        }
    } else {
        asm volatile("; guarded stores of an edge tile" ::: "memory");
#pragma unroll
        for (int r = 0; r < RBT; r++) {
            int lr = rb * RBT + r, gz = gzb + r;
            bool in = lr >= H && lr < H + OH && cg * 4 >= HX && cg * 4 < HX + OW && gz < g.or1 && gx0 < g.cols;
            if (in) {
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (gx0 + e < g.cols) {
                        if (SC1) store4_sc1(dst, (size_t)gz * g.pitch + gx0 + e, v[r][e]);
                        else dst[(size_t)gz * g.pitch + gx0 + e] = v[r][e];
                    }
            }
        }
    }
    NZ_PROBE_T(12);
}

template <int KS, bool UNIT, int NT, bool FAST, int RBT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(conv_waves(KS, NT, RBT)))) void conv_reg_kernel(const float *__restrict__ src, float *__restrict__ dst, nz_geom g,
                                                     nz_kernel_taps taps, int T, int aligned) {
    constexpr int O = (KS - 1) / 2;
    constexpr int TH = NT / 32 * RBT;
    constexpr int NBUF = conv_nbuf(O);
    __shared__ float4 s_edge[NBUF][TH / RBT][2][O][TW / 4];
    const int H = T * O, HX = conv_hx(H);
    src += blockIdx.y * g.bstride;  // batched launch: one independent grid per blockIdx.y
    dst += blockIdx.y * g.bstride;
    int ox0, oz0;
    tile_origin(g, TW - 2 * HX, TH - 2 * H, ox0, oz0);
    NZ_PB_INIT;
    conv_tile<KS, UNIT, NT, false, FAST, RBT>(src, dst, g, taps, T, aligned, ox0, oz0, &s_edge[0][0][0][0][0] NZ_PB_ARG);
}

// ---- the launches of a stage as ONE grid with tile-level dependencies -------------------------------------------------
// KernelFilterStage's `iterations` applications are L fused launches that ping-pong between two planes.  Run as L
// kernels, every launch is two rounds of workgroups that load together (idle SIMDs), compute together (idle memory)
// and store together, and the chip drains at every launch boundary.  Here the tiles of ALL L launches form one grid:
// a workgroup takes the next work item of its class (class = blockIdx.x & 7, the workgroups an XCD receives; items
// are dealt so that a class works through a contiguous eighth of every launch's tiles, in launch order), waits until
// the <= 9 tiles of the previous launch that its input window touches have been stored, and runs the same tile code
// with sc1 accesses.  Loads of launch l + 1 overlap arithmetic of launch l, and nothing drains in between.
//   * a workgroup's work item is its position in the grid (class = blockIdx.x & 7, k = blockIdx.x >> 3; no ticket since round
//     5): the producers of an item are earlier items of its own or of other classes, which must get dispatched: progress rests on
//     the hardware starting a grid's workgroups in index order, round-robin over the XCDs (observed, not promised).
//     Termination does not: the poll is bounded, see below, and a context whose chained launch ever timed out runs separate
//     launches from then on (nz_runtime.cpp, ctx_chain_check);
//   * hand-off: producer = sc1 stores, every wave s_waitcnt vmcnt(0), workgroup barrier, ONE lane stores the tile's
//     flag (agent scope); consumer = up to nine lanes poll one flag each (sc1 loads), workgroup barrier, sc1 loads;
//   * the poll is bounded: a workgroup that gives up raises the context's error word (mapped host memory) and carries on, so the grid always
//     drains and the host reports the failure at its next synchronisation instead of hanging;
//   * flags carry the launch's epoch: nothing is cleared between stages.
// (A persistent form -- resident workgroups that claim item after item and let a tile's stores drain behind the next
// tile's loads -- was built and measured: 0.555 ms for Gauss5 x17 against 0.197 ms, the loop-carried state costs the
// 80-register budget 39 spills.  One item per workgroup it stays.)
constexpr int NZ_CHAIN_MAXL = 8;
struct nz_chain {
    int L, total;
    int T[NZ_CHAIN_MAXL], first[NZ_CHAIN_MAXL + 1], tiles_x[NZ_CHAIN_MAXL];
    unsigned epoch;
    int *flags;       // one per work item, indexed first[l] + tile
    float *plane[2];  // launch l reads plane[l & 1] and writes plane[(l + 1) & 1]
    // test hook (nz_debug_chain_delay): the workgroup that claims work item `delay_item` sleeps `delay_sleeps` x ~1 us
    // between its dependency wait and its loads -- a straggler among the readers of a plane that later launches
    // overwrite; -1: nobody
    int delay_item, delay_sleeps;
    // a consumer gives up after `spin_limit` polls (~2 us each) and raises *err_host, a word in mapped host memory the
    // host reads -- with a plain load -- wherever it waits for the context
    int spin_limit;
    unsigned *err_host;
    unsigned *err_epoch;  // device memory: the lowest epoch of a launch that gave up (which work the failure belongs to)
};

__host__ __device__ __forceinline__ int chain_class_count(int n, int c) { return n > c ? (n - c + 7) >> 3 : 0; }  // #{vb < n : vb % 8 == c}

template <int KS, bool UNIT, int NT, bool FAST, int RBT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(conv_waves(KS, NT, RBT)))) void conv_chain_kernel(nz_geom g, nz_kernel_taps taps,
                                                                                                                       nz_chain ch, int aligned) {
    constexpr int O = (KS - 1) / 2;
    constexpr int TH = NT / 32 * RBT;
    constexpr int NBUF = conv_nbuf(O);
    __shared__ float4 s_edge[NBUF][TH / RBT][2][O][TW / 4];
    NZ_PB_INIT;
    NZ_PROBE_T(18);  // the workgroup's first instruction (after the kernel arguments' scalar loads)
    int l = 0, tile = 0;
    {
        // Work item = position in the grid: vb = blockIdx.x, class = vb & 7 (the XCD that receives the workgroup), k = vb >> 3.
        // (Rounds 2 .. 4 drew k from an atomic ticket per class, which made the order within a class the order in which
        // workgroups actually start -- and put a returning atomic, a write to LDS and a barrier in front of every tile: ~2 us of
        // a tile's ~22, Gauss5 x17 0.203 -> 0.189 ms without it.  The dispatcher starts a grid's workgroups in index order, so
        // within a class a lower k has started whenever a higher one runs; the hand-off already rested on that order ACROSS
        // classes, and the wait is bounded either way.)
        const int vb = blockIdx.x;
        const int cls = vb & 7;
        while (l + 1 < ch.L && vb >= ch.first[l + 1]) l++;
        // the class's share of launch l is a contiguous run of tiles: classes before it, then its own earlier items
        int start = 0;
        for (int c = 0; c < cls; c++) start += chain_class_count(ch.first[l + 1], c) - chain_class_count(ch.first[l], c);
        // Odd classes walk their run backwards.  A run's first row of tiles reads the last row of the run before it:
        // if every class walked forwards, each would start launch l + 1 on the tiles whose producers its neighbour
        // finishes launch l with -- a row of workgroups spinning at every launch boundary.  Walking towards each
        // other, neighbouring classes finish a launch on the rows they share and start the next one at the far ends.
        const int mine = chain_class_count(ch.first[l + 1], cls) - chain_class_count(ch.first[l], cls);
        const int j = chain_class_count(vb, cls) - chain_class_count(ch.first[l], cls);
        tile = start + ((cls & 1) ? mine - 1 - j : j);
    }
    NZ_PROBE_T(7);  // the ticket is back
    NZ_PROBE(13, ((unsigned long long)l << 32) | (unsigned)tile);
    const int T = ch.T[l], H = T * O, HX = conv_hx(H);
    const int OW = TW - 2 * HX, OH = TH - 2 * H;
    const int by = tile / ch.tiles_x[l], bx = tile - by * ch.tiles_x[l];
    const int ox0 = bx * OW, oz0 = g.or0 + by * OH;
    if (l > 0) {
        const int Tp = ch.T[l - 1], Hp = Tp * O, HXp = conv_hx(Hp);
        const int OWp = TW - 2 * HXp, OHp = TH - 2 * Hp;
        // Awaited: the launch-(l-1) tiles whose interior meets this tile's input window (their stores are read here) or
        // lies within THEIR halo of this tile's interior (they read the cells this tile overwrites: the ping-pong plane
        // is shared).  The second set is inside the first while H(l-1) <= H(l), which the host guarantees; the window is
        // widened to the larger halo all the same, so the protocol does not rest on that order.
        const int Hw = max(H, Hp), HXw = max(HX, HXp);
        const int ix0 = max(ox0 - HXw, 0), ix1 = min(ox0 + OW + HXw - 1, g.cols - 1);
        const int iz0 = max(oz0 - Hw, g.or0), iz1 = min(oz0 + OH + Hw - 1, g.or1 - 1);
        const int px0 = ix0 / OWp, px1 = ix1 / OWp, py0 = (iz0 - g.or0) / OHp, py1 = (iz1 - g.or0) / OHp;
        const int nx = px1 - px0 + 1, nd = nx * (py1 - py0 + 1);
        if ((int)threadIdx.x < nd) {
            const int dep = (py0 + (int)threadIdx.x / nx) * ch.tiles_x[l - 1] + px0 + (int)threadIdx.x % nx;
            const int *f = ch.flags + ch.first[l - 1] + dep;
            int spins = 0;
            while ((unsigned)__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ch.epoch) {
                __builtin_amdgcn_s_sleep(32);
                if (++spins > ch.spin_limit) {  // seconds: the producer is never coming
                    atomicMin(ch.err_epoch, ch.epoch);
                    __threadfence();
                    __hip_atomic_store(ch.err_host, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
            }
        }
        __syncthreads();
    }
    NZ_PROBE_T(8);  // the producers' flags are up
    if (ch.first[l] + tile == ch.delay_item)
        for (int i = 0; i < ch.delay_sleeps; i++) __builtin_amdgcn_s_sleep(127);  // 127 x 64 clocks ~ 3.4 us
    conv_tile<KS, UNIT, NT, true, FAST, RBT>(ch.plane[l & 1], ch.plane[(l + 1) & 1], g, taps, T, aligned, ox0, oz0, &s_edge[0][0][0][0][0] NZ_PB_ARG);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its stores have left
    NZ_PROBE_T(10);  // wave 0's stores are acknowledged
    __syncthreads();
    NZ_PROBE_T(11);  // everybody's are
    if (threadIdx.x == 0) {
        __hip_atomic_store(&ch.flags[ch.first[l] + tile], (int)ch.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}



// E applications of the {-1,0} min window fused as one window min over [x-E,x] x [z-E,z], register
// resident like conv_reg_kernel: the X pass reaches E <= 8 cells into the one or two lanes on the left
// (DPP), the Z pass E <= 8 rows into the block above (LDS).  Cells outside the grid are loaded as +FLT_MAX:
// the clamped tap they stand for duplicates a cell that is already inside the window, so they must not win
// a min.  The default five iterations are one launch (one read and one write of the plane).
template <int E>
__global__ __launch_bounds__(CT) void erosion_reg_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                        nz_geom g, int aligned) {
    __shared__ float4 s_edge[TH / RB][E][TW / 4];
    constexpr float BIG = 3.40282347e+38f;
    constexpr int HX = E > 4 ? 8 : 4;
    static_assert(E >= 1 && E <= RB, "the window reaches one 8-row block / two lanes back at most");
    const int tid = threadIdx.x, cg = tid & 31, rb = tid >> 5;
    const int OW = TW - HX, OH = TH - E;
    src += blockIdx.y * g.bstride;  // batched launch: one independent grid per blockIdx.y
    dst += blockIdx.y * g.bstride;
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
            asm volatile("; 16-byte load" ::: "memory");  // after the access: the two forms stay two forms
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
        if constexpr (E <= 4) {
#pragma unroll
            for (int o = 0; o < E; o++) w[o] = dpp_prev(v[r][4 - E + o]);
            if (cg == 0) {  // no lane to the left inside this tile row: halo garbage, keep it neutral
#pragma unroll
                for (int o = 0; o < E; o++) w[o] = BIG;
            }
        } else {  // cells x-E..x-5 sit two lanes to the left
            // every lane must execute the shifts (a lane masked off would hand its neighbour the DPP `old`
            // value): the empty asm pins each result before the select, which the compiler otherwise turns
            // into a branch around the move
            float p1[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float t = dpp_prev(v[r][e]);
                asm volatile("" : "+v"(t));
                p1[e] = cg == 0 ? BIG : t;
            }
#pragma unroll
            for (int o = 0; o < E - 4; o++) {
                float t = dpp_prev(p1[8 - E + o]);
                asm volatile("" : "+v"(t));
                w[o] = cg == 0 ? BIG : t;  // lane 1 receives lane 0's BIG through p1
            }
#pragma unroll
            for (int e = 0; e < 4; e++) w[E - 4 + e] = p1[e];
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
                asm volatile("; 16-byte store" ::: "memory");
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
template <bool FAST>
__global__ __launch_bounds__(CT) void conv_pass_x_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                        nz_geom g, nz_kernel_taps taps) {
    int x = blockIdx.x * CT + threadIdx.x;
    int z = g.or0 + blockIdx.y;
    if (x >= g.cols || z >= g.or1) return;
    const int k_off = (taps.ksize - 1) / 2;
    const float *row = src + (size_t)z * g.pitch;
    float total = 0.0f;
    for (int k = -k_off; k <= k_off; k++) total = tap_acc<FAST>(total, row[clampi(x + k, 0, g.cols - 1)], taps.kx[k_off + k]);
    dst[(size_t)z * g.pitch + x] = total * taps.factor;
}

template <bool FAST>
__global__ __launch_bounds__(CT) void conv_pass_z_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                        nz_geom g, nz_kernel_taps taps) {
    int x = blockIdx.x * CT + threadIdx.x;
    int z = g.or0 + blockIdx.y;
    if (x >= g.cols || z >= g.or1) return;
    const int k_off = (taps.ksize - 1) / 2;
    float total = 0.0f;
    for (int k = k_off; k >= -k_off; k--)
        total = tap_acc<FAST>(total, src[(size_t)clampi(z + k, g.zc0, g.zc1) * g.pitch + x], taps.kz[k_off - k]);
    dst[(size_t)z * g.pitch + x] = total * taps.factor;
}

// ---- wide separable kernels (odd kernelSize 11..25): both passes of one application in one launch -----
// Used by the Gaussian / box blur stages (StageGaussianBlur / StageSmoothBlur, width <= 25,
// Filter/Kernel/Blur/BlurJob.cs:11-52).  A workgroup of NT threads produces an (NT/8) x 128 output tile.  Source rows
// [z0-O, z0+32+O) x columns [x0-16, x0+144) (x0-12 .. x0+140 for the 64-row tiles of 23 / 25 taps), clamped to the grid
// (RWTileData.GetData), are staged in LDS as 160-float (152-float) rows: 16-byte loads.  The X pass runs on all 32+2O rows
// in place (see s_a), 8 consecutive outputs per thread from one register window; the Z pass reads
// those columns row after row, 4 columns x 4 rows per thread, and stores 16 bytes per lane.  Clamped source
// rows give the X-pass value of the clamped row, which is what the reference's Z pass reads after the flush.
constexpr int WD_W = 128;

template <int O, bool UNIT, int WD_NT, bool FAST>
__global__ __launch_bounds__(WD_NT) void conv_wide_kernel(const float *__restrict__ src, float *__restrict__ dst, nz_geom g,
                                                      nz_kernel_taps taps, int aligned) {
    constexpr int KS = 2 * O + 1;
    constexpr int WD_H = WD_NT / 8;           // output rows: 4 per 32 threads
    constexpr int NR = WD_H + 2 * O;          // staged rows
    // x halo staged on each side: 16 columns (five whole 128-byte lines per row) -- except for the 64-row tiles of 23 and 25
    // taps, whose 86 / 88 staged rows of 160 floats are 55 / 56 KB: two workgroups per CU.  With 12 columns a row is 152
    // floats and THREE workgroups fit the CU's 160 KB (round 6, together with the 8 x 2 Z pass below, which brings the
    // 512-thread kernels from 95-124 to <= 85 VGPRs: six waves per SIMD)
    constexpr int WD_XH = (WD_NT == 512 && O >= 11) ? 12 : 16;
    constexpr int WD_AP = WD_W + 2 * WD_XH;
    constexpr int OFF = WD_XH - O;            // first window column of output column 0
    constexpr int OFA = OFF & ~3, SH = OFF & 3;
    constexpr int NF = (SH + 8 + 2 * O + 3) / 4;  // float4s per X window
    static_assert(OFA + 8 * 15 + 4 * NF <= WD_AP, "the last thread's X window leaves the staged row");
    // One LDS plane: the X pass writes a row's results back over that row's source columns [WD_XH, WD_XH + 128).
    // The 16 threads of a row are lanes of one wave, whose LDS loads of the row all precede its stores in program
    // order, so no lane reads a column another lane has already replaced; half the LDS lets twice the workgroups
    // (of other tiles, in other phases) share the CU.
    __shared__ float4 s_a[NR * WD_AP / 4];
    const int tid = threadIdx.x;
    src += blockIdx.y * g.bstride;  // batched launch: one independent grid per blockIdx.y
    dst += blockIdx.y * g.bstride;
    const int tiles_x = (g.cols + WD_W - 1) / WD_W;
    const int bid = xcd_tile_index();
    const int by = bid / tiles_x, bx = bid - by * tiles_x;
    const int x0 = bx * WD_W, z0 = g.or0 + by * WD_H;

    const bool inside = aligned && x0 - WD_XH >= 0 && x0 + WD_W + WD_XH <= g.cols && z0 - O >= g.zc0 &&
                        z0 + WD_H + O - 1 <= g.zc1;
    for (int i = tid; i < NR * (WD_AP / 4); i += WD_NT) {
        int r = i / (WD_AP / 4), c4 = i - r * (WD_AP / 4);
        int gx = x0 - WD_XH + 4 * c4;
        float4 t;
        if (inside) {
            t = *reinterpret_cast<const float4 *>(src + (size_t)(z0 - O + r) * g.pitch + gx);
            asm volatile("; 16-byte load" ::: "memory");  // after the access: the two forms stay two forms
        } else {
            const float *row = src + (size_t)clampi(z0 - O + r, g.zc0, g.zc1) * g.pitch;
            t.x = row[clampi(gx, 0, g.cols - 1)];
            t.y = row[clampi(gx + 1, 0, g.cols - 1)];
            t.z = row[clampi(gx + 2, 0, g.cols - 1)];
            t.w = row[clampi(gx + 3, 0, g.cols - 1)];
        }
        s_a[i] = t;
    }
    __syncthreads();

    // ---- X pass (KernelSampleXOperator: taps k ascending)
    {
        const int tx = tid & 15, ry = tid >> 4;
        for (int r = ry; r < NR; r += WD_NT / 16) {
            float w[NF * 4];
            const float4 *a = s_a + (r * WD_AP + OFA + 8 * tx) / 4;
#pragma unroll
            for (int f = 0; f < NF; f++) {
                float4 t = a[f];
                w[4 * f] = t.x; w[4 * f + 1] = t.y; w[4 * f + 2] = t.z; w[4 * f + 3] = t.w;
            }
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                float total = w[SH + e] * taps.kx[0];  // 0 + a*b == a*b
#pragma unroll
                for (int kk = 1; kk < KS; kk++) total = tap_acc<FAST>(total, w[SH + e + kk], taps.kx[kk]);
                o[e] = UNIT ? total : total * taps.factor;
            }
            float4 *b = s_a + (r * WD_AP + WD_XH + 8 * tx) / 4;
            b[0] = make_float4(o[0], o[1], o[2], o[3]);
            b[1] = make_float4(o[4], o[5], o[6], o[7]);
        }
    }
    __syncthreads();

    // ---- Z pass (KernelSampleZOperator: taps k descending, first term is row z+o with K[0])
    if constexpr (WD_NT == 512) {
        // 8 rows x 2 columns per thread: a window of 8 + 2 O rows of two floats (64 registers at 25 taps; 4 x 4 holds
        // 28 rows of four = 112), 8-byte LDS reads (57 % of the 4 x 4 form's bytes), 8-byte stores (a wave writes 512
        // contiguous bytes of a row)
        const int cg = tid & 63, rg = tid >> 6;
        float v[8 + 2 * O][2];
        const float2 *col = reinterpret_cast<const float2 *>(s_a) + ((rg * 8) * WD_AP + WD_XH) / 2 + cg;
#pragma unroll
        for (int i = 0; i < 8 + 2 * O; i++) {
            float2 t = col[i * (WD_AP / 2)];
            v[i][0] = t.x; v[i][1] = t.y;
        }
        const int gx = x0 + 2 * cg;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float o[2];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                float total = v[j + 2 * O][e] * taps.kz[0];
#pragma unroll
                for (int kk = 1; kk < KS; kk++) total = tap_acc<FAST>(total, v[j + 2 * O - kk][e], taps.kz[kk]);
                o[e] = UNIT ? total : total * taps.factor;
            }
            int gz = z0 + rg * 8 + j;
            if (gz < g.or1 && gx < g.cols) {
                float *out = dst + (size_t)gz * g.pitch + gx;
                if (aligned && gx + 2 <= g.cols) {
                    *reinterpret_cast<float2 *>(out) = make_float2(o[0], o[1]);
                    asm volatile("; 8-byte store" ::: "memory");
                } else {
                    out[0] = o[0];
                    if (gx + 1 < g.cols) out[1] = o[1];
                }
            }
        }
    } else {
        const int cg = tid & 31, rg = tid >> 5;
        float v[4 + 2 * O][4];
#pragma unroll
        for (int i = 0; i < 4 + 2 * O; i++) {
            float4 t = s_a[((rg * 4 + i) * WD_AP + WD_XH) / 4 + cg];
            v[i][0] = t.x; v[i][1] = t.y; v[i][2] = t.z; v[i][3] = t.w;
        }
        const int gx = x0 + 4 * cg;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float total = v[j + 2 * O][e] * taps.kz[0];
#pragma unroll
                for (int kk = 1; kk < KS; kk++) total = tap_acc<FAST>(total, v[j + 2 * O - kk][e], taps.kz[kk]);
                o[e] = UNIT ? total : total * taps.factor;
            }
            int gz = z0 + rg * 4 + j;
            if (gz < g.or1 && gx < g.cols) {
                float *out = dst + (size_t)gz * g.pitch + gx;
                if (aligned && gx + 4 <= g.cols) {
                    *reinterpret_cast<float4 *>(out) = make_float4(o[0], o[1], o[2], o[3]);
                    asm volatile("; 16-byte store" ::: "memory");
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (gx + e < g.cols) out[e] = o[e];
                }
            }
        }
    }
}

template <int O, int NT>
int32_t launch_wide_nt(hipStream_t s, const float *src, float *dst, const nz_geom &g, const nz_kernel_taps &k) {
    constexpr int WD_H = NT / 8;
    long long blocks = (long long)((g.cols + WD_W - 1) / WD_W) * ((g.or1 - g.or0 + WD_H - 1) / WD_H);
    int aligned = (g.pitch % 4 == 0) && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    const bool fast = nz_tls_float_mode >= NZ_FLOAT_FAST;
#define NZ_WD(U, F) NZ_LAUNCH((conv_wide_kernel<O, U, NT, F>), dim3((unsigned)blocks, g.count), dim3(NT), 0, s, src, dst, g, k, aligned)
    if (k.factor == 1.0f) {
        if (fast) NZ_WD(true, true); else NZ_WD(true, false);
    } else {
        if (fast) NZ_WD(false, true); else NZ_WD(false, false);
    }
#undef NZ_WD
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

// 32-row tiles (256 threads) below 17 taps, 64-row tiles (512 threads) from there on: the taller tile stages
// fewer halo rows per output row, which pays once the halo (kernelSize - 1 rows) is a large part of the tile
template <int O>
int32_t launch_wide(hipStream_t s, const float *src, float *dst, const nz_geom &g, const nz_kernel_taps &k) {
    if constexpr (2 * O + 1 >= 17) return launch_wide_nt<O, 512>(s, src, dst, g, k);
    else return launch_wide_nt<O, 256>(s, src, dst, g, k);
}

#ifndef NZ_CONV_NT
#define NZ_CONV_NT 256
#endif
#ifndef NZ_CONV_NT_WIDE
#define NZ_CONV_NT_WIDE 512  // threads per workgroup for the 5-, 7- and 9-tap kernels: 128-row tiles (rows = NT / 4)
#endif

// Threads per workgroup = rows of the register tile / 4 at eight rows per thread.  The 5-, 7- and 9-tap kernels use 128-row
// tiles (512 threads) where throughput counts; a SMALL grid -- a tile of the reference's own sizes (256 ... 1024^2), a stripe's
// ghost-row window -- cannot fill the chip whatever the tile, and what it waits for is the latency of one workgroup's dependent
// applications: 64-row tiles hold half the cells per CU, and FOUR rows per thread (512 threads, two waves per SIMD: a lone
// wave issues a VALU instruction every ~6.6 cycles, two between them every ~4) instead of eight (256 threads, rounds 4 - 5)
// take another fifth off an application: Gauss5 x17 at 512^2 42.0 -> 33.4 us, 1024^2 53.0 -> 45.5, 2048^2 95.9 -> 87.1.
// (Round 4, eight rows per thread, launches one after the other: Gauss5 x17 one tile at a time, 128-row tiles chained -> 64-row
// tiles in four launches: 256^2 80 -> 52 us, 512^2 81 -> 52, 1024^2 89 -> 54, 1536^2 98 -> 74, 2048^2 110 -> 97, 2560^2 126 ->
// 117; 2816^2 127 against 131 and 3072^2 134 against 151.  Round 5, four rows per thread, chained from two launches on
// (nz_stages.cpp): 2048^2 68, 2560^2 95, 3072^2 133 against 124 with the big tiles, 4096^2 221 against 188: from 7 M cells on the
// big tiles stay.)
#ifndef NZ_CONV_SMALL_NT
#define NZ_CONV_SMALL_NT 512  // the 64-row tile of a small grid: NZ_CONV_SMALL_NT threads x NZ_CONV_SMALL_RB rows each
#define NZ_CONV_SMALL_RB 4
#endif
static_assert(NZ_CONV_SMALL_NT / 32 * NZ_CONV_SMALL_RB == 64, "the hosts size a small grid's tiles as 64 rows");
// A TINY grid -- a tile of 256^2 ... 768^2 cells, at most a workgroup per CU -- goes one step further with the 5-tap kernel
// (whose window needs two rows of a neighbouring block at most): 1024 threads x 2 rows, four waves per SIMD.  Gauss5 x17,
// one 512^2 tile at a time: 256 x 8 rows 42.0 us, 512 x 4 rows 33.4 us, 1024 x 2 rows 30.9 us; at 2048^2 (several workgroups per
// CU) 95.9 / 87.1 / 116 us.
constexpr long long NZ_CONV_TINY_CELLS = 600 * 1024;
// (At 1024^2 the 1024-thread shape is faster by itself, 42.8 against 45.4 us, and slower in a tile's pipeline, 9 720 against 9 930
// tiles/s: a 1024-thread workgroup cannot start beside the tail of the stage before.)
static inline bool conv_tiny_grid(const nz_geom &g) { return (long long)g.cols * (g.or1 - g.or0) * g.count <= NZ_CONV_TINY_CELLS; }
#ifndef NZ_CONV_SMALL_CELLS
#define NZ_CONV_SMALL_CELLS (7 * 1024 * 1024)
#endif
static inline bool conv_small_grid(int ksize, const nz_geom &g) {
    if (ksize < 5 || NZ_CONV_NT_WIDE <= 256) return false;
    static const int env = getenv("NZ_CONV_SMALL") ? atoi(getenv("NZ_CONV_SMALL")) : 1;  // 0: never; 2: always (test matrix)
    if (env == 0) return false;
    if (env == 2) return true;
    return (long long)g.cols * (g.or1 - g.or0) * g.count < (long long)NZ_CONV_SMALL_CELLS;
}

template <int KS, int NT, int RBT = RB>
int32_t launch_fused_nt(hipStream_t s, const float *src, float *dst, const nz_geom &g, const nz_kernel_taps &k, int T) {
    constexpr int O = (KS - 1) / 2, RTH = NT / 32 * RBT;  // register tile: RTH rows x 128 columns
    int H = T * O, HX = conv_hx(H);
    int OW = TW - 2 * HX, OH = RTH - 2 * H;
    long long blocks = (long long)((g.cols + OW - 1) / OW) * ((g.or1 - g.or0 + OH - 1) / OH);
    int aligned = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | (uintptr_t)(g.pitch * 4)) & 15) == 0;
    const bool fast = nz_tls_float_mode >= NZ_FLOAT_FAST;
#define NZ_CR(U, F) NZ_LAUNCH((conv_reg_kernel<KS, U, NT, F, RBT>), dim3((unsigned)blocks, g.count), dim3(NT), 0, s, src, dst, g, k, T, aligned)
    if (k.factor == 1.0f) {
        if (fast) NZ_CR(true, true); else NZ_CR(true, false);
    } else {
        if (fast) NZ_CR(false, true); else NZ_CR(false, false);
    }
#undef NZ_CR
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

template <int KS>
int32_t launch_fused(hipStream_t s, const float *src, float *dst, const nz_geom &g, const nz_kernel_taps &k, int T) {
    if (KS >= 5 && conv_small_grid(KS, g)) {
        if constexpr (KS == 5)
            if (conv_tiny_grid(g)) return launch_fused_nt<KS, 1024, 2>(s, src, dst, g, k, T);
        return launch_fused_nt<KS, NZ_CONV_SMALL_NT, NZ_CONV_SMALL_RB>(s, src, dst, g, k, T);
    }
    return launch_fused_nt<KS, (KS >= 5 ? NZ_CONV_NT_WIDE : NZ_CONV_NT)>(s, src, dst, g, k, T);
}

int g_chain_delay_item = -1, g_chain_delay_sleeps = 0;  // nz_debug_chain_delay
int g_chain_spin_limit = 1 << 21;                         // nz_debug_chain_poll_limit

template <int KS, int NT, int RBT = RB>
int32_t launch_chain_nt(hipStream_t s, float *plane0, float *plane1, const nz_geom &g, const nz_kernel_taps &k, const int *Ts,
                        int L, int *flags, unsigned epoch, unsigned *err_host, unsigned *err_epoch) {
    constexpr int O = (KS - 1) / 2;
    constexpr int RTH = NT / 32 * RBT;
    nz_chain ch{};
    ch.L = L;
    ch.first[0] = 0;
    for (int l = 0; l < L; l++) {
        int H = Ts[l] * O, HX = conv_hx(H);
        int OW = TW - 2 * HX, OH = RTH - 2 * H;
        ch.T[l] = Ts[l];
        ch.tiles_x[l] = (g.cols + OW - 1) / OW;
        ch.first[l + 1] = ch.first[l] + ch.tiles_x[l] * ((g.or1 - g.or0 + OH - 1) / OH);
    }
    ch.total = ch.first[L];
    ch.epoch = epoch;
    ch.flags = flags;
    ch.plane[0] = plane0;
    ch.plane[1] = plane1;
    ch.delay_item = g_chain_delay_item;
    ch.delay_sleeps = g_chain_delay_sleeps;
    ch.spin_limit = g_chain_spin_limit;
    ch.err_host = err_host;
    ch.err_epoch = err_epoch;
    int aligned = ((reinterpret_cast<uintptr_t>(plane0) | reinterpret_cast<uintptr_t>(plane1) | (uintptr_t)(g.pitch * 4)) & 15) == 0;
    const bool fast = nz_tls_float_mode >= NZ_FLOAT_FAST;
#define NZ_CC(U, F) NZ_LAUNCH((conv_chain_kernel<KS, U, NT, F, RBT>), dim3((unsigned)ch.total), dim3(NT), 0, s, g, k, ch, aligned)
    if (k.factor == 1.0f) {
        if (fast) NZ_CC(true, true); else NZ_CC(true, false);
    } else {
        if (fast) NZ_CC(false, true); else NZ_CC(false, false);
    }
#undef NZ_CC
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

template <int KS>
int32_t launch_chain(hipStream_t s, float *plane0, float *plane1, const nz_geom &g, const nz_kernel_taps &k, const int *Ts,
                     int L, int *flags, unsigned epoch, unsigned *err_host, unsigned *err_epoch) {
    if (KS >= 5 && conv_small_grid(KS, g)) {
        if constexpr (KS == 5)
            if (conv_tiny_grid(g)) return launch_chain_nt<KS, 1024, 2>(s, plane0, plane1, g, k, Ts, L, flags, epoch, err_host, err_epoch);
        return launch_chain_nt<KS, NZ_CONV_SMALL_NT, NZ_CONV_SMALL_RB>(s, plane0, plane1, g, k, Ts, L, flags, epoch, err_host, err_epoch);
    }
    return launch_chain_nt<KS, (KS >= 5 ? NZ_CONV_NT_WIDE : NZ_CONV_NT)>(s, plane0, plane1, g, k, Ts, L, flags, epoch, err_host, err_epoch);
}

}  // namespace

// a grid the 5-, 7- and 9-tap kernels serve with 64-row tiles (see conv_small_grid)
bool nz_conv_small_grid(int ksize, const nz_geom &g) { return conv_small_grid(ksize, g); }
// ... and so small that a launch is at most a workgroup per CU (the 5-tap kernel then fuses nine applications, nz_stages.cpp)
bool nz_conv_tiny_grid(int ksize, const nz_geom &g) { return conv_small_grid(ksize, g) && conv_tiny_grid(g); }

// work items (= flags) the chained form of L launches needs on this geometry
int nz_conv_chain_items(int ksize, const nz_geom &g, const int *Ts, int L) {
    const int O = (ksize - 1) / 2;
    const int RTH = (ksize >= 5 ? (conv_small_grid(ksize, g) ? 256 : NZ_CONV_NT_WIDE) : NZ_CONV_NT) / 32 * RB;
    int n = 0;
    for (int l = 0; l < L; l++) {
        int H = Ts[l] * O, HX = conv_hx(H);
        int OW = TW - 2 * HX, OH = RTH - 2 * H;
        n += ((g.cols + OW - 1) / OW) * ((g.or1 - g.or0 + OH - 1) / OH);
    }
    return n;
}

// L <= 8 fused launches as ONE grid with tile-level dependencies (conv_chain_kernel): launch l reads plane (l & 1) and
// writes the other one, so the result is in plane0 when L is even.  One grid of the geometry (g.count == 1), planes
// below 4 GiB.  flags: nz_conv_chain_items() ints, never cleared (they carry an epoch).
int32_t nz_launch_conv_chain(hipStream_t s, float *plane0, float *plane1, const nz_geom &g, const nz_kernel_taps &k,
                             const int *Ts, int L, int *flags, unsigned epoch, unsigned *err_host, unsigned *err_epoch) {
    if (L < 1 || L > NZ_CHAIN_MAXL || g.count != 1 || (size_t)g.rows * g.pitch * 4 >= ((size_t)1 << 32)) {
        nz_set_error("conv_chain: %d launches / %d grids / plane of %zu bytes unsupported", L, g.count, (size_t)g.rows * g.pitch * 4);
        return NZ_ERR_INVALID;
    }
    for (int l = 0; l < L; l++)
        if (Ts[l] < 1 || Ts[l] > nz_conv_max_fused(k.ksize)) {
            nz_set_error("conv_chain: T=%d unsupported for kernelSize %d", Ts[l], k.ksize);
            return NZ_ERR_INVALID;
        }
    if (g.or1 <= g.or0) return NZ_OK;
    switch (k.ksize) {
        case 3: return launch_chain<3>(s, plane0, plane1, g, k, Ts, L, flags, epoch, err_host, err_epoch);
        case 5: return launch_chain<5>(s, plane0, plane1, g, k, Ts, L, flags, epoch, err_host, err_epoch);
        case 7: return launch_chain<7>(s, plane0, plane1, g, k, Ts, L, flags, epoch, err_host, err_epoch);
        case 9: return launch_chain<9>(s, plane0, plane1, g, k, Ts, L, flags, epoch, err_host, err_epoch);
    }
    return NZ_ERR_INVALID;
}

// largest T with a useful interior left (OH >= 28, OW >= 88); 0 = this size has no fused kernel
int nz_conv_max_fused(int ksize) {
    switch (ksize) {
        case 3: return 8;
        case 5: return 9;
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
    if (nz_conv_stream_wanted(g, k.ksize, T)) return nz_launch_conv_stream(s, src, dst, g, k, T);
    switch (k.ksize) {
        case 3: return launch_fused<3>(s, src, dst, g, k, T);
        case 5: return launch_fused<5>(s, src, dst, g, k, T);
        case 7: return launch_fused<7>(s, src, dst, g, k, T);
        case 9: return launch_fused<9>(s, src, dst, g, k, T);
    }
    return NZ_ERR_INVALID;
}

// one application (X pass + Z pass) of an odd kernel of 11..25 taps, src -> dst
bool nz_conv_has_wide(int ksize) { return (ksize & 1) && ksize >= 11 && ksize <= 25; }

int32_t nz_launch_conv_wide(hipStream_t s, const float *src, float *dst, const nz_geom &g, const nz_kernel_taps &k) {
    if (g.or1 <= g.or0) return NZ_OK;
    switch (k.ksize) {
        case 11: return launch_wide<5>(s, src, dst, g, k);
        case 13: return launch_wide<6>(s, src, dst, g, k);
        case 15: return launch_wide<7>(s, src, dst, g, k);
        case 17: return launch_wide<8>(s, src, dst, g, k);
        case 19: return launch_wide<9>(s, src, dst, g, k);
        case 21: return launch_wide<10>(s, src, dst, g, k);
        case 23: return launch_wide<11>(s, src, dst, g, k);
        case 25: return launch_wide<12>(s, src, dst, g, k);
    }
    nz_set_error("conv_wide: kernelSize %d unsupported", k.ksize);
    return NZ_ERR_INVALID;
}

int32_t nz_launch_conv_pass_x(hipStream_t s, const float *src, float *dst, const nz_geom &g,
                              const nz_kernel_taps &k) {
    if (g.or1 <= g.or0) return NZ_OK;
    dim3 grid((g.cols + CT - 1) / CT, g.or1 - g.or0);
    if (nz_tls_float_mode >= NZ_FLOAT_FAST) hipLaunchKernelGGL(conv_pass_x_kernel<true>, grid, dim3(CT), 0, s, src, dst, g, k);
    else hipLaunchKernelGGL(conv_pass_x_kernel<false>, grid, dim3(CT), 0, s, src, dst, g, k);
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

int32_t nz_launch_conv_pass_z(hipStream_t s, const float *src, float *dst, const nz_geom &g,
                              const nz_kernel_taps &k) {
    if (g.or1 <= g.or0) return NZ_OK;
    dim3 grid((g.cols + CT - 1) / CT, g.or1 - g.or0);
    if (nz_tls_float_mode >= NZ_FLOAT_FAST) hipLaunchKernelGGL(conv_pass_z_kernel<true>, grid, dim3(CT), 0, s, src, dst, g, k);
    else hipLaunchKernelGGL(conv_pass_z_kernel<false>, grid, dim3(CT), 0, s, src, dst, g, k);
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

int nz_erosion_max_fused() {
    static const int cap = getenv("NZ_EROSION_EMAX") ? atoi(getenv("NZ_EROSION_EMAX")) : 8;
    return cap < 1 ? 1 : (cap > 8 ? 8 : cap);
}

int32_t nz_launch_erosion_fused(hipStream_t s, const float *src, float *dst, const nz_geom &g, int E) {
    if (E < 1 || E > nz_erosion_max_fused()) {
        nz_set_error("erosion_fused: E=%d unsupported", E);
        return NZ_ERR_INVALID;
    }
    if (g.or1 <= g.or0) return NZ_OK;
    int OW = TW - (E > 4 ? 8 : 4), OH = TH - E;
    long long blocks = (long long)((g.cols + OW - 1) / OW) * ((g.or1 - g.or0 + OH - 1) / OH);
    int aligned = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | (uintptr_t)(g.pitch * 4)) & 15) == 0;
    switch (E) {
        case 1: NZ_LAUNCH((erosion_reg_kernel<1>), dim3((unsigned)blocks, g.count), dim3(CT), 0, s, src, dst, g, aligned); break;
        case 2: NZ_LAUNCH((erosion_reg_kernel<2>), dim3((unsigned)blocks, g.count), dim3(CT), 0, s, src, dst, g, aligned); break;
        case 3: NZ_LAUNCH((erosion_reg_kernel<3>), dim3((unsigned)blocks, g.count), dim3(CT), 0, s, src, dst, g, aligned); break;
        case 4: NZ_LAUNCH((erosion_reg_kernel<4>), dim3((unsigned)blocks, g.count), dim3(CT), 0, s, src, dst, g, aligned); break;
        case 5: NZ_LAUNCH((erosion_reg_kernel<5>), dim3((unsigned)blocks, g.count), dim3(CT), 0, s, src, dst, g, aligned); break;
        case 6: NZ_LAUNCH((erosion_reg_kernel<6>), dim3((unsigned)blocks, g.count), dim3(CT), 0, s, src, dst, g, aligned); break;
        case 7: NZ_LAUNCH((erosion_reg_kernel<7>), dim3((unsigned)blocks, g.count), dim3(CT), 0, s, src, dst, g, aligned); break;
        default: NZ_LAUNCH((erosion_reg_kernel<8>), dim3((unsigned)blocks, g.count), dim3(CT), 0, s, src, dst, g, aligned); break;
    }
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

#ifdef NZ_CONV_PROBE
extern "C" int32_t nz_debug_set_conv_probe(void *buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(nz_probe_buf), &buf, sizeof buf) == hipSuccess ? 0 : -3;
}
#endif

// Test hook: the workgroup that claims work item `item` of every chained launch from now on (items count through the
// launches of a chain: launch 0's tiles first) sleeps `sleeps` x ~3.4 us before it loads its tile; item < 0 switches it off.
extern "C" int32_t nz_debug_chain_delay(int32_t item, int32_t sleeps) {
    g_chain_delay_item = item;
    g_chain_delay_sleeps = sleeps < 0 ? 0 : sleeps;
    return NZ_OK;
}

// Test hook: polls (~2 us each) after which a tile of a chained launch stops waiting for a producer; <= 0: the default
// (2^21, seconds)
extern "C" int32_t nz_debug_chain_poll_limit(int32_t polls) {
    g_chain_spin_limit = polls > 0 ? polls : 1 << 21;
    return NZ_OK;
}
