// nz_flow_stream.hip -- the whole FlowMapStage (fill, `iterations` x (outflow, water update), velocity, normalise;
// Geologic/Stage/FlowMapStage.cs:124-195) as ONE row-streaming launch (gfx950).  Built with the max-ILP scheduling
// strategy (Makefile): a wave here is one long dependent stream and only three share a SIMD.
#include <cstdlib>

#include "nz_internal.hpp"
#include "nz_flow_common.hpp"

namespace {

// the same with 0 for the lane that has no neighbour and no zeroed destination to pay for (streaming kernel)
__device__ __forceinline__ float wave_prev0(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_next0(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
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
#ifndef NZ_FS_WPE
#define NZ_FS_WPE 2  // waves per SIMD the four- and five-iteration kernels are register-allocated for: 176 VGPRs, nothing spilled (3: 168 VGPRs and a 25-dword spill; 0.147 against 0.149 ms at 4096^2, round 4)
#endif
constexpr int FS_RING = 16;   // rows of height kept per wave (needs 2n - 1 <= 9)
// NC = columns per lane: 2 (a 128-column strip per wave, 12 registers of state per column and iteration = 120 at n = 5:
// three waves per SIMD) or 1 (64-column strips: 1.22x the halo columns, half the state per lane -- five waves per SIMD)

// a row of the strip: NC cells per lane
template <int NC>
struct alignas(4 * NC) fs_row {  // aligned like float2: one 8-byte LDS / global access for two columns
    float c[NC];
};

template <int NST, int NC>
struct fs_state {
    // stage i (0-based) is about to compute the outflow of row r = t - 2i from the state of iteration i - 1:
    float Tm[NST][NC], T0[NST][NC];   // total height (water + height) of rows r - 1, r after iteration i - 1
    float Wm[NST][NC], W0[NST][NC];   // water of rows r - 1, r after iteration i - 1
    float FA[NST][NC][4], FB[NST][NC][4];  // {W, E, S, N} outflow of rows r - 2, r - 1 of iteration i
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
template <int NST, int NC, int NACT, bool COND, bool XEDGE, bool VEC, bool FAST>
__device__ __forceinline__ void fs_step(fs_state<NST, NC> &st, const int t, const fs_row<NC> hp, const fs_row<NC> hn,
                                        fs_row<NC> *ring, const fs_bounds &b, const int gx, const bool lane_x0,
                                        const bool lane_x1, const bool lane_x1o, const nz_geom &g, const float nmin,
                                        const float nrange, const float inv_range, float *__restrict__ dst,
                                        const bool store_lane) {
    float Tp[NC], Wp[NC], FCprev[NC][4];
#pragma unroll
    for (int e = 0; e < NC; e++) {
        Tp[e] = 0.0001f + hp.c[e];  // fillStage (FlowMapStage.cs:129): water 1e-4 everywhere
        Wp[e] = 0.0001f;
#pragma unroll
        for (int k = 0; k < 4; k++) FCprev[e][k] = 0.0f;
    }
#pragma unroll
    for (int i = 0; i < NST; i++) {
        if (i > NACT) continue;
        if (i == NACT) {  // not computing yet: its windows follow what stage NACT - 1 hands on
#pragma unroll
            for (int e = 0; e < NC; e++) {
                st.Tm[i][e] = st.T0[i][e]; st.T0[i][e] = Tp[e];
                st.Wm[i][e] = st.W0[i][e]; st.W0[i][e] = Wp[e];
            }
            continue;
        }
        const int r = t - 2 * i;
        float FC[NC][4];
        // height of the row whose water this stage updates (read from the ring ahead of the outflow arithmetic)
        fs_row<NC> hh;
#pragma unroll
        for (int e = 0; e < NC; e++) hh.c[e] = 0.0f;
        if (i < NST - 1) hh = ring[((r - 1) & (FS_RING - 1)) * 64];
        // The prefetched row h(t + 2) goes into the ring here, before the last stage's stores: waiting for that load behind
        // a (conditional) store would mean waiting for the store as well -- vmcnt counts both, in order.
        if (i == NACT - 1) ring[((t + 2) & (FS_RING - 1)) * 64] = hn;
        // ---- outflow of row r (ComputeFlowStep)
        const bool fa = !COND || (r >= b.loF[i] && r < b.hiF[i]);
        if (fa) {
            float left = wave_prev0(st.T0[i][NC - 1]), right = wave_next0(st.T0[i][0]);
            if (XEDGE) {
                left = lane_x0 ? st.T0[i][0] : left;
                right = lane_x1 ? st.T0[i][NC - 1] : right;
            }
#pragma unroll
            for (int e = 0; e < NC; e++) {
                const float self = st.T0[i][e];
                const float tW = e == 0 ? left : st.T0[i][e > 0 ? e - 1 : 0];
                float tE = e == NC - 1 ? right : st.T0[i][e + 1 < NC ? e + 1 : e];
                if (XEDGE && NC == 2 && e == 0 && lane_x1o) tE = self;
                float tS = st.Tm[i][e], tN = Tp[e];
                if (COND) {
                    if (r <= g.zc0) tS = self;
                    if (r >= g.zc1) tN = self;
                }
                flux4 old;
                if (i == 0) old = flux4{0.0f, 0.0f, 0.0f, 0.0f};
                else old = flux4{st.FA[i > 0 ? i - 1 : 0][e][0], st.FA[i > 0 ? i - 1 : 0][e][1],
                                 st.FA[i > 0 ? i - 1 : 0][e][2], st.FA[i > 0 ? i - 1 : 0][e][3]};
                const flux4 f = compute_flow_m<FAST>(self, st.W0[i][e], tW, tE, tS, tN, old, true);
                FC[e][0] = f.w; FC[e][1] = f.e; FC[e][2] = f.s; FC[e][3] = f.n;
            }
        } else {
#pragma unroll
            for (int e = 0; e < NC; e++)
#pragma unroll
                for (int k = 0; k < 4; k++) FC[e][k] = 0.0f;
        }
        // the previous stage's row r (its FA) has now been consumed: its window moves on
        if (i > 0) {
#pragma unroll
            for (int e = 0; e < NC; e++)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    st.FA[i > 0 ? i - 1 : 0][e][k] = st.FB[i > 0 ? i - 1 : 0][e][k];
                    st.FB[i > 0 ? i - 1 : 0][e][k] = FCprev[e][k];
                }
        }
        // ---- row r - 1: water update (UpdateWaterStep), or for the last stage velocity + normalise
        const int rw = r - 1;
        const bool wa = !COND || (rw >= b.loW[i] && rw < b.hiW[i]);
        float Wn[NC], Tn[NC];
#pragma unroll
        for (int e = 0; e < NC; e++) Wn[e] = Tn[e] = 0.0f;
        if (wa) {
            // own row rw = FB, row rw - 1 = FA (its fN flows in), row rw + 1 = FC (its fS flows in)
            float eW = wave_prev0(st.FB[i][NC - 1][1]), wE = wave_next0(st.FB[i][0][0]);
            if (XEDGE) {
                eW = lane_x0 ? st.FB[i][0][1] : eW;
                wE = lane_x1 ? st.FB[i][NC - 1][0] : wE;
            }
            float nS[NC], sN[NC];  // fN of row rw - 1, fS of row rw + 1
#pragma unroll
            for (int e = 0; e < NC; e++) {
                nS[e] = st.FA[i][e][3];
                sN[e] = FC[e][2];
                if (COND) {
                    if (rw <= g.zc0) nS[e] = st.FB[i][e][3];
                    if (rw >= g.zc1) sN[e] = st.FB[i][e][2];
                }
            }
            if (i < NST - 1) {
#pragma unroll
                for (int e = 0; e < NC; e++) {
                    const float inE = e == 0 ? eW : st.FB[i][e > 0 ? e - 1 : 0][1];
                    float inW = e == NC - 1 ? wE : st.FB[i][e + 1 < NC ? e + 1 : e][0];
                    if (XEDGE && NC == 2 && e == 0 && lane_x1o) inW = st.FB[i][0][0];
                    Wn[e] = update_water_m<FAST>(st.Wm[i][e], flux4{st.FB[i][e][0], st.FB[i][e][1], st.FB[i][e][2], st.FB[i][e][3]},
                                         inE, inW, nS[e], sN[e]);
                    Tn[e] = Wn[e] + hh.c[e];
                }
            } else {
                // CreateVelocityField + NormalizeMap, FlowMapComponents.cs:120-139,157-165
                float out[NC];
#pragma unroll
                for (int e = 0; e < NC; e++) {
                    const float fE_w = e == 0 ? eW : st.FB[i][e > 0 ? e - 1 : 0][1];
                    float fW_e = e == NC - 1 ? wE : st.FB[i][e + 1 < NC ? e + 1 : e][0];
                    if (XEDGE && NC == 2 && e == 0 && lane_x1o) fW_e = st.FB[i][0][0];
                    const float dl = fE_w - st.FB[i][e][0];
                    const float dr = st.FB[i][e][1] - fW_e;
                    const float dt = sN[e] - st.FB[i][e][3];
                    const float db = st.FB[i][e][2] - nS[e];
                    out[e] = velocity_norm_m<FAST>(dl, dr, dt, db, nmin, nrange, inv_range);
                }
                if (store_lane && (COND || rw >= b.loW[NST - 1])) {
                    float *p = dst + (size_t)rw * g.pitch + gx;
                    if (VEC && NC == 2) {
                        *reinterpret_cast<float2 *>(p) = make_float2(out[0], out[NC - 1]);
                    } else if (VEC) {
                        p[0] = out[0];
                    } else {
#pragma unroll
                        for (int e = 0; e < NC; e++)
                            if (gx + e >= 0 && gx + e < g.cols) p[e] = out[e];
                    }
                }
            }
        }
        // the windows of stage i move on; what it produced feeds stage i + 1 (row r - 2) in this same step
#pragma unroll
        for (int e = 0; e < NC; e++) {
            st.Tm[i][e] = st.T0[i][e]; st.T0[i][e] = Tp[e];
            st.Wm[i][e] = st.W0[i][e]; st.W0[i][e] = Wp[e];
            Tp[e] = Tn[e]; Wp[e] = Wn[e];
#pragma unroll
            for (int k = 0; k < 4; k++) FCprev[e][k] = FC[e][k];
        }
    }
#pragma unroll
    for (int e = 0; e < NC; e++)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            st.FA[NACT - 1][e][k] = st.FB[NACT - 1][e][k];
            st.FB[NACT - 1][e][k] = FCprev[e][k];
        }
}

template <int NC, bool VEC>
__device__ __forceinline__ fs_row<NC> fs_load_row(const float *__restrict__ h, const nz_geom &g, int row, int gx) {
    fs_row<NC> o;
    if (VEC && NC == 2) {
        const float2 t = *reinterpret_cast<const float2 *>(h + (size_t)row * g.pitch + gx);
        o.c[0] = t.x;
        o.c[NC - 1] = t.y;
        return o;
    }
    if (VEC) {
        o.c[0] = h[(size_t)row * g.pitch + gx];
        return o;
    }
    const size_t base = (size_t)row * g.pitch;
#pragma unroll
    for (int e = 0; e < NC; e++) o.c[e] = h[base + clampi(gx + e, 0, g.cols - 1)];
    return o;
}

template <int NST, int NC, bool XEDGE, bool FAST>
__device__ __forceinline__ void flow_stream_body(fs_row<NC> *ring, const float *__restrict__ h, float *__restrict__ dst,
                                                 const nz_geom &g, const int lx0, const int s0, const int s1,
                                                 const float nmin, const float nrange) {
    constexpr bool VEC = !XEDGE;
    const float inv_range = FAST ? 1.0f / nrange : 0.0f;
    constexpr int H = 2 * NST, FS_TW = 64 * NC;
    const int lane = threadIdx.x;
    const int gx = lx0 + NC * lane;
    // the grid's last column is the last of its lane's columns -- or, with two columns per lane and an odd row length,
    // the first of them
    const bool lane_x0 = gx == 0, lane_x1 = gx + NC - 1 == g.cols - 1, lane_x1o = NC == 2 && gx == g.cols - 1;
    const bool store_lane = NC * lane >= H && NC * lane < FS_TW - H;
    fs_bounds b;
#pragma unroll
    for (int i = 0; i < NST; i++) {
        const int m = 2 * (NST - 1 - i) + 1;
        b.loF[i] = max(g.zc0, s0 - m);
        b.hiF[i] = min(g.zc1 + 1, s1 + m);
        b.loW[i] = max(g.zc0, s0 - m + 1);
        b.hiW[i] = min(g.zc1 + 1, s1 + m - 1);
    }
    fs_state<NST, NC> st;
#pragma unroll
    for (int i = 0; i < NST; i++)
#pragma unroll
        for (int e = 0; e < NC; e++) {
            st.Tm[i][e] = 0.0f; st.T0[i][e] = 0.0f; st.Wm[i][e] = 0.0001f; st.W0[i][e] = 0.0001f;
#pragma unroll
            for (int k = 0; k < 4; k++) { st.FA[i][e][k] = 0.0f; st.FB[i][e][k] = 0.0f; }
        }
    // rows read: [t0 - 1, t1], all clamped into the grid's rows (a clamped row is only ever a cell's z-neighbour beyond
    // the grid, which the COND steps replace)
    const int t0 = b.loF[0], t1 = s1 + H - 1;
    {
        const fs_row<NC> hm = fs_load_row<NC, VEC>(h, g, max(t0 - 1, g.zc0), gx), h0 = fs_load_row<NC, VEC>(h, g, t0, gx);
        ring[(t0 & (FS_RING - 1)) * 64] = h0;
#pragma unroll
        for (int e = 0; e < NC; e++) {
            st.Tm[0][e] = 0.0001f + hm.c[e];
            st.T0[0][e] = 0.0001f + h0.c[e];
        }
    }
    fs_row<NC> hp = fs_load_row<NC, VEC>(h, g, min(t0 + 1, g.zc1), gx);
    ring[((t0 + 1) & (FS_RING - 1)) * 64] = hp;
    int t = t0;
#define NZ_FS_STEP(NA, C, T, HP, HN)                                                                              \
    do {                                                                                                          \
        fs_step<NST, NC, NA, C, XEDGE, VEC, FAST>(st, T, HP, HN, ring, b, gx, lane_x0, lane_x1, lane_x1o, g, nmin, nrange, \
                                              inv_range, dst, store_lane);                                        \
    } while (0)
#define NZ_FS_PHASE(K)                                                            \
    if (NST > K) {                                                                \
        for (int q = 0; q < 4 && t < t1; q++, t++) {                              \
            const fs_row<NC> hn = fs_load_row<NC, VEC>(h, g, min(t + 2, g.zc1), gx);  \
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
            const fs_row<NC> hn = fs_load_row<NC, VEC>(h, g, min(t + 2, g.zc1), gx);
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
        const fs_row<NC> hn = fs_load_row<NC, VEC>(h, g, t + 2, gx);
        NZ_FS_STEP(NST, false, t, hp, hn);
        const fs_row<NC> hn2 = fs_load_row<NC, VEC>(h, g, min(t + 3, g.zc1), gx);
        NZ_FS_STEP(NST, false, t + 1, hn, hn2);
        const fs_row<NC> hn3 = fs_load_row<NC, VEC>(h, g, min(t + 4, g.zc1), gx);
        NZ_FS_STEP(NST, false, t + 2, hn2, hn3);
        hp = hn3;
    }
    // the last one or two steps of an inner segment; the drain of a segment that ends on the grid's last row
    for (; t < t1; t++) {
        const fs_row<NC> hn = fs_load_row<NC, VEC>(h, g, min(t + 2, g.zc1), gx);
        NZ_FS_STEP(NST, true, t, hp, hn);
        hp = hn;
    }
#undef NZ_FS_PHASE
#undef NZ_FS_STEP
}

template <int NST, int NC, bool FAST>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NST >= 4 ? NZ_FS_WPE : 4)))
void flow_stream_kernel(const float *__restrict__ h, float *__restrict__ dst, nz_geom g, int S, int nstrips, int Se,
                        int nseg_edge, float nmin, float nrange, int aligned) {
    __shared__ fs_row<NC> s_ring[FS_RING * 64];
    constexpr int FS_TW = 64 * NC, H = 2 * NST, OW = FS_TW - 2 * H;
    // The first blocks are the two border strips (one when the grid is a single strip wide), whose steps carry the border
    // selects and clamped accesses.  A launch on an idle chip places its blocks breadth first (block b is the (b / number
    // of SIMDs)-th wave of its SIMD, tools/probe_flow_stream.py) and a SIMD serves its waves oldest first: as the first
    // wave of their SIMDs the border strips run at full speed and do not finish last (0.158 -> 0.148 ms at 4096^2); they
    // also get shorter segments (Se rows).  Then the strips away from the grid's first / last column, S rows per segment.
    int strip, s0, s1;
    const int ne = min(nstrips, 2), n_edge = ne * nseg_edge;
    if ((int)blockIdx.x < n_edge) {
        const int e = (int)blockIdx.x;
        strip = (e % ne) ? nstrips - 1 : 0;
        s0 = g.or0 + (e / ne) * Se;
        s1 = min(s0 + Se, g.or1);
    } else {
        const int b = (int)blockIdx.x - n_edge;
        strip = 1 + b % (nstrips - 2);
        s0 = g.or0 + (b / (nstrips - 2)) * S;
        s1 = min(s0 + S, g.or1);
    }
    const int lx0 = strip * OW - H;
    const size_t off = blockIdx.y * g.bstride;  // batched launch: one independent grid per blockIdx.y
    const bool inner = aligned && lx0 > 0 && lx0 + FS_TW < g.cols;
    fs_row<NC> *ring = s_ring + threadIdx.x;
    NZ_FPROBE(0, __builtin_amdgcn_s_memrealtime());
    NZ_FPROBE(1, __builtin_amdgcn_s_memtime());
    NZ_FPROBE(6, (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)));   // HW_ID
    NZ_FPROBE(7, (unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) | ((unsigned long long)inner << 32));  // XCC_ID
    if (inner) flow_stream_body<NST, NC, false, FAST>(ring, h + off, dst + off, g, lx0, s0, s1, nmin, nrange);
    else flow_stream_body<NST, NC, true, FAST>(ring, h + off, dst + off, g, lx0, s0, s1, nmin, nrange);
    NZ_FPROBE(4, __builtin_amdgcn_s_memrealtime());
    NZ_FPROBE(5, __builtin_amdgcn_s_memtime());
}

}  // namespace

// The whole stage (first && last) in its row-streaming form.  A segment of S rows per wave: S is chosen so that the grid
// is about one round of waves for the chip (one round of resident waves; longer segments waste less on the pipeline
// fill, but a second, partial round of waves would cost more), never below 16 rows.
bool nz_flow_stream_wanted(const nz_geom &g, int n) {
    static const int mode = getenv("NZ_FLOW_STREAM") ? atoi(getenv("NZ_FLOW_STREAM")) : 1;
    if (mode == 0 || n < 1 || n > FT_MAX_N) return false;
    if (mode == 2) return true;  // test matrix: every size
    // 4096^2: 0.160 against 0.205 ms for the tile kernel, 8192^2: 0.667 against 0.848; 2048^2: 0.084 against 0.075
    return (long long)g.cols * (g.or1 - g.or0) * g.count >= 8ll * 1024 * 1024;
}

int32_t nz_launch_flow_stream(hipStream_t s, const float *h, float *dst, const nz_geom &g, int n, float nmin,
                                     float nrange) {
    // two columns per lane: a 128-column strip per wave (one column per lane -- 64-column strips, five waves per SIMD -- was built
    // in round 4 and lost: 0.174 against 0.148 ms, 1.22x the halo columns; removed in round 5)
    constexpr int NC = 2;
    const int waves = 1024 * NZ_FS_WPE;  // one round of resident waves
    const int H = 2 * n, OW = 64 * NC - 2 * H;
    const int nstrips = (g.cols + OW - 1) / OW, rows = g.or1 - g.or0;
    const long long per = (long long)nstrips * g.count;
    int nseg = (int)(waves / per > 0 ? waves / per : 1);
    int S = (rows + nseg - 1) / nseg;
    if (S < 16) S = 16;
    nseg = (rows + S - 1) / S;
    // border strips: ~12 % shorter segments (their steps are that much longer)
    int Se = (S * 7 + 7) / 8;
    if (Se < 8) Se = S;
    const int nseg_e = (rows + Se - 1) / Se;
    const int nblocks = (nstrips > 2 ? nstrips - 2 : 0) * nseg + (nstrips < 2 ? nstrips : 2) * nseg_e;
    uintptr_t bits = reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(dst) | (uintptr_t)(g.pitch * 4) |
                     (uintptr_t)(g.bstride * 4);
    const int aligned = (bits & 7) == 0;
    const dim3 grid((unsigned)nblocks, g.count);
    const bool fast = nz_tls_float_mode >= NZ_FLOAT_RELAXED;
#define NZ_FSL(N, F) NZ_LAUNCH((flow_stream_kernel<N, 2, F>), grid, dim3(64), 0, s, h, dst, g, S, nstrips, Se, nseg_e, nmin, nrange, aligned)
#define NZ_FS(N)                     \
    do {                             \
        if (fast) NZ_FSL(N, true);   \
        else NZ_FSL(N, false);       \
    } while (0)
    switch (n) {
        case 1: NZ_FS(1); break;
        case 2: NZ_FS(2); break;
        case 3: NZ_FS(3); break;
        case 4: NZ_FS(4); break;
        default: NZ_FS(5); break;
    }
#undef NZ_FS
#undef NZ_FSL
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

#ifdef NZ_FLOW_PROBE
extern "C" int32_t nz_debug_set_flow_probe(void *buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(nz_flow_probe_buf), &buf, sizeof buf) == hipSuccess ? 0 : -3;
}
#endif
