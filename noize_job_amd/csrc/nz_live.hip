// nz_live.hip -- the particle half of noize-job's live erosion (BASELINE config 4), gfx950.
//
// Replaces the Burst jobs of Geologic/ParticleErosion/MultiThreadErosionJob.cs that LiveErosion.TriggerQueuedBeyerMT
// (Component/LiveErosion.cs:378-436) chains per cycle:
//   FillBeyerQueueJob (:21-72)               -> fill_queue_kernel          one lane per 16 particles of a worker
//   QueuedBeyerCycleMultiThreadJob (:178-224) -> descent_kernel             one lane per particle, until it is dead
//   ProcessBeyerErosiveEventsJob (:330-385)   -> process_events_kernel      one lane per cell that received an event
//   ErodeHeightMaps (:438-480, ONE thread)    -> disperse_kernel (every cell gathers) + pile_kernel (one wave per block)
//   PoolAutomataJob, drainParticles (:264-327) -> nz_elementwise.hip, pool_automata_pass_kernel<true>
//   CurvitureMapJob (:387-436), SetRGBA32Job (:482-529) -> curviture_kernel, set_rgba32_kernel
//
// The reference is deterministic per particle but not per run (UnityEngine.Random seeds, a parallel multi-hash-map
// summed in its iteration order, sediment events in the order parallel jobs enqueued them).  Three choices are FIXED
// here, the same three oracle/noize_oracle_live.c fixes, so that the two agree bit for bit:
//   1. the seed is an argument; positions follow Unity.Mathematics.Random (xorshift32);
//   2. per-cell event sums are 2^-40 fixed point (64-bit integer atomics): independent of arrival order;
//   3. ErodeHeightMaps applies the per-cell events in the order one worker would have produced them (job z ascending,
//      x ascending inside): first every KernelDisperse event -- as a GATHER: a target cell folds the <= 25
//      contributions that reach it in that order, reading only its own running value, so all targets run in parallel
//      -- then every PileSolver event: blocks of side 2 * (PILING_RADIUS + 1), 4-coloured, colour after colour, the
//      canonical order inside a block (piles of one colour's blocks cannot see each other; see pile_kernel).
// atan / sin of the velocity model are the Cephes fp32 polynomials written out operation by operation (no libm call),
// the same text as the oracle's.  Planes are indexed x * res + z (WorldTile.getIdx, LiveErosionDataTypes.cs:608-610).
#include <cstring>

#include "nz_internal.hpp"
#include "nz_flow_track.hpp"

struct nz_particle_queue {  // device layout: header + particles
    int32_t *hdr = nullptr;  // {count, capacity, overflow, pad}
    nz_particle *data = nullptr;
    int32_t capacity = 0;
};

struct nz_erosive_events {
    int32_t res = 0;
    unsigned long long *acc = nullptr;  // [3][res^2]: pool, track, sediment sums (2^-40 fixed point, two's complement)
    int32_t *touched = nullptr;         // [res^2] events per cell this cycle
    int32_t *list[2] = {nullptr, nullptr};  // cells touched this cycle / last cycle
    int32_t *counters = nullptr;        // {n_list[0], n_list[1], (unused), events}
    float *sediment = nullptr;          // [res^2] the per-cell ErosiveEvent.deltaSediment (0 where none)
    int cur = 0;
    void *pile_scratch = nullptr;       // the ManhattanVertex offsets of the current PILING_RADIUS
    size_t pile_scratch_bytes = 0;
    // [nb][nb] which PileSolver blocks hold a pile this cycle (1; 2 once the ticket kernel has done it), then PILE_CTL
    // control words {busy blocks of colour 0..3, ticket}, then the busy blocks listed per colour ((nb + 1)^2 / 4 each)
    int32_t *pile_blocks = nullptr;
    size_t pile_blocks_n = 0;
    float *height_snapshot = nullptr;   // [res^2], safe mode only (nz_ctx_set_pile_safe): the plane as ErodeHeightMaps found it
};

namespace {

constexpr int CT = 256;
constexpr double FIX_SCALE = 1099511627776.0;  // 2^40

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ float lmaxf(float a, float b) { return (b != b) || a > b ? a : b; }  // math.max
__device__ __forceinline__ float lminf(float a, float b) { return (b != b) || a < b ? a : b; }  // math.min

// ---- FillBeyerQueueJob + FlowMaster.CreateRandomParticles ----------------------------------------------------------
__device__ __forceinline__ uint32_t rnd_next_state(uint32_t &s) {  // Unity.Mathematics.Random.NextState
    uint32_t t = s;
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    return t;
}

// xorshift32 is linear over GF(2): the state 32 * q steps on is a 32 x 32 bit matrix applied to the state.  FILL_JUMP
// holds the matrices of 32 * 2^l steps (columns = images of the unit vectors), so a lane reaches the first state of
// any 16-particle chunk of its worker's stream in at most FILL_LEVELS products instead of walking there.
constexpr int FILL_CHUNK = 16, FILL_LEVELS = 27;
struct fill_jump_table { uint32_t col[FILL_LEVELS][32]; };
constexpr uint32_t fill_step(uint32_t s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
constexpr uint32_t fill_apply(const uint32_t (&m)[32], uint32_t s) {
    uint32_t r = 0;
    for (int b = 0; b < 32; b++) r ^= (0u - ((s >> b) & 1u)) & m[b];
    return r;
}
constexpr fill_jump_table fill_make_jump() {
    fill_jump_table t{};
    uint32_t m[32] = {}, n[32] = {};
    for (int b = 0; b < 32; b++) m[b] = fill_step(1u << b);
    for (int sq = 0; sq < 5; sq++) {  // 1 step -> 2 * FILL_CHUNK = 32 steps
        for (int b = 0; b < 32; b++) n[b] = fill_apply(m, m[b]);
        for (int b = 0; b < 32; b++) m[b] = n[b];
    }
    for (int l = 0; l < FILL_LEVELS; l++) {
        for (int b = 0; b < 32; b++) t.col[l][b] = m[b];
        for (int b = 0; b < 32; b++) n[b] = fill_apply(m, m[b]);
        for (int b = 0; b < 32; b++) m[b] = n[b];
    }
    return t;
}
__constant__ fill_jump_table FILL_JUMP = fill_make_jump();

// One workgroup; an item is one chunk of FILL_CHUNK consecutive particles of one reference worker (worker i draws
// Random(seed + i), two states per particle, and numbers its particles pid += i * COUNT + k).
__global__ __launch_bounds__(1024) void fill_queue_kernel(int32_t *hdr, nz_particle *data, int generationRound, int res,
                                                          int maxParticles, int seed, int concurrency) {
    __shared__ int s_current, s_count;
    if (threadIdx.x == 0) {
        int current = hdr[0];
        int required = maxParticles - current;
        if (required < 1) required = 1;
        int COUNT = required / concurrency;
        if (COUNT < 1) COUNT = 1;
        if (current + concurrency * COUNT > hdr[1]) {  // the reference's queue grows; this one reports
            hdr[2] = 1;
            COUNT = 0;
        }
        s_current = current;
        s_count = COUNT;
    }
    __syncthreads();
    const int COUNT = s_count, current = s_current;
    const int chunks = (COUNT + FILL_CHUNK - 1) / FILL_CHUNK;
    const uint16_t pid0 = (uint16_t)(generationRound * maxParticles);
    for (long long item = threadIdx.x; item < (long long)concurrency * chunks; item += blockDim.x) {
        const int i = (int)(item / chunks), q = (int)(item - (long long)i * chunks);
        uint32_t st = (uint32_t)(seed + i);
        (void)rnd_next_state(st);  // Random(uint seed): state = seed; NextState()
        for (int l = 0; l < FILL_LEVELS; l++)
            if ((q >> l) & 1) {
                uint32_t r = 0;
#pragma unroll
                for (int b = 0; b < 32; b++) r ^= (0u - ((st >> b) & 1u)) & FILL_JUMP.col[l][b];
                st = r;
            }
        const int k1 = min(COUNT, (q + 1) * FILL_CHUNK);
        for (int k = q * FILL_CHUNK; k < k1; k++) {
            // pid after the additions of particles 0..k: (k + 1) * i * COUNT + k (k + 1) / 2, modulo 2^16
            unsigned long long tri = (unsigned long long)k * (unsigned long long)(k + 1) / 2ull;
            uint32_t lin = (uint32_t)(k + 1) * (uint32_t)i * (uint32_t)COUNT;
            nz_particle p;
            p.px = (int)(((uint64_t)rnd_next_state(st) * (uint64_t)(uint32_t)res) >> 32);
            p.pz = (int)(((uint64_t)rnd_next_state(st) * (uint64_t)(uint32_t)res) >> 32);
            p.water = 1.0f;
            p.pid = (uint16_t)(pid0 + (uint16_t)lin + (uint16_t)tri);
            data[current + (long long)i * COUNT + k] = p;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) hdr[0] = current + concurrency * COUNT;
}

// ---- Heading (LiveErosionDataTypes.cs:1297-1444) ---------------------------------------------------------------------
enum { H_N = 1, H_S = 2, H_E = 4, H_W = 8, H_NONE = 0 };
__device__ __forceinline__ int heading_from(float dx, float dz) {
    int b = 0;
    if (dx > 0.0f) b |= H_E; else if (dx < 0.0f) b |= H_W;
    if (dz > 0.0f) b |= H_N; else if (dz < 0.0f) b |= H_S;
    return b;
}
// nb[] / WTORDER index of a heading (N, E, S, W, NE, SE, SW, NW), -1 for NONE:
//   N = 1 -> 0, E = 4 -> 1, S = 2 -> 2, W = 8 -> 3, NE = 5 -> 4, SE = 6 -> 5, SW = 10 -> 6, NW = 9 -> 7
// a nibble (value + 1) per heading; a switch is a chain of branches on the particle's critical path
__device__ __forceinline__ int heading_wt_idx(int h) {
    const unsigned t = (h & 8) ? 0x784u : 0x06520310u;
    return (unsigned)h < 16u ? (int)((t >> (4 * (h & 7))) & 15u) - 1 : -1;
}
// position in ADJACENT (N, NE, E, SE, S, SW, W, NW) and back; 8 for anything else:
//   1 -> 0, 5 -> 1, 4 -> 2, 6 -> 3, 2 -> 4, 10 -> 5, 8 -> 6, 9 -> 7
__device__ __forceinline__ int heading_adj_idx(int h) {
    const unsigned t = (h & 8) ? 0x88888576u : 0x83128408u;
    return (unsigned)h < 16u ? (int)((t >> (4 * (h & 7))) & 15u) : 8;
}
__device__ __forceinline__ int adj_heading(int i) {
    // {1, 5, 4, 6, 2, 10, 8, 9}[i & 7], a nibble each: a table in memory is a load on the particle's critical path
    return (int)((0x98A26451u >> (4 * (i & 7))) & 15u);
}

__device__ __forceinline__ long long to_fix(float v) {
    if (!(fabsf(v) < 4194304.0f)) return 0;  // NaN, inf and anything beyond 2^22 is dropped
    return __double2ll_rn((double)v * FIX_SCALE);
}
__device__ __forceinline__ float from_fix(unsigned long long a) { return (float)((double)(long long)a * (1.0 / FIX_SCALE)); }

struct live_planes {
    const float *height, *pool, *flow;
    unsigned long long *acc;
    int32_t *touched, *list, *counters;
    int list_slot;
    size_t n;
};

#ifndef NZ_DESCENT_LANES
#define NZ_DESCENT_LANES 64  // particles per wave
#endif
#ifndef NZ_DESCENT_PREFETCH
#define NZ_DESCENT_PREFETCH 1
#endif

#ifdef NZ_DESCENT_PROBE
// -DNZ_DESCENT_PROBE (tools/probe_descent.sh): shader-clock sums over all waves and steps -- waiting for the step's loads,
// the step's arithmetic, the emit -- and the number of wave steps
__device__ unsigned long long g_descent_probe[8];
#define NZ_DPROBE(slot)                                   \
    do {                                                  \
        __builtin_amdgcn_sched_barrier(0);                \
        const unsigned long long t_now = clock64();       \
        probe_t[slot] += t_now - probe_last;              \
        probe_last = t_now;                               \
        __builtin_amdgcn_sched_barrier(0);                \
    } while (0)
#else
#define NZ_DPROBE(slot) do { } while (0)
#endif

__device__ __forceinline__ float ld_off(const float *plane, unsigned byte_off) {
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(plane) + byte_off);
}

struct live_particle {
    float posx, posz, dirx, dirz, vel, water, sediment;
    int age;
};

// ---- one DescendSimultaneous call (LiveErosionDataTypes.cs:273-432) as ONE basic block -----------------------------
// The event a live particle leaves at cell (ix, iz), and whether the particle is dead afterwards; the planes are read
// only.  A wave with its SIMD to itself issues a VALU instruction that depends on the one before every 8 clocks, an independent
// one every ~5 (tools/microbench/lone_wave.hip), and the 64 particles of a wave between them take every branch of the
// step anyway.  So the step is written without branches: every way out becomes a flag, every `if` a select on the
// reference's own condition, the arithmetic of both sides is the reference's operation for operation, and the compiler
// is free to interleave the independent chains (the eight neighbours, the divisions, the fixed-point conversions).
// What the ways out share is computed once: water / HEIGHT and sediment / HEIGHT (the particle's water and sediment do
// not change before the end of a step), and the slope pass (atan, sin, sqrt) that UphillVelocityLoss and the velocity
// update both run on |hDiff| -- a lane needs a second pass only when it cannot climb and falls back to the drain
// direction, which is the one wave-uniform branch left.
// Cephes atanf / sinf in fp32 operations: the oracle's text (oracle/noize_oracle_live.c) with every `if` as a select
// on the same condition
__device__ __forceinline__ float live_atanf(float xx) {
    const bool nan = xx != xx;
    const bool neg = xx < 0.0f;
    const float sign = neg ? -1.0f : 1.0f;
    const float nx = -xx;
    float x = neg ? nx : xx;
    const bool big = x > 2.414213562373095f, mid = !big && x > 0.4142135623730950f;
    const float y0 = big ? 1.5707963267948966192f : (mid ? 0.7853981633974483096f : 0.0f);
    const float xm1 = x - 1.0f, xp1 = x + 1.0f;
    const float num_m = mid ? xm1 : x, den_m = mid ? xp1 : 1.0f;
    const float num = big ? -1.0f : num_m, den = big ? x : den_m;
    x = num / den;  // -(1 / x) == (-1) / x and x / 1 == x, bit for bit
    const float z = x * x;
    float y = y0;
    y += (((8.05374449538e-2f * z - 1.38776856032E-1f) * z + 1.99777106478E-1f) * z - 3.33329491539E-1f) * z * x + x;
    const float r = sign * y;  // both arms of a select are plain values: an expression in an arm is a branch again
    return nan ? xx : r;
}

__device__ __forceinline__ float live_sinf(float xx) {
    const bool nan = xx != xx;
    const bool neg = xx < 0.0f;
    float sign = neg ? -1.0f : 1.0f;
    const float nx = -xx;
    float x = neg ? nx : xx;
    const bool big = x > 8192.0f;
    int j = (int)(1.27323954473516f * x);
    float y = (float)j;
    const bool odd = j & 1;
    j += odd ? 1 : 0;
    const float y1 = y + 1.0f;
    y = odd ? y1 : y;
    j &= 7;
    const bool flip = j > 3;
    const float nsign = -sign;
    sign = flip ? nsign : sign;
    j -= flip ? 4 : 0;
    x = ((x - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
    const float z = x * x;
    float yc = ((2.443315711809948E-005f * z - 1.388731625493765E-003f) * z + 4.166664568298827E-002f) * z * z;
    yc -= 0.5f * z;
    yc += 1.0f;
    float ys = ((-1.9515295891E-4f * z + 8.3321608736E-3f) * z - 1.6666654611E-1f) * z * x;
    ys += x;
    y = (j == 1 || j == 2) ? yc : ys;
    const float r = sign * y;
    const float rb = big ? 0.0f : r;
    return nan ? xx : rb;
}

__device__ __forceinline__ bool descent_step(const live_planes &P, live_particle &p, const nz_erosion_params &ep, int res,
                                             float HEIGHT, float patchRes, int ix, int iz, bool alive, float &eTrack,
                                             float &ePool, float &eSed, float (&pf)[6]
#ifdef NZ_DESCENT_PROBE
                                             , unsigned long long (&probe_t)[4], unsigned long long &probe_last
#endif
) {
    const int NBDX[8] = {0, 1, 0, -1, 1, 1, -1, -1}, NBDZ[8] = {1, 0, -1, 0, 1, -1, -1, 1};
    const int heading0 = heading_from(p.dirx, p.dirz);
    const bool dead_water = p.water < .01f;                    // :276
    const bool dead_age = !dead_water && p.age >= ep.MAXAGE;   // :283
    const float wq = p.water / HEIGHT, sq = p.sediment / HEIGHT;
    const unsigned res4 = 4u * (unsigned)res;
    const unsigned rowb[3] = {(unsigned)clampi(ix - 1, 0, res - 1) * res4, (unsigned)ix * res4,
                              (unsigned)clampi(ix + 1, 0, res - 1) * res4};
    const unsigned zb[3] = {4u * (unsigned)clampi(iz - 1, 0, res - 1), 4u * (unsigned)iz, 4u * (unsigned)clampi(iz + 1, 0, res - 1)};
    const unsigned soff = (unsigned)clampi((int)p.posx, 0, res - 1) * res4 + 4u * (unsigned)clampi((int)p.posz, 0, res - 1);
    float hv[8], pv[8], fv[8];
    const float h0 = ld_off(P.height, soff), p0 = ld_off(P.pool, soff), f0 = ld_off(P.flow, rowb[1] + zb[1]);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const unsigned no = rowb[NBDX[k] + 1] + zb[NBDZ[k] + 1];
        hv[k] = ld_off(P.height, no);
        pv[k] = ld_off(P.pool, no);
        fv[k] = ld_off(P.flow, no);
    }
#if NZ_DESCENT_PREFETCH
    // The next step reads the 3 x 3 cells around one of this step's neighbours: rows ix - 2 and ix + 2 are the only
    // cache lines of it this step has not touched (z is the fast index: iz +- 2 shares the lines of iz, a line end
    // aside).  Asking for them now -- BEHIND this step's own loads, returns come back in order -- hides their way
    // from HBM behind this step's arithmetic; the next step then finds all its lines in the CU's cache.  Nothing looks
    // at the values before the end of the step (descent_kernel).
    __builtin_amdgcn_sched_barrier(0);
    {
        const unsigned xm = (unsigned)clampi(ix - 2, 0, res - 1) * res4 + zb[1], xp = (unsigned)clampi(ix + 2, 0, res - 1) * res4 + zb[1];
        pf[0] = ld_off(P.height, xm); pf[1] = ld_off(P.height, xp);
        pf[2] = ld_off(P.pool, xm);   pf[3] = ld_off(P.pool, xp);
        pf[4] = ld_off(P.flow, xm);   pf[5] = ld_off(P.flow, xp);
    }
    __builtin_amdgcn_sched_barrier(0);
#endif
    const float currentHeight = HEIGHT * (h0 + p0);
    int nb[8];
    int hmin = 0x7fffffff, kmin = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const float all = HEIGHT * (hv[k] + pv[k]) + ep.FLOW_HEIGHT_CONTRIBUTION * (fv[k]);
        nb[k] = (int)(100.0f * all);
        const bool lower = nb[k] < hmin;  // nbSort[0] and its first index (IndexOf)
        hmin = lower ? nb[k] : hmin;
        kmin = lower ? k : kmin;
    }
    const float fl = lmaxf(f0, 0.0f);
    NZ_DPROBE(0);
    const int drainDx0 = (int)((0xA19u >> (2 * kmin)) & 3u) - 1, drainDz0 = (int)((0x8246u >> (2 * kmin)) & 3u) - 1;
    const int heading_drain = heading_from((float)drainDx0, (float)drainDz0);
    const int heading = heading0 == H_NONE ? heading_drain : heading0;
    const float effectiveDrag = ep.DRAG * (1.0f - fl);
    const float effectiveFriction = ep.FRICTION * (1.0f - fl);
    const int ai = heading_adj_idx(heading);
    const int hl = adj_heading(ai + 7), hr = adj_heading(ai + 1);
    const int wl = heading_wt_idx(hl), wc = heading_wt_idx(heading), wr = heading_wt_idx(hr);
    const bool dead_heading = ai >= 8 || wl < 0 || wc < 0 || wr < 0;  // unreachable: the drain direction is never (0, 0)
    int il = 0, ic = 0, ir = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        il = (k == wl) ? nb[k] : il;
        ic = (k == wc) ? nb[k] : ic;
        ir = (k == wr) ? nb[k] : ir;
    }
    const float hx = (float)il / 100.0f, hy = (float)ic / 100.0f, hz = (float)ir / 100.0f;
    const bool take_l = hx < hy && hx < hz, take_r = !take_l && hz < hx && hz < hy;
    const float hh_r = take_r ? hz : hy;
    const int fh_r = take_r ? hr : heading;
    const float headingHeight = take_l ? hx : hh_r;
    const int flowH = take_l ? hl : fh_r;
    const int flowDx = ((flowH >> 2) & 1) ? 1 : (((flowH >> 3) & 1) ? -1 : 0);
    const int flowDz = ((flowH >> 0) & 1) ? 1 : (((flowH >> 1) & 1) ? -1 : 0);
    const float hDiff0 = headingHeight - currentHeight;
    const bool down = hDiff0 < 0.0f;  // up = !down takes NaN along, as the reference's !(hDiff < 0) does
    float vel = p.vel - (p.vel * effectiveDrag);
    // the slope pass on |hDiff0|: UphillVelocityLoss for the lanes going up (+ friction), DownhillVelocityGain for the
    // others (- friction == + (-friction)); hDiff0 == -0 gives the reference's NaN (0 / 0) either way
    const float a1 = fabsf(hDiff0);
    const float th1 = live_atanf(a1 / patchRes);
    const float st1 = live_sinf(th1);
    const float nfric = -effectiveFriction;
    const float fric1 = down ? nfric : effectiveFriction;
    const float acc1 = (ep.GRAVITY * st1) + fric1;
    const float r1 = sqrtf(2.0f * fabsf(acc1) * (a1 / st1));
    const bool uphillOk = !down && r1 <= vel;
    const bool follow = down || uphillOk;  // keeps the heading's direction; the others fall back to the drain
    const float hD = (float)hmin / 100.0f - currentHeight;
    const bool dead_uphill = !follow && hD > 0.0f;  // :330
    float th2 = 0.0f, r2 = 0.0f;
    const float a2 = fabsf(hD);
#ifdef NZ_DESCENT_PROBE
    if (__ballot(alive && !down)) probe_t[3] += 1;
    if (__ballot(alive && !follow && !dead_uphill && a2 > 0.0f)) probe_t[3] += 1ull << 32;
#endif
    if (__ballot(alive && !follow && !dead_uphill && a2 > 0.0f)) {  // wave-uniform: some lane falls back and goes on
        th2 = live_atanf(a2 / patchRes);
        const float st2 = live_sinf(th2);
        const float acc2 = (ep.GRAVITY * st2) - effectiveFriction;
        r2 = sqrtf(2.0f * fabsf(acc2) * (a2 / st2));
    }
    const float hDiff = follow ? hDiff0 : hD;
    const float velocityLoss = (follow && !down) ? r1 : 0.0f;
    const int drainDx = follow ? flowDx : drainDx0, drainDz = follow ? flowDz : drainDz0;
    const float dirx = (float)drainDx, dirz = (float)drainDz;
    const float pnx = p.posx + dirx, pnz = p.posz + dirz;
    const bool dead_edge = (int)pnx < 0 || (int)pnz < 0 || (int)pnx >= res || (int)pnz >= res;  // :344
    const float vDiff = fabsf(hDiff);
    const bool sloped = vDiff > 0.0f;
    const float theta = follow ? th1 : th2;
    const float theta_deg = theta * 180.0f / 3.14159f;
    const float thetaD = sloped ? theta_deg : 0.0f;
    // hDiff > 0: only a lane that climbed (a fallen-back lane with hD > 0 is dead); otherwise the gain of its own pass
    const float loss = -1.0f * velocityLoss, gain = follow ? r1 : r2;
    const float dv = hDiff > 0.0f ? loss : gain;
    const float deltaV = sloped ? dv : 0.0f;
    vel = lmaxf((vel + deltaV), 0.0f);
    const float over = vel - ep.TERMINAL_VELOCITY;
    vel = vel - lmaxf(lminf(over, lmaxf(effectiveDrag * 0.25f * over * over, 0.0f)), 0.0f);
    const bool dead_slow = thetaD < 3.0f && vel < 1.0f;  // :374
    const float currentCapacity = vel * p.water * ep.CAPACITY;
    const float erodes = -1.0f * ep.EROSION * (currentCapacity - p.sediment), lays = ep.DEPOSITION * (p.sediment - currentCapacity);
    const float depositionAmount = p.sediment < currentCapacity ? erodes : lays;
    const bool deposits = fabsf(depositionAmount) > 0.0f;
    const float dq = depositionAmount / HEIGHT;
    // the first way out that applies, in the reference's order
    const bool with_pool = dead_age || (!dead_water && !dead_heading && (dead_uphill || (!dead_edge && dead_slow)));
    const bool with_sed = dead_water || with_pool;
    const bool dies = dead_water || dead_age || dead_heading || dead_uphill || dead_edge || dead_slow;
    eTrack = dies ? 0.0f : p.water;
    ePool = with_pool ? wq : 0.0f;
    const float sed_dead = with_sed ? sq : 0.0f, sed_live = deposits ? dq : 0.0f;
    eSed = dies ? sed_dead : sed_live;
    // a dead particle's state is never looked at again
    const float sed_less = p.sediment - depositionAmount;
    p.sediment = deposits ? sed_less : p.sediment;
    p.water = p.water * (1 - ep.EVAP);
    p.vel = vel;
    p.dirx = dirx;
    p.dirz = dirz;
    p.posx = pnx;
    p.posz = pnz;
    p.age++;
    return dies;
}

// BeyerParticle.DescendSimultaneous repeated until the particle is dead (FlowMaster.BeyerSimultaneousDescentSingle,
// LiveErosionComponents.cs:79-91), one lane per particle.  A particle's steps are one dependent chain -- load the
// neighbourhood, decide, move -- so the kernel lasts as long as the longest-lived particle's chain, and what counts is
// the length of one step:
//   * every step ends in ONE event (a cell, three sums); the wave emits its lanes' events together, dead lanes wait
//     masked until the wave's last particle is dead;
//   * the cell's first event of the cycle (atomicAdd on `touched` returned 0; asked for at the top of the step, so its
//     round trip overlaps the step's loads) puts the cell on the cycle's list.  The lanes of a step share ONE increment
//     of the list counter (160 k single increments of one word serialise in the L2), and its return value is only
//     looked at one step later, behind that step's loads (returns come back in order): the list entry of step s is
//     written at the end of step s + 1;
//   * the look-ahead loads of descent_step, and its branch-free form.
__global__ __launch_bounds__(64) void descent_kernel(live_planes P, const int32_t *hdr, const nz_particle *particles,
                                                    nz_erosion_params ep, int res, float HEIGHT, float patchRes) {
    const int lane = threadIdx.x;
    const int n = min(hdr[0], hdr[1]);  // the count lives on the device; the grid is sized by the host's bound
    int events = 0;
    unsigned sink = 0;
#ifdef NZ_DESCENT_PROBE
    unsigned long long probe_t[4] = {0, 0, 0, 0}, probe_last = clock64(), probe_steps = 0;
#endif
    for (long long first_pi = (long long)blockIdx.x * NZ_DESCENT_LANES; first_pi < n; first_pi += (long long)gridDim.x * NZ_DESCENT_LANES) {
    const long long pi = first_pi + lane;
    bool alive = lane < NZ_DESCENT_LANES && pi < n;
    nz_particle src{0, 0, 0.0f, 0};
    if (alive) src = particles[pi];
    // a particle uploaded with a position outside the tile (nz_particle_queue_upload cannot know the resolution) is
    // dropped: its first event would index the per-cell planes out of bounds
    if ((unsigned)src.px >= (unsigned)res || (unsigned)src.pz >= (unsigned)res) alive = false;
    live_particle p{(float)src.px, (float)src.pz, 0.0f, 0.0f, .01f, src.water, 0.0f, 0};
    // the list entries of the previous step: waiting for their place
    unsigned long long pend_firsts = 0;
    int pend_base = 0, pend_idx = 0;
    // every path of a step either kills or ages the particle, and age >= MAXAGE kills
    while (__ballot(alive)) {
#ifdef NZ_DESCENT_PROBE
        probe_steps++;
#endif
        float eTrack = 0.0f, ePool = 0.0f, eSed = 0.0f;
        float pf[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        bool first = false, dies = false;
        int idx = 0;
        if (alive) {
            events++;
            const int ix = (int)rintf(p.posx), iz = (int)rintf(p.posz);
            idx = ix * res + iz;
            first = atomicAdd(&P.touched[idx], 1) == 0;
#ifdef NZ_DESCENT_PROBE
            dies = descent_step(P, p, ep, res, HEIGHT, patchRes, ix, iz, alive, eTrack, ePool, eSed, pf, probe_t, probe_last);
#else
            dies = descent_step(P, p, ep, res, HEIGHT, patchRes, ix, iz, alive, eTrack, ePool, eSed, pf);
#endif
        }
        NZ_DPROBE(1);
        if (pend_firsts) {  // wave-uniform
            const int base = __builtin_amdgcn_readlane(pend_base, __ffsll((long long)pend_firsts) - 1);
            if ((pend_firsts >> lane) & 1ull) P.list[base + __popcll(pend_firsts & ((1ull << lane) - 1ull))] = pend_idx;
        }
        const unsigned long long firsts = __ballot(alive && first);
        if (firsts) {
            if (lane == __ffsll((long long)firsts) - 1) pend_base = atomicAdd(&P.counters[P.list_slot], __popcll(firsts));
            pend_idx = idx;
        }
        pend_firsts = firsts;
        if (alive) {
            const long long a = to_fix(ePool), b = to_fix(eTrack), c = to_fix(eSed);
            if (a) atomicAdd(&P.acc[idx], (unsigned long long)a);
            if (b) atomicAdd(&P.acc[P.n + idx], (unsigned long long)b);
            if (c) atomicAdd(&P.acc[2 * P.n + idx], (unsigned long long)c);
        }
        alive = alive && !dies;
        NZ_DPROBE(2);
#if NZ_DESCENT_PREFETCH
        __builtin_amdgcn_sched_barrier(0);  // the look-ahead values: looked at last
        sink |= __float_as_uint(pf[0]) & __float_as_uint(pf[1]) & __float_as_uint(pf[2]) & __float_as_uint(pf[3]) &
                __float_as_uint(pf[4]) & __float_as_uint(pf[5]);
#endif
    }
    if (pend_firsts) {
        const int base = __builtin_amdgcn_readlane(pend_base, __ffsll((long long)pend_firsts) - 1);
        if ((pend_firsts >> lane) & 1ull) P.list[base + __popcll(pend_firsts & ((1ull << lane) - 1ull))] = pend_idx;
    }
    }  // the wave's next 64 particles
#ifdef NZ_DESCENT_PROBE
    if (lane == 0 && probe_steps) {
        for (int k = 0; k < 3; k++) atomicAdd(&g_descent_probe[k], probe_t[k]);
        atomicAdd(&g_descent_probe[6], probe_t[3] & 0xffffffffull);
        atomicAdd(&g_descent_probe[7], probe_t[3] >> 32);
        atomicAdd(&g_descent_probe[3], probe_steps);
        atomicAdd(&g_descent_probe[4], 1ull);
        atomicMax(&g_descent_probe[5], probe_steps);
    }
#endif
    for (int o = 32; o; o >>= 1) events += __shfl_xor(events, o);  // every lane of the wave is here
    if (lane == 0 && events) atomicAdd(&P.counters[3], events);
    if (sink == 0xffffffffu && n < 0) P.counters[3] = 0;  // never: n >= 0; keeps the look-ahead loads
}

// ProcessBeyerErosiveEventsJob + CombineBeyerEvents / HandleBeyerEvent (MultiThreadErosionJob.cs:330-385,
// LiveErosionComponents.cs:93-110), over the cells that received an event; first the sediment plane forgets the
// previous cycle's events.
__global__ __launch_bounds__(CT) void forget_kernel(float *sediment, const int32_t *list, const int32_t *counters, int slot) {
    const int n = counters[slot];
    for (int i = blockIdx.x * CT + threadIdx.x; i < n; i += gridDim.x * CT) sediment[list[i]] = 0.0f;
}

#ifndef NZ_EVENTS_BLOCKS
#define NZ_EVENTS_BLOCKS 2048  // a cell's eight scattered lines per lane: many lanes in flight
#endif
__global__ __launch_bounds__(CT) void process_events_kernel(float *pool, float *track, float *sediment,
                                                           unsigned long long *acc, int32_t *touched, const int32_t *list,
                                                           int32_t *counters, int slot, size_t ncell, float poolMul,
                                                           float trackMul) {
    const int n = counters[slot];
    for (int i = blockIdx.x * CT + threadIdx.x; i < n; i += gridDim.x * CT) {
        const int idx = list[i];
        const float poolV = from_fix(acc[idx]), trackV = from_fix(acc[ncell + idx]), sedimentV = from_fix(acc[2 * ncell + idx]);
        if (fabsf(poolV) > 0.0f) {  // Place :181-185
            float last = pool[idx];
            last += poolV * poolMul;
            pool[idx] = last;
        }
        if (fabsf(trackV) > 0.0f) {
            float last = track[idx];
            last += trackV * trackMul;
            track[idx] = last;
        }
        sediment[idx] = sedimentV;
        acc[idx] = 0;
        acc[ncell + idx] = 0;
        acc[2 * ncell + idx] = 0;
        touched[idx] = 0;
    }
}

// FlowMaster.KernelDisperse (LiveErosionComponents.cs:130-157) for every dispersed event at once, as a gather.
__constant__ float KERNEL5[5] = {0.12007838424321349f, 0.23388075658535032f, 0.29208171834287244f, 0.23388075658535032f,
                                 0.12007838424321349f};

// Events are sparse (a cycle's particles touch a few cells in a thousand), so the gather runs over the cycle's event
// list, IN PLACE (a target reads the sediment plane and its own height only):
//   disperse_list_kernel   thread = (listed cell s, tap): the target T = s + tap, if the tap stays two cells inside the
//                          grid, belongs to the FIRST dispersing source of T's window in the canonical order -- that
//                          thread folds the whole window, the others leave; tap 0 of a pile event flags its block
//   disperse_frame_kernel  the two-cell frame of the grid, where clamped taps of one source pile onto one target: every
//                          frame cell gathers in full
__device__ __forceinline__ bool disperses(float val, float pileThreshold) {
    return val != 0.0f && (val < 0.0f || val <= pileThreshold);  // WriteSedimentMap :118-128
}

__global__ __launch_bounds__(CT) void disperse_list_kernel(float *__restrict__ height, const float *__restrict__ sediment,
                                                          const int32_t *__restrict__ list,
                                                          const int32_t *__restrict__ counters, int slot, int res,
                                                          float pileThreshold, int32_t *pile_blocks, int B, int nb) {
    const long long n = (long long)counters[slot] * 25;
    for (long long j = (long long)blockIdx.x * CT + threadIdx.x; j < n; j += (long long)gridDim.x * CT) {
        const int s = list[j / 25], tap = (int)(j % 25);
        const int sx = s / res, sz = s - sx * res;
        const float own = sediment[s];
        if (tap == 0 && pile_blocks && own != 0.0f && !disperses(own, pileThreshold)) pile_blocks[(sx / B) * nb + sz / B] = 1;
        if (!disperses(own, pileThreshold)) continue;
        const int tx = sx + tap / 5 - 2, tz = sz + tap % 5 - 2;
        if (tx < 2 || tz < 2 || tx >= res - 2 || tz >= res - 2) continue;  // the frame kernel's
        // the whole 5 x 5 window in flight at once (walking it load by load, with the early way out, was 25 dependent
        // round trips); q = 5 * (wz - tz + 2) + (wx - tx + 2) is the canonical order inside the window
        float win[25];
        const float *w0 = sediment + (size_t)(tx - 2) * res + (tz - 2);
#pragma unroll
        for (int q = 0; q < 25; q++) win[q] = w0[(size_t)(q % 5) * res + q / 5];
        float v = height[(size_t)tx * res + tz];
        int firstq = 25;
#pragma unroll
        for (int q = 24; q >= 0; q--) firstq = disperses(win[q], pileThreshold) ? q : firstq;
        const bool mine = firstq < 25 && tx - 2 + firstq % 5 == sx && tz - 2 + firstq / 5 == sz;
        if (!mine) continue;
#pragma unroll
        for (int q = 0; q < 25; q++) {
            const float val = win[q];
            const float newDiff = ((val * (KERNEL5[4 - q % 5] * KERNEL5[4 - q / 5])) / 1.0f);  // its one tap here
            const float nextV = v + newDiff;
            if (disperses(val, pileThreshold) && !(nextV > 1.0f) && !(nextV < 0.0f)) v = v + newDiff;
        }
        if (mine) height[(size_t)tx * res + tz] = v;
    }
}

__global__ __launch_bounds__(CT) void disperse_frame_kernel(float *__restrict__ height, const float *__restrict__ sediment,
                                                           int res, float pileThreshold) {
    // frame cell f: rows x = 0, 1, res-2, res-1 in full (z the fast index), then columns z = 0, 1, res-2, res-1 between them
    const int rows = min(res, 4);
    const long long nrow = (long long)rows * res, ncol = (long long)max(res - 4, 0) * 4;
    const long long f = (long long)blockIdx.x * CT + threadIdx.x;
    if (f >= nrow + ncol) return;
    int tx, tz;
    if (f < nrow) {
        const int r = (int)(f / res);
        tx = res <= 4 ? r : (r < 2 ? r : res - 4 + r);
        tz = (int)(f % res);
    } else {
        const long long g = f - nrow;
        const int c = (int)(g & 3);
        tx = 2 + (int)(g >> 2);
        tz = c < 2 ? c : res - 4 + c;
        if (tz < 0 || tz >= res || (c >= 2 && tz < 2)) return;  // res < 4: the rows were everything
    }
    float v = height[(size_t)tx * res + tz];
    const int x0 = max(tx - 2, 0), x1 = min(tx + 2, res - 1), z0 = max(tz - 2, 0), z1 = min(tz + 2, res - 1);
    bool any = false;
    for (int sz = z0; sz <= z1; sz++)
        for (int sx = x0; sx <= x1; sx++) {
            const float val = sediment[(size_t)sx * res + sz];
            if (!disperses(val, pileThreshold)) continue;
            any = true;
            for (int kx = 0; kx < 5; kx++)
                for (int kz = 0; kz < 5; kz++) {
                    if (clampi(sx - 2 + kx, 0, res - 1) != tx || clampi(sz - 2 + kz, 0, res - 1) != tz) continue;
                    const float newDiff = ((val * (KERNEL5[kx] * KERNEL5[kz])) / 1.0f);
                    const float nextV = v + newDiff;
                    if (nextV > 1.0f) continue;
                    if (nextV < 0.0f) continue;
                    v = v + newDiff;
                }
        }
    if (any) height[(size_t)tx * res + tz] = v;
}

// PileSolver (LiveErosionDataTypes.cs:1053-1225).  A pile reads and raises heights within Chebyshev distance
// PILING_RADIUS + 1 of its cell, so piles in different blocks of side B = 2 * (PILING_RADIUS + 1) with one block between
// them cannot see each other.  The grid is cut into such blocks, 4-coloured by (bx & 1, bz & 1); one launch per colour,
// one wave per block; inside a block the events keep the canonical order (z ascending, x ascending inside).  (The
// reference runs all of them on one thread in whatever order its parallel queue writers produced.)
// Per pile the wave loads the vertex values side by side (SetPile), walks DepositSediment -- sequential by nature: a
// running remainder -- over the vertices that qualify (found 64 at a time), and commits the modified vertices in vertex order (the ManhattanVertex list
// names the centre four times and many ring cells twice, each copy with a value of its own: the LAST copy wins).
// Heights as the ticket kernel reads and writes them: blocks of different colours run side by side on other CUs (other
// XCDs), ordered by flags only -- loads that pass the CU's cache, stores that go through (agent scope: `sc1`).
template <bool COH>
__device__ __forceinline__ float pile_ld(const float *p) {
    if (COH) return __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<const int *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    return *p;
}
template <bool COH>
__device__ __forceinline__ void pile_st(float *p, float v) {
    if (COH) __hip_atomic_store(reinterpret_cast<int *>(p), __builtin_bit_cast(int, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

struct pile_lds {
    float *val;           // per vertex: value
    int *idx;             // cell
    unsigned char *flag;  // bit 0 valid, bit 1 modified
    float *sed;           // the block's sediment events, [B][B]
    short2 *ofs;          // the vertex offsets: every pile walks them
};
__device__ __forceinline__ pile_lds pile_carve(unsigned char *s_raw, int nverts, int B) {
    pile_lds L;
    L.val = reinterpret_cast<float *>(s_raw);
    L.idx = reinterpret_cast<int *>(s_raw + (size_t)nverts * 4);
    L.flag = s_raw + (size_t)nverts * 8;
    L.sed = reinterpret_cast<float *>(s_raw + (((size_t)nverts * 9 + 15) & ~(size_t)15));
    L.ofs = reinterpret_cast<short2 *>(L.sed + B * B);
    return L;
}
__device__ __forceinline__ void pile_load_offsets(const pile_lds &L, const short2 *__restrict__ ofs, int nverts, int lane) {
    for (int i0 = lane; i0 < nverts; i0 += 64 * 16) {  // sixteen loads in flight per lane: a loop of single loads waits for each
        short2 t[16];
#pragma unroll
        for (int u = 0; u < 16; u++) t[u] = ofs[min(i0 + 64 * u, nverts - 1)];
#pragma unroll
        for (int u = 0; u < 16; u++)
            if (i0 + 64 * u < nverts) L.ofs[i0 + 64 * u] = t[u];
    }
}

// The piles of block (bx, bz), one wave; the vertex offsets are in LDS already.
template <bool COH>
__device__ __forceinline__ void pile_block(const pile_lds &L, float *height, const float *__restrict__ sediment, int nverts,
                                           int res, int maxDistance, int B, int bx, int bz, float pileThreshold,
                                           float increment, int32_t *beat = nullptr) {
    int beats = 0;  // (ticket launch) piles started: the block's flag carries 1 + 2 * beats while it is busy, see pile_ticket_wave
    float *s_val = L.val;
    int *s_idx = L.idx;
    unsigned char *s_flag = L.flag;
    float *s_sed = L.sed;
    short2 *s_ofs = L.ofs;
    const int lane = threadIdx.x;
    const int x0 = bx * B, z0 = bz * B, x1 = min(x0 + B, res), z1 = min(z0 + B, res);
    // the block's sediment events into LDS first, lanes along z (the planes' fast index), all loads in flight together:
    // walking the block row by row straight from memory cost one dependent round trip per row (32 of them)
    const int bw = x1 - x0, bh = z1 - z0;  // <= B each
    for (int i0 = lane; i0 < bw * bh; i0 += 64 * 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {  // eight loads in flight per lane (no load behind a condition: each would be waited for)
            const int i = min(i0 + 64 * u, bw * bh - 1), xo = i / bh, zo = i - xo * bh;
            t[u] = sediment[(size_t)(x0 + xo) * res + z0 + zo];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = i0 + 64 * u, xo = i / bh, zo = i - xo * bh;
            if (i < bw * bh) s_sed[xo * B + zo] = t[u];
        }
    }
    __syncthreads();
    int px = 0, pz = 0;
    // vertices [v0, v1) of the pile at (px, pz): value, cell and validity side by side in LDS, eight per lane in flight
    auto set_pile = [&](int v0, int v1) {
        for (int i0 = v0 + lane; i0 < v1; i0 += 64 * 8) {
            short2 o[8];
            float hv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) o[u] = s_ofs[min(i0 + 64 * u, v1 - 1)];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int vx = px + o[u].x, vz = pz + o[u].y;
                const bool ok = i0 + 64 * u < v1 && vx >= 0 && vz >= 0 && vx < res && vz < res;
                hv[u] = pile_ld<COH>(height + (ok ? (size_t)vx * res + vz : 0));  // cell 0 for a vertex off the grid: never looked at
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = i0 + 64 * u;
                if (i >= v1) break;
                const int vx = px + o[u].x, vz = pz + o[u].y;
                const bool ok = vx >= 0 && vz >= 0 && vx < res && vz < res;
                s_flag[i] = ok ? 1 : 0;
                s_idx[i] = ok ? vx * res + vz : 0;
                s_val[i] = ok ? hv[u] : 0.0f;
            }
        }
    };
    for (int z = z0; z < z1; z++) {
        for (int xb = x0; xb < x1; xb += 64) {
            const int x = xb + lane;
            const float val = x < x1 ? s_sed[(x - x0) * B + (z - z0)] : 0.0f;
            const bool is_pile = val != 0.0f && !(val < 0.0f || val <= pileThreshold);
            unsigned long long todo = __ballot(is_pile);
            while (todo) {  // wave-uniform
                const int src_lane = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                px = xb + src_lane;
                pz = z;
                if (beat && lane == 0) __hip_atomic_store(beat, 1 + 2 * (++beats & 0x3fffffff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float amount = __shfl(val, src_lane);
                // SetPile, as far as the pile will look: a round r only examines the vertices of the rings dist < r
                // (2 r (r + 3) of them), and most piles are a few increments that the first rounds place -- the first
                // 64 vertices (rounds 1..4) are one load per lane, the other rings follow only if a round asks for them
                int loaded = min(64, nverts);
                if (lane < loaded) {
                    const short2 o = s_ofs[lane];
                    const int vx = px + o.x, vz = pz + o.y;
                    const bool ok = vx >= 0 && vz >= 0 && vx < res && vz < res;
                    const int cell = ok ? vx * res + vz : 0;
                    const float hv = pile_ld<COH>(height + cell);  // cell 0 for a vertex off the grid: never looked at
                    s_flag[lane] = ok ? 1 : 0;
                    s_idx[lane] = cell;
                    s_val[lane] = ok ? hv : 0.0f;
                }
                __syncthreads();  // one wave per workgroup: orders the LDS traffic
                // DepositSediment: a running remainder handed to the vertices below the round's level, in vertex order --
                // sequential by nature, but only over the vertices that QUALIFY.  The wave tests 64 vertices at a time
                // (a vertex's value changes within a round only at its own visit, so the test at the round's start is the
                // test at its visit) and walks the ballot's set bits in order with the reference's arithmetic; every
                // scalar below is wave-uniform.  A pile that used to cost one lane ~8 000 LDS round trips
                // (15 rounds x up to 540 vertices) costs a few ballots per round.
                {
                    float remaining = amount;
                    int cmax = -1;
                    for (int guard = 0; remaining > 0.0f && increment > 0.0f && guard < 4096; guard++) {
                        // (a pile of a million increments is seconds of one wave, ~0.8 ms a pass: a sign of life every 8 passes)
                        if (beat && (guard & 7) == 7 && lane == 0)
                            __hip_atomic_store(beat, 1 + 2 * (++beats & 0x3fffffff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const float amt = remaining;
                        float deposited = 0.0f, rem = amt;
                        bool done = false;
                        for (int round = 1; round <= maxDistance && !done; round++) {
                            const int nv = 2 * round * (round + 3);  // vertices of the rings dist < round: sum of 4 (dist + 2)
                            if (nv > loaded) {  // wave-uniform: the other rings, once
                                set_pile(loaded, nverts);
                                loaded = nverts;
                                __syncthreads();
                            }
                            const float level = s_val[0] + (increment * (float)round);
                            for (int base = 0; base < nv && !done; base += 64) {
                                const int cl = base + lane;
                                const float myv = cl < nv ? s_val[cl] : 0.0f;
                                const bool below = cl < nv && (s_flag[cl] & 1) && myv < level;
                                const unsigned long long mask = __ballot(below);
                                if (!mask) continue;
                                // the running remainder is a chain of float operations, one vertex after the other -- but
                                // a chain of wave-uniform registers only: step k's increment goes to lane k of `incs`, and
                                // the vertices take theirs all at once afterwards (each copy's value is its own)
                                float incs = 0.0f;
                                int served = 0;
                                for (unsigned long long m = mask; m && !done; m &= m - 1) {
                                    const int c = base + __ffsll((long long)m) - 1;
                                    const float inc = lminf(increment, rem);
                                    incs = lane == served ? inc : incs;
                                    served++;
                                    cmax = max(cmax, c);
                                    deposited += inc;
                                    rem = amt - deposited;
                                    if (rem <= 0.0f) done = true;
                                }
                                const int rank = __popcll(mask & ((1ull << lane) - 1ull));
                                const float mine = __shfl(incs, rank);
                                if (below && rank < served) {
                                    s_flag[cl] |= 2;
                                    s_val[cl] = myv + mine;
                                }
                            }
                            __builtin_amdgcn_wave_barrier();  // the LDS stores before the next round's tests
                        }
                        remaining = done ? 0.0f : rem;
                    }
                    // CommitChanges, in vertex order (a cell listed twice keeps its LAST copy): only modified vertices.  The
                    // wave reads 64 vertices' cells and values at once; lane 0 then stores them one after the other from
                    // scalar copies (stores of one lane stay in order, none of them is waited for)
                    for (int base = 0; base <= cmax; base += 64) {
                        const int cl = base + lane;
                        const bool is_mod = cl <= cmax && (s_flag[cl] & 3) == 3;
                        const int mycell = cl <= cmax ? s_idx[cl] : 0;
                        const float myval = cl <= cmax ? s_val[cl] : 0.0f;
                        unsigned long long mod = __ballot(is_mod);
                        while (mod) {
                            const int l = __ffsll((long long)mod) - 1;
                            mod &= mod - 1;
                            const int cell = __builtin_amdgcn_readlane(mycell, l);
                            const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myval), l));
                            if (lane == 0) pile_st<COH>(height + cell, v);
                        }
                    }
                    __threadfence_block();
                }
                __syncthreads();  // the next pile of this block reads the committed heights
            }
        }
    }
}

// One launch per colour, one wave per block of that colour.
__global__ __launch_bounds__(64) void pile_kernel(float *height, const float *__restrict__ sediment,
                                                 const int32_t *__restrict__ pile_blocks,
                                                 const short2 *__restrict__ ofs, int nverts, int res, int maxDistance,
                                                 int B, int nb, int cx, int cz, float pileThreshold, float increment) {
    extern __shared__ unsigned char s_raw[];
    const int per_row = (nb - cx + 1) / 2;
    const int bx = cx + 2 * (int)(blockIdx.x % per_row), bz = cz + 2 * (int)(blockIdx.x / per_row);
    if (!pile_blocks[bx * nb + bz]) return;  // no pile event in this block (disperse_list_kernel flags them)
    const pile_lds L = pile_carve(s_raw, nverts, B);
    pile_load_offsets(L, ofs, nverts, threadIdx.x);
    pile_block<false>(L, height, sediment, nverts, res, maxDistance, B, bx, bz, pileThreshold, increment);
}

// ALL colours in one launch (round 4).  A colour's launch lasts as long as its fullest block (30...40 piles, ~2 us each, at
// 8192^2 / 10 000 droplets: tools/probe_piles.py) while a busy block holds 3...5 on average -- four launches are four such
// tails.  Here the busy blocks (disperse_list_kernel lists them per colour) are work items in colour-major order, handed
// out by an atomic ticket; a block waits only for the busy ones among its eight neighbours that have a LOWER colour --
// the blocks whose piles the canonical order puts before its own -- i.e. for items with a lower ticket, which a resident
// workgroup holds or has finished: progress does not depend on how the hardware dispatches workgroups.  blocks[b]: odd = busy
// (1 when listed, 1 + 2 k once its k-th pile has started: the heartbeat a waiting neighbour's bound watches), 2 = done.  Heights travel between CUs with agent-scope accesses, a block's stores have left (vmcnt 0) before its flag is
// raised.  ctl: {items of colour 0..3, ticket}.  A wait is bounded all the same (err_host, mapped host memory: the context
// reports an internal error at its next synchronisation instead of hanging).
constexpr int PILE_CTL = 8;
int g_pile_spin_limit = 1 << 22;  // nz_debug_pile_poll_limit
// the blocks disperse_list_kernel flagged, listed under their colour (a colour's blocks are independent: any order).  A
// thread per flag; a wave numbers its blocks with a ballot per colour, the workgroup's waves share LDS counters, and ONE
// increment per colour and workgroup reaches memory (7 000 returning atomics on four words took 75 us).
__global__ __launch_bounds__(1024) void pile_list_kernel(const int32_t *__restrict__ blocks, int32_t *ctl, int32_t *list, int cap,
                                                        int nb) {
    __shared__ int s_n[4], s_base[4];
    if (threadIdx.x < 4) s_n[threadIdx.x] = 0;
    __syncthreads();
    const int n = nb * nb, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 1024 + (int)threadIdx.x;
    const bool busy = b < n && blocks[b] != 0;
    const int bx = b / nb, bz = b - bx * nb;
    const int c = (bx & 1) | ((bz & 1) << 1);
    int mine = 0;  // my number among the workgroup's blocks of my colour
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const unsigned long long m = __ballot(busy && c == k);
        if (!m) continue;  // wave-uniform
        int base = 0;
        if (lane == 0) base = atomicAdd(&s_n[k], __popcll(m));
        base = __shfl(base, 0);
        if (busy && c == k) mine = base + __popcll(m & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    if (threadIdx.x < 4) s_base[threadIdx.x] = s_n[threadIdx.x] ? atomicAdd(&ctl[threadIdx.x], s_n[threadIdx.x]) : 0;
    __syncthreads();
    if (busy) list[(size_t)c * cap + s_base[c] + mine] = b;
}

struct pile_ticket_args {
    float *height;
    const float *sediment;
    int32_t *blocks;
    const int32_t *list;
    int32_t *ctl;
    int cap;
    const short2 *ofs;
    int nverts, res, maxDistance, B, nb;
    float pileThreshold, increment;
    unsigned *err_host;
    int spin_limit;  // polls of a neighbour's flag before a block gives up (2^22: seconds; nz_debug_pile_poll_limit)
};

__device__ __forceinline__ void pile_ticket_wave(const pile_ticket_args &a, unsigned char *s_raw) {
    float *height = a.height;
    const float *__restrict__ sediment = a.sediment;
    int32_t *blocks = a.blocks;
    const int32_t *__restrict__ list = a.list;
    int32_t *ctl = a.ctl;
    const short2 *__restrict__ ofs = a.ofs;
    const int cap = a.cap, nverts = a.nverts, res = a.res, maxDistance = a.maxDistance, B = a.B, nb = a.nb;
    const float pileThreshold = a.pileThreshold, increment = a.increment;
    unsigned *err_host = a.err_host;
    const int spin_limit = a.spin_limit;
    const pile_lds L = pile_carve(s_raw, nverts, B);
    const int lane = threadIdx.x;
    const int n0 = ctl[0], n1 = ctl[1], n2 = ctl[2], n3 = ctl[3];
    const int total = n0 + n1 + n2 + n3;
    bool have_ofs = false;
    for (;;) {
        int t = 0;
        if (lane == 0) t = atomicAdd(&ctl[4], 1);
        t = __shfl(t, 0);
        if (t >= total) break;
        if (!have_ofs) {
            pile_load_offsets(L, ofs, nverts, lane);
            have_ofs = true;
        }
        const int c = t < n0 ? 0 : (t < n0 + n1 ? 1 : (t < n0 + n1 + n2 ? 2 : 3));
        const int i = t - (c > 0 ? n0 : 0) - (c > 1 ? n1 : 0) - (c > 2 ? n2 : 0);
        const int b = list[(size_t)c * cap + i];
        const int bx = b / nb, bz = b - bx * nb;
        if (c > 0 && lane < 8) {
            const int k = lane < 4 ? lane : lane + 1;  // the eight neighbours of the 3 x 3
            const int qx = bx + k % 3 - 1, qz = bz + k / 3 - 1;
            const int qc = (qx & 1) | ((qz & 1) << 1);
            if (qx >= 0 && qz >= 0 && qx < nb && qz < nb && qc < c) {
                const int32_t *f = blocks + (size_t)qx * nb + qz;
                // busy = an ODD flag: 1 when listed, then 1 + 2 k as the block starts its k-th pile (a heartbeat: a block of a
                // plane buried in sediment works for seconds -- 5 000 droplets on 89^2 cells, increments of a 60 000th of the
                // sediment: round 6's soak, seed 707 case 7457 -- and the bound below is on polls WITHOUT progress, not on time)
                // A block that WAITS passes the sign of life on: whenever the flag it watches moves, it moves its own, so that a block
                // waiting for a block that waits for a busy block does not run out of patience either.
                int spins = 0, last = 1, passed = 0;
                for (;;) {
                    const int v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (!(v & 1)) break;
                    if (v != last) {
                        last = v;
                        spins = 0;
                        __hip_atomic_store(blocks + b, 1 + 2 * ((++passed << 3 | lane) & 0x3fffffff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    __builtin_amdgcn_s_sleep(16);
                    if (++spins > spin_limit) {  // seconds without a sign of life: never, unless the protocol is broken
                        __hip_atomic_store(err_host + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // word [1]: the pile solver's
                        break;
                    }
                }
            }
        }
        __syncthreads();
        pile_block<true>(L, height, sediment, nverts, res, maxDistance, B, bx, bz, pileThreshold, increment, blocks + b);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the block's stores have left
        __syncthreads();
        if (lane == 0) __hip_atomic_store(blocks + b, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ __launch_bounds__(64) void pile_ticket_kernel(pile_ticket_args a) {
    extern __shared__ unsigned char s_raw[];
    pile_ticket_wave(a, s_raw);
}

// ErodeHeightMaps and UpdateFlowFromTrackJob are siblings in the reference's job graph (CombineDependencies,
// Component/LiveErosion.cs:408-412): the pile solver touches height and the sediment events, the flow update pool, flow and
// track.  The pile solver's launch is a couple of thousand waves that wait for memory and for each other (VALU issue 0.07,
// HBM 0.04 of the chip); this form of it carries the flow update's workgroups BEHIND its own -- workgroups start in index
// order, so every pile wave is resident before the first flow workgroup and no pile wave ever waits for a slot -- and the
// flow update streams through the CUs the pile waves leave idle.  A flow workgroup: one wave, FLOW_UNROLL quads per lane,
// all in flight at once (the launch's LDS -- the pile solver's, ~11 KB per workgroup -- caps a CU at 14 workgroups: few
// waves, so each keeps more loads in flight than flow_from_track_kernel's do).  8192^2, cycle 100: the two jobs 0.294 + 0.192 ms
// one after the other, 0.366-0.377 ms as one launch -- the pile waves' dependent loads take longer next to a streaming
// kernel (alone 0.183 ms), so it is not max(0.29, 0.17); 2 ... 8 quads in flight and 1 ... 16 rounds per workgroup all land
// within 0.366-0.387.
constexpr int FLOW_UNROLL = 4;
constexpr size_t FLOW_WG_CELLS = (size_t)64 * 4 * FLOW_UNROLL;
struct flow_track_args {
    float *pool, *flow, *track;
    size_t n;
    float flowLossRate, evaporation;
    int aligned;
};

__global__ __launch_bounds__(64) void pile_ticket_flow_kernel(pile_ticket_args a, unsigned pile_grid, flow_track_args f) {
    extern __shared__ unsigned char s_raw[];
    if (blockIdx.x < pile_grid) {
        pile_ticket_wave(a, s_raw);
        return;
    }
    const size_t wg0 = (size_t)(blockIdx.x - pile_grid) * FLOW_WG_CELLS;
    const size_t base = wg0 + (size_t)threadIdx.x * 4;
    if (f.aligned && wg0 + FLOW_WG_CELLS <= f.n) {  // wave-uniform: the workgroup's cells are all inside
        float4 pv[FLOW_UNROLL], tv[FLOW_UNROLL], po[FLOW_UNROLL];
#pragma unroll
        for (int u = 0; u < FLOW_UNROLL; u++) {
            const size_t i = base + (size_t)u * 256;
            pv[u] = *reinterpret_cast<const float4 *>(f.flow + i);
            tv[u] = *reinterpret_cast<const float4 *>(f.track + i);
            po[u] = *reinterpret_cast<const float4 *>(f.pool + i);
        }
#pragma unroll
        for (int u = 0; u < FLOW_UNROLL; u++)
            flow_from_track_quad(f.pool, f.flow, f.track, base + (size_t)u * 256, pv[u], tv[u], po[u], f.flowLossRate, f.evaporation);
    } else {
        for (int u = 0; u < FLOW_UNROLL; u++) {
            const size_t i = base + (size_t)u * 256;
            if (i < f.n) flow_from_track_cells(f.pool, f.flow, f.track, i, f.n < i + 4 ? f.n : i + 4, f.flowLossRate, f.evaporation);
        }
    }
}

// WorldTile.Curviture (LiveErosionDataTypes.cs:726-866) + CurvitureMapJob (MultiThreadErosionJob.cs:387-436)
__global__ __launch_bounds__(CT) void curviture_kernel(unsigned char *texture, int channel, const float *__restrict__ height,
                                                      int res, int meshRes, int offset, float HEIGHT, float w) {
    const int x = blockIdx.x * CT + threadIdx.x, z = blockIdx.y;
    if (x >= meshRes) return;
    const int cx = z + offset, cz = x + offset;  // pos = (srcZ, x + offset)
    auto HH = [&](int dx, int dz) {
        return height[(size_t)clampi(cx + dx, 0, res - 1) * res + clampi(cz + dz, 0, res - 1)] * HEIGHT;
    };
    const float w2 = w * w;
    const float z1x = HH(-1, 1), z1y = HH(0, 1), z1z = HH(1, 1), z1w = HH(-1, 0);
    const float z5 = height[(size_t)cx * res + cz] * HEIGHT;
    const float z6x = HH(1, 0), z6y = HH(-1, -1), z6z = HH(0, -1), z6w = HH(1, -1);
    const float zx = (z1z + z6x + z6w - z1x - z1w - z6y) / (6.0f * w);
    const float zy = (z1x + z1y + z1z - z6y - z6z - z6w) / (6.0f * w);
    const float zxx = (z1x + z1z + z1w + z6x + z6y + z6w - 2.0f * (z1y + z5 + z6z)) / (3.0f * w2);
    const float zyy = (z1x + z1y + z1z + z6y + z6z + z6w - 2.0f + (z1w + z5 + z6x)) / (3.0f * w2);  // `- 2.0f +` as written
    const float zxy = (z1z + z6y - z1x - z6w) / (4.0f * w2);
    const float dzx = -zx, dzy = -zy, dxx = -zxx, dyy = -zyy, dxy = -zxy;
    const float zx2 = dzx * dzx, zy2 = dzy * dzy, p = zx2 + zy2;
    const float nn = zy2 * dxx - 2.0f * dxy * dzx * dzy + zx2 * dyy;
    const float d = p * powf(p + 1.0f, 0.5f);
    float v = fabsf(d) < 1e-18f ? 0.0f : nn / d;
    v = fabsf(v);
    const float sign_ = v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f);
    const float log_ = logf(1.0f + powf(10.0f, .05f) * fabsf(v));
    const float r = fabsf(sign_ * log_) / 2.0f;
    texture[((size_t)z * meshRes + x) * 4 + channel] = (unsigned char)(lmaxf(0.0f, lminf(1.0f, r)) * 255.0f);
}

__global__ __launch_bounds__(CT) void set_rgba32_kernel(unsigned char *texture, int channel, const float *__restrict__ src,
                                                       int dataRes, int meshRes, int offset, float scale) {
    const int x = blockIdx.x * CT + threadIdx.x, z = blockIdx.y;
    if (x >= meshRes) return;
    const float c = lmaxf(0.0f, lminf(1.0f, src[(size_t)(z + offset) * dataRes + x + offset] * scale));
    texture[((size_t)z * meshRes + x) * 4 + channel] = (unsigned char)(c * 255.0f);
}

}  // namespace

#define NZ_BEGIN(ctx, dep)                   \
    do {                                     \
        int32_t rc_ = nz_ctx_begin(ctx, dep); \
        if (rc_) return rc_;                 \
    } while (0)

static int32_t check_live(const nz_erosion_params *ep, const nz_tile_set_meta *tm, int32_t res) {
    NZ_REQUIRE(ep && tm, "ep / tm is NULL");
    NZ_REQUIRE(res >= 2 && res <= 32768, "resolution %d out of range [2,32768]", res);
    NZ_REQUIRE(tm->GENERATOR_RES[0] == res && tm->GENERATOR_RES[1] == res, "tm.GENERATOR_RES (%d, %d) is not the plane's resolution %d",
               tm->GENERATOR_RES[0], tm->GENERATOR_RES[1], res);
    NZ_REQUIRE(tm->HEIGHT != 0, "tm.HEIGHT is 0");
    return NZ_OK;
}

// ---- containers --------------------------------------------------------------------------------------------------------
extern "C" int32_t nz_particle_queue_create(nz_ctx *ctx, int32_t capacity, nz_particle_queue **out) {
    NZ_REQUIRE(ctx && out && capacity >= 1, "ctx/out is NULL or capacity < 1");
    NZ_HIP(hipSetDevice(ctx->device));
    nz_particle_queue *q = new nz_particle_queue();
    q->capacity = capacity;
    if (hipMalloc((void **)&q->hdr, 16) != hipSuccess || hipMalloc((void **)&q->data, (size_t)capacity * sizeof(nz_particle)) != hipSuccess) {
        if (q->hdr) (void)hipFree(q->hdr);
        delete q;
        nz_set_error("hipMalloc of a %d-particle queue failed", capacity);
        return NZ_ERR_NOMEM;
    }
    int32_t hdr[4] = {0, capacity, 0, 0};
    NZ_HIP(hipMemcpy(q->hdr, hdr, sizeof hdr, hipMemcpyHostToDevice));
    *out = q;
    return NZ_OK;
}

extern "C" int32_t nz_particle_queue_destroy(nz_ctx *ctx, nz_particle_queue *q) {
    NZ_REQUIRE(ctx, "ctx is NULL");
    if (!q) return NZ_OK;
    NZ_HIP(hipSetDevice(ctx->device));
    NZ_HIP(hipStreamSynchronize(ctx->stream));
    (void)hipFree(q->hdr);
    (void)hipFree(q->data);
    delete q;
    return NZ_OK;
}

// NativeQueue.Count (host synchronises with the ctx's stream); NZ_ERR_NOMEM if a job found the queue too small
extern "C" int32_t nz_particle_queue_count(nz_ctx *ctx, nz_particle_queue *q, int32_t *count) {
    NZ_REQUIRE(ctx && q && count, "ctx/queue/count is NULL");
    NZ_HIP(hipSetDevice(ctx->device));
    int32_t hdr[4];
    NZ_HIP(hipMemcpyAsync(hdr, q->hdr, sizeof hdr, hipMemcpyDeviceToHost, ctx->stream));
    NZ_HIP(hipStreamSynchronize(ctx->stream));
    *count = hdr[0] < hdr[1] ? hdr[0] : hdr[1];
    if (hdr[2] || hdr[0] > hdr[1]) {
        nz_set_error("particle queue overflow: capacity %d, %d wanted", hdr[1], hdr[0]);
        return NZ_ERR_NOMEM;
    }
    return NZ_OK;
}

extern "C" int32_t nz_particle_queue_download(nz_ctx *ctx, nz_particle_queue *q, nz_particle *host, int32_t max_count,
                                              int32_t *count) {
    int32_t n = 0;
    int32_t rc = nz_particle_queue_count(ctx, q, &n);
    if (rc) return rc;
    if (n > max_count) n = max_count;
    if (n > 0) NZ_HIP(hipMemcpy(host, q->data, (size_t)n * sizeof(nz_particle), hipMemcpyDeviceToHost));
    if (count) *count = n;
    return NZ_OK;
}

extern "C" int32_t nz_particle_queue_upload(nz_ctx *ctx, nz_particle_queue *q, const nz_particle *host, int32_t count) {
    NZ_REQUIRE(ctx && q && (host || count == 0) && count >= 0 && count <= q->capacity, "bad queue upload");
    NZ_HIP(hipSetDevice(ctx->device));
    NZ_HIP(hipStreamSynchronize(ctx->stream));
    if (count) NZ_HIP(hipMemcpy(q->data, host, (size_t)count * sizeof(nz_particle), hipMemcpyHostToDevice));
    int32_t hdr[4] = {count, q->capacity, 0, 0};
    NZ_HIP(hipMemcpy(q->hdr, hdr, sizeof hdr, hipMemcpyHostToDevice));
    return NZ_OK;
}

// ClearQueueJob<BeyerParticle>.ScheduleRun(queue, deps), MultiThreadErosionJob.cs:133-153
extern "C" int32_t nz_clear_particle_queue(nz_ctx *ctx, nz_particle_queue *q, nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_REQUIRE(q, "queue is NULL");
    NZ_HIP(hipMemsetAsync(q->hdr, 0, 4, ctx->stream));
    return nz_ctx_finish(ctx, out);
}

extern "C" int32_t nz_erosive_events_create(nz_ctx *ctx, int32_t resolution, nz_erosive_events **out) {
    NZ_REQUIRE(ctx && out && resolution >= 2 && resolution <= 32768, "ctx/out is NULL or resolution out of range");
    NZ_HIP(hipSetDevice(ctx->device));
    nz_erosive_events *ev = new nz_erosive_events();
    ev->res = resolution;
    const size_t n = (size_t)resolution * resolution;
    bool ok = hipMalloc((void **)&ev->acc, 3 * n * 8) == hipSuccess && hipMalloc((void **)&ev->touched, n * 4) == hipSuccess &&
              hipMalloc((void **)&ev->list[0], n * 4) == hipSuccess && hipMalloc((void **)&ev->list[1], n * 4) == hipSuccess &&
              hipMalloc((void **)&ev->counters, 16) == hipSuccess && hipMalloc((void **)&ev->sediment, n * 4) == hipSuccess;
    if (ok) ok = hipMemset(ev->acc, 0, 3 * n * 8) == hipSuccess && hipMemset(ev->touched, 0, n * 4) == hipSuccess &&
                 hipMemset(ev->counters, 0, 16) == hipSuccess && hipMemset(ev->sediment, 0, n * 4) == hipSuccess;
    if (!ok) {
        void *ps[] = {ev->acc, ev->touched, ev->list[0], ev->list[1], ev->counters, ev->sediment};
        for (void *p : ps)
            if (p) (void)hipFree(p);
        delete ev;
        (void)hipGetLastError();
        nz_set_error("hipMalloc of the event planes (%d^2 cells x 40 bytes) failed", resolution);
        return NZ_ERR_NOMEM;
    }
    *out = ev;
    return NZ_OK;
}

extern "C" int32_t nz_erosive_events_destroy(nz_ctx *ctx, nz_erosive_events *ev) {
    NZ_REQUIRE(ctx, "ctx is NULL");
    if (!ev) return NZ_OK;
    NZ_HIP(hipSetDevice(ctx->device));
    NZ_HIP(hipStreamSynchronize(ctx->stream));
    void *ps[] = {ev->acc, ev->touched, ev->list[0], ev->list[1], ev->counters, ev->sediment, ev->pile_scratch, ev->pile_blocks,
                  ev->height_snapshot};
    for (void *p : ps)
        if (p) (void)hipFree(p);
    delete ev;
    return NZ_OK;
}

extern "C" float *nz_erosive_events_sediment(nz_erosive_events *ev) { return ev ? ev->sediment : nullptr; }

// events of the last nz_queued_beyer_cycle (host synchronises)
extern "C" int32_t nz_erosive_events_count(nz_ctx *ctx, nz_erosive_events *ev, int32_t *events) {
    NZ_REQUIRE(ctx && ev && events, "ctx/events is NULL");
    NZ_HIP(hipSetDevice(ctx->device));
    int32_t c[4];
    NZ_HIP(hipMemcpyAsync(c, ev->counters, sizeof c, hipMemcpyDeviceToHost, ctx->stream));
    NZ_HIP(hipStreamSynchronize(ctx->stream));
    *events = c[3];
    return NZ_OK;
}

#ifdef NZ_DESCENT_PROBE
extern "C" int32_t nz_debug_descent_probe(unsigned long long *out8, int32_t reset) {
    if (out8) NZ_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_descent_probe), 64));
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        NZ_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_descent_probe), z, 64));
    }
    return NZ_OK;
}
#endif

// ---- jobs --------------------------------------------------------------------------------------------------------------
extern "C" int32_t nz_fill_beyer_queue(nz_ctx *ctx, nz_particle_queue *particles, const nz_erosion_params *ep,
                                       const nz_tile_set_meta *tm, int32_t generationRound, int32_t res,
                                       int32_t maxParticles, int32_t seed, int32_t concurrency, nz_handle dep,
                                       nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_REQUIRE(particles, "particles is NULL");
    if (int32_t rc = check_live(ep, tm, res)) return rc;
    NZ_REQUIRE(concurrency >= 1 && concurrency <= 1024 && maxParticles >= 0, "concurrency %d out of range [1,1024] or maxParticles < 0",
               concurrency);
    hipLaunchKernelGGL(fill_queue_kernel, dim3(1), dim3(1024), 0, ctx->stream, particles->hdr,
                       particles->data, generationRound, res, maxParticles, seed, concurrency);
    NZ_HIP(hipGetLastError());
    return nz_ctx_finish(ctx, out);
}

extern "C" int32_t nz_queued_beyer_cycle(nz_ctx *ctx, const float *height, const float *pool, const float *flow,
                                         const float *track, nz_particle_queue *particles, nz_erosive_events *events,
                                         const nz_erosion_params *ep, const nz_tile_set_meta *tm, int32_t eventLimit,
                                         int32_t res, nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    (void)track;       // the job's WorldTile carries it; the descent never reads it
    (void)eventLimit;  // unused by the reference job as well (:205)
    NZ_REQUIRE(height && pool && flow && particles && events, "a plane / the queue / the events is NULL");
    if (int32_t rc = check_live(ep, tm, res)) return rc;
    NZ_REQUIRE(events->res == res, "events were created for resolution %d", events->res);
    live_planes P{height, pool, flow, events->acc, events->touched, events->list[events->cur], events->counters, events->cur,
                  (size_t)res * res};
    NZ_HIP(hipMemsetAsync(events->counters + 2, 0, 8, ctx->stream));  // piles and events of this cycle
    // the count lives on the device: waves beyond it leave; a wave per SIMD slot at most, each takes 64 particles at a
    // time (a queue sized for a whole plane launched 131 k waves for 10 000 particles)
    const unsigned blocks = (unsigned)std::min<long long>(((long long)particles->capacity + NZ_DESCENT_LANES - 1) / NZ_DESCENT_LANES, 8192);
    hipLaunchKernelGGL(descent_kernel, dim3(blocks), dim3(64), 0, ctx->stream, P, particles->hdr, particles->data, *ep, res,
                       (float)tm->HEIGHT, tm->PATCH_RES[0]);
    NZ_HIP(hipGetLastError());
    return nz_ctx_finish(ctx, out);
}

extern "C" int32_t nz_process_beyer_erosive_events(nz_ctx *ctx, float *height, float *pool, float *flow, float *track,
                                                   nz_erosive_events *events, const nz_erosion_params *ep,
                                                   const nz_tile_set_meta *tm, int32_t res, nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    (void)height;
    (void)flow;
    NZ_REQUIRE(pool && track && events, "pool / track / events is NULL");
    if (int32_t rc = check_live(ep, tm, res)) return rc;
    NZ_REQUIRE(events->res == res, "events were created for resolution %d", events->res);
    const int cur = events->cur, prev = cur ^ 1;
    hipLaunchKernelGGL(forget_kernel, dim3(512), dim3(CT), 0, ctx->stream, events->sediment, events->list[prev], events->counters, prev);
    NZ_HIP(hipMemsetAsync(events->counters + prev, 0, 4, ctx->stream));
    hipLaunchKernelGGL(process_events_kernel, dim3(NZ_EVENTS_BLOCKS), dim3(CT), 0, ctx->stream, pool, track, events->sediment,
                       events->acc, events->touched, events->list[cur], events->counters, cur, (size_t)res * res,
                       ep->POOL_PLACEMENT_MULTIPLIER, ep->TRACK_PLACEMENT_MULTIPLIER);
    NZ_HIP(hipGetLastError());
    events->cur = prev;  // the next cycle's events go to the other list; this one is forgotten then
    return nz_ctx_finish(ctx, out);
}

// ErodeHeightMaps; `fl` != NULL: and UpdateFlowFromTrackJob, its sibling (nz_erode_height_maps_and_flow)
static int32_t erode_height_maps(nz_ctx *ctx, float *height, nz_erosive_events *events, const nz_erosion_params *ep,
                                 const nz_tile_set_meta *tm, int32_t res, const flow_track_args *fl, nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_REQUIRE(height && events, "height / events is NULL");
    if (int32_t rc = check_live(ep, tm, res)) return rc;
    NZ_REQUIRE(events->res == res, "events were created for resolution %d", events->res);
    NZ_REQUIRE(!fl || (fl->pool != height && fl->flow != height && fl->track != height),
               "the flow update's planes must not be the height plane: the two jobs run side by side");
    NZ_REQUIRE(ep->PILING_RADIUS >= 0 && ep->PILING_RADIUS <= 50, "PILING_RADIUS %d out of range [0,50]", ep->PILING_RADIUS);
    const float thr = ep->PILE_THRESHOLD / (float)tm->HEIGHT;
    // the events ProcessBeyerErosiveEvents has just written are the cells of the list it filled (it flipped `cur` after)
    const int slot = events->cur ^ 1;
    const int D = ep->PILING_RADIUS;
    const int B = 2 * (D + 1), nb = (res + B - 1) / B;
    const int pile_cap = ((nb + 1) / 2) * ((nb + 1) / 2);  // blocks of one colour, at most
    if (D >= 1) {
        if ((size_t)nb * nb != events->pile_blocks_n) {  // another radius: rare
            NZ_HIP(hipStreamSynchronize(ctx->stream));
            if (events->pile_blocks) (void)hipFree(events->pile_blocks);
            events->pile_blocks = nullptr;
            events->pile_blocks_n = 0;
            NZ_HIP(hipMalloc((void **)&events->pile_blocks, ((size_t)nb * nb + PILE_CTL + 4 * (size_t)pile_cap) * 4));
            events->pile_blocks_n = (size_t)nb * nb;
        }
        NZ_HIP(hipMemsetAsync(events->pile_blocks, 0, ((size_t)nb * nb + PILE_CTL) * 4, ctx->stream));
    }
    int32_t *pile_ctl = D >= 1 ? events->pile_blocks + (size_t)nb * nb : nullptr;
    int32_t *pile_list = D >= 1 ? pile_ctl + PILE_CTL : nullptr;
    // safe mode keeps the plane as the job found it (one plane copy per cycle, ~25 us at 8192^2)
    static const bool ticket_wanted = [] { const char *e = getenv("NZ_PILE_TICKET"); return !e || atoi(e) != 0; }();
    const bool safe = ctx->pile_safe && D >= 1 && ticket_wanted && !ctx->pile_ticket_off;
    if (safe) {
        if (!events->height_snapshot) NZ_HIP(hipMalloc((void **)&events->height_snapshot, (size_t)res * res * sizeof(float)));
        NZ_HIP(hipMemcpyAsync(events->height_snapshot, height, (size_t)res * res * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    }
    auto disperse = [&]() -> int32_t {
        hipLaunchKernelGGL(disperse_list_kernel, dim3(2048), dim3(CT), 0, ctx->stream, height, events->sediment, events->list[slot],
                           events->counters, slot, res, thr, D >= 1 ? events->pile_blocks : nullptr, B, nb);
        const long long frame = (long long)std::min(res, 4) * res + (long long)std::max(res - 4, 0) * 4;
        hipLaunchKernelGGL(disperse_frame_kernel, dim3((unsigned)((frame + CT - 1) / CT)), dim3(CT), 0, ctx->stream, height,
                           events->sediment, res, thr);
        NZ_HIP(hipGetLastError());
        return NZ_OK;
    };
    NZ_TRY_(disperse());
    // PileSolver.Init :1058-1098: vertex offsets (host), GetOffset = dist * dirA + i * (dirB - dirA)
    if (D >= 1) {
        std::vector<short2> ofs;
        const int DAX[4] = {0, 1, 0, -1}, DAZ[4] = {1, 0, -1, 0}, DBX[4] = {1, 0, -1, 0}, DBZ[4] = {0, -1, 0, 1};
        for (int dist = 0; dist < D; dist++)
            for (int dir = 0; dir < 4; dir++)
                for (int i = 0; i <= dist + 1; i++)
                    ofs.push_back(make_short2((short)(dist * DAX[dir] + i * (DBX[dir] - DAX[dir])),
                                              (short)(dist * DAZ[dir] + i * (DBZ[dir] - DAZ[dir]))));
        const size_t bytes = ofs.size() * sizeof(short2);
        if (bytes != events->pile_scratch_bytes) {  // the table of another radius: rebuild (rare)
            NZ_HIP(hipStreamSynchronize(ctx->stream));
            if (events->pile_scratch) (void)hipFree(events->pile_scratch);
            events->pile_scratch = nullptr;
            NZ_HIP(hipMalloc(&events->pile_scratch, bytes));
            NZ_HIP(hipMemcpy(events->pile_scratch, ofs.data(), bytes, hipMemcpyHostToDevice));
            events->pile_scratch_bytes = bytes;
        }
        const int nverts = (int)ofs.size();
        // vertices + the block's sediment + the vertex offsets
        const size_t lds = (((size_t)nverts * 9 + 15) & ~(size_t)15) + (size_t)B * B * 4 + (size_t)nverts * 4;
        if (lds > 64 * 1024)
            NZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pile_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        // NZ_PILE_TICKET (1): all colours in one launch, the busy blocks handed out by ticket (pile_ticket_kernel); 0: one
        // launch per colour over every block of it -- also what a context does for good after a ticket launch of its gave up
        static const bool ticket_env = [] { const char *e = getenv("NZ_PILE_TICKET"); return !e || atoi(e) != 0; }();
        const bool ticket = ticket_env && !ctx->pile_ticket_off;
        const float incr = ep->MIN_PILE_INCREMENT / (float)tm->HEIGHT;
        auto colour_launches = [&]() -> int32_t {
            for (int colour = 0; colour < 4; colour++) {
                const int cx = colour & 1, cz = colour >> 1;
                const int bxn = (nb - cx + 1) / 2, bzn = (nb - cz + 1) / 2;
                if (bxn <= 0 || bzn <= 0) continue;
                hipLaunchKernelGGL(pile_kernel, dim3((unsigned)(bxn * bzn)), dim3(64), lds, ctx->stream, height, events->sediment,
                                   events->pile_blocks, (const short2 *)events->pile_scratch, nverts, res, D, B, nb, cx, cz, thr, incr);
                NZ_HIP(hipGetLastError());
            }
            return NZ_OK;
        };
        if (ticket) {
            unsigned *err_host = nullptr;
            NZ_TRY_(nz_ctx_error_word(ctx, &err_host));
            if (lds > 64 * 1024)
                NZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pile_ticket_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(pile_list_kernel, dim3((unsigned)(((size_t)nb * nb + 1023) / 1024)), dim3(1024), 0, ctx->stream,
                               events->pile_blocks, pile_ctl, pile_list, pile_cap, nb);
            // 2048 resident waves: 1024 leave blocks waiting for a wave (248 us), 3584 / 7168 fill the CUs with waves that
            // poll (239 / 279 us against 211)
            const unsigned grid = (unsigned)std::min<long long>((long long)nb * nb, 2048);
            const pile_ticket_args pa{height, events->sediment, events->pile_blocks, pile_list, pile_ctl, pile_cap,
                                      (const short2 *)events->pile_scratch, nverts, res, D, B, nb, thr, incr, err_host,
                                      g_pile_spin_limit};
            const size_t flow_wgs = fl ? (fl->n + FLOW_WG_CELLS - 1) / FLOW_WG_CELLS : 0;
            // the flow workgroups of the one-call form are launched under the pile solver's dynamic LDS size: a few KB at the
            // default radius, ~110 KB at PILING_RADIUS 50 -- where each of the res^2 / 1024 flow workgroups would hold a CU
            // alone with one wave.  Beyond 16 KB the flow update is its own launch again (nz_launch_flow_from_track below);
            // so it is in safe mode, whose retry runs the pile solver alone
            if (fl && !safe && lds <= 16 * 1024 && flow_wgs + grid < 0x7fffffffull) {
                hipLaunchKernelGGL(pile_ticket_flow_kernel, dim3(grid + (unsigned)flow_wgs), dim3(64), lds, ctx->stream, pa, grid, *fl);
                fl = nullptr;  // done
            } else {
                hipLaunchKernelGGL(pile_ticket_kernel, dim3(grid), dim3(64), lds, ctx->stream, pa);
            }
            NZ_HIP(hipGetLastError());
            if (safe) {
                // Safe mode (nz_ctx_set_pile_safe): the ticket launch is waited for here.  Its bounded wait never gives up while
                // the protocol holds (a block waits only for lower tickets, which resident waves hold) -- should it, the plane it
                // left is invalid: put back what the job found, and run the whole job again the way that cannot wait, a launch
                // per colour, on this context from now on.  The caller sees a job that succeeded.
                NZ_HIP(hipStreamSynchronize(ctx->stream));
                volatile unsigned *w = reinterpret_cast<volatile unsigned *>(ctx->chain_err);
                if (w && w[1]) {
                    w[1] = 0;
                    ctx->pile_ticket_off = true;
                    ctx->pile_retries++;
                    NZ_HIP(hipMemcpyAsync(height, events->height_snapshot, (size_t)res * res * sizeof(float), hipMemcpyDeviceToDevice,
                                          ctx->stream));
                    NZ_HIP(hipMemsetAsync(events->pile_blocks, 0, ((size_t)nb * nb + PILE_CTL) * 4, ctx->stream));
                    NZ_TRY_(disperse());
                    NZ_TRY_(colour_launches());
                }
            }
        } else {
            NZ_TRY_(colour_launches());
        }
    }
    // no pile solver launch to carry it (PILING_RADIUS 0, NZ_PILE_TICKET=0, a solver with more than 16 KB of LDS): the flow update by itself
    if (fl) NZ_TRY_(nz_launch_flow_from_track(ctx->stream, fl->pool, fl->flow, fl->track, fl->n, fl->flowLossRate, fl->evaporation));
    return nz_ctx_finish(ctx, out);
}

// Test hook: polls of a lower-coloured neighbour's flag after which a block of the pile solver's ticket launch gives up (<= 0: the
// default, 2^22 = seconds).  With a limit of 1 every block that has to wait at all gives up: the time-out path runs.
extern "C" int32_t nz_debug_pile_poll_limit(int32_t polls) {
    g_pile_spin_limit = polls > 0 ? polls : 1 << 22;
    return NZ_OK;
}
extern "C" int32_t nz_ctx_set_pile_safe(nz_ctx *ctx, int32_t on) {
    NZ_REQUIRE(ctx, "ctx is NULL");
    ctx->pile_safe = on != 0;
    return NZ_OK;
}
extern "C" int32_t nz_ctx_pile_retries(nz_ctx *ctx) { return ctx ? ctx->pile_retries : -1; }

extern "C" int32_t nz_erode_height_maps(nz_ctx *ctx, float *height, nz_erosive_events *events, const nz_erosion_params *ep,
                                        const nz_tile_set_meta *tm, int32_t res, nz_handle dep, nz_handle *out) {
    return erode_height_maps(ctx, height, events, ep, tm, res, nullptr, dep, out);
}

extern "C" int32_t nz_erode_height_maps_and_flow(nz_ctx *ctx, float *height, nz_erosive_events *events, float *pool, float *flow,
                                                 float *track, const nz_erosion_params *ep, const nz_tile_set_meta *tm,
                                                 int32_t res, nz_handle dep, nz_handle *out) {
    if (!ctx) return NZ_ERR_INVALID;
    if (!pool || !flow || !track || !ep || !tm || res < 1) {
        nz_set_error("nz_erode_height_maps_and_flow: pool / flow / track / ep / tm is NULL");
        return NZ_ERR_INVALID;
    }
    flow_track_args fl{pool, flow, track, (size_t)res * res, ep->FLOW_LOSS_RATE, ep->SURFACE_EVAPORATION_RATE / (float)tm->HEIGHT, 0};
    fl.aligned = ((reinterpret_cast<uintptr_t>(pool) | reinterpret_cast<uintptr_t>(flow) | reinterpret_cast<uintptr_t>(track)) & 15) == 0;
    return erode_height_maps(ctx, height, events, ep, tm, res, &fl, dep, out);
}

extern "C" int32_t nz_curviture_map(nz_ctx *ctx, uint8_t *texture, const float *height, const nz_tile_set_meta *tm,
                                    int32_t target, int32_t res, int32_t meshRes, nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_REQUIRE(texture && height && tm, "texture / height / tm is NULL");
    NZ_REQUIRE(res >= 2 && meshRes >= 1 && meshRes <= res && target >= 0 && target < 4, "bad resolution / channel");
    hipLaunchKernelGGL(curviture_kernel, dim3((unsigned)((meshRes + CT - 1) / CT), (unsigned)meshRes), dim3(CT), 0, ctx->stream,
                       texture, target, height, res, meshRes, (res - meshRes) / 2, (float)tm->HEIGHT, tm->PATCH_RES[0]);
    NZ_HIP(hipGetLastError());
    return nz_ctx_finish(ctx, out);
}

extern "C" int32_t nz_set_rgba32(nz_ctx *ctx, const float *src, uint8_t *texture, int32_t target, int32_t dataRes,
                                 int32_t meshRes, float scale, nz_handle dep, nz_handle *out) {
    NZ_BEGIN(ctx, dep);
    NZ_REQUIRE(texture && src, "texture / src is NULL");
    NZ_REQUIRE(dataRes >= 1 && meshRes >= 1 && meshRes <= dataRes && target >= 0 && target < 4, "bad resolution / channel");
    hipLaunchKernelGGL(set_rgba32_kernel, dim3((unsigned)((meshRes + CT - 1) / CT), (unsigned)meshRes), dim3(CT), 0, ctx->stream,
                       texture, target, src, dataRes, meshRes, (dataRes - meshRes) / 2, scale);
    NZ_HIP(hipGetLastError());
    return nz_ctx_finish(ctx, out);
}

// the queue's device pieces, for PoolAutomataJob's drain (nz_stages.cpp)
int32_t *nz_particle_queue_hdr(nz_particle_queue *q) { return q ? q->hdr : nullptr; }
nz_particle *nz_particle_queue_data(nz_particle_queue *q) { return q ? q->data : nullptr; }
