// nz_elementwise.hip -- per-cell stages either side of the hot path:
// ConstantJob<ConstantMultiply|ConstantBinarize> (Filter/ConstantJob.cs, Filter/Operators/SimpleMutation.cs:16-55),
// ReductionJob<Subtract|Multiply|RootSumSquares|Max|Min Tiles> (Filter/ReductionJob.cs, SimpleMutation.cs:57-171),
// CurveJob<CurveOperator> (Filter/Curve/CurveJob.cs:56-89).
// Pure streaming kernels: 16 B per lane, in place (the reference's tmp + flush copy is not needed for an
// element-wise update); 8 or 12 B/cell.
#include "nz_internal.hpp"
#include "nz_flow_track.hpp"

namespace {

constexpr int CT = 256;

template <int OP>
__device__ __forceinline__ float constant_op(float v, float c) {
    if constexpr (OP == 0) return v * c;          // ConstantMultiply
    else return v >= c ? 1.0f : 0.0f;             // ConstantBinarize
}

template <int OP>
__device__ __forceinline__ float reduce_op(float a, float b) {
    if constexpr (OP == 0) return a - b;                         // SubtractTiles
    else if constexpr (OP == 1) return a * b;                    // MultiplyTiles
    else if constexpr (OP == 2) return sqrtf((a * a) + (b * b)); // RootSumSquaresTiles
    else if constexpr (OP == 3) return fmaxf(a, b);              // MaxTiles
    else return fminf(a, b);                                     // MinTiles
}

template <int OP>
__global__ __launch_bounds__(CT) void constant_kernel(float *data, size_t n, float c, int aligned) {
    size_t i = ((size_t)blockIdx.x * CT + threadIdx.x) * 4;
    if (i >= n) return;
    if (aligned && i + 4 <= n) {
        float4 v = *reinterpret_cast<float4 *>(data + i);
        v.x = constant_op<OP>(v.x, c); v.y = constant_op<OP>(v.y, c);
        v.z = constant_op<OP>(v.z, c); v.w = constant_op<OP>(v.w, c);
        *reinterpret_cast<float4 *>(data + i) = v;
    } else {
        for (size_t k = i; k < n && k < i + 4; k++) data[k] = constant_op<OP>(data[k], c);
    }
}

template <int OP>
__global__ __launch_bounds__(CT) void reduce_kernel(float *l, const float *__restrict__ r, size_t n, int aligned) {
    size_t i = ((size_t)blockIdx.x * CT + threadIdx.x) * 4;
    if (i >= n) return;
    if (aligned && i + 4 <= n) {
        float4 a = *reinterpret_cast<float4 *>(l + i);
        float4 b = *reinterpret_cast<const float4 *>(r + i);
        a.x = reduce_op<OP>(a.x, b.x); a.y = reduce_op<OP>(a.y, b.y);
        a.z = reduce_op<OP>(a.z, b.z); a.w = reduce_op<OP>(a.w, b.w);
        *reinterpret_cast<float4 *>(l + i) = a;
    } else {
        for (size_t k = i; k < n && k < i + 4; k++) l[k] = reduce_op<OP>(l[k], r[k]);
    }
}

// CurveOperator.Apply, Filter/Curve/CurveJob.cs:69-80
__device__ __forceinline__ float curve_apply(float v, const float *__restrict__ curve, int curveSize) {
    float rect = fmaxf(0.0f, fminf(1.0f, v)) * (float)curveSize;
    float lowerIdx = fminf(floorf(rect), (float)(curveSize - 2));
    float left = curve[(int)lowerIdx];
    float right = curve[(int)lowerIdx + 1];
    float value = left + (rect - lowerIdx) * (right - left);  // math.lerp
    value = fmaxf(0.0f, value);
    return fminf(1.0f, value);
}

__global__ __launch_bounds__(CT) void curve_kernel(float *data, size_t n, const float *__restrict__ curve, int curveSize,
                                                  int aligned) {
    extern __shared__ float s_curve[];
    for (int i = threadIdx.x; i < curveSize; i += CT) s_curve[i] = curve[i];
    __syncthreads();
    size_t i = ((size_t)blockIdx.x * CT + threadIdx.x) * 4;
    if (i >= n) return;
    if (aligned && i + 4 <= n) {
        float4 v = *reinterpret_cast<float4 *>(data + i);
        v.x = curve_apply(v.x, s_curve, curveSize); v.y = curve_apply(v.y, s_curve, curveSize);
        v.z = curve_apply(v.z, s_curve, curveSize); v.w = curve_apply(v.w, s_curve, curveSize);
        *reinterpret_cast<float4 *>(data + i) = v;
    } else {
        for (size_t k = i; k < n && k < i + 4; k++) data[k] = curve_apply(data[k], s_curve, curveSize);
    }
}

// ---- live erosion: the deterministic grid jobs (planes indexed x * res + z, LiveErosionDataTypes.cs:608-610) -------
// WorldTile.UpdateFlowMapFromTrack, LiveErosionDataTypes.cs:869-886 (UpdateFlowFromTrackJob): nz_flow_track.hpp
__global__ __launch_bounds__(CT) void flow_from_track_kernel(float *__restrict__ pool, float *__restrict__ flow,
                                                            float *__restrict__ track, size_t n, float flowLossRate,
                                                            float evaporation /* SURFACE_EVAPORATION_RATE / tm.HEIGHT */,
                                                            int aligned) {
    const size_t i = ((size_t)blockIdx.x * CT + threadIdx.x) * 4;
    if (i >= n) return;
    if (aligned && i + 4 <= n) {
        const float4 pv = *reinterpret_cast<const float4 *>(flow + i), tv = *reinterpret_cast<const float4 *>(track + i),
                     po = *reinterpret_cast<const float4 *>(pool + i);
        flow_from_track_quad(pool, flow, track, i, pv, tv, po, flowLossRate, evaporation);
    } else {
        flow_from_track_cells(pool, flow, track, i, n < i + 4 ? n : i + 4, flowLossRate, evaporation);
    }
}

// FloodedNeighbor ordering (LiveErosionDataTypes.cs:1013-1050): by the hash of height + water = the float's bits as a
// signed int (+-0 -> 0); the same cell compares equal.
struct flooded {
    int idx;
    float height, water;
};
__device__ __forceinline__ int float_hash(float f) { return f == 0.0f ? 0 : __builtin_bit_cast(int, f); }
__device__ __forceinline__ int flooded_cmp(const flooded &a, const flooded &b) {
    if (a.idx == b.idx) return 0;
    return float_hash(a.height + a.water) > float_hash(b.height + b.water) ? 1 : -1;
}

// One step of the walk: WorldTile.SpreadPool (LiveErosionDataTypes.cs:938-1010) at (x, z) with water standing.  `nh` /
// `nw` = height / water of the up, right, down, left neighbours; `carry` enters as the right-hand neighbour's water and
// leaves as its value after the step (the left-hand neighbour of the walk's next step).
//
// DRAIN (drainParticles): a pool that finds a dry, lower neighbour does not wet it but leaves as ONE BeyerParticle
// (pid 64000, at the neighbour, carrying the water, LiveErosionDataTypes.cs:971-984) appended to the particle queue; the
// order of the queue is whatever the atomics make it -- the descent's results do not depend on it (nz_live.hip).
// A step acts only where at least 1E-3 of water stands when the walk reaches it (SpreadPool returns at once on a dry
// cell, :940, and every transfer is guarded by hWater >= 1E-3, which no skipped transfer can raise); its own cell is
// written by no other step of the pass, so that is decided by the plane as the pass finds it.
__device__ __forceinline__ bool pool_step_acts(float w) { return w > 0.0f && !(w < 1E-3f); }

// The acting-step bits of the four colour passes (class c = 2 xoff + zoff): bit b of word [(c * words + w) * walks + k]
// = step 32 w + b of walk k, the cell (xoff + (k & 1) + 2 (32 w + b), 2 k + zoff).
struct pool_masks {
    unsigned *m;
    int words, walks;
    // the sparse form of a job (pool_sparse_kernel): the non-empty words as a list.  listed[word] != 0: the word has an
    // entry; list[i] = class | word << 2 | walk << 12; ctl = {entries, done, scan_all}.  NULL: no list is kept.
    unsigned *listed = nullptr, *list = nullptr;
    int *ctl = nullptr;
    int cap = 0;
};
__device__ __forceinline__ unsigned pool_list_pack(int c, int w, int k) { return (unsigned)c | ((unsigned)w << 2) | ((unsigned)k << 12); }
// a word that has just received its first bit (or is found non-empty by the masks kernel) gets its one entry
__device__ __forceinline__ void pool_list_add(const pool_masks &pm, int c, int w, int k) {
    const size_t wi = ((size_t)c * pm.words + w) * pm.walks + k;
    if (__hip_atomic_exchange(pm.listed + wi, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    const int slot = __hip_atomic_fetch_add(pm.ctl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (slot < pm.cap) pm.list[slot] = pool_list_pack(c, w, k);  // beyond the capacity: the passes scan every word instead
}
__device__ __forceinline__ void pool_mark_acting(const pool_masks &pm, int cx, int cz) {
    const int k = cz >> 1, t = cx - (k & 1);
    if (k >= pm.walks || t < 0) return;  // the last row of an odd plane, column 0 of an odd walk: no pass has a step there
    const int c = 2 * (t & 1) + (cz & 1), st = t >> 1;
    unsigned *word = pm.m + ((size_t)c * pm.words + (st >> 5)) * pm.walks + k;
    if (pm.list) {
        const unsigned old = __hip_atomic_fetch_or(word, 1u << (st & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == 0u) pool_list_add(pm, c, st >> 5, k);
        return;
    }
    // no value comes back: the walk does not wait for it
    (void)__hip_atomic_fetch_or(word, 1u << (st & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool DRAIN, bool MARK>
__device__ __forceinline__ void spread_pool_step(float *pool, int res, int x, int z, int zu, int zd, float hWater,
                                                 const float hLand, const float nh[4], const float nw[4], float &carry,
                                                 int32_t *drain_hdr, nz_particle *drain_data, const pool_masks &pm) {
    const int xr = min(x + 1, res - 1), xl = max(x - 1, 0);
    const int idx = x * res + z, right_idx = xr * res + z;
    float tHeight = hLand + hWater;
    flooded b[4];
    b[0].idx = x * res + zu;  b[1].idx = right_idx;  b[2].idx = x * res + zd;  b[3].idx = xl * res + z;
    int key[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        b[e].height = nh[e];
        b[e].water = nw[e];
        key[e] = float_hash(nh[e] + nw[e]);
    }
    // NativeArray.Sort() of com.unity.collections 1.4.0 on 4 elements: insertion sort, element i+1 moves
    // left while it compares < 0 (flooded_cmp above: same cell -> 0, else key > key ? 1 : -1).  The same
    // comparisons in the same order, applied with selects so that the lanes of a wave do not diverge (the
    // walk is bound by the length of its instruction stream)
#define NZ_LT(ti, tk, j) ((ti) != b[j].idx && !((tk) > key[j]))
#define NZ_SEL(dst, c, src_t) do { b[dst].idx = (c) ? (src_t##i) : b[dst].idx; b[dst].height = (c) ? (src_t##h) : b[dst].height; \
                                   b[dst].water = (c) ? (src_t##w) : b[dst].water; key[dst] = (c) ? (src_t##k) : key[dst]; } while (0)
#define NZ_MOV(dst, c, src) do { b[dst].idx = (c) ? b[src].idx : b[dst].idx; b[dst].height = (c) ? b[src].height : b[dst].height; \
                                 b[dst].water = (c) ? b[src].water : b[dst].water; key[dst] = (c) ? key[src] : key[dst]; } while (0)
    {
        int ti = b[1].idx, tk = key[1];
        float th = b[1].height, tw = b[1].water;
        bool c0 = NZ_LT(ti, tk, 0);
        NZ_MOV(1, c0, 0);
        NZ_SEL(0, c0, t);
        ti = b[2].idx; tk = key[2]; th = b[2].height; tw = b[2].water;
        bool c1 = NZ_LT(ti, tk, 1);
        c0 = c1 && NZ_LT(ti, tk, 0);
        NZ_MOV(2, c1, 1);
        NZ_MOV(1, c0, 0);
        NZ_SEL(1, c1 && !c0, t);
        NZ_SEL(0, c0, t);
        ti = b[3].idx; tk = key[3]; th = b[3].height; tw = b[3].water;
        bool c2 = NZ_LT(ti, tk, 2);
        c1 = c2 && NZ_LT(ti, tk, 1);
        c0 = c1 && NZ_LT(ti, tk, 0);
        NZ_MOV(3, c2, 2);
        NZ_MOV(2, c1, 1);
        NZ_MOV(1, c0, 0);
        NZ_SEL(2, c2 && !c1, t);
        NZ_SEL(1, c1 && !c0, t);
        NZ_SEL(0, c0, t);
    }
#undef NZ_LT
#undef NZ_SEL
#undef NZ_MOV
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const float bw = b[e].water, bh = b[e].height;
        const float diffV = tHeight - (bh + bw);
        const bool go = !(hWater < 1E-3f);
        const bool spill = bw <= 0.0f && hLand >= bh;                       // the dry neighbour takes it all
        const bool give = !spill && diffV > 0.0f && !(hWater <= 0.0f);
        const bool take = !spill && !(diffV > 0.0f) && diffV < 0.0f && !(bw <= 0.0f);
        const float fill_give = fminf(0.25f * hWater, 0.25f * diffV);
        const float fill_take = fminf(0.25f * bw, -0.25f * diffV);
        const float put = spill ? bw + hWater : (give ? bw + fill_give : bw + (-1.0f * fill_take));
        const float w_new = spill ? 0.0f : (give ? hWater - fill_give : hWater + fill_take);
        if (go && (spill || give || take)) {
            if (DRAIN && spill) {
                const int slot = atomicAdd(&drain_hdr[0], 1);
                if (slot < drain_hdr[1]) {
                    nz_particle p;
                    p.px = b[e].idx / res;  // getPos(idx)
                    p.pz = b[e].idx % res;
                    p.water = hWater;
                    p.pid = 64000;
                    drain_data[slot] = p;
                }
            } else {
                pool[b[e].idx] = put;
                if (b[e].idx == right_idx) carry = put;
                // the neighbour's own pass must find it; one that was acting before has its bit already
                if (MARK && pool_step_acts(put) && !pool_step_acts(bw)) {
                    const int nx = b[e].idx == right_idx ? xr : (b[e].idx == xl * res + z ? xl : x);
                    pool_mark_acting(pm, nx, b[e].idx - nx * res);
                }
            }
            hWater = w_new;
            tHeight = spill ? hLand : hLand + w_new;
        }
    }
    pool[idx] = hWater;
    if (idx == right_idx) carry = hWater;  // last column: the right-hand neighbour is the cell itself
}

// One colour pass of PoolAutomataJob (MultiThreadErosionJob.cs:264-327): walk k follows row z = 2k + zoff over
// x = xoff (+1 for odd k), step 2, calling SpreadPool wherever water stands.  The walk along a row is sequential in the
// reference; rows of one pass share no cell.  z is the fast index of the planes, so lanes = walks touch neighbouring
// addresses at every step.
//
// What one step leaves for a later step of the same walk is a single cell: the right-hand neighbour (x+1, z) of
// step x is the left-hand neighbour of step x+2; every other cell a step reads (itself, (x, z+-1), (x+1, z)) is
// written by no earlier step of the walk and by no other walk of the pass.  A step that does not act writes nothing,
// so the chain only links CONSECUTIVE ACTING steps: a walk falls apart into independent runs of acting steps, and the
// runs of all rows proceed in parallel, each in its row's order -- the same values as the one-lane-per-row walk, bit
// for bit, with res^2 / 4 threads instead of res / 2.
//
//   pool_masks_kernel   once per job: the acting-step bits of all four passes from one read of the plane
//   pool_runs_kernel    thread (walk k, mask word w): for every run that STARTS among its 32 steps, walks the run to
//                       its end (into the following words if need be); the loads of the next step are issued before
//                       the arithmetic of the current one, the carried cell travels in a register.  Water a step
//                       hands to a neighbour sets that neighbour's bit (its pass comes later; a pass only reads its own
//                       class's bits, and no step changes those: the own cell of an acting step already has its bit).
//   pool_masks_clean_kernel  between two iterations: drops the bits of steps that have stopped acting (only the words
//                       that have bits look at the plane)
// A pass cannot clear bits itself (its own class's bits are what the other threads cut their runs by, and the other
// classes' bits are being set by it), and does not have to: a step that has stopped acting stays in its run and does
// nothing there -- any SUPERSET of the acting steps cuts the walks into runs that give the row walk's values.
#ifndef NZ_POOL_MASKS_NT
#define NZ_POOL_MASKS_NT 256
#endif
template <bool LIST>
__global__ __launch_bounds__(NZ_POOL_MASKS_NT) void pool_masks_kernel(const float *__restrict__ pool, pool_masks pm, int res) {
    // a lane owns walk k of both z parities: rows z = 2k and 2k + 1 are neighbours in memory, one 8-byte load per column;
    // a workgroup's waves sit side by side in z, so every row is read in pieces of NT * 8 bytes
    const int k = blockIdx.x * NZ_POOL_MASKS_NT + threadIdx.x, w = blockIdx.y;
    const bool valid = k < pm.walks;  // (no early return: the wave's lanes count and number their entries together below)
    const int odd = k & 1, z = 2 * k;
    unsigned m0[2] = {0, 0}, m1[2] = {0, 0};  // xoff = 0: x = 64 w + odd + 2 b; xoff = 1: x = 64 w + 1 + odd + 2 b
#pragma unroll
    for (int b = 0; b < 32; b++) {
        const int xa = 64 * w + odd + 2 * b, xb = xa + 1;
        float va[2] = {0.0f, 0.0f}, vb[2] = {0.0f, 0.0f};
        if (valid && xa < res) __builtin_memcpy(va, pool + (size_t)xa * res + z, 8);
        if (valid && xb < res) __builtin_memcpy(vb, pool + (size_t)xb * res + z, 8);
#pragma unroll
        for (int e = 0; e < 2; e++) {
            if (pool_step_acts(va[e])) m0[e] |= 1u << b;
            if (pool_step_acts(vb[e])) m1[e] |= 1u << b;
        }
    }
    if (LIST) {  // steps that act, over the whole plane: what the host's choice between runs and row walks rests on
        int bits = __builtin_popcount(m0[0]) + __builtin_popcount(m0[1]) + __builtin_popcount(m1[0]) + __builtin_popcount(m1[1]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) bits += __shfl_down(bits, off);
        if ((threadIdx.x & 63) == 0 && bits) (void)__hip_atomic_fetch_add(pm.ctl + 2, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // The non-empty words as a list, one entry each.  One counter increment per WAVE (the lanes' entries are numbered by a
    // prefix sum over the wave), and none at all once the list has overflowed -- a plane with water everywhere has half a
    // million non-empty words, and as many returning atomics on one address cost the job 5 ms.
    int mine = 0, base = 0;
    if (LIST) {
        mine = (m0[0] != 0u) + (m0[1] != 0u) + (m1[0] != 0u) + (m1[1] != 0u);
        int incl = mine;  // inclusive prefix sum over the wave's lanes
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if ((int)(threadIdx.x & 63) >= off) incl += t;
        }
        const int total = __shfl(incl, 63);
        int wave_base = 0;
        if ((threadIdx.x & 63) == 63 && total > 0) {
            // beyond the capacity only the count matters, and only that it is beyond: stop counting there
            const int seen = __hip_atomic_load(pm.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            wave_base = seen > pm.cap ? pm.cap + 1 : __hip_atomic_fetch_add(pm.ctl, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        wave_base = __shfl(wave_base, 63);
        base = wave_base + incl - mine;
    }
    if (!valid) return;
#pragma unroll
    for (int zoff = 0; zoff < 2; zoff++) {
        const size_t i0 = ((size_t)(0 + zoff) * pm.words + w) * pm.walks + k, i1 = ((size_t)(2 + zoff) * pm.words + w) * pm.walks + k;
        pm.m[i0] = m0[zoff];
        pm.m[i1] = m1[zoff];
        if (LIST) {
            pm.listed[i0] = m0[zoff] != 0u;
            pm.listed[i1] = m1[zoff] != 0u;
            if (m0[zoff]) {
                if (base < pm.cap) pm.list[base] = pool_list_pack(0 + zoff, w, k);
                base++;
            }
            if (m1[zoff]) {
                if (base < pm.cap) pm.list[base] = pool_list_pack(2 + zoff, w, k);
                base++;
            }
        }
    }
}

// a mask word as the other threads of a running kernel have left it: an agent-scope load (atomic ORs land in the L2, a
// plain load may find an older copy of the line in the CU's L1)
template <bool COH>
__device__ __forceinline__ unsigned pool_mask_load(const unsigned *p) {
    return COH ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}

// one word of the clean step: drops the bits of steps that have stopped acting
template <bool COH>
__device__ __forceinline__ void pool_clean_word(const float *__restrict__ pool, const pool_masks &pm, int res, int c, int w, int k,
                                                unsigned mword) {
    if (mword == 0) return;
    const int z = 2 * k + (c & 1);
    const int x0 = (c >> 1) + (k & 1) + 64 * w;
    unsigned keep = mword;
    for (unsigned rest = mword; rest;) {  // eight cells per round trip (a word of a lake has all 32 bits set)
        int b[8];
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            b[e] = rest ? __builtin_ctz(rest) : -1;
            rest &= rest - 1;  // 0 stays 0
        }
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = b[e] >= 0 ? pool[(size_t)(x0 + 2 * b[e]) * res + z] : 1.0f;
#pragma unroll
        for (int e = 0; e < 8; e++)
            if (b[e] >= 0 && !pool_step_acts(v[e])) keep &= ~(1u << b[e]);
    }
    if (keep != mword) {
        unsigned *word = pm.m + ((size_t)c * pm.words + w) * pm.walks + k;
        if (COH) __hip_atomic_store(word, keep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *word = keep;
    }
}

constexpr int PCW = 8;  // mask words per thread of the clean kernel
__global__ __launch_bounds__(256) void pool_masks_clean_kernel(const float *__restrict__ pool, pool_masks pm, int res) {
    const int k = blockIdx.x * 256 + threadIdx.x, c = blockIdx.z;
    if (k >= pm.walks) return;
    if (pm.ctl && (pm.ctl[1] || pm.ctl[3])) return;  // the sparse kernel has run the whole job, or the passes walk whole rows
    const int w0 = blockIdx.y * PCW;
    unsigned m[PCW];
#pragma unroll
    for (int u = 0; u < PCW; u++)  // the words first, all loads in flight
        m[u] = w0 + u < pm.words ? pm.m[((size_t)c * pm.words + w0 + u) * pm.walks + k] : 0u;
#pragma unroll
    for (int u = 0; u < PCW; u++) pool_clean_word<false>(pool, pm, res, c, w0 + u, k, m[u]);
}

constexpr int PRT = 256;  // threads per workgroup: a pass is 50 tiny launches, 4x fewer workgroups dispatch faster
#ifndef NZ_POOL_PR
#define NZ_POOL_PR 8
#endif
constexpr int PR = NZ_POOL_PR;  // steps of a run whose loads are in flight together (all-wet 8192^2: 305 ms one step at a time, 212 with 8, 222 with 16: what is left is the ~1 us instruction stream of a step)
// The one-lane-per-row form of the same pass (NZ_POOL_RUNS=0): the loads of PU consecutive steps are issued together,
// ahead of the arithmetic, and the one carried cell travels in a register, so the walk pays one memory round trip per
// PU steps.  Faster than the run form only where standing water covers most of the plane.
constexpr int PU = 8;

template <bool DRAIN>
__device__ __forceinline__ void pool_row_walk(float *pool, const float *__restrict__ height, int res, int xoff, int zoff, int k,
                                              int32_t *drain_hdr, nz_particle *drain_data) {
    if (k >= res / 2) return;
    const int z = 2 * k + zoff;
    const int zu = min(z + 1, res - 1), zd = max(z - 1, 0);  // SafeIdx clamps (:585-589)
    float carry = 0.0f;
    bool have_carry = false;
    for (int x0 = xoff + ((k & 1) ? 1 : 0); x0 < res; x0 += 2 * PU) {
        float sw[PU], sh[PU], nh[PU][4], nw[PU][4];
#pragma unroll
        for (int u = 0; u < PU; u++) {
            int x = min(x0 + 2 * u, res - 1);  // steps past the end of the row load a valid cell and are skipped below
            int xr = min(x + 1, res - 1), xl = max(x - 1, 0);
            size_t c = (size_t)x * res;
            sw[u] = pool[c + z];
            sh[u] = height[c + z];
            nh[u][0] = height[c + zu];                 nw[u][0] = pool[c + zu];                  // up
            nh[u][1] = height[(size_t)xr * res + z];   nw[u][1] = pool[(size_t)xr * res + z];    // right
            nh[u][2] = height[c + zd];                 nw[u][2] = pool[c + zd];                  // down
            nh[u][3] = height[(size_t)xl * res + z];   nw[u][3] = pool[(size_t)xl * res + z];    // left
        }
#pragma unroll
        for (int u = 0; u < PU; u++) {
            const int x = x0 + 2 * u;
            if (x >= res) break;
            if (have_carry) nw[u][3] = carry;
            carry = nw[u][1];
            have_carry = true;
            if (!(sw[u] > 0.0f)) continue;
            spread_pool_step<DRAIN, false>(pool, res, x, z, zu, zd, sw[u], sh[u], nh[u], nw[u], carry, drain_hdr, drain_data,
                                           pool_masks{nullptr, 0, 0});
        }
    }
}

template <bool DRAIN>
__global__ __launch_bounds__(64) void pool_automata_pass_kernel(float *pool, const float *__restrict__ height, int res,
                                                               int xoff, int zoff, int32_t *drain_hdr,
                                                               nz_particle *drain_data) {
    pool_row_walk<DRAIN>(pool, height, res, xoff, zoff, blockIdx.x * 64 + threadIdx.x, drain_hdr, drain_data);
}
// The same walk as the second half of a dense pass: it runs only when the sparse launch found the plane under water
// (ctl[3]); the runs kernel launched before it then does nothing.  (One kernel with both forms needs 176 VGPRs instead of
// 122 and halves the runs' occupancy: 30 % wet 4096^2 3.2 -> 8.3 ms.)
template <bool DRAIN>
__global__ __launch_bounds__(64) void pool_rows_if_wet_kernel(float *pool, const float *__restrict__ height, int res, int xoff,
                                                             int zoff, int32_t *drain_hdr, nz_particle *drain_data,
                                                             const int *ctl) {
    if (ctl[1] || !ctl[3]) return;
    pool_row_walk<DRAIN>(pool, height, res, xoff, zoff, blockIdx.x * 64 + threadIdx.x, drain_hdr, drain_data);
}

// thread (walk k, mask word w) of a pass: every run that STARTS among the word's 32 steps, walked to its end
template <bool DRAIN, bool COH>
__device__ __forceinline__ void pool_walk_word(float *pool, const float *__restrict__ height, const pool_masks &pm, int res,
                                               int xoff, int zoff, int k, int w, int32_t *drain_hdr, nz_particle *drain_data) {
    const int walks = pm.walks, words = pm.words;
    const unsigned *mask = pm.m + (size_t)(2 * xoff + zoff) * words * walks;  // read-only for the whole pass
    const unsigned m0 = pool_mask_load<COH>(mask + (size_t)w * walks + k);
    if (m0 == 0) return;
    const unsigned before = w > 0 ? pool_mask_load<COH>(mask + (size_t)(w - 1) * walks + k) >> 31 : 0u;
    unsigned starts = m0 & ~((m0 << 1) | before);
    const int z = 2 * k + zoff;
    const int zu = min(z + 1, res - 1), zd = max(z - 1, 0);  // SafeIdx clamps (:585-589)
    const int xbase = xoff + (k & 1);
    while (starts) {
        const int first = __builtin_ctz(starts);
        starts &= starts - 1;
        int ww = w, bit = first;
        unsigned m = m0;
        int x = xbase + 2 * (32 * w + first);
        float carry = 0.0f;
        bool have_carry = false;  // the run's first step reads its left-hand cell from memory, the later ones carry it
        for (;;) {
            // the steps of this run inside the current mask word, PR at a time: their cells are written by no earlier
            // step of the walk except the left-hand one, which travels in `carry` -- so the loads of PR steps go out
            // together and a long run (standing water) pays one memory round trip per PR steps, not one per step
            const unsigned rest = m >> bit;
            int ones = rest == 0xffffffffu ? 32 : __builtin_ctz(~rest);  // >= 1: bit `bit` is set
            // a run that reaches the end of the word goes on in the next one: its bits are asked for now
            const bool to_end = bit + ones == 32 && ww + 1 < words;
            const unsigned mnext = to_end ? pool_mask_load<COH>(mask + (size_t)(ww + 1) * walks + k) : 0u;
            while (ones > 0) {
                const int n = min(ones, PR);
                float sw[PR], sh[PR], nh[PR][4], nw[PR][4];
#pragma unroll
                for (int u = 0; u < PR; u++) {
                    if (u < n) {
                        const int xs = x + 2 * u;
                        const int xr = min(xs + 1, res - 1), xl = max(xs - 1, 0);
                        const size_t c = (size_t)xs * res;
                        sw[u] = pool[c + z];                       sh[u] = height[c + z];
                        nh[u][0] = height[c + zu];                 nw[u][0] = pool[c + zu];                  // up
                        nh[u][1] = height[(size_t)xr * res + z];   nw[u][1] = pool[(size_t)xr * res + z];    // right
                        nh[u][2] = height[c + zd];                 nw[u][2] = pool[c + zd];                  // down
                        nh[u][3] = height[(size_t)xl * res + z];                                             // left
                        nw[u][3] = (u == 0 && !have_carry) ? pool[(size_t)xl * res + z] : 0.0f;
                    }
                }
#pragma unroll
                for (int u = 0; u < PR; u++) {
                    if (u < n) {
                        if (u > 0 || have_carry) nw[u][3] = carry;
                        carry = nw[u][1];
                        have_carry = true;
                        if (pool_step_acts(sw[u]))  // a stale bit: the step stays in its run and does nothing
                            spread_pool_step<DRAIN, true>(pool, res, x + 2 * u, z, zu, zd, sw[u], sh[u], nh[u], nw[u], carry,
                                                          drain_hdr, drain_data, pm);
                    }
                }
                x += 2 * n;
                bit += n;
                ones -= n;
            }
            if (!to_end) break;  // the run ended inside this word (or with the row)
            ww++;
            m = mnext;
            bit = 0;
            if (!(m & 1u)) break;
        }
    }
}

template <bool DRAIN>
__global__ __launch_bounds__(PRT) void pool_runs_kernel(float *pool, const float *__restrict__ height, pool_masks pm,
                                                      int res, int xoff, int zoff, int32_t *drain_hdr,
                                                      nz_particle *drain_data) {
    const int k = blockIdx.x * PRT + threadIdx.x, w = blockIdx.y;
    if (k >= pm.walks) return;
    if (pm.ctl && pm.ctl[1]) return;  // the sparse kernel has already run the whole job
    if (pm.ctl && pm.ctl[3]) return;  // a plane under water: pool_rows_if_wet_kernel, launched next, walks whole rows
    pool_walk_word<DRAIN, false>(pool, height, pm, res, xoff, zoff, k, w, drain_hdr, drain_data);
}

// ---- the sparse form of a whole job -----------------------------------------------------------------------------
// With the reference's settings almost no cell holds enough water to act (a share of 1e-6 ... 2e-4 of the plane over
// thousands of cycles): the job's `iterations` x 4 colour passes and the cleans between the iterations are
// then ~50 dependent launches that find nothing to do, ~4.3 us each.  Here ONE workgroup runs them all from the list of
// non-empty mask words the masks kernel left (and the walks extend when water reaches a word for the first time), with a
// workgroup barrier where the launches had a kernel boundary: all the plane's cells a pass writes are written by this
// workgroup, whose waves share one L1, so plain accesses see them across the barrier; mask words, which atomic ORs
// change in the L2, are read with agent-scope loads.  Same walk code, same order inside every run: same values.
//   entries <= limit           : the list form; ctl[1] = 1 tells the dense launches that may follow to do nothing
//   more, dense launches follow: leaves everything to them (ctl[1] = 0)
//   more, nothing follows      : (the host's hint was out of date) every word of every pass is scanned by this one
//                                workgroup -- slow, and correct
// `hint` (mapped host memory): the entry count as this job found it, for the host's choice at the NEXT job.
constexpr int PST = 1024;
template <bool DRAIN>
__global__ __launch_bounds__(PST) void pool_sparse_kernel(float *pool, const float *__restrict__ height, pool_masks pm, int res,
                                                        int iterations, int limit, int dense_follows, int32_t *drain_hdr,
                                                        nz_particle *drain_data, unsigned long long *hint,
                                                        unsigned long long seq) {
    // the entry count lives in LDS while the kernel runs (this workgroup is the only one that adds entries): a pass
    // that finds nothing to do costs one workgroup barrier, not a round trip to the L2
    __shared__ int s_count;
    const int tid = threadIdx.x;
    int *const gctl = pm.ctl;
    if (tid == 0) {
        const int n = __hip_atomic_load(gctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_count = n;
        // (job number, entries) in one 8-byte store to mapped host memory
        if (hint) __hip_atomic_store(hint, (seq << 32) | (unsigned)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // A plane under water (nearly) everywhere is one run per row: the one-lane-per-row walk, its loads eight steps
        // ahead, is then the faster form (165 against 212 ms all-wet at 8192^2).  Decided here, from THIS job's plane, for
        // the dense launches that follow: ctl[3] != 0 = at least seven steps in eight act, the passes walk whole rows.  (At
        // half the plane under water the runs are still short enough to win: 3.8 against 8.2 ms at 4096^2.)
        const unsigned acting = (unsigned)__hip_atomic_load(gctl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        gctl[3] = (unsigned long long)acting * 8 >= (unsigned long long)res * res * 7 ? 1 : 0;
        gctl[2] = 0;
    }
    __syncthreads();
    const int n0 = s_count;
    if (n0 > limit && dense_follows) {
        if (tid == 0) {
            gctl[1] = 0;
            gctl[0] = 0;  // the next job's masks kernel counts from zero
        }
        return;
    }
    pm.ctl = &s_count;  // pool_list_add counts here from now on
    const bool scan_from_start = n0 > limit || n0 > pm.cap;
    const int total = pm.words * pm.walks;
    for (int it = 0; it < iterations; it++) {
        if (it > 0) {  // pool_masks_clean_kernel
            __syncthreads();  // the last pass of the previous iteration is complete
            const int n = s_count;
            if (!(scan_from_start || n > pm.cap)) {
                for (int i = tid; i < n; i += PST) {
                    const unsigned e = pm.list[i];
                    const int c = e & 3, w = (e >> 2) & 1023, k = e >> 12;
                    pool_clean_word<true>(pool, pm, res, c, w, k, pool_mask_load<true>(pm.m + ((size_t)c * pm.words + w) * pm.walks + k));
                }
            } else {
                for (int c = 0; c < 4; c++)
                    for (int i = tid; i < total; i += PST)
                        pool_clean_word<true>(pool, pm, res, c, i / pm.walks, i % pm.walks, pool_mask_load<true>(pm.m + (size_t)c * total + i));
            }
        }
        for (int xoff = 0; xoff < 2; xoff++)
            for (int zoff = 0; zoff < 2; zoff++) {
                __syncthreads();  // the previous pass (or the clean) is complete, its stores visible to the workgroup
                // entries added while this pass runs belong to other classes: whichever of them a thread already sees, it skips
                const int n = s_count;
                const int cls = 2 * xoff + zoff;
                if (!(scan_from_start || n > pm.cap)) {
                    for (int i = tid; i < n; i += PST) {
                        const unsigned e = pm.list[i];
                        if ((int)(e & 3) != cls) continue;
                        pool_walk_word<DRAIN, true>(pool, height, pm, res, xoff, zoff, (int)(e >> 12), (int)((e >> 2) & 1023), drain_hdr,
                                                    drain_data);
                    }
                } else {
                    for (int i = tid; i < total; i += PST)
                        pool_walk_word<DRAIN, true>(pool, height, pm, res, xoff, zoff, i % pm.walks, i / pm.walks, drain_hdr, drain_data);
                }
            }
    }
    __syncthreads();
    if (tid == 0) {
        gctl[1] = 1;
        gctl[0] = 0;  // the next job's masks kernel counts from zero
    }
}

// ---- GetMapRangeJob (Filter/NormalizeJob.cs:17-55): {min, max, max - min} of a plane, folded from the two limits the
// caller passes.  The reference folds sequentially with math.min / math.max (NaN elements are skipped; on a tie the
// LATER operand stays).  Values that compare equal have equal bits except the two zeros, so the fold is a plain
// parallel min / max plus the position of the last +0 and the last -0: when the minimum (maximum) is zero, its sign is
// that of the last zero the sequential fold would have met (the limit itself counts as position -1).
struct map_range_part {
    float mn, mx;
    long long last_pz, last_nz;  // -2: none
};
__device__ __forceinline__ float umin(float x, float y) { return (y != y) || x < y ? x : y; }  // Unity.Mathematics math.min
__device__ __forceinline__ float umax(float x, float y) { return (y != y) || x > y ? x : y; }
__device__ __forceinline__ void map_range_take(map_range_part &p, float v, long long i) {
    p.mn = umin(p.mn, v);
    p.mx = umax(p.mx, v);
    if (v == 0.0f) {
        if (__builtin_signbit(v)) p.last_nz = i; else p.last_pz = i;  // i ascends within a thread
    }
}
// The streaming form of the same fold, for the kernel that reads the plane: v_min_f32 / v_max_f32 (IEEE minNum / maxNum: a NaN
// operand is skipped, like math.min / math.max) -- the sign of a zero result is whatever the instruction picks, which does not
// matter: a zero minimum (maximum) takes its sign from the LAST zero of the plane (map_range_finish), kept here as ONE key per
// lane, position << 1 | sign, overwritten in ascending order.  7 VALU instructions per cell instead of 28.
struct map_range_lane {
    float mn, mx;
    unsigned long long key;  // 0: no zero met
};
__device__ __forceinline__ void map_range_take4(map_range_lane &p, const float4 &t, unsigned long long b) {
    p.mn = __builtin_fminf(__builtin_fminf(p.mn, t.x), __builtin_fminf(t.y, __builtin_fminf(t.z, t.w)));
    p.mx = __builtin_fmaxf(__builtin_fmaxf(p.mx, t.x), __builtin_fmaxf(t.y, __builtin_fmaxf(t.z, t.w)));
    // a lane's four cells are consecutive: the last zero among them
    const unsigned bx = __float_as_uint(t.x), by = __float_as_uint(t.y), bz = __float_as_uint(t.z), bw = __float_as_uint(t.w);
    if (((bx << 1) == 0) | ((by << 1) == 0) | ((bz << 1) == 0) | ((bw << 1) == 0)) {   // rare: a zero among the four
        unsigned long long k = p.key;
        if ((bx << 1) == 0) k = ((b + 1) << 1) | (bx >> 31);
        if ((by << 1) == 0) k = ((b + 2) << 1) | (by >> 31);
        if ((bz << 1) == 0) k = ((b + 3) << 1) | (bz >> 31);
        if ((bw << 1) == 0) k = ((b + 4) << 1) | (bw >> 31);
        p.key = k;   // (positions are stored + 1, so that 0 means none)
    }
}
__device__ __forceinline__ map_range_part map_range_from_lane(const map_range_lane &l) {
    map_range_part p{l.mn, l.mx, -2, -2};
    if (l.key) {
        const long long pos = (long long)(l.key >> 1) - 1;
        if (l.key & 1) p.last_nz = pos; else p.last_pz = pos;
    }
    return p;
}
__device__ __forceinline__ void map_range_merge(map_range_part &a, const map_range_part &b) {
    a.mn = umin(a.mn, b.mn);
    a.mx = umax(a.mx, b.mx);
    a.last_pz = a.last_pz > b.last_pz ? a.last_pz : b.last_pz;
    a.last_nz = a.last_nz > b.last_nz ? a.last_nz : b.last_nz;
}
__device__ __forceinline__ map_range_part map_range_block(map_range_part p, map_range_part *s_part) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        map_range_part q;
        q.mn = __shfl_down(p.mn, off);
        q.mx = __shfl_down(p.mx, off);
        q.last_pz = __shfl_down(p.last_pz, off);
        q.last_nz = __shfl_down(p.last_nz, off);
        map_range_merge(p, q);
    }
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = p;
    __syncthreads();
    if (threadIdx.x == 0)
        for (int w = 1; w < CT / 64; w++) map_range_merge(p, s_part[w]);
    return p;  // thread 0 holds the block's
}

__device__ __forceinline__ void map_range_finish(const map_range_part &p, float lim_min, float lim_max, float *__restrict__ res) {
    float mn = lim_min, mx = lim_max;
    if (!(p.mn == __builtin_inff() && p.mx == -__builtin_inff())) {  // some element was not NaN
        mn = umin(lim_min, p.mn);
        mx = umax(lim_max, p.mx);
        const bool last_is_neg = p.last_nz > p.last_pz;  // both -2: no zero in the plane, the limit's own zero stays
        if (mn == 0.0f && (p.last_pz >= 0 || p.last_nz >= 0)) mn = last_is_neg ? -0.0f : 0.0f;
        if (mx == 0.0f && (p.last_pz >= 0 || p.last_nz >= 0)) mx = last_is_neg ? -0.0f : 0.0f;
    }
    res[0] = mn;
    res[1] = mx;
    res[2] = mx - mn;
}

// Two launches: every workgroup folds its share into a partial, then one workgroup folds the partials.  (One launch
// with the last workgroup to arrive doing the second fold is slower: 2048 arrivals on one counter serialise, 0.042 ms;
// with 512 workgroups 0.028 ms against 0.024 ms for this form.)
__global__ __launch_bounds__(CT) void map_range_partial_kernel(const float *__restrict__ map, size_t n, int vec,
                                                              map_range_part *__restrict__ parts) {
    __shared__ map_range_part s_part[CT / 64];
    map_range_part p{__builtin_inff(), -__builtin_inff(), -2, -2};
    const size_t stride = (size_t)gridDim.x * CT;
    if (vec) {
        const size_t n4 = n / 4;
#ifndef NZ_MAP_RANGE_U
#define NZ_MAP_RANGE_U 4
#endif
        constexpr int U = NZ_MAP_RANGE_U;  // 16-byte loads in flight per thread (rocprofv3, 4096^2: 4 -> 13.2 us, 8 -> 14.3; round 5's 28-instruction fold: 16.7)
        map_range_lane l{__builtin_inff(), -__builtin_inff(), 0ull};
        for (size_t i0 = (size_t)blockIdx.x * CT + threadIdx.x; i0 < n4; i0 += U * stride) {
            float4 t[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const size_t i = i0 + u * stride;
                typedef float f4v __attribute__((ext_vector_type(4)));
                if (i < n4) {  // read once, by one workgroup: non-temporal (no line of the plane is worth keeping in the L2)
                    const f4v q = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(map) + i);
                    t[u] = make_float4(q.x, q.y, q.z, q.w);
                } else {
                    t[u] = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++)  // ascending indices: the last zero a lane meets is its highest
                map_range_take4(l, t[u], (unsigned long long)(4 * (i0 + u * stride)));
        }
        p = map_range_from_lane(l);
        for (size_t i = n4 * 4 + (size_t)blockIdx.x * CT + threadIdx.x; i < n; i += stride) map_range_take(p, map[i], (long long)i);
    } else {
        for (size_t i = (size_t)blockIdx.x * CT + threadIdx.x; i < n; i += stride) map_range_take(p, map[i], (long long)i);
    }
    p = map_range_block(p, s_part);
    if (threadIdx.x == 0) parts[blockIdx.x] = p;
}

__global__ __launch_bounds__(CT) void map_range_final_kernel(const map_range_part *__restrict__ parts, int nparts,
                                                            float lim_min, float lim_max, float *__restrict__ res) {
    __shared__ map_range_part s_part[CT / 64];
    map_range_part p{__builtin_inff(), -__builtin_inff(), -2, -2};
    for (int i = threadIdx.x; i < nparts; i += CT) map_range_merge(p, parts[i]);
    p = map_range_block(p, s_part);
    if (threadIdx.x == 0) map_range_finish(p, lim_min, lim_max, res);
}

// NormalizeMap with its args in device memory ({min, max, range}: what GetMapRangeJob leaves), NormalizeJob.cs:57-92
__device__ __forceinline__ float normalize_cell(float v, float nmin, float nrange) {
    if (nrange < 1e-12f) v = 0.0f;
    return (v - nmin) / nrange;
}
__global__ __launch_bounds__(CT) void normalize_args_kernel(float *__restrict__ data, size_t n, int vec,
                                                           const float *__restrict__ args) {
    const float nmin = args[0], nrange = args[2];
    const size_t i = (size_t)blockIdx.x * CT + threadIdx.x;
    if (vec && 4 * i + 4 <= n) {
        float4 t = reinterpret_cast<float4 *>(data)[i];
        t.x = normalize_cell(t.x, nmin, nrange);
        t.y = normalize_cell(t.y, nmin, nrange);
        t.z = normalize_cell(t.z, nmin, nrange);
        t.w = normalize_cell(t.w, nmin, nrange);
        reinterpret_cast<float4 *>(data)[i] = t;
    } else {
        for (size_t j = 4 * i; j < n && j < 4 * i + 4; j++) data[j] = normalize_cell(data[j], nmin, nrange);
    }
}

// the sharded GetMapRangeJob (nz_comm_allgather_range): the ranks' {min, max, range} triples as one array of minima and
// one of maxima, which the same fold then walks in rank order; and the triple it leaves
__global__ void range_split_kernel(const float *__restrict__ triples, int n, float *__restrict__ mins, float *__restrict__ maxs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        mins[i] = triples[3 * i];
        maxs[i] = triples[3 * i + 1];
    }
}
__global__ void range_compose_kernel(const float *__restrict__ lo, const float *__restrict__ hi, float *__restrict__ res) {
    const float mn = lo[0], mx = hi[1];
    res[0] = mn;
    res[1] = mx;
    res[2] = mx - mn;  // `max_ - min_` of the job, IEEE fp32
}

// CropJob (Filter/Sample/CropJob.cs:34-41): out(x,z) = in(clamp(x + Offset), clamp(z + Offset)); the reference never
// sets Offset, so it is 0 (top-left crop).  One thread per output cell; rows of different pitch on either side.
__global__ __launch_bounds__(CT) void crop_kernel(const float *__restrict__ in, int in_res, float *__restrict__ out,
                                                 int out_res, int offset) {
    int x = blockIdx.x * CT + threadIdx.x;
    int z = blockIdx.y;
    if (x >= out_res) return;
    int sx = min(max(x + offset, 0), in_res - 1), sz = min(max(z + offset, 0), in_res - 1);
    out[(size_t)z * out_res + x] = in[(size_t)sz * in_res + sx];
}

// ThermalErosionFilter (Filter/Kernel/Blur/ThermalErosionFilter.cs:21-147): one launch per phase; a phase
// relaxes disjoint 2x2 blocks in place (rectify :84-99 applied to the six pairs in the order xy, xz, xw, yz,
// yw, zw :74-81), so the result does not depend on the execution order within a phase.
__device__ __forceinline__ void thermal_rectify(float &a, float &b, float maxDiff, float increment) {
    float diff = fabsf(a - b);
    if (diff > maxDiff) {
        float excess = diff - maxDiff;
        if (a > b) {
            b += increment * excess;
            a -= increment * excess;
        } else {
            a += increment * excess;
            b -= increment * excess;
        }
    }
}

__global__ __launch_bounds__(CT) void thermal_phase_kernel(float *data, int resolution, int flip, float maxDiff,
                                                          float increment) {
    int job = blockIdx.y;  // IJobFor index, (resolution / 2) - 1 jobs
    int z = (job + 1) * 2 - (flip > 1 ? 1 : 0);
    int offset = 1 + ((flip % 2 != 0) ? 1 : 0);
    int x = offset + 2 * (blockIdx.x * CT + threadIdx.x);
    if (x >= resolution - 1) return;
    size_t i0 = (size_t)z * resolution + x, i2 = (size_t)(z + 1) * resolution + x;
    float vx = data[i0], vy = data[i0 + 1], vz = data[i2], vw = data[i2 + 1];
    thermal_rectify(vx, vy, maxDiff, increment);
    thermal_rectify(vx, vz, maxDiff, increment);
    thermal_rectify(vx, vw, maxDiff, increment);
    thermal_rectify(vy, vz, maxDiff, increment);
    thermal_rectify(vy, vw, maxDiff, increment);
    thermal_rectify(vz, vw, maxDiff, increment);
    data[i0] = vx; data[i0 + 1] = vy; data[i2] = vz; data[i2 + 1] = vw;
}

// Phases 0 and 1 relax blocks of the SAME row pairs (z, z + 1), z even, at x offsets 1 and 2; phases 2 and 3 those of the
// odd pairs.  A row pair depends on no other row pair within such a pair of phases, so one workgroup takes a row pair
// through both phases in LDS: one read and one write of the plane instead of two, in place.
__global__ __launch_bounds__(1024) void thermal_pair_kernel(float *data, int resolution, int zodd, float maxDiff,
                                                         float increment, int vec) {
    extern __shared__ __attribute__((aligned(16))) float s_rows[];  // [2][resolution]
    const int z = (blockIdx.x + 1) * 2 - zodd, nt = blockDim.x;
    float *g0 = data + (size_t)z * resolution;
    float *s0 = s_rows, *s1 = s_rows + resolution;
    // what this thread loaded stays in registers: a quad the relaxation left as it was is not written back (on terrain at
    // rest -- most of a long run's plane -- that is nearly every quad, and the pass moves 4 bytes per cell instead of 8)
    constexpr int QMAX = 8;  // float4s per thread: resolution / 2 / nt <= 8 for the thread counts of nz_launch_thermal_pair
    float4 orig[QMAX];
    if (vec) {  // resolution % 4 == 0 and the plane 16-byte aligned: both rows are
#pragma unroll
        for (int k = 0; k < QMAX; k++) {  // 2 rows x resolution / 4 float4s, contiguous in memory and in LDS
            const int i = threadIdx.x + k * nt;
            if (i < resolution / 2) {
                orig[k] = reinterpret_cast<const float4 *>(g0)[i];
                reinterpret_cast<float4 *>(s_rows)[i] = orig[k];
            }
        }
    } else {
        for (int i = threadIdx.x; i < 2 * resolution; i += nt) s_rows[i] = g0[i];
    }
    __syncthreads();
#pragma unroll
    for (int offset = 1; offset <= 2; offset++) {
        for (int x = offset + 2 * (int)threadIdx.x; x < resolution - 1; x += 2 * nt) {
            float vx = s0[x], vy = s0[x + 1], vz = s1[x], vw = s1[x + 1];
            thermal_rectify(vx, vy, maxDiff, increment);
            thermal_rectify(vx, vz, maxDiff, increment);
            thermal_rectify(vx, vw, maxDiff, increment);
            thermal_rectify(vy, vz, maxDiff, increment);
            thermal_rectify(vy, vw, maxDiff, increment);
            thermal_rectify(vz, vw, maxDiff, increment);
            s0[x] = vx; s0[x + 1] = vy; s1[x] = vz; s1[x + 1] = vw;
        }
        __syncthreads();
    }
    if (vec) {
#pragma unroll
        for (int k = 0; k < QMAX; k++) {
            const int i = threadIdx.x + k * nt;
            if (i < resolution / 2) {
                const float4 v = reinterpret_cast<const float4 *>(s_rows)[i];
                const unsigned diff = (__float_as_uint(v.x) ^ __float_as_uint(orig[k].x)) | (__float_as_uint(v.y) ^ __float_as_uint(orig[k].y)) |
                                      (__float_as_uint(v.z) ^ __float_as_uint(orig[k].z)) | (__float_as_uint(v.w) ^ __float_as_uint(orig[k].w));
                if (diff) reinterpret_cast<float4 *>(g0)[i] = v;
            }
        }
    } else {
        for (int i = threadIdx.x; i < 2 * resolution; i += nt) g0[i] = s_rows[i];
    }
}

unsigned blocks_for(size_t n) { return (unsigned)((n + (size_t)CT * 4 - 1) / ((size_t)CT * 4)); }

}  // namespace

int32_t nz_launch_constant(hipStream_t s, int op, float *data, size_t n, float c) {
    if (n == 0) return NZ_OK;
    int aligned = (reinterpret_cast<uintptr_t>(data) & 15) == 0;
    if (op == 0) NZ_LAUNCH(constant_kernel<0>, dim3(blocks_for(n)), dim3(CT), 0, s, data, n, c, aligned);
    else if (op == 1) NZ_LAUNCH(constant_kernel<1>, dim3(blocks_for(n)), dim3(CT), 0, s, data, n, c, aligned);
    else {
        nz_set_error("unknown ConstantOperationType %d", op);
        return NZ_ERR_INVALID;
    }
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

int32_t nz_launch_reduce(hipStream_t s, int op, float *l, const float *r, size_t n) {
    if (n == 0) return NZ_OK;
    int aligned = ((reinterpret_cast<uintptr_t>(l) | reinterpret_cast<uintptr_t>(r)) & 15) == 0;
    dim3 grid(blocks_for(n)), block(CT);
    switch (op) {
        case 0: NZ_LAUNCH(reduce_kernel<0>, grid, block, 0, s, l, r, n, aligned); break;
        case 1: NZ_LAUNCH(reduce_kernel<1>, grid, block, 0, s, l, r, n, aligned); break;
        case 2: NZ_LAUNCH(reduce_kernel<2>, grid, block, 0, s, l, r, n, aligned); break;
        case 3: NZ_LAUNCH(reduce_kernel<3>, grid, block, 0, s, l, r, n, aligned); break;
        case 4: NZ_LAUNCH(reduce_kernel<4>, grid, block, 0, s, l, r, n, aligned); break;
        default: nz_set_error("unknown ReductionType %d", op); return NZ_ERR_INVALID;
    }
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

int32_t nz_launch_curve(hipStream_t s, float *data, size_t n, const float *curve, int curveSize) {
    if (n == 0) return NZ_OK;
    int aligned = (reinterpret_cast<uintptr_t>(data) & 15) == 0;
    NZ_LAUNCH(curve_kernel, dim3(blocks_for(n)), dim3(CT), (size_t)curveSize * sizeof(float), s, data, n, curve,
                       curveSize, aligned);
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

int32_t nz_launch_flow_from_track(hipStream_t s, float *pool, float *flow, float *track, size_t n, float flowLossRate,
                                  float evaporation) {
    if (n == 0) return NZ_OK;
    const int aligned = ((reinterpret_cast<uintptr_t>(pool) | reinterpret_cast<uintptr_t>(flow) | reinterpret_cast<uintptr_t>(track)) & 15) == 0;
    hipLaunchKernelGGL(flow_from_track_kernel, dim3((unsigned)(((n + 3) / 4 + CT - 1) / CT)), dim3(CT), 0, s, pool, flow, track, n,
                       flowLossRate, evaporation, aligned);
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}


// scratch of a job in its parallel-runs form: the mask words of the four classes, and for the sparse form the `listed`
// words and the list itself
constexpr int NZ_POOL_LIST_CAP = 8192;
static pool_masks pool_masks_of(int res, unsigned *scratch, int *ctl, bool with_list) {
    pool_masks pm;
    pm.m = scratch;
    pm.words = ((res + 1) / 2 + 31) / 32;
    pm.walks = res / 2;
    pm.ctl = ctl;
    if (with_list) {
        const size_t n = (size_t)4 * pm.words * pm.walks;
        pm.listed = scratch + n;
        pm.list = scratch + 2 * n;
        pm.cap = NZ_POOL_LIST_CAP;
    }
    return pm;
}
size_t nz_pool_automata_mask_words(int res) {
    const size_t n = (size_t)4 * (((res + 1) / 2 + 31) / 32) * (size_t)(res / 2);
    return 2 * n + NZ_POOL_LIST_CAP;
}

// The acting-step bits of a job's four passes, from the plane as the job finds it (with_list: and the non-empty words
// as a list, counted in ctl[0])
int32_t nz_launch_pool_automata_masks(hipStream_t s, const float *pool, int res, unsigned *mask, int *ctl, int with_list) {
    if (res / 2 <= 0) return NZ_OK;
    const pool_masks pm = pool_masks_of(res, mask, ctl, with_list != 0);
    const dim3 grid((unsigned)((pm.walks + NZ_POOL_MASKS_NT - 1) / NZ_POOL_MASKS_NT), (unsigned)pm.words);
    if (with_list) hipLaunchKernelGGL(pool_masks_kernel<true>, grid, dim3(NZ_POOL_MASKS_NT), 0, s, pool, pm, res);
    else hipLaunchKernelGGL(pool_masks_kernel<false>, grid, dim3(NZ_POOL_MASKS_NT), 0, s, pool, pm, res);
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

// The whole job from the list (pool_sparse_kernel): one launch of one workgroup
int32_t nz_launch_pool_automata_sparse(hipStream_t s, float *pool, const float *height, int res, int iterations, int limit,
                                       int dense_follows, int32_t *drain_hdr, nz_particle *drain_data, unsigned *mask, int *ctl,
                                       unsigned long long *hint_dev, unsigned long long seq) {
    if (res / 2 <= 0) return NZ_OK;
    const pool_masks pm = pool_masks_of(res, mask, ctl, true);
    if (drain_hdr)
        hipLaunchKernelGGL(pool_sparse_kernel<true>, dim3(1), dim3(PST), 0, s, pool, height, pm, res, iterations, limit,
                           dense_follows, drain_hdr, drain_data, hint_dev, seq);
    else
        hipLaunchKernelGGL(pool_sparse_kernel<false>, dim3(1), dim3(PST), 0, s, pool, height, pm, res, iterations, limit,
                           dense_follows, drain_hdr, drain_data, hint_dev, seq);
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

int32_t nz_launch_pool_automata_clean(hipStream_t s, const float *pool, int res, unsigned *mask, int *ctl) {
    if (res / 2 <= 0) return NZ_OK;
    const pool_masks pm = pool_masks_of(res, mask, ctl, false);
    hipLaunchKernelGGL(pool_masks_clean_kernel, dim3((unsigned)((pm.walks + 255) / 256), (unsigned)((pm.words + PCW - 1) / PCW), 4),
                       dim3(256), 0, s, pool, pm, res);
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

// `mask` = the bits nz_launch_pool_automata_masks prepared (nz_pool_automata_mask_words(res) words, kept up to date by
// the passes themselves): the pass runs as parallel runs; NULL: one lane per row.  ctl (nullable): ctl[1] != 0 = the
// sparse kernel has already run the job, the pass does nothing.
int32_t nz_launch_pool_automata_pass(hipStream_t s, float *pool, const float *height, int res, int xoff, int zoff,
                                     int32_t *drain_hdr, nz_particle *drain_data, unsigned *mask, int *ctl) {
    int jobs = res / 2;
    if (jobs <= 0) return NZ_OK;
    dim3 grid((unsigned)((jobs + 63) / 64));
    if (mask) {
        const pool_masks pm = pool_masks_of(res, mask, ctl, false);
        dim3 rgrid((unsigned)((jobs + PRT - 1) / PRT), (unsigned)pm.words);
        if (drain_hdr)
            hipLaunchKernelGGL(pool_runs_kernel<true>, rgrid, dim3(PRT), 0, s, pool, height, pm, res, xoff, zoff, drain_hdr,
                               drain_data);
        else
            hipLaunchKernelGGL(pool_runs_kernel<false>, rgrid, dim3(PRT), 0, s, pool, height, pm, res, xoff, zoff, drain_hdr,
                               drain_data);
        if (ctl) {  // the row-walk half of the pass: does something only on a plane under water
            if (drain_hdr)
                hipLaunchKernelGGL(pool_rows_if_wet_kernel<true>, grid, dim3(64), 0, s, pool, height, res, xoff, zoff, drain_hdr,
                                   drain_data, ctl);
            else
                hipLaunchKernelGGL(pool_rows_if_wet_kernel<false>, grid, dim3(64), 0, s, pool, height, res, xoff, zoff, drain_hdr,
                                   drain_data, ctl);
        }
    } else if (drain_hdr) {
        hipLaunchKernelGGL(pool_automata_pass_kernel<true>, grid, dim3(64), 0, s, pool, height, res, xoff, zoff, drain_hdr,
                           drain_data);
    } else {
        hipLaunchKernelGGL(pool_automata_pass_kernel<false>, grid, dim3(64), 0, s, pool, height, res, xoff, zoff, drain_hdr,
                           drain_data);
    }
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

constexpr int MAP_RANGE_BLOCKS = 2048;
size_t nz_map_range_scratch_floats() { return (size_t)MAP_RANGE_BLOCKS * sizeof(map_range_part) / sizeof(float); }

int32_t nz_launch_map_range(hipStream_t s, const float *map, size_t n, float lim_min, float lim_max, float *res, void *scratch) {
    map_range_part *parts = reinterpret_cast<map_range_part *>(scratch);
    const int vec = (reinterpret_cast<uintptr_t>(map) & 15) == 0;
    size_t want = (n / 4 + CT - 1) / CT;
    const int blocks = (int)(want < 1 ? 1 : (want > (size_t)MAP_RANGE_BLOCKS ? (size_t)MAP_RANGE_BLOCKS : want));
    hipLaunchKernelGGL(map_range_partial_kernel, dim3((unsigned)blocks), dim3(CT), 0, s, map, n, vec, parts);
    hipLaunchKernelGGL(map_range_final_kernel, dim3(1), dim3(CT), 0, s, parts, blocks, lim_min, lim_max, res);
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

int32_t nz_launch_normalize_args(hipStream_t s, float *data, size_t n, const float *args) {
    if (n == 0) return NZ_OK;
    const int vec = (reinterpret_cast<uintptr_t>(data) & 15) == 0;
    NZ_LAUNCH(normalize_args_kernel, dim3((unsigned)(((n + 3) / 4 + CT - 1) / CT)), dim3(CT), 0, s, data, n, vec, args);
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

int32_t nz_launch_range_split(hipStream_t s, const float *triples, int n, float *mins, float *maxs) {
    if (n <= 0) return NZ_OK;
    hipLaunchKernelGGL(range_split_kernel, dim3((n + 63) / 64), dim3(64), 0, s, triples, n, mins, maxs);
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

int32_t nz_launch_range_compose(hipStream_t s, const float *lo, const float *hi, float *res) {
    hipLaunchKernelGGL(range_compose_kernel, dim3(1), dim3(1), 0, s, lo, hi, res);
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

int32_t nz_launch_crop(hipStream_t s, const float *in, int in_res, float *out, int out_res) {
    if (out_res <= 0) return NZ_OK;
    dim3 grid((out_res + CT - 1) / CT, out_res);
    NZ_LAUNCH(crop_kernel, grid, dim3(CT), 0, s, in, in_res, out, out_res, 0);
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

// phases (0, 1) for zodd = 0, (2, 3) for zodd = 1; false if a row pair does not fit the LDS (resolution > 16384)
bool nz_thermal_pair_fits(int resolution) { return (size_t)resolution * 8 <= 128 * 1024; }
int32_t nz_launch_thermal_pair(hipStream_t s, float *data, int resolution, int zodd, float maxDiff, float increment) {
    int jobs = resolution / 2 - 1;
    if (jobs <= 0 || resolution < 3) return NZ_OK;
    const int vec = resolution % 4 == 0 && (reinterpret_cast<uintptr_t>(data) & 15) == 0;
    const size_t lds = (size_t)resolution * 8;
    if (lds > 64 * 1024)
        NZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(thermal_pair_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // a row pair of 2 x resolution cells per workgroup: enough threads to keep the loads of a 64 KB pair in flight
    const int nt = resolution >= 8192 ? 1024 : (resolution >= 2048 ? 512 : 256);
    NZ_LAUNCH(thermal_pair_kernel, dim3((unsigned)jobs), dim3(nt), lds, s, data, resolution, zodd, maxDiff, increment, vec);
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

int32_t nz_launch_thermal_phase(hipStream_t s, float *data, int resolution, int flip, float maxDiff, float increment) {
    int jobs = resolution / 2 - 1;
    if (jobs <= 0) return NZ_OK;
    int per_row = (resolution - 1) / 2;  // upper bound on the blocks of a row
    if (per_row <= 0) return NZ_OK;
    dim3 grid((per_row + CT - 1) / CT, jobs);
    NZ_LAUNCH(thermal_phase_kernel, grid, dim3(CT), 0, s, data, resolution, flip, maxDiff, increment);
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}
