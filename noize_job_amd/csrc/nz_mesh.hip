// nz_mesh.hip -- heightmap -> interleaved vertex stream + uint32 triangle indices (gfx950).
//
// Replaces HeightMapMeshJob<{Overshoot,}SquareGridHeightMap, PositionStream32>
// (Mesh/Job/HeightMapMeshJob.cs:8-52, Mesh/Generators/OvershootSquareGridHeightMap.cs:12-103,
// Mesh/Generators/SquareGridHeightMap.cs:12-106, Mesh/Streams/PositionStream.cs:75-134,
// Mesh/Streams/Triangle.cs:19-27).
//
// Vertex record = {float3 position; float3 normal; float4 tangent; float2 texCoord0} = 48 B.  A
// workgroup builds the records of 256 consecutive vertices in LDS (12 KB) and streams them out as a
// flat float4 array, so every store instruction writes 1 KB contiguous per wave instead of 16-byte
// pieces at a 48-byte stride.  The index buffer has a closed form (vi = (R+1) z + x,
// ti = 2R(z-1) + 2(x-1)), so it is written as a flat array, one uint4 (16 B) per lane, fully
// coalesced, without reading anything.  Integer output is bit-exact by construction.
#include "nz_internal.hpp"

namespace {

constexpr int CT = 256;

struct mesh_params {
    int res, in_res, off, type;
    float height, tile_size, normal_strength;
    // batched launch: mesh blockIdx.y reads the height plane hstride floats on and writes vstride float4s /
    // istride indices on
    size_t hstride, vstride, istride;
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// getIdx: Overshoot :54-59 / Square :59-64
__device__ __forceinline__ float mesh_h(const mesh_params &g, const float *__restrict__ heights, int x, int z) {
    if (g.type == NZ_MESH_OVERSHOOT_SQUARE_GRID) {
        x = clampi(x, 0 - g.off, g.res + g.off);
        z = clampi(z, 0 - g.off, g.res + g.off);
    } else {
        x = clampi(x, 0, g.res + 1);
        z = clampi(z, 0, g.res + 1);
    }
    return heights[((z + g.off) * g.in_res) + x + g.off];
}

// write-once output streams: non-temporal 16-byte stores
template <typename T>
__device__ __forceinline__ void nt_store(T *p, const T &v) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    static_assert(sizeof(T) == 16, "16-byte records");
    __builtin_nontemporal_store(*reinterpret_cast<const v4f *>(&v), reinterpret_cast<v4f *>(p));
}

__device__ __forceinline__ float interpolate_edge(float a, float b) { return a - (b - a); }

__global__ __launch_bounds__(CT) void mesh_vertex_kernel(float4 *__restrict__ vtx, const float *__restrict__ heights,
                                                        mesh_params g) {
    __shared__ float4 s_rec[CT * 3];
    heights += blockIdx.y * g.hstride;
    vtx += blockIdx.y * g.vstride;
    const int R = g.res;
    const size_t base = (size_t)blockIdx.x * CT;
    size_t vi = base + threadIdx.x;
    const size_t nv = (size_t)(R + 1) * (R + 1);
    if (vi >= nv) vi = nv - 1;  // lanes past the end rebuild the last vertex; their records are not stored
    int z = (int)(vi / (R + 1));
    int x = (int)(vi - (size_t)z * (R + 1));
    // Execute(): Overshoot :77-102 / Square :84-105
    float px = x == 0 ? -(0.5f * g.tile_size / (float)R) : (float)x * g.tile_size / (float)R - 0.5f;
    float pz = (float)z * g.tile_size / (float)R - 0.5f;
    // SetVertexValues(): Overshoot :62-75 / Square :67-82
    float t = mesh_h(g, heights, x, z);
    float py = t * g.height;
    float l, r, u, d;
    if (g.type == NZ_MESH_OVERSHOOT_SQUARE_GRID) {
        l = mesh_h(g, heights, x - 1, z);
        r = mesh_h(g, heights, x + 1, z);
        u = mesh_h(g, heights, x, z - 1);
        d = mesh_h(g, heights, x, z + 1);
    } else {
        l = x > 0 ? mesh_h(g, heights, x - 1, z) : interpolate_edge(t, mesh_h(g, heights, x + 1, z));
        r = x < R - 1 ? mesh_h(g, heights, x + 1, z) : interpolate_edge(t, mesh_h(g, heights, x - 1, z));
        u = z > 0 ? mesh_h(g, heights, x, z - 1) : interpolate_edge(mesh_h(g, heights, x, z + 1), t);
        d = z < R - 1 ? mesh_h(g, heights, x, z + 1) : interpolate_edge(mesh_h(g, heights, x, z - 1), t);
    }
    float t1x = 4.0f, t1y = (r - l) / 2.0f, t1z = 0.0f;
    float t2x = 0.0f, t2y = (u - d) / 2.0f, t2z = 4.0f;
    float tgx = t2y * t1z - t2z * t1y;  // math.cross(t2, t1)
    float tgy = t2z * t1x - t2x * t1z;
    float tgz = t2x * t1y - t2y * t1x;
    float nx = (l - r) / 2.0f * g.normal_strength;
    float ny = 2.0f / g.height;
    float nz = (u - d) / 2.0f * g.normal_strength;
    float rs = 1.0f / sqrtf(nx * nx + ny * ny + nz * nz);  // math.normalize = rsqrt(dot) * v
    float uvx, uvy;
    if (g.type == NZ_MESH_OVERSHOOT_SQUARE_GRID) {
        uvx = ((float)x) / (((float)R) - 0.5f);
        uvy = ((float)z) / (((float)R) - 0.5f);
    } else {
        uvx = ((float)x) / ((float)R + 1.0f);
        uvy = ((float)z) / ((float)R + 1.0f);
    }
    float4 *o = s_rec + threadIdx.x * 3;  // 12-dword stride: the 16-byte LDS stores of 16 lanes hit distinct banks
    o[0] = make_float4(px, py, pz, rs * nx);
    o[1] = make_float4(rs * ny, rs * nz, tgx, tgy);
    o[2] = make_float4(tgz, 0.0f /* tangent.w of `new Vertex()` */, uvx, uvy);
    __syncthreads();
    const size_t left = nv - base;
    const int nrec = (int)(left < (size_t)CT ? left : (size_t)CT) * 3;
    float4 *out = vtx + base * 3;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        int j = k * CT + threadIdx.x;
        if (j < nrec) nt_store(out + j, s_rec[j]);
    }
}

// MeshJob<SharedSquareGridPosition, PositionStream32> (Mesh/Generators/SharedSquareGridPosition.cs:20-50): the flat
// unit-square grid.  The reference reuses one Vertex along a row, so column 0 keeps position.x = -0.5 (a literal)
// and texCoord0.x = 0 (never assigned).
__global__ __launch_bounds__(CT) void mesh_planar_vertex_kernel(float4 *__restrict__ vtx, int R) {
    size_t vi = (size_t)blockIdx.x * CT + threadIdx.x;
    size_t nv = (size_t)(R + 1) * (R + 1);
    if (vi >= nv) return;
    int z = (int)(vi / (R + 1));
    int x = (int)(vi - (size_t)z * (R + 1));
    float px = x == 0 ? -0.5f : (float)x / (float)R - 0.5f;
    float pz = (float)z / (float)R - 0.5f;
    float u = x == 0 ? 0.0f : ((float)x) / ((float)R + 1.0f);
    float v = ((float)z) / ((float)R + 1.0f);
    float4 *o = vtx + vi * 3;
    nt_store(o + 0, make_float4(px, 0.0f, pz, 0.0f));     // position, normal.x
    nt_store(o + 1, make_float4(0.0f, -1.0f, 1.0f, 0.0f));  // normal.yz, tangent.xy
    nt_store(o + 2, make_float4(0.0f, -1.0f, u, v));      // tangent.zw, texCoord0
}

// flat index i -> triangle i/3 (ti), corner i%3.  ti = 2R(z-1) + 2(x-1) + s, s in {0,1}:
//   s=0: vi + (-R-2, -1, -R-1);  s=1: vi + (-R-1, -1, 0);  vi = (R+1) z + x
__device__ __forceinline__ uint32_t mesh_index(uint32_t i, uint32_t R) {
    uint32_t ti = i / 3u, corner = i - ti * 3u;
    uint32_t quad = ti >> 1, s = ti & 1u;
    uint32_t zq = quad / R, xq = quad - zq * R;  // z-1, x-1
    uint32_t vi = (R + 1u) * (zq + 1u) + xq + 1u;
    uint32_t a = s ? vi - R - 1u : vi - R - 2u;
    uint32_t b = vi - 1u;
    uint32_t c = s ? vi : vi - R - 1u;
    return corner == 0 ? a : (corner == 1 ? b : c);
}

// Four consecutive indices (one 16-byte store) lie in at most two neighbouring quads: one division by the run-time R
// per thread (reciprocal multiply + two corrections, exact for every 32-bit quad number) instead of one per index.
//   quad q = (zq, xq):  e0..e5 = vi-R-2, vi-1, vi-R-1 | vi-R-1, vi-1, vi   with vi = (R+1)(zq+1) + xq+1
__global__ __launch_bounds__(CT) void mesh_index_kernel(uint32_t *__restrict__ idx, uint32_t R, uint32_t Rinv, size_t n,
                                                       size_t istride) {
    idx += blockIdx.y * istride;
    size_t i = ((size_t)blockIdx.x * CT + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 4 <= n && ((reinterpret_cast<uintptr_t>(idx) & 15) == 0)) {
        const uint32_t i0 = (uint32_t)i, q = i0 / 6u, rem = i0 - q * 6u;  // rem = 0, 2 or 4
        uint32_t zq = __umulhi(q, Rinv);  // floor(2^32 / R): at most two short of q / R
        uint32_t xq = q - zq * R;
        if (xq >= R) { zq += 1u; xq -= R; }
        if (xq >= R) { zq += 1u; xq -= R; }
        const uint32_t vi = (R + 1u) * (zq + 1u) + xq + 1u;
        const bool wrap = xq + 1u >= R;                      // the next quad starts the next row of quads
        const uint32_t vi2 = wrap ? vi + 2u : vi + 1u;       // (R+1)(zq+2) + 1 = vi - xq + R + 1 with xq = R - 1
        const uint32_t e0 = vi - R - 2u, e1 = vi - 1u, e2 = vi - R - 1u, e5 = vi;
        uint4 v;
        if (rem == 0u) {
            v = make_uint4(e0, e1, e2, e2);
        } else if (rem == 2u) {
            v = make_uint4(e2, e2, e1, e5);
        } else {
            v = make_uint4(e1, e5, vi2 - R - 2u, vi2 - 1u);
        }
        nt_store(reinterpret_cast<uint4 *>(idx + i), v);
    } else {
        for (size_t k = i; k < n && k < i + 4; k++) idx[k] = mesh_index((uint32_t)k, R);
    }
}

// PositionStream16 / TriangleUInt16 (Mesh/Streams/PositionStream.cs:11-74, Triangle.cs:7-17): the same indices
// truncated to 16 bits -- `(ushort) t.x`, valid only while (R + 1)^2 <= 65536 as the reference notes; four per thread
__global__ __launch_bounds__(CT) void mesh_index16_kernel(uint16_t *__restrict__ idx, uint32_t R, size_t n, size_t istride) {
    idx += blockIdx.y * istride;
    size_t i = ((size_t)blockIdx.x * CT + threadIdx.x) * 4;
    for (size_t k = i; k < n && k < i + 4; k++) idx[k] = (uint16_t)mesh_index((uint32_t)k, R);
}

}  // namespace

// floor(2^32 / R), saturated (R = 1), for mesh_index_kernel's quad -> (row, column) split
static uint32_t mesh_rinv(int res) {
    uint64_t v = 0x100000000ULL / (uint64_t)res;
    return v > 0xffffffffULL ? 0xffffffffu : (uint32_t)v;
}

int32_t nz_launch_mesh(hipStream_t s, int meshType, void *vertices, uint32_t *indices, int res, int in_res,
                       float tile_height, float tile_size, const float *heights, int count, int index16) {
    mesh_params g;
    g.res = res;
    g.in_res = in_res;
    g.off = (in_res - res) / 2;  // PixOffset
    g.type = meshType;
    g.height = tile_height;
    g.tile_size = tile_size;
    g.normal_strength = 8.0f;  // HeightMapMeshJob.cs:41
    size_t nv = (size_t)(res + 1) * (res + 1);
    size_t ni = (size_t)6 * res * res;
    g.hstride = (size_t)in_res * in_res;
    g.vstride = nv * 3;
    g.istride = ni;
    if (ni > 0xffffffffULL) {
        nz_set_error("mesh index count overflows uint32");
        return NZ_ERR_INVALID;
    }
    if ((reinterpret_cast<uintptr_t>(vertices) & 15) != 0) {
        nz_set_error("vertex buffer must be 16-byte aligned");
        return NZ_ERR_INVALID;
    }
    if ((count > 1 && (nv * 48) % 16 != 0) || count < 1 || count > 65535) {
        nz_set_error("mesh batch: count %d unsupported", count);
        return NZ_ERR_INVALID;
    }
    hipLaunchKernelGGL(mesh_vertex_kernel, dim3((unsigned)((nv + CT - 1) / CT), count), dim3(CT), 0, s,
                       reinterpret_cast<float4 *>(vertices), heights, g);
    NZ_HIP(hipGetLastError());
    size_t nthreads = (ni + 3) / 4;
    if (index16)
        NZ_LAUNCH(mesh_index16_kernel, dim3((unsigned)((nthreads + CT - 1) / CT), count), dim3(CT), 0, s,
                           reinterpret_cast<uint16_t *>(indices), (uint32_t)res, ni, ni);
    else
        NZ_LAUNCH(mesh_index_kernel, dim3((unsigned)((nthreads + CT - 1) / CT), count), dim3(CT), 0, s, indices,
                           (uint32_t)res, mesh_rinv(res), ni, ni);
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}

int32_t nz_launch_mesh_planar(hipStream_t s, void *vertices, uint32_t *indices, int res) {
    size_t nv = (size_t)(res + 1) * (res + 1);
    size_t ni = (size_t)6 * res * res;
    if (ni > 0xffffffffULL) {
        nz_set_error("mesh index count overflows uint32");
        return NZ_ERR_INVALID;
    }
    if ((reinterpret_cast<uintptr_t>(vertices) & 15) != 0) {
        nz_set_error("vertex buffer must be 16-byte aligned");
        return NZ_ERR_INVALID;
    }
    hipLaunchKernelGGL(mesh_planar_vertex_kernel, dim3((unsigned)((nv + CT - 1) / CT)), dim3(CT), 0, s,
                       reinterpret_cast<float4 *>(vertices), res);
    NZ_HIP(hipGetLastError());
    size_t nthreads = (ni + 3) / 4;
    NZ_LAUNCH(mesh_index_kernel, dim3((unsigned)((nthreads + CT - 1) / CT)), dim3(CT), 0, s, indices,
                       (uint32_t)res, mesh_rinv(res), ni, (size_t)0);
    NZ_HIP(hipGetLastError());
    return NZ_OK;
}
