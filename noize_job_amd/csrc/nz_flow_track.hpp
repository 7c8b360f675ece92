// WorldTile.UpdateFlowMapFromTrack (LiveErosionDataTypes.cs:869-886, UpdateFlowFromTrackJob) per cell and per quad of cells:
// shared by flow_from_track_kernel (nz_elementwise.hip) and by the pile solver's launch that carries the flow update's
// workgroups behind its own (pile_ticket_flow_kernel, nz_live.hip).
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ float flow_from_track_cell(float pv, float tv, float poolV, float flowLossRate) {
    const float MINFLOWPOOL = .00005f;
    if (poolV > MINFLOWPOOL) return ((1.0f - 0.1f * flowLossRate) * pv);
    if (tv > 0.0f) return ((1.0f - flowLossRate) * pv) + (flowLossRate * 50.0f * tv) / (1.0f + 50.0f * tv);
    return (1.0f - flowLossRate) * pv;
}

// Four cells, 16-byte accesses, values already loaded.  The flow plane decays wherever it is not zero, the track is zero
// wherever no particle went this cycle and the pool wherever no water stands -- nearly everywhere -- and decaying a zero,
// zeroing a zero or drying a dry cell changes no bit: those stores are left out (a quad is stored when any of its four cells
// changes), 12 ... 16 instead of 24 bytes per cell.
__device__ __forceinline__ void flow_from_track_quad(float *__restrict__ pool, float *__restrict__ flow, float *__restrict__ track,
                                                     size_t i, const float4 pv, const float4 tv, const float4 po,
                                                     float flowLossRate, float evaporation) {
    const float4 f = make_float4(flow_from_track_cell(pv.x, tv.x, po.x, flowLossRate), flow_from_track_cell(pv.y, tv.y, po.y, flowLossRate),
                                 flow_from_track_cell(pv.z, tv.z, po.z, flowLossRate), flow_from_track_cell(pv.w, tv.w, po.w, flowLossRate));
    const float4 pn = make_float4(fmaxf(po.x - evaporation, 0.0f), fmaxf(po.y - evaporation, 0.0f), fmaxf(po.z - evaporation, 0.0f),
                                  fmaxf(po.w - evaporation, 0.0f));
    // (flow that is zero stays zero where no particle went: (1 - loss) * 0 == 0, the same bits -- on a map that is mostly
    // untouched the flow store of most quads is left out as well, 12 + a little instead of 16 bytes per cell; round 5)
    const unsigned flow_diff = (__float_as_uint(f.x) ^ __float_as_uint(pv.x)) | (__float_as_uint(f.y) ^ __float_as_uint(pv.y)) |
                               (__float_as_uint(f.z) ^ __float_as_uint(pv.z)) | (__float_as_uint(f.w) ^ __float_as_uint(pv.w));
    if (flow_diff) *reinterpret_cast<float4 *>(flow + i) = f;
    const unsigned track_bits = __float_as_uint(tv.x) | __float_as_uint(tv.y) | __float_as_uint(tv.z) | __float_as_uint(tv.w);
    if (track_bits) *reinterpret_cast<float4 *>(track + i) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const unsigned pool_diff = (__float_as_uint(pn.x) ^ __float_as_uint(po.x)) | (__float_as_uint(pn.y) ^ __float_as_uint(po.y)) |
                               (__float_as_uint(pn.z) ^ __float_as_uint(po.z)) | (__float_as_uint(pn.w) ^ __float_as_uint(po.w));
    if (pool_diff) *reinterpret_cast<float4 *>(pool + i) = pn;
}

// the cells a quad access cannot take (unaligned planes, the last cells of a plane whose size is not a multiple of four)
__device__ __forceinline__ void flow_from_track_cells(float *pool, float *flow, float *track, size_t i0, size_t i1,
                                                      float flowLossRate, float evaporation) {
    for (size_t k = i0; k < i1; k++) {
        const float pv = flow[k], tv = track[k], poolV = pool[k];
        flow[k] = flow_from_track_cell(pv, tv, poolV, flowLossRate);
        track[k] = 0.0f;
        pool[k] = fmaxf(poolV - evaporation, 0.0f);
    }
}
