// probe: sc1 buffer loads / stores through the rsrc builtins round-trip data (used by conv_chain_kernel)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__global__ void k(const float* src, float* dst, int n4) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, -1, 0x00020000);
    __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, -1, 0x00020000);
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    v4u t = __builtin_amdgcn_raw_buffer_load_b128(rs, i * 16, 0, 16);
    __builtin_amdgcn_raw_buffer_store_b128(t, rd, i * 16, 0, 16);
}
int main() {
    const int n = 1 << 22;
    std::vector<float> h(n), o(n);
    for (int i = 0; i < n; i++) h[i] = (float)i;
    float *a, *b;
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 4);
    hipMemcpy(a, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(b, 0, n * 4);
    hipLaunchKernelGGL(k, dim3(n / 4 / 256), dim3(256), 0, 0, a, b, n / 4);
    hipMemcpy(o.data(), b, n * 4, hipMemcpyDeviceToHost);
    int bad = 0, first = -1;
    for (int i = 0; i < n; i++) if (o[i] != h[i]) { if (first < 0) first = i; bad++; }
    printf("bad %d first %d  o[0..7] %g %g %g %g %g %g %g %g\n", bad, first, o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]);
    return bad != 0;
}
