// What does a stage marker cost the stream?  N x (kernel, marker) back to back, marker = nothing / hipEventRecord /
// hipStreamWriteValue32 to mapped host memory.   hipcc --offload-arch=gfx950 -O2 tools/probe_write_value.hip -o /tmp/pwv
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void work(float *p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0001f + 1.0f;
}
int main() {
    const size_t n = (size_t)4096 * 4096;
    float *p;
    CK(hipMalloc(&p, n * 4));
    CK(hipMemset(p, 0, n * 4));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    uint32_t *host, *dev;
    CK(hipHostMalloc((void **)&host, 64, hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void **)&dev, host, 0));
    *host = 0;
    const int N = 2000;
    std::vector<hipEvent_t> ev(N);
    for (auto &e : ev) CK(hipEventCreate(&e));
    for (int mode = 0; mode < 4; mode++)
        for (int rep = 0; rep < 3; rep++) {
            CK(hipStreamSynchronize(s));
            auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < N; k++) {
                if (mode == 3) hipExtLaunchKernelGGL(work, dim3((unsigned)(n / 256)), dim3(256), 0, s, nullptr, ev[k], 0, p, n);
                else hipLaunchKernelGGL(work, dim3((unsigned)(n / 256)), dim3(256), 0, s, p, n);
                if (mode == 1) CK(hipEventRecord(ev[k], s));
                if (mode == 2) CK(hipStreamWriteValue32(s, dev, (uint32_t)(k + 1), 0));
            }
            CK(hipStreamSynchronize(s));
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            std::printf("%-28s %.2f us per kernel + marker (host word %u)\n",
                        mode == 0 ? "no marker" : mode == 1 ? "hipEventRecord" : mode == 2 ? "hipStreamWriteValue32" : "hipExtLaunchKernelGGL stop event", us, *host);
        }
    float ms = -1.0f;
    hipError_t e = hipEventElapsedTime(&ms, ev[10], ev[N - 1]);
    std::printf("elapsed between ext stop events 10 and %d: %s, %.3f ms; query of the last: %s\n", N - 1, hipGetErrorString(e), ms,
                hipGetErrorString(hipEventQuery(ev[N - 1])));
    return 0;
}
