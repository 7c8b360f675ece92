// What does an instruction cost a wave that has its SIMD to itself?  One wave per workgroup, `blocks` workgroups; every
// wave runs the same stream and lane 0 of block 0 reports shader clocks per instruction:
//   dep / indep : a loop of 64 v_fma_f32, each depending on the one before / on the one 8 before
//   long body   : the same 64-instruction group unrolled to 256 / 1024 / 4096 instructions per loop trip (2 / 8 / 32 KB of
//                 code: does a lone wave's instruction fetch keep up?)
// build: hipcc --offload-arch=gfx950 -O3 lone_wave.hip -o lone_wave ; run: ./lone_wave
#include <hip/hip_runtime.h>
#include <cstdio>

#define DEP8(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[0]) : "v"(c), "v"(a));
#define IND8(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c), "v"(a));
#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define R64(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X)
#define R256(X) R64(X) R64(X) R64(X) R64(X)
#define R1024(X) R256(X) R256(X) R256(X) R256(X)
#define R4096(X) R1024(X) R1024(X) R1024(X) R1024(X)

template <int KIND>
__global__ __launch_bounds__(64) void k(float *out, unsigned long long *clk, float a, float c, int trips) {
    float r[8];
    for (int i = 0; i < 8; i++) r[i] = a + threadIdx.x * 1e-6f + i;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < trips; it++) {
        if (KIND == 0) { R64(DEP8) }
        if (KIND == 1) { R64(IND8) }
        if (KIND == 2) { R256(DEP8) }
        if (KIND == 3) { R1024(DEP8) }
        if (KIND == 4) { R4096(DEP8) }
        if (KIND == 5) { R4096(IND8) }
    }
    const unsigned long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 8; i++) s += r[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}

template <int KIND>
void run(const char *name, int per_trip, int blocks) {
    float *out;
    unsigned long long *clk, h = 0;
    hipMalloc(&out, blocks * 64 * 4);
    hipMalloc(&clk, 8);
    const int trips = 65536 / per_trip * 4;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, clk, 1.0f, 0.999f, trips);
        hipDeviceSynchronize();
    }
    hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    printf("%-44s %4d waves  %6.2f clocks per instruction\n", name, blocks, (double)h / ((double)trips * per_trip));
    hipFree(out);
    hipFree(clk);
}

int main() {
    for (int blocks : {1, 160, 1024}) {
        run<0>("dependent, 64 per trip (0.5 KB)", 64, blocks);
        run<1>("independent (8 chains), 64 per trip", 64, blocks);
        run<2>("dependent, 256 per trip (2 KB)", 256, blocks);
        run<3>("dependent, 1024 per trip (8 KB)", 1024, blocks);
        run<4>("dependent, 4096 per trip (32 KB)", 4096, blocks);
        run<5>("independent, 4096 per trip (32 KB)", 4096, blocks);
    }
    return 0;
}
