// Issue rate of a few VALU instructions on gfx950: every wave runs ITER x 8 independent instructions of one kind.
// build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 4096
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, float a, float b) {
    float r[8];
    for (int i = 0; i < 8; i++) r[i] = a + threadIdx.x * 1e-6f + i;
    float c = b;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f q[8], cc = {b, b}, aa = {a, a};
    for (int i = 0; i < 8; i++) q[i] = (v2f){r[i], r[i] + 0.5f};
    asm volatile("s_mov_b32 vcc_lo, 0x55555555\n\ts_mov_b32 vcc_hi, 0x55555555" ::: "vcc");
    asm volatile("s_mov_b32 s10, 0x33333333\n\ts_mov_b32 s11, 0x33333333" ::: "s10", "s11");
    for (int it = 0; it < ITER; it++) {
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c), "v"(a));
#define MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
#define RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
#define DIVSCALE(i) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(r[i]) : "v"(c) : "vcc");
#define DIVFMAS(i) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c), "v"(a) : "vcc");
#define DIVFIXUP(i) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c), "v"(a));
#define MAXF(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
#define CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(c));
#define DPP(i) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r[i]) : "v"(c));
#define ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
#define CNDE64(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(r[i]) : "v"(c));
#define FLOOR(i) asm volatile("v_floor_f32 %0, %0" : "+v"(r[i]));
#define CVTI(i) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(r[i]));
#define CMP(i) asm volatile("v_cmp_gt_f32 s[10:11], %0, %1" : : "v"(r[i]), "v"(c) : "s10", "s11");
#define MED3(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c), "v"(a));
#define CNDE64V(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(c));
#define CND2(i) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(r[i]) : "v"(c));
#define PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(q[i]) : "v"(cc));
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(q[i]) : "v"(cc), "v"(aa));
#define SQRT(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(r[i]));
        if (KIND == 0) { REP8(FMA) }
        if (KIND == 1) { REP8(MUL) }
        if (KIND == 2) { REP8(RCP) }
        if (KIND == 3) { REP8(DIVSCALE) }
        if (KIND == 4) { REP8(DIVFMAS) }
        if (KIND == 5) { REP8(DIVFIXUP) }
        if (KIND == 6) { REP8(MAXF) }
        if (KIND == 7) { REP8(CNDMASK) }
        if (KIND == 8) { REP8(DPP) }
        if (KIND == 9) { REP8(SQRT) }
        if (KIND == 10) { REP8(ADD) }
        if (KIND == 16) { REP8(CNDE64V) }
        if (KIND == 20) { REP8(PKMUL) }
        if (KIND == 21) { REP8(PKFMA) }
        if (KIND == 18) { REP8(FMA) CNDMASK(0) REP8(FMA) CNDMASK(1) }   // 16 fma + 2 e32 selects
        if (KIND == 19) { REP8(FMA) CNDE64V(0) REP8(FMA) CNDE64V(1) }  // 16 fma + 2 e64 selects
        if (KIND == 17) { REP8(CND2) }
        if (KIND == 12) { REP8(FLOOR) }
        if (KIND == 13) { REP8(CVTI) }
        if (KIND == 14) { REP8(CMP) }
        if (KIND == 15) { REP8(MED3) }
        if (KIND == 11) { REP8(CNDE64) }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += r[i] + q[i].x + q[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
double run(float *d, const char *name) {
    const int blocks = 256 * 8;  // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.9999f);
    hipEventRecord(e0);
    for (int w = 0; w < 5; w++) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.9999f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double inst = 5.0 * blocks * 4 /*waves*/ * (double)ITER * 8;
    double per_simd_cycle = inst / (1024.0) / (ms * 1e-3 * 2.4e9);  // wave-instructions per SIMD per cycle at 2.4 GHz
    printf("%-16s %7.3f ms  %.3f wave-instr / SIMD / clk  = %.2f clk per wave-instr\n", name, ms / 5, per_simd_cycle, 1.0 / per_simd_cycle);
    return ms;
}

int main() {
    float *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>(d, "v_fma_f32"); run<0>(d, "v_fma_f32"); run<1>(d, "v_mul_f32"); run<20>(d, "v_pk_mul_f32 (2 results)"); run<21>(d, "v_pk_fma_f32 (2 results)"); run<10>(d, "v_add_f32"); run<11>(d, "v_cndmask_e64 sgpr"); run<6>(d, "v_max_f32"); run<12>(d, "v_floor_f32"); run<13>(d, "v_cvt_i32_f32"); run<14>(d, "v_cmp_gt_f32 -> sgpr"); run<15>(d, "v_med3_f32"); run<7>(d, "v_cndmask_b32 e32 vcc"); run<16>(d, "v_cndmask_e64 vcc"); run<17>(d, "v_cndmask e32 swapped"); run<18>(d, "(16 fma + 2 cnd e32)/8"); run<19>(d, "(16 fma + 2 cnd e64)/8"); run<8>(d, "v_mov_b32_dpp");
    run<2>(d, "v_rcp_f32"); run<9>(d, "v_sqrt_f32"); run<3>(d, "v_div_scale_f32"); run<4>(d, "v_div_fmas_f32"); run<5>(d, "v_div_fixup_f32");
    return 0;
}
