// Probe (round 6): what a completion event costs a replayed graph -- recorded behind hipGraphLaunch, or as a captured
// event-record node inside the graph -- and what a kernel's read of its parameters from mapped host memory costs.
// build: hipcc --offload-arch=gfx950 -O2 tools/microbench/graph_event.hip -o tools/microbench/graph_event
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void work(float *out, int n, const int *box, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    float a = box ? (float)box[0] : 1.0f;
    float v = a * i;
    for (int k = 0; k < iters; k++) v = v * 1.0001f + 0.5f;
    if (i < n) out[i] = v;
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const int n = 256 * 128, reps = 3000;
    float *t; CK(hipMalloc(&t, n * 4));
    int *box_h, *box_d; CK(hipHostMalloc((void **)&box_h, 64, hipHostMallocMapped)); CK(hipHostGetDevicePointer((void **)&box_d, box_h, 0));
    int *box_dev; CK(hipMalloc(&box_dev, 64)); CK(hipMemset(box_dev, 0, 64));
    box_h[0] = 1;
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (int variant = 0; variant < 5; variant++) {
        // 0: five kernels; 1: + event node captured at the end; 2: five kernels, event recorded after each launch
        // 3: first kernel reads its parameter from mapped host memory; 4: ... from device memory
        const int *box = variant == 3 ? box_d : variant == 4 ? box_dev : nullptr;
        hipGraph_t g; hipGraphExec_t ex;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        for (int k = 0; k < 5; k++) hipLaunchKernelGGL(work, dim3(128), dim3(256), 0, s, t, n, k == 0 ? box : nullptr, 2000);
        if (variant == 1) CK(hipEventRecord(ev, s));
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        for (int r = 0; r < 50; r++) CK(hipGraphLaunch(ex, s));
        CK(hipStreamSynchronize(s));
        double t0 = now();
        for (int r = 0; r < reps; r++) {
            if (variant == 3) box_h[0] = r;
            CK(hipGraphLaunch(ex, s));
            if (variant == 2) CK(hipEventRecord(ev, s));
        }
        CK(hipStreamSynchronize(s));
        double dt = (now() - t0) / reps;
        const char *names[] = {"5 kernels", "5 kernels + captured event node", "5 kernels, hipEventRecord behind the launch",
                               "first kernel reads mapped host memory", "first kernel reads device memory"};
        printf("%-48s %.2f us per replay", names[variant], dt);
        if (variant == 1) { hipError_t q = hipEventQuery(ev); printf("  (event query after sync: %s)", hipGetErrorString(q)); }
        printf("\n");
        CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(g));
    }
    {   // does an event recorded by a captured node behave like a recorded event for the host right after hipGraphLaunch?
        hipGraph_t g; hipGraphExec_t ex;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        for (int k = 0; k < 5; k++) hipLaunchKernelGGL(work, dim3(128), dim3(256), 0, s, t, n, nullptr, 200000);  // ~3 ms each
        CK(hipEventRecord(ev, s));
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        for (int round = 0; round < 3; round++) {
            CK(hipStreamSynchronize(s));
            double t0 = now();
            CK(hipGraphLaunch(ex, s));
            hipError_t q = hipEventQuery(ev);
            double t1 = now();
            hipError_t w = hipEventSynchronize(ev);
            double t2 = now();
            hipError_t q2 = hipStreamQuery(s);
            printf("round %d: query right after launch: %s; hipEventSynchronize: %s after %.0f us; stream then: %s\n", round,
                   q == hipSuccess ? "COMPLETE (stale)" : hipGetErrorString(q), hipGetErrorString(w), t2 - t1, hipGetErrorString(q2));
            (void)hipGetLastError();
            (void)t0;
        }
        // and as a dependency of another stream
        hipStream_t s2; CK(hipStreamCreate(&s2));
        CK(hipStreamSynchronize(s));
        CK(hipGraphLaunch(ex, s));
        CK(hipStreamWaitEvent(s2, ev, 0));
        double t1 = now();
        CK(hipStreamSynchronize(s2));
        printf("another stream waiting on the event: released after %.0f us (stream 1 then: %s)\n", now() - t1, hipGetErrorString(hipStreamQuery(s)));
        (void)hipGetLastError();
        CK(hipStreamSynchronize(s));
        CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(g));
    }
    // eager reference: five launches, with and without the riding stop event
    for (int variant = 0; variant < 2; variant++) {
        for (int r = 0; r < 50; r++) hipLaunchKernelGGL(work, dim3(128), dim3(256), 0, s, t, n, nullptr, 2000);
        CK(hipStreamSynchronize(s));
        double t0 = now();
        for (int r = 0; r < reps; r++) {
            for (int k = 0; k < 4; k++) hipLaunchKernelGGL(work, dim3(128), dim3(256), 0, s, t, n, nullptr, 2000);
            if (variant) hipExtLaunchKernelGGL(work, dim3(128), dim3(256), 0, s, nullptr, ev, 0, t, n, nullptr, 2000);
            else hipLaunchKernelGGL(work, dim3(128), dim3(256), 0, s, t, n, nullptr, 2000);
        }
        CK(hipStreamSynchronize(s));
        printf("%-48s %.2f us per 5 launches\n", variant ? "eager, last launch carries the event" : "eager", (now() - t0) / reps);
    }
    return 0;
}
