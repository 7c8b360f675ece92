// Probe (round 6): can a captured graph be replayed with new by-value kernel arguments on ROCm 7.2 / gfx950?
//   (a) hipGraphExecKernelNodeSetParams on a node found by its function pointer, the argument patched through the
//       pointers hipGraphKernelNodeGetParams hands out;
//   (b) hipGraphExecUpdate from a fresh capture of the same launches.
// Checks results (also two replays in flight with different arguments) and times the host side of each form.
// build: hipcc --offload-arch=gfx950 -O2 tools/microbench/graph_update.hip -o tools/microbench/graph_update
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)
struct P { float a; int pad[5]; unsigned epoch; };
__global__ void k1(float *out, int n, P p) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = p.a * i + p.epoch; }
__global__ void k2(const float *in, float *acc, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) acc[i] += in[i]; }
static void enqueue(hipStream_t s, float *t, float *acc, int n, P p) {
    hipLaunchKernelGGL(k1, dim3((n + 255) / 256), dim3(256), 0, s, t, n, p);
    hipLaunchKernelGGL(k2, dim3((n + 255) / 256), dim3(256), 0, s, t, acc, n);
}
int main() {
    const int n = 1 << 16;
    float *t, *acc;
    CK(hipMalloc(&t, n * 4)); CK(hipMalloc(&acc, n * 4)); CK(hipMemset(acc, 0, n * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    P p{1.0f, {0}, 0};
    hipGraph_t g; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed)); enqueue(s, t, acc, n, p); CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn));
    std::vector<hipGraphNode_t> nodes(nn); CK(hipGraphGetNodes(g, nodes.data(), &nn));
    hipGraphNode_t n1 = nullptr; hipKernelNodeParams np{};
    for (auto nd : nodes) {
        hipGraphNodeType ty; CK(hipGraphNodeGetType(nd, &ty));
        if (ty != hipGraphNodeTypeKernel) continue;
        hipKernelNodeParams q{}; CK(hipGraphKernelNodeGetParams(nd, &q));
        printf("node func %p (k1 %p, k2 %p) grid %u\n", q.func, (void *)k1, (void *)k2, q.gridDim.x);
        if (q.func == (void *)k1) { n1 = nd; np = q; }
    }
    if (!n1) { printf("k1's node not found by function pointer\n"); return 2; }
    // (a) three replays in flight: a = 1, 2, 4 -> acc[i] = 7 i + (0 + 1 + 2)
    for (int r = 0; r < 3; r++) {
        P *pp = (P *)np.kernelParams[2];
        pp->a = (float)(1 << r); pp->epoch = r;
        CK(hipGraphExecKernelNodeSetParams(ex, n1, &np));
        CK(hipGraphLaunch(ex, s));
    }
    CK(hipStreamSynchronize(s));
    std::vector<float> h(n); CK(hipMemcpy(h.data(), acc, n * 4, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < n; i++) bad += h[i] != 7.0f * i + 3.0f;
    printf("(a) SetParams, three replays in flight: %s (%d wrong; acc[5] = %g, want 38)\n", bad ? "WRONG" : "ok", bad, h[5]);
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < 2000; r++) { ((P *)np.kernelParams[2])->epoch = r; CK(hipGraphExecKernelNodeSetParams(ex, n1, &np)); CK(hipGraphLaunch(ex, s)); }
    auto t1 = std::chrono::steady_clock::now(); CK(hipStreamSynchronize(s)); auto t2 = std::chrono::steady_clock::now();
    printf("    host %.2f us per SetParams + launch, %.2f us per replay drained\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / 2000,
           std::chrono::duration<double, std::micro>(t2 - t0).count() / 2000);
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < 2000; r++) CK(hipGraphLaunch(ex, s));
    t1 = std::chrono::steady_clock::now(); CK(hipStreamSynchronize(s)); t2 = std::chrono::steady_clock::now();
    printf("    host %.2f us per launch alone, %.2f us per replay drained\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / 2000,
           std::chrono::duration<double, std::micro>(t2 - t0).count() / 2000);
    // (b) whole-graph update from a fresh capture
    CK(hipMemset(acc, 0, n * 4));
    for (int r = 0; r < 3; r++) {
        P q{(float)(1 << r), {0}, (unsigned)r};
        hipGraph_t g2; CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed)); enqueue(s, t, acc, n, q); CK(hipStreamEndCapture(s, &g2));
        hipGraphNode_t err; hipGraphExecUpdateResult res;
        hipError_t e = hipGraphExecUpdate(ex, g2, &err, &res);
        if (e != hipSuccess) { printf("(b) hipGraphExecUpdate: %s (result %d)\n", hipGetErrorString(e), (int)res); return 3; }
        CK(hipGraphLaunch(ex, s)); CK(hipGraphDestroy(g2));
    }
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), acc, n * 4, hipMemcpyDeviceToHost));
    bad = 0; for (int i = 0; i < n; i++) bad += h[i] != 7.0f * i + 3.0f;
    printf("(b) ExecUpdate, three replays in flight: %s (%d wrong)\n", bad ? "WRONG" : "ok", bad);
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < 2000; r++) {
        P q{1.0f, {0}, (unsigned)r};
        hipGraph_t g2; CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed)); enqueue(s, t, acc, n, q); CK(hipStreamEndCapture(s, &g2));
        hipGraphNode_t err; hipGraphExecUpdateResult res; CK(hipGraphExecUpdate(ex, g2, &err, &res)); CK(hipGraphLaunch(ex, s)); CK(hipGraphDestroy(g2));
    }
    t1 = std::chrono::steady_clock::now(); CK(hipStreamSynchronize(s));
    printf("    host %.2f us per capture + update + launch\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / 2000);
    return 0;
}
