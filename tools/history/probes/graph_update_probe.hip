// Probe: what updating the kernel arguments of an instantiated hipGraph costs on the host (hipGraphExecKernelNodeSetParams)
// against plain launches, for a chain of dependent kernels with a 3.4 KB by-value argument (the epoch loop's shape).
// build: hipcc -O2 --offload-arch=gfx950 -o graph_update_probe graph_update_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
struct Big { int v[850]; int* out; };
__global__ void __launch_bounds__(1024) k_big(Big b) { if (threadIdx.x == 0 && blockIdx.x == 0) b.out[0] += b.v[0]; }
struct Small { int v[60]; int* out; };
__global__ void __launch_bounds__(320) k_small(Small s) { if (threadIdx.x == 0 && blockIdx.x == 0) s.out[1] += s.v[0]; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    int* out; CK(hipMalloc(&out, 64)); CK(hipMemset(out, 0, 64));
    hipStream_t s; CK(hipStreamCreate(&s));
    const int NB = 64;      // mini-batches per graph: 2 nodes each
    Big b; Small sm; for (int i = 0; i < 850; ++i) b.v[i] = 1; for (int i = 0; i < 60; ++i) sm.v[i] = 1; b.out = out; sm.out = out;
    CK(hipFuncSetAttribute((const void*)k_big, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    // plain launches
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipStreamSynchronize(s));
        double t0 = now();
        for (int i = 0; i < NB; ++i) {
            hipLaunchKernelGGL(k_big, dim3(256), dim3(1024), 150 * 1024, s, b);
            hipLaunchKernelGGL(k_small, dim3(165), dim3(320), 0, s, sm);
        }
        double t1 = now();
        CK(hipStreamSynchronize(s));
        double t2 = now();
        printf("plain: host %.2f us per pair, done %.2f us per pair\n", (t1 - t0) / NB * 1e6, (t2 - t0) / NB * 1e6);
    }
    // graph
    hipGraph_t g; CK(hipGraphCreate(&g, 0));
    std::vector<hipGraphNode_t> nodes;
    hipGraphNode_t prev = nullptr;
    for (int i = 0; i < 2 * NB; ++i) {
        hipKernelNodeParams p = {};
        void* args[1];
        if (i & 1) { p.func = (void*)k_small; p.gridDim = dim3(165); p.blockDim = dim3(320); p.sharedMemBytes = 0; args[0] = &sm; }
        else { p.func = (void*)k_big; p.gridDim = dim3(256); p.blockDim = dim3(1024); p.sharedMemBytes = 150 * 1024; args[0] = &b; }
        p.kernelParams = args; p.extra = nullptr;
        hipGraphNode_t n;
        CK(hipGraphAddKernelNode(&n, g, prev ? &prev : nullptr, prev ? 1 : 0, &p));
        nodes.push_back(n); prev = n;
    }
    double t0 = now();
    hipGraphExec_t ex[2];
    CK(hipGraphInstantiate(&ex[0], g, nullptr, nullptr, 0));
    double t1 = now();
    CK(hipGraphInstantiate(&ex[1], g, nullptr, nullptr, 0));
    printf("instantiate: %.1f us for %d nodes\n", (t1 - t0) * 1e6, 2 * NB);
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipStreamSynchronize(s));
        double a0 = now();
        for (int r = 0; r < 4; ++r) CK(hipGraphLaunch(ex[r & 1], s));
        double a1 = now();
        CK(hipStreamSynchronize(s));
        double a2 = now();
        printf("graph replay: host %.2f us per pair, done %.2f us per pair\n", (a1 - a0) / (4 * NB) * 1e6, (a2 - a0) / (4 * NB) * 1e6);
    }
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipStreamSynchronize(s));
        double a0 = now(), upd = 0;
        for (int r = 0; r < 4; ++r) {
            double u0 = now();
            for (int i = 0; i < 2 * NB; ++i) {
                hipKernelNodeParams p = {};
                void* args[1];
                b.v[1] = rep * 100 + r; sm.v[1] = i;
                if (i & 1) { p.func = (void*)k_small; p.gridDim = dim3(165); p.blockDim = dim3(320); p.sharedMemBytes = 0; args[0] = &sm; }
                else { p.func = (void*)k_big; p.gridDim = dim3(256 - (i & 2)); p.blockDim = dim3(1024); p.sharedMemBytes = 150 * 1024; args[0] = &b; }
                p.kernelParams = args; p.extra = nullptr;
                CK(hipGraphExecKernelNodeSetParams(ex[r & 1], nodes[i], &p));
            }
            upd += now() - u0;
            CK(hipGraphLaunch(ex[r & 1], s));
        }
        double a1 = now();
        CK(hipStreamSynchronize(s));
        double a2 = now();
        printf("update + replay: SetParams %.2f us per node, host %.2f us per pair, done %.2f us per pair\n",
               upd / (4 * 2 * NB) * 1e6, (a1 - a0) / (4 * NB) * 1e6, (a2 - a0) / (4 * NB) * 1e6);
    }
    int h[2]; CK(hipMemcpy(h, out, 8, hipMemcpyDeviceToHost));
    printf("counters %d %d\n", h[0], h[1]);
    return 0;
}
