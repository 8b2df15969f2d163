// How long is a workgroup barrier of 16 waves on gfx950, and what does a phase cost in which ONE wave runs a short dependent
// chain while fifteen wait?  (measurement probe; hipcc --offload-arch=gfx950 -O3 -o barrier_cost barrier_cost.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(1024) k(float* out, int iters, int chain) {
    extern __shared__ float lds[];
    float v = (float)threadIdx.x;
    lds[threadIdx.x] = v;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {                       // barrier only (+ one LDS write/read per wave so that it is a real hand-off)
            lds[threadIdx.x] = v;
            __syncthreads();
            v += lds[(threadIdx.x + 64) & 1023];
        } else if (MODE == 1) {                // wave 0 runs `chain` dependent LDS round trips, the others wait
            __syncthreads();
            if (threadIdx.x < 64) {
                int idx = threadIdx.x;
                for (int c = 0; c < chain; ++c) idx = (int)lds[idx & 1023] & 1023;
                lds[threadIdx.x] = (float)idx;
            }
            __syncthreads();
            v += lds[threadIdx.x & 63];
        } else {                               // wave 0 runs `chain` dependent FMAs
            __syncthreads();
            if (threadIdx.x < 64) {
                float a = v;
                for (int c = 0; c < chain; ++c) a = fmaf(a, 1.0001f, 0.5f);
                lds[threadIdx.x] = a;
            }
            __syncthreads();
            v += lds[threadIdx.x & 63];
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * 1024 + threadIdx.x] = v;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}

template <int MODE> int run(const char* name, int chain, float* d, int blocks) {
    const int iters = 200;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 150 * 1024, 0, d, iters, chain);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 150 * 1024, 0, d, iters, chain);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    float cyc = 0; CHECK(hipMemcpy(&cyc, d, 4, hipMemcpyDeviceToHost));
    printf("%-46s chain %3d blocks %3d: %.0f ns per iteration (%.0f clock64 ticks)\n", name, chain, blocks, ms * 1e6 / iters, cyc / iters);
    return 0;
}

int main() {
    float* d; CHECK(hipMalloc(&d, 256 * 1024 * 4));
    CHECK(hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    CHECK(hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    CHECK(hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    for (int blocks : {1, 128}) {
        if (run<0>("one barrier + LDS hand-off per iteration", 0, d, blocks)) return 1;
        for (int chain : {0, 4, 16}) if (run<1>("two barriers, wave 0: dependent LDS round trips", chain, d, blocks)) return 1;
        for (int chain : {16, 64}) if (run<2>("two barriers, wave 0: dependent FMAs", chain, d, blocks)) return 1;
    }
    return 0;
}
