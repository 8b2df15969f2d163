// kernarg_preload_probe.hip -- does kernel-argument PRELOAD (the first dwords of the argument segment arrive in SGPRs with the
// wave; -mllvm -amdgpu-kernarg-preload-count=N) take the argument block's trip to memory off a kernel's critical path?
// Eager launches (every launch gets a freshly written argument block, as in an epoch loop): 128 workgroups x 1024 lanes, each
// loads 16 bytes per lane from a pointer argument (warm data), sums, writes one word; ~3 us of dependent FMAs behind it so
// that the launch is longer than the command processor's floor.  Built twice by the caller:
//   hipcc -O3 --offload-arch=gfx950 -o probe_plain  kernarg_preload_probe.hip
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=8 -o probe_preload kernarg_preload_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct Big { int v[600]; };      // 2.4 KB behind the scalar arguments, like the step's descriptors
template <int SPIN>
__global__ void __launch_bounds__(1024) k(const float4* p, float* out, int pick, Big b) {
    const float4 v = p[blockIdx.x * 1024 + threadIdx.x];
    float acc = v.x + v.y + v.z + v.w;
#pragma unroll 1
    for (int i = 0; i < SPIN; ++i) acc = __builtin_fmaf(acc, 1.0000001f, 1e-9f);      // (a constant: the same loop in both builds)
    acc += (float)b.v[pick];                       // one late read of the block (a line nobody touched before)
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}
int main() {
    float4* p; float* out;
    CHECK(hipMalloc(&p, 128 * 1024 * 16)); CHECK(hipMemset(p, 0, 128 * 1024 * 16)); CHECK(hipMalloc(&out, 4096));
    Big b; for (int i = 0; i < 600; ++i) b.v[i] = i;
    hipStream_t s; CHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 6; ++rep) {
        const int spin = (rep & 1) ? 600 : 300;      // device-bound either way (the host enqueues a launch in ~3.7 us)
#define LAUNCH(i) do { if (spin == 600) hipLaunchKernelGGL(k<600>, dim3(128), dim3(1024), 0, s, p, out, (i) % 600, b); \
                       else hipLaunchKernelGGL(k<300>, dim3(128), dim3(1024), 0, s, p, out, (i) % 600, b); } while (0)
        for (int i = 0; i < 200; ++i) LAUNCH(i);
        CHECK(hipStreamSynchronize(s));
        const int n = 8000;
        CHECK(hipEventRecord(e0, s));
        for (int i = 0; i < n; ++i) LAUNCH(i);
        CHECK(hipEventRecord(e1, s));
        CHECK(hipStreamSynchronize(s));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("spin %3d: %.2f us per launch (eager, %d launches)\n", spin, ms * 1e3 / n, n);
    }
    float h = 0; CHECK(hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost));
    printf("check %.1f\n", h);
    return 0;
}
