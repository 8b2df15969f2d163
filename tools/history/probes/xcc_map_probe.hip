// xcc_map_probe.hip -- which XCD (and CU) a workgroup of a 1-D grid lands on: the step kernels place the workgroups that share a
// graph "8 block ids apart = same XCD" and the builder / prefetch workgroups of graph g on "XCD g % 8"; this prints the
// hardware's XCC_ID per block id for a grid shaped like the step launch (256 workgroups x 1024 lanes, 80 KB of LDS).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(1024) k(int* out) {
    extern __shared__ int lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        out[2 * blockIdx.x] = (int)(xcc & 0xf);
        out[2 * blockIdx.x + 1] = (int)hwid + lds[5] - 5;
    }
}
int main() {
    const int n = 256;
    int* d;
    hipMalloc(&d, 2 * n * 4);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k, dim3(n), dim3(1024), 80 * 1024, 0, d);
        std::vector<int> h(2 * n);
        hipMemcpy(h.data(), d, 2 * n * 4, hipMemcpyDeviceToHost);
        printf("launch %d: xcc of blocks 0..31:", rep);
        for (int i = 0; i < 32; ++i) printf(" %d", h[2 * i]);
        int ok = 0;
        for (int i = 0; i < n; ++i) ok += (h[2 * i] == (h[0] + i) % 8) ? 1 : 0;
        printf("\n   blocks with xcc == (xcc(0) + i) %% 8: %d of %d; xcc(0) = %d\n", ok, n, h[0]);
    }
    return 0;
}
