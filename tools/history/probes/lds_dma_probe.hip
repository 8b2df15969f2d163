// Probe of the gfx950 global->LDS DMA loads (buffer_load_dword ... lds, global_load_lds_dwordx4):
// where does lane L's data land, what do EXEC-masked lanes and out-of-range lanes do?
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 -o /tmp/lds_dma_probe tools/probes/lds_dma_probe.hip && /tmp/lds_dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__global__ void k_dword(const int* src, int* out, int n_valid, int n_exec) {
    extern __shared__ int lds[];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = -7;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(src), 0, n_valid * 4, 0x00020000);
    const int wave = threadIdx.x >> 6;
    if ((int)threadIdx.x < n_exec)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(lds + wave * 64), 4, threadIdx.x * 4, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) out[i] = lds[i];
}
// lane L fetches the 16-byte chunk perm(L) of the source; lands at lds_base + 16 * L
__global__ void k_x4(const float4* src, float* out, int n4) {
    extern __shared__ int lds[];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = -7;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = wave * 64 + (lane ^ 5);
    if ((int)threadIdx.x < n4)
        __builtin_amdgcn_global_load_lds((glb_void*)(src + q), (lds_void*)(lds + wave * 256), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) out[i] = ((float*)lds)[i];
}
int main() {
    int *src, *out;
    hipMalloc(&src, 4096 * 4); hipMalloc(&out, 4096 * 4);
    std::vector<int> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = 1000 + i;
    hipMemcpy(src, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    std::vector<int> o(4096);
    // 256 threads; valid range 100 words; exec for threads < 150
    hipLaunchKernelGGL(k_dword, dim3(1), dim3(256), 8192, 0, src, out, 100, 150);
    hipMemcpy(o.data(), out, 1024 * 4, hipMemcpyDeviceToHost);
    printf("dword DMA: valid 100 words, exec lanes < 150, 256 threads\n");
    int ok_data = 0, oob_zero = 0, oob_keep = 0, masked_keep = 0, masked_other = 0;
    for (int i = 0; i < 256; ++i) {
        if (i < 100) ok_data += (o[i] == 1000 + i);
        else if (i < 150) { oob_zero += (o[i] == 0); oob_keep += (o[i] == -7); }
        else { masked_keep += (o[i] == -7); masked_other += (o[i] != -7); }
    }
    printf("  in-range lanes correct: %d/100   out-of-range lanes: zero %d, untouched %d (of 50)   masked lanes untouched %d, changed %d (of 106)\n",
           ok_data, oob_zero, oob_keep, masked_keep, masked_other);
    std::vector<float> hf(4096), of(4096);
    for (int i = 0; i < 4096; ++i) hf[i] = (float)i;
    hipMemcpy(src, hf.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_x4, dim3(1), dim3(128), 16384, 0, (const float4*)src, (float*)out, 100);
    hipMemcpy(of.data(), out, 2048 * 4, hipMemcpyDeviceToHost);
    int good = 0, bad = 0, keep = 0;
    for (int t = 0; t < 128; ++t) {
        const int wave = t >> 6, lane = t & 63, q = wave * 64 + (lane ^ 5);
        for (int j = 0; j < 4; ++j) {
            const float v = of[t * 4 + j];
            if (t < 100) { if (v == (float)(q * 4 + j)) ++good; else ++bad; }
            else keep += (((int*)of.data())[t * 4 + j] == -7);
        }
    }
    printf("dwordx4 DMA with permuted per-lane sources: %d correct, %d wrong (of 400); masked lanes' words untouched %d (of 112)\n", good, bad, keep);
    hipError_t e = hipDeviceSynchronize();
    printf("status %s\n", hipGetErrorString(e));
    return 0;
}
