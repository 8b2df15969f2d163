#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i4 __attribute__((ext_vector_type(4)));
__global__ void k(const int* src, int n, int* out) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(src), 0, n * 4, 0x00020000);
    i4 v = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16, 0, 0);
    out[threadIdx.x * 4 + 0] = v.x; out[threadIdx.x * 4 + 1] = v.y; out[threadIdx.x * 4 + 2] = v.z; out[threadIdx.x * 4 + 3] = v.w;
}
int main() {
    int *d, *o; hipMalloc(&d, 4096); hipMalloc(&o, 4096);
    int h[64]; for (int i = 0; i < 64; ++i) h[i] = 100 + i;
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    for (int off = 0; off < 2; ++off) {      // aligned and +1 word misaligned source
        k<<<1, 4>>>(d + off, 10, o);
        int r[16]; hipMemcpy(r, o, 64, hipMemcpyDeviceToHost);
        printf("src+%d n=10:", off); for (int i = 0; i < 16; ++i) printf(" %d", r[i]); printf("\n");
    }
    return 0;
}
