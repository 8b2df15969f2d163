// Instruction-cache behaviour of gfx950 for code that a workgroup executes ONCE:
//   (a) does the cache keep a kernel's code across launches (same kernel back to back / another kernel in between)?
//   (b) what does cold straight-line code cost per instruction, and what does a cold TAKEN branch cost?
// One workgroup of W waves per CU-ish (grid = 256), thread 0 of block 0 stamps the shader clock around the body.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R256(x) R16(R16(x))
#define R1024(x) R4(R256(x))
// 4096 independent-ish VALU instructions, 8 bytes each (32-bit literal) = 32 KB of straight-line code
#define BODY_STRAIGHT R4(R1024(asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(v));))
// the same amount of work, but after every 16 instructions a taken branch over 16 instructions of dead code
#define CHUNK_BR asm volatile( \
    "v_add_u32 %0, 0x12345, %0\n v_add_u32 %0, 0x12345, %0\n v_add_u32 %0, 0x12345, %0\n v_add_u32 %0, 0x12345, %0\n" \
    "v_add_u32 %0, 0x12345, %0\n v_add_u32 %0, 0x12345, %0\n v_add_u32 %0, 0x12345, %0\n v_add_u32 %0, 0x12345, %0\n" \
    "v_add_u32 %0, 0x12345, %0\n v_add_u32 %0, 0x12345, %0\n v_add_u32 %0, 0x12345, %0\n v_add_u32 %0, 0x12345, %0\n" \
    "v_add_u32 %0, 0x12345, %0\n v_add_u32 %0, 0x12345, %0\n v_add_u32 %0, 0x12345, %0\n v_add_u32 %0, 0x12345, %0\n" \
    "s_branch 1f\n" \
    "v_add_u32 %0, 0x54321, %0\n v_add_u32 %0, 0x54321, %0\n v_add_u32 %0, 0x54321, %0\n v_add_u32 %0, 0x54321, %0\n" \
    "v_add_u32 %0, 0x54321, %0\n v_add_u32 %0, 0x54321, %0\n v_add_u32 %0, 0x54321, %0\n v_add_u32 %0, 0x54321, %0\n" \
    "v_add_u32 %0, 0x54321, %0\n v_add_u32 %0, 0x54321, %0\n v_add_u32 %0, 0x54321, %0\n v_add_u32 %0, 0x54321, %0\n" \
    "v_add_u32 %0, 0x54321, %0\n v_add_u32 %0, 0x54321, %0\n v_add_u32 %0, 0x54321, %0\n v_add_u32 %0, 0x54321, %0\n" \
    "1:\n" : "+v"(v));
#define BODY_BRANCHY R256(CHUNK_BR)      // 256 x 16 = 4096 executed instructions, 256 taken branches, 64 KB of code
template <int KIND>
__global__ void __launch_bounds__(1024) k(unsigned long long* out, int slot) {
    unsigned v = threadIdx.x;
    const unsigned long long t0 = clock64();
    if (KIND == 0) { BODY_STRAIGHT } else { BODY_BRANCHY }
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[slot] = t1 - t0;
    if (v == 0xdeadbeef) out[100] = v;
}
__global__ void other(unsigned long long* out) { if (threadIdx.x == 4096) out[101] = 1; }
int main(int argc, char** argv) {
    unsigned long long* d; hipMalloc(&d, 4096); hipMemset(d, 0, 4096);
    unsigned long long h[64];
    for (int threads : {64, 256, 1024}) {
        for (int kind = 0; kind < 2; ++kind) {
            for (int rep = 0; rep < 4; ++rep) {
                if (kind == 0) k<0><<<256, threads>>>(d, rep); else k<1><<<256, threads>>>(d, rep);
                if (rep == 2) other<<<256, 256>>>(d);        // launch 3 follows ANOTHER kernel
            }
            hipDeviceSynchronize();
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            printf("%s, %4d threads/WG: cycles of 4096 executed instructions, launch 0..3 (3 follows another kernel): %llu %llu %llu %llu\n",
                   kind ? "branchy (256 taken branches, 64 KB)" : "straight (32 KB)", threads, h[0], h[1], h[2], h[3]);
        }
    }
    return 0;
}
