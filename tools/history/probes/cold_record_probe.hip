// cold_record_probe.hip -- what the PROLOGUE of a step workgroup costs as a function of where its graph's data lies.
//
// Round 4 measured that a step on a mini-batch that is not cache-resident costs 1.6 - 2.2 us more than a replayed one, the
// same at batch 32 and 64 ("dependent round trips, not bandwidth"), and left one idea untried: ONE contiguous
// record per graph instead of thirteen arrays in ten regions of an array-major workspace.  This probe isolates the question:
// 128 workgroups of 1024 lanes each load ~40 KB (the bytes a GINet branch workgroup stages: S rows, IHORD, the pooled
// level's arrays, counts) into LDS and write one word, nothing else.
//   layout  scattered: array a of graph g lies at base_a + g * stride_a (array-major, as drgnn_topology_layout does)
//           record   : everything of graph g lies in [g * REC, (g + 1) * REC)
//   order   one burst: all loads issued, then waited for
//           two hops : a first load (the counts) is waited for before the rest is issued (the pre-round-2 prologue)
//   data    warm: the same mini-batch every launch;  cold: a cycle of K mini-batches (K x ~5 MB >> the L2s)
// Build: hipcc -O3 --offload-arch=gfx950 -o cold_record_probe cold_record_probe.hip ; run: ./cold_record_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NARR = 10;
struct Layout {
    // words per graph of each array (float4-aligned)
    int words[NARR];
    long base[NARR];      // scattered: word offset of array a inside one mini-batch's block
    long stride[NARR];    // scattered: words between graph g and g + 1 of array a
    long rec;             // record: words per graph
    long rec_off[NARR];   // record: offset of array a inside the record
    long batch_words;     // words of one mini-batch's block (either layout)
};

#define MAXK 160
struct Args {
    const float* batch[MAXK];      // base of every mini-batch's block (one allocation each, or slices of one block)
    Layout L;
    int record, two_hops, n_batches, launch, spin;
    float* out;
};

__global__ void __launch_bounds__(1024) k_probe(Args a) {
    extern __shared__ float lds[];
    const int g = blockIdx.x >> 1;                       // two workgroups (branches) read the same graph
    const int b = a.launch % a.n_batches;
    const float* blk = a.batch[b];
    float acc = 0.0f;
    int first = 0;
    if (a.two_hops) {
        // the counts: one word of the LAST array, waited for before anything else is issued
        const float* p = a.record ? blk + (long)g * a.L.rec + a.L.rec_off[NARR - 1] : blk + a.L.base[NARR - 1] + (long)g * a.L.stride[NARR - 1];
        first = (int)__builtin_nontemporal_load(p);
        first = __builtin_amdgcn_readfirstlane(first);
    }
    float4 v[3 + NARR];
    int nv = 0;
    // array 0: the S rows (25.6 KB = 1600 float4: two per lane); the others: at most one float4 per lane
#pragma unroll
    for (int arr = 0; arr < NARR; ++arr) {
        const float* p = a.record ? blk + (long)g * a.L.rec + a.L.rec_off[arr] : blk + a.L.base[arr] + (long)g * a.L.stride[arr];
        const int n4 = (a.L.words[arr] + first) >> 2;
        if (arr == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = threadIdx.x + j * 1024;
                v[nv++] = (q < n4) ? ((const float4*)p)[q] : float4{0, 0, 0, 0};
            }
        } else {
            v[nv++] = ((int)threadIdx.x < n4) ? ((const float4*)p)[threadIdx.x] : float4{0, 0, 0, 0};
        }
    }
#pragma unroll
    for (int i = 0; i < 2 + NARR - 1; ++i) { lds[(i * 1024 + threadIdx.x) & 8191] = v[i].x + v[i].y + v[i].z + v[i].w; }
    __syncthreads();
    acc = lds[(threadIdx.x * 7) & 8191];
    // ~5 us of dependent work behind the staging: the launch is then longer than the command processor's per-node floor
    // (2.8 us for this grid), so that what the staging costs shows in the launch's duration instead of hiding under it
    for (int i = 0; i < a.spin; ++i) acc = __builtin_fmaf(acc, 1.0000001f, 1e-9f);
    if (threadIdx.x == 0) a.out[blockIdx.x] = acc;
}

static double time_us(Args a, int n_wg, int iters) {
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 64; ++i) { a.launch = i; hipLaunchKernelGGL(k_probe, dim3(n_wg), dim3(1024), 32768, s, a); }
    CHECK(hipStreamSynchronize(s));
    // 160 launches per graph (>= one per mini-batch of the longest cycle), replayed
    hipGraph_t gr; hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 160; ++i) { a.launch = i; hipLaunchKernelGGL(k_probe, dim3(n_wg), dim3(1024), 32768, s, a); }
    CHECK(hipStreamEndCapture(s, &gr));
    CHECK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CHECK(hipGraphLaunch(ge, s));
    CHECK(hipStreamSynchronize(s));
    CHECK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) CHECK(hipGraphLaunch(ge, s));
    CHECK(hipEventRecord(e1, s));
    CHECK(hipStreamSynchronize(s));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(gr));
    CHECK(hipStreamDestroy(s));
    return ms * 1e3 / (iters * 160.0);
}

int main() {
    const int G = 64, n_wg = 2 * G, K = 160;
    // S rows | IHORD | HMP0 | MEM1 | MPTR1 | ROWPTR1 | COL1 | COLPTR1 | ROWIDX1 | counts
    const int words[NARR] = {6400, 200, 52, 52, 20, 52, 400, 52, 400, 4};
    Layout L;
    long off = 0, rec = 0;
    for (int a = 0; a < NARR; ++a) {
        L.words[a] = words[a];
        const long w4 = (words[a] + 3) & ~3;
        // scattered: node-indexed arrays are laid out for 200 slots per graph, edge-indexed ones for 1024, whatever they hold
        const long slot = (a == 0) ? 6400 : (a == 6 || a == 8) ? 1024 : (a == NARR - 1) ? 4 : 204;
        L.base[a] = off; L.stride[a] = slot; off += slot * G;
        off = (off + 1023) & ~1023L;                      // (every array starts on a 4 KB boundary)
        L.rec_off[a] = rec; rec += w4;
    }
    L.rec = (rec + 63) & ~63L;                             // 256-byte aligned records
    const long scattered_words = off, record_words = L.rec * G;
    const long bw = ((scattered_words > record_words ? scattered_words : record_words) + (1 << 18)) & ~((1L << 18) - 1);   // 1 MB multiples
    L.batch_words = bw;
    printf("bytes per graph %ld (record), mini-batch block %.2f MB, cycle of %d = %.1f MB\n", L.rec * 4, bw * 4 / 1e6, K, K * bw * 4 / 1e6);
    float* data; float* out;
    CHECK(hipMalloc(&data, (size_t)K * bw * 4));
    CHECK(hipMemset(data, 0, (size_t)K * bw * 4));
    CHECK(hipMalloc(&out, n_wg * 4));
    std::vector<float*> sep(K);
    for (int k = 0; k < K; ++k) { CHECK(hipMalloc(&sep[k], (size_t)bw * 4)); CHECK(hipMemset(sep[k], 0, (size_t)bw * 4)); }
    Args a; a.L = L; a.out = out; a.spin = 100;
    for (int alloc = 0; alloc < 2; ++alloc) {
        for (int k = 0; k < K; ++k) a.batch[k] = alloc ? sep[k] : data + (long)k * bw;
        for (int record = 0; record < 2; ++record)
            for (int cyc = 0; cyc < 5; ++cyc) {
                const int ks[5] = {1, 4, 16, 32, K};
                a.record = record; a.two_hops = 0; a.n_batches = ks[cyc];
                const double us = time_us(a, n_wg, 100);
                printf("%-22s %-9s cycle of %3d (%6.1f MB) %7.2f us per launch\n", alloc ? "one hipMalloc per batch" : "one block", record ? "record" : "scattered", ks[cyc], ks[cyc] * bw * 4 / 1e6, us);
            }
    }
    // the same with only the small arrays (no S rows): what the bulk costs
    L.words[0] = 0; a.L = L;
    for (int record = 0; record < 2; ++record)
        for (int cold = 0; cold < 2; ++cold) {
            a.record = record; a.two_hops = 0; a.n_batches = cold ? K : 1;
            printf("%-10s %-9s %-5s %7.2f us per launch (small arrays only)\n", "one burst", record ? "record" : "scattered", cold ? "cold" : "warm", time_us(a, n_wg, 300));
        }
    // empty kernel floor: zero arrays
    for (int i = 0; i < NARR; ++i) L.words[i] = 0;
    a.L = L; a.record = 1; a.two_hops = 0; a.n_batches = 1;
    printf("%-10s %-9s %-5s %7.2f us per launch (no loads)\n", "-", "-", "-", time_us(a, n_wg, 300));
    return 0;
}
