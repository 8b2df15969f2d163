// drgnn_rt.h -- execution-model glue shared by all kernels.
//
// Kernels are written as a sequence of barrier-separated phases over a workgroup:
//     FOR_TID(i, n) { ... }   work items i = tid, tid + nthreads, ...
//     BARRIER();
// No per-thread value survives a BARRIER except through memory.  That discipline lets the
// SAME source be compiled twice:
//   * hipcc --offload-arch=gfx950 : the product (256-thread workgroups, 64-wide waves,
//     MFMA, LDS);
//   * g++ -DDRGNN_EMU             : a host emulation used ONLY by the CPU test-suite
//     (tests/emu), where a "workgroup" is a plain loop that runs every work item of a
//     phase in order.  It checks index logic without a GPU; it is never loaded by the
//     package.
#pragma once
#include <stdint.h>

#ifdef DRGNN_EMU
// ---------------------------------------------------------------- host emulation
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#define DEV static inline
#define HD static inline
#define FOR_TID(i, n) for (int i = 0; i < (int)(n); ++i)
#define FOR_TID_FROM(i, n, first) for (int i = 0; i < (int)(n); ++i)
#define BARRIER() ((void)0)
#define PHASE_MARK() ((void)0)
#define PHASE_BEGIN() ((void)0)
#define DRGNN_NTHREADS 1024
struct WG { int block; int nthreads; };
#define WG_TID0(wg) (true)
DEV int emu_atomic_add(int* p, int v) { int o = *p; *p = o + v; return o; }
DEV void emu_atomic_or(int* p, int v) { *p |= v; }
DEV void emu_atomic_min64(long long* p, long long v) { if (v < *p) *p = v; }
DEV void emu_atomic_max64(long long* p, long long v) { if (v > *p) *p = v; }
#define ATOMIC_ADD(p, v) emu_atomic_add((p), (v))
#define ATOMIC_OR(p, v) emu_atomic_or((p), (v))
#define ATOMIC_MIN64(p, v) emu_atomic_min64((p), (v))
#define ATOMIC_MAX64(p, v) emu_atomic_max64((p), (v))
DEV void emu_atomic_add64(long long* p, long long v) { *p += v; }
DEV void emu_atomic_max(int* p, int v) { if (v > *p) *p = v; }
#define ATOMIC_ADD64(p, v) emu_atomic_add64((p), (v))
#define ATOMIC_MAX(p, v) emu_atomic_max((p), (v))
typedef void* drgnn_stream_t;
#else
// ---------------------------------------------------------------- gfx950
#include <hip/hip_runtime.h>
#include <limits.h>
#define DEV __device__ __forceinline__
#define HD __host__ __device__ static inline
#ifndef DRGNN_NTHREADS
#define DRGNN_NTHREADS 1024
#endif
// (nounroll: these loops run once per workgroup with 1 - 3 trips; unrolled copies and their remainder loops are instructions for nothing)
#define FOR_TID(i, n) _Pragma("nounroll") for (int i = (int)threadIdx.x; i < (int)(n); i += DRGNN_NTHREADS)
// the same with item 0 on thread `first` (a multiple of the wave size): several short loops of one phase land on different
// waves instead of queueing up on waves 0, 1, ..
#define FOR_TID_FROM(i, n, first) _Pragma("nounroll") \
    for (int i = (int)((threadIdx.x + DRGNN_NTHREADS - (first)) & (DRGNN_NTHREADS - 1)); i < (int)(n); i += DRGNN_NTHREADS)
#ifdef DRGNN_PHASE_TIMING
// profiling build only (libdrgnn_prof.so, tools/phase_timing.py): thread 0 of workgroup 0
// stamps (source line, shader clock) after every barrier into a global buffer.
__device__ unsigned long long* g_phase_buf = nullptr;
// the mark counter lives in LDS (set by PHASE_BEGIN at kernel entry) so that a mark costs one
// scalar pointer load + s_memtime + a fire-and-forget store, not two global round trips
__shared__ unsigned int drgnn_phase_k;
// which workgroup of a launch stamps: word 1 of the buffer, set by the host (e.g. 8: the second builder workgroup of graph 0)
__device__ __forceinline__ void phase_mark(int line) {
    if (threadIdx.x == 0 && g_phase_buf != nullptr && blockIdx.x == (unsigned)g_phase_buf[1]) {
        const unsigned int k = drgnn_phase_k;
        if (k < 2000) {
            g_phase_buf[2 + 2 * k] = (unsigned long long)line;
            g_phase_buf[3 + 2 * k] = clock64();
            g_phase_buf[0] = k + 1;
            drgnn_phase_k = k + 1;
        }
    }
}
#define PHASE_BEGIN() do { if (threadIdx.x == 0) drgnn_phase_k = 0; } while (0)
#define BARRIER() do { __syncthreads(); phase_mark(__LINE__); } while (0)
#define PHASE_MARK() phase_mark(__LINE__)
#else
#define BARRIER() __syncthreads()
#define PHASE_MARK() ((void)0)
#define PHASE_BEGIN() ((void)0)
#endif
struct WG { int block; int nthreads; };
#define ATOMIC_ADD(p, v) atomicAdd((p), (v))
#define ATOMIC_OR(p, v) atomicOr((p), (v))
#define ATOMIC_MIN64(p, v) atomicMin((p), (long long)(v))
#define ATOMIC_MAX64(p, v) atomicMax((p), (long long)(v))
#define ATOMIC_ADD64(p, v) atomicAdd((unsigned long long*)(p), (unsigned long long)(v))
#define ATOMIC_MAX(p, v) atomicMax((p), (v))
typedef hipStream_t drgnn_stream_t;
#endif

// row * stride for LDS addresses (rows < 2^23): a full-rate 24-bit multiply-add instead of the quarter-rate 32-bit one the
// compiler picks for strides that are not powers of two (the product sits between an index read and the read it addresses)
#ifdef DRGNN_EMU
#define ROW24(r, ld) ((r) * (ld))
#else
#define ROW24(r, ld) __mul24((int)(r), (int)(ld))
#endif
#define DRGNN_WAVE 64
#define DRGNN_BSCALE (1024 / DRGNN_NTHREADS)   // burst capacities are quoted per 1024 lanes
#define DRGNN_BCAP 1024
#define DRGNN_NWAVES (DRGNN_NTHREADS / DRGNN_WAVE)

// Division by a run-time constant without the ~40-instruction integer divide: one real
// division per kernel (the magic), then a mul-hi per use.  Exact while n * d < 2^32.
struct FastDiv {
    unsigned d, m;
};
DEV FastDiv fastdiv_make(int d) {
    FastDiv f;
    f.d = (unsigned)(d > 0 ? d : 1);
    f.m = (unsigned)(0xFFFFFFFFu / f.d) + 1u;
    return f;
}
#ifdef DRGNN_EMU
DEV int fastdiv(const FastDiv& f, int n) { return (int)((unsigned)n / f.d); }
#else
DEV int fastdiv(const FastDiv& f, int n) { return f.d == 1u ? n : (int)__umulhi((unsigned)n, f.m); }
#endif
DEV int fastmod(const FastDiv& f, int n, int q) { return n - q * (int)f.d; }

DEV int imin(int a, int b) { return a < b ? a : b; }
DEV int imax(int a, int b) { return a > b ? a : b; }

// ---------------------------------------------------------------------------------
// Workgroup-wide exclusive scan, in place, of a[0..n).  `part` is DRGNN_NTHREADS+1 ints of
// workgroup-visible scratch.  Returns the total in part[DRGNN_NTHREADS] (valid after the
// call for every thread).  Three barriers.
// ---------------------------------------------------------------------------------
#ifdef DRGNN_EMU
DEV int wg_exscan(int* a, int n, int* part) {
    int run = 0;
    for (int i = 0; i < n; ++i) { int v = a[i]; a[i] = run; run += v; }
    part[DRGNN_NTHREADS] = run;
    return run;
}
// two independent exclusive scans in the barrier intervals of one (totals: *ta, *tb)
DEV void wg_exscan2(int* a, int na, int* b, int nb, int* part, int* ta, int* tb) {
    *ta = wg_exscan(a, na, part);
    *tb = wg_exscan(b, nb, part);
}
#else
// inclusive scan over the 64 lanes of a wave on the DPP path (no LDS round trips): Hillis-Steele inside the rows of 16
// lanes (row_shr 1, 2, 4, 8; lanes without a source add 0), then lane 15 / lane 31 broadcast into the following rows
template <int CTRL, int ROWMASK> DEV int dpp_add_i(int v) {
    return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROWMASK, 0xF, true);
}
DEV int wave_incl_scan(int v) {
    v = dpp_add_i<0x111, 0xF>(v);
    v = dpp_add_i<0x112, 0xF>(v);
    v = dpp_add_i<0x114, 0xF>(v);
    v = dpp_add_i<0x118, 0xF>(v);
    v = dpp_add_i<0x142, 0xA>(v);      // row_bcast:15 into rows 1 and 3
    v = dpp_add_i<0x143, 0xC>(v);      // row_bcast:31 into rows 2 and 3
    return v;
}
DEV int wg_exscan(int* a, int n, int* part) {
    const int t = threadIdx.x;
    if (n <= DRGNN_NTHREADS) {
        // one element per lane: wave-level inclusive scan, the totals of the waves that HOLD elements go through LDS and
        // every lane adds the totals of the waves before its own -> 2 barriers.  Waves past the end only pass the barriers
        // and add up the totals (an instruction of any wave occupies its SIMD for 4 cycles: typical scans here are a few
        // hundred elements, 3 - 4 of the 16 waves).
        const int lane = t & (DRGNN_WAVE - 1), wave = t >> 6;
        const int nw = (n + DRGNN_WAVE - 1) / DRGNN_WAVE;
        int v = 0, inc = 0;
        if (wave < nw) {
            v = (t < n) ? a[t] : 0;
            inc = wave_incl_scan(v);
            if (lane == DRGNN_WAVE - 1) part[wave] = inc;
        }
        __syncthreads();
        int base = 0, total = 0;
        for (int w = 0; w < nw; ++w) {
            const int tw = part[w];
            base += (w < wave) ? tw : 0;
            total += tw;
        }
        if (t < n) a[t] = base + inc - v;
        __syncthreads();
        return total;
    }
    const int chunk = (n + DRGNN_NTHREADS - 1) / DRGNN_NTHREADS;
    const int lo = imin(t * chunk, n), hi = imin(lo + chunk, n);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += a[i];
    part[t] = s;
    __syncthreads();
    if (t < DRGNN_WAVE) {
        constexpr int PER = DRGNN_NTHREADS / DRGNN_WAVE;
        int v[PER];
        int loc = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) { v[j] = part[t * PER + j]; loc += v[j]; }
        int inc = loc;
#pragma unroll
        for (int d = 1; d < DRGNN_WAVE; d <<= 1) {
            int o = __shfl_up(inc, d, DRGNN_WAVE);
            if (t >= d) inc += o;
        }
        int run = inc - loc;
#pragma unroll
        for (int j = 0; j < PER; ++j) { part[t * PER + j] = run; run += v[j]; }
        if (t == DRGNN_WAVE - 1) part[DRGNN_NTHREADS] = inc;
    }
    __syncthreads();
    int run = part[t];
    for (int i = lo; i < hi; ++i) { int v = a[i]; a[i] = run; run += v; }
    const int total = part[DRGNN_NTHREADS];
    __syncthreads();
    return total;
}
// two independent exclusive scans in the barrier intervals of one (totals: *ta, *tb): array a on the waves from 0, array b on the
// waves from DRGNN_NWAVES / 2 (both at most half a workgroup of elements; anything longer: one scan after the other)
DEV void wg_exscan2(int* a, int na, int* b, int nb, int* part, int* ta, int* tb) {
    constexpr int HALF = DRGNN_NTHREADS / 2, HW = HALF / DRGNN_WAVE;
    if (na > HALF || nb > HALF) { *ta = wg_exscan(a, na, part); *tb = wg_exscan(b, nb, part); return; }
    const int t = threadIdx.x, lane = t & (DRGNN_WAVE - 1), wave = t >> 6;
    const bool second = t >= HALF;
    int* arr = second ? b : a;
    const int n = second ? nb : na, i = second ? t - HALF : t;
    const int nwa = (na + DRGNN_WAVE - 1) / DRGNN_WAVE, nwb = (nb + DRGNN_WAVE - 1) / DRGNN_WAVE;
    int v = 0, inc = 0;
    if (i < ((n + DRGNN_WAVE - 1) & ~(DRGNN_WAVE - 1))) {
        v = (i < n) ? arr[i] : 0;
        inc = wave_incl_scan(v);
        if (lane == DRGNN_WAVE - 1) part[wave] = inc;
    }
    __syncthreads();
    int base = 0, tot_a = 0, tot_b = 0;
    for (int w = 0; w < nwa; ++w) { const int tw = part[w]; base += (!second && w < wave) ? tw : 0; tot_a += tw; }
    for (int w = 0; w < nwb; ++w) { const int tw = part[HW + w]; base += (second && HW + w < wave) ? tw : 0; tot_b += tw; }
    if (i < n) arr[i] = base + inc - v;
    __syncthreads();
    *ta = tot_a; *tb = tot_b;
}
#endif

// ---------------------------------------------------------------------------------
// Burst staging (global -> registers -> LDS).  A kernel prologue that copies a dozen small
// arrays with one loop each pays one full memory latency PER ARRAY (the store of loop k
// waits for the load of loop k before loop k+1 may issue).  Instead every array is first
// loaded into registers (burst_load: J elements per lane, all loads in flight together) and
// only then written to LDS (burst_store).  Requires n <= J * DRGNN_NTHREADS.
// ---------------------------------------------------------------------------------
#ifdef DRGNN_EMU
template <class T, int J> struct Burst { const T* src; int n; };
template <class T, int J> DEV void burst_load(Burst<T, J>& b, const T* src, int n) { b.src = src; b.n = src ? n : 0; }
template <class T, int J> DEV void burst_store(const Burst<T, J>& b, T* dst) { for (int i = 0; i < b.n; ++i) dst[i] = b.src[i]; }
template <int J> struct BurstW { const float* src; long sk, sh; int K, H; };
template <int J> DEV void burst_load_w(BurstW<J>& b, const float* src, long sk, long sh, int K, int H) {
    b.src = src; b.sk = sk; b.sh = sh; b.K = src ? K : 0; b.H = H;
}
template <int J> DEV void burst_store_w(const BurstW<J>& b, float* dst, int ld) {
    for (int k = 0; k < b.K; ++k) for (int h = 0; h < b.H; ++h) dst[k * ld + h] = b.src[k * b.sk + h * b.sh];
}
template <int J> DEV void burst_store_wt(const BurstW<J>& b, float* dst, int ld) {     // dst[h*ld + k]
    for (int k = 0; k < b.K; ++k) for (int h = 0; h < b.H; ++h) dst[h * ld + k] = b.src[k * b.sk + h * b.sh];
}
template <int J> struct BurstX { const float* src; int rows, F; };
template <int J> DEV void burst_load_x(BurstX<J>& b, const float* src, int rows, int F) { b.src = src; b.rows = rows; b.F = F; }
template <int J> DEV void burst_store_x(const BurstX<J>& b, float* dst) {
    for (int i = 0; i < b.rows; ++i) for (int f = 0; f < b.F; ++f) dst[i * (b.F + 1) + f] = b.src[i * b.F + f];
}
#else
template <class T, int J> struct Burst { T v[J * DRGNN_BSCALE]; int n; };
template <class T, int J> DEV void burst_load(Burst<T, J>& b, const T* src, int n) {
    b.n = src ? n : 0;
#pragma unroll
    for (int j = 0; j < J * DRGNN_BSCALE; ++j) {
        const int i = threadIdx.x + j * DRGNN_NTHREADS;
        b.v[j] = (i < b.n) ? src[i] : T(0);
    }
}
template <class T, int J> DEV void burst_store(const Burst<T, J>& b, T* dst) {
#pragma unroll
    for (int j = 0; j < J * DRGNN_BSCALE; ++j) {
        const int i = threadIdx.x + j * DRGNN_NTHREADS;
        if (i < b.n) dst[i] = b.v[j];
    }
}
// strided [K,H] weight matrix -> dense padded rows dst[k*ld + h]
template <int J> struct BurstW { float v[J * DRGNN_BSCALE]; int n, H; };
template <int J> DEV void burst_load_w(BurstW<J>& b, const float* src, long sk, long sh, int K, int H) {
    b.n = src ? K * H : 0; b.H = H;
    const FastDiv fd = fastdiv_make(H);
#pragma unroll
    for (int j = 0; j < J * DRGNN_BSCALE; ++j) {
        const int e = threadIdx.x + j * DRGNN_NTHREADS;
        const int k = fastdiv(fd, e), h = fastmod(fd, e, k);
        b.v[j] = (e < b.n) ? src[(long)k * sk + (long)h * sh] : 0.0f;
    }
}
template <int J> DEV void burst_store_w(const BurstW<J>& b, float* dst, int ld) {
    const FastDiv fd = fastdiv_make(b.H);
#pragma unroll
    for (int j = 0; j < J * DRGNN_BSCALE; ++j) {
        const int e = threadIdx.x + j * DRGNN_NTHREADS;
        const int k = fastdiv(fd, e), h = fastmod(fd, e, k);
        if (e < b.n) dst[k * ld + h] = b.v[j];
    }
}
template <int J> DEV void burst_store_wt(const BurstW<J>& b, float* dst, int ld) {     // dst[h*ld + k]
    const FastDiv fd = fastdiv_make(b.H);
#pragma unroll
    for (int j = 0; j < J * DRGNN_BSCALE; ++j) {
        const int e = threadIdx.x + j * DRGNN_NTHREADS;
        const int k = fastdiv(fd, e), h = fastmod(fd, e, k);
        if (e < b.n) dst[h * ld + k] = b.v[j];
    }
}
// x tile [rows, F] (F % 4 == 0, 16-byte aligned) -> padded rows dst[i*(F+1) + f], float4 loads
typedef float drgnn_f4 __attribute__((ext_vector_type(4)));
template <int J> struct BurstX { drgnn_f4 v[J * DRGNN_BSCALE]; int n4, F; };
template <int J> DEV void burst_load_x(BurstX<J>& b, const float* src, int rows, int F) {
    b.n4 = rows * F / 4; b.F = F;
    const drgnn_f4* s4 = (const drgnn_f4*)src;
#pragma unroll
    for (int j = 0; j < J * DRGNN_BSCALE; ++j) {
        if (j > 0 && b.n4 <= j * DRGNN_NTHREADS) break;          // uniform: the tile ends before this chunk
        const int q = threadIdx.x + j * DRGNN_NTHREADS;
        b.v[j] = (q < b.n4) ? s4[q] : drgnn_f4{0.f, 0.f, 0.f, 0.f};
    }
}
template <int J> DEV void burst_store_x(const BurstX<J>& b, float* dst) {
    const FastDiv fd = fastdiv_make(b.F);
#pragma unroll
    for (int j = 0; j < J * DRGNN_BSCALE; ++j) {
        const int q = threadIdx.x + j * DRGNN_NTHREADS;
        if (q < b.n4) {
            const int e = q * 4;
            const int row = fastdiv(fd, e);
            float* d = dst + row * (b.F + 1) + fastmod(fd, e, row);
            d[0] = b.v[j][0]; d[1] = b.v[j][1]; d[2] = b.v[j][2]; d[3] = b.v[j][3];
        }
    }
}
#endif

#ifndef DRGNN_EMU
// buffer resource of a global array (gfx950 `buffer_load ... offen`): the hardware range check of the descriptor returns 0
// for words past the end, so loads need no per-lane compare / exec mask
DEV __amdgpu_buffer_rsrc_t buf_rsrc(const void* p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
#endif

// ---------------------------------------------------------------------------------
// Wave-specialised staging of the SMALL arrays (offset tables, index lists, bias vectors): one staging job = one array
// (or half of one) of at most 1024 32-bit words, handled by ONE wave with up to four 128-bit buffer loads (64 lanes x 4
// words x 4) that stay in its registers until the LDS store one phase later.  Why: every instruction that all 16 waves of
// a workgroup execute occupies every SIMD (4 waves each, no issue slack) whether its lanes hold data or not; a dozen arrays of a few
// hundred words each, staged by all waves, cost ~150 instructions per wave and burst -- as one job per wave they cost ~25.
// `src` / `n` / `dst` are wave-uniform.  dst is 16-byte aligned and padded to a multiple of 4 words (step_carve); the
// source needs 4-byte alignment only (the hardware range check is per word: tools/probes/buffer_x4_range_probe.hip).
// ---------------------------------------------------------------------------------
struct StageJob { const void* src; int n; void* dst; int narrow; };      // narrow: store the words as 16-bit values
// the two halves of an array too long for one job (n <= 2048): [0, h) and [h, n), h a multiple of 4
DEV StageJob stage_half(StageJob j, int which) {
    int h = ((j.n + 7) >> 3) << 2;
    if (h > j.n) h = j.n;
    if (which == 0) { j.n = h; return j; }
    j.src = (const int32_t*)j.src + h;
    j.dst = j.narrow ? (void*)((unsigned short*)j.dst + h) : (void*)((int32_t*)j.dst + h);
    j.n -= h;
    return j;
}
#ifdef DRGNN_EMU
struct WaveStage { int dummy; };
DEV void stage_copy(const StageJob& j) {
    for (int i = 0; i < j.n; ++i) {
        const int32_t v = ((const int32_t*)j.src)[i];
        if (j.narrow) ((unsigned short*)j.dst)[i] = (unsigned short)v;
        else ((int32_t*)j.dst)[i] = v;
    }
}
#else
typedef int drgnn_i4 __attribute__((ext_vector_type(4)));
struct WaveStage { drgnn_i4 v[4]; int n; void* dst; int narrow; };
DEV void wstage_load(WaveStage& w, const StageJob& j) {
    w.n = j.src ? j.n : 0; w.dst = j.dst; w.narrow = j.narrow;
    const __amdgpu_buffer_rsrc_t r = buf_rsrc(j.src, w.n * 4);
    const int voff = (threadIdx.x & 63) * 16;
#pragma unroll
    for (int it = 0; it < 4; ++it)
        if (it * 256 < w.n) w.v[it] = __builtin_amdgcn_raw_buffer_load_b128(r, voff, it * 1024, 0);
}
DEV void wstage_store(const WaveStage& w) {
    const int lane4 = (threadIdx.x & 63) * 4;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        if (it * 256 >= w.n) break;
        const int e = it * 256 + lane4;
        if (e < w.n) {
            if (w.narrow) {
                const unsigned int lo = ((unsigned int)w.v[it][0] & 0xffffu) | ((unsigned int)w.v[it][1] << 16);
                const unsigned int hi = ((unsigned int)w.v[it][2] & 0xffffu) | ((unsigned int)w.v[it][3] << 16);
                typedef unsigned int drgnn_u2 __attribute__((ext_vector_type(2)));
                *(drgnn_u2*)((unsigned short*)w.dst + e) = drgnn_u2{lo, hi};
            } else {
                *(drgnn_i4*)((int32_t*)w.dst + e) = w.v[it];
            }
        }
    }
}
#endif

// x tile with rows padded to `ld` floats (ld % 4 == 0: 16-byte aligned rows, one 128-bit LDS store per
// float4; ld = 36 keeps the 16 row lanes of an MFMA A-operand read on distinct banks)
#ifdef DRGNN_EMU
template <int J> DEV void burst_store_x4(const BurstX<J>& b, float* dst, int ld) {
    for (int i = 0; i < b.rows; ++i) for (int f = 0; f < b.F; ++f) dst[i * ld + f] = b.src[i * b.F + f];
}
#else
template <int J> DEV void burst_store_x4(const BurstX<J>& b, float* dst, int ld) {
    const FastDiv fd = fastdiv_make(b.F >> 2);
#pragma unroll
    for (int j = 0; j < J * DRGNN_BSCALE; ++j) {
        if (j > 0 && b.n4 <= j * DRGNN_NTHREADS) break;
        const int q = threadIdx.x + j * DRGNN_NTHREADS;
        if (q < b.n4) {
            const int row = fastdiv(fd, q);
            *(drgnn_f4*)(dst + row * ld + 4 * fastmod(fd, q, row)) = b.v[j];
        }
    }
}

// ---- cross-lane sums on the DPP path (no LDS round trip): fixed combination order ---------------
template <int CTRL> DEV float dpp_take(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
// sum over aligned groups of 8 / 16 lanes, every lane of the group gets the result
DEV float lanes8_sum(float v) {
    v += dpp_take<0xB1>(v);     // quad_perm [1,0,3,2]
    v += dpp_take<0x4E>(v);     // quad_perm [2,3,0,1]
    v += dpp_take<0x141>(v);    // row_half_mirror
    return v;
}
DEV float lanes16_sum(float v) {
    v = lanes8_sum(v);
    v += dpp_take<0x140>(v);    // row_mirror
    return v;
}
DEV float lane_get(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
DEV float lanes32_sum(float v) {
    v = lanes16_sum(v);
    const float lo = lane_get(v, 0) + lane_get(v, 16), hi = lane_get(v, 32) + lane_get(v, 48);
    return (threadIdx.x & 32) ? hi : lo;
}
DEV float lanes64_sum(float v) {
    v = lanes16_sum(v);
    return (lane_get(v, 0) + lane_get(v, 16)) + (lane_get(v, 32) + lane_get(v, 48));
}
// min / max of a 64-bit value over the wave, every lane gets both: four DPP steps inside the rows of 16 lanes (two 32-bit
// moves + a 64-bit compare + two selects each), then the four row results meet through v_readlane -- against six
// ds_bpermute round trips per 32-bit half with __shfl_xor
template <int CTRL> DEV long long dpp_take_i64(long long v) {
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned long long)v, CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)((unsigned long long)v >> 32), CTRL, 0xF, 0xF, false);
    return (long long)(((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo);
}
DEV long long lane_get_i64(long long v, int lane) {
    const int lo = __builtin_amdgcn_readlane((int)(unsigned long long)v, lane);
    const int hi = __builtin_amdgcn_readlane((int)((unsigned long long)v >> 32), lane);
    return (long long)(((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo);
}
DEV void wave_minmax_i64(long long& lo, long long& hi) {
#define DRGNN_MM_STEP(CTRL) { const long long ol = dpp_take_i64<CTRL>(lo), oh = dpp_take_i64<CTRL>(hi); \
                              lo = ol < lo ? ol : lo; hi = oh > hi ? oh : hi; }
    DRGNN_MM_STEP(0xB1) DRGNN_MM_STEP(0x4E) DRGNN_MM_STEP(0x141) DRGNN_MM_STEP(0x140)
#undef DRGNN_MM_STEP
    long long l = lane_get_i64(lo, 0), h = lane_get_i64(hi, 0);
#pragma unroll
    for (int r = 16; r < 64; r += 16) {
        const long long ol = lane_get_i64(lo, r), oh = lane_get_i64(hi, r);
        l = ol < l ? ol : l; h = oh > h ? oh : h;
    }
    lo = l; hi = h;
}
DEV float lanes64_max(float v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { const float o = __shfl_xor(v, m, 64); v = o > v ? o : v; }
    return v;
}
#endif
