// drgnn_p2p.h -- one-shot all-reduce of the flat gradient over peer-mapped exchange buffers.
//
// The data-parallel exchange of this path is ONE sum of a ~43 KB fp32 vector over <= 8 GPUs of one node
// (SURVEY.md 8(e)).  A ring all-reduce pays 2 (W-1) latency-bound hops for that; here every rank publishes its
// (weighted) vector in its own fine-grained exchange buffer, mapped into every peer through hipIpc, and every rank
// reads all W vectors over xGMI (point-to-point links, one hop) and adds them IN RANK ORDER -- one xGMI round trip, and
// bit-identical sums on all ranks (the replicas' parameters cannot drift).  No host involvement: the launch is
// hipGraph-capturable, the sequence number lives in device memory.
//
// Exchange buffer of one rank:   [slot 0: n_pad floats][slot 1: n_pad floats][flags: 2 x P2P_WGS uint32][pad]
// Workgroup j owns the slice [j * per, (j+1) * per) of the vector: it publishes that slice in slot (seq & 1), raises
// flag[slot][j] = seq (release, system scope), waits for the same flag of every peer (acquire, system scope; bounded:
// an expired wait sets a status word instead of hanging the queue) and sums the peers' slices.  Two slots: a rank
// that runs ahead publishes step k+1 in the other slot while a slower peer still reads step k; it cannot reach step
// k+2 before that peer has raised its step-k+1 flag, i.e. has finished reading step k.
#pragma once
#include "drgnn_rt.h"

#define DRGNN_P2P_MAX 16          // ranks
#define DRGNN_P2P_WGS 16          // workgroups (= slices) per launch
#define DRGNN_P2P_THREADS 256

struct P2PArgs {
    float* grad;                      // [n] in: this rank's gradient; out: the weighted sum over the ranks
    int64_t n;
    float* peer[DRGNN_P2P_MAX];       // exchange buffers of ranks 0..world-1 as mapped HERE (peer[rank] = own)
    int world, rank;
    float weight;                     // n_local / n_global (1 / world for equal shards)
    uint32_t* seq;                    // [DRGNN_P2P_WGS] device counters: completed exchanges, per workgroup
    int32_t* status;                  // [1] device word: != 0 after an expired wait
    int part;                         // 0: whole exchange; 1: publish only; 2: consume only (single-process tests)
};

HD int64_t p2p_pad(int64_t n) { return (n + 63) & ~(int64_t)63; }
HD int64_t p2p_bytes(int64_t n) { return (2 * p2p_pad(n) + 2 * DRGNN_P2P_WGS + 64) * 4; }

#ifdef DRGNN_EMU
DEV void p2p_store_flag(uint32_t* p, uint32_t v) { *p = v; }
DEV uint32_t p2p_load_flag(const uint32_t* p) { return *p; }
DEV float p2p_load(const float* p) { return *p; }
#else
DEV void p2p_store_flag(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
DEV uint32_t p2p_load_flag(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); }
// a peer's slot is rewritten every other exchange: the read must not be served from a stale local cache line
DEV float p2p_load(const float* p) {
    return __uint_as_float(__hip_atomic_load((const uint32_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}
#endif

DEV void p2p_block(const P2PArgs& a, int j) {
    const int64_t npad = p2p_pad(a.n);
    const int64_t per = (a.n + DRGNN_P2P_WGS - 1) / DRGNN_P2P_WGS;
    const int64_t lo = (int64_t)j * per, hi = (lo + per < a.n) ? lo + per : a.n;
    const uint32_t seq = a.seq[j] + (a.part == 2 ? 0u : 1u);
    const int slot = (int)(seq & 1u);
    if (a.part != 2) {
        float* mine = a.peer[a.rank] + slot * npad;
#ifdef DRGNN_EMU
        for (int64_t i = lo; i < hi; ++i) mine[i] = a.grad[i] * a.weight;
        a.seq[j] = seq;
#else
        for (int64_t i = lo + threadIdx.x; i < hi; i += DRGNN_P2P_THREADS)
            __hip_atomic_store((uint32_t*)(mine + i), __float_as_uint(a.grad[i] * a.weight), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) a.seq[j] = seq;
#endif
        uint32_t* flags = (uint32_t*)(a.peer[a.rank] + 2 * npad);
#ifdef DRGNN_EMU
        p2p_store_flag(flags + slot * DRGNN_P2P_WGS + j, seq);
#else
        if (threadIdx.x == 0) p2p_store_flag(flags + slot * DRGNN_P2P_WGS + j, seq);
#endif
    }
    if (a.part == 1) return;
#ifdef DRGNN_EMU
    for (int r = 0; r < a.world; ++r) {
        const uint32_t* f = (const uint32_t*)(a.peer[r] + 2 * npad) + slot * DRGNN_P2P_WGS + j;
        if (p2p_load_flag(f) != seq) a.status[0] = 1;
    }
    for (int64_t i = lo; i < hi; ++i) {
        float acc = 0.0f;
        for (int r = 0; r < a.world; ++r) acc += p2p_load(a.peer[r] + slot * npad + i);
        a.grad[i] = acc;
    }
#else
    if ((int)threadIdx.x < a.world) {
        const uint32_t* f = (const uint32_t*)(a.peer[threadIdx.x] + 2 * npad) + slot * DRGNN_P2P_WGS + j;
        const unsigned long long t0 = wall_clock64();
        while (p2p_load_flag(f) != seq) {
            if (wall_clock64() - t0 > 200000000ull) { atomicExch(a.status, 1 + (int)threadIdx.x); break; }   // ~2 s at 100 MHz
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    for (int64_t i = lo + threadIdx.x; i < hi; i += DRGNN_P2P_THREADS) {
        float v[DRGNN_P2P_MAX];
#pragma unroll
        for (int r = 0; r < DRGNN_P2P_MAX; ++r) v[r] = (r < a.world) ? p2p_load(a.peer[r] + slot * npad + i) : 0.0f;
        float acc = 0.0f;
#pragma unroll
        for (int r = 0; r < DRGNN_P2P_MAX; ++r) if (r < a.world) acc += v[r];      // rank order: same bits everywhere
        a.grad[i] = acc;
    }
#endif
}
