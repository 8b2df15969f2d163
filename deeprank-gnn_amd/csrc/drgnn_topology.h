// drgnn_topology.h -- per-graph topology construction (one workgroup per graph).
//
// Everything the convolutions and poolings need that depends only on the index tensors:
//   CSR0/CSC0 of the input graph, consecutive depth-0 clusters + member lists,
//   pooled graph CSR1/CSC1 (pool_edge: relabel, drop self loops, sort, merge duplicates
//   with summed edge_attr), consecutive depth-1 clusters + member lists.
// Deterministic: integer atomics are only used where the result is order-independent
// (histograms, slot claiming followed by a rank sort on unique keys); every float sum has
// a fixed order.
#pragma once
#include "drgnn_rt.h"
#include "../../include/drgnn.h"

// element offset of the padded X copy inside an aggregation-tiles buffer laid out for n nodes (include/drgnn.h, DRGNN_TOPO_TILES:
// S [n][TF] | D [n] | C [n] | X [n][TF]): behind D and C, rounded up to a multiple of 4 floats so that X rows are 16-byte aligned
// whatever the parity of n (the step kernels read them with 128-bit requests)
HD long long drgnn_tiles_x_off(long long n, int TF) { return n * TF + ((2 * n + 3) & ~3LL); }

#ifdef DRGNN_EMU
#define LAMBDA_DEV
#else
#define LAMBDA_DEV __device__
#endif

// ---- workspace layout (host + device) ---------------------------------------------
struct TopoLayout {
    int64_t i32[DRGNN_TI_COUNT + 1];
    int64_t f32[DRGNN_TF_COUNT + 1];
};

static inline void topo_layout(int64_t N, int64_t E, int64_t B, TopoLayout* L) {
    int64_t o = 0;
    auto take = [&](int which, int64_t n) { L->i32[which] = o; o += (n + 3) & ~(int64_t)3; };
    take(DRGNN_TI_NPTR, B + 1);
    take(DRGNN_TI_EPTR, B + 1);
    take(DRGNN_TI_ROWPTR0, N + B);
    take(DRGNN_TI_COL0, E);
    take(DRGNN_TI_EID0, E);
    take(DRGNN_TI_COLPTR0, N + B);
    take(DRGNN_TI_ROWIDX0, E);
    take(DRGNN_TI_TSLOT0, E);
    take(DRGNN_TI_CL0, N);
    take(DRGNN_TI_NC0, B);
    take(DRGNN_TI_MPTR0, N + B);
    take(DRGNN_TI_MEM0, N);
    take(DRGNN_TI_ROWPTR1, N + B);
    take(DRGNN_TI_COL1, E);
    take(DRGNN_TI_NE1, B);
    take(DRGNN_TI_COLPTR1, N + B);
    take(DRGNN_TI_ROWIDX1, E);
    take(DRGNN_TI_TSLOT1, E);
    take(DRGNN_TI_CL1, N);
    take(DRGNN_TI_NC1, B);
    take(DRGNN_TI_MPTR1, N + B);
    take(DRGNN_TI_MEM1, N);
    take(DRGNN_TI_CPTR0, B + 1);
    take(DRGNN_TI_E1PTR, B + 1);
    take(DRGNN_TI_CPTR1, B + 1);
    take(DRGNN_TI_ERR, 4);
    take(DRGNN_TI_GSTAT, 2 * B);
    take(DRGNN_TI_HORD, N);
    take(DRGNN_TI_HMP0, N + B);
    take(DRGNN_TI_HSPLIT, 4 * B);
    take(DRGNN_TI_IHORD, N);
    L->i32[DRGNN_TI_COUNT] = o;
    int64_t f = 0;
    L->f32[DRGNN_TF_W0] = f; f += (E + 3) & ~(int64_t)3;
    L->f32[DRGNN_TF_W1] = f; f += (E + 3) & ~(int64_t)3;
    L->f32[DRGNN_TF_COUNT] = f;
}

// Device-side view of the workspace.
struct TopoView {
    int32_t* p[DRGNN_TI_COUNT];
    float* w0;
    float* w1;
};

static inline TopoView topo_view(int32_t* ws_i32, float* ws_f32, const TopoLayout& L) {
    TopoView v;
    for (int k = 0; k < DRGNN_TI_COUNT; ++k) v.p[k] = ws_i32 + L.i32[k];
    v.w0 = ws_f32 ? ws_f32 + L.f32[DRGNN_TF_W0] : nullptr;
    v.w1 = ws_f32 ? ws_f32 + L.f32[DRGNN_TF_W1] : nullptr;
    return v;
}

// ---- scratch carving for one graph -------------------------------------------------
// capT >= max(capN, capE) + 1 ; capF >= capN + capE + 2 (cluster-id flag range)
struct TopoScratch {
    long long* mm;   // [2] min / max of the cluster ids
    int* part;       // [NTHREADS + 1]
    int* rp;         // [capN+1] rowptr0
    int* cp;         // [capN+1] colptr (level 0 then level 1)
    int* cur;        // [capN+1]
    int* cl;         // [capN]
    int* mp;         // [capN+1]
    int* mem;        // [capN]
    int* pp;         // [capN+1]
    int* nb;         // [capN]
    int* rp1;        // [capN+1]
    int* col;        // [capE]
    int* seg;        // [capE]
    int* col1;       // [capE]
    int* er;         // [capE] local row of every edge (staged once from the int64 edge_index)
    int* ec;         // [capE] local col
    float* w0;       // [capE] edge_attr in CSR0 slot order
    float* wr;       // [capE] lean rows chain: edge weight by CSR0 slot (for the aggregation tiles)
    int* t1;         // [capT] x5
    int* t2;
    int* t3;
    int* t4;
    int* t5;
    int* fl;         // [capF]
    float* xs;       // x tile of the graph for the aggregation tiles ([capN][tile_f + 4] floats behind the carve; null: none)
    int capF;
    int capT;        // ints in each of t1..t5
};

#define TOPO_PAD4(n) (((n) + 3) & ~3LL)
// mm[0], mm[1]: min / max of the ids being ranked; mm[2 + 2w], mm[3 + 2w]: wave w's partial min / max (wg_rank_prepare)
// (+ a second set of wave slots, mm[2 + 2 NW + 2w ..]: the depth-1 ids of the lean clusters chain, prepared in the same phase)
// (+ one word: bits of the largest |edge weight| of the graph, the fixed-point scale of the pooled weight sums)
#define TOPO_MM_INTS (8 + 8 * (DRGNN_NTHREADS / DRGNN_WAVE))
#define TOPO_MM1(s) ((s).mm + 2 + 2 * (DRGNN_NTHREADS / DRGNN_WAVE))
#define TOPO_WMAX(s) ((int*)(s).mm + 4 + 8 * (DRGNN_NTHREADS / DRGNN_WAVE))
// (+ one word: a cluster id of the lean clusters chain lay outside its flag array)
#define TOPO_OVF(s) ((int*)(s).mm + 5 + 8 * (DRGNN_NTHREADS / DRGNN_WAVE))
// number of ints: linear in (capN, capE, capT, capF) -- keep in sync with topo_carve
HD int64_t topo_scratch_ints(int64_t capN, int64_t capE, int64_t capT, int64_t capF) {
    return TOPO_MM_INTS + TOPO_PAD4(DRGNN_NTHREADS + 1) + 6 * TOPO_PAD4(capN + 1) + 3 * TOPO_PAD4(capN) +
           7 * TOPO_PAD4(capE) + 5 * TOPO_PAD4(capT) + TOPO_PAD4(capF);
}

// Global-memory placement of graph g's scratch when it does not live in LDS.  With
// capT = N+E+1 and capF = N+E+2 the carve needs at most 15*N + 13*E + TOPO_GSCRATCH_CONST ints
// (the constant absorbs the fixed arrays and every PAD4 rounding), so regions placed at
// 15*n0 + 13*e0 + CONST*g (rounded up to even for the 64-bit min/max slot) never overlap.
#define TOPO_GSCRATCH_CONST (DRGNN_NTHREADS + 188 + 8 * (DRGNN_NTHREADS / DRGNN_WAVE))
HD int64_t topo_gscratch_base(int64_t n0, int64_t e0, int64_t g) {
    return ((15 * n0 + 13 * e0 + (int64_t)TOPO_GSCRATCH_CONST * g) + 1) & ~(int64_t)1;
}

template <class IntPtr>
DEV TopoScratch topo_carve(IntPtr base, int capN, int capE, int capT, int capF) {
    TopoScratch s;
    int o = 0;
    s.mm = (long long*)(base + o); o += TOPO_MM_INTS;
    s.part = base + o; o += (int)TOPO_PAD4(DRGNN_NTHREADS + 1);
    s.rp = base + o;   o += (int)TOPO_PAD4(capN + 1);
    s.cp = base + o;   o += (int)TOPO_PAD4(capN + 1);
    s.cur = base + o;  o += (int)TOPO_PAD4(capN + 1);
    s.mp = base + o;   o += (int)TOPO_PAD4(capN + 1);
    s.pp = base + o;   o += (int)TOPO_PAD4(capN + 1);
    s.rp1 = base + o;  o += (int)TOPO_PAD4(capN + 1);
    s.cl = base + o;   o += (int)TOPO_PAD4(capN);
    s.mem = base + o;  o += (int)TOPO_PAD4(capN);
    s.nb = base + o;   o += (int)TOPO_PAD4(capN);
    s.col = base + o;  o += (int)TOPO_PAD4(capE);
    s.seg = base + o;  o += (int)TOPO_PAD4(capE);
    s.col1 = base + o; o += (int)TOPO_PAD4(capE);
    s.er = base + o;   o += (int)TOPO_PAD4(capE);
    s.ec = base + o;   o += (int)TOPO_PAD4(capE);
    s.w0 = (float*)(base + o); o += (int)TOPO_PAD4(capE);
    s.wr = (float*)(base + o); o += (int)TOPO_PAD4(capE);
    s.t1 = base + o;   o += (int)TOPO_PAD4(capT);
    s.t2 = base + o;   o += (int)TOPO_PAD4(capT);
    s.t3 = base + o;   o += (int)TOPO_PAD4(capT);
    s.t4 = base + o;   o += (int)TOPO_PAD4(capT);
    s.t5 = base + o;   o += (int)TOPO_PAD4(capT);
    s.fl = base + o;   o += (int)TOPO_PAD4(capF);
    s.xs = nullptr;
    s.capF = capF;
    s.capT = capT;
    return s;
}

DEV int topo_f2i(float v) { int b; memcpy(&b, &v, 4); return b; }
DEV float topo_i2f(int b) { float v; memcpy(&v, &b, 4); return v; }
// Pooled edge weights are sums of raw edge weights; they are formed in 64-bit FIXED POINT at a scale set by the largest |w|
// of the graph (2^40 steps per binade of it: < 2^-40 of that weight per addend, no overflow below 2^16 edges): exact, hence
// independent of the order of the addends -- the lean chain adds with atomics, the general chain along sorted runs, both
// get the same bits.
DEV double topo_wscale(int wmax_bits, double* inv) {
    int ex = 0;
    (void)frexpf(topo_i2f(wmax_bits), &ex);
    *inv = ldexp(1.0, ex - 40);
    return ldexp(1.0, 40 - ex);
}
DEV long long topo_wfix(float w, double scale) { return (long long)rint((double)w * scale); }

// per-graph status word: cleared by the graph's own workgroup at the start of every build
DEV void topo_flag(const TopoView& tv, int bit, int graph) { ATOMIC_OR(&tv.p[DRGNN_TI_GSTAT][graph], bit); }

// ---------------------------------------------------------------------------------
// Stable bucket sort of items 0..n-1 by bucket_of(item) in [0, nb):
//   ptr[0..nb]   bucket offsets            order[p]  item at sorted position p
//   slot_bucket[p] bucket of sorted position p
// Items inside a bucket keep ascending item order (rank sort on the unique item id).
// tmp: n ints.  cur: nb+1 ints.

// number of entries of a[lo, hi) below `me`, four entries per trip in flight: for LONG buckets (the 20 - 60 edges that leave one
// cluster) -- for the 5 - 10 entries of a node's edge list the plain loop is faster (measured: +0.9 us on the unweighted builder)
DEV int rank_below(const int* a, int lo, int hi, int me) {
    int rank = 0, q = lo;
    for (; q + 3 < hi; q += 4) {
        const int a0 = a[q], a1 = a[q + 1], a2 = a[q + 2], a3 = a[q + 3];
        rank += ((a0 < me) ? 1 : 0) + ((a1 < me) ? 1 : 0) + ((a2 < me) ? 1 : 0) + ((a3 < me) ? 1 : 0);
    }
    for (; q < hi; ++q) rank += (a[q] < me) ? 1 : 0;
    return rank;
}
// ---------------------------------------------------------------------------------
// inv (optional): inv[item] = its sorted position
template <class F>
DEV void wg_bucket_sort(int n, int nb, F bucket_of, int* ptr, int* cur, int* tmp,
                        int* slot_bucket, int* order, int* part, bool prezeroed = false, int* inv = nullptr) {
    if (!prezeroed) {      // callers that can clear ptr/cur in an earlier phase save this barrier
        FOR_TID(b, nb + 1) { ptr[b] = 0; cur[b] = 0; }
        BARRIER();
    }
    FOR_TID(i, n) { ATOMIC_ADD(&ptr[bucket_of(i)], 1); }
    BARRIER();
    wg_exscan(ptr, nb + 1, part);
    FOR_TID(i, n) {
        const int b = bucket_of(i);
        const int pos = ptr[b] + ATOMIC_ADD(&cur[b], 1);
        tmp[pos] = i;
        slot_bucket[pos] = b;
    }
    BARRIER();
    FOR_TID(p, n) {
        const int b = slot_bucket[p];
        const int me = tmp[p];
        const int lo = ptr[b], hi = ptr[b + 1];
        int rank = 0;
        for (int q = lo; q < hi; ++q) rank += (tmp[q] < me) ? 1 : 0;
        order[lo + rank] = me;
        if (inv) inv[me] = lo + rank;
    }
    BARRIER();
}

// ---------------------------------------------------------------------------------
// consecutive_cluster [3P] for one graph: ids (any int64) -> rank among the distinct ids
// present (order preserving), plus member lists with ascending member index.
// Outputs in scratch: s.cl[0..n), s.mp[0..C], s.mem[0..n); returns C.
// ---------------------------------------------------------------------------------
// min / max of ids[0..n) into mm[0], mm[1]: per-lane running values, a wave butterfly, then
// ONE atomic per wave (same-address LDS atomics serialise: 2 per element cost ~12k cycles)
DEV void wg_minmax64(const int64_t* ids, int n, long long* mm, bool preinit = false) {
    if (!preinit) {
        FOR_TID(i, 1) { mm[0] = LLONG_MAX; mm[1] = LLONG_MIN; }
        BARRIER();
    }
#ifdef DRGNN_EMU
    for (int i = 0; i < n; ++i) {
        if ((long long)ids[i] < mm[0]) mm[0] = (long long)ids[i];
        if ((long long)ids[i] > mm[1]) mm[1] = (long long)ids[i];
    }
#else
    long long lo = LLONG_MAX, hi = LLONG_MIN;
    for (int i = threadIdx.x; i < n; i += DRGNN_NTHREADS) {
        const long long v = (long long)ids[i];
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
    }
    wave_minmax_i64(lo, hi);
    if ((threadIdx.x & (DRGNN_WAVE - 1)) == 0 && lo <= hi) { ATOMIC_MIN64(&mm[0], lo); ATOMIC_MAX64(&mm[1], hi); }
#endif
    BARRIER();
}

// Preparation of wg_cluster_rank inside an EARLIER phase of the caller (so that the ids' memory latency overlaps that
// phase's own loads and no phase of its own is spent on the min / max): every wave leaves the min / max of the ids its
// lanes read in mm[2 + 2w], mm[3 + 2w] and the ids' low words go to s.pp (id - min fits 32 bits whenever the flag path is
// taken, so the low words are all that path needs).  The caller also clears fl[0..capF) and ends the phase with a barrier.
DEV void wg_rank_prepare_to(const int64_t* ids, int n, int* low, long long* slots) {
#ifdef DRGNN_EMU
    long long lo = LLONG_MAX, hi = LLONG_MIN;
    for (int i = 0; i < n; ++i) {
        const long long v = (long long)ids[i];
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
        low[i] = (int)(unsigned int)((unsigned long long)v & 0xffffffffull);
    }
    for (int w = 0; w < DRGNN_NTHREADS / DRGNN_WAVE; ++w) { slots[2 * w] = LLONG_MAX; slots[2 * w + 1] = LLONG_MIN; }
    slots[0] = lo; slots[1] = hi;
#else
    long long lo = LLONG_MAX, hi = LLONG_MIN;
    for (int i = threadIdx.x; i < n; i += DRGNN_NTHREADS) {
        const long long v = (long long)ids[i];
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
        low[i] = (int)(unsigned int)((unsigned long long)v & 0xffffffffull);
    }
    // only the waves that read ids take part (the rank routine reads the first ceil(min(n, threads) / 64) slots)
    if ((int)(threadIdx.x & ~(DRGNN_WAVE - 1)) < n) {
        wave_minmax_i64(lo, hi);
        if ((threadIdx.x & (DRGNN_WAVE - 1)) == 0) {
            slots[2 * (threadIdx.x / DRGNN_WAVE)] = lo;
            slots[2 * (threadIdx.x / DRGNN_WAVE) + 1] = hi;
        }
    }
#endif
}
DEV void wg_rank_prepare(const int64_t* ids, int n, TopoScratch& s) { wg_rank_prepare_to(ids, n, s.pp, s.mm + 2); }
#ifndef DRGNN_EMU
// the same for n <= DRGNN_NTHREADS with the ids ALREADY in registers (v = ids[threadIdx.x], requested by the caller ahead of
// its other loads: one memory round trip for everything a phase reads instead of one per list)
DEV void wg_rank_prepare_val(long long v, int n, int* low, long long* slots) {
    long long lo = LLONG_MAX, hi = LLONG_MIN;
    if ((int)threadIdx.x < n) {
        lo = v; hi = v;
        low[threadIdx.x] = (int)(unsigned int)((unsigned long long)v & 0xffffffffull);
    }
    if ((int)(threadIdx.x & ~(DRGNN_WAVE - 1)) < n) {
        wave_minmax_i64(lo, hi);
        if ((threadIdx.x & (DRGNN_WAVE - 1)) == 0) {
            slots[2 * (threadIdx.x / DRGNN_WAVE)] = lo;
            slots[2 * (threadIdx.x / DRGNN_WAVE) + 1] = hi;
        }
    }
}
#endif
// min / max of n prepared ids from the wave slots wg_rank_prepare_to left
DEV void wg_prepared_minmax(const long long* slots, int n, long long& mn, long long& mx) {
    mn = LLONG_MAX; mx = LLONG_MIN;
    const int nw = imin(DRGNN_NTHREADS / DRGNN_WAVE, (n + DRGNN_WAVE - 1) / DRGNN_WAVE);     // slots that were written
    if (nw == 1) { mn = slots[0]; mx = slots[1]; return; }
    for (int w = 0; w < nw; ++w) {
        const long long l = slots[2 * w], h = slots[2 * w + 1];
        mn = l < mn ? l : mn;
        mx = h > mx ? h : mx;
    }
}

// `prepared`: the caller has run wg_rank_prepare(ids, n, s) and cleared fl[0..capF) in an earlier phase
DEV int wg_cluster_rank(const TopoView& tv, int graph, const int64_t* ids, int n, TopoScratch& s,
                        bool prepared = false, bool with_members = true, int* inv = nullptr) {
    long long mn, mx;
    if (prepared) {
        mn = LLONG_MAX; mx = LLONG_MIN;
        const int nw = imin(DRGNN_NTHREADS / DRGNN_WAVE, (n + DRGNN_WAVE - 1) / DRGNN_WAVE);     // slots that were written
        for (int w = 0; w < nw; ++w) {
            const long long l = s.mm[2 + 2 * w], h = s.mm[3 + 2 * w];
            mn = l < mn ? l : mn;
            mx = h > mx ? h : mx;
        }
    } else {
        wg_minmax64(ids, n, s.mm, false);
        mn = s.mm[0]; mx = s.mm[1];
    }
    const long long span = (n > 0) ? (mx - mn + 1) : 0;
    int C;
    if (span >= 0 && span <= (long long)(s.capF - 1)) {
        // usual case (ids are small labels): presence flags over [min, max] + scan, O(n + span)
        const int range = (int)span;
        if (!prepared) {
            FOR_TID(v, range + 1) { s.fl[v] = 0; }
            BARRIER();
        }
        // (prepared: id - min from the staged low words, exact because 0 <= id - min < capF)
        const unsigned int mn_lo = (unsigned int)((unsigned long long)mn & 0xffffffffull);
        if (prepared) { FOR_TID(i, n) { s.fl[(int)((unsigned int)s.pp[i] - mn_lo)] = 1; } }
        else { FOR_TID(i, n) { s.fl[(int)((long long)ids[i] - mn)] = 1; } }
        BARRIER();
        C = wg_exscan(s.fl, range + 1, s.part);
        if (prepared) { FOR_TID(i, n) { s.cl[i] = s.fl[(int)((unsigned int)s.pp[i] - mn_lo)]; } }
        else { FOR_TID(i, n) { s.cl[i] = s.fl[(int)((long long)ids[i] - mn)]; } }
        FOR_TID(b, n + 1) { s.mp[b] = 0; s.cur[b] = 0; }       // for the member bucket sort below
        BARRIER();
    } else {
        // arbitrary ids: rank sort of the members by (id, position), O(n^2) comparisons spread
        // over the workgroup; rank of an id = number of distinct smaller ids
        FOR_TID(i, n) {
            const long long me = (long long)ids[i];
            int r = 0;
            for (int j = 0; j < n; ++j) {
                const long long o = (long long)ids[j];
                r += (o < me || (o == me && j < i)) ? 1 : 0;
            }
            s.t1[r] = i;
        }
        BARRIER();
        FOR_TID(p, n + 1) {
            int head = 0;
            if (p < n) head = (p == 0 || ids[s.t1[p]] != ids[s.t1[p - 1]]) ? 1 : 0;
            s.t2[p] = head;
        }
        BARRIER();
        C = wg_exscan(s.t2, n + 1, s.part);
        FOR_TID(p, n) {
            const bool head = (p == 0 || ids[s.t1[p]] != ids[s.t1[p - 1]]);
            s.cl[s.t1[p]] = s.t2[p] - (head ? 0 : 1);
        }
        FOR_TID(b, n + 1) { s.mp[b] = 0; s.cur[b] = 0; }
        BARRIER();
    }
    if (with_members) {
        const int* cl = s.cl;
        wg_bucket_sort(n, C, [cl] LAMBDA_DEV(int i) { return cl[i]; }, s.mp, s.cur, s.t1, s.t2, s.mem,
                       s.part, true, inv);
    }
    return C;
}

// CSC of a CSR matrix with n rows/cols and m stored entries (slot_row[k] = row of slot k).
DEV void wg_csc_build(int n, int m, const int* col, const int* slot_row, TopoScratch& s,
                      int32_t* g_colptr, int32_t* g_rowidx, int32_t* g_tslot, bool prezeroed = false) {
    wg_bucket_sort(m, n, [col] LAMBDA_DEV(int k) { return col[k]; }, s.cp, s.cur, s.t1, s.t2, s.t3,
                   s.part, prezeroed);
    FOR_TID(j, m) {
        const int k = s.t3[j];
        g_tslot[j] = k;
        g_rowidx[j] = slot_row[k];
    }
    FOR_TID(i, n + 1) { g_colptr[i] = s.cp[i]; }
    BARRIER();
}

// ---------------------------------------------------------------------------------
// The per-graph builder.
// ---------------------------------------------------------------------------------
struct TopoArgs {
    const int64_t* edge_index;   // [2, Etot]
    const float* edge_attr;      // [Etot] or null
    const int64_t* cluster0;     // [Ntot]
    const int64_t* cluster1;     // [L1] or null
    const int32_t* c1_ptr;       // [B+1] or null (then level 1 is built by topo_graph_level1)
    int64_t n_edges;
    int64_t len_cluster1;
    int n_graphs;
    // Resident-set mode (set_ids != null): the arrays above are those of the WHOLE graph set (graph-major, local
    // node ids, n_edges = the set's edge total) and slot g of the mini-batch is graph set_ids[g] of the set; the
    // builder then also gathers the slot's node features and target into the mini-batch's compact buffers.
    const int32_t* set_ids;       // [B] or null
    const int64_t* set_node_ptr;  // [G+1]
    const int64_t* set_edge_ptr;  // [G+1]
    const int64_t* set_c1_ptr;    // [G+1]
    const float* set_x;           // [sumN, F]
    const void* set_y;            // [G] elements of y_bytes bytes, or null
    float* x_out;                 // [N, F] of the mini-batch
    void* y_out;                  // [B]
    int64_t n_set;
    int n_feat, y_bytes;
    int flags;                    // DRGNN_TOPO_* (include/drgnn.h)
    // DRGNN_TOPO_TILES: x_in = the mini-batch's node features (null in resident-set mode: set_x), tiles = output
    // (S [tile_nodes][tile_f] | D [tile_nodes] | C [tile_nodes]); tile_f = 0: no tiles
    const float* x_in;
    float* tiles;
    int64_t tile_nodes;
    int tile_f;
};

// where slot g's index data lives: in the mini-batch tensors (global ids, shifted by n0) or in the resident set
struct TopoSrc {
    const int64_t* row; const int64_t* col; const float* attr; const int64_t* cl0; const int64_t* cl1;
    long long shift; long long x_row;
};
DEV TopoSrc topo_src(const TopoArgs& a, int g, int n0, int e0) {
    TopoSrc t;
    if (a.set_ids == nullptr) {
        t.row = a.edge_index + e0;
        t.col = a.edge_index + a.n_edges + e0;
        t.attr = a.edge_attr ? a.edge_attr + e0 : nullptr;
        t.cl0 = a.cluster0 ? a.cluster0 + n0 : nullptr;
        t.cl1 = (a.cluster1 && a.c1_ptr) ? a.cluster1 + a.c1_ptr[g] : nullptr;
        t.shift = n0;
        t.x_row = n0;
    } else {
        long long id = a.set_ids[g];
        if (id < 0 || id >= a.n_set) id = 0;      // the caller validated the ids; never read outside the set
        const long long se = a.set_edge_ptr[id], sn = a.set_node_ptr[id];
        t.row = a.edge_index + se;
        t.col = a.edge_index + a.n_edges + se;
        t.attr = a.edge_attr ? a.edge_attr + se : nullptr;
        t.cl0 = a.cluster0 ? a.cluster0 + sn : nullptr;
        t.cl1 = (a.cluster1 && a.set_c1_ptr) ? a.cluster1 + a.set_c1_ptr[id] : nullptr;
        t.shift = 0;
        t.x_row = sn;
    }
    return t;
}
// resident-set mode: node features and target of slot g -> the mini-batch's compact buffers
DEV void topo_gather_rows(const TopoArgs& a, const TopoSrc& src, int g, int n0, int N) {
    if (a.set_ids == nullptr || a.x_out == nullptr) return;
    const int F = a.n_feat;
    const float* xs = a.set_x + src.x_row * F;
    float* xd = a.x_out + (long long)n0 * F;
#ifndef DRGNN_EMU
    if ((F & 3) == 0) {
        const drgnn_f4* xs4 = (const drgnn_f4*)xs;
        drgnn_f4* xd4 = (drgnn_f4*)xd;
        FOR_TID(i, N * (F >> 2)) { xd4[i] = xs4[i]; }
    } else
#endif
    {
        FOR_TID(i, N * F) { xd[i] = xs[i]; }
    }
    if (a.y_out && a.set_y) {
        long long id = a.set_ids[g];
        if (id < 0 || id >= a.n_set) id = 0;
        FOR_TID(i, 1) {
            if (a.y_bytes == 8) ((int64_t*)a.y_out)[g] = ((const int64_t*)a.set_y)[id];
            else ((int32_t*)a.y_out)[g] = ((const int32_t*)a.set_y)[id];
        }
    }
}

// ---- hierarchical node order (DRGNN_TI_HORD / HMP0 / HSPLIT; consumer: the node-split step kernels, drgnn_step2.h) ----
// Nodes sorted by (depth-1 cluster of their depth-0 cluster, depth-0 cluster, node id) = a stable bucket sort of the nodes
// by the POSITION q of their depth-0 cluster in the depth-1 member list (MEM1 lists the depth-0 clusters depth-1-major):
// the bucket offsets are HMP0, the sorted items HORD.  cl0 [N]: depth-0 cluster rank of every node (scratch copy, or the
// workspace's CL0); s.mem / s.mp: the depth-1 member lists just built; qpos = s.t4: position of every depth-0 cluster in
// s.mem (the depth-1 sort's inverse; clusters it did not cover -- a flagged graph -- keep the initial last bucket: outputs
// stay structurally valid, the graph is poisoned anyway).  topo_hier_init has cleared ptr = s.rp1, cur = s.nb and set
// qpos in an earlier phase.  Four barriers: histogram | scan (2) | claim | rank.
// Split point: the number k of leading depth-1 clusters whose node total is closest to N / 2 (smallest k on ties).
DEV void topo_hier_init(int C, TopoScratch& s) {
    FOR_TID(c, C + 1) { s.t4[c] = (C > 0) ? C - 1 : 0; s.rp1[c] = 0; s.nb[c] = 0; }
}
DEV void topo_hier(const TopoView& tv, int g, int n0, int N, int C, int C1, const int* cl0, TopoScratch& s) {
    const int rowbase = n0 + g;
    const int* qpos = s.t4;
    int* ptr = s.rp1;
    int* cur = s.nb;
    int* tmp = s.t1;
    int* slot_bucket = s.t2;
    long long* best = (long long*)s.fl;      // (the depth-1 ranking is done with its flags; 16-byte aligned)
    FOR_TID(i, N) { ATOMIC_ADD(&ptr[qpos[cl0[i]]], 1); }
    FOR_TID(i, 1) { best[0] = LLONG_MAX; }
    BARRIER();
    wg_exscan(ptr, C + 1, s.part);
    int32_t* g_hord = tv.p[DRGNN_TI_HORD] + n0;
    int32_t* g_ihord = tv.p[DRGNN_TI_IHORD] + n0;
    int32_t* g_hmp = tv.p[DRGNN_TI_HMP0] + rowbase;
    FOR_TID(i, N) {
        const int b = qpos[cl0[i]];
        const int pos = ptr[b] + ATOMIC_ADD(&cur[b], 1);
        tmp[pos] = i;
        slot_bucket[pos] = b;
    }
    FOR_TID(q, C + 1) { g_hmp[q] = ptr[q]; }
    FOR_TID(k, C1 + 1) {
        const int pos = ptr[imin(s.mp[k], C)];
        int d = 2 * pos - N;
        d = d < 0 ? -d : d;
        ATOMIC_MIN64(&best[0], ((long long)d << 32) | (long long)k);
    }
    BARRIER();
    FOR_TID(p, N) {
        const int b = slot_bucket[p];
        const int me = tmp[p];
        const int lo = ptr[b], hi = ptr[b + 1];
        int rank = 0;
        for (int q = lo; q < hi; ++q) rank += (tmp[q] < me) ? 1 : 0;
        g_hord[lo + rank] = me;
        g_ihord[me] = lo + rank;
    }
    FOR_TID(i, 1) {
        const int k = (int)(best[0] & 0xffffffffLL);
        const int q = imin(s.mp[k], C);
        int32_t* hs = tv.p[DRGNN_TI_HSPLIT] + 4 * g;
        hs[0] = k; hs[1] = q; hs[2] = ptr[q]; hs[3] = C1;
    }
}

// C0 = number of depth-0 clusters of the graph; sidx = this workgroup's status word
// cl0: depth-0 cluster rank of every node for the hierarchical order (null: not built)
DEV void topo_graph_level1(const TopoView& tv, const TopoArgs& a, int g, int n0, int C0, const int64_t* ids1,
                           int c1_len, TopoScratch& s, int sidx, bool prepared = false, const int* cl0 = nullptr,
                           int N = 0, bool hier_ready = false) {
    const int rowbase = n0 + g;
    if (c1_len != C0) {
        FOR_TID(i, 1) { topo_flag(tv, DRGNN_S_CLUSTER1_LEN, sidx); }
    }
    const int n = imin(C0, imax(c1_len, 0));
    if (cl0 != nullptr && !hier_ready) { topo_hier_init(C0, s); BARRIER(); }
    const int C1 = wg_cluster_rank(tv, sidx, ids1, n, s, prepared, true, cl0 != nullptr ? s.t4 : nullptr);
    int32_t* g_cl1 = tv.p[DRGNN_TI_CL1] + n0;
    int32_t* g_mptr1 = tv.p[DRGNN_TI_MPTR1] + rowbase;
    int32_t* g_mem1 = tv.p[DRGNN_TI_MEM1] + n0;
    FOR_TID(i, n) { g_cl1[i] = s.cl[i]; g_mem1[i] = s.mem[i]; }
    FOR_TID(c, C1 + 1) { g_mptr1[c] = s.mp[c]; }
    FOR_TID(i, 1) { tv.p[DRGNN_TI_NC1][g] = C1; }
    BARRIER();
    if (cl0 != nullptr) topo_hier(tv, g, n0, N, C0, C1, cl0, s);
}

// role 0: everything in one workgroup.  With two workgroups per graph the work splits into two INDEPENDENT chains of
// about equal length, each starting from its own staged copy of the edge list:
//   role 1 "pool"      : depth-0 cluster ranks + member lists -> pooled graph (pool_edge) -> CSC1
//   role 2 "structure" : node-feature gather, CSR0 / CSC0 (+ edge weights), depth-0 cluster count, depth-1 clusters + members
// (round 1 had CSR0 / CSC0 in front of the pooling chain, ~41 k stamped cycles against ~29 k for the member lists; the
// pooled graph only needs (cluster(row), cluster(col)) of the RAW edges, so CSR0 / CSC0 moved to the other chain and the
// depth-0 member lists came over in exchange: ~37 k / ~38 k.)
#define TOPO_ROLE_ALL 0
#define TOPO_ROLE_EDGES 1        // "pool"
#define TOPO_ROLE_MEMBERS 2      // "structure"

// ---- the edge list of the graph, once: int64 global ids -> int32 local (s.er, s.ec), weights by edge id (s.w0) ----
DEV void topo_stage_edges(const TopoView& tv, const TopoSrc& src, int g, int N, int E, bool has_w, TopoScratch& s) {
    const int64_t* src_row = src.row;
    const int64_t* src_col = src.col;
    const long long shift = src.shift;
    const int Nm1 = N - 1;
    FOR_TID(e, E) {
        long long r = (long long)src_row[e] - shift, c = (long long)src_col[e] - shift;
        if (r < 0 || r > Nm1 || c < 0 || c > Nm1) {
            topo_flag(tv, DRGNN_S_EDGE_RANGE, g);
            r = (r < 0 || r > Nm1) ? 0 : r;
            c = (c < 0 || c > Nm1) ? 0 : c;
        }
        s.er[e] = (int)r;
        s.ec[e] = (int)c;
        if (has_w) s.w0[e] = src.attr[e];
    }
}

// ---- CSR0 and CSC0 together: histogram, one scan, slot claim, rank sort by edge id.  Needs the staged edges; the
// caller has cleared rp[0 .. 2N+2), cur[0 .. N], nb[0 .. N) in an earlier phase.  Ends with a barrier. ----
DEV void topo_csr0(const TopoView& tv, int g, int n0, int e0, int N, int E, bool has_w, TopoScratch& s) {
    const int rowbase = n0 + g;
    int32_t* g_rowptr0 = tv.p[DRGNN_TI_ROWPTR0] + rowbase;
    int32_t* g_col0 = tv.p[DRGNN_TI_COL0] + e0;
    int32_t* g_eid0 = tv.p[DRGNN_TI_EID0] + e0;
    float* g_w0 = tv.w0 ? tv.w0 + e0 : nullptr;
    int* rp = s.rp;             // [N+1]   rows
    int* cp = s.rp + (N + 1);   // [N+1]   cols, contiguous with rp so ONE scan serves both
    int* cur_r = s.cur;
    int* cur_c = s.nb;
    FOR_TID(e, E) {
        ATOMIC_ADD(&rp[s.er[e]], 1);
        ATOMIC_ADD(&cp[s.ec[e]], 1);
    }
    BARRIER();
    wg_exscan(rp, 2 * N + 2, s.part);      // cp[] now carries an extra +E (the row total)
    FOR_TID(e, E) {
        const int r = s.er[e], c = s.ec[e];
        s.t1[rp[r] + ATOMIC_ADD(&cur_r[r], 1)] = e;
        s.t2[cp[c] - E + ATOMIC_ADD(&cur_c[c], 1)] = e;
    }
    BARRIER();
    FOR_TID(p, E) {
        {
            const int e = s.t1[p];
            const int lo = rp[s.er[e]], hi = rp[s.er[e] + 1];
            int rank = 0;
            for (int q = lo; q < hi; ++q) rank += (s.t1[q] < e) ? 1 : 0;
            s.t3[lo + rank] = e;                       // CSR0 slot -> edge
        }
        {
            const int e = s.t2[p];
            const int lo = cp[s.ec[e]] - E, hi = cp[s.ec[e] + 1] - E;
            int rank = 0;
            for (int q = lo; q < hi; ++q) rank += (s.t2[q] < e) ? 1 : 0;
            s.t4[lo + rank] = e;                       // CSC0 entry -> edge
        }
    }
    BARRIER();
    FOR_TID(k, E) {
        const int e = s.t3[k];
        s.t5[e] = k;                                   // edge -> CSR0 slot
        s.col[k] = s.ec[e];                            // (column / weight by slot stay in LDS for topo_tiles_rows)
        if (has_w) ((float*)s.seg)[k] = s.w0[e];
        g_col0[k] = s.ec[e];
        g_eid0[k] = e;
        if (has_w) g_w0[k] = s.w0[e];
    }
    {
        int32_t* g_colptr0 = tv.p[DRGNN_TI_COLPTR0] + rowbase;
        FOR_TID(i, N + 1) { g_rowptr0[i] = rp[i]; g_colptr0[i] = cp[i] - E; }
    }
    BARRIER();
    {
        int32_t* g_rowidx0 = tv.p[DRGNN_TI_ROWIDX0] + e0;
        int32_t* g_tslot0 = tv.p[DRGNN_TI_TSLOT0] + e0;
        FOR_TID(j, E) {
            const int e = s.t4[j];
            g_rowidx0[j] = s.er[e];
            g_tslot0[j] = s.t5[e];
        }
    }
    BARRIER();
}

// ---- pool_edge + CSC1 from the RAW staged edges and the depth-0 cluster ranks s.cl[0..N) (C clusters) -------------
// Without edge weights the pooled graph is just the SET of (cluster(row), cluster(col)) pairs of the edges minus self
// loops: every pooled row keeps a bitmap of its target clusters (LDS atomic OR, order independent), the sorted unique
// targets are its set bits in ascending order.  No sorting at all; used whenever the C x ceil(C/32) words fit the sort
// scratch.  With weights: the edges are bucketed by pooled row, every bucket ranked by (target cluster, edge id), runs of
// equal target are the coalesced pooled edges and their weights are summed in that (fixed) order.
DEV void topo_pool(const TopoView& tv, int g, int n0, int e0, int E, int C, bool has_w, TopoScratch& s) {
    const int rowbase = n0 + g;
    float* g_w1 = tv.w1 ? tv.w1 + e0 : nullptr;
    const int BW = (C + 31) >> 5;
    const bool bitmap = !has_w && (long)C * BW + 1 <= (long)s.capT;
    int E1;
    int32_t* g_rowptr1 = tv.p[DRGNN_TI_ROWPTR1] + rowbase;
    int32_t* g_col1 = tv.p[DRGNN_TI_COL1] + e0;
    if (bitmap) {
        int* bm = s.t1;          // [C][BW] target bitmaps
        int* pre = s.t2;         // [C*BW + 1] set bits before each word
        FOR_TID(q, C * BW) { bm[q] = 0; }
        BARRIER();
        FOR_TID(e, E) {
            const int r = s.cl[s.er[e]], cc = s.cl[s.ec[e]];
            if (cc != r) ATOMIC_OR(&bm[r * BW + (cc >> 5)], (int)(1u << (cc & 31)));
        }
        BARRIER();
        FOR_TID(q, C * BW + 1) { pre[q] = (q < C * BW) ? __builtin_popcount((unsigned)bm[q]) : 0; }
        BARRIER();
        E1 = wg_exscan(pre, C * BW + 1, s.part);
        FOR_TID(q, C * BW) {
            unsigned bits = (unsigned)bm[q];
            const int r = q / BW, base_col = (q - r * BW) << 5;
            int slot = pre[q];
            while (bits) {
                const int b = __builtin_ctz(bits);
                bits &= bits - 1;
                s.col1[slot] = base_col + b;
                s.seg[slot] = r;               // row of pooled CSR slot
                g_col1[slot] = base_col + b;
                ++slot;
            }
        }
        FOR_TID(r, C + 1) {
            const int v = pre[r * BW];
            s.rp1[r] = v;
            g_rowptr1[r] = v;
            s.cp[r] = 0;                                          // histogram / cursors of the CSC1 build
            s.cur[r] = 0;
        }
        FOR_TID(i, 1) { tv.p[DRGNN_TI_NE1][g] = E1; }
        BARRIER();
    } else {
        FOR_TID(r, C + 1) { s.pp[r] = 0; s.cur[r] = 0; }
        FOR_TID(i, 1) { TOPO_WMAX(s)[0] = 0; }
        BARRIER();
        {
            int mine = 0;
            FOR_TID(e, E) {
                ATOMIC_ADD(&s.pp[s.cl[s.er[e]]], 1);
                if (has_w) { const int b = topo_f2i(fabsf(s.w0[e])) & 0x7fffffff; mine = b > mine ? b : mine; }
            }
#ifdef DRGNN_EMU
            if (has_w) ATOMIC_MAX(TOPO_WMAX(s), mine);
#else
            if (has_w && (int)(threadIdx.x & ~(DRGNN_WAVE - 1)) < E) {      // (one atomic per wave)
#pragma unroll
                for (int m = 1; m < DRGNN_WAVE; m <<= 1) { const int o = __shfl_xor(mine, m, DRGNN_WAVE); mine = o > mine ? o : mine; }
                if ((threadIdx.x & (DRGNN_WAVE - 1)) == 0) ATOMIC_MAX(TOPO_WMAX(s), mine);
            }
#endif
        }
        BARRIER();
        wg_exscan(s.pp, C + 1, s.part);
        // (target cluster, edge id) packed into ONE word -- clusters < 2^15 (max_nodes <= 32767), edge ids < 2^16
        // (max_edges <= 65535): the rank loop below then reads and compares one word per candidate.  A self loop of the
        // pooled graph (dropped) carries the cluster field DROPPED, which sorts behind every real target and stays unique.
        // (graphs beyond those bounds -- only the global-scratch builder takes them -- keep two words per candidate)
        constexpr int DROPPED = 0x7FFF;
        const bool packed = (E <= 0x10000) && (C <= DROPPED);
        FOR_TID(e, E) {                                   // one work item per edge
            const int r = s.cl[s.er[e]];
            const int cc = s.cl[s.ec[e]];
            const int j = s.pp[r] + ATOMIC_ADD(&s.cur[r], 1);
            if (packed) {
                s.t1[j] = (((cc == r) ? DROPPED : cc) << 16) | e;
            } else {
                s.t1[j] = (cc == r) ? INT_MAX : cc;       // self loop of the pooled graph: dropped
                s.t2[j] = e;
            }
            s.t3[j] = r;
        }
        BARRIER();
        // rank sort of every pooled row's candidates by (target cluster, edge id).  A bucket holds the 20 - 60 edges that leave
        // one cluster: four candidates per trip in flight (a trip was two dependent LDS round trips per candidate: the slowest
        // lane's bucket set the pace of the whole phase, 4 us of the builder's 17 for the SYN graphs)
        FOR_TID(j, E) {
            const int r = s.t3[j];
            const int mine = s.t1[j];
            const int lo = s.pp[r], hi = s.pp[r + 1];
            if (packed) {
                const int rank = rank_below(s.t1, lo, hi, mine);
                const int key = mine >> 16;
                s.t4[lo + rank] = (key == DROPPED) ? INT_MAX : key;
                s.t5[lo + rank] = mine & 0xFFFF;
            } else {
                const int id = s.t2[j];
                int rank = 0;
                for (int q = lo; q < hi; ++q) {
                    const int kq = s.t1[q];
                    rank += (kq < mine || (kq == mine && s.t2[q] < id)) ? 1 : 0;
                }
                s.t4[lo + rank] = mine;
                s.t5[lo + rank] = id;
            }
        }
        BARRIER();
        // heads of runs of equal target = the coalesced pooled edges, already (row, col) sorted
        // (t3[j] = pooled row of sorted position j: positions of a bucket stay inside the bucket)
        // + the weight of every sorted position, fetched once by independent reads (the run heads below add runs of them;
        // t2 is free again: the ranking above was its last reader)
        float* const wv = reinterpret_cast<float*>(s.t2);
        FOR_TID(j, E + 1) {
            int head = 0;
            if (j < E) {
                const int key = s.t4[j];
                head = (key != INT_MAX && (j == s.pp[s.t3[j]] || s.t4[j - 1] != key)) ? 1 : 0;
                if (has_w) wv[j] = s.w0[s.t5[j]];
            }
            s.t1[j] = head;
        }
        BARRIER();
        E1 = wg_exscan(s.t1, E + 1, s.part);
        double winv = 1.0;
        const double wsc = has_w ? topo_wscale(TOPO_WMAX(s)[0], &winv) : 1.0;
        FOR_TID(j, E) {
            const int key = s.t4[j];
            const int r = s.t3[j];
            if (key != INT_MAX && (j == s.pp[r] || s.t4[j - 1] != key)) {
                const int slot = s.t1[j];
                s.col1[slot] = key;
                s.seg[slot] = r;               // row of pooled CSR slot
                g_col1[slot] = key;
                if (has_w) {
                    // the run's weights in fixed point (topo_wscale: exact, the same bits as the lean chain's atomics); four
                    // positions per trip in flight
                    const int hi = s.pp[r + 1];
                    long long w = 0;
                    int q = j;
                    for (; q + 3 < hi; q += 4) {
                        const int k0 = s.t4[q], k1 = s.t4[q + 1], k2 = s.t4[q + 2], k3 = s.t4[q + 3];
                        const float v0 = wv[q], v1 = wv[q + 1], v2 = wv[q + 2], v3 = wv[q + 3];
                        const bool e0 = k0 == key, e1 = e0 && k1 == key, e2 = e1 && k2 == key, e3 = e2 && k3 == key;
                        if (e0) w += topo_wfix(v0, wsc);
                        if (e1) w += topo_wfix(v1, wsc);
                        if (e2) w += topo_wfix(v2, wsc);
                        if (e3) w += topo_wfix(v3, wsc);
                        if (!e3) { q = hi; break; }
                    }
                    for (; q < hi && s.t4[q] == key; ++q) w += topo_wfix(wv[q], wsc);
                    g_w1[slot] = (float)((double)w * winv);
                }
            }
        }
        FOR_TID(r, C + 1) {
            const int v = s.t1[s.pp[r]];
            s.rp1[r] = v;
            g_rowptr1[r] = v;
            s.cp[r] = 0;                                          // histogram / cursors of the CSC1 build
            s.cur[r] = 0;
        }
        FOR_TID(i, 1) { tv.p[DRGNN_TI_NE1][g] = E1; }
        BARRIER();
    }
    wg_csc_build(C, E1, s.col1, s.seg, s, tv.p[DRGNN_TI_COLPTR1] + rowbase,
                 tv.p[DRGNN_TI_ROWIDX1] + e0, tv.p[DRGNN_TI_TSLOT1] + e0, true);
}

// depth-0 cluster ranks; with_members: + member lists, both written out.  Returns C.
// ranks_out: without member lists, still write the ranks and the count (lean build of a weighted graph)
DEV int topo_clusters0(const TopoView& tv, int g, int n0, int N, const TopoSrc& src, TopoScratch& s, int sidx,
                       bool with_members, bool ranks_out = false) {
    const int rowbase = n0 + g;
    const int C = wg_cluster_rank(tv, sidx, src.cl0, N, s, true, with_members);
    if (!with_members && ranks_out) {
        int32_t* g_cl0 = tv.p[DRGNN_TI_CL0] + n0;
        FOR_TID(i, N) { g_cl0[i] = s.cl[i]; }
        FOR_TID(i, 1) { tv.p[DRGNN_TI_NC0][g] = C; }
    }
    if (with_members) {
        int32_t* g_cl0 = tv.p[DRGNN_TI_CL0] + n0;
        int32_t* g_mptr0 = tv.p[DRGNN_TI_MPTR0] + rowbase;
        int32_t* g_mem0 = tv.p[DRGNN_TI_MEM0] + n0;
        FOR_TID(i, N) { g_cl0[i] = s.cl[i]; g_mem0[i] = s.mem[i]; }
        FOR_TID(c, C + 1) { g_mptr0[c] = s.mp[c]; }
        FOR_TID(i, 1) { tv.p[DRGNN_TI_NC0][g] = C; }
    }
    return C;
}
// depth-1 clusters + member lists of a graph with C depth-0 clusters (when the caller located this graph's ids)
// N: nodes of the graph; s.cl holds their depth-0 cluster ranks (saved to s.t5 here: the depth-1 ranking reuses s.cl)
DEV void topo_clusters1(const TopoView& tv, const TopoArgs& a, int g, int n0, int N, int C, const TopoSrc& src, TopoScratch& s,
                        int sidx) {
    if (a.cluster1 == nullptr || a.c1_ptr == nullptr) return;
    const bool hier = (a.flags & DRGNN_TOPO_HIER) != 0;
    FOR_TID(v, s.capF) { s.fl[v] = 0; }
    if (hier) {
        FOR_TID(i, N) { s.t5[i] = s.cl[i]; }
        topo_hier_init(C, s);
    }
    const int b = a.c1_ptr[g];
    const int c1_len = a.c1_ptr[g + 1] - b;
    wg_rank_prepare(src.cl1, imin(C, imax(c1_len, 0)), s);
    BARRIER();
    topo_graph_level1(tv, a, g, n0, C, src.cl1, c1_len, s, sidx, true, hier ? s.t5 : nullptr, N, true);
}

// =========================================================================================================================
// Level-0 aggregation tiles (DRGNN_TOPO_TILES, include/drgnn.h): S_i = sum over the CSR0 row of node i of [w_k] x_col(k), D, C
// -- what conv1 of every net starts from, formed HERE because it depends on the inputs only.  The x tile of the graph was
// requested in phase 0 and sits in LDS (rows of F + 4 floats); rp / col_s / w_s: CSR0 of the graph in LDS (slot order =
// edge-id order).  F / 4 lanes per node, each with one float4 of the row; results go straight to global memory (node order).
// Rows of S are padded to TF = pad4(F) floats (zeros), so that the step kernels load them with 128-bit requests whatever the
// feature count; with F % 4 != 0 the tiles also carry a padded copy of the node features (tx: what sGAT / FoutNet multiply with
// their self weights), the kernels then never touch the unaligned input rows.
struct TopoTile { const float* x; float* ts; float* td; float* tc; float* tx; float* xs; int F, TF; };
// L2 touch of an output range (see topo_graph): up to 4 float4 per lane, requested with the phase's other loads, dropped unread
#ifdef DRGNN_EMU
struct TopoTouch { int dummy; };
DEV void topo_touch_load(TopoTouch&, const float*, int) {}
DEV void topo_touch_done(const TopoTouch&) {}
#else
struct TopoTouch { drgnn_f4 v[4]; };
DEV void topo_touch_load(TopoTouch& t, const float* p, int words) {
#ifdef DRGNN_NO_TOUCH      // (A/B switch: tools/r05/build_af_variant.sh)
    const int n4 = 0;
#else
    const int n4 = p ? (words >> 2) : 0;
#endif
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = threadIdx.x + j * DRGNN_NTHREADS;
        t.v[j] = drgnn_f4{0.f, 0.f, 0.f, 0.f};
        if (q < n4) t.v[j] = ((const drgnn_f4*)p)[q];
    }
}
DEV void topo_touch_done(const TopoTouch& t) {
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" :: "v"(t.v[j]));
}
#endif
DEV TopoTile topo_tile_of(const TopoArgs& a, const TopoSrc& src, int n0, float* xs) {
    TopoTile t;
    t.F = (a.tiles != nullptr) ? a.tile_f : 0;
    const float* xsrc = a.set_ids ? a.set_x : a.x_in;
    if (xsrc == nullptr || xs == nullptr) t.F = 0;
    t.x = (t.F > 0) ? xsrc + src.x_row * t.F : nullptr;
    t.TF = (t.F + 3) & ~3;
    t.ts = (t.F > 0) ? a.tiles + (long long)n0 * t.TF : nullptr;
    t.td = (t.F > 0) ? a.tiles + a.tile_nodes * t.TF + n0 : nullptr;
    t.tc = (t.F > 0) ? t.td + a.tile_nodes : nullptr;
    t.tx = (t.F > 0 && (t.F & 3)) ? a.tiles + drgnn_tiles_x_off(a.tile_nodes, t.TF) + (long long)n0 * t.TF : nullptr;
    t.xs = xs;
    return t;
}
#ifdef DRGNN_EMU
struct TopoTileRegs { int dummy; };
DEV void topo_tile_load(TopoTileRegs&, const TopoTile&, int) {}
DEV void topo_tile_store(const TopoTileRegs&, const TopoTile& t, int N) {
    for (int i = 0; i < N && t.F > 0; ++i)
        for (int f = 0; f < t.TF; ++f) t.xs[i * (t.TF + 4) + f] = (f < t.F) ? t.x[(long long)i * t.F + f] : 0.0f;
}
#else
struct TopoTileRegs { BurstX<4> bx; };
DEV void topo_tile_load(TopoTileRegs& r, const TopoTile& t, int N) { if (t.F > 0 && !(t.F & 3)) burst_load_x(r.bx, t.x, N, t.F); }
DEV void topo_tile_store(const TopoTileRegs& r, const TopoTile& t, int N) {
    if (t.F <= 0) return;
    if (!(t.F & 3)) { burst_store_x4(r.bx, t.xs, t.F + 4); return; }
    // feature counts that are not a multiple of 4: word by word, rows padded with zeros
    const FastDiv fd = fastdiv_make(t.TF);
    FOR_TID(e, N * t.TF) {
        const int i = fastdiv(fd, e), f = fastmod(fd, e, i);
        t.xs[i * (t.TF + 4) + f] = (f < t.F) ? t.x[(long long)i * t.F + f] : 0.0f;
    }
}
#endif
DEV void topo_tiles_rows(const TopoTile& t, int N, const int* rp, const int* col_s, const float* w_s) {
    if (t.F <= 0) return;
    const int F = t.TF, G4 = F >> 2, XLD = F + 4;      // (F: the padded row length from here on)
    const FastDiv fd = fastdiv_make(G4);
    FOR_TID(item, N * G4) {
        const int i = fastdiv(fd, item), c = fastmod(fd, item, i) * 4;
        const int lo = rp[i], hi = rp[i + 1], deg = hi - lo;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, asum = 0.f;
        if (w_s != nullptr) {
            // batches of four independent (index -> row) chains, the last one padded under a zero coefficient
            for (int k = lo; k < hi; k += 4) {
                int kk[4];
                float cf[4];
                for (int j = 0; j < 4; ++j) { kk[j] = (k + j < hi) ? k + j : hi - 1; cf[j] = (k + j < hi) ? 1.0f : 0.0f; }
                for (int j = 0; j < 4; ++j) { cf[j] *= w_s[kk[j]]; asum += cf[j]; }
                const float* xj[4];
                for (int j = 0; j < 4; ++j) xj[j] = t.xs + ROW24(col_s[kk[j]], XLD) + c;
                for (int j = 0; j < 4; ++j) {
                    a0 = fmaf(cf[j], xj[j][0], a0); a1 = fmaf(cf[j], xj[j][1], a1);
                    a2 = fmaf(cf[j], xj[j][2], a2); a3 = fmaf(cf[j], xj[j][3], a3);
                }
            }
        } else {
#ifndef DRGNN_EMU
#pragma unroll 4
#endif
            for (int k = lo; k < hi; ++k) {
                const float* xj = t.xs + ROW24(col_s[k], XLD) + c;
                a0 += xj[0]; a1 += xj[1]; a2 += xj[2]; a3 += xj[3];
            }
        }
        float* dst = t.ts + (long long)i * F + c;
        dst[0] = a0; dst[1] = a1; dst[2] = a2; dst[3] = a3;
        if (t.tx != nullptr) {
            const float* xi = t.xs + ROW24(i, XLD) + c;
            float* dx = t.tx + (long long)i * F + c;
            dx[0] = xi[0]; dx[1] = xi[1]; dx[2] = xi[2]; dx[3] = xi[3];
        }
        if (c == 0) {
            float d, sc;
            if (w_s != nullptr) { d = 1.0f / (float)(deg > 0 ? deg : 1); sc = asum * d; }
            else { d = deg > 0 ? 1.0f / (float)deg : 0.0f; sc = 1.0f; }
            t.td[i] = d; t.tc[i] = sc;
        }
    }
}

// =========================================================================================================================
// LEAN build (DRGNN_TOPO_LEAN, include/drgnn.h): only what the aggregation-first training kernels read, by two short chains.
//   rows chain     : CSR0 (+ W0) -- histogram, scan, claim, rank by edge id -- and the aggregation tiles;
//   clusters chain : both consecutive-cluster rankings from ONE exclusive scan over the concatenated presence flags, the
//                    pooled CSR and CSC from a target bitmap and its TRANSPOSE (both emitted behind one popcount scan), the
//                    depth-1 member lists and the hierarchical node order by COUNTING and size sums instead of bucket sorts:
//                    position of a cluster = number of smaller (depth-1 cluster, id) keys (C^2 comparisons), first position of
//                    its nodes = sum of the sizes of the clusters in front (C^2 additions), a node claims a slot of its
//                    cluster's run and the few nodes of a run are ranked by id.
//                    With edge weights the pooled weights are sums over the raw edges of a (row, column) pair: accumulated by
//                    64-bit fixed-point atomics (exact, order independent -- no sorting; the general chain sorts every pooled
//                    row's edges by (target, id) and adds runs).
// Two workgroups per graph: the "structure" workgroup runs the rows chain and the "pool" workgroup the clusters chain
// (20 k / 29 k clock ticks at SYN size; the general chains: 33 k / 41 k).  One workgroup per graph: rows, then clusters.
// Items of the counting loops are shared by G consecutive lanes that meet in DPP adds (the emulation runs items serially).
#ifdef DRGNN_EMU
#define TOPO_LANES(G) 1
template <int G> DEV int topo_group_sum(int v) { return v; }
#else
#define TOPO_LANES(G) (G)
template <int CTRL> DEV int dpp_take_int(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }
template <int G> DEV int topo_group_sum(int v) {
    if (G >= 2) v += dpp_take_int<0xB1>(v);      // quad_perm [1,0,3,2]
    if (G >= 4) v += dpp_take_int<0x4E>(v);      // quad_perm [2,3,0,1]
    if (G >= 8) v += dpp_take_int<0x141>(v);     // row_half_mirror
    if (G >= 16) v += dpp_take_int<0x140>(v);    // row_mirror
    return v;
}
#endif
// number of keys[0 .. n) strictly below `me`: lane `sub` of the item's G lanes takes the keys sub, sub + G, ... (four per trip in
// flight); every lane of the group gets the total
template <int G> DEV int topo_count_below(const int* keys, int n, int me, int sub) {
    constexpr int L = TOPO_LANES(G);
    int cnt = 0;
    for (int j = sub; j < n; j += 4 * L) {
        const int j1 = j + L, j2 = j + 2 * L, j3 = j + 3 * L;
        const int k0 = keys[j], k1 = keys[j1 < n ? j1 : j], k2 = keys[j2 < n ? j2 : j], k3 = keys[j3 < n ? j3 : j];
        cnt += ((k0 < me) ? 1 : 0) + ((j1 < n && k1 < me) ? 1 : 0) + ((j2 < n && k2 < me) ? 1 : 0) + ((j3 < n && k3 < me) ? 1 : 0);
    }
    return topo_group_sum<L>(cnt);
}

// Phase 0 of the lean chains (next to the edge staging; ends with the caller's barrier).  rows: histogram and cursors cleared.
// clusters: the presence flags of both id lists are set HERE -- depth-0 ids in Z[0, R), depth-1 ids in Z[R, 2 R), R = N + 1,
// Z = s.fl, the id itself as the index (cluster ids are labels in [0, N): the reference's are node or cluster numbers;
// anything else raises TOPO_OVF and the caller takes the general chain) -- so the ranking needs no min / max pass and no flag
// phase of its own.  Z and TOPO_OVF were cleared BEFORE the previous barrier (topo_lean_preclear, from topo_block).
// `pre`: the ids are already in registers, requested ahead of the edge list.
// `rows` / `clusters`: which chains this workgroup will run (the rows chain's histogram and cursors are cleared here too: its
// histogram is formed by the lanes that stage the edges, in the builder's first phase)
DEV void topo_lean_preclear(int N, TopoScratch& s, bool rows, bool clusters) {
    if (rows) { FOR_TID(i, N + 1) { s.rp[i] = 0; s.cur[i] = 0; } }
    if (clusters) {
        FOR_TID(v, imin(2 * N + 2, s.capF)) { s.fl[v] = 0; }
        FOR_TID(i, 1) { TOPO_OVF(s)[0] = (2 * N + 2 > s.capF) ? 1 : 0; TOPO_WMAX(s)[0] = 0; }
    }
}
DEV void topo_lean_flag(long long v, int i, int* Z, int N, int* key, TopoScratch& s) {
    if ((unsigned long long)v < (unsigned long long)N) { Z[(int)v] = 1; key[i] = (int)v; }
    else { TOPO_OVF(s)[0] = 1; key[i] = 0; }
}
DEV void topo_lean_prepare(const TopoSrc& src, int N, int E, int n1, bool rows, bool clusters, TopoScratch& s, bool pre = false,
                           long long v0 = 0, long long v1 = 0) {
    // (the rows chain's histogram: every lane counts the edges it has just staged itself, topo_stage_edges' loop)
    if (rows) { FOR_TID(e, E) { ATOMIC_ADD(&s.rp[s.er[e]], 1); } }
    if (clusters && 2 * N + 2 <= s.capF) {
        const int R = N + 1;
        FOR_TID(i, N + 1) { s.cp[i] = 0; }
#ifndef DRGNN_EMU
        if (pre) {
            if ((int)threadIdx.x < N) topo_lean_flag(v0, threadIdx.x, s.fl, N, s.pp, s);
            if ((int)threadIdx.x < n1) topo_lean_flag(v1, threadIdx.x, s.fl + R, N, s.nb, s);
        } else
#endif
        {
            FOR_TID(i, N) { topo_lean_flag((long long)src.cl0[i], i, s.fl, N, s.pp, s); }
            FOR_TID(c, n1) { topo_lean_flag((long long)src.cl1[c], c, s.fl + R, N, s.nb, s); }
        }
    }
    (void)pre; (void)v0; (void)v1;
}

// ---- rows chain: CSR0 (+ W0) and the aggregation tiles.  Needs the staged edges and the row histogram (topo_lean_prepare(rows))
// behind a barrier.  In pieces, one per barrier interval, so that a workgroup that runs BOTH chains (one workgroup per graph:
// batches beyond the resident size) can work them off in the clusters chain's intervals (topo_lean_clusters, `rows`):
//   scan of the histogram | claim | rank + emit | tiles.     Scratch of its own: rp (Z), cur (until the claim is over), t1, col, wr.
DEV void topo_lean_rows_claim(const TopoView& tv, int g, int n0, int N, int E, TopoScratch& s) {
    const int* Z = s.rp;
    int32_t* g_rowptr0 = tv.p[DRGNN_TI_ROWPTR0] + (n0 + g);
    FOR_TID(e, E) { const int r = s.er[e]; s.t1[Z[r] + ATOMIC_ADD(&s.cur[r], 1)] = e; }
    FOR_TID(i, N + 1) { g_rowptr0[i] = Z[i]; }
}
// rank of every edge inside its row by edge id, emitted at once; column / weight by slot stay in LDS for the tiles
DEV void topo_lean_rows_emit(const TopoView& tv, int e0, int E, bool has_w, TopoScratch& s) {
    const int* Z = s.rp;
    int32_t* g_col0 = tv.p[DRGNN_TI_COL0] + e0;
    int32_t* g_eid0 = tv.p[DRGNN_TI_EID0] + e0;
    float* g_w0 = tv.w0 ? tv.w0 + e0 : nullptr;
    FOR_TID(p, E) {
        const int e = s.t1[p];
        const int r = s.er[e];
        const int lo = Z[r], hi = Z[r + 1];
        int rank = 0;
        for (int q = lo; q < hi; ++q) rank += (s.t1[q] < e) ? 1 : 0;
        const int cc = s.ec[e];
        g_col0[lo + rank] = cc;
        g_eid0[lo + rank] = e;
        s.col[lo + rank] = cc;
        if (has_w) { const float w = s.w0[e]; g_w0[lo + rank] = w; s.wr[lo + rank] = w; }
    }
}
// from the claim on (the histogram is scanned); leaves no barrier behind its last phase
DEV void topo_lean_rows_tail(const TopoView& tv, int g, int n0, int e0, int N, int E, bool has_w, TopoScratch& s, const TopoTile& tile) {
    topo_lean_rows_claim(tv, g, n0, N, E, s);
    BARRIER();
    topo_lean_rows_emit(tv, e0, E, has_w, s);
    if (tile.F > 0) {
        BARRIER();
        topo_tiles_rows(tile, N, s.rp, s.col, has_w ? (const float*)s.wr : nullptr);
    }
}
DEV void topo_lean_rows(const TopoView& tv, int g, int n0, int e0, int N, int E, bool has_w, TopoScratch& s, const TopoTile& tile) {
    wg_exscan(s.rp, N + 1, s.part);
    topo_lean_rows_tail(tv, g, n0, e0, N, E, has_w, s, tile);
}

// ---- clusters chain: depth-0 / depth-1 cluster ranks, MEM1 / MPTR1, HORD / IHORD / HMP0 / HSPLIT and (with_pool, no edge
// weights) the pooled CSR + CSC.  Needs the staged edges and topo_lean_prepare(clusters) behind a barrier.  false
// (workgroup-uniform; at most ranks and counts written, which the general chain writes again): ids too sparse for the flag
// array or too many clusters for the bitmaps -- the caller takes the general chain.
// `rows` (one workgroup per graph): the rows chain's pieces are worked off in this chain's barrier intervals (its scan shares
// the first scan's two barriers, wg_exscan2).  Returns 0: done.  Otherwise the caller takes the general chain for the clusters
// and, with `rows`, finishes the rows chain itself: 1 = its histogram is not scanned yet, 2 = scanned (topo_lean_rows_tail).
DEV int topo_lean_clusters(const TopoView& tv, int g, int n0, int e0, int N, int E, int c1_len, bool with_pool, bool has_w,
                           TopoScratch& s, int sidx, bool rows, const TopoTile& tile) {
    const int rowbase = n0 + g;
    const int n1 = imin(N, imax(c1_len, 0));
    if (TOPO_OVF(s)[0] != 0 || N > 0x7FFF) return 1;
    int* Z = s.fl;
    const int H = N + 1;
    int total = 0, rows_total = 0;
    if (rows) wg_exscan2(Z, 2 * H, s.rp, N + 1, s.part, &total, &rows_total);
    else total = wg_exscan(Z, 2 * H, s.part);
    const int C = Z[H], C1 = total - C;
    int* wmax_bits = TOPO_WMAX(s);
    if (with_pool && has_w) {
        // the largest |w| of the graph, for the fixed-point scale of the pooled sums below (non-negative floats order like
        // their bit patterns; one atomic per wave: same-address LDS atomics serialise)
        int mine = 0;
        FOR_TID(e, E) { const int b = topo_f2i(fabsf(s.w0[e])) & 0x7fffffff; mine = b > mine ? b : mine; }
#ifdef DRGNN_EMU
        ATOMIC_MAX(wmax_bits, mine);
#else
        if ((int)(threadIdx.x & ~(DRGNN_WAVE - 1)) < E) {
#pragma unroll
            for (int m = 1; m < DRGNN_WAVE; m <<= 1) { const int o = __shfl_xor(mine, m, DRGNN_WAVE); mine = o > mine ? o : mine; }
            if ((threadIdx.x & (DRGNN_WAVE - 1)) == 0) ATOMIC_MAX(wmax_bits, mine);
        }
#endif
    }
    const int n = imin(C, n1);                      // depth-0 clusters the depth-1 list covers
    const int BW = (C + 31) >> 5, CB = C * BW;
    if (with_pool && 2L * CB + 1 > (long)s.capT) return 2;
    if (c1_len != C) { FOR_TID(i, 1) { topo_flag(tv, DRGNN_S_CLUSTER1_LEN, sidx); } }
    int* csize = s.cp;      // nodes per depth-0 cluster (cleared by topo_lean_prepare)
    int* cl1 = s.mem;       // depth-1 cluster of every depth-0 cluster
    int* key1 = s.mp;       // ... its sort key (depth-1 cluster, id)
    int* bm = s.t2;         // [C][BW] target bitmaps of the pooled rows
    int* bmT = s.t3;        // [C][BW] source bitmaps of the pooled columns
    {
        int32_t* g_cl0 = tv.p[DRGNN_TI_CL0] + n0;
        int32_t* g_cl1 = tv.p[DRGNN_TI_CL1] + n0;
        FOR_TID(i, N) {
            const int c = Z[s.pp[i]];
            s.cl[i] = c; g_cl0[i] = c;
            ATOMIC_ADD(&csize[c], 1);
        }
        FOR_TID(c, n) {
            const int k = Z[H + s.nb[c]] - C;
            cl1[c] = k;
            key1[c] = (k << 16) | c;                // (k < 2^15, c < 2^15)
            g_cl1[c] = k;
        }
        if (with_pool) { FOR_TID(q, CB) { bm[q] = 0; bmT[q] = 0; } }
        if (with_pool && has_w) { FOR_TID(q, 2 * E) { s.seg[q] = 0; } }      // 64-bit accumulators of the pooled weights: [seg | col1]
        FOR_TID(i, 1) { tv.p[DRGNN_TI_NC0][g] = C; tv.p[DRGNN_TI_NC1][g] = C1; }
    }
    if (rows) topo_lean_rows_claim(tv, g, n0, N, E, s);      // (its cursors, s.cur, are this chain's only from the next interval on)
    BARRIER();
    int* qpos = s.pp;       // position of every depth-0 cluster in the depth-1-major order (clusters the list does not cover: last)
    int* mp1 = s.rp1;
    {
        if (with_pool) {
            FOR_TID(e, E) {
                const int r = s.cl[s.er[e]], cc = s.cl[s.ec[e]];
                if (cc != r) {
                    ATOMIC_OR(&bm[r * BW + (cc >> 5)], (int)(1u << (cc & 31)));
                    ATOMIC_OR(&bmT[cc * BW + (r >> 5)], (int)(1u << (r & 31)));
                }
            }
        }
        // (the counting loops start on wave 8: the edge loop above keeps every wave busy, its first trip ends on waves 0 - 7 last)
        constexpr int G = TOPO_LANES(8);
        FOR_TID_FROM(item, C * G, 512) {
            const int c = item / G, sub = item % G;
            int q = C - 1;
            if (c < n) q = topo_count_below<8>(key1, n, key1[c], sub);
            if (sub == 0) qpos[c] = q;
        }
        int32_t* g_mptr1 = tv.p[DRGNN_TI_MPTR1] + rowbase;
        FOR_TID_FROM(item, (C1 + 1) * G, 384) {
            const int k = item / G, sub = item % G;
            const int cnt = topo_count_below<8>(cl1, n, k, sub);
            if (sub == 0) { mp1[k] = cnt; g_mptr1[k] = cnt; }
        }
        FOR_TID(c, C + 1) { s.cur[c] = 0; }      // cursors of the node claim below
    }
    if (rows) topo_lean_rows_emit(tv, e0, E, has_w, s);
    BARRIER();
    int* hmp = s.fl;        // [C + 1] <= capN + 1 <= capF
    {
        int32_t* g_mem1 = tv.p[DRGNN_TI_MEM1] + n0;
        int32_t* g_hmp = tv.p[DRGNN_TI_HMP0] + rowbase;
        FOR_TID(c, n) { g_mem1[qpos[c]] = c; }
        constexpr int G = TOPO_LANES(8);
        FOR_TID(item, (C + 1) * G) {       // first position of the q-th cluster = sizes of the clusters in front of it
            const int q = item / G, sub = item % G;
            int acc = 0;
            for (int c = sub; c < C; c += G) acc += (qpos[c] < q) ? csize[c] : 0;
            acc = topo_group_sum<G>(acc);
            if (sub == 0) { hmp[q] = acc; g_hmp[q] = acc; }
        }
    }
    if (rows && tile.F > 0) topo_tiles_rows(tile, N, s.rp, s.col, has_w ? (const float*)s.wr : nullptr);
    // the three pieces the end of the chain is made of
    auto emit_pooled = [&](const int* Y, const int E1) {
        int32_t* g_col1 = tv.p[DRGNN_TI_COL1] + e0;
        int32_t* g_rowidx1 = tv.p[DRGNN_TI_ROWIDX1] + e0;
        int32_t* g_tslot1 = tv.p[DRGNN_TI_TSLOT1] + e0;
        FOR_TID_FROM(q, 2 * CB, 512) {      // (waves 8 ..: waves 0 .. 6 formed the run offsets, every wave takes part in the scan)
            const bool tr = q >= CB;
            const int qq = tr ? q - CB : q;
            unsigned bits = (unsigned)(tr ? bmT[qq] : bm[qq]);
            const int r = qq / BW, base = (qq - r * BW) << 5;
            int slot = Y[q] - (tr ? E1 : 0);
            int32_t* dst = tr ? g_rowidx1 : g_col1;
            while (bits) {
                const int b = __builtin_ctz(bits);
                bits &= bits - 1;
                dst[slot] = base + b;
                if (tr && has_w) {      // CSR1 slot of this CSC1 entry (column r, row base + b): the set bits of row's bitmap in front of it
                    const int row = base + b, w_ = row * BW + (r >> 5);
                    g_tslot1[slot] = Y[w_] + __builtin_popcount((unsigned)bm[w_] & ((1u << (r & 31)) - 1u));
                }
                ++slot;
            }
        }
        if (has_w) {
            // Pooled weights: the sum of the raw weights of every (row cluster, column cluster) pair.  Order-independent and
            // exact: 64-bit fixed-point atomics at a scale set by the graph's largest |w| (2^40 steps per binade of it: < 2^-40
            // relative to that weight per addend, no overflow below 2^16 edges) -- deterministic without sorting the edges.
            double inv_unused;
            const double scale = topo_wscale(wmax_bits[0], &inv_unused);
            long long* acc = (long long*)s.seg;
            FOR_TID(e, E) {
                const int r = s.cl[s.er[e]], cc = s.cl[s.ec[e]];
                if (cc != r) {
                    const int w_ = r * BW + (cc >> 5);
                    const int slot = Y[w_] + __builtin_popcount((unsigned)bm[w_] & ((1u << (cc & 31)) - 1u));
                    ATOMIC_ADD64(&acc[slot], topo_wfix(s.w0[e], scale));
                }
            }
        }
        int32_t* g_rowptr1 = tv.p[DRGNN_TI_ROWPTR1] + rowbase;
        int32_t* g_colptr1 = tv.p[DRGNN_TI_COLPTR1] + rowbase;
        FOR_TID(r, C + 1) { g_rowptr1[r] = Y[r * BW]; g_colptr1[r] = Y[CB + r * BW] - E1; }
        FOR_TID(i, 1) { tv.p[DRGNN_TI_NE1][g] = E1; }
    };
    auto node_claim = [&]() {
    FOR_TID_FROM(i, N, 768) {
        const int c = s.cl[i], b = qpos[c];
        const int pos = hmp[b] + ATOMIC_ADD(&s.cur[c], 1);
        s.t5[pos] = i;
        s.nb[pos] = b;
    }
    };
    auto hier_tail = [&]() {
    {
        int32_t* g_hord = tv.p[DRGNN_TI_HORD] + n0;
        int32_t* g_ihord = tv.p[DRGNN_TI_IHORD] + n0;
        FOR_TID(p, N) {
            const int me = s.t5[p], b = s.nb[p];
            const int lo = hmp[b], hi = hmp[b + 1];
            int rank = 0;
            for (int q = lo; q < hi; ++q) rank += (s.t5[q] < me) ? 1 : 0;
            g_hord[lo + rank] = me;
            g_ihord[me] = lo + rank;
        }
    }
    {   // split point: the number k of leading depth-1 clusters whose node total is closest to N / 2 (smallest k on ties)
        int32_t* hs = tv.p[DRGNN_TI_HSPLIT] + 4 * g;
#ifdef DRGNN_EMU
        long long best = LLONG_MAX;
        for (int k = 0; k <= C1; ++k) {
            int d = 2 * hmp[imin(mp1[k], C)] - N;
            d = d < 0 ? -d : d;
            const long long v = ((long long)d << 32) | (long long)k;
            best = v < best ? v : best;
        }
        const int kb = (int)(best & 0xffffffffLL);
        const int qb = imin(mp1[kb], C);
        hs[0] = kb; hs[1] = qb; hs[2] = hmp[qb]; hs[3] = C1;
#else
        if (threadIdx.x < DRGNN_WAVE) {
            long long best = LLONG_MAX, other = LLONG_MIN;
            for (int k = threadIdx.x; k <= C1; k += DRGNN_WAVE) {
                int d = 2 * hmp[imin(mp1[k], C)] - N;
                d = d < 0 ? -d : d;
                const long long v = ((long long)d << 32) | (long long)k;
                best = v < best ? v : best;
            }
            wave_minmax_i64(best, other);
            if (threadIdx.x == 0) {
                const int kb = (int)(best & 0xffffffffLL);
                const int qb = imin(mp1[kb], C);
                hs[0] = kb; hs[1] = qb; hs[2] = hmp[qb]; hs[3] = C1;
            }
        }
#endif
    }
    };
#ifndef DRGNN_EMU
    if (with_pool && !has_w && 2 * CB + 1 <= DRGNN_NTHREADS) {
        // Without edge weights the chain ENDS inside the popcount scan's own two barriers (one element per lane): behind the
        // first one the cluster positions (hmp) are visible -> the node claim runs next to the scan's second half; behind
        // the second one the claims are visible -> the rank inside the runs (HORD / IHORD) and the split point are formed
        // next to the pooled graph's emission.  One barrier interval less than scan | emit + claim | barrier | rank
        // (profiles/r05_builder_phases.txt).
        int* Y = s.t4;
        const int t = threadIdx.x, ny = 2 * CB + 1;
        const int lane = t & (DRGNN_WAVE - 1), wave = t >> 6, nw = (ny + DRGNN_WAVE - 1) / DRGNN_WAVE;
        int v = 0, inc = 0;
        if (wave < nw) {
            v = (t < CB) ? __builtin_popcount((unsigned)bm[t]) : (t < 2 * CB) ? __builtin_popcount((unsigned)bmT[t - CB]) : 0;
            inc = wave_incl_scan(v);
            if (lane == DRGNN_WAVE - 1) s.part[wave] = inc;
        }
        BARRIER();
        int base = 0, total = 0;
        for (int w = 0; w < nw; ++w) {
            const int tw = s.part[w];
            base += (w < wave) ? tw : 0;
            total += tw;
        }
        if (t < ny) Y[t] = base + inc - v;
        node_claim();
        BARRIER();
        emit_pooled(Y, total >> 1);
        hier_tail();
        return 0;
    }
#endif
    if (with_pool) {
        int* Y = s.t4;           // [2 CB + 1] set bits before each word of [bm | bmT]
        // (a thread scans the element it wrote while 2 CB + 1 <= threads: no barrier in between)
        FOR_TID(q, 2 * CB + 1) { Y[q] = (q < CB) ? __builtin_popcount((unsigned)bm[q]) : (q < 2 * CB) ? __builtin_popcount((unsigned)bmT[q - CB]) : 0; }
        if (2 * CB + 1 > DRGNN_NTHREADS) BARRIER();
        const int E1 = wg_exscan(Y, 2 * CB + 1, s.part) >> 1;
        emit_pooled(Y, E1);
    } else {
        BARRIER();
    }
    // (the barriers inside the popcount scan have made the run offsets visible: the node claim needs no phase of its own)
    node_claim();
    BARRIER();
    if (with_pool && has_w) {
        double inv;
        (void)topo_wscale(wmax_bits[0], &inv);
        const long long* acc = (const long long*)s.seg;
        float* g_w1 = tv.w1 + e0;
        const int E1 = s.t4[CB];      // (= Y[CB]: the pooled edge count)
        FOR_TID(q, E1) { g_w1[q] = (float)((double)acc[q] * inv); }
    }
    hier_tail();
    return 0;
}

// WEIGHTS: -1 = decided at run time (edge_attr and a weight workspace given); 0 = never (the builder co-launched with a
// GINet / FoutNet step: the weighted pooled-edge path -- bucket ranking, run sums -- is not even compiled into those kernels,
// whose register allocation and code layout it otherwise shapes: +0.5 us on the unweighted builder, +0.2 us on the GINet step)
template <int WEIGHTS = -1>
DEV void topo_graph(const TopoView& tv, const TopoArgs& a, int g, int n0, int n1, int e0, int e1,
                    TopoScratch& s, int role = TOPO_ROLE_ALL) {
    const int N = n1 - n0;
    int E = e1 - e0;
    const int sidx = (role == TOPO_ROLE_MEMBERS) ? a.n_graphs + g : g;
    if (N <= 0 && E > 0) {
        FOR_TID(i, 1) { topo_flag(tv, DRGNN_S_EDGE_RANGE, sidx); }
        E = 0;
    }
    PHASE_MARK();
    const TopoSrc src = topo_src(a, g, n0, e0);
    const bool has_w = (WEIGHTS != 0) && (a.edge_attr != nullptr) && (tv.w0 != nullptr);
    const bool structure = (role != TOPO_ROLE_EDGES);      // CSR0 / CSC0, member lists, node-feature gather
    const bool pool = (role != TOPO_ROLE_MEMBERS);         // pooled graph
    if (structure) topo_gather_rows(a, src, g, n0, N);

    // ---- phase 0: stage the edge list, clear what the next phases accumulate into ----
    // LEAN (request flag, with the hierarchical order and cluster1): the short chains above (which role runs which: see there).
    const bool lean = (a.flags & DRGNN_TOPO_LEAN) != 0 && (a.flags & DRGNN_TOPO_HIER) != 0 && a.cluster0 != nullptr &&
                      a.cluster1 != nullptr && a.c1_ptr != nullptr;
    const int c1_len = lean ? a.c1_ptr[g + 1] - a.c1_ptr[g] : 0;
    const int nc1 = imin(N, imax(c1_len, 0));
    const bool run_rows = lean && structure;
    const bool run_clusters = lean && pool;
    // the aggregation tiles (DRGNN_TOPO_TILES): the x tile of the graph is requested first of all
    const TopoTile tile = topo_tile_of(a, src, n0, structure ? s.xs : nullptr);
    TopoTileRegs treg;
    topo_tile_load(treg, tile, N);
    // L2 touch of the tile rows this workgroup is going to WRITE: a store that misses the XCD's L2 does not allocate the line
    // (profiles/r04_epoch_loop_probes.txt, 6), so the step workgroups of the next launch -- placed on the same XCD -- would read
    // freshly written tiles from memory; a load of the (stale) destination lines allocates them, the stores then update them in
    // place.  Values discarded.  Worth 0.35 us per step over rotating workspaces (profiles/r05_cold_path.txt); nothing where two
    // workspaces alternate (the epoch loop's slots: their lines are resident from two launches ago).  The same for the index
    // arrays the clusters chain writes did not pay (17.78 vs 17.81 us, and +0.1 us on the replayed step).
    TopoTouch touch;
    topo_touch_load(touch, (run_rows || (structure && !lean)) ? tile.ts : nullptr, N * tile.TF);
    // the cluster ids of the clusters chain are requested AHEAD of the edge list (one round trip for all three)
    long long pre0 = 0, pre1 = 0;
    bool pre = false;
#ifndef DRGNN_EMU
    if (run_clusters && N <= DRGNN_NTHREADS) {
        pre = true;
        if ((int)threadIdx.x < N) pre0 = (long long)src.cl0[threadIdx.x];
        if ((int)threadIdx.x < nc1) pre1 = (long long)src.cl1[threadIdx.x];
    }
#endif
    topo_stage_edges(tv, src, sidx, N, E, has_w, s);
    topo_tile_store(treg, tile, N);
    topo_touch_done(touch);
    bool rows_done = false, need_c1 = false;
    if (run_rows || run_clusters) {
        // (one call site per chain: they are inlined)
        topo_lean_prepare(src, N, E, nc1, run_rows, run_clusters, s, pre, pre0, pre1);
        BARRIER();
        if (run_rows && !run_clusters) {
            topo_lean_rows(tv, g, n0, e0, N, E, has_w, s, tile);
            return;
        }
        // (one workgroup per graph: the rows chain rides in the clusters chain's barrier intervals)
        const int rc = topo_lean_clusters(tv, g, n0, e0, N, E, c1_len, true, has_w, s, sidx, run_rows, tile);
        if (rc == 0) return;
        if (run_rows) {
            if (rc == 1) topo_lean_rows(tv, g, n0, e0, N, E, has_w, s, tile);
            else topo_lean_rows_tail(tv, g, n0, e0, N, E, has_w, s, tile);
            rows_done = true;
        }
        need_c1 = true;     // the general chain for the clusters: this workgroup builds depth 1 and the hierarchical order too
        BARRIER();          // (every thread is past its reads of the prepared arrays)
    }
    if (structure && !rows_done) {
        FOR_TID(i, 2 * N + 2) { s.rp[i] = 0; }
        FOR_TID(i, N + 1) { s.cur[i] = 0; }
        FOR_TID(i, N) { s.nb[i] = 0; }
    }
    FOR_TID(v, s.capF) { s.fl[v] = 0; }                       // for the depth-0 cluster ranks
    if (a.cluster0 != nullptr) wg_rank_prepare(src.cl0, N, s);
    BARRIER();

    if (structure && !rows_done) {
        topo_csr0(tv, g, n0, e0, N, E, has_w, s);
        topo_tiles_rows(tile, N, s.rp, s.col, has_w ? (const float*)s.seg : nullptr);
    }
    if (a.cluster0 == nullptr) {          // graph-only build (stand-alone conv layers): no pooling
        FOR_TID(i, 1) { tv.p[DRGNN_TI_NC0][g] = 0; tv.p[DRGNN_TI_NE1][g] = 0; tv.p[DRGNN_TI_NC1][g] = 0; }
        BARRIER();
        return;
    }
    if (role == TOPO_ROLE_MEMBERS) {       // structure: only the cluster COUNT of depth 0 is needed for depth 1
        const int C = topo_clusters0(tv, g, n0, N, src, s, sidx, false);
        topo_clusters1(tv, a, g, n0, N, C, src, s, sidx);
        return;
    }
    // pool (or everything): ranks + member lists of depth 0 (lean with edge weights, two workgroups per graph: ranks only),
    // then the pooled graph
    const bool members0 = !(lean && role == TOPO_ROLE_EDGES);
    const int C = topo_clusters0(tv, g, n0, N, src, s, sidx, members0, true);
    BARRIER();
    topo_pool(tv, g, n0, e0, E, C, has_w, s);
    if (role == TOPO_ROLE_ALL || need_c1) topo_clusters1(tv, a, g, n0, N, C, src, s, sidx);
}
