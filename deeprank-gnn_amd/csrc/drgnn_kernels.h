// drgnn_kernels.h -- launch descriptions, per-workgroup drivers and the __global__ kernels of the library.
//
// Included by drgnn_capi.hip (which defines DRGNN_KERNELS_MAIN: it owns the plain kernels and the host side) and by
// drgnn_step_tu.hip, compiled once per kind of net: the instantiations of the fused step kernel are by far the largest
// part of the build, so three more translation units compile them in parallel (make -j).
#pragma once
#include "drgnn_head.h"
#include "drgnn_step.h"
#include "drgnn_step1.h"
#include "drgnn_step2.h"
#include "drgnn_step3.h"
#include "drgnn_layers.h"
#include "drgnn_mcl.h"
#include "drgnn_collate.h"
#include "drgnn_p2p.h"

// =====================================================================================
// kernels
// =====================================================================================
struct PtrArgs {
    const int64_t* batch;
    const int64_t* edge_row;
    int64_t n_nodes, n_edges;
    int n_graphs;
    int32_t* nptr;
    int32_t* eptr;
    int32_t* err;
};

// per-graph offsets from the (sorted) batch vector and the graph-grouped edge list
DEV void ptr_item(const PtrArgs& a, int64_t i) {
    const int B = a.n_graphs;
    if (i < a.n_nodes) {
        long long b = a.batch[i];
        long long prev = (i > 0) ? (long long)a.batch[i - 1] : -1;
        if (b < prev || b < 0 || b >= B) { ATOMIC_OR(&a.err[0], DRGNN_S_UNSORTED); a.err[1] = 0; b = prev < 0 ? 0 : prev; }
        for (long long gph = prev + 1; gph <= b && gph <= B; ++gph) a.nptr[gph] = (int32_t)i;
        if (i == a.n_nodes - 1)
            for (long long gph = b + 1; gph <= B; ++gph) a.nptr[gph] = (int32_t)a.n_nodes;
    }
    if (i < a.n_edges) {
        auto graph_of = [&](int64_t e) -> long long {
            const long long r = a.edge_row[e];
            if (r < 0 || r >= a.n_nodes) return -2;
            return (long long)a.batch[r];
        };
        long long b = graph_of(i);
        long long prev = (i > 0) ? graph_of(i - 1) : -1;
        if (b == -2 || prev == -2) { ATOMIC_OR(&a.err[0], DRGNN_S_EDGE_RANGE); a.err[1] = 0; if (b == -2) b = prev < 0 ? 0 : prev; if (prev == -2) prev = b; }
        if (b < prev || b >= B) { ATOMIC_OR(&a.err[0], DRGNN_S_UNSORTED); a.err[1] = 0; b = prev; }
        for (long long gph = prev + 1; gph <= b && gph <= B; ++gph) a.eptr[gph] = (int32_t)i;
        if (i == a.n_edges - 1)
            for (long long gph = b + 1; gph <= B; ++gph) a.eptr[gph] = (int32_t)a.n_edges;
    }
    if (i == 0) {
        if (a.n_nodes == 0) for (int gph = 0; gph <= B; ++gph) a.nptr[gph] = 0;
        if (a.n_edges == 0) for (int gph = 0; gph <= B; ++gph) a.eptr[gph] = 0;
    }
}

struct TopoLaunch {
    TopoView tv;
    TopoArgs args;
    const int32_t* user_nptr;   // caller-supplied per-graph offsets (null: k_ptrs filled the workspace)
    const int32_t* user_eptr;
    int32_t* gscratch;     // global scratch (when not in LDS)
    int capN, capE;        // LDS capacities (0 = use global scratch)
    int level1_only;       // second pass: only depth-1 clusters, c1 offsets from NC0
    int roles;             // 1: one workgroup per graph; 2: edge structures / member lists split
    // Cached-topology launches build nothing: the extra workgroups of the launch (CUs the step leaves idle) instead PREFETCH
    // the graphs of the NEXT mini-batch into the L2 of the XCD that will step them (prefetch_block).  pf_ids: their numbers in
    // the cached set (DEVICE memory), null = a builder launch
    const int32_t* pf_ids; int pf_n; int pf_graphs;      // pf_graphs: graphs of the cached set (numbers outside it are skipped)
    const float* pf_tiles; const float* pf_x; int pf_f; int pf_coef;      // tiles / x rows of the set (x: nets that read it), F, D / C too
    long long pf_tile_nodes;
    const void* pf_y; int pf_y_bytes;
};

// ---- L2 prefetch of one graph of a cached set (the launch's workgroup n_net + g: XCD g % 8, where slot g of the next launch
// runs, step_block).  A step on graphs that are not cache-resident waits 1.6 - 1.7 us longer in its prologue than a replayed one
// (profiles/r05_cold_path.txt: the aggregation tiles 1.05 us, the index arrays 0.45 us): first-byte latency, not bandwidth.  In
// cached mode half the CUs are idle: one workgroup per graph of the NEXT mini-batch requests everything that graph's step
// workgroups will stage -- tile rows, (x rows, D, C), the hierarchical order, the pooled level's arrays, counts, target -- and
// drops it; the lines stay in the XCD's L2 across the kernel boundary (that residency is what a replayed mini-batch lives on).
// Reads of valid addresses only; no LDS, no barrier, no effect on any result.
#ifdef DRGNN_EMU
DEV void prefetch_block(const TopoLaunch&, int) {}
#else
DEV void prefetch_block(const TopoLaunch& L, int g) {
    if (g >= L.pf_n) return;
    const int32_t* const* P = L.tv.p;
    const int id = __builtin_amdgcn_readfirstlane(L.pf_ids[g]);
    if (id < 0 || id >= L.pf_graphs) return;      // (the list is the caller's; nothing is read on its word alone)
    const int n0 = P[DRGNN_TI_NPTR][id], n1 = P[DRGNN_TI_NPTR][id + 1];
    const int e0 = P[DRGNN_TI_EPTR][id], e1 = P[DRGNN_TI_EPTR][id + 1];
    const int N = n1 - n0, E = e1 - e0, rowbase = n0 + id, t = threadIdx.x;
    int acc = P[DRGNN_TI_NC0][id] ^ P[DRGNN_TI_NE1][id] ^ P[DRGNN_TI_NC1][id] ^ P[DRGNN_TI_GSTAT][id] ^ P[DRGNN_TI_HSPLIT][4 * id];
    if (L.pf_y) acc ^= (L.pf_y_bytes == 8) ? (int)((const long long*)L.pf_y)[id] : ((const int*)L.pf_y)[id];
    drgnn_f4 f = {0.f, 0.f, 0.f, 0.f};
    // (pf_f: the padded row length pad4(F); pf_x: the rows the step kernels read -- the input's, or the tiles' padded copy)
    const int n4 = (N * L.pf_f) >> 2;
    const drgnn_f4* s4 = (const drgnn_f4*)(L.pf_tiles + (long long)n0 * L.pf_f);
    const drgnn_f4* x4 = L.pf_x ? (const drgnn_f4*)(L.pf_x + (long long)n0 * L.pf_f) : nullptr;
    for (int q = t; q < n4; q += DRGNN_NTHREADS) {
        const drgnn_f4 v = s4[q];
        f[0] += v[0]; f[1] += v[1]; f[2] += v[2]; f[3] += v[3];
        if (x4) { const drgnn_f4 u = x4[q]; f[0] += u[0]; f[1] += u[1]; f[2] += u[2]; f[3] += u[3]; }
    }
    for (int i = t; i <= N; i += DRGNN_NTHREADS) {
        acc ^= P[DRGNN_TI_HMP0][rowbase + i] ^ P[DRGNN_TI_MPTR1][rowbase + i] ^ P[DRGNN_TI_ROWPTR1][rowbase + i] ^
               P[DRGNN_TI_COLPTR1][rowbase + i];
        if (i < N) {
            acc ^= P[DRGNN_TI_IHORD][n0 + i] ^ P[DRGNN_TI_MEM1][n0 + i];
            if (L.pf_coef) {
                const float* td = L.pf_tiles + L.pf_tile_nodes * L.pf_f + n0;
                f[0] += td[i] + td[L.pf_tile_nodes + i];
            }
        }
    }
    for (int e = t; e < E; e += DRGNN_NTHREADS) {
        acc ^= P[DRGNN_TI_COL1][e0 + e] ^ P[DRGNN_TI_ROWIDX1][e0 + e];
        if (L.pf_coef && L.tv.w1) { acc ^= P[DRGNN_TI_TSLOT1][e0 + e]; f[1] += L.tv.w1[e0 + e]; }
    }
    // (keeps the loads: a value no data produces)
    if (acc == 0x7fc12345 && f[0] + f[1] + f[2] + f[3] == 12345.678f) { const_cast<int32_t*>(P[DRGNN_TI_ERR])[1] = acc; }
}
#endif

// LDS is a compile-time property so that every scratch access is a ds_* instruction (a
// run-time choice between LDS and global would make them all flat_* accesses)
template <bool LDS, int WEIGHTS = -1>
DEV void topo_block(const TopoLaunch& L, int blk, int* lds) {
    TopoScratch s;
    // Two workgroups per graph: within full groups of 8 graphs both land on XCD (graph % 8) -- workgroups are dealt
    // round-robin to the 8 XCDs and the step kernels put graph g there too (step_block), so what the builder writes
    // is read back through the same L2 by the launch that trains on it.
    if (L.pf_ids != nullptr) { prefetch_block(L, blk); return; }      // (cached-topology launch: nothing to build)
#ifdef DRGNN_TOPO_EXIT0
    return;      // (experiment: the builder's workgroups are dispatched and return at once)
#endif
    int g = blk, role = TOPO_ROLE_ALL;
    if (L.roles == 2) {
        const int full = (L.args.n_graphs >> 3) << 4;
        if (blk < full) { g = ((blk >> 4) << 3) + (blk & 7); role = ((blk >> 3) & 1) ? TOPO_ROLE_MEMBERS : TOPO_ROLE_EDGES; }
        else { const int t = blk - full; g = (full >> 1) + (t >> 1); role = (t & 1) ? TOPO_ROLE_MEMBERS : TOPO_ROLE_EDGES; }
    }
    const int sidx = (role == TOPO_ROLE_MEMBERS) ? L.args.n_graphs + g : g;
    const int32_t* NP = L.user_nptr ? L.user_nptr : L.tv.p[DRGNN_TI_NPTR];
    const int32_t* EP = L.user_eptr ? L.user_eptr : L.tv.p[DRGNN_TI_EPTR];
    const int n0 = NP[g], n1 = NP[g + 1], N = n1 - n0;
    const int e0 = EP[g], e1 = EP[g + 1], E = e1 - e0;
    if (LDS) {
        const int capT = imax(L.capN, L.capE) + 1;
        s = topo_carve(lds, L.capN, L.capE, capT, L.capN + L.capE + 2);
        // the x tile of the aggregation tiles sits behind the carve (topo_lds_bytes counts it)
        if (L.args.tile_f > 0 && L.args.tiles != nullptr)
            s.xs = (float*)(lds + topo_scratch_ints(L.capN, L.capE, capT, (int64_t)L.capN + L.capE + 2));
    } else {
        // linear placement (see topo_gscratch_base): disjoint regions without a scan
        s = topo_carve(L.gscratch + topo_gscratch_base(n0, e0, g), N, E, N + E + 1, N + E + 2);
    }
    const bool fits = !LDS || (N <= L.capN && E <= L.capE);
    if (!L.level1_only) {
        FOR_TID(i, 1) {
            L.tv.p[DRGNN_TI_GSTAT][sidx] = 0;
            if (role == TOPO_ROLE_ALL) L.tv.p[DRGNN_TI_GSTAT][L.args.n_graphs + g] = 0;
            if (L.user_nptr && role != TOPO_ROLE_MEMBERS) {       // publish the offsets for the kernels that follow
                L.tv.p[DRGNN_TI_NPTR][g] = n0; L.tv.p[DRGNN_TI_NPTR][g + 1] = n1;
                L.tv.p[DRGNN_TI_EPTR][g] = e0; L.tv.p[DRGNN_TI_EPTR][g + 1] = e1;
                if (g == 0) L.tv.p[DRGNN_TI_ERR][0] = 0;
            }
        }
        // (the lean clusters chain sets its presence flags in the builder's first phase: cleared here, before this barrier)
        if (fits && (L.args.flags & DRGNN_TOPO_LEAN)) topo_lean_preclear(N, s, role != TOPO_ROLE_EDGES, role != TOPO_ROLE_MEMBERS);
        BARRIER();
    }
    if (!fits) {   // caller's bound was wrong: refuse loudly
        FOR_TID(i, 1) { topo_flag(L.tv, DRGNN_S_EDGE_RANGE, sidx); }
        return;
    }
    if (!L.level1_only) {
        topo_graph<WEIGHTS>(L.tv, L.args, g, n0, n1, e0, e1, s, role);
    } else {
        // offset of this graph's ids inside cluster1 = number of depth-0 clusters before it
        FOR_TID(i, 1) { s.part[0] = 0; }
        BARRIER();
        FOR_TID(t, DRGNN_NTHREADS) {
            int acc = 0;
            for (int q = t; q < g; q += DRGNN_NTHREADS) acc += L.tv.p[DRGNN_TI_NC0][q];
            if (acc) ATOMIC_ADD(&s.part[0], acc);
        }
        BARRIER();
        const int begin = s.part[0];
        const int C0 = L.tv.p[DRGNN_TI_NC0][g];
        BARRIER();
        int len = C0;
        if (g == L.args.n_graphs - 1 && (int64_t)begin + C0 != L.args.len_cluster1) len = -1;
        if ((int64_t)begin + C0 > L.args.len_cluster1) len = -1;
        topo_graph_level1(L.tv, L.args, g, n0, C0, L.args.cluster1 + begin, len, s, g, false,
                          (L.args.flags & DRGNN_TOPO_HIER) ? L.tv.p[DRGNN_TI_CL0] + n0 : nullptr, N);
    }
}

// aggregation tiles from a built workspace (drgnn_topology_tiles): one workgroup per graph, everything read from global memory
struct TilesArgs { TopoView tv; const float* x; float* tiles; int64_t n_nodes; int n_feat, use_weights; };
DEV void tiles_block(const TilesArgs& a, int g) {
    const int n0 = a.tv.p[DRGNN_TI_NPTR][g], N = a.tv.p[DRGNN_TI_NPTR][g + 1] - n0;
    const int e0 = a.tv.p[DRGNN_TI_EPTR][g];
    const int32_t* rp = a.tv.p[DRGNN_TI_ROWPTR0] + n0 + g;
    const int32_t* col = a.tv.p[DRGNN_TI_COL0] + e0;
    const float* w = (a.use_weights && a.tv.w0) ? a.tv.w0 + e0 : nullptr;
    const int F = a.n_feat, TF = (F + 3) & ~3;      // (rows of the tiles are padded to TF floats: drgnn_topology.h, TopoTile)
    const float* x = a.x + (long long)n0 * F;
    float* ts = a.tiles + (long long)n0 * TF;
    float* td = a.tiles + a.n_nodes * TF + n0;
    float* tc = td + a.n_nodes;
    float* tx = (F & 3) ? a.tiles + drgnn_tiles_x_off(a.n_nodes, TF) + (long long)n0 * TF : nullptr;
    FOR_TID(pad, N * (TF - F)) {
        const int i = pad / (TF - F), f = F + pad % (TF - F);
        ts[(long long)i * TF + f] = 0.0f;
        tx[(long long)i * TF + f] = 0.0f;
    }
    FOR_TID(item, N * F) {
        const int i = item / F, f = item - i * F;
        if (tx) tx[(long long)i * TF + f] = x[(long long)i * F + f];
        const int lo = rp[i], hi = rp[i + 1], deg = hi - lo;
        float acc = 0.0f, asum = 0.0f;
        if (w != nullptr) {
            // (the builder's order: batches of four products, the last one padded under a zero coefficient -- same bits)
            for (int k = lo; k < hi; k += 4)
                for (int j = 0; j < 4; ++j) {
                    const int kk = (k + j < hi) ? k + j : hi - 1;
                    const float cf = ((k + j < hi) ? 1.0f : 0.0f) * w[kk];
                    asum += cf;
                    acc = fmaf(cf, x[(long long)col[kk] * F + f], acc);
                }
        } else {
            for (int k = lo; k < hi; ++k) acc += x[(long long)col[k] * F + f];
        }
        ts[(long long)i * TF + f] = acc;
        if (f == 0) {
            float d, sc;
            if (w != nullptr) { d = 1.0f / (float)(deg > 0 ? deg : 1); sc = asum * d; }
            else { d = deg > 0 ? 1.0f / (float)deg : 0.0f; sc = 1.0f; }
            td[i] = d; tc[i] = sc;
        }
    }
}

struct ScanArgs { TopoView tv; int n_graphs; };
DEV void finalize_block(const ScanArgs& a, int* part) {
    const int B = a.n_graphs;
    const int src[3] = {DRGNN_TI_NC0, DRGNN_TI_NE1, DRGNN_TI_NC1};
    const int dst[3] = {DRGNN_TI_CPTR0, DRGNN_TI_E1PTR, DRGNN_TI_CPTR1};
    for (int w = 0; w < 3; ++w) {
        int32_t* out = a.tv.p[dst[w]];
        const int32_t* in = a.tv.p[src[w]];
        FOR_TID(i, B + 1) { out[i] = (i < B) ? in[i] : 0; }
        BARRIER();
        wg_exscan(out, B + 1, part);
    }
}

struct ReduceArgs {
    const float* partials;
    int n_graphs, n_branch, n_feat, n_partial;
    int kind;
    drgnn_conv_params lay1[DRGNN_MAX_BRANCH], lay2[DRGNN_MAX_BRANCH];   // striding (pointers unused)
    drgnn_conv_grads g1[DRGNN_MAX_BRANCH], g2[DRGNN_MAX_BRANCH];
    float* grad_x; int64_t n_nodes;    // [n_branch][Ntot][F] -> summed into branch 0
};

// where one reduced partial element lives inside the model's own (strided) gradient tensor
DEV float* reduce_dst(const ReduceArgs& a, int br, int p) {
    const int F = a.n_feat;
    const int o_w1s = F * DRGNN_H1, o_b1 = 2 * F * DRGNN_H1, o_w2n = o_b1 + DRGNN_H1;
    const int o_w2s = o_w2n + DRGNN_H1 * DRGNN_H2, o_b2 = o_w2s + DRGNN_H1 * DRGNN_H2;
    if (p < o_w1s)
        return a.g1[br].w_nbr ? a.g1[br].w_nbr + (int64_t)(p / DRGNN_H1) * a.lay1[br].nbr_sk + (int64_t)(p % DRGNN_H1) * a.lay1[br].nbr_sh : nullptr;
    if (p < o_b1) {
        const int q = p - o_w1s;
        return a.g1[br].w_self ? a.g1[br].w_self + (int64_t)(q / DRGNN_H1) * a.lay1[br].self_sk + (int64_t)(q % DRGNN_H1) * a.lay1[br].self_sh : nullptr;
    }
    if (p < o_w2n) return a.g1[br].bias ? a.g1[br].bias + (p - o_b1) : nullptr;
    if (p < o_w2s) {
        const int q = p - o_w2n;
        return a.g2[br].w_nbr ? a.g2[br].w_nbr + (int64_t)(q / DRGNN_H2) * a.lay2[br].nbr_sk + (int64_t)(q % DRGNN_H2) * a.lay2[br].nbr_sh : nullptr;
    }
    if (p < o_b2) {
        const int q = p - o_w2s;
        return a.g2[br].w_self ? a.g2[br].w_self + (int64_t)(q / DRGNN_H2) * a.lay2[br].self_sk + (int64_t)(q % DRGNN_H2) * a.lay2[br].self_sh : nullptr;
    }
    return a.g2[br].bias ? a.g2[br].bias + (p - o_b2) : nullptr;
}
DEV void reduce_write(const ReduceArgs& a, int br, int p, float acc) {
    float* d = reduce_dst(a, br, p);
    if (d) *d = acc;
}

// is this partial slot ever written by the backward kernel?  (GINet has no self / bias terms)
DEV bool reduce_live(const ReduceArgs& a, int p) {
    if (a.kind != DRGNN_GINET) return true;
    const int F = a.n_feat;
    const int o_w1s = F * DRGNN_H1, o_w2n = 2 * F * DRGNN_H1 + DRGNN_H1, o_w2s = o_w2n + DRGNN_H1 * DRGNN_H2;
    return p < o_w1s || (p >= o_w2n && p < o_w2s);
}

// sum of one partial element over the graphs g = first, first+stride, ... (ascending)
DEV float reduce_sum(const ReduceArgs& a, int br, int p, int first, int stride) {
    const float* src = a.partials + (int64_t)br * a.n_partial + p;
    const int64_t step = (int64_t)a.n_branch * a.n_partial;
    float acc = 0.0f;
    // all the loads of a lane's share in flight at once (16 for 64 graphs, 4 waves): the launch is one memory round
    // trip deep instead of four
    float v[16];
    int g = first;
    for (; g + 15 * stride < a.n_graphs; g += 16 * stride) {
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = src[(int64_t)(g + k * stride) * step];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += v[k];
    }
    for (; g < a.n_graphs; g += stride) acc += src[(int64_t)g * step];
    return acc;
}

// ... with slab row g weighted by w[g >> wshift] (drgnn_step_gradients: the slabs hold d pred_g / d theta, w = d loss / d pred)
DEV float reduce_sum_w(const ReduceArgs& a, int br, int p, int first, int stride, const float* w, int wshift) {
    const float* src = a.partials + (int64_t)br * a.n_partial + p;
    const int64_t step = (int64_t)a.n_branch * a.n_partial;
    float acc = 0.0f;
    float v[16], c[16];
    int g = first;
    for (; g + 15 * stride < a.n_graphs; g += 16 * stride) {
#pragma unroll
        for (int k = 0; k < 16; ++k) { v[k] = src[(int64_t)(g + k * stride) * step]; c[k] = w[(g + k * stride) >> wshift]; }
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = fmaf(c[k], v[k], acc);
    }
    for (; g < a.n_graphs; g += stride) acc = fmaf(w[g >> wshift], src[(int64_t)g * step], acc);
    return acc;
}

DEV void reduce_grad_x(const ReduceArgs& a, int64_t item) {
    if (a.grad_x != nullptr && a.n_branch > 1 && item < a.n_nodes * a.n_feat) {
        float acc = a.grad_x[item];
        for (int br = 1; br < a.n_branch; ++br) acc += a.grad_x[(int64_t)br * a.n_nodes * a.n_feat + item];
        a.grad_x[item] = acc;
    }
}

struct NetLaunch {
    NetArgs a;
    float* gscratch;
    int capN, capE, capC;   // LDS capacities (capN == 0 -> global scratch)
    int64_t n_edges;
};

// A body launch that ALSO builds the topology of the next mini-batch: workgroups
// [0, n_net) run the body, [n_net, n_net + n_graphs_next) the topology builder.  The two jobs
// are independent (the builder reads index tensors only), so this hides one of them behind
// the other and saves a kernel boundary -- software pipelining across training steps.
struct CoLaunch {
    NetLaunch net;
    TopoLaunch topo;
    int n_net;
};

// global-scratch placement: net_scratch_words is affine in (capN, capE, capC)
HD int64_t net_gscratch_base(int kind, int F, int64_t n0, int64_t e0, int64_t g, int bwd) {
    const int64_t c0 = net_scratch_words(kind, F, 0, 0, 0, bwd);
    return net_scratch_words(kind, F, n0, e0, n0, bwd) - c0 + c0 * g;
}

template <int KIND, bool BWD, bool LDS>
DEV void net_block(const NetLaunch& L, int blk, float* lds) {
    const int nb = L.a.net.n_branch;
    const int g = blk / nb, br = blk % nb;
    float* scratch;
    int capN, capE, capC;
    if (LDS) {
        scratch = lds; capN = L.capN; capE = L.capE; capC = L.capC;
        const int n0 = L.a.tv.p[DRGNN_TI_NPTR][g], e0 = L.a.tv.p[DRGNN_TI_EPTR][g];
        if (L.a.tv.p[DRGNN_TI_NPTR][g + 1] - n0 > capN || L.a.tv.p[DRGNN_TI_EPTR][g + 1] - e0 > capE ||
            L.a.tv.p[DRGNN_TI_NC0][g] > capC) {
            // the caller's bounds were wrong: poison the output instead of overrunning LDS
            if (!BWD) {
                FOR_TID(c, DRGNN_H2) { L.a.readout[(long)g * DRGNN_H2 * nb + br * DRGNN_H2 + c] = DRGNN_NAN; }
            }
            return;
        }
    } else {
        const int n0 = L.a.tv.p[DRGNN_TI_NPTR][g], e0 = L.a.tv.p[DRGNN_TI_EPTR][g];
        capN = L.a.tv.p[DRGNN_TI_NPTR][g + 1] - n0;
        capE = L.a.tv.p[DRGNN_TI_EPTR][g + 1] - e0;
        capC = capN;
        const int F = L.a.net.n_feat;
        const int64_t per_branch = net_gscratch_base(KIND, F, L.a.n_nodes, L.n_edges, L.a.n_graphs, BWD);
        scratch = L.gscratch + (int64_t)br * per_branch + net_gscratch_base(KIND, F, n0, e0, g, BWD);
    }
    if (BWD) net_backward_graph<KIND>(L.a, g, br, scratch, capN, capE, capC);
    else net_forward_graph<KIND>(L.a, g, br, scratch, capN, capE, capC);
}

// ---- fused training step (drgnn_step.h): one launch for body fwd + head/loss + body bwd, sharing the
// grid with the topology builder of the next mini-batch exactly like CoLaunch above ---------------
// Offsets and sizes of the launch's graphs when the HOST knows them (mini-batches assembled from host-side size tables:
// Batch.from_data_list, the resident set's epoch loops): carried in the kernel arguments, so a workgroup does not have
// to fetch them from the workspace before it can address anything (one dependent memory round trip less).
#define DRGNN_STEP_DIMS_MAX 64
struct StepDims {
    int count;                              // 0: not supplied (sizes come from the workspace tables)
    int32_t n0[DRGNN_STEP_DIMS_MAX], n[DRGNN_STEP_DIMS_MAX], e0[DRGNN_STEP_DIMS_MAX], e[DRGNN_STEP_DIMS_MAX];
    int32_t gi[DRGNN_STEP_DIMS_MAX];        // cached-topology mode: the graph's number in the set
};
struct StepLaunch {
    StepArgs a;
    int capN, capE, capC;
    int64_t words;          // scratch words per workgroup (emulation: one persistent slab each)
    StepDims dims;
};
struct StepCoLaunch {
    StepLaunch step;
    TopoLaunch topo;
    int n_net;
};

// GATHER: cached-topology mode (slot g of the launch = graph gather_ids[g] of a whole-set workspace).  A template
// parameter, not a run-time branch: the per-mini-batch kernels keep exactly the code (and register allocation) they had
template <int KIND, int XF, bool GATHER = false, int CLS = 0>
DEV void step_block(const StepLaunch& L, int blk, float* lds, int part) {
    constexpr int nb = (KIND == DRGNN_GINET) ? 2 : 1;
    // Workgroups go to the 8 XCDs round robin (block id mod 8) and each XCD has its own L2.  The two
    // branch workgroups of a graph read the same x tile and the same topology: they are placed 8
    // block ids apart so that they share an L2 (the second one's misses merge with the first one's).
    int g, br;
    if (nb == 2) { g = ((blk >> 4) << 3) + (blk & 7); br = (blk >> 3) & 1; }
    else { g = blk; br = 0; }
    if (g >= L.a.n_graphs) return;                // padding of the last group of 8 graphs
    if (L.dims.count > 0) {
        // host-supplied offsets / sizes: everything the prologue needs to address its loads is in the kernel arguments;
        // the device-computed counts are requested here and resolved inside (late)
        const int gi = GATHER ? L.dims.gi[g] : g;
        GraphDims d;
        d.n0 = L.dims.n0[g]; d.N = L.dims.n[g]; d.e0 = L.dims.e0[g]; d.E = L.dims.e[g];
        d.rowbase = d.n0 + gi;
        d.C = 0; d.E1 = 0; d.C1 = 0;
        const int cnt_c = L.a.tv.p[DRGNN_TI_NC0][gi], cnt_e1 = L.a.tv.p[DRGNN_TI_NE1][gi], cnt_c1 = L.a.tv.p[DRGNN_TI_NC1][gi];
        // (N <= capN, E <= capE hold by construction: the host derived the capacities from the same table; a cluster
        // count beyond capC can only come from malformed input, which the builder has flagged: such graphs poison
        // their outputs through the status words like any other bad graph, and the loads below stay inside LDS because
        // their bounds are clamped to the capacities)
        net_step_graph<KIND, XF, GATHER, CLS>(L.a, d, g, gi, br, lds, L.capN, L.capE, L.capC, part, true, cnt_c, cnt_e1, cnt_c1);
        return;
    }
    const int gi = GATHER ? WG_UNIFORM(L.a.gather_ids[g]) : g;      // cached mode: graph number in the set
    const GraphDims d = net_dims(L.a.tv, gi);     // ONE round trip for all per-graph sizes
    if (d.N > L.capN || d.E > L.capE || d.C > L.capC) {
        // the caller's bounds were wrong: poison the outputs instead of overrunning LDS
        if (part != 2) {
            const uint32_t tag = (uint32_t)L.a.step2[0] + 1u;
            FOR_TID(c, DRGNN_H2) { const_cast<float*>(L.a.hf.readout)[(long)g * L.a.hf.R + br * DRGNN_H2 + c] = DRGNN_NAN; }
            if (nb > 1) {
                FOR_TID(h, L.a.hf.H) { xchg_publish(L.a.xchg + ((long)g * nb + br) * L.a.hf.H + h, tag, DRGNN_NAN); }
            }
        }
        if (part != 1 && br == 0) {
            if (L.a.hf.train) {
                float* hp = L.a.hf.partials + (long)g * head_compact_floats(L.a.hf.R, L.a.hf.H, L.a.hf.O);
                FOR_TID(i, (int)head_compact_floats(L.a.hf.R, L.a.hf.H, L.a.hf.O)) { hp[i] = DRGNN_NAN; }
            }
            FOR_TID(o, L.a.hf.O) { L.a.hf.pred[(long)g * L.a.hf.O + o] = DRGNN_NAN; }
        }
        return;
    }
    net_step_graph<KIND, XF, GATHER, CLS>(L.a, d, g, gi, br, lds, L.capN, L.capE, L.capC, part);
}

// GINet, one workgroup per graph, both branches one after the other (drgnn_step1.h): the launch layout whenever the
// two-workgroup exchange could meet a non-resident partner (2 B + builder workgroups > CUs).  No cross-workgroup wait.
template <int XF, bool GATHER = false, bool PAIRED = false, int CLS = 0>
DEV void step_block_both(const StepLaunch& L, int g, float* lds) {
    if (g >= L.a.n_graphs) return;
    if (L.dims.count > 0) {
        const int gi = GATHER ? L.dims.gi[g] : g;
        GraphDims d;
        d.n0 = L.dims.n0[g]; d.N = L.dims.n[g]; d.e0 = L.dims.e0[g]; d.E = L.dims.e[g];
        d.rowbase = d.n0 + gi;
        d.C = 0; d.E1 = 0; d.C1 = 0;
        const int cnt_c = L.a.tv.p[DRGNN_TI_NC0][gi], cnt_e1 = L.a.tv.p[DRGNN_TI_NE1][gi], cnt_c1 = L.a.tv.p[DRGNN_TI_NC1][gi];
        net_step_graph_both<XF, GATHER, PAIRED, CLS>(L.a, d, g, gi, lds, L.capN, L.capE, L.capC, true, cnt_c, cnt_e1, cnt_c1);
        return;
    }
    const int gi = GATHER ? WG_UNIFORM(L.a.gather_ids[g]) : g;      // cached mode: graph number in the set
    const GraphDims d = net_dims(L.a.tv, gi);
    if (d.N > L.capN || d.E > L.capE || d.C > L.capC) {
        // the caller's bounds were wrong: poison the outputs instead of overrunning LDS
        FOR_TID(c, L.a.hf.R) { const_cast<float*>(L.a.hf.readout)[(long)g * L.a.hf.R + c] = DRGNN_NAN; }
        if (L.a.hf.train) {
            float* hp = L.a.hf.partials + (long)g * head_compact_floats(L.a.hf.R, L.a.hf.H, L.a.hf.O);
            FOR_TID(i, (int)head_compact_floats(L.a.hf.R, L.a.hf.H, L.a.hf.O)) { hp[i] = DRGNN_NAN; }
        }
        FOR_TID(o, L.a.hf.O) { L.a.hf.pred[(long)g * L.a.hf.O + o] = DRGNN_NAN; }
        return;
    }
    net_step_graph_both<XF, GATHER, PAIRED, CLS>(L.a, d, g, gi, lds, L.capN, L.capE, L.capC);
}

#ifndef DRGNN_EMU
// sGAT / FoutNet, aggregation first, SPLIT workgroups per graph (drgnn_step2.h).  SPLIT = 2: the two halves of a graph are 8
// block ids apart (same XCD: they read the same x tile and topology and hand each other pooled rows), graphs in groups
// of 8 like GINet's branch workgroups.
template <int KIND, int XF, bool GATHER, int CLS, int SPLIT, bool TRAIN, int XG = 0>
DEV void step2_block(const StepLaunch& L, int blk, float* lds) {
    int g, half;
    if (SPLIT == 2) { g = ((blk >> 4) << 3) + (blk & 7); half = (blk >> 3) & 1; }
    else { g = blk; half = 0; }
    if (g >= L.a.n_graphs) return;
    if (L.dims.count > 0) {
        const int gi = GATHER ? L.dims.gi[g] : g;
        GraphDims d;
        d.n0 = L.dims.n0[g]; d.N = L.dims.n[g]; d.e0 = L.dims.e0[g]; d.E = L.dims.e[g];
        d.rowbase = d.n0 + gi;
        d.C = 0; d.E1 = 0; d.C1 = 0;
        const int cnt_c = L.a.tv.p[DRGNN_TI_NC0][gi], cnt_e1 = L.a.tv.p[DRGNN_TI_NE1][gi], cnt_c1 = L.a.tv.p[DRGNN_TI_NC1][gi];
        const int32_t* hs = L.a.tv.p[DRGNN_TI_HSPLIT] + 4 * gi;
        const int hk = (SPLIT == 2) ? hs[0] : 0, hq = (SPLIT == 2) ? hs[1] : 0, hn = (SPLIT == 2) ? hs[2] : 0;
        net_step2_graph<KIND, XF, GATHER, CLS, SPLIT, TRAIN, XG>(L.a, d, g, gi, half, lds, L.capN, L.capE, L.capC, true, cnt_c, cnt_e1,
                                                             cnt_c1, hk, hq, hn);
        return;
    }
    const int gi = GATHER ? WG_UNIFORM(L.a.gather_ids[g]) : g;
    const GraphDims d = net_dims(L.a.tv, gi);
    const int32_t* hs = L.a.tv.p[DRGNN_TI_HSPLIT] + 4 * gi;
    const int hk = (SPLIT == 2) ? WG_UNIFORM(hs[0]) : 0, hq = (SPLIT == 2) ? WG_UNIFORM(hs[1]) : 0, hn = (SPLIT == 2) ? WG_UNIFORM(hs[2]) : 0;
    if (d.N > L.capN || d.E > L.capE || d.C > L.capC) {
        // the caller's bounds were wrong: poison the outputs instead of overrunning LDS (the partner does the same: no wait)
        if (half == 0) {
            FOR_TID(c, L.a.hf.R) { const_cast<float*>(L.a.hf.readout)[(long)g * L.a.hf.R + c] = DRGNN_NAN; }
            if (TRAIN) {
                float* hp = L.a.hf.partials + (long)g * head_compact_floats(L.a.hf.R, L.a.hf.H, L.a.hf.O);
                FOR_TID(i, (int)head_compact_floats(L.a.hf.R, L.a.hf.H, L.a.hf.O)) { hp[i] = DRGNN_NAN; }
            }
            FOR_TID(o, L.a.hf.O) { L.a.hf.pred[(long)g * L.a.hf.O + o] = DRGNN_NAN; }
        }
        return;
    }
    net_step2_graph<KIND, XF, GATHER, CLS, SPLIT, TRAIN, XG>(L.a, d, g, gi, half, lds, L.capN, L.capE, L.capC, false, 0, 0, 0, hk, hq, hn);
}
#endif

#ifndef DRGNN_EMU
// GINet, aggregation first (drgnn_step3.h): two branch workgroups per graph, placed like step_block's
template <int XF, bool GATHER, int CLS, bool TRAIN>
DEV void step3_block(const StepLaunch& L, int blk, float* lds) {
    const int g = ((blk >> 4) << 3) + (blk & 7), br = (blk >> 3) & 1;
    if (g >= L.a.n_graphs) return;
    if (L.dims.count > 0) {
        const int gi = GATHER ? L.dims.gi[g] : g;
        GraphDims d;
        d.n0 = L.dims.n0[g]; d.N = L.dims.n[g]; d.e0 = L.dims.e0[g]; d.E = L.dims.e[g];
        d.rowbase = d.n0 + gi;
        d.C = 0; d.E1 = 0; d.C1 = 0;
        const int cnt_c = L.a.tv.p[DRGNN_TI_NC0][gi], cnt_e1 = L.a.tv.p[DRGNN_TI_NE1][gi], cnt_c1 = L.a.tv.p[DRGNN_TI_NC1][gi];
        net_step3_graph<XF, GATHER, CLS, TRAIN>(L.a, d, g, gi, br, lds, L.capN, L.capE, L.capC, true, cnt_c, cnt_e1, cnt_c1);
        return;
    }
    const int gi = GATHER ? WG_UNIFORM(L.a.gather_ids[g]) : g;
    const GraphDims d = net_dims(L.a.tv, gi);
    if (d.N > L.capN || d.E > L.capE || d.C > L.capC) {
        // the caller's bounds were wrong: poison the outputs instead of overrunning LDS (as step_block does)
        const uint32_t tag = (uint32_t)L.a.step2[0] + 1u;
        FOR_TID(c, DRGNN_H2) { const_cast<float*>(L.a.hf.readout)[(long)g * L.a.hf.R + br * DRGNN_H2 + c] = DRGNN_NAN; }
        FOR_TID(h, L.a.hf.H) { xchg_publish(L.a.xchg + ((long)g * 2 + br) * L.a.hf.H + h, tag, DRGNN_NAN); }
        if (br == 0) {
            if (TRAIN) {
                float* hp = L.a.hf.partials + (long)g * head_compact_floats(L.a.hf.R, L.a.hf.H, L.a.hf.O);
                FOR_TID(i, (int)head_compact_floats(L.a.hf.R, L.a.hf.H, L.a.hf.O)) { hp[i] = DRGNN_NAN; }
            }
            FOR_TID(o, L.a.hf.O) { L.a.hf.pred[(long)g * L.a.hf.O + o] = DRGNN_NAN; }
        }
        return;
    }
    net_step3_graph<XF, GATHER, CLS, TRAIN>(L.a, d, g, gi, br, lds, L.capN, L.capE, L.capC, false, 0, 0, 0);
}
#endif

#ifndef DRGNN_EMU
// GINet, aggregation first, both branches of a graph in one workgroup (drgnn_step3.h, net_step3_graph_both)
template <int XF, bool GATHER, int CLS, bool TRAIN, bool SG = false>
DEV void step3b_block(const StepLaunch& L, int g, float* lds) {
    // the capacity-class training instance of the 32-wide kernel works the two branches off side by side (drgnn_step3.h: DUAL;
    // step_pick sizes the launch's LDS for it)
    constexpr bool DUAL = !SG && STEP3B_DUAL(XF, CLS, TRAIN);
    if (g >= L.a.n_graphs) return;
    if (L.dims.count > 0) {
        const int gi = GATHER ? L.dims.gi[g] : g;
        GraphDims d;
        d.n0 = L.dims.n0[g]; d.N = L.dims.n[g]; d.e0 = L.dims.e0[g]; d.E = L.dims.e[g];
        d.rowbase = d.n0 + gi;
        d.C = 0; d.E1 = 0; d.C1 = 0;
        const int cnt_c = L.a.tv.p[DRGNN_TI_NC0][gi], cnt_e1 = L.a.tv.p[DRGNN_TI_NE1][gi], cnt_c1 = L.a.tv.p[DRGNN_TI_NC1][gi];
        net_step3_graph_both<XF, GATHER, CLS, TRAIN, DUAL, SG>(L.a, d, g, gi, lds, L.capN, L.capE, L.capC, true, cnt_c, cnt_e1, cnt_c1);
        return;
    }
    const int gi = GATHER ? WG_UNIFORM(L.a.gather_ids[g]) : g;
    const GraphDims d = net_dims(L.a.tv, gi);
    if (d.N > L.capN || d.E > L.capE || d.C > L.capC) {
        // the caller's bounds were wrong: poison the outputs instead of overrunning LDS
        FOR_TID(c, L.a.hf.R) { const_cast<float*>(L.a.hf.readout)[(long)g * L.a.hf.R + c] = DRGNN_NAN; }
        if (TRAIN) {
            float* hp = L.a.hf.partials + (long)g * head_compact_floats(L.a.hf.R, L.a.hf.H, L.a.hf.O);
            FOR_TID(i, (int)head_compact_floats(L.a.hf.R, L.a.hf.H, L.a.hf.O)) { hp[i] = DRGNN_NAN; }
        }
        FOR_TID(o, L.a.hf.O) { L.a.hf.pred[(long)g * L.a.hf.O + o] = DRGNN_NAN; }
        return;
    }
    net_step3_graph_both<XF, GATHER, CLS, TRAIN, DUAL, SG>(L.a, d, g, gi, lds, L.capN, L.capE, L.capC, false, 0, 0, 0);
}
#endif

// ---- single-launch parameter update: reduce the conv + head partials and apply Adam --------
struct UpdateArgs {
    ReduceArgs r;
    HeadReduceArgs h;       // h.grad = flat gradient block of the FC head; h.step unused here
    AdamArgs ad;            // flat buffers; ad.grad = base of the flat gradient
    int conv_blocks;        // blocks [0, conv_blocks) reduce conv partials, the rest the head's
    int blocks_per_branch;  // conv blocks of one branch (a block never straddles branches: the branch is block-uniform)
    int apply_adam;         // 0: only produce the flat gradient (data parallel: all-reduce comes next)
    // fused-step mode: head slabs are compact ([dhid H][dW2][db2][loss][weight], u.h.P floats each) and
    // dW_fc1[h][r] = sum_g dhid[g][h] * readout[g][r] is formed here
    const float* readout;   // [n_wg][R] or null (legacy slabs that already hold dW_fc1)
    int hR, hH;
    int32_t* step2;         // non-null: commit step2[0] = step2[1] (the step index Adam just used)
    float* loss2;           // optional second destination of the batch loss (the epoch loop's last mini-batch: the trainer's loss word)
};

DEV void update_store(const UpdateArgs& u, float* dst, float g) {
    *dst = g;                                   // keep p.grad inspectable
    if (u.apply_adam) adam_item(u.ad, (int64_t)(dst - u.ad.grad));
}

// number of head items one update launch produces: the gradient block + the loss
DEV int update_head_items(const UpdateArgs& u) { return (u.readout ? u.hH * u.hR : 0) + u.h.P - 1; }
// sum of one head-partial element over the slabs w = first, first+stride, ...
DEV float update_head_sum(const UpdateArgs& u, int item, int first, int stride) {
    float acc = 0.0f;
    if (u.readout) {
        const int HR = u.hH * u.hR;
        if (item < HR) {
            const int h = item / u.hR, r = item - h * u.hR;
            const float* dh = u.h.partials + h;
            const float* xr = u.readout + r;
            float va[16], vb[16];
            int w = first;
            for (; w + 15 * stride < u.h.n_wg; w += 16 * stride) {
#pragma unroll
                for (int k = 0; k < 16; ++k) { va[k] = dh[(long)(w + k * stride) * u.h.P]; vb[k] = xr[(long)(w + k * stride) * u.hR]; }
#pragma unroll
                for (int k = 0; k < 16; ++k) acc = fmaf(va[k], vb[k], acc);
            }
            for (; w < u.h.n_wg; w += stride) acc = fmaf(dh[(long)w * u.h.P], xr[(long)w * u.hR], acc);
            return acc;
        }
        item -= HR;
    }
    const float* src = u.h.partials + item;
    float v[16];
    int w = first;
    for (; w + 15 * stride < u.h.n_wg; w += 16 * stride) {
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = src[(long)(w + k * stride) * u.h.P];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += v[k];
    }
    for (; w < u.h.n_wg; w += stride) acc += src[(long)w * u.h.P];
    return acc;
}
// ... with slab w weighted by gw[w] (see reduce_sum_w)
DEV float update_head_sum_w(const UpdateArgs& u, int item, int first, int stride, const float* gw) {
    float acc = 0.0f;
    if (u.readout) {
        const int HR = u.hH * u.hR;
        if (item < HR) {
            const int h = item / u.hR, r = item - h * u.hR;
            const float* dh = u.h.partials + h;
            const float* xr = u.readout + r;
            // (all the loads of a lane's share in flight at once, as update_head_sum)
            float va[16], vb[16], vc[16];
            int w = first;
            for (; w + 15 * stride < u.h.n_wg; w += 16 * stride) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    va[k] = dh[(long)(w + k * stride) * u.h.P]; vb[k] = xr[(long)(w + k * stride) * u.hR]; vc[k] = gw[w + k * stride];
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) acc = fmaf(vc[k] * va[k], vb[k], acc);
            }
            for (; w < u.h.n_wg; w += stride) acc = fmaf(gw[w] * dh[(long)w * u.h.P], xr[(long)w * u.hR], acc);
            return acc;
        }
        item -= HR;
    }
    const float* src = u.h.partials + item;
    float v[16], c[16];
    int w = first;
    for (; w + 15 * stride < u.h.n_wg; w += 16 * stride) {
#pragma unroll
        for (int k = 0; k < 16; ++k) { v[k] = src[(long)(w + k * stride) * u.h.P]; c[k] = gw[w + k * stride]; }
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = fmaf(c[k], v[k], acc);
    }
    for (; w < u.h.n_wg; w += stride) acc = fmaf(gw[w], src[(long)w * u.h.P], acc);
    return acc;
}
struct GradArgs {
    UpdateArgs u;                  // (u.ad unused, u.apply_adam = 0)
    const float* graph_weight;     // [n_graphs] or null
    int wshift;                    // conv slab row -> graph: row >> wshift (1 for the split layout's two half-graph slabs)
    int n_zero;
    float* zero_ptr[DRGNN_ZERO_RANGES];
    int64_t zero_len[DRGNN_ZERO_RANGES];
};
DEV void update_head_store(const UpdateArgs& u, int item, float acc) {
    const int n_grad = update_head_items(u) - 1;
    if (item < n_grad) update_store(u, u.h.grad + item, acc);
    else if (item == n_grad && u.h.loss) { u.h.loss[0] = acc; if (u.loss2) u.loss2[0] = acc; }
}

#ifndef DRGNN_EMU
#ifdef DRGNN_KERNELS_MAIN
__global__ void __launch_bounds__(256) k_ptrs(PtrArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    ptr_item(a, i);
}
#endif  // DRGNN_KERNELS_MAIN
// One batch of scalar loads over a range of the kernel-argument segment (as co_kernarg_touch below does for the co-launch's
// arguments): one word of each of the N 64-byte lines from byte offset off0 (rounded down to a line) on, results discarded.  The co-launched BUILDER's workgroups read the topo part of StepCoLaunch (0x230 bytes behind 2 KB of step
// arguments) in five or six dependent rounds of scalar loads -- first touches of a line each; when the argument block is not
// cache-resident (every launch of an epoch loop, a long recorded graph: the block was written by the host or last read
// milliseconds ago) each round is a trip to memory, on the chain that is the launch's tail (tools/r05/cached_prefetch_probe.py:
// steps on the SAME graphs cost 16.8 us inside a long graph against 15.6 in a short one).
template <int N>
DEV void kernarg_touch_lines(int off0) {
    static_assert(N >= 1 && N <= 12, "lines per call");
    const char* base = (const char*)__builtin_amdgcn_kernarg_segment_ptr() + (off0 & ~63);
    int t;
    // ONE asm statement with its own wait: the compiler does not know that the destination is written late
    // (lines past N - 1 touch line N - 1 again)
#define DRGNN_KA_OFF(k) "i"(64 * ((k) < N ? (k) : N - 1))
    asm volatile("s_load_dword %0, %1, %2\n s_load_dword %0, %1, %3\n s_load_dword %0, %1, %4\n s_load_dword %0, %1, %5\n"
                 "s_load_dword %0, %1, %6\n s_load_dword %0, %1, %7\n s_load_dword %0, %1, %8\n s_load_dword %0, %1, %9\n"
                 "s_load_dword %0, %1, %10\n s_load_dword %0, %1, %11\n s_load_dword %0, %1, %12\n s_load_dword %0, %1, %13\n"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(t)
                 : "s"(base), DRGNN_KA_OFF(0), DRGNN_KA_OFF(1), DRGNN_KA_OFF(2), DRGNN_KA_OFF(3), DRGNN_KA_OFF(4), DRGNN_KA_OFF(5),
                   DRGNN_KA_OFF(6), DRGNN_KA_OFF(7), DRGNN_KA_OFF(8), DRGNN_KA_OFF(9), DRGNN_KA_OFF(10), DRGNN_KA_OFF(11)
                 : "memory");
#undef DRGNN_KA_OFF
}
#define DRGNN_KA_LINES(first_byte, end_byte) ((((end_byte) - 1) >> 6) - ((first_byte) >> 6) + 1)
template <bool LDS>
__global__ void __launch_bounds__(DRGNN_NTHREADS) k_topo(TopoLaunch L) {
    extern __shared__ __attribute__((aligned(16))) int smem_i[];
    PHASE_BEGIN();
    kernarg_touch_lines<DRGNN_KA_LINES(0, (int)sizeof(TopoLaunch))>(0);
    topo_block<LDS>(L, blockIdx.x, smem_i);
}
#ifdef DRGNN_KERNELS_MAIN
__global__ void __launch_bounds__(DRGNN_NTHREADS) k_tiles(TilesArgs a) { tiles_block(a, blockIdx.x); }
__global__ void __launch_bounds__(DRGNN_NTHREADS) k_finalize(ScanArgs a) {
    __shared__ int part[DRGNN_NTHREADS + 4];
    finalize_block(a, part);
}
// 64 gradient elements per workgroup; the 4 waves sum interleaved quarters of the graphs, the
// quarter sums are combined in fixed order -> deterministic, and 4x16 loads in flight per lane
__global__ void __launch_bounds__(256) k_reduce(ReduceArgs a, int64_t n_items) {
    __shared__ float quarter[4][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    // (the branch of a block is uniform, as in k_update: ceil(n_partial / 64) blocks per branch)
    const int bpb = (a.n_partial + 63) / 64;
    int br = 0, blk = (int)blockIdx.x;
    while (blk >= bpb && br + 1 < a.n_branch) { blk -= bpb; ++br; }
    const int p_raw = blk * 64 + lane;
    const bool live = p_raw < a.n_partial && reduce_live(a, p_raw);
    const int p = live ? p_raw : 0;
    quarter[q][lane] = live ? reduce_sum(a, br, p, q, 4) : 0.0f;
    __syncthreads();
    if (q == 0) {
        if (live) reduce_write(a, br, p, (quarter[0][lane] + quarter[1][lane]) + (quarter[2][lane] + quarter[3][lane]));
    } else {
        // the other waves fold the per-branch d loss / d x slabs (only when it was requested)
        for (int64_t e = (int64_t)blockIdx.x * 192 + (threadIdx.x - 64); e < a.n_nodes * a.n_feat && a.grad_x; e += (int64_t)gridDim.x * 192)
            reduce_grad_x(a, e);
    }
}
#endif  // DRGNN_KERNELS_MAIN
template <int KIND, bool BWD, bool LDS>
__global__ void __launch_bounds__(DRGNN_NTHREADS) k_net(NetLaunch L) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    PHASE_BEGIN();
    net_block<KIND, BWD, LDS>(L, blockIdx.x, smem_f);
}
template <int KIND, bool BWD>
__global__ void __launch_bounds__(DRGNN_NTHREADS) k_net_co_topo(CoLaunch C) {
    extern __shared__ __attribute__((aligned(16))) float smem_c[];
    PHASE_BEGIN();
    if ((int)blockIdx.x < C.n_net) net_block<KIND, BWD, true>(C.net, blockIdx.x, smem_c);
    else topo_block<true>(C.topo, (int)blockIdx.x - C.n_net, (int*)smem_c);
}
// The step workgroups read ~0.8 KB of kernel arguments (descriptors, pointers, strides) in the order the prologue's code
// happens to need them: every first touch of a 64-byte line is a scalar-cache miss the next instructions wait for, one
// after the other.  One batch of scalar loads (one word of every line, results discarded) makes it a single miss time:
// co_kernarg_touch below (round 2: the step's lines; round 5: + the builder's, before the role is known).
// The step kernels read their arguments through the kernel-argument segment pointer, not through the by-value parameter:
// with run-time indices into its arrays (dims) the compiler may keep a PRIVATE COPY of the whole 2.4 KB argument block in
// scratch memory (2.5 KB of scratch per lane, a ~10x slower kernel).  Which instances are hit changes with unrelated edits
// and with -O3 / -Os: in round 3 the generic-width sGAT kernels were (profiles/r03_kernarg_scratch.txt).
DEV const StepCoLaunch& step_kernarg() {
    return *reinterpret_cast<const StepCoLaunch*>((const char*)__builtin_amdgcn_kernarg_segment_ptr());
}
// Both parts of a co-launch's arguments in ONE batch, before the workgroup knows its role (n_net lies in the last line): the step's
// lines [0, 0x340) (descriptors, pointers, strides, the head: everything but the per-graph dims) and the builder's.
DEV void co_kernarg_touch() {
    constexpr int first = (int)offsetof(StepCoLaunch, topo), end = (int)sizeof(StepCoLaunch);
    constexpr int N0 = 13, N1 = DRGNN_KA_LINES(first, end);
    static_assert(offsetof(StepLaunch, dims) + sizeof(int) >= 0x2c0 && sizeof(StepLaunch) >= 0x340, "the touched lines lie inside the step arguments");
    static_assert(N1 >= 1 && N1 <= 13, "one asm statement covers 13 lines per part");
    const char* b0 = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
    const char* b1 = b0 + (first & ~63);
    int t;
#define DRGNN_KA_OFF(k, N) "i"(64 * ((k) < N ? (k) : N - 1))
#define DRGNN_KA_13(b) "s_load_dword %0, " b ", %3\n s_load_dword %0, " b ", %4\n s_load_dword %0, " b ", %5\n s_load_dword %0, " b ", %6\n" \
                       "s_load_dword %0, " b ", %7\n s_load_dword %0, " b ", %8\n s_load_dword %0, " b ", %9\n s_load_dword %0, " b ", %10\n" \
                       "s_load_dword %0, " b ", %11\n s_load_dword %0, " b ", %12\n s_load_dword %0, " b ", %13\n s_load_dword %0, " b ", %14\n" \
                       "s_load_dword %0, " b ", %15\n"
    // (offsets past a part's last line are clamped to that line: it is touched again)
    asm volatile(DRGNN_KA_13("%1")
                 "s_load_dword %0, %2, %16\n s_load_dword %0, %2, %17\n s_load_dword %0, %2, %18\n s_load_dword %0, %2, %19\n"
                 "s_load_dword %0, %2, %20\n s_load_dword %0, %2, %21\n s_load_dword %0, %2, %22\n s_load_dword %0, %2, %23\n"
                 "s_load_dword %0, %2, %24\n s_load_dword %0, %2, %25\n s_load_dword %0, %2, %26\n s_load_dword %0, %2, %27\n"
                 "s_load_dword %0, %2, %28\n s_waitcnt lgkmcnt(0)"
                 : "=&s"(t)
                 : "s"(b0), "s"(b1),
                   DRGNN_KA_OFF(0, N0), DRGNN_KA_OFF(1, N0), DRGNN_KA_OFF(2, N0), DRGNN_KA_OFF(3, N0), DRGNN_KA_OFF(4, N0), DRGNN_KA_OFF(5, N0),
                   DRGNN_KA_OFF(6, N0), DRGNN_KA_OFF(7, N0), DRGNN_KA_OFF(8, N0), DRGNN_KA_OFF(9, N0), DRGNN_KA_OFF(10, N0), DRGNN_KA_OFF(11, N0),
                   DRGNN_KA_OFF(12, N0),
                   DRGNN_KA_OFF(0, N1), DRGNN_KA_OFF(1, N1), DRGNN_KA_OFF(2, N1), DRGNN_KA_OFF(3, N1), DRGNN_KA_OFF(4, N1), DRGNN_KA_OFF(5, N1),
                   DRGNN_KA_OFF(6, N1), DRGNN_KA_OFF(7, N1), DRGNN_KA_OFF(8, N1), DRGNN_KA_OFF(9, N1), DRGNN_KA_OFF(10, N1), DRGNN_KA_OFF(11, N1),
                   DRGNN_KA_OFF(12, N1)
                 : "memory");
#undef DRGNN_KA_13
#undef DRGNN_KA_OFF
}
// Which workgroups of a co-launch are the step's and which the builder's (or, cached mode, the prefetcher's).
// DRGNN_TOPO_FIRST (experiment switch): the builder's workgroups take the FIRST block ids -- they are dispatched first, and
// the builder's chain is what a rebuilt launch ends with (profiles/r05_builder_first.txt).  Block ids keep their XCD (id % 8)
// as long as the builder's workgroup count is a multiple of 8.
#ifdef DRGNN_TOPO_FIRST
#define STEP_CO_ROLES(C)                                                       \
    const int co_extra_ = (int)gridDim.x - (C).n_net;                          \
    const bool co_is_step_ = (int)blockIdx.x >= co_extra_;                     \
    const int co_step_blk_ = (int)blockIdx.x - co_extra_, co_topo_blk_ = (int)blockIdx.x
#else
#define STEP_CO_ROLES(C)                                                       \
    const bool co_is_step_ = (int)blockIdx.x < (C).n_net;                      \
    const int co_step_blk_ = (int)blockIdx.x, co_topo_blk_ = (int)blockIdx.x - (C).n_net
#endif
template <int KIND, int XF, bool GATHER, int CLS = 0>
__global__ void __launch_bounds__(DRGNN_NTHREADS) k_step_co_topo(StepCoLaunch C_by_value) {
    extern __shared__ __attribute__((aligned(16))) float smem_s[];
    PHASE_BEGIN();
    const StepCoLaunch& C = step_kernarg();
    co_kernarg_touch();
    STEP_CO_ROLES(C);
    if (co_is_step_) step_block<KIND, XF, GATHER, CLS>(C.step, co_step_blk_, smem_s, 0);
    else topo_block<true, (KIND == DRGNN_SGAT) ? -1 : 0>(C.topo, co_topo_blk_, (int*)smem_s);      // (train_step_impl keeps weighted requests of the other kinds out of the launch)
}
// GINet, one workgroup per graph (both branches), + the builder's workgroups of the next mini-batch
// PAIRED: both branches share every phase (drgnn_step1.h); instantiated for the generic and the 32-wide kernels
template <int XF, bool GATHER, bool PAIRED, int CLS = 0>
__global__ void __launch_bounds__(DRGNN_NTHREADS) k_step1_co_topo(StepCoLaunch C_by_value) {
    extern __shared__ __attribute__((aligned(16))) float smem_s1[];
    PHASE_BEGIN();
    const StepCoLaunch& C = step_kernarg();
    co_kernarg_touch();
    STEP_CO_ROLES(C);
    if (co_is_step_) step_block_both<XF, GATHER, PAIRED, CLS>(C.step, co_step_blk_, smem_s1);
    else topo_block<true, 0>(C.topo, co_topo_blk_, (int*)smem_s1);
}
// sGAT / FoutNet, aggregation first, SPLIT workgroups per graph (drgnn_step2.h) + the builder's workgroups
template <int KIND, int XF, bool GATHER, int CLS, int SPLIT, bool TRAIN, int XG = 0>
__global__ void __launch_bounds__(DRGNN_NTHREADS) k_step2_co_topo(StepCoLaunch C_by_value) {
    extern __shared__ __attribute__((aligned(16))) float smem_s2[];
    PHASE_BEGIN();
    const StepCoLaunch& C = step_kernarg();
    co_kernarg_touch();
    STEP_CO_ROLES(C);
    if (co_is_step_) step2_block<KIND, XF, GATHER, CLS, SPLIT, TRAIN, XG>(C.step, co_step_blk_, smem_s2);
    else topo_block<true, (KIND == DRGNN_SGAT) ? -1 : 0>(C.topo, co_topo_blk_, (int*)smem_s2);
}
// GINet, aggregation first (drgnn_step3.h) + the builder's workgroups
template <int XF, bool GATHER, int CLS, bool TRAIN>
__global__ void __launch_bounds__(DRGNN_NTHREADS) k_step3_co_topo(StepCoLaunch C_by_value) {
    extern __shared__ __attribute__((aligned(16))) float smem_s3[];
    PHASE_BEGIN();
    const StepCoLaunch& C = step_kernarg();
    co_kernarg_touch();
    STEP_CO_ROLES(C);
    if (co_is_step_) step3_block<XF, GATHER, CLS, TRAIN>(C.step, co_step_blk_, smem_s3);
    else topo_block<true, 0>(C.topo, co_topo_blk_, (int*)smem_s3);
}
// ... its form with both branches of a graph in one workgroup (beyond the resident batch size)
// SG: the S-from-memory form (drgnn_step3.h: graphs beyond the staged form's LDS budget)
template <int XF, bool GATHER, int CLS, bool TRAIN, bool SG = false>
__global__ void __launch_bounds__(DRGNN_NTHREADS) k_step3b_co_topo(StepCoLaunch C_by_value) {
    extern __shared__ __attribute__((aligned(16))) float smem_s3b[];
    PHASE_BEGIN();
    const StepCoLaunch& C = step_kernarg();
    co_kernarg_touch();
    STEP_CO_ROLES(C);
    if (co_is_step_) step3b_block<XF, GATHER, CLS, TRAIN, SG>(C.step, co_step_blk_, smem_s3b);
    else topo_block<true, 0>(C.topo, co_topo_blk_, (int*)smem_s3b);
}
#ifdef DRGNN_KERNELS_MAIN
__global__ void __launch_bounds__(DRGNN_NTHREADS) k_conv_gemm(ConvLayerArgs a) { conv_gemm_block(a, blockIdx.x); }
__global__ void __launch_bounds__(256) k_conv_aggregate(ConvLayerArgs a) {
    conv_aggregate_item(a, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}
__global__ void __launch_bounds__(256) k_conv_bwd_du(ConvLayerArgs a) {
    conv_bwd_du_item(a, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}
__global__ void __launch_bounds__(DRGNN_NTHREADS) k_conv_bwd_dw(ConvLayerArgs a) { conv_bwd_dw_block(a, blockIdx.x); }
__global__ void __launch_bounds__(256) k_conv_reduce(ConvReduceArgs a) {
    conv_reduce_item(a, blockIdx.x * blockDim.x + threadIdx.x);
}
__global__ void __launch_bounds__(256) k_conv_bwd_dx(ConvLayerArgs a) {
    conv_bwd_dx_item(a, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}
__global__ void __launch_bounds__(DRGNN_NTHREADS) k_segpool_fwd(SegPoolArgs a) { segpool_fwd_block(a, blockIdx.x); }
__global__ void __launch_bounds__(256) k_segmax_bwd(SegPoolArgs a, int64_t n_items, int64_t n_nodes) {
    segmax_bwd_item(a, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, n_items, n_nodes);
}
__global__ void __launch_bounds__(DRGNN_NTHREADS) k_edge_export(EdgeExportArgs a) { edge_export_block(a, blockIdx.x); }
__global__ void __launch_bounds__(DRGNN_NTHREADS) k_cluster_max(ClusterOffsetArgs a) {
    __shared__ long long mm[2];
    cluster_max_block(a, blockIdx.x, mm);
}
__global__ void k_cluster_scan(ClusterOffsetArgs a) { cluster_scan_single(a); }
__global__ void __launch_bounds__(DRGNN_NTHREADS) k_cluster_add(ClusterOffsetArgs a) { cluster_add_block(a, blockIdx.x); }
__global__ void __launch_bounds__(DRGNN_NTHREADS) k_mcl(MclArgs a) {
    __shared__ double red[DRGNN_NTHREADS];
    __shared__ int flag[2];
    mcl_graph(a, blockIdx.x, flag, red);
}
__global__ void __launch_bounds__(DRGNN_NTHREADS) k_graclus(GraclusArgs a) {
    extern __shared__ __attribute__((aligned(16))) int smem_g[];
    graclus_block(a, blockIdx.x, smem_g);
}
__global__ void __launch_bounds__(DRGNN_NTHREADS) k_batch_offsets(OffsetsArgs a) {
    extern __shared__ __attribute__((aligned(16))) int smem_o[];
    batch_offsets_block(a, blockIdx.x, smem_o + DRGNN_NTHREADS + 4, smem_o);
}
__global__ void __launch_bounds__(DRGNN_NTHREADS) k_collate(CollateArgs a) {
    __shared__ int sh[4];
    collate_block(a, blockIdx.x, sh);
}
__global__ void __launch_bounds__(DRGNN_NTHREADS) k_head(HeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem_h[];
    head_block(a, blockIdx.x, smem_h);
}
__global__ void __launch_bounds__(256) k_head_reduce(HeadReduceArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.P - 1) head_reduce_item(a, i);
}
// 64 gradient elements per workgroup; the 4 waves sum interleaved quarters of the partial
// slabs, the quarter sums are combined in fixed order, then Adam is applied to that element
// A FIFTH wave (q == 4) forms Adam's bias corrections meanwhile -- a chain of ~35 dependent double-precision operations
// behind the load of the step index, which used to sit in front of wave 0's slab loads -- and hands them over through LDS.
#define DRGNN_UPDATE_THREADS 320
__global__ void __launch_bounds__(DRGNN_UPDATE_THREADS) k_update(UpdateArgs u) {
    kernarg_touch_lines<DRGNN_KA_LINES(0, (int)sizeof(UpdateArgs))>(0);
    __shared__ float quarter[4][64];
    __shared__ float bias_scalars[2];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const bool conv = (int)blockIdx.x < u.conv_blocks;
    // ONE textual barrier for all five waves: before it the four quarter waves issue their loads and sums and the
    // fifth forms Adam's two scalars; behind it wave 0 owns the element's update.
    float* d = nullptr;
    bool upd = false, live = false;
    int64_t idx = -1;
    int item = 0, n_grad = 0;
    AdamPre pre;
    pre.ok = false; pre.p = pre.m = pre.v = 0.0f; pre.step_size = 0.0f; pre.sqrt_bc2 = 1.0f;
    if (q == 4) {
        if (lane == 0) {
            float step_size = 0.0f, sqrt_bc2 = 1.0f;
            if (u.apply_adam) adam_bias_scalars(u.ad, step_size, sqrt_bc2);
            bias_scalars[0] = step_size; bias_scalars[1] = sqrt_bc2;
        }
    } else if (conv) {
        const ReduceArgs& a = u.r;
        // The branch of a block is UNIFORM (blocks_per_branch blocks per branch): the per-branch descriptors (gradient
        // pointers, strides) are then scalar loads from the argument block.  With a per-lane branch (item / n_partial) the
        // compiler fetched them with vector loads from argument memory, one dependent round trip after the other in
        // reduce_dst's chain of cases -- microseconds in front of wave 0's slab loads.
        int br = 0, blk = (int)blockIdx.x;
        while (blk >= u.blocks_per_branch && br + 1 < a.n_branch) { blk -= u.blocks_per_branch; ++br; }
        const int p_raw = blk * 64 + lane;
        live = p_raw < a.n_partial && reduce_live(a, p_raw);
        const int p = live ? p_raw : 0;
        // wave 0 owns the element's update: its parameter / moment loads are issued BEFORE the slab loads, so that Adam
        // starts from registers once the quarter sums are in
        d = (q == 0 && live) ? reduce_dst(a, br, p) : nullptr;
        upd = d != nullptr && u.apply_adam;
        idx = upd ? (int64_t)(d - u.ad.grad) : -1;
        pre = adam_prefetch_state(u.ad, idx);
        quarter[q][lane] = live ? reduce_sum(a, br, p, q, 4) : 0.0f;
    } else {
        item = ((int)blockIdx.x - u.conv_blocks) * 64 + lane;
        live = item < update_head_items(u);
        n_grad = update_head_items(u) - 1;
        d = (q == 0 && live && item < n_grad) ? u.h.grad + item : nullptr;
        upd = d != nullptr && u.apply_adam;
        idx = upd ? (int64_t)(d - u.ad.grad) : -1;
        pre = adam_prefetch_state(u.ad, idx);
        quarter[q][lane] = live ? update_head_sum(u, item, q, 4) : 0.0f;
    }
    __syncthreads();
    if (q == 0) {
        pre.step_size = bias_scalars[0]; pre.sqrt_bc2 = bias_scalars[1];
        const float g = (quarter[0][lane] + quarter[1][lane]) + (quarter[2][lane] + quarter[3][lane]);
        if (d) {
            *d = g;                                   // keep p.grad inspectable
            if (upd) adam_apply(u.ad, idx, g, pre);
        } else if (!conv && live && item == n_grad && u.h.loss) {
            u.h.loss[0] = g;
            if (u.loss2) u.loss2[0] = g;
        }
    }
    // nobody reads step2[0] in this launch (Adam reads step2[1]): safe to commit it here
    if (u.step2 && blockIdx.x == 0 && threadIdx.x == 0) u.step2[0] = u.step2[1];
}
// drgnn_step_gradients: k_update's fixed-order sums without the optimiser -- the gradient of the step's slabs, each slab
// optionally weighted by its graph's d loss / d pred (the autograd boundary, include/drgnn.h), plus one block that clears the
// ranges no kernel writes (GINet's dead attention parameters).  4 waves: interleaved quarters of the slabs, combined in fixed order.
__global__ void __launch_bounds__(256) k_gradients(GradArgs ga) {
    kernarg_touch_lines<DRGNN_KA_LINES(0, (int)sizeof(GradArgs))>(0);
    __shared__ float quarter[4][64];
    const UpdateArgs& u = ga.u;
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int n_sum = u.conv_blocks + (update_head_items(u) - 1 + 63) / 64;
    if ((int)blockIdx.x >= n_sum) {          // the clearing block
        for (int r = 0; r < ga.n_zero; ++r)
            for (int64_t i = threadIdx.x; i < ga.zero_len[r]; i += 256) ga.zero_ptr[r][i] = 0.0f;
        if (u.step2 && threadIdx.x == 0) u.step2[0] = u.step2[1];      // (nobody reads step2 in this launch)
        return;
    }
    const bool conv = (int)blockIdx.x < u.conv_blocks;
    const float* gw = ga.graph_weight;
    float* d = nullptr;
    bool live = false;
    if (conv) {
        const ReduceArgs& a = u.r;
        int br = 0, blk = (int)blockIdx.x;
        while (blk >= u.blocks_per_branch && br + 1 < a.n_branch) { blk -= u.blocks_per_branch; ++br; }
        const int p_raw = blk * 64 + lane;
        live = p_raw < a.n_partial && reduce_live(a, p_raw);
        const int p = live ? p_raw : 0;
        d = (q == 0 && live) ? reduce_dst(a, br, p) : nullptr;
        quarter[q][lane] = !live ? 0.0f : gw ? reduce_sum_w(a, br, p, q, 4, gw, ga.wshift) : reduce_sum(a, br, p, q, 4);
    } else {
        const int item = ((int)blockIdx.x - u.conv_blocks) * 64 + lane;
        live = item < update_head_items(u) - 1;       // (the loss item is not a gradient)
        d = (q == 0 && live) ? u.h.grad + item : nullptr;
        quarter[q][lane] = !live ? 0.0f : gw ? update_head_sum_w(u, item, q, 4, gw) : update_head_sum(u, item, q, 4);
    }
    __syncthreads();
    if (d) *d = (quarter[0][lane] + quarter[1][lane]) + (quarter[2][lane] + quarter[3][lane]);
}
__global__ void __launch_bounds__(DRGNN_P2P_THREADS) k_allreduce_oneshot(P2PArgs a) { p2p_block(a, blockIdx.x); }
__global__ void __launch_bounds__(256) k_adam(AdamArgs a) {
    adam_item(a, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}
// topology prologue when the caller supplies the per-graph offsets: copy them into the
// workspace and clear the status words (one launch instead of two copies and a fill)
__global__ void __launch_bounds__(256) k_topo_begin(const int32_t* node_ptr, const int32_t* edge_ptr,
                                                    int32_t* nptr, int32_t* eptr, int32_t* err, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        if (node_ptr) { nptr[i] = node_ptr[i]; eptr[i] = edge_ptr[i]; }
    }
    if (i < 4) err[i] = 0;
}
#endif  // DRGNN_KERNELS_MAIN
// The instantiations of the aggregation-first step kernels live in drgnn_step_tu.hip (one translation unit per (family, width),
// drgnn_step_af.h) when the library is built from several translation units (Makefile: DRGNN_SPLIT_TU); a single-unit build
// (profiling variants) instantiates them at their lookup functions.  k_step_co_topo / k_step1_co_topo (the product-first family)
// are not instantiated in the device library at all: the host emulation steps through their bodies.
#endif  // !DRGNN_EMU
#include "drgnn_step_af.h"
