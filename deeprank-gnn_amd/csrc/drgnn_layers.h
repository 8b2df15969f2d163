// drgnn_layers.h -- stand-alone (non-fused) pieces behind the reference's function-level API:
// one convolution layer of arbitrary width on ONE graph (custom nets call
// conv(x, edge_index, edge_attr): ginet.py:50, sGAT.py:62, foutnet.py:56), cluster max / mean
// pooling over a Topology's member lists (scatter_max / scatter_mean inside
// community_pooling.py:197,212 and max_pool_x), export of the pooled edges in the reference's
// COO form (pool_edge output), and get_preloaded_cluster (community_pooling.py:25-30).
// Grid-parallel kernels working out of global memory; the fused per-graph kernels of
// drgnn_net.h remain the fast path of the three shipped nets.
#pragma once
#include "drgnn_net.h"

#define DRGNN_LAYER_ROWS 64          // node rows per workgroup in the dense products
#define DRGNN_LAYER_MAXH 128         // widest layer output supported

struct ConvLayerArgs {
    int kind;
    const float* x;            // [N, F]
    int F, H;
    drgnn_conv_params p;
    const int32_t* rowptr; const int32_t* col; const float* w;      // CSR  (w: sGAT)
    const int32_t* colptr; const int32_t* ridx; const int32_t* tslot;   // CSC  (backward)
    int N;
    float* u;                  // [N, HC]   x W (nbr | self)   /  backward: dU
    float* out;                // [N, H]    forward result (pre-activation, like the reference)
    const float* grad_out;     // [N, H]
    float* partials;           // [n_wg][F*HC + H]
    float* grad_x;             // [N, F] or null
};

HD int conv_hc(int kind, int H) { return kind == DRGNN_GINET ? H : 2 * H; }

// ---- forward 1/2: U[rows] = X[rows] W ------------------------------------------------------
DEV void conv_gemm_block(const ConvLayerArgs& a, int blk) {
    const int r0 = blk * DRGNN_LAYER_ROWS;
    const int rows = imin(DRGNN_LAYER_ROWS, a.N - r0);
    const int HC = conv_hc(a.kind, a.H);
    wg_gemm(rows, a.H, a.F, a.x + (long)r0 * a.F, a.F, 1, a.p.w_nbr, (int)a.p.nbr_sk, (int)a.p.nbr_sh,
            a.u + (long)r0 * HC, HC, 1);
    if (a.kind != DRGNN_GINET)
        wg_gemm(rows, a.H, a.F, a.x + (long)r0 * a.F, a.F, 1, a.p.w_self, (int)a.p.self_sk, (int)a.p.self_sh,
                a.u + (long)r0 * HC + a.H, HC, 1);
}

// row statistics of the weighted adjacency (edge scale dv, self scale sc), recomputed on the fly
DEV void conv_row_coef(const ConvLayerArgs& a, int i, float& dv, float& sc) {
    const int lo = a.rowptr[i], hi = a.rowptr[i + 1];
    const int deg = hi - lo;
    if (a.kind == DRGNN_SGAT) {
        float asum = 0.0f;
        for (int k = lo; k < hi; ++k) asum += a.w[k];
        dv = 1.0f / (float)(deg > 0 ? deg : 1);
        sc = asum * dv;
    } else if (a.kind == DRGNN_FOUT) {
        dv = deg > 0 ? 1.0f / (float)deg : 0.0f;
        sc = 1.0f;
    } else {
        dv = 1.0f; sc = 0.0f;
    }
}

// ---- forward 2/2: out[i,h] = sc_i U[i,H+h] + dv_i sum_k w_k U[col_k,h] + b_h ----------------
DEV void conv_aggregate_item(const ConvLayerArgs& a, int64_t item) {
    const int H = a.H, HC = conv_hc(a.kind, H);
    if (item >= (int64_t)a.N * H) return;
    const int i = (int)(item / H), h = (int)(item % H);
    const int lo = a.rowptr[i], hi = a.rowptr[i + 1];
    float acc = 0.0f;
    for (int k = lo; k < hi; ++k) {
        const float cf = (a.kind == DRGNN_SGAT) ? a.w[k] : 1.0f;
        acc = fmaf(cf, a.u[(long)a.col[k] * HC + h], acc);
    }
    if (a.kind != DRGNN_GINET) {
        float dv, sc;
        conv_row_coef(a, i, dv, sc);
        acc = fmaf(sc, a.u[(long)i * HC + H + h], acc * dv) + a.p.bias[h];
        if (a.kind == DRGNN_FOUT && hi == lo) acc = DRGNN_NAN;      // mean of an empty slice
    }
    a.out[item] = acc;
}

// ---- backward 1/4: dU from d out -----------------------------------------------------------
DEV void conv_bwd_du_item(const ConvLayerArgs& a, int64_t item) {
    const int H = a.H, HC = conv_hc(a.kind, H);
    if (item >= (int64_t)a.N * H) return;
    const int j = (int)(item / H), h = (int)(item % H);
    float acc = 0.0f;
    for (int t = a.colptr[j]; t < a.colptr[j + 1]; ++t) {
        const int i = a.ridx[t];
        float cf = 1.0f;
        if (a.kind != DRGNN_GINET) {
            float dv, sc;
            conv_row_coef(a, i, dv, sc);
            cf = (a.kind == DRGNN_SGAT) ? a.w[a.tslot[t]] * dv : dv;
        }
        acc = fmaf(cf, a.grad_out[(long)i * H + h], acc);
    }
    a.u[(long)j * HC + h] = acc;
    if (a.kind != DRGNN_GINET) {
        float dv, sc;
        conv_row_coef(a, j, dv, sc);
        // (FoutLayer, node without out-edges: the forward row is NaN through the empty mean, but x_j Wc still receives
        // the upstream gradient -- autograd of `alpha + gamma`, foutnet.py:75 -- so the self path is NOT masked)
        a.u[(long)j * HC + H + h] = sc * a.grad_out[(long)j * H + h];
    }
}

// ---- backward 2/4: per-workgroup partial dW = X[rows]^T dU[rows], db = sum d out[rows] --------
HD int64_t conv_partial_floats(int kind, int F, int H) { return (int64_t)F * conv_hc(kind, H) + H; }
DEV void conv_bwd_dw_block(const ConvLayerArgs& a, int blk) {
    const int r0 = blk * DRGNN_LAYER_ROWS;
    const int rows = imin(DRGNN_LAYER_ROWS, a.N - r0);
    const int H = a.H, HC = conv_hc(a.kind, H);
    float* part = a.partials + (long)blk * conv_partial_floats(a.kind, a.F, H);
    // part[f*HC + c] = sum_k x[(r0+k)*F + f] * dU[(r0+k)*HC + c]
    wg_gemm(a.F, HC, rows, a.x + (long)r0 * a.F, 1, a.F, a.u + (long)r0 * HC, HC, 1, part, HC, 1);
    FOR_TID(h, H) {
        float acc = 0.0f;
        for (int k = 0; k < rows; ++k) acc += a.grad_out[(long)(r0 + k) * H + h];
        part[(long)a.F * HC + h] = acc;
    }
}

struct ConvReduceArgs {
    const float* partials; int n_wg; int kind, F, H;
    drgnn_conv_params lay; drgnn_conv_grads g;
};
DEV void conv_reduce_item(const ConvReduceArgs& a, int item) {
    const int HC = conv_hc(a.kind, a.H);
    const int P = (int)conv_partial_floats(a.kind, a.F, a.H);
    if (item >= P) return;
    float acc = 0.0f;
    for (int w = 0; w < a.n_wg; ++w) acc += a.partials[(long)w * P + item];
    if (item < a.F * HC) {
        const int f = item / HC, c = item % HC;
        if (c < a.H) { if (a.g.w_nbr) a.g.w_nbr[(long)f * a.lay.nbr_sk + (long)c * a.lay.nbr_sh] = acc; }
        else if (a.g.w_self) a.g.w_self[(long)f * a.lay.self_sk + (long)(c - a.H) * a.lay.self_sh] = acc;
    } else if (a.g.bias) {
        a.g.bias[item - a.F * HC] = acc;
    }
}

// ---- backward 4/4: dX[rows] = dU[rows] Wcat^T ---------------------------------------------------
DEV void conv_bwd_dx_item(const ConvLayerArgs& a, int64_t item) {
    const int H = a.H, HC = conv_hc(a.kind, H), F = a.F;
    if (item >= (int64_t)a.N * F) return;
    const int i = (int)(item / F), f = (int)(item % F);
    const float* du = a.u + (long)i * HC;
    float acc = 0.0f;
    for (int h = 0; h < H; ++h) acc = fmaf(du[h], a.p.w_nbr[(long)f * a.p.nbr_sk + (long)h * a.p.nbr_sh], acc);
    if (a.kind != DRGNN_GINET)
        for (int h = 0; h < H; ++h) acc = fmaf(du[H + h], a.p.w_self[(long)f * a.p.self_sk + (long)h * a.p.self_sh], acc);
    a.grad_x[item] = acc;
}

// ---- cluster pooling over a Topology's depth-0 member lists (any feature width) ---------------
struct SegPoolArgs {
    TopoView tv;
    const float* x;        // [N, H]   rows in node order
    int H, n_graphs;
    int op;                // 0 = max (+arg), 1 = mean
    float* out;            // [C0tot, H]  compact, cluster c of graph g at row CPTR0[g] + c
    int64_t* arg;          // [C0tot, H]  GLOBAL node id of the maximum (N = none), max only
    const float* grad_out; // backward
    float* grad_x;         // [N, H], zero-filled by the caller
    const int64_t* arg_in;
};
DEV void segpool_fwd_block(const SegPoolArgs& a, int g) {
    const TopoView& tv = a.tv;
    const int n0 = tv.p[DRGNN_TI_NPTR][g], rowbase = n0 + g;
    const int C = tv.p[DRGNN_TI_NC0][g];
    const int c0 = tv.p[DRGNN_TI_CPTR0][g];
    const int ntot = tv.p[DRGNN_TI_NPTR][a.n_graphs];
    const int32_t* mp = tv.p[DRGNN_TI_MPTR0] + rowbase;
    const int32_t* mem = tv.p[DRGNN_TI_MEM0] + n0;
    const int H = a.H;
    const FastDiv dH = fastdiv_make(H);
    FOR_TID(item, C * H) {
        const int r = fastdiv(dH, item), h = fastmod(dH, item, r);
        if (a.op == 0) {
            float best = DRGNN_NEG_INF;
            long arg = ntot;
            for (int p = mp[r]; p < mp[r + 1]; ++p) {
                const int m = n0 + mem[p];
                const float v = a.x[(long)m * H + h];
                if (v > best) { best = v; arg = m; }
            }
            if (arg == ntot) best = 0.0f;
            a.out[(long)(c0 + r) * H + h] = best;
            if (a.arg) a.arg[(long)(c0 + r) * H + h] = arg;
        } else {
            float acc = 0.0f;
            const int cnt = mp[r + 1] - mp[r];
            for (int p = mp[r]; p < mp[r + 1]; ++p) acc += a.x[(long)(n0 + mem[p]) * H + h];
            a.out[(long)(c0 + r) * H + h] = acc / (float)(cnt > 0 ? cnt : 1);
        }
    }
}
DEV void segmax_bwd_item(const SegPoolArgs& a, int64_t item, int64_t n_items, int64_t n_nodes) {
    if (item >= n_items) return;
    const int64_t m = a.arg_in[item];
    if (m >= 0 && m < n_nodes) a.grad_x[m * a.H + (item % a.H)] = a.grad_out[item];
}

// ---- pooled edges of a Topology in the reference's COO form ----------------------------------
struct EdgeExportArgs {
    TopoView tv; int n_graphs;
    int64_t* edge_index;   // [2, E1tot]   global consecutive cluster ids, sorted by (row, col)
    float* edge_attr;      // [E1tot] or null
    int64_t e1_total;
};
DEV void edge_export_block(const EdgeExportArgs& a, int g) {
    const TopoView& tv = a.tv;
    const int n0 = tv.p[DRGNN_TI_NPTR][g], e0 = tv.p[DRGNN_TI_EPTR][g], rowbase = n0 + g;
    const int C = tv.p[DRGNN_TI_NC0][g];
    const int c0 = tv.p[DRGNN_TI_CPTR0][g], o0 = tv.p[DRGNN_TI_E1PTR][g];
    const int32_t* rp = tv.p[DRGNN_TI_ROWPTR1] + rowbase;
    const int32_t* col = tv.p[DRGNN_TI_COL1] + e0;
    FOR_TID(r, C) {
        for (int k = rp[r]; k < rp[r + 1]; ++k) {
            a.edge_index[o0 + k] = c0 + r;
            a.edge_index[a.e1_total + o0 + k] = c0 + col[k];
            if (a.edge_attr && tv.w1) a.edge_attr[o0 + k] = tv.w1[e0 + k];
        }
    }
}

// ---- graclus: greedy maximal matching of every graph (README.md:98-126 custom net; torch_cluster) ----
// Visits the nodes of a graph in the order `perm` (identity when null); an unmatched node u takes, among its
// unmatched neighbours v != u, the one with the largest edge weight (the first one in edge-id order when there are
// no weights or on ties) and both get the label min(u, v); a node without a free neighbour keeps its own label.
// Inherently sequential per graph: the graph's CSR is staged in LDS and one lane walks it; graphs run in parallel.
// lds: (N+1) row pointers, E columns, E weights, N labels.
struct GraclusArgs {
    TopoView tv; int n_graphs;
    const float* weight;       // [E] by input edge id, or null
    const int64_t* perm;       // [N] visiting order (LOCAL node ids per graph, graph-major), or null
    int64_t* cluster;          // [N] out: batch-global label n0 + min(u, v)
    int capN, capE;
};
DEV void graclus_block(const GraclusArgs& a, int g, int* lds) {
    const TopoView& tv = a.tv;
    const int n0 = tv.p[DRGNN_TI_NPTR][g], N = tv.p[DRGNN_TI_NPTR][g + 1] - n0;
    const int e0 = tv.p[DRGNN_TI_EPTR][g], E = tv.p[DRGNN_TI_EPTR][g + 1] - e0;
    if (N > a.capN || E > a.capE) return;                  // the host sized the carve from max_nodes / max_edges
    int* rp = lds;
    int* col = rp + (a.capN + 1);
    float* w = (float*)(col + a.capE);
    int* lab = (int*)(w + a.capE);
    const int32_t* g_rp = tv.p[DRGNN_TI_ROWPTR0] + n0 + g;
    const int32_t* g_col = tv.p[DRGNN_TI_COL0] + e0;
    const int32_t* g_eid = tv.p[DRGNN_TI_EID0] + e0;
    FOR_TID(i, N + 1) { rp[i] = g_rp[i]; }
    FOR_TID(k, E) {
        col[k] = g_col[k];
        w[k] = a.weight ? a.weight[e0 + g_eid[k]] : 0.0f;
    }
    FOR_TID(i, N) { lab[i] = -1; }
    BARRIER();
    FOR_TID(t, 1) {
        for (int k = 0; k < N; ++k) {
            long long u = a.perm ? a.perm[n0 + k] : k;
            if (u < 0 || u >= N || lab[u] >= 0) continue;
            int best = -1;
            float bw = 0.0f;
            for (int j = rp[u]; j < rp[u + 1]; ++j) {
                const int v = col[j];
                if (v == (int)u || lab[v] >= 0) continue;
                if (!a.weight) { best = v; break; }
                if (best < 0 || w[j] > bw) { best = v; bw = w[j]; }
            }
            const int m = (best >= 0 && best < (int)u) ? best : (int)u;
            lab[u] = m;
            if (best >= 0) lab[best] = m;
        }
    }
    BARRIER();
    FOR_TID(i, N) { a.cluster[n0 + i] = (long long)n0 + (lab[i] >= 0 ? lab[i] : i); }
}

// ---- get_preloaded_cluster: per-graph running offset, in place ---------------------------------
struct ClusterOffsetArgs {
    int64_t* cluster;          // [n] in/out
    const int32_t* nptr;       // [B+1]
    int n_graphs;
    long long* maxes;          // [B+1] scratch: per-graph max, then exclusive offsets
};
DEV void cluster_max_block(const ClusterOffsetArgs& a, int g, long long* mm) {
    const int n0 = a.nptr[g], n = a.nptr[g + 1] - n0;
    wg_minmax64(a.cluster + n0, n, mm);
    FOR_TID(i, 1) { a.maxes[g] = (n > 0) ? mm[1] : -1; }
}
DEV void cluster_scan_single(const ClusterOffsetArgs& a) {
    // offsets[g] = sum_{q<g} (max_q + 1), exactly the reference's running update (one thread:
    // B is the number of graphs of a mini-batch)
    FOR_TID(i, 1) {
        long long run = 0;
        for (int g = 0; g < a.n_graphs; ++g) {
            const long long m = a.maxes[g];
            a.maxes[g] = run;
            // reference: cluster[batch==g] += max(cluster[batch==g-1]) + 1 where the previous
            // graph has ALREADY been shifted -> shifted max = run + m
            run = run + m + 1;
        }
    }
}
DEV void cluster_add_block(const ClusterOffsetArgs& a, int g) {
    const int n0 = a.nptr[g], n = a.nptr[g + 1] - n0;
    const long long off = a.maxes[g];
    FOR_TID(i, n) { a.cluster[n0 + i] += off; }
}
