// drgnn_mcl.h -- Markov clustering of every graph of a batch (offline preprocessing step of the
// reference: community_detection(..., method='mcl'), community_pooling.py:95-158, which runs
// markov_clustering.run_mcl with its defaults on the unweighted internal-contact graph, once per
// dataset, inside PreCluster, DataSet.py:45-88).  One workgroup per graph, dense N x N fp64
// matrices in a global scratch slab (L2 resident for interface-sized graphs):
//     M <- A + I, columns normalised;  repeat <= 100x: T = M M; T = colnorm(T o T);
//     prune (< 1e-3, but every column keeps its maximum); stop when allclose(T, M); M <- T
//     clusters = nonzero patterns of the attractor rows, sorted; label = index of the last cluster
//     (in that order) containing the node.
// fp64 like the reference (scipy); the result is discrete, tests pin it on the fixture's stored
// clustering (bit-exact labels for all graphs and both depths).
#pragma once
#include "drgnn_rt.h"
#include "../../include/drgnn.h"

struct MclArgs {
    const int64_t* edge_index;   // [2, Etot] global node ids (both directions or one, any duplication)
    int64_t n_edges;
    const int32_t* node_ptr;     // [B+1]
    const int32_t* edge_ptr;     // [B+1]
    const int64_t* mat_ptr;      // [B+1] prefix sums of N_g^2
    int n_graphs;
    double* mat;                 // [3 * mat_ptr[B]]   M, T (and column scratch)
    int32_t* iscr;               // [4 * Ntot]
    int64_t* labels;             // [Ntot] out
    int32_t* info;               // [B] out: iterations used (negative: not converged in 100)
    int iterations;
    double prune_threshold;
};

// lexicographic order of the sorted member lists of rows a and b of the 0/1 pattern of M
DEV int mcl_tuple_cmp(const double* M, int N, int a, int b) {
    const double* ra = M + (long)a * N;
    const double* rb = M + (long)b * N;
    for (int c = 0; c < N; ++c) {
        const bool ia = ra[c] != 0.0, ib = rb[c] != 0.0;
        if (ia != ib) {
            // the row that contains c has the smaller next element -- unless the other row has no
            // element left at all (it is then a proper prefix, hence smaller)
            const double* other = ia ? rb : ra;
            bool more = false;
            for (int d = c + 1; d < N; ++d) more = more || (other[d] != 0.0);
            if (ia) return more ? -1 : 1;
            return more ? 1 : -1;
        }
    }
    return 0;
}

DEV void mcl_graph(const MclArgs& a, int g, int* lds_i, double* lds_d) {
    const int n0 = a.node_ptr[g], N = a.node_ptr[g + 1] - n0;
    const int e0 = a.edge_ptr[g], E = a.edge_ptr[g + 1] - e0;
    const long NN = (long)N * N;
    const int64_t total = a.mat_ptr[a.n_graphs];
    double* M = a.mat + a.mat_ptr[g];
    double* T = a.mat + total + a.mat_ptr[g];
    double* cs = a.mat + 2 * total + a.mat_ptr[g];        // column sums (first N entries used)
    int* am = a.iscr + 4L * n0;                            // column arg-max / attractor flag
    int* isrep = am + N;
    int* krank = isrep + N;
    double* red = lds_d;                                   // [NTHREADS]
    int* flag = lds_i;                                     // [2]
    const FastDiv dN = fastdiv_make(N);
    const int64_t* src_row = a.edge_index + e0;
    const int64_t* src_col = a.edge_index + a.n_edges + e0;

    FOR_TID(e, NN) { M[e] = 0.0; }
    BARRIER();
    FOR_TID(e, E) {
        const long i = (long)src_row[e] - n0, j = (long)src_col[e] - n0;
        if (i >= 0 && i < N && j >= 0 && j < N) { M[i * N + j] = 1.0; M[j * N + i] = 1.0; }
    }
    BARRIER();
    FOR_TID(i, N) { M[(long)i * N + i] = 1.0; }            // loop_value = 1
    BARRIER();
    FOR_TID(j, N) {
        double s = 0.0;
        for (int i = 0; i < N; ++i) s += fabs(M[(long)i * N + j]);
        cs[j] = (s == 0.0) ? 1.0 : s;
    }
    BARRIER();
    FOR_TID(e, NN) { const int i = fastdiv(dN, (int)e); M[e] = M[e] / cs[fastmod(dN, (int)e, i)]; }
    BARRIER();

    int used = 0;
    bool done = false;
    for (int it = 1; it <= a.iterations && !done; ++it) {
        used = it;
        // expansion (M M) and inflation (element-wise square), T = (M M) o (M M)
        FOR_TID(e, NN) {
            const int i = fastdiv(dN, (int)e), j = fastmod(dN, (int)e, i);
            const double* mi = M + (long)i * N;
            double acc = 0.0;
            for (int k = 0; k < N; ++k) acc += mi[k] * M[(long)k * N + j];
            T[e] = acc * acc;
        }
        BARRIER();
        FOR_TID(j, N) {
            double s = 0.0;
            for (int i = 0; i < N; ++i) s += fabs(T[(long)i * N + j]);
            cs[j] = (s == 0.0) ? 1.0 : s;
        }
        BARRIER();
        FOR_TID(e, NN) { const int i = fastdiv(dN, (int)e); T[e] = T[e] / cs[fastmod(dN, (int)e, i)]; }
        BARRIER();
        FOR_TID(j, N) {                                     // first maximum of the column
            double best = T[j];
            int arg = 0;
            for (int i = 1; i < N; ++i) { const double v = T[(long)i * N + j]; if (v > best) { best = v; arg = i; } }
            am[j] = arg;
        }
        BARRIER();
        FOR_TID(t, DRGNN_NTHREADS) {
            double worst = -1.0;
            for (long e = t; e < NN; e += DRGNN_NTHREADS) {
                const int i = fastdiv(dN, (int)e), j = fastmod(dN, (int)e, i);
                double v = T[e];
                if (!(v >= a.prune_threshold) && am[j] != i) v = 0.0;
                T[e] = v;
                const double last = M[e];
                const double c = fabs(v - last) - 1e-5 * fabs(last);
                worst = c > worst ? c : worst;
            }
            red[t] = worst;
        }
        BARRIER();
        FOR_TID(i, 1) {
            double worst = -1.0;
            for (int t = 0; t < DRGNN_NTHREADS; ++t) worst = red[t] > worst ? red[t] : worst;
            flag[0] = (worst <= 1e-8) ? 1 : 0;
        }
        BARRIER();
        done = flag[0] != 0;
        double* sw = M; M = T; T = sw;
        BARRIER();
    }
    FOR_TID(i, 1) { a.info[g] = done ? used : -used; }

    // ---- clusters -> labels ---------------------------------------------------------------
    FOR_TID(i, N) { am[i] = (M[(long)i * N + i] != 0.0) ? 1 : 0; }      // attractor rows
    BARRIER();
    FOR_TID(i, N) {                                                       // first row of each distinct tuple
        int rep = am[i];
        for (int q = 0; q < i && rep; ++q)
            if (am[q] && mcl_tuple_cmp(M, N, q, i) == 0) rep = 0;
        isrep[i] = rep;
    }
    BARRIER();
    FOR_TID(i, N) {                                                       // rank among the distinct tuples
        int k = 0;
        if (am[i])
            for (int q = 0; q < N; ++q)
                if (isrep[q] && mcl_tuple_cmp(M, N, q, i) < 0) ++k;
        krank[i] = k;
    }
    BARRIER();
    FOR_TID(v, N) {                                                       // later clusters overwrite
        int lab = 0;
        for (int i = 0; i < N; ++i)
            if (am[i] && M[(long)i * N + v] != 0.0 && krank[i] > lab) lab = krank[i];
        a.labels[n0 + v] = lab;
    }
}
