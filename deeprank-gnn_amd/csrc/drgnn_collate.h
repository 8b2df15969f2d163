// Mini-batch assembly from a graph set that lives in HBM.
//
// The reference collates on the host, per mini-batch: PyG DataLoader -> Batch.from_data_list over the
// Data objects of HDF5DataSet.load_one_graph (NeuralNet.py:153-154, DataSet.py:231-366): tensors are
// concatenated along dim 0, keys containing "index" along the last dim and shifted by the running node
// count, `batch` = graph slot of every node.  Here the whole set is uploaded ONCE, graph-major with LOCAL
// node ids, and a mini-batch is the list of graph numbers: one workgroup per slot copies its graph's
// slices to the slot's offsets (coalesced, 16-byte rows when F % 4 == 0), shifts the edge ends, writes
// `batch` and the per-slot offset tables the topology builder and the step kernel take -- no host tensor
// work, no host<->device copy besides the id list.
#pragma once
#include "drgnn_rt.h"

struct CollateArgs {
    drgnn_graph_set set;
    const int32_t* ids;        // [B] graph numbers in slot order
    int n_graphs;              // B
    int64_t n_edges;           // E of the batch = row stride of edge_index
    float* x;                  // [N, F]
    int64_t* edge_index;       // [2, E] batch-global node ids
    float* edge_attr;          // [E] or null
    int64_t* batch;            // [N]
    int64_t* cluster0;         // [N] or null
    int64_t* cluster1;         // [sum C0] or null
    void* y;                   // [B] elements of set.y_bytes bytes, or null
    int32_t* node_ptr;         // [B+1]
    int32_t* edge_ptr;         // [B+1]
    int32_t* c1_ptr;           // [B+1] or null
};

// extent of graph `id` in one of the set's offset tables; ids outside the set select nothing
DEV void collate_extent(const int64_t* ptr, int n_set, int id, int64_t* first, int* count) {
    const bool ok = ptr && id >= 0 && id < n_set;
    *first = ok ? ptr[id] : 0;
    *count = ok ? (int)(ptr[id + 1] - ptr[id]) : 0;
}

// sh: 4 ints of workgroup-shared memory
DEV void collate_block(const CollateArgs& a, int g, int* sh) {
    const drgnn_graph_set& s = a.set;
    const int G = (int)s.n_graphs;
    FOR_TID(i, 4) { sh[i] = 0; }
    BARRIER();
    // slot offsets = sums of the extents of the slots in front (B is a mini-batch: tens to thousands)
    FOR_TID(q, g) {
        const int id = a.ids[q];
        int64_t f; int c;
        collate_extent(s.node_ptr, G, id, &f, &c);
        if (c) ATOMIC_ADD(&sh[0], c);
        collate_extent(s.edge_ptr, G, id, &f, &c);
        if (c) ATOMIC_ADD(&sh[1], c);
        collate_extent(s.c1_ptr, G, id, &f, &c);
        if (c) ATOMIC_ADD(&sh[2], c);
    }
    BARRIER();
    const int id = a.ids[g];
    const int n0 = sh[0], e0 = sh[1], c0 = sh[2];
    int64_t sn, se, sc; int n, e, c;
    collate_extent(s.node_ptr, G, id, &sn, &n);
    collate_extent(s.edge_ptr, G, id, &se, &e);
    collate_extent(s.c1_ptr, G, id, &sc, &c);
    const int F = s.n_feat;
    FOR_TID(i, 1) {
        a.node_ptr[g] = n0;
        a.edge_ptr[g] = e0;
        if (a.c1_ptr) a.c1_ptr[g] = c0;
        if (g == a.n_graphs - 1) {
            a.node_ptr[g + 1] = n0 + n;
            a.edge_ptr[g + 1] = e0 + e;
            if (a.c1_ptr) a.c1_ptr[g + 1] = c0 + c;
        }
        if (a.y && s.y && id >= 0 && id < G) {
            if (s.y_bytes == 8) ((int64_t*)a.y)[g] = ((const int64_t*)s.y)[id];
            else ((int32_t*)a.y)[g] = ((const int32_t*)s.y)[id];
        }
    }
    // node rows: one contiguous block of n * F floats
    const float* xs = s.x + sn * F;
    float* xd = a.x + (int64_t)n0 * F;
#ifndef DRGNN_EMU
    if ((F & 3) == 0) {           // rows are multiples of 16 bytes: 128-bit copies
        const drgnn_f4* xs4 = (const drgnn_f4*)xs;
        drgnn_f4* xd4 = (drgnn_f4*)xd;
        FOR_TID(i, n * (F >> 2)) { xd4[i] = xs4[i]; }
    } else
#endif
    {
        FOR_TID(i, n * F) { xd[i] = xs[i]; }
    }
    FOR_TID(i, n) {
        a.batch[n0 + i] = g;
        if (a.cluster0) a.cluster0[n0 + i] = s.cluster0[sn + i];
    }
    // edges: both ends shifted by the slot's node offset ("index" keys of the PyG collate)
    const int64_t* r_in = s.edge_index + se;
    const int64_t* c_in = s.edge_index + s.n_edges + se;
    FOR_TID(k, e) {
        a.edge_index[e0 + k] = r_in[k] + n0;
        a.edge_index[a.n_edges + e0 + k] = c_in[k] + n0;
        if (a.edge_attr) a.edge_attr[e0 + k] = s.edge_attr[se + k];
    }
    if (a.cluster1) {
        FOR_TID(j, c) { a.cluster1[c0 + j] = s.cluster1[sc + j]; }
    }
}

// prefix sums of the slots' extents, one workgroup per mini-batch (cnt: 3 * (batch_size + 1) ints + scan scratch)
struct OffsetsArgs {
    drgnn_graph_set set; const int32_t* ids; int64_t n_ids; int batch_size; int32_t* ptrs;
};
DEV void batch_offsets_block(const OffsetsArgs& a, int k, int* cnt, int* part) {
    const int W = a.batch_size + 1;
    const int64_t first = (int64_t)k * a.batch_size;
    const int B = (int)((a.n_ids - first) < a.batch_size ? (a.n_ids - first) : a.batch_size);
    const int G = (int)a.set.n_graphs;
    FOR_TID(q, 3 * W) { cnt[q] = 0; }
    BARRIER();
    FOR_TID(q, B) {
        const int id = a.ids[first + q];
        int64_t f; int c;
        collate_extent(a.set.node_ptr, G, id, &f, &c); cnt[q] = c;
        collate_extent(a.set.edge_ptr, G, id, &f, &c); cnt[W + q] = c;
        collate_extent(a.set.c1_ptr, G, id, &f, &c); cnt[2 * W + q] = c;
    }
    BARRIER();
    for (int t = 0; t < 3; ++t) wg_exscan(cnt + t * W, W, part);
    int32_t* out = a.ptrs + (int64_t)k * 3 * W;
    FOR_TID(q, 3 * W) { out[q] = cnt[q]; }
}
