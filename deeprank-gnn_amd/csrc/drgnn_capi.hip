// drgnn_capi.hip -- kernels + the extern "C" surface declared in include/drgnn.h.
//
// Built twice from this one source:
//   hipcc --offload-arch=gfx950 -> csrc/libdrgnn.so        (the product)
//   g++ -x c++ -DDRGNN_EMU      -> tests/emu/build/libdrgnn_emu.so (CPU test-suite only:
//        every "launch" becomes a loop over workgroups on host pointers)
#define DRGNN_KERNELS_MAIN
#include "drgnn_kernels.h"

#include <vector>
#include <string.h>
#include <stdlib.h>
#ifdef DRGNN_EMU
#define DRGNN_LDS_LIMIT (160 * 1024)
#else
#define DRGNN_LDS_LIMIT (160 * 1024)
#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t e_ = (expr);                         \
        if (e_ != hipSuccess) return (int)e_;           \
    } while (0)
#endif



// =====================================================================================
// host side
// =====================================================================================
extern "C" {

int drgnn_abi_version(void) { return DRGNN_ABI_VERSION; }

#if defined(DRGNN_PHASE_TIMING) && !defined(DRGNN_EMU)
// profiling build only: where phase_mark() writes (device buffer of >= 4002 uint64)
int drgnn_debug_set_phase_buffer(void* dev_buf) {
    unsigned long long* p = (unsigned long long*)dev_buf;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_buf), &p, sizeof(p));
}
#endif

int drgnn_topology_layout(int64_t n_nodes, int64_t n_edges, int64_t n_graphs, int64_t* off_i32,
                          int64_t* off_f32) {
    if (n_nodes < 0 || n_edges < 0 || n_graphs < 0 || !off_i32 || !off_f32) return DRGNN_E_ARG;
    TopoLayout L;
    topo_layout(n_nodes, n_edges, n_graphs, &L);
    for (int k = 0; k <= DRGNN_TI_COUNT; ++k) off_i32[k] = L.i32[k];
    for (int k = 0; k <= DRGNN_TF_COUNT; ++k) off_f32[k] = L.f32[k];
    return 0;
}

int64_t drgnn_topology_scratch_elems(int64_t n_nodes, int64_t n_edges, int64_t n_graphs) {
    return topo_gscratch_base(n_nodes, n_edges, n_graphs) + TOPO_GSCRATCH_CONST;
}

// tile_f > 0: + the x tile of the aggregation tiles (DRGNN_TOPO_TILES)
static int64_t topo_lds_bytes(int capN, int capE, int tile_f = 0) {
    const int64_t capT = (capN > capE ? capN : capE) + 1;
    // (the x tile's rows are padded to pad4(F) + 4 floats)
    return 4 * (topo_scratch_ints(capN, capE, capT, (int64_t)capN + capE + 2) + (tile_f > 0 ? (int64_t)capN * (((tile_f + 3) & ~3) + 4) : 0));
}
static bool topo_tiles_shape_ok(int capN, int capE, int F) {
    const int TF = (F + 3) & ~3;
    if (capN <= 0 || F <= 0 || F > 256 || (int64_t)capN * TF > 16 * DRGNN_BCAP) return false;      // (x tile: 4 float4 per lane)
    return topo_lds_bytes(capN, capE > 0 ? capE : 1, F) <= DRGNN_LDS_LIMIT;
}

int64_t drgnn_topology_lds_bytes(int32_t max_nodes, int32_t max_edges) {
    if (max_nodes <= 0) return 0;
    return topo_lds_bytes(max_nodes, max_edges > 0 ? max_edges : 1);
}

int64_t drgnn_topology_tiles_elems(int64_t n_nodes, int32_t n_feat) {
    if (n_nodes < 0 || n_feat <= 0) return 0;
    const int64_t TF = ((int64_t)n_feat + 3) & ~(int64_t)3;      // S [n][TF] | D [n] | C [n] | (F % 4 != 0) X [n][TF]
    return (n_feat & 3) ? drgnn_tiles_x_off(n_nodes, TF) + n_nodes * TF : n_nodes * (TF + 2);
}
int32_t drgnn_topology_tiles_ok(int32_t max_nodes, int32_t max_edges, int32_t n_feat) {
    return topo_tiles_shape_ok(max_nodes, max_edges, n_feat) ? 1 : 0;
}

}  // extern "C"

#ifndef DRGNN_TOPO_SPLIT_MAX_GRAPHS
#define DRGNN_TOPO_SPLIT_MAX_GRAPHS 160
#endif
// fills a TopoLaunch from the public arguments; *lds_out = LDS bytes (0: global scratch path)
static int topo_prepare(TopoLaunch& L, int64_t* lds_out, const int64_t* edge_index, const float* edge_attr,
                        const int64_t* batch, const int64_t* cluster0, const int64_t* cluster1,
                        const int32_t* node_ptr, const int32_t* edge_ptr, const int32_t* c1_ptr,
                        int64_t n_nodes, int64_t n_edges, int64_t len_cluster1, int64_t n_graphs,
                        int32_t max_nodes, int32_t max_edges, int32_t* ws_i32, float* ws_f32,
                        int32_t* scratch_i32, int32_t flags = DRGNN_TOPO_HIER, const float* x_in = nullptr,
                        float* tiles = nullptr, int32_t tile_f = 0) {
    if (n_nodes < 0 || n_edges < 0 || n_graphs < 0 || !ws_i32) return DRGNN_E_ARG;
    if (!batch && !(node_ptr && edge_ptr)) return DRGNN_E_ARG;      // the offsets are derived from `batch`
    if (!cluster0 && cluster1) return DRGNN_E_ARG;
    if (n_edges > 0 && !edge_index) return DRGNN_E_ARG;
    if (edge_attr && !ws_f32) return DRGNN_E_ARG;
    if (n_nodes + n_graphs >= INT32_MAX || n_edges >= INT32_MAX) return DRGNN_E_CAPACITY;
    TopoLayout lay;
    topo_layout(n_nodes, n_edges, n_graphs, &lay);
    L.tv = topo_view(ws_i32, ws_f32, lay);
    L.args.edge_index = edge_index;
    L.args.edge_attr = edge_attr;
    L.args.cluster0 = cluster0;
    L.args.cluster1 = cluster1;
    L.args.c1_ptr = c1_ptr;
    L.args.n_edges = n_edges;
    L.args.len_cluster1 = len_cluster1;
    L.args.n_graphs = (int)n_graphs;
    L.args.set_ids = nullptr; L.args.set_node_ptr = L.args.set_edge_ptr = L.args.set_c1_ptr = nullptr;
    L.args.set_x = nullptr; L.args.set_y = nullptr; L.args.x_out = nullptr; L.args.y_out = nullptr;
    L.args.n_set = 0; L.args.n_feat = 0; L.args.y_bytes = 0;
    L.args.flags = flags;
    L.args.x_in = nullptr; L.args.tiles = nullptr; L.args.tile_nodes = n_nodes; L.args.tile_f = 0;
    if (flags & DRGNN_TOPO_TILES) {
        // the aggregation tiles need the hierarchical order's companions, an output buffer and float4-loadable features
        // (x_in null: resident-set mode, the caller fills in the set's x and checks it)
        if (!(flags & DRGNN_TOPO_HIER) || !tiles || (x_in && !(tile_f & 3) && (((uintptr_t)x_in) & 15)) || (((uintptr_t)tiles) & 15)) return DRGNN_E_ARG;
        if (!topo_tiles_shape_ok(max_nodes, max_edges, tile_f)) return DRGNN_E_CAPACITY;
        L.args.x_in = x_in; L.args.tiles = tiles; L.args.tile_f = tile_f;
    }
    L.gscratch = scratch_i32;
    L.level1_only = 0;
    L.roles = 1;
    L.pf_ids = nullptr; L.pf_n = 0; L.pf_graphs = 0;
    L.user_nptr = (node_ptr && edge_ptr) ? node_ptr : nullptr;
    L.user_eptr = (node_ptr && edge_ptr) ? edge_ptr : nullptr;
    int64_t lds = 0;
    L.capN = 0; L.capE = 0;
    if (max_nodes > 0) {
        lds = topo_lds_bytes(max_nodes, max_edges > 0 ? max_edges : 1, L.args.tile_f);
        if (lds <= DRGNN_LDS_LIMIT) { L.capN = max_nodes; L.capE = max_edges > 0 ? max_edges : 1; }
        else lds = 0;
    }
    if (L.args.tile_f > 0 && L.capN == 0) return DRGNN_E_CAPACITY;      // (tiles are formed by the LDS builder only)
    // two independent workgroups per graph (edge structures / member lists) when nothing forces the
    // single-chain order: LDS path, clusters present, depth-1 ids located by the caller
    // ... and when the extra workgroups find idle CUs: two workgroups per graph shorten the builder's critical path
    // (what counts while a mini-batch leaves CUs idle) but repeat the cluster ranking; beyond ~one workgroup per CU the
    // total work decides and one workgroup per graph does less of it
    if (L.capN > 0 && cluster0 != nullptr && L.user_nptr != nullptr && !(cluster1 != nullptr && c1_ptr == nullptr) &&
        n_graphs <= DRGNN_TOPO_SPLIT_MAX_GRAPHS)
        L.roles = 2;
    *lds_out = lds;
    return 0;
}

// a request in either mode -> launch description
static int topo_prepare_req(TopoLaunch& T, int64_t* tlds, const drgnn_topology_request* r) {
    const drgnn_graph_set* gs = r->set;
    if (!gs)
        return topo_prepare(T, tlds, r->edge_index, r->edge_attr, r->batch, r->cluster0, r->cluster1, r->node_ptr,
                            r->edge_ptr, r->c1_ptr, r->n_nodes, r->n_edges, r->len_cluster1, r->n_graphs, r->max_nodes,
                            r->max_edges, r->ws_i32, r->ws_f32, r->scratch_i32, r->flags, r->x, r->tiles, r->n_feat);
    if (!r->ids || !r->node_ptr || !r->edge_ptr || !gs->node_ptr || !gs->edge_ptr || !gs->cluster0) return DRGNN_E_ARG;
    if (gs->cluster1 && (!gs->c1_ptr || !r->c1_ptr)) return DRGNN_E_ARG;
    if (r->x_out && (!gs->x || gs->n_feat <= 0)) return DRGNN_E_ARG;
    if (r->y_out && (!gs->y || (gs->y_bytes != 4 && gs->y_bytes != 8))) return DRGNN_E_ARG;
    const float* attr = (r->ws_f32 && gs->edge_attr) ? gs->edge_attr : nullptr;
    if ((r->flags & DRGNN_TOPO_TILES) && (!gs->x || gs->n_feat <= 0 || (!(gs->n_feat & 3) && (((uintptr_t)gs->x) & 15)))) return DRGNN_E_ARG;
    const int rc = topo_prepare(T, tlds, gs->edge_index, attr, nullptr, gs->cluster0, gs->cluster1, r->node_ptr,
                                r->edge_ptr, r->c1_ptr, r->n_nodes, r->n_edges, r->len_cluster1, r->n_graphs,
                                r->max_nodes, r->max_edges, r->ws_i32, r->ws_f32, r->scratch_i32, r->flags, nullptr, r->tiles,
                                gs->n_feat);
    if (rc) return rc;
    TopoArgs& a = T.args;
    a.n_edges = gs->n_edges;                  // row stride of the SET's edge_index
    a.set_ids = r->ids; a.set_node_ptr = gs->node_ptr; a.set_edge_ptr = gs->edge_ptr; a.set_c1_ptr = gs->c1_ptr;
    a.set_x = gs->x; a.set_y = gs->y; a.x_out = r->x_out; a.y_out = r->y_out;
    a.n_set = gs->n_graphs; a.n_feat = gs->n_feat; a.y_bytes = gs->y_bytes;
    return 0;
}

static int topology_build_impl(const int64_t* edge_index, const float* edge_attr, const int64_t* batch,
                         const int64_t* cluster0, const int64_t* cluster1, const int32_t* node_ptr,
                         const int32_t* edge_ptr, const int32_t* c1_ptr, int64_t n_nodes,
                         int64_t n_edges, int64_t len_cluster1, int64_t n_graphs, int32_t max_nodes,
                         int32_t max_edges, int32_t* ws_i32, float* ws_f32, int32_t* scratch_i32,
                         int32_t flags, void* stream_, const float* x_in = nullptr, float* tiles = nullptr,
                         int32_t tile_f = 0) {
    drgnn_stream_t stream = (drgnn_stream_t)stream_;
    TopoLaunch L;
    int64_t lds = 0;
    int rc0 = topo_prepare(L, &lds, edge_index, edge_attr, batch, cluster0, cluster1, node_ptr, edge_ptr, c1_ptr,
                           n_nodes, n_edges, len_cluster1, n_graphs, max_nodes, max_edges, ws_i32, ws_f32,
                           scratch_i32, flags, x_in, tiles, tile_f);
    if (rc0) return rc0;
    if (L.capN == 0 && !scratch_i32) return DRGNN_E_CAPACITY;
    if (n_graphs == 0) return 0;
    PtrArgs pa;
    pa.batch = batch; pa.edge_row = edge_index; pa.n_nodes = n_nodes; pa.n_edges = n_edges;
    pa.n_graphs = (int)n_graphs; pa.nptr = L.tv.p[DRGNN_TI_NPTR]; pa.eptr = L.tv.p[DRGNN_TI_EPTR];
    pa.err = L.tv.p[DRGNN_TI_ERR];
    const bool need_ptrs = !(node_ptr && edge_ptr);
    const int64_t span = n_nodes > n_edges ? n_nodes : n_edges;
#ifdef DRGNN_EMU
    if (need_ptrs) {
        for (int k = 0; k < 4; ++k) pa.err[k] = 0;
        for (int64_t i = 0; i < (span > 0 ? span : 1); ++i) ptr_item(pa, i);
    }
    std::vector<int> lds_buf((size_t)(lds / 4) + 16);
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) { if (!(cluster1 && !c1_ptr)) break; L.level1_only = 1; L.roles = 1; }
        for (int gph = 0; gph < n_graphs * (pass == 0 ? L.roles : 1); ++gph) {
            if (L.capN > 0) topo_block<true>(L, gph, lds_buf.data());
            else topo_block<false>(L, gph, lds_buf.data());
        }
    }
    (void)stream;
#else
    if (need_ptrs) {
        const int n = (int)n_graphs + 1;
        hipLaunchKernelGGL(k_topo_begin, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                           (const int32_t*)nullptr, (const int32_t*)nullptr, pa.nptr, pa.eptr, pa.err, n);
        const int64_t items = span > 0 ? span : 1;
        hipLaunchKernelGGL(k_ptrs, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream, pa);
    }
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void*)k_topo<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) { if (!(cluster1 && !c1_ptr)) break; L.level1_only = 1; L.roles = 1; }
        if (L.capN > 0) hipLaunchKernelGGL(k_topo<true>, dim3((unsigned)(n_graphs * L.roles)), dim3(DRGNN_NTHREADS), (size_t)lds, stream, L);
        else hipLaunchKernelGGL(k_topo<false>, dim3((unsigned)n_graphs), dim3(DRGNN_NTHREADS), 0, stream, L);
    }
    HIP_TRY(hipGetLastError());
#endif
    return 0;
}

extern "C" {

int drgnn_topology_build(const int64_t* edge_index, const float* edge_attr, const int64_t* batch,
                         const int64_t* cluster0, const int64_t* cluster1, const int32_t* node_ptr,
                         const int32_t* edge_ptr, const int32_t* c1_ptr, int64_t n_nodes,
                         int64_t n_edges, int64_t len_cluster1, int64_t n_graphs, int32_t max_nodes,
                         int32_t max_edges, int32_t* ws_i32, float* ws_f32, int32_t* scratch_i32,
                         void* stream_) {
    return topology_build_impl(edge_index, edge_attr, batch, cluster0, cluster1, node_ptr, edge_ptr, c1_ptr, n_nodes, n_edges,
                               len_cluster1, n_graphs, max_nodes, max_edges, ws_i32, ws_f32, scratch_i32, DRGNN_TOPO_HIER,
                               stream_);
}

int drgnn_topology_build_request(const drgnn_topology_request* r, void* stream_) {
    if (!r) return DRGNN_E_ARG;
    if (!r->set)
        return topology_build_impl(r->edge_index, r->edge_attr, r->batch, r->cluster0, r->cluster1, r->node_ptr,
                                   r->edge_ptr, r->c1_ptr, r->n_nodes, r->n_edges, r->len_cluster1, r->n_graphs,
                                   r->max_nodes, r->max_edges, r->ws_i32, r->ws_f32, r->scratch_i32, r->flags, stream_,
                                   r->x, r->tiles, r->n_feat);
    TopoLaunch L;
    int64_t lds = 0;
    const int rc = topo_prepare_req(L, &lds, r);
    if (rc) return rc;
    if (L.capN == 0 && !r->scratch_i32) return DRGNN_E_CAPACITY;
    if (r->n_graphs == 0) return 0;
#ifdef DRGNN_EMU
    std::vector<int> lds_buf((size_t)(lds / 4) + 16);
    for (int gph = 0; gph < r->n_graphs * L.roles; ++gph) {
        if (L.capN > 0) topo_block<true>(L, gph, lds_buf.data());
        else topo_block<false>(L, gph, lds_buf.data());
    }
    (void)stream_;
#else
    hipStream_t stream = (hipStream_t)stream_;
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void*)k_topo<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (L.capN > 0) hipLaunchKernelGGL(k_topo<true>, dim3((unsigned)(r->n_graphs * L.roles)), dim3(DRGNN_NTHREADS), (size_t)lds, stream, L);
    else hipLaunchKernelGGL(k_topo<false>, dim3((unsigned)r->n_graphs), dim3(DRGNN_NTHREADS), 0, stream, L);
    HIP_TRY(hipGetLastError());
#endif
    return 0;
}

int drgnn_topology_tiles(const int32_t* ws_i32, const float* ws_f32, int64_t n_nodes, int64_t n_edges, int64_t n_graphs,
                         const float* x, int32_t n_feat, int32_t use_weights, float* tiles, void* stream_) {
    if (!ws_i32 || !x || !tiles || n_nodes < 0 || n_edges < 0 || n_graphs < 0 || n_feat <= 0) return DRGNN_E_ARG;
    if (use_weights && !ws_f32) return DRGNN_E_ARG;
    if (n_graphs == 0) return 0;
    TopoLayout lay;
    topo_layout(n_nodes, n_edges, n_graphs, &lay);
    TilesArgs a;
    a.tv = topo_view(const_cast<int32_t*>(ws_i32), const_cast<float*>(ws_f32), lay);
    a.x = x; a.tiles = tiles; a.n_nodes = n_nodes; a.n_feat = n_feat; a.use_weights = use_weights ? 1 : 0;
#ifdef DRGNN_EMU
    for (int64_t g = 0; g < n_graphs; ++g) tiles_block(a, (int)g);
    (void)stream_;
#else
    hipLaunchKernelGGL(k_tiles, dim3((unsigned)n_graphs), dim3(DRGNN_NTHREADS), 0, (hipStream_t)stream_, a);
    HIP_TRY(hipGetLastError());
#endif
    return 0;
}

int drgnn_batch_offsets(const drgnn_graph_set* set, const int32_t* ids, int64_t n_ids, int32_t batch_size,
                        int32_t* ptrs, void* stream_) {
    if (!set || !ids || !ptrs || n_ids < 0 || batch_size < 1 || !set->node_ptr || !set->edge_ptr) return DRGNN_E_ARG;
    if (batch_size > 4096) return DRGNN_E_CAPACITY;
    if (n_ids == 0) return 0;
    OffsetsArgs a;
    a.set = *set; a.ids = ids; a.n_ids = n_ids; a.batch_size = batch_size; a.ptrs = ptrs;
    const int64_t nb = (n_ids + batch_size - 1) / batch_size;
    const size_t words = (size_t)DRGNN_NTHREADS + 4 + 3 * ((size_t)batch_size + 1);
#ifdef DRGNN_EMU
    std::vector<int> buf(words);
    for (int64_t k = 0; k < nb; ++k) batch_offsets_block(a, (int)k, buf.data() + DRGNN_NTHREADS + 4, buf.data());
    (void)stream_;
#else
    hipLaunchKernelGGL(k_batch_offsets, dim3((unsigned)nb), dim3(DRGNN_NTHREADS), words * 4, (hipStream_t)stream_, a);
    HIP_TRY(hipGetLastError());
#endif
    return 0;
}

int drgnn_topology_finalize(int32_t* ws_i32, int64_t n_nodes, int64_t n_edges, int64_t n_graphs,
                            void* stream_) {
    if (!ws_i32 || n_graphs < 0) return DRGNN_E_ARG;
    TopoLayout lay;
    topo_layout(n_nodes, n_edges, n_graphs, &lay);
    ScanArgs a;
    a.tv = topo_view(ws_i32, nullptr, lay);
    a.n_graphs = (int)n_graphs;
#ifdef DRGNN_EMU
    int part[DRGNN_NTHREADS + 4];
    finalize_block(a, part);
    (void)stream_;
#else
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(DRGNN_NTHREADS), 0, (hipStream_t)stream_, a);
    HIP_TRY(hipGetLastError());
#endif
    return 0;
}

int drgnn_topology_status(const int32_t* ws_i32, int64_t n_nodes, int64_t n_edges, int64_t n_graphs,
                          int32_t* status4, void* stream_) {
    if (!ws_i32 || !status4) return DRGNN_E_ARG;
    TopoLayout lay;
    topo_layout(n_nodes, n_edges, n_graphs, &lay);
    const int32_t* err = ws_i32 + lay.i32[DRGNN_TI_ERR];
    const int32_t* gst = ws_i32 + lay.i32[DRGNN_TI_GSTAT];
    std::vector<int32_t> host((size_t)(2 * n_graphs) + 4);
#ifdef DRGNN_EMU
    for (int k = 0; k < 4; ++k) host[k] = err[k];
    for (int64_t g = 0; g < 2 * n_graphs; ++g) host[4 + g] = gst[g];
    (void)stream_;
#else
    HIP_TRY(hipMemcpyAsync(host.data(), err, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream_));
    if (n_graphs > 0)
        HIP_TRY(hipMemcpyAsync(host.data() + 4, gst, 2 * n_graphs * sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream_));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream_));
#endif
    status4[0] = host[0]; status4[1] = -1; status4[2] = 0; status4[3] = 0;
    for (int64_t g = 0; g < n_graphs; ++g) {
        host[4 + g] |= host[4 + n_graphs + g];       // edge-structure and member-list workgroups
        if (host[4 + g]) {
            status4[0] |= host[4 + g];
            if (status4[1] < 0) status4[1] = (int32_t)g;
            status4[2] += 1;
        }
    }
    return 0;
}

// ---- fused net ----------------------------------------------------------------------
int64_t drgnn_net_lds_bytes(int32_t kind, int32_t n_feat, int32_t max_nodes, int32_t max_edges,
                            int32_t max_c0, int32_t backward) {
    if (max_nodes <= 0) return 0;
    const int capC = (max_c0 > 0 && max_c0 < max_nodes) ? max_c0 : max_nodes;
    return 4 * net_scratch_words(kind, n_feat, max_nodes, max_edges > 0 ? max_edges : 1, capC, backward);
}

int64_t drgnn_net_partial_elems(int32_t kind, int32_t n_feat) { (void)kind; return net_partial_floats(n_feat); }

int64_t drgnn_net_scratch_elems(int32_t kind, int32_t n_feat, int64_t n_nodes, int64_t n_edges,
                                int64_t n_graphs) {
    const int nb = (kind == DRGNN_GINET) ? 2 : 1;
    const int64_t f = net_gscratch_base(kind, n_feat, n_nodes, n_edges, n_graphs, 0);
    const int64_t b = net_gscratch_base(kind, n_feat, n_nodes, n_edges, n_graphs, 1);
    return nb * (f > b ? f : b) + 64;
}

}  // extern "C"

static int net_check(const drgnn_net_desc* net) {
    if (!net) return DRGNN_E_ARG;
    if (net->kind < DRGNN_GINET || net->kind > DRGNN_FOUT) return DRGNN_E_ARG;
    if (net->n_branch < 1 || net->n_branch > DRGNN_MAX_BRANCH) return DRGNN_E_ARG;
    if (net->n_feat < 1) return DRGNN_E_WIDTH;
    for (int b = 0; b < net->n_branch; ++b) {
        if (!net->conv1[b].w_nbr || !net->conv2[b].w_nbr) return DRGNN_E_ARG;
        if (net->kind != DRGNN_GINET &&
            (!net->conv1[b].w_self || !net->conv2[b].w_self || !net->conv1[b].bias || !net->conv2[b].bias))
            return DRGNN_E_ARG;
    }
    return 0;
}

template <bool BWD>
static int net_launch(NetLaunch& L, int32_t max_nodes, int32_t max_edges, int32_t max_c0, float* scratch,
                      void* stream_, const TopoLaunch* co = nullptr, int64_t co_lds = 0) {
    const int kind = L.a.net.kind;
    const int64_t lds = drgnn_net_lds_bytes(kind, L.a.net.n_feat, max_nodes, max_edges, max_c0, BWD ? 1 : 0);
    L.capN = 0; L.capE = 0; L.capC = 0; L.gscratch = scratch;
    int64_t use_lds = 0;
    L.a.hf.stage = 0;
    if (lds > 0 && lds <= DRGNN_LDS_LIMIT) {
        L.capN = max_nodes;
        L.capE = max_edges > 0 ? max_edges : 1;
        L.capC = (max_c0 > 0 && max_c0 < max_nodes) ? max_c0 : max_nodes;
        use_lds = lds;
        if (BWD && L.a.hf.enabled) {    // room to keep the head's weights in LDS as well?
            const int64_t extra = 4 * head_stage_words(L.a.hf.R, L.a.hf.H, L.a.hf.O);
            if (use_lds + extra <= DRGNN_LDS_LIMIT) { use_lds += extra; L.a.hf.stage = 1; }
        }
    } else if (!scratch) {
        return DRGNN_E_CAPACITY;
    }
    const int blocks = L.a.n_graphs * L.a.net.n_branch;
    // co-launched topology build of the next mini-batch: only when both jobs run out of LDS and the
    // builder needs no helper passes (offsets supplied, depth-1 ids located)
    const bool co_ok = co != nullptr && use_lds > 0 && co->capN > 0 && co->user_nptr != nullptr &&
                       !(co->args.cluster1 != nullptr && co->args.c1_ptr == nullptr) && co->args.n_graphs > 0 &&
                       blocks > 0;
    if (blocks == 0 && !co) return 0;
#ifdef DRGNN_EMU
    std::vector<float> buf((size_t)((use_lds > co_lds ? use_lds : co_lds) / 4) + 16);
    for (int b = 0; b < blocks; ++b) {
        if (use_lds) {
            if (kind == DRGNN_GINET) net_block<DRGNN_GINET, BWD, true>(L, b, buf.data());
            else if (kind == DRGNN_SGAT) net_block<DRGNN_SGAT, BWD, true>(L, b, buf.data());
            else net_block<DRGNN_FOUT, BWD, true>(L, b, buf.data());
        } else {
            if (kind == DRGNN_GINET) net_block<DRGNN_GINET, BWD, false>(L, b, buf.data());
            else if (kind == DRGNN_SGAT) net_block<DRGNN_SGAT, BWD, false>(L, b, buf.data());
            else net_block<DRGNN_FOUT, BWD, false>(L, b, buf.data());
        }
    }
    if (co_ok)
        for (int g = 0; g < co->args.n_graphs * co->roles; ++g) topo_block<true>(*co, g, (int*)buf.data());
    (void)stream_;
    return co_ok ? 1 : 0;       // 1: the co-launched topology has been built as well
#else
    hipStream_t stream = (hipStream_t)stream_;
    if (co_ok) {
        CoLaunch C;
        C.net = L; C.topo = *co; C.n_net = blocks;
        const int64_t both = use_lds > co_lds ? use_lds : co_lds;
#define DRGNN_CO_LAUNCH(K)                                                                               \
    do {                                                                                                 \
        if (both > 64 * 1024)                                                                            \
            HIP_TRY(hipFuncSetAttribute((const void*)k_net_co_topo<K, BWD>,                              \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)both));         \
        hipLaunchKernelGGL((k_net_co_topo<K, BWD>), dim3((unsigned)(blocks + co->args.n_graphs * co->roles)), \
                           dim3(DRGNN_NTHREADS), (size_t)both, stream, C);                               \
    } while (0)
        if (kind == DRGNN_GINET) DRGNN_CO_LAUNCH(DRGNN_GINET);
        else if (kind == DRGNN_SGAT) DRGNN_CO_LAUNCH(DRGNN_SGAT);
        else DRGNN_CO_LAUNCH(DRGNN_FOUT);
#undef DRGNN_CO_LAUNCH
        HIP_TRY(hipGetLastError());
        return 1;
    }
    if (blocks == 0) return 0;
#define DRGNN_NET_LAUNCH(K)                                                                         \
    do {                                                                                            \
        if (use_lds > 64 * 1024)                                                                    \
            HIP_TRY(hipFuncSetAttribute((const void*)k_net<K, BWD, true>,                           \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)use_lds)); \
        if (use_lds)                                                                                \
            hipLaunchKernelGGL((k_net<K, BWD, true>), dim3((unsigned)blocks), dim3(DRGNN_NTHREADS), \
                               (size_t)use_lds, stream, L);                                         \
        else                                                                                        \
            hipLaunchKernelGGL((k_net<K, BWD, false>), dim3((unsigned)blocks), dim3(DRGNN_NTHREADS),\
                               0, stream, L);                                                       \
    } while (0)
    if (kind == DRGNN_GINET) DRGNN_NET_LAUNCH(DRGNN_GINET);
    else if (kind == DRGNN_SGAT) DRGNN_NET_LAUNCH(DRGNN_SGAT);
    else DRGNN_NET_LAUNCH(DRGNN_FOUT);
#undef DRGNN_NET_LAUNCH
    HIP_TRY(hipGetLastError());
    return 0;
#endif
}

extern "C" {

int drgnn_net_forward(const drgnn_net_desc* net, const float* x, const int32_t* ws_i32,
                      const float* ws_f32, int64_t n_nodes, int64_t n_edges, int64_t n_graphs,
                      int32_t max_nodes, int32_t max_edges, int32_t max_c0, float* xp, int32_t* arg0,
                      int32_t* arg1, float* readout, float* scratch_f32, int32_t* step_inc, void* stream_) {
    int rc = net_check(net);
    if (rc) return rc;
    if (!x || !ws_i32 || !xp || !arg0 || !arg1 || !readout) return DRGNN_E_ARG;
    if (net->kind == DRGNN_SGAT && !ws_f32) return DRGNN_E_ARG;
    TopoLayout lay;
    topo_layout(n_nodes, n_edges, n_graphs, &lay);
    NetLaunch L;
    L.a.net = *net;
    L.a.x = x;
    L.a.tv = topo_view(const_cast<int32_t*>(ws_i32), const_cast<float*>(ws_f32), lay);
    L.a.n_nodes = n_nodes;
    L.a.n_graphs = (int)n_graphs;
    L.a.xp = xp; L.a.arg0 = arg0; L.a.arg1 = arg1; L.a.readout = readout;
    L.a.grad_readout = nullptr; L.a.partials = nullptr; L.a.grad_x = nullptr; L.a.n_partial = 0;
    L.a.step_inc = step_inc;
    L.a.hf.enabled = 0;
    L.n_edges = n_edges;
    const int rc2 = net_launch<false>(L, max_nodes, max_edges, max_c0, scratch_f32, stream_);
    return rc2 < 0 ? rc2 : (rc2 > 1 ? rc2 : 0);
}

static int net_backward_impl(const drgnn_net_desc* net, const float* x, const float* grad_readout,
                             const HeadFused* hf, const int32_t* ws_i32, const float* ws_f32,
                             int64_t n_nodes, int64_t n_edges, int64_t n_graphs, int32_t max_nodes,
                             int32_t max_edges, int32_t max_c0, const float* xp, const int32_t* arg0,
                             const int32_t* arg1, float* grad_x, float* partials, float* scratch_f32,
                             int32_t* step_inc, void* stream_, const drgnn_topology_request* next = nullptr) {
    int rc = net_check(net);
    if (rc) return rc;
    if (!x || (!grad_readout && !hf) || !ws_i32 || !xp || !arg0 || !arg1 || !partials) return DRGNN_E_ARG;
    TopoLayout lay;
    topo_layout(n_nodes, n_edges, n_graphs, &lay);
    NetLaunch L;
    L.a.net = *net;
    L.a.x = x;
    L.a.tv = topo_view(const_cast<int32_t*>(ws_i32), const_cast<float*>(ws_f32), lay);
    L.a.n_nodes = n_nodes;
    L.a.n_graphs = (int)n_graphs;
    L.a.xp = const_cast<float*>(xp); L.a.arg0 = const_cast<int32_t*>(arg0);
    L.a.arg1 = const_cast<int32_t*>(arg1); L.a.readout = nullptr;
    L.a.grad_readout = grad_readout; L.a.partials = partials; L.a.grad_x = grad_x;
    L.a.n_partial = (int)net_partial_floats(net->n_feat);
    L.a.step_inc = step_inc;
    if (hf) L.a.hf = *hf; else L.a.hf.enabled = 0;
    L.n_edges = n_edges;
    if (!next) {
        const int rc2 = net_launch<true>(L, max_nodes, max_edges, max_c0, scratch_f32, stream_);
        return rc2 < 0 ? rc2 : (rc2 > 1 ? rc2 : 0);
    }
    TopoLaunch T;
    int64_t tlds = 0;
    rc = topo_prepare_req(T, &tlds, next);
    if (rc) return rc;
    const int rc2 = net_launch<true>(L, max_nodes, max_edges, max_c0, scratch_f32, stream_, &T, tlds);
    if (rc2 < 0 || rc2 > 1) return rc2;
    if (rc2 == 1) return 0;
    // could not share the launch: build the next topology with its own launches
    return drgnn_topology_build_request(next, stream_);
}

}  // extern "C"
extern "C" {

int drgnn_net_backward(const drgnn_net_desc* net, const float* x, const float* grad_readout,
                       const int32_t* ws_i32, const float* ws_f32, int64_t n_nodes, int64_t n_edges,
                       int64_t n_graphs, int32_t max_nodes, int32_t max_edges, int32_t max_c0,
                       const float* xp, const int32_t* arg0, const int32_t* arg1, float* grad_x,
                       float* partials, float* scratch_f32, int32_t* step_inc, void* stream_) {
    return net_backward_impl(net, x, grad_readout, nullptr, ws_i32, ws_f32, n_nodes, n_edges, n_graphs,
                             max_nodes, max_edges, max_c0, xp, arg0, arg1, grad_x, partials, scratch_f32,
                             step_inc, stream_);
}

int drgnn_net_backward_fused_head(const drgnn_net_desc* net, const drgnn_head_desc* hd, const float* x,
                                  const float* readout, const void* target, const int32_t* step,
                                  const int32_t* ws_i32, const float* ws_f32, int64_t n_nodes,
                                  int64_t n_edges, int64_t n_graphs, int32_t max_nodes, int32_t max_edges,
                                  int32_t max_c0, const float* xp, const int32_t* arg0, const int32_t* arg1,
                                  float* pred, float* head_partials, float* grad_x, float* partials,
                                  float* scratch_f32, const drgnn_topology_request* next_topology,
                                  void* stream_) {
    if (!hd || !hd->w1 || !hd->b1 || !hd->w2 || !hd->b2 || !readout || !target || !pred || !head_partials)
        return DRGNN_E_ARG;
    if (!net || hd->R != DRGNN_H2 * net->n_branch || hd->H < 1 || hd->H > 512 || hd->O < 1 || hd->O > DRGNN_MAX_OUT)
        return DRGNN_E_WIDTH;
    HeadFused hf;
    hf.enabled = 1; hf.B = (int)n_graphs; hf.R = hd->R; hf.H = hd->H; hf.O = hd->O; hf.task = hd->task;
    hf.p_drop = hd->train ? hd->p_drop : 0.0f; hf.seed = hd->seed; hf.step_bias = -1; hf.train = 1; hf.sigmoid = hd->transform_sigmoid;
    hf.drop_mask = hd->train ? hd->drop_mask : nullptr;
    hf.w1 = hd->w1; hf.b1 = hd->b1; hf.w2 = hd->w2; hf.b2 = hd->b2; hf.class_w = hd->class_w;
    hf.y_reg = (hd->task == DRGNN_TASK_REG) ? (const float*)target : nullptr;
    hf.y_cls = (hd->task == DRGNN_TASK_CLASS) ? (const int64_t*)target : nullptr;
    hf.readout = readout; hf.step = step; hf.pred = pred; hf.partials = head_partials;
    return net_backward_impl(net, x, nullptr, &hf, ws_i32, ws_f32, n_nodes, n_edges, n_graphs, max_nodes,
                             max_edges, max_c0, xp, arg0, arg1, grad_x, partials, scratch_f32, nullptr, stream_,
                             next_topology);
}

// ---- fused training step ---------------------------------------------------------------------
static int64_t step_lds_bytes(int kind, int F, int capN, int capE, int capC, int R, int H, int O) {
    return 4 * step_scratch_words(kind, F, capN, capE, capC, R, H, O);
}

int64_t drgnn_net_step_lds_bytes(int32_t kind, int32_t n_feat, int32_t max_nodes, int32_t max_edges,
                                 int32_t max_c0, int32_t R, int32_t H, int32_t O) {
    if (max_nodes <= 0 || max_nodes > 32767 || max_edges > 65535 || n_feat > 256) return 0;   // else: the launch pair
    const int capC = (max_c0 > 0 && max_c0 < max_nodes) ? max_c0 : max_nodes;
    return step_lds_bytes(kind, n_feat, max_nodes, max_edges > 0 ? max_edges : 1, capC, R, H, O);
}

int64_t drgnn_head_compact_elems(int32_t R, int32_t H, int32_t O) { return head_compact_floats(R, H, O); }

// ---- launch plan of the fused step (include/drgnn.h: drgnn_step_plan) ---------------------------------------------------
// CUs this process may count on for co-residency of a launch's workgroups: the device's CU count, per device id (ADVICE r03:
// a process may drive devices of different sizes).  DRGNN_RESIDENT_CUS=<n> overrides it (a CU mask -- HSA_CU_MASK /
// ROC_GLOBAL_CU_MASK -- or a share of the GPU leaves fewer CUs than the attribute says); DRGNN_SHARED_GPU=1 means "assume
// nothing": every layout that waits for a partner workgroup is off (one workgroup per graph).
static int device_cu_count() {
#ifdef DRGNN_EMU
    return 256;
#else
    static int cus[64];
    static int env_cus = -2;
    if (env_cus == -2) {
        const char* sh = getenv("DRGNN_SHARED_GPU");
        const char* lim = getenv("DRGNN_RESIDENT_CUS");
        env_cus = (sh && sh[0] == '1') ? 0 : (lim && atoi(lim) > 0) ? atoi(lim) : -1;
    }
    if (env_cus >= 0) return env_cus;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 1;      // unknown device: never assume co-residency
    if (cus[dev] == 0) {
        int n = 0;
        cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 1;
    }
    return cus[dev];
#endif
}
#ifndef DRGNN_EMU
static int step_current_device() { int dev = 0; return hipGetDevice(&dev) == hipSuccess ? dev : 0; }
#endif

// the overrides of a plan (all 0 = automatic).  DRGNN_STEP_PLAN=<one|two|noclass|product|nosplit|seq>[,...] (read once) gives
// the defaults of a process for launches whose plan overrides nothing: same-box A/B runs of whole programs
struct StepOverrides { int force_wgs, no_class, no_aggregate, no_split, no_paired; };
static StepOverrides step_env_overrides() {
    static int parsed = 0;
    static StepOverrides env = {0, 0, 0, 0, 0};
    if (!parsed) {
        StepOverrides o = {0, 0, 0, 0, 0};
        const char* v = getenv("DRGNN_STEP_PLAN");
        if (v) {
            if (strstr(v, "one")) o.force_wgs = 1;
            if (strstr(v, "two")) o.force_wgs = 2;
            if (strstr(v, "noclass")) o.no_class = 1;
            if (strstr(v, "product")) o.no_aggregate = 1;
            if (strstr(v, "nosplit")) o.no_split = 1;
            if (strstr(v, "seq")) o.no_paired = 1;
        }
        env = o;
        parsed = 1;
    }
    return env;
}
// DRGNN_NO_PREFETCH (A/B runs): cached-topology launches without the next mini-batch's prefetch workgroups; read once
static bool step_no_prefetch() {
    static int v = -1;
    if (v < 0) v = getenv("DRGNN_NO_PREFETCH") != nullptr ? 1 : 0;
    return v != 0;
}
static StepOverrides step_overrides_of(const drgnn_step_plan* p) {
    if (p && (p->force_wgs || p->no_class || p->no_aggregate || p->no_split || p->no_paired))
        return StepOverrides{p->force_wgs, p->no_class, p->no_aggregate, p->no_split, p->no_paired};
    return step_env_overrides();
}

static int64_t step1_lds_bytes_form(int F, int capN, int capE, int capC, int H, int O, int paired) {
    return 4 * step1_scratch_words(F, capN, capE, capC, H, O, paired);
}
// padded feature width of the width-specialised kernels a launch of these bounds may take (16 / 32 / 48 / 64), 0 = generic.
// x == nullptr: alignment not judged here
static int step_variant(int kind, const float* x, int F, int capN, int capE, int capC, int H, int O) {
    if (!step_burst_guaranteed(kind, x, F, capN, capE, capC, H, O)) return 0;
    const int f16 = step_pad16(F);
    return (f16 == 16 || f16 == 32 || f16 == 48 || f16 == 64) ? f16 : 0;
}

// ... of the AGGREGATION-FIRST kernels: they read padded tile rows, so any feature count up to 64 has a width class; what
// remains of step_burst_guaranteed are the head width and the burst capacities
// from_memory: the form that reads the S rows of the tiles from memory (GINet, drgnn_step3.h SG): no register burst of the tile
static int step_af_width(int kind, int F, int capN, int capE, int capC, int H, int O, bool from_memory = false) {
    if (H != ((kind == DRGNN_GINET) ? 128 : 64) || F < 1 || F > 64) return 0;
    const int TF = (F + 3) & ~3;
    const bool ok = (F * DRGNN_H1 <= DRGNN_BCAP) && (from_memory || (long)capN * TF <= 16L * DRGNN_BCAP) && (capN + 1 <= DRGNN_BCAP) &&
                    (capE <= 2 * DRGNN_BCAP) && (capC * DRGNN_H1 <= 4 * DRGNN_BCAP) && O * H <= 2 * DRGNN_BCAP &&
                    H * 8 <= STEP_WB_J * DRGNN_BCAP;
    return ok ? step_pad16(F) : 0;
}

// what a launch is, as far as its layout goes
struct StepAsk {
    int kind, F, capN, capE, capC, R, H, O;
    int64_t B;
    int64_t co;             // graphs of the topology the launch co-builds (0: none, or not co-launchable)
    int co_roles;           // workgroups per graph that builder may take: 2 or 1
    bool train;
    bool x_ok;              // x is 16-byte aligned (or not known to be otherwise)
    int topo_flags;         // HIER / TILES (tiles usable) / LEAN of the workspace the launch reads
    int commit_wgs;         // single-branch nets: workgroups per graph the caller sized its buffers for (0: free choice)
    StepOverrides ov;
};
enum { SK_STEP = 0, SK_STEP1 = 1, SK_AF2 = 2, SK_AF3 = 3, SK_AF3B = 4 };
struct StepPick {
    int rc;                 // 0, or the error a launch of this shape returns (family NONE)
    int family, kernel, wgs, slabs, width, cls, paired, lean_ok, builder_roles;
    int xg;                 // sGAT / FoutNet: 1 = the x-from-memory form (the S and the x tile together do not fit the LDS), 2 = S too
    int sg;                 // GINet, one workgroup per graph: the S-from-memory form (the S tile does not fit the LDS)
    int64_t lds, xchg_words;
    int capN, capE, capC;   // the LDS capacities the kernel is launched with (the class's when cls)
};
static const int64_t STEP_LDS_NEVER = (int64_t)1 << 50;

static StepPick step_pick(const StepAsk& q) {
    StepPick k;
    memset(&k, 0, sizeof(k));
    k.capN = q.capN; k.capE = q.capE; k.capC = q.capC;
    k.builder_roles = q.co > 0 ? q.co_roles : 0;
    k.slabs = (q.kind == DRGNN_GINET) ? 2 : 1;
    const int cus = device_cu_count();
    const int64_t co = q.co > 0 ? q.co : 0;
    auto two_ok = [&](int64_t extra) {
        if (q.ov.force_wgs == 1) return false;
        if (q.ov.force_wgs == 2) return true;
        return 2 * q.B + extra <= cus;
    };
    // the aggregation-first family: a workspace with the hierarchical order and usable tiles, a width class, the reference head
    k.width = step_variant(q.kind, nullptr, q.F, q.capN, q.capE, q.capC, q.H, q.O);
    if (!q.x_ok) k.width = 0;
    const int old_width = k.width;
#ifdef DRGNN_EMU
    const bool af_shape = false;
    const int af_w = 0;
#else
    // (the single-branch nets read x rows next to the tiles: the input's when F % 4 == 0 -- 16-byte aligned then --, else the
    // tiles' padded copy; GINet reads the tiles only)
    int af_w = step_af_width(q.kind, q.F, q.capN, q.capE, q.capC, q.H, q.O);
    const bool af_ws = !q.ov.no_aggregate && (q.topo_flags & DRGNN_TOPO_HIER) && (q.topo_flags & DRGNN_TOPO_TILES);
    const bool af_shape = af_ws && af_w != 0 && (q.kind == DRGNN_GINET || (q.F & 3) != 0 || q.x_ok);
    // (the forms that leave the tiles' S rows in memory -- GINet's one-workgroup kernel, the 48- / 64-wide single-branch kernels:
    // graphs the staged forms have no LDS -- or no register burst -- for)
    const int sg_w = af_ws ? step_af_width(q.kind, q.F, q.capN, q.capE, q.capC, q.H, q.O, true) : 0;
#endif
    // The PRODUCT-FIRST family (drgnn_step.h / drgnn_step1.h, rounds 2 - 3) is the host emulation's only: the device library
    // instantiates the aggregation-first kernels alone (round 6), and a launch they do not cover -- a head that is not the
    // reference's, more than 64 features, a workspace without the hierarchical order or usable tiles, the no_aggregate override --
    // is family NONE: the launch pair (drgnn_net_forward + drgnn_net_backward_fused_head) steps it.
#ifdef DRGNN_EMU
    const bool old_ok = true;
#else
    const bool old_ok = false;
#endif
    if (q.kind == DRGNN_GINET) {
        const bool narrow = q.H < DRGNN_H2;      // (the exchange words of a graph are 2 x 32 of its 2 x H: a narrower head runs one workgroup per graph)
        const int64_t l2af = af_shape ? 4 * step3_scratch_words(q.F, q.capN, q.capE, q.capC, q.H, q.O) : STEP_LDS_NEVER;
        const int64_t l2old = old_ok ? step_lds_bytes(q.kind, q.F, q.capN, q.capE, q.capC, q.R, q.H, q.O) : STEP_LDS_NEVER;
        int64_t l1af = af_shape ? 4 * step3b_scratch_words(q.F, q.capN, q.capE, q.capC, q.H, q.O) : STEP_LDS_NEVER;
#ifndef DRGNN_EMU
        if (l1af > DRGNN_LDS_LIMIT && sg_w != 0) {
            const int64_t l1sg = 4 * step3b_scratch_words(q.F, q.capN, q.capE, q.capC, q.H, q.O, 1);
            if (l1sg <= DRGNN_LDS_LIMIT) { l1af = l1sg; k.sg = 1; af_w = sg_w; }
        }
#endif
        const int64_t l1p = old_ok ? step1_lds_bytes_form(q.F, q.capN, q.capE, q.capC, q.H, q.O, 1) : STEP_LDS_NEVER;
        const bool paired = !q.ov.no_paired && l1p <= DRGNN_LDS_LIMIT;
        const int64_t l1old = paired ? l1p : old_ok ? step1_lds_bytes_form(q.F, q.capN, q.capE, q.capC, q.H, q.O, 0) : STEP_LDS_NEVER;
        const bool af_two = l2af <= DRGNN_LDS_LIMIT, af_one = l1af <= DRGNN_LDS_LIMIT;
        const bool can_two = !narrow && (af_two || l2old <= DRGNN_LDS_LIMIT);
        const bool can_one = af_one || l1old <= DRGNN_LDS_LIMIT;
        // two workgroups per graph only while every workgroup of the launch is resident -- with the builder at two
        // workgroups per graph if that fits, else at one; else one workgroup per graph; else two with the builder on its own
        int wgs = 0;
        if (can_two && two_ok(co * k.builder_roles)) wgs = 2;
        else if (can_two && k.builder_roles == 2 && two_ok(co)) { wgs = 2; k.builder_roles = 1; }
        else if (can_one) wgs = 1;
        else if (can_two && two_ok(0)) { wgs = 2; k.builder_roles = 0; }
        else { k.rc = DRGNN_E_CAPACITY; return k; }
        k.wgs = wgs;
        if (wgs == 2) { k.kernel = af_two ? SK_AF3 : SK_STEP; k.lds = af_two ? l2af : l2old; }
        else { k.kernel = af_one ? SK_AF3B : SK_STEP1; k.lds = af_one ? l1af : l1old; k.paired = (!af_one && paired) ? 1 : 0; }
        if (wgs == 2) k.sg = 0;      // (the S-from-memory form is the one-workgroup kernel's)
        k.xchg_words = 2 * (int64_t)(q.H > DRGNN_H2 ? q.H : DRGNN_H2);
    } else {
        int64_t laf = af_shape ? 4 * step2_scratch_words(q.kind, q.F, q.capN, q.capE, q.capC, q.H, q.O) : STEP_LDS_NEVER;
        if (af_shape && laf > DRGNN_LDS_LIMIT) {
            // graphs whose S and x tiles do not fit together (49 - 64 features at 200 nodes, 260+ nodes at the narrower widths):
            // the x rows stay in memory (drgnn_step2.h, XG)
            const int64_t lxg = 4 * step2_scratch_words(q.kind, q.F, q.capN, q.capE, q.capC, q.H, q.O, 1);
            if (lxg <= DRGNN_LDS_LIMIT) { laf = lxg; k.xg = 1; }
        }
#ifndef DRGNN_EMU
        if (laf > DRGNN_LDS_LIMIT && sg_w >= 32 && ((q.F & 3) != 0 || q.x_ok)) {
            // ... and where the S tile alone is too much (the 32-, 48- and 64-wide kernels beyond 250 - 360 nodes): the S rows too
            const int64_t lxg = 4 * step2_scratch_words(q.kind, q.F, q.capN, q.capE, q.capC, q.H, q.O, 2);
            if (lxg <= DRGNN_LDS_LIMIT) { laf = lxg; k.xg = 2; af_w = sg_w; }
        }
#endif
        const int64_t lold = old_ok ? step_lds_bytes(q.kind, q.F, q.capN, q.capE, q.capC, q.R, q.H, q.O) : STEP_LDS_NEVER;
        const bool af_ok = laf <= DRGNN_LDS_LIMIT;
        // the node-split layout: training launches of the aggregation-first kernels under GINet's residency rule
        int wgs = 1;
        // (training launches only.  Round 6 instantiated the split layout for inference launches too and measured it on the same
        // box, predict_epoch at batch 64: sGAT 14.8 us per mini-batch against 13.35 with one workgroup per graph (cached topology),
        // FoutNet 14.6 against 13.1 -- a forward alone is too short for its two hand-offs, profiles/r06_split_inference_ab.txt)
        const bool may_split = q.train && af_ok && !q.ov.no_split;
        if (q.commit_wgs == 2) {
            // the caller sized its buffers for two workgroups per graph: no silent fallback to another layout
            if (!may_split) { k.rc = DRGNN_E_CAPACITY; return k; }
            if (two_ok(co * k.builder_roles)) wgs = 2;
            else if (k.builder_roles == 2 && two_ok(co)) { wgs = 2; k.builder_roles = 1; }
            else if (two_ok(0)) { wgs = 2; k.builder_roles = 0; }
            else { k.rc = DRGNN_E_CAPACITY; return k; }
        } else if (q.commit_wgs == 0 && may_split) {
            // (a launch without a builder counts one idle CU per graph all the same: with every CU holding a half graph the
            // hand-offs cost more than the halved phases give and k_update sums twice the slabs -- cached topology at batch
            // 128, us per step, split / whole: sGAT 20.7 / 19.2, FoutNet 20.5 / 19.5; profiles/r05_batch_sweep.txt)
            if (co == 0) { if (two_ok(q.B)) wgs = 2; }
            else if (two_ok(co * k.builder_roles)) wgs = 2;
            else if (k.builder_roles == 2 && two_ok(co)) { wgs = 2; k.builder_roles = 1; }
        }
        k.wgs = wgs;
        if (af_ok) { k.kernel = SK_AF2; k.lds = laf; }
        else if (lold <= DRGNN_LDS_LIMIT) { k.kernel = SK_STEP; k.lds = lold; }
        else { k.rc = DRGNN_E_CAPACITY; return k; }
        k.slabs = wgs;
        k.xchg_words = (wgs == 2) ? step2_xchg_words(q.capC > 0 ? q.capC : 1) : 0;
    }
    // one round of workgroups instead of two: the builder at ONE workgroup per graph when that makes the launch resident
    // (measured, tools/r03_split_sweep.sh: sGAT at batch 128, 128 step + 256 builder workgroups = two rounds, 37.2 us per
    // step; 128 + 128 = one round, 29.0 us)
    if (k.builder_roles == 2 && k.wgs == 1 && q.B + 2 * co > cus && q.B + co <= cus) k.builder_roles = 1;
    k.lean_ok = (k.kernel == SK_AF2 || k.kernel == SK_AF3 || k.kernel == SK_AF3B) ? 1 : 0;
    k.width = k.lean_ok ? af_w : old_width;
    k.family = k.lean_ok ? DRGNN_STEP_FAMILY_AGGREGATE : DRGNN_STEP_FAMILY_PRODUCT;
    // Capacity class (drgnn_step.h: STEP_CLS_*): a batch whose maxima lie inside the class is stepped by the 32-wide kernels
    // whose LDS layout is a compile-time constant (of the one-workgroup product-first GINet layouts the paired form; of the
    // aggregation-first kernels the training and the inference instances)
#ifndef DRGNN_EMU
    // (48-wide: the aggregation-first kernels only -- the feature count of the reference's shipped regression models)
    if (!q.ov.no_class && !k.sg && !k.xg && (k.width == 32 || (k.width == 48 && k.lean_ok)) && q.capN <= STEP_CLS_N && q.capE <= STEP_CLS_E &&
        q.capC <= STEP_CLS_C && !(k.kernel == SK_STEP1 && !k.paired) &&
        (k.lean_ok ? step_af_width(q.kind, q.F, STEP_CLS_N, STEP_CLS_E, STEP_CLS_C, q.H, q.O)
                   : step_variant(q.kind, nullptr, q.F, STEP_CLS_N, STEP_CLS_E, STEP_CLS_C, q.H, q.O)) == k.width) {
        // (the 32-wide class instance of the one-workgroup kernel keeps Z1 / XP / dS per branch: STEP3B_DUAL)
        const int64_t lc = k.kernel == SK_AF3B ? 4 * (step3b_scratch_words(q.F, STEP_CLS_N, STEP_CLS_E, STEP_CLS_C, q.H, q.O) +
                                                      (STEP3B_DUAL(k.width, 1, q.train) ? step3b_dual_extra_words(STEP_CLS_N, STEP_CLS_C) : 0))
                         : k.kernel == SK_AF3 ? 4 * step3_scratch_words(q.F, STEP_CLS_N, STEP_CLS_E, STEP_CLS_C, q.H, q.O)
                         : k.kernel == SK_AF2 ? 4 * step2_scratch_words(q.kind, q.F, STEP_CLS_N, STEP_CLS_E, STEP_CLS_C, q.H, q.O)
                         : k.kernel == SK_STEP1 ? step1_lds_bytes_form(q.F, STEP_CLS_N, STEP_CLS_E, STEP_CLS_C, q.H, q.O, 1)
                                                : step_lds_bytes(q.kind, q.F, STEP_CLS_N, STEP_CLS_E, STEP_CLS_C, q.R, q.H, q.O);
        if (lc <= DRGNN_LDS_LIMIT) {
            k.cls = 1;
            k.capN = STEP_CLS_N; k.capE = STEP_CLS_E; k.capC = STEP_CLS_C;
            k.lds = lc;
        }
    }
#endif
    return k;
}

static bool step_bounds_ok(int32_t max_nodes, int32_t max_edges, int32_t n_feat) {
    return max_nodes > 0 && max_nodes <= 32767 && max_edges <= 65535 && n_feat <= 256;
}

int32_t drgnn_net_step_plan(drgnn_step_plan* p) {
    if (!p) return 0;
    p->family = DRGNN_STEP_FAMILY_NONE; p->wgs_per_graph = 0; p->slabs_per_graph = 0; p->width = 0; p->cls = 0; p->lean_ok = 0;
    p->builder_wgs_per_graph = 0; p->lds_bytes = 0; p->xchg_words = 0; p->from_memory = 0; p->reserved = 0;
    if (!step_bounds_ok(p->max_nodes, p->max_edges, p->n_feat) || p->n_graphs < 0 || p->H < 1 || p->H > 512 || p->O < 1 ||
        p->O > DRGNN_MAX_OUT || p->kind < 0 || p->kind > DRGNN_FOUT)
        return 0;
    StepAsk q;
    q.kind = p->kind; q.F = p->n_feat; q.capN = p->max_nodes; q.capE = p->max_edges > 0 ? p->max_edges : 1;
    q.capC = (p->max_c0 > 0 && p->max_c0 < p->max_nodes) ? p->max_c0 : p->max_nodes;
    q.R = p->R; q.H = p->H; q.O = p->O; q.B = p->n_graphs;
    q.co = p->co_built_graphs > 0 ? p->co_built_graphs : 0;
    q.co_roles = (q.co > 0 && q.co <= DRGNN_TOPO_SPLIT_MAX_GRAPHS) ? 2 : 1;
    q.train = p->train != 0; q.x_ok = true; q.topo_flags = p->topo_flags; q.commit_wgs = 0;
    q.ov = step_overrides_of(p);
    const StepPick k = step_pick(q);
    if (k.rc) return 0;
    p->family = k.family; p->wgs_per_graph = k.wgs; p->slabs_per_graph = k.slabs; p->width = k.width; p->cls = k.cls;
    p->lean_ok = k.lean_ok; p->builder_wgs_per_graph = k.builder_roles; p->lds_bytes = k.lds; p->xchg_words = k.xchg_words;
    p->from_memory = (k.sg || k.xg) ? 1 : 0;
    return k.wgs;
}

int64_t drgnn_net_step_xchg_elems(int32_t kind, int32_t max_nodes, int32_t max_c0, int32_t H) {
    if (kind == DRGNN_GINET) return 2 * (int64_t)(H > DRGNN_H2 ? H : DRGNN_H2);
    const int capC = (max_c0 > 0 && max_c0 < max_nodes) ? max_c0 : max_nodes;
    return step2_xchg_words(capC > 0 ? capC : 1);
}

int32_t drgnn_net_step_variant(int32_t kind, const float* x, int32_t n_feat, int32_t max_nodes, int32_t max_edges,
                               int32_t max_c0, int32_t H, int32_t O) {
    const int capC = (max_c0 > 0 && max_c0 < max_nodes) ? max_c0 : max_nodes;
    return step_variant(kind, x, n_feat, max_nodes, max_edges > 0 ? max_edges : 1, capC, H, O);
}

#ifndef DRGNN_EMU
// One launch of a step kernel instance (+ the co-launched builder's workgroups).  These kernels use up to the whole 160 KiB of
// LDS: the attribute is raised once per kernel instance and device (the call costs host time on every launch otherwise).
static int step_launch(drgnn_step_kernel_t kern, int64_t lds_bytes, unsigned grid, hipStream_t stream, const StepCoLaunch& C) {
    if (!kern) return DRGNN_E_ARG;
    if (lds_bytes > 64 * 1024) {
        struct Seen { const void* fn; int dev; };
        static Seen seen[256];
        static int n_seen = 0;
        const int dev = step_current_device();
        bool known = false;
        const int n = __atomic_load_n(&n_seen, __ATOMIC_ACQUIRE);
        for (int i = 0; i < n && !known; ++i) known = seen[i].fn == (const void*)kern && seen[i].dev == dev;
        if (!known) {
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DRGNN_LDS_LIMIT) != hipSuccess) {
                (void)hipGetLastError();      // (profiling builds carry a static LDS word: ask for what this launch needs)
                HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            } else {
                // (a cache, not state: a lost race only repeats the idempotent call above)
                const int slot = __atomic_fetch_add(&n_seen, 0, __ATOMIC_RELAXED);
                if (slot < 256) { seen[slot].fn = (const void*)kern; seen[slot].dev = dev; __atomic_store_n(&n_seen, slot + 1, __ATOMIC_RELEASE); }
            }
        }
    }
    void* args[] = {const_cast<StepCoLaunch*>(&C)};
    HIP_TRY(hipLaunchKernel((const void*)kern, dim3(grid), dim3(DRGNN_NTHREADS), args, (size_t)lds_bytes, stream));
    return 0;
}
#endif

// n_graphs graphs of the launch; (n_nodes, n_edges, ws_graphs): the shape the workspace was laid out for (the
// mini-batch itself, or a whole cached set whose graphs gather_ids[0..n_graphs) are stepped)
static int train_step_impl(const drgnn_net_desc* net, const drgnn_head_desc* hd, const float* x,
                           const void* target, int32_t* step2, const int32_t* ws_i32, const float* ws_f32,
                           int64_t n_nodes, int64_t n_edges, int64_t n_graphs, int64_t ws_graphs,
                           const int32_t* gather_ids, int32_t max_nodes,
                           int32_t max_edges, int32_t max_c0, float* pred, float* readout,
                           float* head_partials, float* partials, uint64_t* xchg,
                           const drgnn_topology_request* next, const drgnn_step_hints* hints, void* stream_) {
    int rc = net_check(net);
    if (rc) return rc;
    if (!hd || !hd->w1 || !hd->b1 || !hd->w2 || !hd->b2 || !x || !step2 || !ws_i32 || !pred || !readout)
        return DRGNN_E_ARG;
    // hd->train: 1 = the whole step; 0 = inference; 2 = the forward of a training step (predictions only, dropout on)
    if (hd->train < 0 || hd->train > 2) return DRGNN_E_ARG;
    const bool full_step = hd->train == 1;
    if (full_step && (!target || !head_partials || !partials)) return DRGNN_E_ARG;      // inference needs neither
    if (hd->task != DRGNN_TASK_REG && hd->task != DRGNN_TASK_CLASS && hd->task != DRGNN_TASK_GRAD) return DRGNN_E_ARG;
    if (hd->task == DRGNN_TASK_GRAD && gather_ids) return DRGNN_E_ARG;      // (the upstream gradient is indexed by slot)
    if (net->n_branch > 1 && !xchg) return DRGNN_E_ARG;
    if (net->kind == DRGNN_SGAT && !ws_f32) return DRGNN_E_ARG;
    if (hd->R != DRGNN_H2 * net->n_branch || hd->H < 1 || hd->H > 512 || hd->O < 1 || hd->O > DRGNN_MAX_OUT)
        return DRGNN_E_WIDTH;
    if (!step_bounds_ok(max_nodes, max_edges, net->n_feat)) return DRGNN_E_CAPACITY;
    const int kind = net->kind, F = net->n_feat;
    // the co-launched builder of the next mini-batch's topology
    TopoLaunch T;
    int64_t tlds = 0;
    bool co_ok = false;
    if (next) {
        rc = topo_prepare_req(T, &tlds, next);
        if (rc) return rc;
        co_ok = T.capN > 0 && T.user_nptr != nullptr && !(T.args.cluster1 != nullptr && T.args.c1_ptr == nullptr) &&
                T.args.n_graphs > 0 && n_graphs > 0;
        // the builder inside a GINet / FoutNet step launch is compiled without the edge-weight path: a request that wants
        // weights (another kind's topology) gets a launch of its own
        if (kind != DRGNN_SGAT && T.args.edge_attr != nullptr && T.tv.w0 != nullptr) co_ok = false;
    }
    // ---- the layout: one decision procedure for this launch and for drgnn_net_step_plan ----------------------------------
    StepAsk q;
    q.kind = kind; q.F = F; q.capN = max_nodes; q.capE = max_edges > 0 ? max_edges : 1;
    q.capC = (max_c0 > 0 && max_c0 < max_nodes) ? max_c0 : max_nodes;
    q.R = hd->R; q.H = hd->H; q.O = hd->O; q.B = n_graphs;
    q.co = co_ok ? (int64_t)T.args.n_graphs : 0;
    q.co_roles = co_ok ? T.roles : 1;
    q.train = full_step;
    q.x_ok = (((uintptr_t)x) & 15) == 0;
    q.topo_flags = hints ? hints->topo_flags : 0;
    // (the caller vouches for the tiles' flavour: weighted sums for sGAT, plain sums for FoutNet / GINet)
    if (!(hints && hints->tiles && ((((uintptr_t)hints->tiles) & 15) == 0))) q.topo_flags &= ~DRGNN_TOPO_TILES;
    const drgnn_step_plan* plan = hints ? hints->plan : nullptr;
    q.ov = step_overrides_of(plan);
    q.commit_wgs = (kind == DRGNN_GINET) ? 0 : (plan && plan->wgs_per_graph == 2) ? 2 : 1;
    if (q.commit_wgs == 2 && !xchg) return DRGNN_E_CAPACITY;
    const StepPick k = step_pick(q);
    if (k.rc) return k.rc;
    // a workspace built with DRGNN_TOPO_LEAN holds only what the aggregation-first kernels read
    if (hints && (hints->topo_flags & DRGNN_TOPO_LEAN) && !k.lean_ok) return DRGNN_E_ARG;
    // the autograd boundary (an upstream gradient instead of a target, a forward with the dropout mask of the step) is the
    // aggregation-first kernels': the product-first family predates it
    if ((hd->task == DRGNN_TASK_GRAD || hd->train == 2) && !k.lean_ok) return DRGNN_E_ARG;
    // a plan handed along with its `out` members filled (drgnn_net_step_plan) is a commitment: the launch takes the kernel
    // family, width class and slab layout the caller planned (and sized its buffers / made its assertions from) or fails --
    // never a silent third layout (a plan of another mode -- training against inference -- is only a carrier of overrides)
    if (plan && plan->family != DRGNN_STEP_FAMILY_NONE && (plan->train != 0) == q.train &&
        (plan->family != k.family || plan->width != k.width || plan->cls != k.cls || plan->slabs_per_graph != k.slabs))
        return DRGNN_E_CAPACITY;
    if (co_ok && k.builder_roles == 0) co_ok = false;      // the builder gets a launch of its own
    if (co_ok) T.roles = k.builder_roles;
    const bool one_wg = (kind == DRGNN_GINET) && k.wgs == 1;

    StepLaunch L;
    L.capN = k.capN; L.capE = k.capE; L.capC = k.capC;
    L.words = k.lds / 4;
    TopoLayout lay;
    topo_layout(n_nodes, n_edges, ws_graphs, &lay);
    StepArgs& a = L.a;
    a.gather_ids = gather_ids; a.ws_graphs = (int)ws_graphs;
    a.tiles = (q.topo_flags & DRGNN_TOPO_TILES) ? hints->tiles : nullptr; a.tile_nodes = n_nodes;
    L.dims.count = 0;
    // host-known graph numbers of a cached-topology launch are range-checked whatever the batch size: the kernel reads the
    // set's tables at gather_ids[g] as is
    if (gather_ids && hints && hints->host_ids)
        for (int64_t g = 0; g < n_graphs; ++g)
            if (hints->host_ids[g] < 0 || hints->host_ids[g] >= ws_graphs) return DRGNN_E_ARG;
    if (hints && n_graphs > 0 && n_graphs <= DRGNN_STEP_DIMS_MAX) {
        StepDims& D = L.dims;
        bool ok = true;
        if (!gather_ids && hints->host_node_ptr && hints->host_edge_ptr) {
            for (int g = 0; g < (int)n_graphs; ++g) {
                D.n0[g] = hints->host_node_ptr[g]; D.n[g] = hints->host_node_ptr[g + 1] - hints->host_node_ptr[g];
                D.e0[g] = hints->host_edge_ptr[g]; D.e[g] = hints->host_edge_ptr[g + 1] - hints->host_edge_ptr[g];
                D.gi[g] = g;
                ok = ok && D.n[g] >= 0 && D.e[g] >= 0 && D.n[g] <= q.capN && D.e[g] <= q.capE;
            }
            if (ok && (hints->host_node_ptr[n_graphs] != n_nodes || hints->host_edge_ptr[n_graphs] != n_edges)) ok = false;
        } else if (gather_ids && hints->host_ids && hints->set_node_ptr && hints->set_edge_ptr) {
            for (int g = 0; g < (int)n_graphs; ++g) {
                const int64_t id = hints->host_ids[g];
                if (id < 0 || id >= ws_graphs) { ok = false; break; }
                D.n0[g] = (int32_t)hints->set_node_ptr[id]; D.n[g] = (int32_t)(hints->set_node_ptr[id + 1] - hints->set_node_ptr[id]);
                D.e0[g] = (int32_t)hints->set_edge_ptr[id]; D.e[g] = (int32_t)(hints->set_edge_ptr[id + 1] - hints->set_edge_ptr[id]);
                D.gi[g] = (int32_t)id;
                ok = ok && D.n[g] <= q.capN && D.e[g] <= q.capE;
            }
        } else {
            ok = false;
        }
        if (ok) D.count = (int)n_graphs;       // (a table that contradicts the capacities is ignored: the device path decides)
    }
    a.net = *net; a.x = x;
    a.tv = topo_view(const_cast<int32_t*>(ws_i32), const_cast<float*>(ws_f32), lay);
    a.n_nodes = n_nodes; a.n_graphs = (int)n_graphs;
    a.partials = partials; a.n_partial = (int)net_partial_floats(F);
    a.xchg = (unsigned long long*)xchg; a.step2 = step2;
    a.xchg_stride = (int)step2_xchg_words(q.capC > 0 ? q.capC : 1);
    HeadFused& hf = a.hf;
    hf.enabled = 1; hf.B = (int)n_graphs; hf.R = hd->R; hf.H = hd->H; hf.O = hd->O; hf.task = hd->task;
    hf.p_drop = hd->train ? hd->p_drop : 0.0f; hf.seed = hd->seed; hf.step_bias = 0; hf.train = full_step ? 1 : 0; hf.sigmoid = hd->transform_sigmoid;
    hf.drop_mask = hd->train ? hd->drop_mask : nullptr;
    hf.w1 = hd->w1; hf.b1 = hd->b1; hf.w2 = hd->w2; hf.b2 = hd->b2; hf.class_w = hd->class_w;
    hf.y_reg = (hd->task != DRGNN_TASK_CLASS) ? (const float*)target : nullptr;
    hf.y_cls = (hd->task == DRGNN_TASK_CLASS) ? (const int64_t*)target : nullptr;
    hf.readout = readout; hf.step = step2; hf.pred = pred; hf.partials = head_partials; hf.stage = 0;

    // grid of the step part: graphs in groups of 8 x 2 when a graph has two workgroups (see step_block)
    const int blocks = (k.wgs == 2) ? (int)((n_graphs + 7) / 8) * 16 : (int)n_graphs;
    if (blocks > 0) {
#ifdef DRGNN_EMU
        // workgroups run one after the other here: two passes (up to the readout exchange, then the
        // rest), each workgroup keeping its "LDS" in a slab of its own between the passes
        std::vector<float> slabs((size_t)blocks * (size_t)(L.words + 16));
        if (one_wg) {
            for (int b = 0; b < blocks; ++b) {
                if (k.paired) {
                    if (gather_ids) step_block_both<0, true, true>(L, b, slabs.data());
                    else step_block_both<0, false, true>(L, b, slabs.data());
                } else {
                    if (gather_ids) step_block_both<0, true, false>(L, b, slabs.data());
                    else step_block_both<0, false, false>(L, b, slabs.data());
                }
            }
        } else
        for (int pass = 1; pass <= 2; ++pass)
            for (int b = 0; b < blocks; ++b) {
                float* lds_b = slabs.data() + (size_t)b * (size_t)(L.words + 16);
                if (gather_ids) {
                    if (kind == DRGNN_GINET) step_block<DRGNN_GINET, 0, true>(L, b, lds_b, pass);
                    else if (kind == DRGNN_SGAT) step_block<DRGNN_SGAT, 0, true>(L, b, lds_b, pass);
                    else step_block<DRGNN_FOUT, 0, true>(L, b, lds_b, pass);
                } else {
                    if (kind == DRGNN_GINET) step_block<DRGNN_GINET, 0>(L, b, lds_b, pass);
                    else if (kind == DRGNN_SGAT) step_block<DRGNN_SGAT, 0>(L, b, lds_b, pass);
                    else step_block<DRGNN_FOUT, 0>(L, b, lds_b, pass);
                }
            }
        if (co_ok) {
            std::vector<int> tbuf((size_t)(tlds / 4) + 16);
            for (int g = 0; g < T.args.n_graphs * T.roles; ++g) topo_block<true>(T, g, tbuf.data());
        }
        (void)stream_;
#else
        StepCoLaunch C;
        C.step = L; C.n_net = blocks;
        int64_t both = k.lds;
        int extra = 0;
        C.topo.pf_ids = nullptr; C.topo.pf_n = 0; C.topo.pf_graphs = 0;
        if (co_ok) { C.topo = T; both = k.lds > tlds ? k.lds : tlds; extra = T.args.n_graphs * T.roles; }
        else if (gather_ids && hints && hints->next_ids && hints->n_next > 0 && (blocks % 8) == 0 &&
                 blocks + hints->n_next <= device_cu_count() && !step_no_prefetch()) {
            // cached topology, CUs to spare: one extra workgroup per graph of the NEXT mini-batch warms the L2 of the XCD that
            // will step it (prefetch_block; beyond the resident size the workgroups would only queue behind the step's)
            TopoLaunch& Q = C.topo;
            memset(&Q, 0, sizeof(Q));
            Q.tv = a.tv;
            Q.pf_ids = hints->next_ids; Q.pf_n = (int)hints->n_next; Q.pf_graphs = a.ws_graphs;
            const int TF = (F + 3) & ~3;
            Q.pf_tiles = a.tiles; Q.pf_f = TF; Q.pf_tile_nodes = n_nodes;
            Q.pf_x = (kind == DRGNN_GINET) ? nullptr : (F & 3) ? a.tiles + drgnn_tiles_x_off(n_nodes, TF) : x;
            Q.pf_coef = (kind != DRGNN_GINET) ? 1 : 0;
            Q.pf_y = full_step ? target : nullptr; Q.pf_y_bytes = (hd->task == DRGNN_TASK_REG) ? 4 : 8;
            if (Q.pf_tiles != nullptr) extra = Q.pf_n; else Q.pf_ids = nullptr;
        }
        drgnn_step_kernel_t kern = nullptr;
        const bool gather = gather_ids != nullptr;
        switch (k.kernel) {
            case SK_AF3: kern = af_step_kernel(DRGNN_AF_GINET_TWO, k.width, gather, k.cls, 1, q.train); break;
            case SK_AF3B: kern = af_step_kernel(k.sg ? DRGNN_AF_GINET_SG : DRGNN_AF_GINET_ONE, k.width, gather, k.cls, 1, q.train); break;
            case SK_AF2: kern = af_step_kernel(k.xg ? (kind == DRGNN_SGAT ? DRGNN_AF_SGAT_XG : DRGNN_AF_FOUT_XG)
                                                    : (kind == DRGNN_SGAT ? DRGNN_AF_SGAT : DRGNN_AF_FOUT), k.width, gather, k.cls, k.wgs, q.train, k.xg); break;
            default: return DRGNN_E_CAPACITY;      // (the product-first family is not part of the device library: step_pick never picks it)
        }
        if ((rc = step_launch(kern, both, (unsigned)(blocks + extra), (hipStream_t)stream_, C))) return rc;
#endif
    }
    if (next && (!co_ok || blocks == 0))
        return drgnn_topology_build_request(next, stream_);
    return 0;
}

int drgnn_net_train_step(const drgnn_net_desc* net, const drgnn_head_desc* hd, const float* x,
                         const void* target, int32_t* step2, const int32_t* ws_i32, const float* ws_f32,
                         int64_t n_nodes, int64_t n_edges, int64_t n_graphs, int32_t max_nodes,
                         int32_t max_edges, int32_t max_c0, float* pred, float* readout,
                         float* head_partials, float* partials, uint64_t* xchg,
                         const drgnn_topology_request* next, const drgnn_step_hints* hints, void* stream_) {
    return train_step_impl(net, hd, x, target, step2, ws_i32, ws_f32, n_nodes, n_edges, n_graphs, n_graphs, nullptr,
                           max_nodes, max_edges, max_c0, pred, readout, head_partials, partials, xchg, next, hints, stream_);
}

int drgnn_net_train_step_cached(const drgnn_net_desc* net, const drgnn_head_desc* hd,
                                const drgnn_topology_cache* cache, const int32_t* ids, int64_t n_graphs,
                                int32_t max_nodes, int32_t max_edges, int32_t max_c0, int32_t* step2, float* pred,
                                float* readout, float* head_partials, float* partials, uint64_t* xchg,
                                const drgnn_step_hints* hints, void* stream_) {
    if (!cache || !ids || !cache->ws_i32 || !cache->x || n_graphs < 0 || n_graphs > cache->n_graphs) return DRGNN_E_ARG;
    if (hd && hd->train == 1) {
        if (!cache->y || cache->y_bytes != (hd->task == DRGNN_TASK_REG ? 4 : 8)) return DRGNN_E_ARG;
    }
    return train_step_impl(net, hd, cache->x, cache->y, step2, cache->ws_i32, cache->ws_f32, cache->n_nodes,
                           cache->n_edges, n_graphs, cache->n_graphs, ids, max_nodes, max_edges, max_c0, pred, readout,
                           head_partials, partials, xchg, nullptr, hints, stream_);
}

int drgnn_net_reduce_grads(const drgnn_net_desc* net, const float* partials, int64_t n_nodes,
                           int64_t n_graphs, drgnn_conv_grads* g_conv1, drgnn_conv_grads* g_conv2,
                           float* grad_x, void* stream_) {
    int rc = net_check(net);
    if (rc) return rc;
    if (!partials || !g_conv1 || !g_conv2) return DRGNN_E_ARG;
    ReduceArgs r;
    r.partials = partials; r.n_graphs = (int)n_graphs; r.n_branch = net->n_branch;
    r.n_feat = net->n_feat; r.n_partial = (int)net_partial_floats(net->n_feat); r.kind = net->kind;
    for (int b = 0; b < DRGNN_MAX_BRANCH; ++b) {
        r.lay1[b] = net->conv1[b]; r.lay2[b] = net->conv2[b];
        if (b < net->n_branch) { r.g1[b] = g_conv1[b]; r.g2[b] = g_conv2[b]; }
        else { r.g1[b] = drgnn_conv_grads{nullptr, nullptr, nullptr}; r.g2[b] = r.g1[b]; }
    }
    r.grad_x = grad_x; r.n_nodes = n_nodes;
    int64_t items = (int64_t)net->n_branch * r.n_partial;
    if (grad_x && net->n_branch > 1 && n_nodes * net->n_feat > items) items = n_nodes * net->n_feat;
#ifdef DRGNN_EMU
    for (int64_t i = 0; i < (int64_t)net->n_branch * r.n_partial; ++i) {
        const int br = (int)(i / r.n_partial), p = (int)(i % r.n_partial);
        if (reduce_live(r, p)) reduce_write(r, br, p, reduce_sum(r, br, p, 0, 1));
    }
    for (int64_t i = 0; grad_x && i < n_nodes * net->n_feat; ++i) reduce_grad_x(r, i);
    (void)stream_; (void)items;
#else
    const int reduce_blocks = net->n_branch * ((r.n_partial + 63) / 64);      // (blocks do not straddle branches)
    hipLaunchKernelGGL(k_reduce, dim3((unsigned)reduce_blocks), dim3(256), 0, (hipStream_t)stream_, r, items);
    HIP_TRY(hipGetLastError());
#endif
    return 0;
}

// ---- dense head + loss + optimiser ----------------------------------------------------------
int64_t drgnn_head_partial_elems(int32_t R, int32_t H, int32_t O) { return head_partial_floats(R, H, O); }
int64_t drgnn_head_num_slabs(int64_t n_graphs) { const int t = head_tile(n_graphs); return (n_graphs + t - 1) / t; }

static int head_check(const drgnn_head_desc* hd) {
    if (!hd || !hd->w1 || !hd->b1 || !hd->w2 || !hd->b2) return DRGNN_E_ARG;
    if (hd->R < 1 || hd->H < 1 || hd->O < 1 || hd->O > DRGNN_MAX_OUT) return DRGNN_E_WIDTH;
    if (4 * head_lds_words(hd->R, hd->H, hd->O, DRGNN_HEAD_TILE_LARGE) > DRGNN_LDS_LIMIT) return DRGNN_E_WIDTH;
    if (hd->task != DRGNN_TASK_REG && hd->task != DRGNN_TASK_CLASS) return DRGNN_E_ARG;
    if (!(hd->p_drop >= 0.0f && hd->p_drop < 1.0f)) return DRGNN_E_ARG;
    return 0;
}

int drgnn_head_step(const drgnn_head_desc* hd, const float* readout, const void* target,
                    int64_t n_graphs, const int32_t* step, float* pred, float* grad_readout,
                    float* partials, void* stream_) {
    int rc = head_check(hd);
    if (rc) return rc;
    if (!readout || !pred || n_graphs < 0) return DRGNN_E_ARG;
    if (hd->train && (!target || !grad_readout || !partials)) return DRGNN_E_ARG;
    if (n_graphs == 0) return 0;
    HeadArgs a;
    a.readout = readout;
    a.y_reg = (hd->task == DRGNN_TASK_REG) ? (const float*)target : nullptr;
    a.y_cls = (hd->task == DRGNN_TASK_CLASS) ? (const int64_t*)target : nullptr;
    a.class_w = hd->class_w;
    a.w1 = hd->w1; a.b1 = hd->b1; a.w2 = hd->w2; a.b2 = hd->b2;
    a.step = step; a.pred = pred; a.grad_readout = hd->train ? grad_readout : nullptr;
    a.partials = partials;
    a.B = (int)n_graphs; a.R = hd->R; a.H = hd->H; a.O = hd->O;
    a.task = hd->task; a.train = hd->train; a.p_drop = hd->p_drop; a.seed = hd->seed; a.sigmoid = hd->transform_sigmoid;
    a.T = head_tile(n_graphs);
    const int blocks = (int)((n_graphs + a.T - 1) / a.T);
    const int64_t lds = 4 * head_lds_words(hd->R, hd->H, hd->O, a.T);
#ifdef DRGNN_EMU
    std::vector<float> buf((size_t)(lds / 4) + 16);
    for (int b = 0; b < blocks; ++b) head_block(a, b, buf.data());
    (void)stream_;
#else
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void*)k_head, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_head, dim3((unsigned)blocks), dim3(DRGNN_NTHREADS), (size_t)lds, (hipStream_t)stream_, a);
    HIP_TRY(hipGetLastError());
#endif
    return 0;
}

int drgnn_head_reduce(const float* partials, int64_t n_graphs, int32_t R, int32_t H, int32_t O,
                      float* grad_block, float* loss, int32_t* step, void* stream_) {
    if (!partials || !grad_block) return DRGNN_E_ARG;
    HeadReduceArgs a;
    a.partials = partials;
    a.n_wg = (int)((n_graphs + head_tile(n_graphs) - 1) / head_tile(n_graphs));
    a.P = (int)head_partial_floats(R, H, O);
    a.grad = grad_block; a.loss = loss; a.step = step;
#ifdef DRGNN_EMU
    for (int i = 0; i < a.P - 1; ++i) head_reduce_item(a, i);
    (void)stream_;
#else
    hipLaunchKernelGGL(k_head_reduce, dim3((unsigned)((a.P - 1 + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, a);
    HIP_TRY(hipGetLastError());
#endif
    return 0;
}

int drgnn_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                    const int32_t* step, int64_t n, float lr, float beta1, float beta2, float eps,
                    float weight_decay, void* stream_) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !step || n < 0) return DRGNN_E_ARG;
    if (n == 0) return 0;
    AdamArgs a;
    a.param = param; a.grad = grad; a.exp_avg = exp_avg; a.exp_avg_sq = exp_avg_sq; a.step = step; a.n = n;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
#ifdef DRGNN_EMU
    for (int64_t i = 0; i < n; ++i) adam_item(a, i);
    (void)stream_;
#else
    hipLaunchKernelGGL(k_adam, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, a);
    HIP_TRY(hipGetLastError());
#endif
    return 0;
}

}  // extern "C"

extern "C" {

int64_t drgnn_p2p_bytes(int64_t n_floats) { return n_floats < 0 ? (int64_t)DRGNN_E_ARG : p2p_bytes(n_floats); }

int drgnn_p2p_alloc(int64_t bytes, void** dev_ptr, void* handle) {
    if (bytes <= 0 || !dev_ptr) return DRGNN_E_ARG;
#ifdef DRGNN_EMU
    *dev_ptr = calloc(1, (size_t)bytes);
    if (handle) { memset(handle, 0, 64); memcpy(handle, dev_ptr, sizeof(void*)); }     // same-process "handle"
    return *dev_ptr ? 0 : DRGNN_E_CAPACITY;
#else
    // fine-grained: coherent across devices WITHIN a kernel (coarse-grained memory is only coherent at launch boundaries)
    HIP_TRY(hipExtMallocWithFlags(dev_ptr, (size_t)bytes, hipDeviceMallocFinegrained));
    HIP_TRY(hipMemset(*dev_ptr, 0, (size_t)bytes));
    HIP_TRY(hipDeviceSynchronize());
    if (handle) {
        static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
        HIP_TRY(hipIpcGetMemHandle((hipIpcMemHandle_t*)handle, *dev_ptr));
    }
    return 0;
#endif
}

int drgnn_p2p_open(const void* handle, void** dev_ptr) {
    if (!handle || !dev_ptr) return DRGNN_E_ARG;
#ifdef DRGNN_EMU
    memcpy(dev_ptr, handle, sizeof(void*));
    return 0;
#else
    const hipIpcMemHandle_t h = *(const hipIpcMemHandle_t*)handle;
    HIP_TRY(hipIpcOpenMemHandle(dev_ptr, h, hipIpcMemLazyEnablePeerAccess));
    return 0;
#endif
}

int drgnn_p2p_close(void* dev_ptr) {
#ifdef DRGNN_EMU
    (void)dev_ptr;
    return 0;
#else
    if (!dev_ptr) return DRGNN_E_ARG;
    HIP_TRY(hipIpcCloseMemHandle(dev_ptr));
    return 0;
#endif
}

int drgnn_p2p_free(void* dev_ptr) {
#ifdef DRGNN_EMU
    free(dev_ptr);
    return 0;
#else
    if (!dev_ptr) return DRGNN_E_ARG;
    HIP_TRY(hipFree(dev_ptr));
    return 0;
#endif
}

int drgnn_allreduce_oneshot(float* grad, int64_t n, void* const* peer_bufs, int32_t world, int32_t rank, float weight,
                            uint32_t* seq, int32_t* status, int32_t part, void* stream_) {
    if (!grad || n < 0 || !peer_bufs || world < 1 || world > DRGNN_P2P_MAX || rank < 0 || rank >= world || !seq || !status ||
        part < 0 || part > 2)
        return DRGNN_E_ARG;
    if (n == 0) return 0;
    P2PArgs a;
    a.grad = grad; a.n = n; a.world = world; a.rank = rank; a.weight = weight; a.seq = seq; a.status = status; a.part = part;
    for (int r = 0; r < DRGNN_P2P_MAX; ++r) a.peer[r] = (r < world) ? (float*)peer_bufs[r] : nullptr;
    for (int r = 0; r < world; ++r) if (!a.peer[r]) return DRGNN_E_ARG;
#ifdef DRGNN_EMU
    for (int j = 0; j < DRGNN_P2P_WGS; ++j) p2p_block(a, j);
    (void)stream_;
#else
    hipLaunchKernelGGL(k_allreduce_oneshot, dim3(DRGNN_P2P_WGS), dim3(DRGNN_P2P_THREADS), 0, (hipStream_t)stream_, a);
    HIP_TRY(hipGetLastError());
#endif
    return 0;
}

}  // extern "C"

// `readout` null: legacy head slabs [head_slabs][head_partial_floats] (dW_fc1 inside), Adam reads step[0].
// `readout` given (fused step): compact slabs [n_graphs][head_compact_floats] + readout [n_graphs][R];
// `step` is then the 2-word counter of drgnn_net_train_step: Adam reads step[1], step[0] is committed.
// slabs_per_graph: conv slabs per graph (0: n_branch; 2 for the split layout of a single-branch net, drgnn_step2.h)
static int update_impl(int32_t slabs_per_graph, const drgnn_net_desc* net, const float* conv_partials, int64_t n_graphs,
                       drgnn_conv_grads* g_conv1, drgnn_conv_grads* g_conv2, const float* head_partials,
                       int64_t head_slabs, const float* readout, int32_t R, int32_t H, int32_t O,
                       int64_t head_offset, float* flat_param,
                       float* flat_grad, float* exp_avg, float* exp_avg_sq, int64_t n_param,
                       int32_t* step, float* loss, float lr, float beta1, float beta2, float eps,
                       int32_t apply_adam, void* stream_, float* loss2 = nullptr) {
    int rc = net_check(net);
    if (rc) return rc;
    if (!conv_partials || !g_conv1 || !g_conv2 || !head_partials || !flat_grad) return DRGNN_E_ARG;
    if (apply_adam && (!flat_param || !exp_avg || !exp_avg_sq || !step)) return DRGNN_E_ARG;
    UpdateArgs u;
    ReduceArgs& r = u.r;
    if (slabs_per_graph != 0 && slabs_per_graph != net->n_branch && !(net->n_branch == 1 && slabs_per_graph == 2)) return DRGNN_E_ARG;
    // (a single-branch net's two half-graph slabs are two more "graphs" to the fixed-order sum)
    const int64_t conv_rows = (net->n_branch == 1 && slabs_per_graph == 2) ? 2 * n_graphs : n_graphs;
    r.partials = conv_partials; r.n_graphs = (int)conv_rows; r.n_branch = net->n_branch;
    r.n_feat = net->n_feat; r.n_partial = (int)net_partial_floats(net->n_feat); r.kind = net->kind;
    for (int b = 0; b < DRGNN_MAX_BRANCH; ++b) {
        r.lay1[b] = net->conv1[b]; r.lay2[b] = net->conv2[b];
        if (b < net->n_branch) { r.g1[b] = g_conv1[b]; r.g2[b] = g_conv2[b]; }
        else { r.g1[b] = drgnn_conv_grads{nullptr, nullptr, nullptr}; r.g2[b] = r.g1[b]; }
    }
    r.grad_x = nullptr; r.n_nodes = 0;
    u.h.partials = head_partials;
    u.h.n_wg = (int)head_slabs;
    u.h.P = (int)(readout ? head_compact_floats(R, H, O) : head_partial_floats(R, H, O));
    u.h.grad = flat_grad + head_offset; u.h.loss = loss; u.h.step = nullptr;
    u.readout = readout; u.hR = R; u.hH = H;
    u.step2 = readout ? step : nullptr;
    u.loss2 = loss ? loss2 : nullptr;
    u.ad.param = flat_param; u.ad.grad = flat_grad; u.ad.exp_avg = exp_avg; u.ad.exp_avg_sq = exp_avg_sq;
    u.ad.step = (readout && step) ? step + 1 : step; u.ad.n = n_param;
    u.ad.lr = lr; u.ad.beta1 = beta1; u.ad.beta2 = beta2; u.ad.eps = eps; u.ad.weight_decay = 0.0f;
    u.apply_adam = apply_adam ? 1 : 0;
    const int64_t pitems = (int64_t)net->n_branch * r.n_partial;
    u.blocks_per_branch = (r.n_partial + 63) / 64;
    u.conv_blocks = net->n_branch * u.blocks_per_branch;
    (void)pitems;
    const int head_items = (readout ? H * R : 0) + u.h.P - 1;
    const int head_blocks = (head_items + 63) / 64;
#ifdef DRGNN_EMU
    for (int64_t i = 0; i < pitems; ++i) {
        const int br = (int)(i / r.n_partial), p = (int)(i % r.n_partial);
        if (!reduce_live(r, p)) continue;
        float* d = reduce_dst(r, br, p);
        if (d) update_store(u, d, reduce_sum(r, br, p, 0, 1));
    }
    for (int i = 0; i < head_items; ++i) update_head_store(u, i, update_head_sum(u, i, 0, 1));
    if (u.step2) u.step2[0] = u.step2[1];
    (void)stream_; (void)head_blocks;
#else
    hipLaunchKernelGGL(k_update, dim3((unsigned)(u.conv_blocks + head_blocks)), dim3(DRGNN_UPDATE_THREADS), 0, (hipStream_t)stream_, u);
    HIP_TRY(hipGetLastError());
#endif
    return 0;
}

extern "C" {

int drgnn_train_update(const drgnn_net_desc* net, const float* conv_partials, int64_t n_graphs,
                       drgnn_conv_grads* g_conv1, drgnn_conv_grads* g_conv2, const float* head_partials,
                       int64_t head_slabs, int32_t R, int32_t H, int32_t O, int64_t head_offset,
                       float* flat_param,
                       float* flat_grad, float* exp_avg, float* exp_avg_sq, int64_t n_param,
                       const int32_t* step, float* loss, float lr, float beta1, float beta2, float eps,
                       int32_t apply_adam, void* stream_) {
    if (!head_partials) return DRGNN_E_ARG;
    return update_impl(0, net, conv_partials, n_graphs, g_conv1, g_conv2, head_partials, head_slabs, nullptr, R, H, O,
                       head_offset, flat_param, flat_grad, exp_avg, exp_avg_sq, n_param,
                       const_cast<int32_t*>(step), loss, lr, beta1, beta2, eps, apply_adam, stream_);
}

int drgnn_step_update(const drgnn_net_desc* net, const float* conv_partials, int64_t n_graphs,
                      drgnn_conv_grads* g_conv1, drgnn_conv_grads* g_conv2, const float* head_partials,
                      const float* readout, int32_t R, int32_t H, int32_t O, int64_t head_offset,
                      float* flat_param, float* flat_grad, float* exp_avg, float* exp_avg_sq, int64_t n_param,
                      int32_t* step2, float* loss, float lr, float beta1, float beta2, float eps,
                      int32_t apply_adam, int32_t slabs_per_graph, void* stream_) {
    if (!head_partials || !readout || !step2) return DRGNN_E_ARG;
    return update_impl(slabs_per_graph, net, conv_partials, n_graphs, g_conv1, g_conv2, head_partials, n_graphs, readout, R, H, O,
                       head_offset, flat_param, flat_grad, exp_avg, exp_avg_sq, n_param, step2, loss, lr, beta1,
                       beta2, eps, apply_adam, stream_);
}

int drgnn_step_gradients(const drgnn_net_desc* net, const float* conv_partials, int64_t n_graphs,
                         drgnn_conv_grads* g_conv1, drgnn_conv_grads* g_conv2, const float* head_partials,
                         const float* readout, int32_t R, int32_t H, int32_t O, float* head_grad,
                         const float* graph_weight, float* const* zero_ptr, const int64_t* zero_len, int32_t n_zero,
                         int32_t* step2, int32_t slabs_per_graph, void* stream_) {
    int rc = net_check(net);
    if (rc) return rc;
    if (!conv_partials || !g_conv1 || !g_conv2 || !head_partials || !readout || !head_grad || n_graphs < 0) return DRGNN_E_ARG;
    if (n_zero < 0 || n_zero > DRGNN_ZERO_RANGES || (n_zero > 0 && (!zero_ptr || !zero_len))) return DRGNN_E_ARG;
    if (slabs_per_graph != 0 && slabs_per_graph != net->n_branch && !(net->n_branch == 1 && slabs_per_graph == 2)) return DRGNN_E_ARG;
    if (R != DRGNN_H2 * net->n_branch || H < 1 || H > 512 || O < 1 || O > DRGNN_MAX_OUT) return DRGNN_E_WIDTH;
    GradArgs ga;
    memset(&ga, 0, sizeof(ga));
    UpdateArgs& u = ga.u;
    ReduceArgs& r = u.r;
    const bool split = net->n_branch == 1 && slabs_per_graph == 2;
    r.partials = conv_partials; r.n_graphs = (int)(split ? 2 * n_graphs : n_graphs); r.n_branch = net->n_branch;
    r.n_feat = net->n_feat; r.n_partial = (int)net_partial_floats(net->n_feat); r.kind = net->kind;
    for (int b = 0; b < DRGNN_MAX_BRANCH; ++b) {
        r.lay1[b] = net->conv1[b]; r.lay2[b] = net->conv2[b];
        if (b < net->n_branch) { r.g1[b] = g_conv1[b]; r.g2[b] = g_conv2[b]; }
        else { r.g1[b] = drgnn_conv_grads{nullptr, nullptr, nullptr}; r.g2[b] = r.g1[b]; }
    }
    r.grad_x = nullptr; r.n_nodes = 0;
    u.h.partials = head_partials; u.h.n_wg = (int)n_graphs; u.h.P = (int)head_compact_floats(R, H, O);
    u.h.grad = head_grad; u.h.loss = nullptr; u.h.step = nullptr;
    u.readout = readout; u.hR = R; u.hH = H; u.step2 = step2; u.apply_adam = 0;
    u.blocks_per_branch = (r.n_partial + 63) / 64;
    u.conv_blocks = net->n_branch * u.blocks_per_branch;
    ga.graph_weight = graph_weight; ga.wshift = split ? 1 : 0; ga.n_zero = n_zero;
    for (int i = 0; i < n_zero; ++i) {
        if (!zero_ptr[i] || zero_len[i] < 0) return DRGNN_E_ARG;
        ga.zero_ptr[i] = zero_ptr[i]; ga.zero_len[i] = zero_len[i];
    }
    const int head_items = H * R + u.h.P - 2;             // the gradient block (the loss / weight slots are not gradients)
    const int head_blocks = (head_items + 63) / 64;
#ifdef DRGNN_EMU
    for (int64_t i = 0; i < (int64_t)net->n_branch * r.n_partial; ++i) {
        const int br = (int)(i / r.n_partial), p = (int)(i % r.n_partial);
        if (!reduce_live(r, p)) continue;
        float* d = reduce_dst(r, br, p);
        if (d) *d = graph_weight ? reduce_sum_w(r, br, p, 0, 1, graph_weight, ga.wshift) : reduce_sum(r, br, p, 0, 1);
    }
    for (int i = 0; i < head_items; ++i)
        head_grad[i] = graph_weight ? update_head_sum_w(u, i, 0, 1, graph_weight) : update_head_sum(u, i, 0, 1);
    for (int i = 0; i < n_zero; ++i) memset(ga.zero_ptr[i], 0, sizeof(float) * (size_t)ga.zero_len[i]);
    if (step2) step2[0] = step2[1];
    (void)stream_; (void)head_blocks;
#else
    hipLaunchKernelGGL(k_gradients, dim3((unsigned)(u.conv_blocks + head_blocks + 1)), dim3(256), 0, (hipStream_t)stream_, ga);
    HIP_TRY(hipGetLastError());
#endif
    return 0;
}

// ---- stand-alone layers / pooling functions ---------------------------------------------------
#ifdef DRGNN_EMU
#define DRGNN_GRID_ITEMS(kern, item_fn, n_items, stream, ...) \
    do { for (int64_t it_ = 0; it_ < (int64_t)(n_items); ++it_) item_fn(__VA_ARGS__, it_); (void)stream; } while (0)
#define DRGNN_GRID_BLOCKS(kern, block_fn, n_blocks, stream, ...) \
    do { for (int b_ = 0; b_ < (int)(n_blocks); ++b_) block_fn(__VA_ARGS__, b_); (void)stream; } while (0)
#else
#define DRGNN_GRID_ITEMS(kern, item_fn, n_items, stream, ...)                                              \
    do { if ((n_items) > 0) hipLaunchKernelGGL(kern, dim3((unsigned)(((n_items) + 255) / 256)), dim3(256), \
                                               0, (hipStream_t)(stream), __VA_ARGS__); } while (0)
#define DRGNN_GRID_BLOCKS(kern, block_fn, n_blocks, stream, ...)                                          \
    do { if ((n_blocks) > 0) hipLaunchKernelGGL(kern, dim3((unsigned)(n_blocks)), dim3(DRGNN_NTHREADS),   \
                                                0, (hipStream_t)(stream), __VA_ARGS__); } while (0)
#endif

static int conv_layer_fill(ConvLayerArgs& a, int32_t kind, const float* x, int64_t n_nodes, int32_t F,
                           int32_t H, const drgnn_conv_params* p, const int32_t* ws_i32,
                           const float* ws_f32, int64_t n_edges) {
    if (kind < DRGNN_GINET || kind > DRGNN_FOUT || !x || !p || !p->w_nbr || !ws_i32) return DRGNN_E_ARG;
    if (kind != DRGNN_GINET && (!p->w_self || !p->bias)) return DRGNN_E_ARG;
    if (kind == DRGNN_SGAT && !ws_f32) return DRGNN_E_ARG;
    if (F < 1 || H < 1 || H > DRGNN_LAYER_MAXH || n_nodes >= INT32_MAX) return DRGNN_E_WIDTH;
    TopoLayout lay;
    topo_layout(n_nodes, n_edges, 1, &lay);
    TopoView tv = topo_view(const_cast<int32_t*>(ws_i32), const_cast<float*>(ws_f32), lay);
    a.kind = kind; a.x = x; a.F = F; a.H = H; a.p = *p; a.N = (int)n_nodes;
    a.rowptr = tv.p[DRGNN_TI_ROWPTR0]; a.col = tv.p[DRGNN_TI_COL0]; a.w = tv.w0;
    a.colptr = tv.p[DRGNN_TI_COLPTR0]; a.ridx = tv.p[DRGNN_TI_ROWIDX0]; a.tslot = tv.p[DRGNN_TI_TSLOT0];
    a.u = nullptr; a.out = nullptr; a.grad_out = nullptr; a.partials = nullptr; a.grad_x = nullptr;
    return 0;
}

int64_t drgnn_conv_layer_slabs(int64_t n_nodes) { return (n_nodes + DRGNN_LAYER_ROWS - 1) / DRGNN_LAYER_ROWS; }
int64_t drgnn_conv_layer_partial_elems(int32_t kind, int32_t F, int32_t H) { return conv_partial_floats(kind, F, H); }

int drgnn_conv_layer_forward(int32_t kind, const float* x, int64_t n_nodes, int32_t F, int32_t H,
                             const drgnn_conv_params* p, const int32_t* ws_i32, const float* ws_f32,
                             int64_t n_edges, float* u, float* out, void* stream) {
    ConvLayerArgs a;
    int rc = conv_layer_fill(a, kind, x, n_nodes, F, H, p, ws_i32, ws_f32, n_edges);
    if (rc) return rc;
    if (!u || !out) return DRGNN_E_ARG;
    a.u = u; a.out = out;
    const int64_t slabs = drgnn_conv_layer_slabs(n_nodes);
    DRGNN_GRID_BLOCKS(k_conv_gemm, conv_gemm_block, slabs, stream, a);
    DRGNN_GRID_ITEMS(k_conv_aggregate, conv_aggregate_item, n_nodes * H, stream, a);
#ifndef DRGNN_EMU
    HIP_TRY(hipGetLastError());
#endif
    return 0;
}

int drgnn_conv_layer_backward(int32_t kind, const float* x, int64_t n_nodes, int32_t F, int32_t H,
                              const drgnn_conv_params* p, const int32_t* ws_i32, const float* ws_f32,
                              int64_t n_edges, const float* grad_out, float* du, float* partials,
                              const drgnn_conv_grads* g, float* grad_x, void* stream) {
    ConvLayerArgs a;
    int rc = conv_layer_fill(a, kind, x, n_nodes, F, H, p, ws_i32, ws_f32, n_edges);
    if (rc) return rc;
    if (!grad_out || !du || !partials || !g) return DRGNN_E_ARG;
    a.u = du; a.grad_out = grad_out; a.partials = partials; a.grad_x = grad_x;
    const int64_t slabs = drgnn_conv_layer_slabs(n_nodes);
    DRGNN_GRID_ITEMS(k_conv_bwd_du, conv_bwd_du_item, n_nodes * H, stream, a);
    DRGNN_GRID_BLOCKS(k_conv_bwd_dw, conv_bwd_dw_block, slabs, stream, a);
    ConvReduceArgs r;
    r.partials = partials; r.n_wg = (int)slabs; r.kind = kind; r.F = F; r.H = H; r.lay = *p; r.g = *g;
    const int64_t P = conv_partial_floats(kind, F, H);
#ifdef DRGNN_EMU
    for (int i = 0; i < (int)P; ++i) conv_reduce_item(r, i);
#else
    hipLaunchKernelGGL(k_conv_reduce, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, r);
#endif
    if (grad_x) DRGNN_GRID_ITEMS(k_conv_bwd_dx, conv_bwd_dx_item, n_nodes * F, stream, a);
#ifndef DRGNN_EMU
    HIP_TRY(hipGetLastError());
#endif
    return 0;
}

int drgnn_segpool_forward(const int32_t* ws_i32, int64_t n_nodes, int64_t n_edges, int64_t n_graphs,
                          const float* x, int32_t H, int32_t op, float* out, int64_t* arg, void* stream) {
    if (!ws_i32 || !x || !out || H < 1 || (op != 0 && op != 1)) return DRGNN_E_ARG;
    TopoLayout lay;
    topo_layout(n_nodes, n_edges, n_graphs, &lay);
    SegPoolArgs a;
    a.tv = topo_view(const_cast<int32_t*>(ws_i32), nullptr, lay);
    a.x = x; a.H = H; a.n_graphs = (int)n_graphs; a.op = op; a.out = out; a.arg = arg;
    a.grad_out = nullptr; a.grad_x = nullptr; a.arg_in = nullptr;
    DRGNN_GRID_BLOCKS(k_segpool_fwd, segpool_fwd_block, n_graphs, stream, a);
#ifndef DRGNN_EMU
    HIP_TRY(hipGetLastError());
#endif
    return 0;
}

int drgnn_segmax_backward(const float* grad_out, const int64_t* arg, int64_t n_clusters, int32_t H,
                          int64_t n_nodes, float* grad_x, void* stream) {
    if (!grad_out || !arg || !grad_x) return DRGNN_E_ARG;
    SegPoolArgs a;
    a.H = H; a.grad_out = grad_out; a.arg_in = arg; a.grad_x = grad_x;
    a.x = nullptr; a.out = nullptr; a.arg = nullptr; a.n_graphs = 0; a.op = 0;
    const int64_t n_items = n_clusters * H;
#ifdef DRGNN_EMU
    for (int64_t i = 0; i < n_items; ++i) segmax_bwd_item(a, i, n_items, n_nodes);
    (void)stream;
#else
    if (n_items > 0)
        hipLaunchKernelGGL(k_segmax_bwd, dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, n_items, n_nodes);
    HIP_TRY(hipGetLastError());
#endif
    return 0;
}

int drgnn_pooled_edges_export(const int32_t* ws_i32, const float* ws_f32, int64_t n_nodes, int64_t n_edges,
                              int64_t n_graphs, int64_t e1_total, int64_t* edge_index, float* edge_attr,
                              void* stream) {
    if (!ws_i32 || (e1_total > 0 && !edge_index)) return DRGNN_E_ARG;
    TopoLayout lay;
    topo_layout(n_nodes, n_edges, n_graphs, &lay);
    EdgeExportArgs a;
    a.tv = topo_view(const_cast<int32_t*>(ws_i32), const_cast<float*>(ws_f32), lay);
    a.n_graphs = (int)n_graphs; a.edge_index = edge_index; a.edge_attr = edge_attr; a.e1_total = e1_total;
    DRGNN_GRID_BLOCKS(k_edge_export, edge_export_block, n_graphs, stream, a);
#ifndef DRGNN_EMU
    HIP_TRY(hipGetLastError());
#endif
    return 0;
}

int drgnn_cluster_offset(int64_t* cluster, const int32_t* node_ptr, int64_t n_graphs, int64_t* scratch,
                         void* stream) {
    if (!cluster || !node_ptr || !scratch || n_graphs < 0) return DRGNN_E_ARG;
    if (n_graphs == 0) return 0;
    ClusterOffsetArgs a;
    a.cluster = cluster; a.nptr = node_ptr; a.n_graphs = (int)n_graphs; a.maxes = (long long*)scratch;
#ifdef DRGNN_EMU
    long long mm[2];
    for (int g = 0; g < n_graphs; ++g) cluster_max_block(a, g, mm);
    cluster_scan_single(a);
    for (int g = 0; g < n_graphs; ++g) cluster_add_block(a, g);
    (void)stream;
#else
    hipLaunchKernelGGL(k_cluster_max, dim3((unsigned)n_graphs), dim3(DRGNN_NTHREADS), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(k_cluster_scan, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(k_cluster_add, dim3((unsigned)n_graphs), dim3(DRGNN_NTHREADS), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
#endif
    return 0;
}

// ---- graclus ----------------------------------------------------------------------------------------
int drgnn_graclus(const int32_t* ws_i32, int64_t n_nodes, int64_t n_edges, int64_t n_graphs, int32_t max_nodes,
                  int32_t max_edges, const float* weight, const int64_t* perm, int64_t* cluster, void* stream) {
    if (!ws_i32 || n_nodes < 0 || n_edges < 0 || n_graphs < 0 || max_nodes < 0 || max_edges < 0) return DRGNN_E_ARG;
    if (n_nodes > 0 && !cluster) return DRGNN_E_ARG;
    if (n_graphs == 0) return 0;
    TopoLayout lay;
    topo_layout(n_nodes, n_edges, n_graphs, &lay);
    GraclusArgs a;
    a.tv = topo_view(const_cast<int32_t*>(ws_i32), nullptr, lay);
    a.n_graphs = (int)n_graphs; a.weight = weight; a.perm = perm; a.cluster = cluster;
    a.capN = max_nodes > 0 ? max_nodes : 1; a.capE = max_edges > 0 ? max_edges : 1;
    const int64_t words = (int64_t)(a.capN + 1) + 2 * (int64_t)a.capE + a.capN;
    if (words * 4 > DRGNN_LDS_LIMIT) return DRGNN_E_CAPACITY;
#ifdef DRGNN_EMU
    std::vector<int> buf((size_t)words + 16);
    for (int g = 0; g < n_graphs; ++g) graclus_block(a, g, buf.data());
    (void)stream;
#else
    if (words * 4 > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void*)k_graclus, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(words * 4)));
    hipLaunchKernelGGL(k_graclus, dim3((unsigned)n_graphs), dim3(DRGNN_NTHREADS), (size_t)(words * 4), (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
#endif
    return 0;
}

// ---- offline clustering ---------------------------------------------------------------------------
int drgnn_mcl(const int64_t* edge_index, int64_t n_edges, const int32_t* node_ptr, const int32_t* edge_ptr,
              const int64_t* mat_ptr, int64_t n_graphs, double* mat_scratch, int32_t* int_scratch,
              int64_t* labels, int32_t* info, void* stream) {
    if (!node_ptr || !edge_ptr || !mat_ptr || !mat_scratch || !int_scratch || !labels || !info || n_graphs < 0)
        return DRGNN_E_ARG;
    if (n_edges > 0 && !edge_index) return DRGNN_E_ARG;
    if (n_graphs == 0) return 0;
    MclArgs a;
    a.edge_index = edge_index; a.n_edges = n_edges; a.node_ptr = node_ptr; a.edge_ptr = edge_ptr;
    a.mat_ptr = mat_ptr; a.n_graphs = (int)n_graphs; a.mat = mat_scratch; a.iscr = int_scratch;
    a.labels = labels; a.info = info; a.iterations = 100; a.prune_threshold = 1e-3;
#ifdef DRGNN_EMU
    std::vector<double> red(DRGNN_NTHREADS);
    int flag[2];
    for (int g = 0; g < n_graphs; ++g) mcl_graph(a, g, flag, red.data());
    (void)stream;
#else
    hipLaunchKernelGGL(k_mcl, dim3((unsigned)n_graphs), dim3(DRGNN_NTHREADS), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
#endif
    return 0;
}

// ---- mini-batch assembly from the resident graph set -----------------------------------------------
int drgnn_collate(const drgnn_graph_set* set, const int32_t* ids, int64_t n_graphs, int64_t n_nodes,
                  int64_t n_edges, float* x, int64_t* edge_index, float* edge_attr, int64_t* batch,
                  int64_t* cluster0, int64_t* cluster1, void* y, int32_t* node_ptr, int32_t* edge_ptr,
                  int32_t* c1_ptr, void* stream) {
    if (!set || !ids || n_graphs < 0 || n_nodes < 0 || n_edges < 0 || !node_ptr || !edge_ptr) return DRGNN_E_ARG;
    if (!set->node_ptr || !set->edge_ptr || set->n_feat <= 0 || set->n_graphs < 0) return DRGNN_E_ARG;
    if (n_nodes > 0 && (!x || !batch || !set->x)) return DRGNN_E_ARG;
    if (n_edges > 0 && (!edge_index || !set->edge_index)) return DRGNN_E_ARG;
    if ((edge_attr && !set->edge_attr) || (cluster0 && !set->cluster0)) return DRGNN_E_ARG;
    if (cluster1 && (!set->cluster1 || !set->c1_ptr || !c1_ptr)) return DRGNN_E_ARG;
    if (y && (!set->y || (set->y_bytes != 4 && set->y_bytes != 8))) return DRGNN_E_ARG;
    if (n_nodes > 0x7fffffffLL || n_edges > 0x7fffffffLL) return DRGNN_E_CAPACITY;   // int32 offset tables
    if (n_graphs == 0) return 0;
    CollateArgs a;
    a.set = *set; a.ids = ids; a.n_graphs = (int)n_graphs; a.n_edges = n_edges;
    a.x = x; a.edge_index = edge_index; a.edge_attr = edge_attr; a.batch = batch;
    a.cluster0 = cluster0; a.cluster1 = cluster1; a.y = y;
    a.node_ptr = node_ptr; a.edge_ptr = edge_ptr; a.c1_ptr = set->c1_ptr ? c1_ptr : nullptr;
#ifdef DRGNN_EMU
    int sh[4];
    for (int g = 0; g < n_graphs; ++g) collate_block(a, g, sh);
    (void)stream;
#else
    hipLaunchKernelGGL(k_collate, dim3((unsigned)n_graphs), dim3(DRGNN_NTHREADS), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
#endif
    return 0;
}

// ---- one training epoch, driven from here ---------------------------------------------------------
namespace {
struct EpochBatch { int64_t first, B, N, E, C; int maxN, maxE, maxC; };
struct EpochSlot { float* x; void* y; int32_t* ws_i32; float* ws_f32; float* tiles; };
struct EpochCarve {
    EpochSlot slot[2];
    int32_t* ptrs;                 // [n_batches][3][batch_size + 1]
    float* readout; float* partials; float* head_partials; uint64_t* xchg; int64_t xchg_bytes;
    int64_t bytes;
};
int epoch_batch(const drgnn_epoch_plan* p, int64_t k, EpochBatch* b) {
    const int64_t G = p->set->n_graphs;
    b->first = k * p->batch_size;
    b->B = p->n_ids - b->first < p->batch_size ? p->n_ids - b->first : p->batch_size;
    b->N = b->E = b->C = 0; b->maxN = b->maxE = b->maxC = 0;
    for (int64_t q = 0; q < b->B; ++q) {
        const int64_t id = p->host_ids[b->first + q];
        if (id < 0 || id >= G) return DRGNN_E_ARG;
        const int64_t n = p->host_node_ptr[id + 1] - p->host_node_ptr[id];
        const int64_t e = p->host_edge_ptr[id + 1] - p->host_edge_ptr[id];
        const int64_t c = p->host_c1_ptr[id + 1] - p->host_c1_ptr[id];
        if (n < 0 || e < 0 || c < 0) return DRGNN_E_ARG;
        b->N += n; b->E += e; b->C += c;
        if (n > b->maxN) b->maxN = (int)n;
        if (e > b->maxE) b->maxE = (int)e;
        if (c > b->maxC) b->maxC = (int)c;
    }
    return 0;
}
int epoch_check(const drgnn_epoch_plan* p) {
    if (!p || !p->set || !p->host_node_ptr || !p->host_edge_ptr || !p->ids || !p->host_ids || p->n_ids < 0 ||
        p->batch_size < 1 || !p->net || !p->head || !p->step2)
        return DRGNN_E_ARG;
    if (!p->inference && (!p->g_conv1 || !p->g_conv2 || !p->flat_param || !p->flat_grad || !p->exp_avg || !p->exp_avg_sq))
        return DRGNN_E_ARG;
    const drgnn_graph_set* gs = p->set;
    if (p->cache) {      // cached topology: the set is only consulted for sizes (host tables)
        const drgnn_topology_cache* tc = p->cache;
        if (!tc->ws_i32 || !tc->x || tc->n_graphs != gs->n_graphs || !p->host_c1_ptr) return DRGNN_E_ARG;
        if (p->need_weights && !tc->ws_f32) return DRGNN_E_ARG;
        if (!p->inference && (!tc->y || tc->y_bytes != (p->head->task == DRGNN_TASK_REG ? 4 : 8))) return DRGNN_E_ARG;
        if (gs->n_feat != p->net->n_feat) return DRGNN_E_WIDTH;
        if (p->batch_size > 4096) return DRGNN_E_CAPACITY;
        return 0;
    }
    if (!gs->node_ptr || !gs->edge_ptr || !gs->x || !gs->cluster0 || !gs->cluster1 || !gs->c1_ptr || !p->host_c1_ptr)
        return DRGNN_E_ARG;
    if (!p->inference && !gs->y) return DRGNN_E_ARG;
    if (gs->n_edges > 0 && !gs->edge_index) return DRGNN_E_ARG;
    if (p->need_weights && !gs->edge_attr) return DRGNN_E_ARG;
    if (gs->n_feat != p->net->n_feat) return DRGNN_E_WIDTH;
    if (!p->inference && gs->y_bytes != (p->head->task == DRGNN_TASK_REG ? 4 : 8)) return DRGNN_E_ARG;
    if (p->batch_size > 4096) return DRGNN_E_CAPACITY;
    return 0;
}
// the plan of mini-batch b's step launch on a workspace with `topo_flags` (co = graphs of the topology co-built by the same
// launch); family NONE: a graph does not fit the fused kernels
drgnn_step_plan epoch_plan_of(const drgnn_epoch_plan* p, const EpochBatch& b, int64_t co, int32_t topo_flags) {
    const drgnn_head_desc* hd = p->head;
    drgnn_step_plan pl;
    memset(&pl, 0, sizeof(pl));
    pl.kind = p->net->kind; pl.n_feat = p->net->n_feat; pl.max_nodes = b.maxN; pl.max_edges = b.maxE; pl.max_c0 = b.maxC;
    pl.R = hd->R; pl.H = hd->H; pl.O = hd->O; pl.n_graphs = b.B; pl.co_built_graphs = co;
    pl.train = p->inference ? 0 : 1; pl.topo_flags = topo_flags;
    if (p->step_overrides) {
        const drgnn_step_plan* o = p->step_overrides;
        pl.force_wgs = o->force_wgs; pl.no_class = o->no_class; pl.no_aggregate = o->no_aggregate; pl.no_split = o->no_split;
        pl.no_paired = o->no_paired;
    }
    drgnn_net_step_plan(&pl);
    return pl;
}
// what the workspace of mini-batch b holds when its step is launched: the cache's flags, or what the loop asks its own builder
// for -- the hierarchical order, the aggregation tiles and nothing else (DRGNN_TOPO_LEAN) when the step that consumes it is one
// of the aggregation-first kernels, else the plain build (+ the order for the single-branch nets)
int32_t epoch_topo_flags(const drgnn_epoch_plan* p, const EpochBatch& b, int64_t co, const float* slot_x) {
    if (p->cache) {
        int32_t f = p->cache->flags;
        if (!p->cache->tiles) f &= ~DRGNN_TOPO_TILES;
        return f;
    }
    const bool f4 = (p->net->n_feat & 3) == 0;      // (16-byte aligned rows are only needed where rows ARE multiples of 16 bytes)
    const bool tiles_ok = drgnn_topology_tiles_ok(b.maxN, b.maxE, p->net->n_feat) != 0 && p->set->x != nullptr &&
                          (!f4 || (((((uintptr_t)p->set->x) & 15) == 0) && ((((uintptr_t)slot_x) & 15) == 0)));
    const int32_t af = DRGNN_TOPO_HIER | DRGNN_TOPO_LEAN | DRGNN_TOPO_TILES;
    if (tiles_ok && epoch_plan_of(p, b, co, af).lean_ok) return af;
    return (p->net->kind != DRGNN_GINET && !p->inference) ? DRGNN_TOPO_HIER : 0;
}
int64_t epoch_next_b(const drgnn_epoch_plan* p, int64_t k) {      // graphs of mini-batch k + 1 (0: none)
    const int64_t first = (k + 1) * p->batch_size;
    if (first >= p->n_ids) return 0;
    return p->n_ids - first < p->batch_size ? p->n_ids - first : p->batch_size;
}
// sizes the two mini-batch slots and the step slabs for the largest mini-batch of the plan
int epoch_carve(const drgnn_epoch_plan* p, char* base, EpochCarve* c) {
    int rc = epoch_check(p);
    if (rc) return rc;
    const int64_t nb = (p->n_ids + p->batch_size - 1) / p->batch_size;
    int64_t capN = 1, capB = 1, ws_i = 4, ws_f = 4, xchg_words = 0;
    bool any_split = false;
    const drgnn_head_desc* hd = p->head;
    for (int64_t k = 0; k < nb; ++k) {
        EpochBatch b;
        if ((rc = epoch_batch(p, k, &b))) return rc;
        if (b.maxN <= 0) return DRGNN_E_CAPACITY;
        // (slot buffers are 256-byte aligned: the alignment of x does not depend on the slot)
        const int64_t co = p->cache ? 0 : epoch_next_b(p, k);
        const drgnn_step_plan pl = epoch_plan_of(p, b, co, epoch_topo_flags(p, b, co, nullptr));
        if (pl.slabs_per_graph == 2 && p->net->kind != DRGNN_GINET) any_split = true;
        if (pl.xchg_words * b.B > xchg_words) xchg_words = pl.xchg_words * b.B;
        if (pl.family == DRGNN_STEP_FAMILY_NONE ||
            pl.lds_bytes > DRGNN_LDS_LIMIT || b.maxN > 32767 || b.maxE > 65535 ||
            (!p->cache && drgnn_topology_lds_bytes(b.maxN, b.maxE > 0 ? b.maxE : 1) > DRGNN_LDS_LIMIT))
            return DRGNN_E_CAPACITY;
        TopoLayout lay;
        topo_layout(b.N, b.E, b.B, &lay);
        if (lay.i32[DRGNN_TI_COUNT] > ws_i) ws_i = lay.i32[DRGNN_TI_COUNT];
        if (lay.f32[DRGNN_TF_COUNT] > ws_f) ws_f = lay.f32[DRGNN_TF_COUNT];
        if (b.N > capN) capN = b.N;
        if (b.B > capB) capB = b.B;
    }
    int64_t o = 0;
    auto take = [&](int64_t bytes) { char* q = base ? base + o : nullptr; o += (bytes + 255) & ~(int64_t)255; return (void*)q; };
    const int F = p->net->n_feat, nbr = p->net->n_branch;
    for (int s = 0; s < 2; ++s) {
        EpochSlot& t = c->slot[s];
        if (p->cache) { t.x = nullptr; t.y = nullptr; t.ws_i32 = nullptr; t.ws_f32 = nullptr; t.tiles = nullptr; continue; }   // nothing to build
        t.x = (float*)take(capN * F * 4);
        t.tiles = (float*)take(drgnn_topology_tiles_elems(capN, F) * 4);      // (aggregation tiles of the slot's mini-batch)
        t.y = take(capB * 8);
        t.ws_i32 = (int32_t*)take(ws_i * 4);
        t.ws_f32 = p->need_weights ? (float*)take(ws_f * 4) : nullptr;
    }
    c->ptrs = p->cache ? nullptr : (int32_t*)take((nb > 0 ? nb : 1) * 3 * ((int64_t)p->batch_size + 1) * 4);
    c->readout = (float*)take(capB * hd->R * 4);
    c->partials = (float*)take(capB * (any_split ? 2 : nbr) * drgnn_net_partial_elems(p->net->kind, F) * 4);
    c->head_partials = (float*)take(capB * drgnn_head_compact_elems(hd->R, hd->H, hd->O) * 4);
    c->xchg_bytes = capB * nbr * (int64_t)(hd->H > DRGNN_H2 ? hd->H : DRGNN_H2) * 8;
    if (xchg_words * 8 > c->xchg_bytes) c->xchg_bytes = xchg_words * 8;
    c->xchg = (uint64_t*)take(c->xchg_bytes);
    c->bytes = o;
    return 0;
}
}  // namespace

int64_t drgnn_train_epoch_scratch_bytes(const drgnn_epoch_plan* plan) {
    EpochCarve c;
    const int rc = epoch_carve(plan, nullptr, &c);
    return rc ? (int64_t)rc : c.bytes;
}

// reduction of the step's slabs + optimiser update of mini-batch k: one launch, or (data parallel) gradient launch ->
// the caller's exchange -> Adam launch
static int epoch_update(const drgnn_epoch_plan* p, const EpochCarve& c, int64_t B, int64_t k, float* losses, void* stream,
                        int slabs_per_graph = 0) {
    const drgnn_head_desc* hd = p->head;
    const int fused = p->exchange ? 0 : 1;
    // (the last mini-batch of the call also writes the caller's loss word, drgnn_epoch_plan.last_loss)
    const int64_t nb = (p->n_ids + p->batch_size - 1) / p->batch_size;
    int rc = update_impl(slabs_per_graph, p->net, c.partials, B, p->g_conv1, p->g_conv2, c.head_partials, B, c.readout, hd->R, hd->H,
                         hd->O, p->head_offset, p->flat_param, p->flat_grad, p->exp_avg, p->exp_avg_sq, p->n_param,
                         p->step2, losses + k, p->lr, p->beta1, p->beta2, p->eps, fused, stream,
                         (k == nb - 1) ? p->last_loss : nullptr);
    if (rc || fused) return rc;
    if ((rc = p->exchange(p->exchange_user, k, B, stream))) return rc;
    return drgnn_adam_step(p->flat_param, p->flat_grad, p->exp_avg, p->exp_avg_sq, p->step2, p->n_param, p->lr, p->beta1,
                           p->beta2, p->eps, 0.0f, stream);
}

int drgnn_train_epoch(const drgnn_epoch_plan* p, void* scratch, int64_t scratch_bytes, float* pred, float* losses,
                      void* stream) {
    EpochCarve c;
    int rc = epoch_carve(p, (char*)scratch, &c);
    if (rc) return rc;
    if (p->n_ids == 0) return 0;
    const bool train = !p->inference;
    if (!scratch || !pred || (train && !losses) || ((uintptr_t)scratch & 15)) return DRGNN_E_ARG;
    if (scratch_bytes < c.bytes) return DRGNN_E_CAPACITY;
    const int64_t nb = (p->n_ids + p->batch_size - 1) / p->batch_size;
    const drgnn_head_desc* hd = p->head;
    drgnn_head_desc head = *hd;
    head.train = train ? 1 : 0;
    // the exchange words carry the step index as a tag: they only have to start from a value no step uses
#ifdef DRGNN_EMU
    memset(c.xchg, 0, (size_t)c.xchg_bytes);
#else
    HIP_TRY(hipMemsetAsync(c.xchg, 0, (size_t)c.xchg_bytes, (hipStream_t)stream));
#endif
    if (p->cache) {
        // cached topology: a mini-batch is its list of graph numbers -- step launch (+ update launch), nothing else
        for (int64_t k = 0; k < nb; ++k) {
            EpochBatch b;
            if ((rc = epoch_batch(p, k, &b))) return rc;
            drgnn_topology_cache tc = *p->cache;
            if (!train) tc.y = nullptr;
            drgnn_step_hints hints = {};
            hints.set_node_ptr = p->host_node_ptr; hints.set_edge_ptr = p->host_edge_ptr; hints.host_ids = p->host_ids + b.first;
            const drgnn_step_plan pl = epoch_plan_of(p, b, 0, epoch_topo_flags(p, b, 0, nullptr));
            const bool split = pl.slabs_per_graph == 2 && p->net->kind != DRGNN_GINET;
            hints.topo_flags = pl.topo_flags; hints.plan = &pl;
            hints.tiles = p->cache->tiles;
            // the graphs of mini-batch k + 1: prefetched by spare workgroups of this launch
            if (k + 1 < nb) { hints.next_ids = p->ids + b.first + b.B; hints.n_next = epoch_next_b(p, k); }
            rc = drgnn_net_train_step_cached(p->net, &head, &tc, p->ids + b.first, b.B, b.maxN, b.maxE, b.maxC, p->step2,
                                             pred + b.first * hd->O, c.readout, train ? c.head_partials : nullptr,
                                             train ? c.partials : nullptr, c.xchg, &hints, stream);
            if (rc) return rc;
            if (!train) continue;
            if ((rc = epoch_update(p, c, b.B, k, losses, stream, split ? 2 : 0))) return rc;
        }
        return 0;
    }
    if ((rc = drgnn_batch_offsets(p->set, p->ids, p->n_ids, p->batch_size, c.ptrs, stream))) return rc;
    const int64_t W = (int64_t)p->batch_size + 1;
    // mini-batch k's topology + node features + targets, straight from the resident set into slot k & 1
    auto request = [&](int64_t k, const EpochBatch& b) {
        const EpochSlot& u = c.slot[k & 1];
        drgnn_topology_request r = {};
        int32_t* pk = c.ptrs + k * 3 * W;
        r.node_ptr = pk; r.edge_ptr = pk + W; r.c1_ptr = pk + 2 * W;
        r.n_nodes = b.N; r.n_edges = b.E; r.len_cluster1 = b.C; r.n_graphs = b.B;
        r.max_nodes = b.maxN; r.max_edges = b.maxE;
        r.ws_i32 = u.ws_i32; r.ws_f32 = u.ws_f32; r.scratch_i32 = nullptr;
        r.set = p->set; r.ids = p->ids + b.first; r.x_out = u.x; r.y_out = train ? u.y : nullptr;
        // (what the step that consumes this workspace reads: epoch_topo_flags)
        r.flags = epoch_topo_flags(p, b, epoch_next_b(p, k), u.x);
        if (r.flags & DRGNN_TOPO_TILES) { r.tiles = u.tiles; r.n_feat = p->net->n_feat; }
        return r;
    };
    EpochBatch cur, nxt;
    std::vector<int32_t> hn, he;       // (slot offset tables of the current mini-batch, reused)
    if ((rc = epoch_batch(p, 0, &cur))) return rc;
    int32_t cur_flags = 0;
    {
        const drgnn_topology_request r0 = request(0, cur);
        cur_flags = r0.flags;
        if ((rc = drgnn_topology_build_request(&r0, stream))) return rc;
    }
    for (int64_t k = 0; k < nb; ++k) {
        const EpochSlot& t = c.slot[k & 1];
        drgnn_topology_request req;
        const bool more = k + 1 < nb;
        if (more) {
            if ((rc = epoch_batch(p, k + 1, &nxt))) return rc;
            req = request(k + 1, nxt);
        }
        // the slot offsets of this mini-batch are known here (host size tables): hand them to the launch
        drgnn_step_hints hints = {};
        const drgnn_step_plan pl = epoch_plan_of(p, cur, more ? nxt.B : 0, cur_flags);      // (cur_flags: what request() asked the builder for)
        const bool split = pl.slabs_per_graph == 2 && p->net->kind != DRGNN_GINET;
        hints.topo_flags = cur_flags;
        hints.tiles = (cur_flags & DRGNN_TOPO_TILES) ? t.tiles : nullptr;
        hints.plan = &pl;
        if (cur.B <= DRGNN_STEP_DIMS_MAX) {
            hn.resize((size_t)cur.B + 1); he.resize((size_t)cur.B + 1);
            hn[0] = 0; he[0] = 0;
            for (int64_t q = 0; q < cur.B; ++q) {
                const int64_t id = p->host_ids[cur.first + q];
                hn[q + 1] = hn[q] + (int32_t)(p->host_node_ptr[id + 1] - p->host_node_ptr[id]);
                he[q + 1] = he[q] + (int32_t)(p->host_edge_ptr[id + 1] - p->host_edge_ptr[id]);
            }
            hints.host_node_ptr = hn.data(); hints.host_edge_ptr = he.data();
        }
        rc = drgnn_net_train_step(p->net, &head, t.x, train ? t.y : nullptr, p->step2, t.ws_i32, t.ws_f32, cur.N, cur.E,
                                  cur.B, cur.maxN, cur.maxE, cur.maxC, pred + cur.first * hd->O, c.readout,
                                  train ? c.head_partials : nullptr, train ? c.partials : nullptr, c.xchg,
                                  more ? &req : nullptr, &hints, stream);
        if (rc) return rc;
        if (more) cur_flags = req.flags;
        if (!train) { if (more) cur = nxt; continue; }
        if ((rc = epoch_update(p, c, cur.B, k, losses, stream, split ? 2 : 0))) return rc;
        if (more) cur = nxt;
    }
    return 0;
}

}  // extern "C"
