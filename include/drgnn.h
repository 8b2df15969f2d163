/*
 * drgnn.h -- C ABI of the MI355X-native Deeprank-GNN message-passing hot path.
 *
 * One shared object (deeprank-gnn_amd/csrc/libdrgnn.so, built by hipcc for gfx950).
 * Every entry point takes raw DEVICE pointers, explicit sizes and a hipStream_t (passed
 * as void*), never allocates, never synchronises, and is hipGraph-capturable.  Return
 * value: 0 on success, a positive hipError_t from the launch, or a negative DRGNN_E_*
 * for argument errors detected on the host.  Data-dependent errors detected on the
 * device (malformed edge lists, wrong cluster1 length ...) are recorded per graph in the
 * topology workspace (DRGNN_TI_GSTAT / DRGNN_TI_ERR) and poison that graph's outputs with
 * NaN; read them back with drgnn_topology_status().
 *
 * The reference has no C/FFI boundary (it is pure Python on torch_geometric /
 * torch_scatter / torch_sparse, which are un-vendored).  Each entry point cites the
 * reference function (file:line under the reference repository) whose arithmetic it
 * replaces; the Python class protocol above this ABI (GINet / sGAT / FoutNet,
 * community_pooling ...) lives in deeprank-gnn_amd/ and mirrors the reference names.
 *
 * Index tensors arrive as the reference stores them (int64 edge_index [2,E], cluster
 * ids, batch vector: DataSet.py:268-269,354-357); everything derived is int32.
 */
#ifndef DRGNN_H
#define DRGNN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DRGNN_ABI_VERSION 4

/* host-side argument errors */
#define DRGNN_E_ARG      (-1)   /* null pointer / negative size / bad mode            */
#define DRGNN_E_CAPACITY (-2)   /* workspace too small for the request                */
#define DRGNN_E_WIDTH    (-3)   /* feature width not supported by the fused kernels   */

/* device-side status bits (DRGNN_TI_ERR[0]) */
#define DRGNN_S_EDGE_RANGE    1  /* an edge endpoint lies outside its graph's node range */
#define DRGNN_S_UNSORTED      2  /* batch vector / edge list not grouped by graph        */
#define DRGNN_S_CLUSTER_RANGE 4  /* a graph's cluster ids span more than N_g+E_g+1 values*/
#define DRGNN_S_CLUSTER1_LEN  8  /* len(cluster1) != number of depth-0 clusters          */
/* fault bits of step2[2] (fused training step) */
#define DRGNN_FAULT_EXCHANGE  1  /* a GINet branch workgroup waited in vain for its partner's half of fc1 (value = NaN) */
#define DRGNN_FAULT_SPLIT     2  /* a half-graph workgroup of the node-split layout waited in vain for its partner's hand-off */

/* layer kinds (what "conv" means) */
#define DRGNN_GINET 0  /* z_i = sum_{e:row=i} W x_col            ginet.py:50-73 (alpha == 1)   */
#define DRGNN_SGAT  1  /* z_i = mean_e a_e [x_i||x_col] W + b     sGAT.py:62-93                 */
#define DRGNN_FOUT  2  /* z_i = x_i Wc + mean_e (x Wn)_col + b    foutnet.py:56-82 (NaN if deg 0)*/

/* ---- topology workspace ----------------------------------------------------------
 * Built once per mini-batch from the index tensors only (no learned quantity), shared
 * by both GINet branches and by forward and backward.  Layout: "padded per-graph
 * segments" -- graph g owns node slots [nptr[g], nptr[g+1]) and edge slots
 * [eptr[g], eptr[g+1]) of every array, pooled levels use a prefix of the same slots
 * (C0_g <= N_g, E1_g <= E_g), pointer-like arrays of length n+1 start at nptr[g]+g.
 * All node/cluster ids stored inside are LOCAL to their graph.
 */
enum drgnn_topo_i32 {
    DRGNN_TI_NPTR = 0,   /* [B+1]  node offsets per graph                                   */
    DRGNN_TI_EPTR,       /* [B+1]  edge offsets per graph                                   */
    DRGNN_TI_ROWPTR0,    /* [N+B]  CSR by row (edge_index[0]) of the input graph            */
    DRGNN_TI_COL0,       /* [E]    neighbour (edge_index[1]), rows ordered by edge id       */
    DRGNN_TI_EID0,       /* [E]    original edge id (local) of every CSR slot               */
    DRGNN_TI_COLPTR0,    /* [N+B]  CSC (transpose) of the input graph                        */
    DRGNN_TI_ROWIDX0,    /* [E]                                                              */
    DRGNN_TI_TSLOT0,     /* [E]    CSR slot of every CSC entry (to look up its weight)      */
    DRGNN_TI_CL0,        /* [N]    consecutive depth-0 cluster id of every node             */
    DRGNN_TI_NC0,        /* [B]    number of depth-0 clusters                                */
    DRGNN_TI_MPTR0,      /* [N+B]  member lists of depth-0 clusters                          */
    DRGNN_TI_MEM0,       /* [N]    members, ascending node id inside a cluster               */
    DRGNN_TI_ROWPTR1,    /* [N+B]  CSR of the pooled graph (pool_edge output, sorted)        */
    DRGNN_TI_COL1,       /* [E]                                                              */
    DRGNN_TI_NE1,        /* [B]    number of pooled edges                                    */
    DRGNN_TI_COLPTR1,    /* [N+B]  CSC of the pooled graph                                   */
    DRGNN_TI_ROWIDX1,    /* [E]                                                              */
    DRGNN_TI_TSLOT1,     /* [E]                                                              */
    DRGNN_TI_CL1,        /* [N]    consecutive depth-1 cluster id of every depth-0 cluster   */
    DRGNN_TI_NC1,        /* [B]                                                              */
    DRGNN_TI_MPTR1,      /* [N+B]                                                            */
    DRGNN_TI_MEM1,       /* [N]                                                              */
    DRGNN_TI_CPTR0,      /* [B+1]  exclusive scan of NC0  (filled by drgnn_topology_finalize)*/
    DRGNN_TI_E1PTR,      /* [B+1]  exclusive scan of NE1                                     */
    DRGNN_TI_CPTR1,      /* [B+1]  exclusive scan of NC1                                     */
    DRGNN_TI_ERR,        /* [4]    [0] batch-level status bits (offset derivation)           */
    DRGNN_TI_GSTAT,      /* [2B]   per-graph status bits (edge half, member half), rewritten by every build */
    /* Hierarchical node order (built with depth 1): nodes sorted by (depth-1 cluster of their depth-0 cluster, depth-0
     * cluster, node id), i.e. the members of a depth-0 cluster are consecutive positions and the depth-0 clusters of a
     * depth-1 cluster are consecutive runs, in the order of MEM1.  The node-split step kernels (csrc/drgnn_step2.h) keep
     * their rows in this order: cluster maxima run over contiguous rows and a prefix of the depth-1 clusters is a prefix
     * of the positions (what lets two workgroups share one graph with both poolings local). */
    DRGNN_TI_HORD,       /* [N]    local node id at hierarchical position p                                      */
    DRGNN_TI_HMP0,       /* [N+B]  first position of the q-th depth-0 cluster IN MEM1 ORDER, q = 0..C0            */
    DRGNN_TI_HSPLIT,     /* [4B]   per graph {k, MPTR1[k], HMP0[MPTR1[k]], C1}: k = the number of leading depth-1
                                   clusters whose node total is closest to N/2 (the two-workgroup split point)   */
    DRGNN_TI_IHORD,      /* [N]    hierarchical position of local node i (the inverse of HORD)                    */
    DRGNN_TI_COUNT
};
enum drgnn_topo_f32 {
    DRGNN_TF_W0 = 0,     /* [E] edge_attr in CSR0 slot order                                 */
    DRGNN_TF_W1,         /* [E] pooled edge_attr (duplicates summed) in CSR1 slot order      */
    DRGNN_TF_COUNT
};

/* Element offsets of every array inside the two workspaces, and their total sizes.
 * off_i32 has DRGNN_TI_COUNT+1 entries (last = total int32 elements), off_f32 has
 * DRGNN_TF_COUNT+1 entries. */
int drgnn_topology_layout(int64_t n_nodes, int64_t n_edges, int64_t n_graphs,
                          int64_t* off_i32, int64_t* off_f32);

/* Build the topology workspace.
 * Replaces, for the whole mini-batch and without host synchronisation:
 *   - the implicit COO use of edge_index in the conv layers (ginet.py:52, sGAT.py:64,
 *     foutnet.py:58,71-73)                                      -> CSR0 / CSC0
 *   - get_preloaded_cluster (community_pooling.py:25-30) + consecutive_cluster [3P]
 *                                                               -> CL0/CL1, member lists
 *   - pool_edge [3P] on the external edges (community_pooling.py:200-201): relabel, drop
 *     self loops, sort by (row,col), merge duplicates with edge_attr summed
 *                                                               -> CSR1 / CSC1 / W1
 * edge_index  int64 [2,E] (global node ids, edges grouped by graph)
 * edge_attr   float [E] or NULL (one edge feature, as the reference supports)
 * batch       int64 [N] ascending graph id per node
 * cluster0    int64 [N]  per-graph cluster ids (any integers; made consecutive here);
 *             NULL = graph-only build (CSR0/CSC0/W0), used by the stand-alone conv layers
 * cluster1    int64 [L1] per-graph ids of the depth-0 clusters; L1 must equal sum C0_g;
 *             may be NULL (then only depth 0 is built: CL1.. untouched)
 * node_ptr / edge_ptr / c1_ptr  int32 [B+1] or NULL: per-graph offsets if the caller
 *             knows them (our DataLoader does); otherwise derived on the device.
 * max_nodes / max_edges: upper bounds on any single graph's size (used to size LDS);
 *             pass 0 if unknown (kernels then use their global-memory scratch path,
 *             scratch_i32 must hold drgnn_topology_scratch_elems() ints).
 */
int64_t drgnn_topology_scratch_elems(int64_t n_nodes, int64_t n_edges, int64_t n_graphs);
/* LDS bytes the builder needs for the given per-graph bounds; it runs out of LDS when this
 * is <= 160 KiB and max_nodes > 0, else out of scratch_i32. */
int64_t drgnn_topology_lds_bytes(int32_t max_nodes, int32_t max_edges);
int drgnn_topology_build(const int64_t* edge_index, const float* edge_attr,
                         const int64_t* batch, const int64_t* cluster0, const int64_t* cluster1,
                         const int32_t* node_ptr, const int32_t* edge_ptr, const int32_t* c1_ptr,
                         int64_t n_nodes, int64_t n_edges, int64_t len_cluster1, int64_t n_graphs,
                         int32_t max_nodes, int32_t max_edges,
                         int32_t* ws_i32, float* ws_f32, int32_t* scratch_i32, void* stream);

/* Exclusive scans CPTR0 / E1PTR / CPTR1 (only needed to materialise compact pooled
 * tensors for the function-level API: community_pooling(), max_pool_x()). */
int drgnn_topology_finalize(int32_t* ws_i32, int64_t n_nodes, int64_t n_edges,
                            int64_t n_graphs, void* stream);

/* status4[0] = OR of all status bits, status4[1] = first offending graph (or -1); this one
 * DOES synchronise the stream. */
int drgnn_topology_status(const int32_t* ws_i32, int64_t n_nodes, int64_t n_edges,
                          int64_t n_graphs, int32_t* status4, void* stream);

/* ---- fused network body ------------------------------------------------------------
 * One convolution branch of GINet / sGAT / FoutNet:
 *   conv1 -> relu -> community_pooling(max) -> conv2 -> relu -> max_pool_x -> graph mean
 * (ginet.py:103-114,133; sGAT.py:119-133; foutnet.py:108-120), one workgroup per
 * (graph, branch), intermediates in LDS.  The FC head (fc1/relu/dropout/fc2) stays
 * outside.  H1 = 16, H2 = 32 as hard-coded in the reference (ginet.py:87-92).
 *
 * Weight operands are described as strided [K,H] matrices so that each model's own
 * parameter layout is used in place: element (k,h) = ptr[k*sk + h*sh].
 *   GINet   nbr = fc.weight [H,F]           (sk=1, sh=F)   self = NULL   bias = NULL
 *   sGAT    nbr = weight[F:2F,:], self = weight[0:F,:]     (sk=H, sh=1)  bias [H]
 *   FoutNet nbr = Wn, self = Wc  [F,H]                     (sk=H, sh=1)  bias [H]
 */
typedef struct drgnn_conv_params {
    const float* w_nbr;  int64_t nbr_sk, nbr_sh;
    const float* w_self; int64_t self_sk, self_sh;   /* NULL for GINet */
    const float* bias;                                /* NULL for GINet */
} drgnn_conv_params;

typedef struct drgnn_conv_grads {       /* same striding as the parameters; NULL = skip */
    float* w_nbr;
    float* w_self;
    float* bias;
} drgnn_conv_grads;

#define DRGNN_MAX_BRANCH 2

typedef struct drgnn_net_desc {
    int32_t kind;          /* DRGNN_GINET / DRGNN_SGAT / DRGNN_FOUT                    */
    int32_t n_branch;      /* 2 for GINet (conv*, conv*_ext), else 1                   */
    int32_t n_feat;        /* F: input node features                                    */
    int32_t reserved;
    drgnn_conv_params conv1[DRGNN_MAX_BRANCH];
    drgnn_conv_params conv2[DRGNN_MAX_BRANCH];
} drgnn_net_desc;

/* Saved-for-backward tensors, all in the padded per-graph layout, per branch b:
 *   xp   float [n_branch][N][16]   pooled node features (input of conv2)
 *   arg0 int32 [n_branch][N][16]   local node id that won the depth-0 max, -1 = no grad
 *   arg1 int32 [n_branch][N][32]   local depth-0 cluster that won the depth-1 max, -1 = no grad
 * readout float [B][32*n_branch]   per-graph mean of the depth-1 pooled features
 */
/* max_nodes / max_edges / max_c0: upper bounds on any single graph's node count, edge count
 * and number of depth-0 clusters (max_c0 = 0: unknown -> max_nodes).  They size the LDS
 * carve (forward and backward differ); when it exceeds 160 KiB (or max_nodes == 0) the
 * kernels run out of `scratch_f32` (global, drgnn_net_scratch_elems() floats) instead. */
int64_t drgnn_net_lds_bytes(int32_t kind, int32_t n_feat, int32_t max_nodes, int32_t max_edges,
                            int32_t max_c0, int32_t backward);

int drgnn_net_forward(const drgnn_net_desc* net, const float* x,
                      const int32_t* ws_i32, const float* ws_f32,
                      int64_t n_nodes, int64_t n_edges, int64_t n_graphs,
                      int32_t max_nodes, int32_t max_edges, int32_t max_c0,
                      float* xp, int32_t* arg0, int32_t* arg1, float* readout,
                      float* scratch_f32, int32_t* step_inc /* optional: ++*step_inc once */,
                      void* stream);

/* Backward of the above.  grad_readout float [B][32*n_branch].  Every workgroup writes its
 * graph's parameter-gradient contribution into `partials` (float [n_graphs*n_branch][P],
 * P = drgnn_net_partial_elems(); K x H row-major blocks
 * [dW1nbr F*16][dW1self F*16][db1 16][dW2nbr 16*32][dW2self 16*32][db2 32]) and, when grad_x
 * is given, d loss / d x per branch into grad_x float [n_branch][N][F]. */
int64_t drgnn_net_partial_elems(int32_t kind, int32_t n_feat);
int64_t drgnn_net_scratch_elems(int32_t kind, int32_t n_feat, int64_t n_nodes, int64_t n_edges,
                                int64_t n_graphs);
int drgnn_net_backward(const drgnn_net_desc* net, const float* x, const float* grad_readout,
                       const int32_t* ws_i32, const float* ws_f32,
                       int64_t n_nodes, int64_t n_edges, int64_t n_graphs,
                       int32_t max_nodes, int32_t max_edges, int32_t max_c0,
                       const float* xp, const int32_t* arg0, const int32_t* arg1,
                       float* grad_x, float* partials, float* scratch_f32,
                       int32_t* step_inc /* optional: ++*step_inc once per launch */, void* stream);

/* Fixed-order (deterministic) sum of the per-graph partials into the strided gradient
 * tensors (overwritten, not accumulated); with grad_x, branches 1.. are summed into
 * branch 0.  Replaces the scatter-add autograd performs for the shared weights. */
int drgnn_net_reduce_grads(const drgnn_net_desc* net, const float* partials, int64_t n_nodes,
                           int64_t n_graphs, drgnn_conv_grads* g_conv1, drgnn_conv_grads* g_conv2,
                           float* grad_x, void* stream);

/* ---- stand-alone layers and pooling functions (the reference's function-level API) ----------
 * For custom nets built from the reference's layers (README "custom GNN" snippet).  A single
 * convolution of arbitrary width H <= 128 on ONE graph described by a graph-only topology
 * workspace (drgnn_topology_build with n_graphs = 1, cluster0 = NULL):
 *   GINetConvLayer.forward (ginet.py:50-73), sGraphAttentionLayer.forward (sGAT.py:62-93),
 *   FoutLayer.forward (foutnet.py:56-82).  `u` / `du` are [N, H] (GINet) or [N, 2H] scratch;
 *   partials holds drgnn_conv_layer_slabs(N) * drgnn_conv_layer_partial_elems() floats.
 */
int64_t drgnn_conv_layer_slabs(int64_t n_nodes);
int64_t drgnn_conv_layer_partial_elems(int32_t kind, int32_t F, int32_t H);
int drgnn_conv_layer_forward(int32_t kind, const float* x, int64_t n_nodes, int32_t F, int32_t H,
                             const drgnn_conv_params* p, const int32_t* ws_i32, const float* ws_f32,
                             int64_t n_edges, float* u, float* out, void* stream);
int drgnn_conv_layer_backward(int32_t kind, const float* x, int64_t n_nodes, int32_t F, int32_t H,
                              const drgnn_conv_params* p, const int32_t* ws_i32, const float* ws_f32,
                              int64_t n_edges, const float* grad_out, float* du, float* partials,
                              const drgnn_conv_grads* g, float* grad_x, void* stream);
/* Cluster pooling over the depth-0 member lists of a FINALIZED topology workspace:
 * op 0 = max with argmax (scatter_max inside community_pooling.py:197 / max_pool_x [3P]; first
 * maximum in member order, empty cluster -> 0 / arg = N), op 1 = mean (scatter_mean,
 * community_pooling.py:212).  out [C0_total, H] compact, arg = GLOBAL node id. */
int drgnn_segpool_forward(const int32_t* ws_i32, int64_t n_nodes, int64_t n_edges, int64_t n_graphs,
                          const float* x, int32_t H, int32_t op, float* out, int64_t* arg, void* stream);
int drgnn_segmax_backward(const float* grad_out, const int64_t* arg, int64_t n_clusters, int32_t H,
                          int64_t n_nodes, float* grad_x /* zero-filled */, void* stream);
/* pool_edge [3P] result of a FINALIZED workspace in the reference's form: edge_index int64
 * [2, e1_total] (consecutive global cluster ids, sorted by (row, col)), edge_attr [e1_total]. */
int drgnn_pooled_edges_export(const int32_t* ws_i32, const float* ws_f32, int64_t n_nodes, int64_t n_edges,
                              int64_t n_graphs, int64_t e1_total, int64_t* edge_index, float* edge_attr,
                              void* stream);
/* get_preloaded_cluster (community_pooling.py:25-30): cluster[batch == g] += running offset, in
 * place, no host round trips.  scratch: n_graphs + 1 int64. */
int drgnn_cluster_offset(int64_t* cluster, const int32_t* node_ptr, int64_t n_graphs, int64_t* scratch,
                         void* stream);

/* ---- dense head, loss and optimiser (the rest of one training step) -----------------------
 * What the reference trainer runs around the message-passing body for every mini-batch
 * (NeuralNet.py:489-506): the FC head of the nets (ginet.py:136-139, sGAT.py:134-135,
 * foutnet.py:121-122: fc1 -> relu -> dropout(p) -> fc2), MSELoss / (weighted)
 * CrossEntropyLoss with mean reduction (NeuralNet.py:239-263), their backward, and the Adam
 * update (NeuralNet.py:183-184, torch defaults).  torch.nn.Linear layouts: w1 [H,R], b1 [H],
 * w2 [O,H], b2 [O].
 */
#define DRGNN_TASK_REG   0
#define DRGNN_TASK_CLASS 1
/* DRGNN_TASK_GRAD -- the autograd boundary (NeuralNet.py:493-502: pred = model(batch); loss = loss_fn(pred, y); loss.backward()):
 * the loss lives OUTSIDE the library, `target` is float [B, O] = d loss / d pred as the caller's autograd hands it over, and
 * the launch back-propagates exactly that (the loss slot of the head slabs is written as 0).  With transform_sigmoid the
 * upstream gradient is taken with respect to the transformed prediction.  Fused step launches of the aggregation-first
 * family on a per-mini-batch workspace only (drgnn_net_train_step; DRGNN_E_ARG elsewhere). */
#define DRGNN_TASK_GRAD  2
typedef struct drgnn_head_desc {
    int32_t R, H, O;          /* readout width, hidden width, outputs (O <= 16)              */
    int32_t task;             /* DRGNN_TASK_REG: target float [B]; _CLASS: target int64 [B]   */
    int32_t train;            /* 1: dropout + loss + gradients; 0: predictions only (dropout off);
                                 2 (fused step launches only): the FORWARD of a training step -- predictions only, dropout ON
                                 with the mask of optimiser step step2[0], i.e. the mask a train = 1 launch draws until
                                 step2[0] is committed (model(batch) in training mode, its backward being a later launch) */
    float   p_drop;           /* dropout probability (GINet 0.4, others 0)                    */
    uint32_t seed;            /* dropout stream seed (mixed with the device step counter)     */
    int32_t transform_sigmoid; /* regression only: pred = sigmoid(fc2 output) before the loss (reference
                                  NeuralNet.format_output, NeuralNet.py:616-631); predictions are reported transformed */
    const float* w1; const float* b1; const float* w2; const float* b2;
    const float* class_w;     /* [O] class weights or NULL                                    */
    const float* drop_mask;   /* NULL (product): the counter-hash dropout stream.  Otherwise float [B, H] of 0 / 1: hidden unit h
                                 of the launch's graph g is kept iff drop_mask[g*H + h] != 0 and scaled by 1 / (1 - p_drop) --
                                 exactly F.dropout's arithmetic (ginet.py:138) with the mask given, so that a dropout-on launch
                                 can be compared element-wise with the oracle (tests; fused step kernels only)      */
} drgnn_head_desc;

/* The arguments of drgnn_topology_build bundled, to ask a body launch to ALSO build the topology
 * of the NEXT mini-batch in the same launch (its workgroups are appended to the grid).  The two
 * jobs are independent -- the builder only reads index tensors -- so one hides behind the other and
 * a kernel boundary disappears (software pipelining across training steps).  Falls back to
 * separate launches when LDS does not fit or the per-graph offsets are not supplied. */
/* The dataset resident in HBM, graph-major (described at drgnn_collate below). */
typedef struct drgnn_graph_set {
    int64_t n_graphs, n_nodes, n_edges, len_cluster1;
    int32_t n_feat, y_bytes;
    const int64_t* node_ptr; const int64_t* edge_ptr; const int64_t* c1_ptr;   /* c1_ptr null: no cluster1 */
    const float* x; const int64_t* edge_index; const float* edge_attr;          /* edge_attr may be null */
    const int64_t* cluster0; const int64_t* cluster1; const void* y;            /* each may be null */
} drgnn_graph_set;
typedef struct drgnn_topology_request {
    const int64_t* edge_index; const float* edge_attr; const int64_t* batch;
    const int64_t* cluster0; const int64_t* cluster1;
    const int32_t* node_ptr; const int32_t* edge_ptr; const int32_t* c1_ptr;
    int64_t n_nodes, n_edges, len_cluster1, n_graphs;
    int32_t max_nodes, max_edges;
    int32_t* ws_i32; float* ws_f32; int32_t* scratch_i32;
    /* Resident-set mode (set != NULL; the five tensor pointers above are ignored): slot g of the mini-batch is
     * graph ids[g] of `set`, whose index data is read in place (local ids, no collate); node_ptr / edge_ptr /
     * c1_ptr are the mini-batch's slot offset tables (drgnn_batch_offsets) and are required.  The builder also
     * gathers the slots' node features into x_out [n_nodes, F] and targets into y_out [n_graphs] (either may be
     * NULL).  Pooled edge weights are built when ws_f32 and set->edge_attr are both given. */
    const drgnn_graph_set* set; const int32_t* ids; float* x_out; void* y_out;
    int32_t flags, reserved;      /* DRGNN_TOPO_* below */
    /* DRGNN_TOPO_TILES: node features the level-0 aggregation is formed from -- x [n_nodes, n_feat] of the mini-batch (NULL in
     * resident-set mode: the set's x) -- and where it goes: tiles [drgnn_topology_tiles_elems(n_nodes, n_feat)] */
    const float* x; float* tiles;
    int32_t n_feat, reserved2;
} drgnn_topology_request;
/* request flags.  DRGNN_TOPO_HIER: also build the hierarchical node order (DRGNN_TI_HORD / HMP0 / HSPLIT; needs cluster1) --
 * what the node-split step kernels of the single-branch nets consume (4 more phases on the builder's member-list chain, so
 * callers that never run those kernels -- GINet -- leave it out).  drgnn_topology_build always builds it. */
#define DRGNN_TOPO_HIER 1
/* DRGNN_TOPO_LEAN (with DRGNN_TOPO_HIER and cluster1): build ONLY what the aggregation-first training kernels read
 * (csrc/drgnn_step2.h, drgnn_step3.h): ROWPTR0 / COL0 / EID0 (+ W0), CL0 / NC0, ROWPTR1 / COL1 / NE1 (+ W1), COLPTR1 / ROWIDX1
 * (+ TSLOT1 with edge weights), CL1 / NC1 / MPTR1 / MEM1, HORD / HMP0 / HSPLIT.  NOT built: CSC0 (COLPTR0 / ROWIDX0 / TSLOT0)
 * and the depth-0 member lists (MPTR0 / MEM0) -- those arrays keep whatever an earlier build left there.  The builder then
 * runs two short chains (8 barrier-separated phases each instead of ~19: one concatenated scan for the row pointers and
 * both cluster rankings, orders by counting instead of bucket sorts, the pooled CSC from a transposed bitmap), which keeps
 * it hidden behind the step workgroups it is co-launched with.  A launch that is not an aggregation-first training step
 * refuses a workspace built this way (drgnn_step_hints.topo_flags): DRGNN_E_ARG. */
#define DRGNN_TOPO_LEAN 2
/* DRGNN_TOPO_TILES (with DRGNN_TOPO_HIER): the builder also forms the LEVEL-0 NEIGHBOUR AGGREGATION of every node -- it depends
 * on the inputs only (node features, edges, edge weights), not on a parameter -- so that the training step of the mini-batch
 * starts from it instead of gathering x rows over edge_index inside the step kernel: the aggregation rides in the builder's
 * workgroups of the PREVIOUS launch like the topology itself (cached-topology mode: it is formed once per graph).  Layout of
 * `tiles` (node order, n = n_nodes of the workspace, F = n_feat, TF = F rounded up to a multiple of 4: rows are zero padded so
 * that the step kernels load them with 128-bit requests whatever the feature count):
 *     S [n][TF]  S_i = sum over the edges e = (i, j) in edge-id order of [w_e] x_j     (w_e: with edge weights only)
 *     D [n]      1 / deg_i (without weights: 0 for an isolated node; with weights: 1 / max(deg_i, 1))
 *     C [n]      with weights: mean edge weight of the row (sum_e w_e) D_i; without: 1
 *     X [n][TF]  only when F % 4 != 0: the node features, rows zero padded (what sGAT / FoutNet multiply with their self weights);
 *                starts at element n TF + (2 n rounded up to a multiple of 4): 16-byte aligned rows whatever the parity of n
 * GINetConvLayer (ginet.py:50-73) is relu(S W); FoutLayer (foutnet.py:56-82) relu(D (S Wn) + x Wc + b); sGraphAttentionLayer
 * (sGAT.py:62-93) relu(D (S Wn) + C (x Ws) + b).  Needs drgnn_topology_tiles_ok() and, when F % 4 == 0, 16-byte aligned x. */
#define DRGNN_TOPO_TILES 4
/* elements of a tiles buffer / 1 when the builder can form tiles for graphs of these bounds (its LDS holds an x tile then) */
int64_t drgnn_topology_tiles_elems(int64_t n_nodes, int32_t n_feat);
int32_t drgnn_topology_tiles_ok(int32_t max_nodes, int32_t max_edges, int32_t n_feat);
/* The same tiles from a workspace that is already built (its CSR0, and with use_weights its W0): a second flavour for a
 * workspace shared by nets with and without edge weights (a resident set's cached topology).  x [n_nodes, n_feat]: the node
 * features the workspace's graphs are laid out over.  Own launch; not a hot path. */
int drgnn_topology_tiles(const int32_t* ws_i32, const float* ws_f32, int64_t n_nodes, int64_t n_edges, int64_t n_graphs,
                         const float* x, int32_t n_feat, int32_t use_weights, float* tiles, void* stream);
/* drgnn_topology_build from a request (either mode), own launch. */
int drgnn_topology_build_request(const drgnn_topology_request* request, void* stream);
/* Slot offset tables of EVERY mini-batch of an epoch in one launch: mini-batch k = ids[k*batch_size, ...) gets
 * ptrs[k][0] = node offsets, ptrs[k][1] = edge offsets, ptrs[k][2] = cluster1 offsets, each batch_size+1 int32
 * (ptrs: [n_batches][3][batch_size+1]; a short last mini-batch uses a prefix of each row).  batch_size <= 4096. */
int drgnn_batch_offsets(const drgnn_graph_set* set, const int32_t* ids, int64_t n_ids, int32_t batch_size,
                        int32_t* ptrs, void* stream);

/* Backward of the body with the FC head, loss and their backward evaluated per graph INSIDE the
 * same launch (the head is row-wise), instead of taking grad_readout from drgnn_head_step:
 * reads readout [B,32*n_branch] (forward output) and the targets, writes pred [B,O], one head
 * partial slab per GRAPH (head_partials [B][drgnn_head_partial_elems]) and the conv partials.
 * Dropout stream id = *step - 1 (the forward launch of the same step has incremented it). */
int drgnn_net_backward_fused_head(const drgnn_net_desc* net, const drgnn_head_desc* head, const float* x,
                                  const float* readout, const void* target, const int32_t* step,
                                  const int32_t* ws_i32, const float* ws_f32, int64_t n_nodes,
                                  int64_t n_edges, int64_t n_graphs, int32_t max_nodes, int32_t max_edges,
                                  int32_t max_c0, const float* xp, const int32_t* arg0, const int32_t* arg1,
                                  float* pred, float* head_partials, float* grad_x, float* partials,
                                  float* scratch_f32,
                                  const drgnn_topology_request* next_topology /* optional */,
                                  void* stream);

/* One workgroup per tile of graphs (16 for B <= 512, else 64; drgnn_head_num_slabs()).  Writes pred [B,O]; when train: grad_readout [B,R]
 * (d loss / d readout for the mean loss over the B graphs) and one partial slab per workgroup
 * ([dW1 H*R][db1 H][dW2 O*H][db2 O][loss][weight], drgnn_head_partial_elems() floats).
 * `step` (device int32) selects the dropout stream; it is NOT modified here. */
int64_t drgnn_head_partial_elems(int32_t R, int32_t H, int32_t O);
int64_t drgnn_head_num_slabs(int64_t n_graphs);
int drgnn_head_step(const drgnn_head_desc* head, const float* readout, const void* target,
                    int64_t n_graphs, const int32_t* step, float* pred, float* grad_readout,
                    float* partials, void* stream);
/* Fixed-order sum of the head partials into grad_block = [fc1.weight | fc1.bias | fc2.weight |
 * fc2.bias] (contiguous), the batch loss into *loss, and ++*step (the optimiser step count). */
int drgnn_head_reduce(const float* partials, int64_t n_graphs, int32_t R, int32_t H, int32_t O,
                      float* grad_block, float* loss, int32_t* step, void* stream);
/* torch.optim.Adam single-tensor semantics on flat fp32 buffers; *step must already count
 * this update (>= 1). */
int drgnn_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                    const int32_t* step, int64_t n, float lr, float beta1, float beta2, float eps,
                    float weight_decay, void* stream);

/* Single-launch parameter update for one process: fixed-order reduction of the conv partials
 * (as drgnn_net_reduce_grads) and of the head partials (as drgnn_head_reduce, without
 * touching the step counter), each reduced element immediately followed by its Adam update.
 * g_conv1/g_conv2 must point INTO flat_grad; head_offset = element offset of fc1.weight in
 * the flat buffers; *step must already count this update (the forward launch's step_inc).
 * apply_adam = 0 only produces the flat gradient + loss (data parallel: all-reduce, then
 * drgnn_adam_step).  weight_decay is not supported here (use drgnn_adam_step). */
int drgnn_train_update(const drgnn_net_desc* net, const float* conv_partials, int64_t n_graphs,
                       drgnn_conv_grads* g_conv1, drgnn_conv_grads* g_conv2,
                       const float* head_partials, int64_t head_slabs /* rows of head_partials */,
                       int32_t R, int32_t H, int32_t O, int64_t head_offset, float* flat_param, float* flat_grad, float* exp_avg,
                       float* exp_avg_sq, int64_t n_param, const int32_t* step, float* loss,
                       float lr, float beta1, float beta2, float eps, int32_t apply_adam, void* stream);

/* ---- fused training step --------------------------------------------------------------------
 * Body forward + FC head + loss + body backward of one mini-batch in ONE launch (every workgroup
 * keeps its graph, activations and argmax indices in LDS between the forward and the backward
 * half), optionally sharing the grid with the topology build of the NEXT mini-batch.  Replaces,
 * together with drgnn_step_update, the whole loop body of NeuralNet._epoch (NeuralNet.py:489-506:
 * zero_grad, model(batch), loss, backward, optimizer.step) for GINet / sGAT / FoutNet.
 *   step2     device int32[4], zero-initialised by the owner: [0] = optimiser steps completed (selects the dropout
 *             stream, read only); [1] = index of this step, written here (drgnn_step_update commits [0] = [1]);
 *             [2] = sticky fault bits (DRGNN_FAULT_*), only ever OR-ed into by the kernels; [3] reserved.
 *   readout   OUT [B, 32*n_branch]; pred OUT [B, O]
 *   head_partials OUT [B][drgnn_head_compact_elems]: [dhid H][dW_fc2 O*H][db_fc2 O][loss][weight]
 *             (dW_fc1 = dhid^T readout is formed by drgnn_step_update)
 *   partials  OUT [B*n_branch][drgnn_net_partial_elems]  (split layout: [B*2][...], one slab per half graph)
 *   xchg      uint64 [B][drgnn_net_step_xchg_elems], zero-filled ONCE by the caller and then left alone: the two
 *             branch workgroups of a GINet graph hand each other their 32 readout values through it, the two half-graph
 *             workgroups of the split layout their pooled rows (may be NULL when n_branch == 1 and no plan with wgs_per_graph == 2 is passed)
 * head->train == 0: inference -- forward + head only (dropout off), writes pred and readout; target,
 * head_partials and partials may be NULL and the step counters are left alone.
 * Needs max_nodes/max_edges/max_c0 bounds; returns DRGNN_E_CAPACITY when a graph of that size does
 * not fit the 160 KiB LDS (drgnn_net_step_lds_bytes): use drgnn_net_forward +
 * drgnn_net_backward_fused_head + drgnn_train_update then. */
/* ---- launch plan of the fused step ---------------------------------------------------------------------------------------
 * Which kernel family and launch layout a fused step of a given shape takes, as ONE function of the launch's description:
 * drgnn_net_step_plan() fills the `out` members from the `in` members, drgnn_net_train_step* and drgnn_train_epoch decide
 * through the same code, so a caller that sizes its buffers from a plan and passes that plan along (drgnn_step_hints.plan)
 * gets the layout it planned or an error, never a silent third one.
 *   family   DRGNN_STEP_FAMILY_AGGREGATE: csrc/drgnn_step2.h (sGAT / FoutNet) / drgnn_step3.h (GINet) -- conv1 starts from the
 *            aggregation tiles the topology builder formed (DRGNN_TOPO_TILES), rows in the builder's hierarchical order
 *            (DRGNN_TOPO_HIER); padded feature widths 16 / 32 / 48 / 64, the reference heads (fc1 width 128 / 64), training and
 *            inference launches.  DRGNN_STEP_FAMILY_NONE: no fused kernel covers the launch -- a graph beyond the LDS budget,
 *            more than 64 features, a head that is not the reference's, a workspace without hierarchical order / tiles (use
 *            drgnn_net_forward + drgnn_net_backward_fused_head).  DRGNN_STEP_FAMILY_PRODUCT (csrc/drgnn_step.h / drgnn_step1.h,
 *            the product-first kernels of rounds 2 - 3): the host emulation build of the CPU test suite only; the device
 *            library does not instantiate them and never reports this family.
 *   wgs_per_graph  GINet (ginet.py:99-141: two branches over the same edge_index): 2 = one workgroup per branch, readouts
 *            exchanged, taken ONLY while all 2 * n_graphs (+ the co-launched builder's) workgroups are resident at once (one
 *            workgroup per CU: HIP promises nothing about dispatch order); 1 = both branches in one workgroup, no cross-workgroup
 *            wait.  sGAT / FoutNet: 2 = the node-split layout (each workgroup owns the depth-1 clusters of one half of the graph;
 *            training launches of the aggregation-first family under the same residency rule); 1 otherwise.
 *   slabs_per_graph  conv gradient slabs per graph in `partials` (what drgnn_step_update must be told)
 *   width    the padded feature width of the specialised kernel instance, 0 = the generic instance
 *   cls      1: the instance with the compile-time LDS layout of the capacity class (200 nodes, 1024 edges, 52 depth-0 clusters
 *            per graph; 32-wide kernels and the 48-wide aggregation-first training kernels) -- the same arithmetic in the same
 *            order, bit-identical results
 *   lean_ok  1: the launch reads nothing a DRGNN_TOPO_LEAN build leaves out
 *   builder_wgs_per_graph  of the topology the same launch builds: 2 / 1; 0 = it gets a launch of its own
 *   lds_bytes  LDS one workgroup needs (<= 160 KiB whenever family != NONE); xchg_words: uint64 exchange words per graph
 *   from_memory  1: graphs beyond the LDS budget of the staged kernels (200 - 270 nodes, by width) -- the instance that reads the
 *            node-sized input tile from memory (L2) where it is used instead of staging it: GINet's one-workgroup kernel with the S
 *            rows of the tiles left in memory, sGAT / FoutNet with the x rows -- and, where that is not enough, the S rows too --
 *            left in memory; run-time LDS layout; every net and width up to ~400 nodes / 2048 edges per graph.  Beyond what the
 *            builder stages an x tile for (drgnn_topology_tiles_ok) the tiles come from drgnn_topology_tiles on the built workspace
 * Overrides (0 = automatic; tests and same-box A/B runs): force_wgs 1 / 2 = always that many workgroups per graph (2 beyond the
 * resident size is MEASUREMENT ONLY: the exchange then leans on in-order dispatch; bounded wait + fault bit); no_class;
 * no_aggregate (never the aggregation-first family: family NONE on the device, the launch pair steps the mini-batch);
 * no_split (sGAT / FoutNet never divided); no_paired (emulation build: the one-workgroup product-first GINet kernel runs
 * branch after branch).  The environment variable DRGNN_STEP_PLAN (comma list of one, two, noclass, product, nosplit, seq;
 * read once) sets the defaults of a process for plans that override nothing. */
#define DRGNN_STEP_FAMILY_NONE 0
#define DRGNN_STEP_FAMILY_PRODUCT 1
#define DRGNN_STEP_FAMILY_AGGREGATE 2
typedef struct drgnn_step_plan {
    /* in: the launch */
    int32_t kind, n_feat, max_nodes, max_edges, max_c0, R, H, O;
    int64_t n_graphs;
    int64_t co_built_graphs;   /* graphs of the topology the same launch builds for the next mini-batch (0: none) */
    int32_t train;             /* 1: training launch; 0: inference */
    int32_t topo_flags;        /* DRGNN_TOPO_* flags of the workspace the launch READS.  DRGNN_TOPO_TILES only if it comes with
                                  tiles of this kind's flavour (sGAT: weighted sums) and x is 16-byte aligned */
    /* in: overrides */
    int32_t force_wgs, no_class, no_aggregate, no_split, no_paired;
    /* out */
    int32_t family, wgs_per_graph, slabs_per_graph, width, cls, lean_ok, builder_wgs_per_graph;
    int64_t lds_bytes, xchg_words;
    int32_t from_memory, reserved;   /* (ABI 4) 1: the instance that leaves the node-sized input tile in memory */
} drgnn_step_plan;
/* Returns wgs_per_graph (0: family NONE).  Host-side only. */
int32_t drgnn_net_step_plan(drgnn_step_plan* plan);

/* Host-known offsets / sizes of the mini-batch's graphs (Batch.from_data_list and the resident set record them): for up to
 * 64 graphs they travel in the kernel arguments, so a workgroup need not fetch them from the workspace
 * first (one dependent memory round trip less).  All members optional. */
typedef struct drgnn_step_hints {
    const int32_t* host_node_ptr; const int32_t* host_edge_ptr;       /* per mini-batch (drgnn_net_train_step) */
    const int64_t* set_node_ptr; const int64_t* set_edge_ptr; const int32_t* host_ids;   /* cached mode */
    /* topo_flags: the DRGNN_TOPO_* flags the topology workspace was BUILT with -- what the caller vouches for (0: a caller
     *             that knows nothing of the hierarchical order: no fused kernel, DRGNN_E_CAPACITY) */
    int32_t topo_flags, reserved;
    /* DRGNN_TOPO_TILES in topo_flags: the aggregation tiles the builder formed for this workspace (DEVICE memory, laid out
     * for the workspace's node count): the aggregation-first kernels start conv1 from them. */
    const float* tiles;
    /* the plan the caller sized `partials` (slabs_per_graph) and `xchg` (xchg_words) from, and its overrides.  NULL: no
     * overrides, and a single-branch net is never divided between two workgroups (one slab per graph).  A launch that
     * cannot take the plan's wgs_per_graph returns DRGNN_E_CAPACITY. */
    const drgnn_step_plan* plan;
    /* cached-topology launches (drgnn_net_train_step_cached) only, optional: the graph numbers of the NEXT mini-batch (DEVICE
     * memory, n_next of them).  While the launch leaves CUs idle, one extra workgroup per such graph requests everything its
     * step will stage, so that the next launch finds it in the L2 of the XCD that steps it (no effect on any result; numbers
     * outside the cached set are skipped). */
    const int32_t* next_ids; int64_t n_next;
} drgnn_step_hints;
/* uint64 exchange words per graph the fused step needs for these bounds (GINet: n_branch x max(H, 32); the split layout
 * of sGAT / FoutNet: two hand-offs of max_c0 x 16 values per half + the partial readouts) */
int64_t drgnn_net_step_xchg_elems(int32_t kind, int32_t max_nodes, int32_t max_c0, int32_t H);
int64_t drgnn_net_step_lds_bytes(int32_t kind, int32_t n_feat, int32_t max_nodes, int32_t max_edges,
                                 int32_t max_c0, int32_t R, int32_t H, int32_t O);
int64_t drgnn_head_compact_elems(int32_t R, int32_t H, int32_t O);
/* Which instantiation of the PRODUCT-FIRST step kernel (emulation build; kept in the ABI for hosts that query it) a launch of
 * these bounds would take: the padded feature width (16/32/48/64) of the width-specialised kernel, or 0 for the generic
 * one.  The device library's launches are described by drgnn_net_step_plan (width / cls).  Host-side only. */
int32_t drgnn_net_step_variant(int32_t kind, const float* x, int32_t n_feat, int32_t max_nodes, int32_t max_edges,
                               int32_t max_c0, int32_t H, int32_t O);
int drgnn_net_train_step(const drgnn_net_desc* net, const drgnn_head_desc* head, const float* x,
                         const void* target, int32_t* step2, const int32_t* ws_i32, const float* ws_f32,
                         int64_t n_nodes, int64_t n_edges, int64_t n_graphs, int32_t max_nodes,
                         int32_t max_edges, int32_t max_c0, float* pred, float* readout,
                         float* head_partials, float* partials, uint64_t* xchg,
                         const drgnn_topology_request* next_topology /* optional */,
                         const drgnn_step_hints* hints /* optional */, void* stream);
/* drgnn_train_update for the slabs of drgnn_net_train_step: also forms dW_fc1 from head_partials'
 * dhid rows and readout, applies Adam with step index step2[1] and commits step2[0] = step2[1]
 * (also when apply_adam = 0: data parallel, all-reduce + drgnn_adam_step(step2) follow). */
/* slabs_per_graph: conv slabs per graph in conv_partials -- 0 = n_branch (every layout but one); 2 for the split layout of a
 * single-branch net (drgnn_step_plan.slabs_per_graph of the plan the launch was given). */
int drgnn_step_update(const drgnn_net_desc* net, const float* conv_partials, int64_t n_graphs,
                      drgnn_conv_grads* g_conv1, drgnn_conv_grads* g_conv2, const float* head_partials,
                      const float* readout, int32_t R, int32_t H, int32_t O, int64_t head_offset,
                      float* flat_param, float* flat_grad, float* exp_avg, float* exp_avg_sq, int64_t n_param,
                      int32_t* step2, float* loss, float lr, float beta1, float beta2, float eps,
                      int32_t apply_adam, int32_t slabs_per_graph, void* stream);

/* The gradient half of drgnn_step_update alone, for a caller whose optimiser lives outside the library (the drop-in boundary:
 * an unchanged NeuralNet._epoch runs torch.optim.Adam over model.parameters(), NeuralNet.py:183-184,502-503): fixed-order sum of
 * the conv slabs into g_conv1 / g_conv2 and of the head slabs (+ dW_fc1 = dhid^T readout) into head_grad = [fc1.weight |
 * fc1.bias | fc2.weight | fc2.bias] (contiguous), no parameter update.
 *   graph_weight  NULL, or float [n_graphs]: slab g enters the sum multiplied by graph_weight[g].  Every quantity of a graph's
 *                 slab is linear in d loss / d pred_g, so a DRGNN_TASK_GRAD launch with O = 1 and an upstream gradient of ONE
 *                 leaves d pred_g / d theta in the slabs, and this call with graph_weight = d loss / d pred contracts them with
 *                 the real gradient of ANY loss: model(batch) is one launch, loss.backward() is this one.
 *   zero_ptr / zero_len  up to DRGNN_ZERO_RANGES float ranges filled with 0 by the same launch (parameters the step kernels
 *                 never touch: GINetConvLayer's fc_attention / fc_edge_attr, whose gradient is identically 0, ginet.py:63-66)
 *   step2         optional: commit step2[0] = step2[1] (a new dropout stream for the next step), as drgnn_step_update does. */
#define DRGNN_ZERO_RANGES 8
int drgnn_step_gradients(const drgnn_net_desc* net, const float* conv_partials, int64_t n_graphs,
                         drgnn_conv_grads* g_conv1, drgnn_conv_grads* g_conv2, const float* head_partials,
                         const float* readout, int32_t R, int32_t H, int32_t O, float* head_grad,
                         const float* graph_weight, float* const* zero_ptr, const int64_t* zero_len, int32_t n_zero,
                         int32_t* step2, int32_t slabs_per_graph, void* stream);

/* ---- graclus ---------------------------------------------------------------------
 * Greedy maximal matching of every graph of a built topology (CSR0), the clustering the README's custom net
 * feeds to max_pool / max_pool_x (README.md:98-126; torch_geometric.nn.graclus -> torch_cluster, whose node
 * visiting order is a random permutation: pass it as `perm`, LOCAL ids per graph, to reproduce a given run;
 * NULL = identity).  An unmatched node takes its unmatched neighbour of largest weight (weight [E] indexed by
 * input edge id; NULL or ties: first in edge-id order); both are labelled min(u, v).  cluster [N] int64 receives
 * batch-global labels.  max_nodes / max_edges size the LDS carve (DRGNN_E_CAPACITY beyond 160 KiB). */
int drgnn_graclus(const int32_t* ws_i32, int64_t n_nodes, int64_t n_edges, int64_t n_graphs, int32_t max_nodes,
                  int32_t max_edges, const float* weight, const int64_t* perm, int64_t* cluster, void* stream);

/* ---- offline clustering --------------------------------------------------------
 * Markov clustering of every graph of a batch: community_detection(edge_index, num_nodes,
 * method='mcl') (community_pooling.py:95-158; markov_clustering.run_mcl defaults), as PreCluster
 * runs it on the internal-contact graph (DataSet.py:77-86).  One workgroup per graph, dense fp64.
 * mat_ptr int64 [B+1] = prefix sums of N_g^2; mat_scratch 3 * mat_ptr[B] doubles; int_scratch
 * 4 * N ints; labels int64 [N] (per-graph cluster ids, the reference's numbering); info int32 [B]
 * = iterations used (negative: no convergence within 100). */
int drgnn_mcl(const int64_t* edge_index, int64_t n_edges, const int32_t* node_ptr, const int32_t* edge_ptr,
              const int64_t* mat_ptr, int64_t n_graphs, double* mat_scratch, int32_t* int_scratch,
              int64_t* labels, int32_t* info, void* stream);

/* ---- device-resident graph set and mini-batch assembly --------------------
 * Replaces the host collate of every mini-batch: torch_geometric DataLoader -> Batch.from_data_list over
 * HDF5DataSet.load_one_graph's Data objects (NeuralNet.py:153-154, DataSet.py:231-366).  The set is the
 * whole dataset uploaded once, graph-major: x [sumN, F]; edge_index [2, sumE] with LOCAL node ids (the
 * per-graph tensors of DataSet.py:266-269 back to back, row block then column block); edge_attr [sumE];
 * cluster0 [sumN]; cluster1 [sumC0]; y [G] (4- or 8-byte elements); int64 offset tables [G+1].
 * drgnn_collate assembles the mini-batch ids[0..B) (graph numbers, any order, repeats allowed) exactly as
 * the PyG collate does: x / cluster0 / cluster1 / edge_attr concatenated, both rows of edge_index shifted by
 * the slot's node offset, batch[n] = slot, y gathered; plus the int32 per-slot offset tables node_ptr /
 * edge_ptr / c1_ptr [B+1] that drgnn_topology_build and the step kernels take.  The caller sizes the
 * outputs from its host copy of the tables (N = sum of the selected node counts, ...).  One workgroup per
 * slot; ids outside [0, G) select an empty graph. */
int drgnn_collate(const drgnn_graph_set* set, const int32_t* ids, int64_t n_graphs, int64_t n_nodes,
                  int64_t n_edges, float* x, int64_t* edge_index, float* edge_attr, int64_t* batch,
                  int64_t* cluster0, int64_t* cluster1, void* y, int32_t* node_ptr, int32_t* edge_ptr,
                  int32_t* c1_ptr, void* stream);

/* ---- one training epoch, driven natively --------------------------------------------
 * The body of NeuralNet._epoch's loop (NeuralNet.py:486-506) for EVERY mini-batch of an epoch, enqueued on
 * `stream` without touching the host language in between and without any synchronisation: for mini-batch k
 *     drgnn_collate(k+1)  ->  drgnn_net_train_step(k)  [+ topology of k+1 in the same launch]  ->  drgnn_step_update
 * -- in fact without the collate launch: the topology builder of mini-batch k+1 (extra workgroups of step k's
 * launch) reads the graphs' index data in place from the resident set and gathers their node features and
 * targets into the slot's compact buffers (drgnn_topology_request, resident-set mode); two mini-batch slots are
 * carved out of `scratch` (k trains from one while k+1 is assembled in the other).
 * ids / host_ids: the epoch's visiting order (graph numbers of `set`) on the device and on the host; mini-batch k
 * = ids[k*batch_size, min((k+1)*batch_size, n_ids)).  host_*_ptr: host copies of the set's offset tables (they size
 * every launch).  Outputs: pred [n_ids, O] in visiting order, losses [ceil(n_ids / batch_size)] (mean loss of each
 * mini-batch, what the reference accumulates with loss.item()).  Model / optimiser arguments as for
 * drgnn_net_train_step / drgnn_step_update.  Returns DRGNN_E_CAPACITY when some graph does not fit the fused step
 * kernel's or the topology builder's LDS budget (the caller then steps mini-batch by mini-batch).
 * drgnn_train_epoch_scratch_bytes: bytes of device scratch the plan needs (< 0: error code). */
/* ---- cached topology (declared mode) -----------------------------------------------------------------
 * Per-graph topology does not depend on the mini-batch (all ids inside a graph's segment are local), so a
 * resident graph set can build it ONCE: one topology workspace over the whole set (drgnn_topology_build_request
 * in resident-set mode with ids = 0..G-1 and the set's own offset tables), kept next to the set's node features
 * and targets.  A mini-batch is then just a list of graph numbers: the fused step reads graph ids[g]'s segments
 * of the cached workspace in place -- no builder workgroups, no per-step index work at all.
 * (The reference redoes this work in every forward pass: community_pooling.py:25-30,197-201; "rebuilt every
 * step" stays the default mode of this library and of bench.py's headline.) */
typedef struct drgnn_topology_cache {
    int64_t n_graphs, n_nodes, n_edges;          /* of the WHOLE set = the shape the workspace was laid out for */
    const int32_t* ws_i32; const float* ws_f32;  /* ws_f32 may be NULL (nets without edge weights) */
    const float* x;                              /* [n_nodes, F] node features, graph-major (the set's x) */
    const void* y; int32_t y_bytes, flags;       /* [n_graphs] targets: 4 = float32, 8 = int64; may be NULL (inference);
                                                    flags: the DRGNN_TOPO_* flags the workspace was built with */
    const float* tiles;                          /* flags & DRGNN_TOPO_TILES: the set's aggregation tiles (n_nodes rows), else NULL */
} drgnn_topology_cache;
/* drgnn_net_train_step over the graphs ids[0..n_graphs) of a cached set: same outputs, same arithmetic (slot g
 * of every output = graph ids[g]); max_* bound the graphs of THIS mini-batch. */
int drgnn_net_train_step_cached(const drgnn_net_desc* net, const drgnn_head_desc* head,
                                const drgnn_topology_cache* cache, const int32_t* ids, int64_t n_graphs,
                                int32_t max_nodes, int32_t max_edges, int32_t max_c0, int32_t* step2, float* pred,
                                float* readout, float* head_partials, float* partials, uint64_t* xchg,
                                const drgnn_step_hints* hints /* optional */, void* stream);

/* Data-parallel hook of the native epoch loop: called on the host once per mini-batch, after the launches that leave
 * this rank's gradient of mini-batch `batch_index` in flat_grad have been ENQUEUED on `stream`; it must enqueue the
 * exchange (one all-reduce of flat_grad, weighted n_local / n_global) on the same stream and return 0.  The loop then
 * enqueues Adam (drgnn_adam_step).  No host synchronisation is implied. */
typedef int (*drgnn_exchange_fn)(void* user, int64_t batch_index, int64_t n_local, void* stream);

typedef struct drgnn_epoch_plan {
    const drgnn_graph_set* set;
    const int64_t* host_node_ptr; const int64_t* host_edge_ptr; const int64_t* host_c1_ptr;
    const int32_t* ids; const int32_t* host_ids; int64_t n_ids;
    int32_t batch_size, need_weights;
    int32_t inference, reserved;   /* inference != 0: forward + head only (dropout off), pred is the only output;
                                      the optimiser members, step2 excepted, and the set's targets may be NULL */
    const drgnn_net_desc* net; const drgnn_head_desc* head;
    drgnn_conv_grads* g_conv1; drgnn_conv_grads* g_conv2;
    int64_t head_offset;
    float* flat_param; float* flat_grad; float* exp_avg; float* exp_avg_sq; int64_t n_param;
    int32_t* step2;
    float lr, beta1, beta2, eps;
    /* cached-topology mode (non-NULL): mini-batches are stepped straight out of the cache, the loop issues no
     * offset-table, builder or gather work; `set` is then only consulted for the graphs' sizes on the host */
    const drgnn_topology_cache* cache;
    /* data parallel (non-NULL): per mini-batch  gradient launches -> exchange(...) -> Adam launch  instead of the fused
     * reduce+Adam launch; every rank must run the same number of mini-batches */
    drgnn_exchange_fn exchange; void* exchange_user;
    /* optional: the override members of this plan apply to every step launch of the loop (tests, A/B runs) */
    const drgnn_step_plan* step_overrides;
    /* optional: one more destination of the LAST mini-batch's loss (its update launch writes losses[n_batches - 1] and this
     * word): a trainer's fixed "loss of the last step" word stays current without a copy behind the epoch */
    float* last_loss;
} drgnn_epoch_plan;
int64_t drgnn_train_epoch_scratch_bytes(const drgnn_epoch_plan* plan);
int drgnn_train_epoch(const drgnn_epoch_plan* plan, void* scratch, int64_t scratch_bytes, float* pred,
                      float* losses, void* stream);

/* ---- one-shot all-reduce over peer-mapped exchange buffers (csrc/drgnn_p2p.h) ------------------------------
 * Data-parallel exchange of the flat gradient without a ring: each rank owns a fine-grained exchange buffer
 * (drgnn_p2p_alloc), hands its 64-byte IPC handle to the peers (e.g. torch.distributed.all_gather), maps theirs
 * (drgnn_p2p_open) and then every step  drgnn_allreduce_oneshot  = publish own weighted vector, read all W vectors
 * over xGMI, add them in rank order.  One launch, no host work, hipGraph-capturable; sums are bit-identical on all
 * ranks.  `seq` = DRGNN_P2P_SEQ_WORDS zero-initialised uint32 device words, `status` one int32 device word (non-zero
 * after a wait that expired: a peer did not arrive).  world <= DRGNN_P2P_MAX_RANKS. */
#define DRGNN_P2P_MAX_RANKS 16
#define DRGNN_P2P_SEQ_WORDS 16
int64_t drgnn_p2p_bytes(int64_t n_floats);
int drgnn_p2p_alloc(int64_t bytes, void** dev_ptr, void* ipc_handle_64);     /* zero-filled, fine-grained */
int drgnn_p2p_open(const void* ipc_handle_64, void** dev_ptr);
int drgnn_p2p_close(void* dev_ptr);
int drgnn_p2p_free(void* dev_ptr);
/* part: 0 = the whole exchange (product); 1 = publish only, 2 = consume only (single-process protocol tests) */
int drgnn_allreduce_oneshot(float* grad, int64_t n, void* const* peer_bufs, int32_t world, int32_t rank,
                            float weight, uint32_t* seq, int32_t* status, int32_t part, void* stream);

int drgnn_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* DRGNN_H */
