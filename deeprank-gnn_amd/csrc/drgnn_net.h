// drgnn_net.h -- fused network body: one workgroup per (graph, branch).
//
//   conv1 -> relu -> cluster max (depth 0) -> conv2 -> relu -> cluster max (depth 1) -> mean
//
// "conv" is the common form of the three reference layers
//       z_i = s_i * (x_i Wself) + sum_{e: row(e)=i} c_e * (x_col(e) Wnbr) + b
//   GINet   s = 0          c_e = 1                       no bias     (ginet.py:50-73, alpha==1)
//   sGAT    s_i = mean a   c_e = a_e / max(deg_i,1)      bias        (sGAT.py:62-93)
//   FoutNet s = 1          c_e = 1 / deg_i (NaN if 0)    bias        (foutnet.py:56-82)
//
// Data movement: every input of a graph (its x tile, CSR/CSC, member lists, weights, saved
// activations) is staged into LDS by ONE burst of independent coalesced loads at kernel
// start; all later phases touch LDS only, and only what autograd must keep (pooled
// features, argmax indices, readout / per-graph weight-gradient partials) goes back to HBM.
// The dense products run on the f32 MFMA (v_mfma_f32_16x16x4_f32, exact fp32) with operands
// read from padded LDS rows; neighbour aggregation is a CSR gather with a fixed summation
// order; cluster max keeps the first maximum in ascending member order (torch_scatter CPU
// tie rule).
#pragma once
#include "drgnn_topology.h"

#define DRGNN_H1 16
#define DRGNN_H2 32
#define DRGNN_W1LD (DRGNN_H1 + 1)   // padded LDS row strides of the staged weights
#define DRGNN_W2LD (DRGNN_H2 + 1)

#ifdef DRGNN_EMU
#define DRGNN_NAN (NAN)
#define DRGNN_NEG_INF (-INFINITY)
#else
#define DRGNN_NAN (__builtin_nanf(""))
#define DRGNN_NEG_INF (-__builtin_inff())
#endif

// ---------------------------------------------------------------------------------
// Workgroup GEMM  C(i,j) = sum_k A(i,k) B(k,j),  i<M, j<N, k<K, fully strided operands.
// gfx950: (16x16 output tile, K slice) units are spread over the 4 waves; K is consumed 4 at
// a time by v_mfma_f32_16x16x4_f32 (A: lane -> row l&15, k l>>4;  B: lane -> k l>>4, col
// l&15;  D: col l&15, rows 4*(l>>4)+r), operands of 8 steps are fetched before the MFMA
// chain so their latencies overlap.  With KS > 1 the K range is cut in KS slices whose
// partial tiles go to `part` ([KS][M][N] floats) and are summed in slice order afterwards
// (deterministic); callers put a BARRIER before using C.
// ---------------------------------------------------------------------------------
#ifdef DRGNN_EMU
DEV void wg_gemm(int M, int N, int K, const float* A, int sai, int sak, const float* B, int sbk,
                 int sbj, float* C, int sci, int scj, int KS = 1, float* part = nullptr) {
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < N; ++j) {
            float acc = 0.0f;
            for (int k = 0; k < K; ++k) acc = fmaf(A[i * sai + k * sak], B[k * sbk + j * sbj], acc);
            C[i * sci + j * scj] = acc;
        }
}
#else
typedef float drgnn_f32x4 __attribute__((ext_vector_type(4)));
DEV void wg_gemm(int M, int N, int K, const float* A, int sai, int sak, const float* B, int sbk,
                 int sbj, float* C, int sci, int scj, int KS = 1, float* part = nullptr) {
    // wave id as a scalar so that the unit loop below is uniform control flow (SALU only)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int lr = lane & 15, lq = lane >> 4;
    const int kslice = (KS == 1) ? ((K + 3) & ~3) : ((((K + KS - 1) / KS) + 3) & ~3);
    int unit = 0;      // (tile, K slice) units are dealt round-robin to the waves; all loop
                       // variables are wave-uniform, no per-lane integer division
    for (int i0 = 0; i0 < M; i0 += 16) {
        for (int j0 = 0; j0 < N; j0 += 16) {
            for (int ks = 0; ks < KS; ++ks, ++unit) {
                if ((unit & (DRGNN_NWAVES - 1)) != wave) continue;
                const int ai = i0 + lr, bj = j0 + lr;
                const bool a_ok = ai < M, b_ok = bj < N;
                const float* ap = A + (long)ai * sai;
                const float* bp = B + (long)bj * sbj;
                const int kbeg = ks * kslice, kend = imin(K, kbeg + kslice);
                drgnn_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                for (int k0 = kbeg; k0 < kend; k0 += 32) {
                    float a[8], b[8];
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        const int k = k0 + 4 * s + lq;
                        const bool k_ok = k < kend;
#if defined(DRGNN_SKIP) && DRGNN_SKIP == 31
                        a[s] = 1.0f; b[s] = 1.0f;
#else
                        a[s] = (a_ok && k_ok) ? ap[(long)k * sak] : 0.0f;
                        b[s] = (b_ok && k_ok) ? bp[(long)k * sbk] : 0.0f;
#endif
                    }
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
#if defined(DRGNN_SKIP) && DRGNN_SKIP == 30
                        acc[0] += a[s] * b[s];
#else
                        if (k0 + 4 * s < kend) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc, 0, 0, 0);
#endif
                    }
                }
#if defined(DRGNN_SKIP) && DRGNN_SKIP == 32
                if (b_ok && acc[0] == 12345.0f) {
#else
                if (b_ok) {
#endif
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ci = i0 + lq * 4 + r;
                        if (ci < M) {
                            if (KS == 1) C[(long)ci * sci + (long)bj * scj] = acc[r];
                            else part[((long)ks * M + ci) * N + bj] = acc[r];
                        }
                    }
                }
            }
        }
    }
    if (KS > 1) {
        __syncthreads();
        const FastDiv dn = fastdiv_make(N);
        for (int e = threadIdx.x; e < M * N; e += DRGNN_NTHREADS) {
            float s = 0.0f;
            for (int ks = 0; ks < KS; ++ks) s += part[(long)ks * M * N + e];
            const int i = fastdiv(dn, e);
            C[(long)i * sci + (long)fastmod(dn, e, i) * scj] = s;
        }
    }
}
#endif

// ---------------------------------------------------------------------------------
// Per-graph FC head + loss + its backward, run INSIDE the body-backward workgroups (the head
// is row-wise: pred_g = fc2(dropout(relu(fc1(readout_g)))) and d loss / d pred_g only needs
// the batch size), so a training step needs no separate head launch.  Both branch workgroups
// of a graph evaluate it redundantly (8k MACs); branch 0 alone writes predictions and the
// head's weight-gradient partial slab  [dW1 H*R][db1 H][dW2 O*H][db2 O][loss][weight].
// ---------------------------------------------------------------------------------
#define DRGNN_MAX_OUT 16
#define DRGNN_TASK_REG 0
#define DRGNN_TASK_CLASS 1

// lowbias32-style counter hash -> uniform 32-bit value for (seed, step, element)
HD uint32_t drgnn_hash(uint32_t seed, uint32_t step, uint32_t idx) {
    uint32_t h = seed ^ (step * 0x9E3779B9u) ^ (idx * 0x85EBCA6Bu + 0xC2B2AE35u);
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return h;
}
HD int64_t head_partial_floats(int R, int H, int O) { return (int64_t)H * R + H + (int64_t)O * H + O + 2; }

struct HeadFused {
    int enabled;
    int B, R, H, O, task;
    float p_drop; uint32_t seed; int step_bias;
    const float* w1; const float* b1; const float* w2; const float* b2; const float* class_w;
    const float* y_reg; const int64_t* y_cls;
    const float* readout;        // [B, R] (forward output)
    const int32_t* step;
    float* pred;                 // [B, O]
    float* partials;             // [B][head_partial_floats]
    int stage;                   // 1: the launch reserved LDS for W1/b1/W2/b2/readout row
    int train;                   // fused step only: 0 = inference (predictions, no loss / backward / slabs)
    int sigmoid;                 // regression: pred = sigmoid(output) before the loss (NeuralNet.py:625)
    const float* drop_mask;      // [B][H] 0 / 1 or null: explicit dropout mask instead of the hash stream (parity tests)
};
// dropout decision of hidden unit h of graph g (ginet.py:138): the given mask, else the counter hash against `thresh`
DEV bool drgnn_keep(const HeadFused& hf, uint32_t step, int g, int H, int h, uint32_t thresh) {
    const uint32_t idx = (uint32_t)(g * H + h);
    if (__builtin_expect(hf.drop_mask != nullptr, 0)) return hf.drop_mask[idx] != 0.0f;
    return drgnn_hash(hf.seed, step, idx) >= thresh;
}
DEV float drgnn_sigmoid(float v) { return 1.0f / (1.0f + expf(-v)); }
HD int64_t head_stage_words(int R, int H, int O) { return (int64_t)H * (R + 1) + H + (int64_t)O * H + O + R + 16; }

// scratch `gp`: >= 1024 + H + R + 2*DRGNN_MAX_OUT floats; dr_out: this branch's 32 columns
// w1/b1/w2/b2/xrow: either the model's tensors in global memory (ldw = R) or their LDS copies
// made by the prologue burst (ldw = R + 1, padded rows)
DEV void head_graph(const HeadFused& hf, int g, int br, float* gp, float* dr_out, const float* w1, int ldw,
                    const float* b1, const float* w2, const float* b2, const float* xrow) {
    const int R = hf.R, H = hf.H, O = hf.O;
    float* tmp = gp;
    float* hid = gp + 1024;
    float* xr = hid + H;
    float* outs = xr + R;
    float* douts = outs + DRGNN_MAX_OUT;
    const uint32_t step = hf.step ? (uint32_t)(hf.step[0] + hf.step_bias) : 0u;
    const float keep_scale = (hf.p_drop > 0.0f) ? 1.0f / (1.0f - hf.p_drop) : 1.0f;
    const double pt = (double)hf.p_drop * 4294967296.0;
    const uint32_t thresh = (hf.p_drop > 0.0f) ? (uint32_t)(pt > 4294967295.0 ? 4294967295.0 : pt) : 0u;
    float* part = hf.partials + (long)g * head_partial_floats(R, H, O);
    float* p_w1 = part;
    float* p_b1 = p_w1 + (long)H * R;
    float* p_w2 = p_b1 + H;
    float* p_b2 = p_w2 + (long)O * H;
    float* p_loss = p_b2 + O;

    FOR_TID(r, R) { xr[r] = xrow[r]; }
    BARRIER();
    {   // fc1: 8 lanes per hidden unit, each a contiguous slice of the row (coalesced W1 read)
        const int per = (R + 7) >> 3;
        FOR_TID(t, H * 8) {
            const int h = t >> 3, q = t & 7;
            const int lo = q * per, hi = imin(R, lo + per);
            float acc = 0.0f;
            for (int r = lo; r < hi; ++r) acc = fmaf(w1[h * ldw + r], xr[r], acc);
            tmp[t] = acc;
        }
    }
    BARRIER();
    FOR_TID(h, H) {
        float v = b1[h];
        for (int q = 0; q < 8; ++q) v += tmp[h * 8 + q];
        v = v > 0.0f ? v : 0.0f;
        if (thresh) v = drgnn_keep(hf, step, g, H, h, thresh) ? v * keep_scale : 0.0f;
        hid[h] = v;
    }
    BARRIER();
    FOR_TID(t, O * 16) {
        const int o = t >> 4, q = t & 15;
        float acc = 0.0f;
        for (int h = q; h < H; h += 16) acc = fmaf(hid[h], w2[o * H + h], acc);
        tmp[t] = acc;
    }
    BARRIER();
    FOR_TID(i, 1) {
        const bool sig = hf.sigmoid && hf.task == DRGNN_TASK_REG;
        for (int o = 0; o < O; ++o) {
            float acc = b2[o];
            for (int q = 0; q < 16; ++q) acc += tmp[o * 16 + q];
            if (sig) acc = drgnn_sigmoid(acc);
            outs[o] = acc;
            if (br == 0) hf.pred[(long)g * O + o] = acc;
        }
        float loss = 0.0f, wsum = 1.0f;
        if (hf.task == DRGNN_TASK_REG) {
            const float inv = 1.0f / (float)(hf.B * O);
            for (int o = 0; o < O; ++o) {
                const float d = outs[o] - hf.y_reg[g];
                loss += d * d * inv;
                douts[o] = 2.0f * d * inv * (sig ? outs[o] * (1.0f - outs[o]) : 1.0f);
            }
        } else {
            float denom = 0.0f;
            for (int q = 0; q < hf.B; ++q) denom += hf.class_w ? hf.class_w[hf.y_cls[q]] : 1.0f;
            const int yc = (int)hf.y_cls[g];
            const float wy = hf.class_w ? hf.class_w[yc] : 1.0f;
            float mx = outs[0];
            for (int o = 1; o < O; ++o) mx = outs[o] > mx ? outs[o] : mx;
            float se = 0.0f;
            for (int o = 0; o < O; ++o) se += expf(outs[o] - mx);
            const float lse = logf(se) + mx;
            loss = wy * (lse - outs[yc]) / denom;
            for (int o = 0; o < O; ++o) douts[o] = wy * (expf(outs[o] - lse) - (o == yc ? 1.0f : 0.0f)) / denom;
            wsum = wy;
        }
        if (br == 0) { p_loss[0] = loss; p_loss[1] = wsum; }
    }
    BARRIER();
    FOR_TID(h, H) {
        float acc = 0.0f;
        for (int o = 0; o < O; ++o) acc = fmaf(douts[o], w2[o * H + h], acc);
        const float hv = hid[h];
        const float d = (hv != 0.0f) ? acc * keep_scale : 0.0f;     // relu' and dropout mask
        if (br == 0) {
            for (int o = 0; o < O; ++o) p_w2[(long)o * H + h] = douts[o] * hv;
            p_b1[h] = d;
        }
        hid[h] = d;
    }
    if (br == 0) { FOR_TID(o, O) { p_b2[o] = douts[o]; } }
    BARRIER();
    if (br == 0) {
        const FastDiv dR = fastdiv_make(R);
        FOR_TID(e, H * R) {
            const int h = fastdiv(dR, e);
            p_w1[e] = hid[h] * xr[fastmod(dR, e, h)];
        }
    }
    FOR_TID(t, DRGNN_H2 * 32) {     // d readout, this branch's 32 columns: 32 partial sums per column
        const int r = t & 31, q = t >> 5;
        float acc = 0.0f;
        for (int h = q; h < H; h += 32) acc = fmaf(hid[h], w1[h * ldw + br * DRGNN_H2 + r], acc);
        tmp[t] = acc;
    }
    BARRIER();
    FOR_TID(r, DRGNN_H2) {
        float acc = 0.0f;
        for (int q = 0; q < 32; ++q) acc += tmp[q * 32 + r];
        dr_out[r] = acc;
    }
    BARRIER();
}

// ---- per-launch description ----------------------------------------------------------
struct NetArgs {
    drgnn_net_desc net;
    const float* x;          // [Ntot, F]
    TopoView tv;
    int64_t n_nodes;         // Ntot
    int n_graphs;
    // forward outputs / backward inputs (padded per-graph layout)
    float* xp;               // [n_branch][Ntot][16]
    int32_t* arg0;           // [n_branch][Ntot][16]
    int32_t* arg1;           // [n_branch][Ntot][32]
    float* readout;          // [B][32*n_branch]
    // backward
    const float* grad_readout;
    float* partials;         // [B*n_branch][P]
    float* grad_x;           // [n_branch][Ntot][F] or null (summed over branches by the reducer)
    int n_partial;           // P
    int32_t* step_inc;       // optional optimiser step counter, incremented once per launch
    HeadFused hf;            // backward: per-graph FC head + loss instead of grad_readout
};

// ---- scratch (LDS, or a global slab for graphs that do not fit) -------------------------
// One carve serves forward and backward; sizes in 4-byte words.  capE bounds the edges of a
// graph, capC its number of depth-0 clusters.
struct NetScratch {
    // staged inputs
    float* xs;                 // [capN][F+1]          x tile, padded rows
    float* wn1; float* ws1; float* b1;   // [F][17] x2, [16]
    float* wn2; float* ws2; float* b2;   // [16][33] x2, [32]
    int* rp0; int* ix0; int* ts0;        // [capN+1], [capE], [capE]  CSR0 (fwd) / CSC0 (bwd)
    float* ew0;                          // [capE] edge weights in CSR0 slot order (sGAT)
    int* dg0;                            // [capN+1] CSR0 rowptr when ix0 holds the CSC (bwd)
    int* mp0; int* mem0;                 // [capN+1], [capN]
    int* rp1; int* ix1; int* ts1;        // [capC+1], [capE], [capE]
    float* ew1;                          // [capE]
    int* dg1;                            // [capC+1]
    int* mp1; int* mem1;                 // [capC+1], [capC]
    int* a0; int* a1;                    // [capC][16], [capC][32]   saved argmax (bwd)
    // compute
    float* u1;    // [capN][hc1]   x W1 (nbr | self)      bwd: dU1
    float* z1;    // [capN][16]    relu(conv1)            bwd: dZ1
    float* dv0;   // [capN]
    float* sc0;   // [capN]
    float* xp;    // [capC][16]    pooled features        bwd: saved xp
    float* dxp;   // [capC][16]    bwd: dXP (nbr part)
    float* u2;    // [capC][hc2]                           bwd: dU2
    float* z2;    // [capC][32]                            bwd: dZ2
    float* p2;    // [capC][32]    depth-1 pooled          bwd: dXP (self part)
    float* dv1;   // [capC]
    float* sc1;   // [capC]
    float* gp;    // [2048]        K-split GEMM partials
    float* misc;  // [64]
    float* end;   // first word after the carve (head staging area of the fused backward)
};

// Forward and backward stage different subsets; `bwd` selects the carve.  Keep the two
// functions below in step: net_scratch_words() is what the host sizes LDS with.
#define NET_CARVE_LIST(X)                                                                      \
    X(xs, (long)capN * (F + 1), 1)                                                             \
    X(wn1, F * DRGNN_W1LD, 1)                                                                  \
    X(ws1, F * DRGNN_W1LD, !gin)                                                               \
    X(b1, DRGNN_H1, !gin)                                                                      \
    X(wn2, DRGNN_H1 * DRGNN_W2LD, 1)                                                           \
    X(ws2, DRGNN_H1 * DRGNN_W2LD, !gin)                                                        \
    X(b2, DRGNN_H2, !gin)                                                                      \
    X(rp0, capN + 1, 1)                                                                        \
    X(ix0, capE, 1)                                                                            \
    X(ts0, capE, bwd && sg)                                                                    \
    X(ew0, capE, sg)                                                                           \
    X(dg0, capN + 1, bwd && !gin)                                                              \
    X(mp0, capN + 1, !bwd)                                                                     \
    X(mem0, capN, !bwd)                                                                        \
    X(rp1, capC + 1, 1)                                                                        \
    X(ix1, capE, 1)                                                                            \
    X(ts1, capE, bwd && sg)                                                                    \
    X(ew1, capE, sg)                                                                           \
    X(dg1, capC + 1, bwd && !gin)                                                              \
    X(mp1, capC + 1, !bwd)                                                                     \
    X(mem1, capC, !bwd)                                                                        \
    X(a0, (long)capC * DRGNN_H1, bwd)                                                          \
    X(a1, (long)capC * DRGNN_H2, bwd)                                                          \
    X(u1, (long)capN * hc1, 1)                                                                 \
    X(z1, (long)capN * DRGNN_H1, 1)                                                            \
    X(dv0, capN, !gin)                                                                         \
    X(sc0, capN, !gin)                                                                         \
    X(xp, (long)capC * DRGNN_H1, 1)                                                            \
    X(dxp, (long)capC * DRGNN_H1, bwd)                                                         \
    X(u2, (long)capC * hc2, 1)                                                                 \
    X(z2, (long)capC * DRGNN_H2, 1)                                                            \
    X(p2, (long)capC * DRGNN_H2, (!bwd) || (!gin))                                             \
    X(dv1, capC, !gin)                                                                         \
    X(sc1, capC, !gin)                                                                         \
    X(gp, 2048, bwd)                                                                           \
    X(misc, 64, 1)

HD int64_t net_scratch_words(int kind, int64_t F, int64_t capN, int64_t capE, int64_t capC, int bwd) {
    const int64_t hc1 = (kind == DRGNN_GINET) ? DRGNN_H1 : 2 * DRGNN_H1;
    const int64_t hc2 = (kind == DRGNN_GINET) ? DRGNN_H2 : 2 * DRGNN_H2;
    const int sg = (kind == DRGNN_SGAT) ? 1 : 0;
    const int gin = (kind == DRGNN_GINET) ? 1 : 0;
    int64_t w = 0;
#define X(name, words, cond) w += (cond) ? (int64_t)(words) : 0;
    NET_CARVE_LIST(X)
#undef X
    return w + 16;
}

DEV NetScratch net_carve(float* base, int kind, int F, int capN, int capE, int capC, int bwd) {
    const int hc1 = (kind == DRGNN_GINET) ? DRGNN_H1 : 2 * DRGNN_H1;
    const int hc2 = (kind == DRGNN_GINET) ? DRGNN_H2 : 2 * DRGNN_H2;
    const int sg = (kind == DRGNN_SGAT) ? 1 : 0;
    const int gin = (kind == DRGNN_GINET) ? 1 : 0;
    NetScratch s;
    float* p = base;
#define X(name, words, cond) s.name = (decltype(s.name))p; p += (cond) ? (long)(words) : 0;
    NET_CARVE_LIST(X)
#undef X
    s.end = p;
    return s;
}

struct GraphDims { int n0, N, e0, E, C, E1, C1, rowbase; };

// workgroup-uniform value -> scalar register (the compiler cannot prove these loads uniform itself)
#ifdef DRGNN_EMU
#define WG_UNIFORM(x) (x)
#else
#define WG_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#endif
DEV GraphDims net_dims(const TopoView& tv, int g) {
    GraphDims d;
    const int n0 = tv.p[DRGNN_TI_NPTR][g], n1 = tv.p[DRGNN_TI_NPTR][g + 1];
    const int e0 = tv.p[DRGNN_TI_EPTR][g], e1 = tv.p[DRGNN_TI_EPTR][g + 1];
    const int c = tv.p[DRGNN_TI_NC0][g], m1 = tv.p[DRGNN_TI_NE1][g], c1 = tv.p[DRGNN_TI_NC1][g];
    d.n0 = WG_UNIFORM(n0);
    d.N = WG_UNIFORM(n1) - d.n0;
    d.e0 = WG_UNIFORM(e0);
    d.E = WG_UNIFORM(e1) - d.e0;
    d.C = WG_UNIFORM(c);
    d.E1 = WG_UNIFORM(m1);
    d.C1 = WG_UNIFORM(c1);
    d.rowbase = d.n0 + g;
    return d;
}

// strided [K,H] weight -> dense padded LDS rows  dst[k*ld + h]
DEV void stage_weight(float* dst, int ld, const float* src, long sk, long sh, int K, int H) {
    if (src == nullptr) return;
    FOR_TID(e, K * H) {
        const int k = e / H, h = e % H;
        dst[k * ld + h] = src[(long)k * sk + (long)h * sh];
    }
}
DEV void stage_i32(int* dst, const int32_t* src, int n) { FOR_TID(i, n) { dst[i] = src[i]; } }
DEV void stage_f32(float* dst, const float* src, int n) { FOR_TID(i, n) { dst[i] = src[i]; } }

// sizes for which the register-burst prologue applies (its per-lane element counts are fixed)
DEV bool net_burst_ok(const float* xg, int F, int N, int E, int C) {
    return ((((uintptr_t)xg) & 15) == 0) && (F % 4 == 0) && (F * DRGNN_H1 <= DRGNN_BCAP) && ((long)N * F <= 16L * DRGNN_BCAP) &&
           (N + 1 <= DRGNN_BCAP) && (E <= 2 * DRGNN_BCAP) && (C * DRGNN_H1 <= 4 * DRGNN_BCAP);
}

// weights + x tile
template <int KIND>
DEV void net_stage_common(const NetArgs& a, const GraphDims& d, int br, NetScratch& s) {
    const int F = a.net.n_feat;
    const drgnn_conv_params& c1 = a.net.conv1[br];
    const drgnn_conv_params& c2 = a.net.conv2[br];
    const float* xg = a.x + (long)d.n0 * F;
    FOR_TID(e, d.N * F) {
        const int i = e / F, f = e % F;
        s.xs[i * (F + 1) + f] = xg[e];
    }
    stage_weight(s.wn1, DRGNN_W1LD, c1.w_nbr, c1.nbr_sk, c1.nbr_sh, F, DRGNN_H1);
    stage_weight(s.wn2, DRGNN_W2LD, c2.w_nbr, c2.nbr_sk, c2.nbr_sh, DRGNN_H1, DRGNN_H2);
    if (KIND != DRGNN_GINET) {
        stage_weight(s.ws1, DRGNN_W1LD, c1.w_self, c1.self_sk, c1.self_sh, F, DRGNN_H1);
        stage_weight(s.ws2, DRGNN_W2LD, c2.w_self, c2.self_sk, c2.self_sh, DRGNN_H1, DRGNN_H2);
        stage_f32(s.b1, c1.bias, DRGNN_H1);
        stage_f32(s.b2, c2.bias, DRGNN_H2);
    }
}

// per-row coefficients of one level:  dv[i] (edge scale), sc[i] (self scale)
template <int KIND>
DEV void net_row_coefs(int n, const int* rp, const float* w, float* dv, float* sc) {
    if (KIND == DRGNN_GINET) return;
    FOR_TID(i, n) {
        const int lo = rp[i], hi = rp[i + 1];
        const int deg = hi - lo;
        if (KIND == DRGNN_SGAT) {
            float asum = 0.0f;
            for (int k = lo; k < hi; ++k) asum += w[k];
            const float inv = 1.0f / (float)(deg > 0 ? deg : 1);
            dv[i] = inv;
            sc[i] = asum * inv;
        } else {
            dv[i] = deg > 0 ? 1.0f / (float)deg : 0.0f;   // deg == 0 handled explicitly (NaN fwd, 0 bwd)
            sc[i] = 1.0f;
        }
    }
}

// z[i, :] = relu( sc[i]*u[i, H:2H] + sum_k coef_k * u[col[k], 0:H] + bias )   (H multiple of 4)
// A16: u / z rows are 16-byte aligned (LDS carve of the fused step) -> 128-bit LDS accesses
#ifdef DRGNN_EMU
#define NET_LD4(A16, p, v0, v1, v2, v3) do { v0 = (p)[0]; v1 = (p)[1]; v2 = (p)[2]; v3 = (p)[3]; } while (0)
#define NET_ST4(A16, p, v0, v1, v2, v3) do { (p)[0] = v0; (p)[1] = v1; (p)[2] = v2; (p)[3] = v3; } while (0)
#else
#define NET_LD4(A16, p, v0, v1, v2, v3)                                                        \
    do {                                                                                       \
        if (A16) { const drgnn_f4 t_ = *(const drgnn_f4*)(p); v0 = t_[0]; v1 = t_[1]; v2 = t_[2]; v3 = t_[3]; } \
        else { v0 = (p)[0]; v1 = (p)[1]; v2 = (p)[2]; v3 = (p)[3]; }                           \
    } while (0)
#define NET_ST4(A16, p, v0, v1, v2, v3)                                                        \
    do {                                                                                       \
        if (A16) { drgnn_f4 t_ = {v0, v1, v2, v3}; *(drgnn_f4*)(p) = t_; }                     \
        else { (p)[0] = v0; (p)[1] = v1; (p)[2] = v2; (p)[3] = v3; }                           \
    } while (0)
#endif
// LDU: row stride of u in floats (0: dense rows of HC = H or 2H)
// COEF: the per-row coefficients (net_row_coefs: dv = 1/deg, sc = mean edge weight | 1) are formed HERE from the row's
// own entry list (same summation order) and filed in dv / sc for the backward pass by the lane that owns the row's
// first channel group -- no coefficient phase, no barrier of its own (fused step kernel)
template <int KIND, int H, bool A16 = false, int LDU = 0, class IdxT = int, bool COEF = false>
DEV void net_aggregate(int n, const int* rp, const IdxT* col, const float* w, float* dv,
                       float* sc, const float* u, const float* bias, float* z) {
    constexpr int HC = LDU ? LDU : ((KIND == DRGNN_GINET) ? H : 2 * H);
    constexpr int G = H / 4;
    FOR_TID(item, n * G) {
        const int i = item / G, c = (item % G) * 4;
        const int lo = rp[i], hi = rp[i + 1];
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, asum = 0.f;
        // sGAT (a coefficient per entry anyway): entries in batches of four independent (index -> row) chains, the last
        // batch padded with repeats of the row's last entry under a zero coefficient (x + 0 * v = x: same sum, same order)
        // instead of the 1 - 3 serial round trips of a remainder loop: -0.25 us per step.  GINet / FoutNet sum plain rows
        // with packed adds; for them the padding's extra instructions cost more than the remainder loop (+0.25 us).
        if (KIND == DRGNN_SGAT)
        for (int k = lo; k < hi; k += 4) {
            int kk[4];
            float cf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                kk[j] = (k + j < hi) ? k + j : hi - 1;
                cf[j] = (k + j < hi) ? 1.0f : 0.0f;
            }
            if (KIND == DRGNN_SGAT) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { cf[j] *= w[kk[j]]; if (COEF) asum += cf[j]; }
            }
            const float* uj[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) uj[j] = u + col[kk[j]] * HC + c;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v0, v1, v2, v3;
                NET_LD4(A16, uj[j], v0, v1, v2, v3);
                a0 = fmaf(cf[j], v0, a0); a1 = fmaf(cf[j], v1, a1);
                a2 = fmaf(cf[j], v2, a2); a3 = fmaf(cf[j], v3, a3);
            }
        }
        else
#pragma unroll 4
        for (int k = lo; k < hi; ++k) {
            const float* uj = u + col[k] * HC + c;
            float cf = 1.0f, v0, v1, v2, v3;
            if (KIND == DRGNN_SGAT) { cf = w[k]; if (COEF) asum += cf; }
            NET_LD4(A16, uj, v0, v1, v2, v3);
            a0 = fmaf(cf, v0, a0); a1 = fmaf(cf, v1, a1);
            a2 = fmaf(cf, v2, a2); a3 = fmaf(cf, v3, a3);
        }
        if (KIND != DRGNN_GINET) {
            float d, s;
            if (COEF) {
                const int deg = hi - lo;
                if (KIND == DRGNN_SGAT) { d = 1.0f / (float)(deg > 0 ? deg : 1); s = asum * d; }
                else { d = deg > 0 ? 1.0f / (float)deg : 0.0f; s = 1.0f; }
                if (c == 0) { dv[i] = d; sc[i] = s; }
            } else {
                d = dv[i]; s = sc[i];
            }
            const float* us = u + i * HC + H + c;
            float s0, s1, s2, s3;
            NET_LD4(A16, us, s0, s1, s2, s3);
            a0 = fmaf(s, s0, a0 * d) + bias[c + 0];
            a1 = fmaf(s, s1, a1 * d) + bias[c + 1];
            a2 = fmaf(s, s2, a2 * d) + bias[c + 2];
            a3 = fmaf(s, s3, a3 * d) + bias[c + 3];
            if (KIND == DRGNN_FOUT && hi == lo) { a0 = a1 = a2 = a3 = DRGNN_NAN; }
        }
        float* zi = z + i * H + c;
        // relu that lets NaN through, like torch (max(x,0) would swallow it)
        a0 = (a0 < 0.f) ? 0.f : a0; a1 = (a1 < 0.f) ? 0.f : a1;
        a2 = (a2 < 0.f) ? 0.f : a2; a3 = (a3 < 0.f) ? 0.f : a3;
        NET_ST4(A16, zi, a0, a1, a2, a3);
    }
}

// cluster max with argmax (first maximum in ascending member order; NaN never wins;
// empty cluster -> 0).  arg = -1 where no gradient can flow (value <= 0 or empty).
// LDO: row stride of `out` in floats (0: dense rows of H)
template <int H, int LDO = 0, class ArgT = int32_t, int LDZ = 0>
DEV void net_cluster_max(int nc, const int* mp, const int* mem, const float* z, float* out,
                         float* g_out, ArgT* g_arg) {
    FOR_TID(item, nc * H) {
        const int r = item / H, c = item % H;
        float best = DRGNN_NEG_INF;
        int arg = -1;
        // batches of four independent (member -> value) chains, a short last batch repeating the last member (which cannot
        // win again under strict >): no serial remainder loop
        const int plo = mp[r], phi = mp[r + 1];
        for (int p = plo; p < phi; p += 4) {
            int mm[4];
            float vv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) mm[j] = mem[(p + j < phi) ? p + j : phi - 1];
#pragma unroll
            for (int j = 0; j < 4; ++j) vv[j] = z[mm[j] * (LDZ ? LDZ : H) + c];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (vv[j] > best) { best = vv[j]; arg = mm[j]; }
        }
        if (arg < 0) best = 0.0f;
        out[LDO ? ROW24(r, LDO) + c : item] = best;
        if (g_out) g_out[item] = best;
        g_arg[item] = (ArgT)((best > 0.0f) ? arg : -1);
    }
}

// ---------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------
template <int KIND>
DEV void net_forward_graph(const NetArgs& a, int g, int br, float* scratch, int capN, int capE,
                           int capC) {
    constexpr int HC1 = (KIND == DRGNN_GINET) ? DRGNN_H1 : 2 * DRGNN_H1;
    constexpr int HC2 = (KIND == DRGNN_GINET) ? DRGNN_H2 : 2 * DRGNN_H2;
    const TopoView& tv = a.tv;
    const GraphDims d = net_dims(tv, g);
    const int F = a.net.n_feat;
    NetScratch s = net_carve(scratch, KIND, F, capN, capE, capC, 0);
    const long nodeoff = (long)br * a.n_nodes + d.n0;

    if (a.step_inc != nullptr && g == 0 && br == 0) {
        FOR_TID(i, 1) { a.step_inc[0] = a.step_inc[0] + 1; }
    }
    // ---- one burst of independent loads: everything this graph needs -> LDS ------------
    PHASE_MARK();
    const drgnn_conv_params& c1 = a.net.conv1[br];
    const drgnn_conv_params& c2 = a.net.conv2[br];
    const float* xg = a.x + (long)d.n0 * F;
    const bool burst = net_burst_ok(xg, F, d.N, d.E, d.C);
    if (burst) {
        BurstX<4> bx;       burst_load_x(bx, xg, d.N, F);
        BurstW<1> bw1, bw2, bs1, bs2;
        burst_load_w(bw1, c1.w_nbr, c1.nbr_sk, c1.nbr_sh, F, DRGNN_H1);
        burst_load_w(bw2, c2.w_nbr, c2.nbr_sk, c2.nbr_sh, DRGNN_H1, DRGNN_H2);
        Burst<int, 1> brp0, bmp0, bmem0, brp1, bmp1, bmem1;
        Burst<int, 2> bix0, bix1;
        burst_load(brp0, tv.p[DRGNN_TI_ROWPTR0] + d.rowbase, d.N + 1);
        burst_load(bix0, tv.p[DRGNN_TI_COL0] + d.e0, d.E);
        burst_load(bmp0, tv.p[DRGNN_TI_MPTR0] + d.rowbase, d.C + 1);
        burst_load(bmem0, tv.p[DRGNN_TI_MEM0] + d.n0, d.N);
        burst_load(brp1, tv.p[DRGNN_TI_ROWPTR1] + d.rowbase, d.C + 1);
        burst_load(bix1, tv.p[DRGNN_TI_COL1] + d.e0, d.E1);
        burst_load(bmp1, tv.p[DRGNN_TI_MPTR1] + d.rowbase, d.C1 + 1);
        burst_load(bmem1, tv.p[DRGNN_TI_MEM1] + d.n0, d.C);
        Burst<float, 1> bb1, bb2;
        Burst<float, 2> bew0, bew1;
        if (KIND != DRGNN_GINET) {
            burst_load_w(bs1, c1.w_self, c1.self_sk, c1.self_sh, F, DRGNN_H1);
            burst_load_w(bs2, c2.w_self, c2.self_sk, c2.self_sh, DRGNN_H1, DRGNN_H2);
            burst_load(bb1, c1.bias, DRGNN_H1);
            burst_load(bb2, c2.bias, DRGNN_H2);
        }
        if (KIND == DRGNN_SGAT) {
            burst_load(bew0, (const float*)(tv.w0 + d.e0), d.E);
            burst_load(bew1, (const float*)(tv.w1 + d.e0), d.E1);
        }
        burst_store_x(bx, s.xs);
        burst_store_w(bw1, s.wn1, DRGNN_W1LD);
        burst_store_w(bw2, s.wn2, DRGNN_W2LD);
        burst_store(brp0, s.rp0); burst_store(bix0, s.ix0);
        burst_store(bmp0, s.mp0); burst_store(bmem0, s.mem0);
        burst_store(brp1, s.rp1); burst_store(bix1, s.ix1);
        burst_store(bmp1, s.mp1); burst_store(bmem1, s.mem1);
        if (KIND != DRGNN_GINET) {
            burst_store_w(bs1, s.ws1, DRGNN_W1LD);
            burst_store_w(bs2, s.ws2, DRGNN_W2LD);
            burst_store(bb1, s.b1); burst_store(bb2, s.b2);
        }
        if (KIND == DRGNN_SGAT) { burst_store(bew0, s.ew0); burst_store(bew1, s.ew1); }
    } else {
        net_stage_common<KIND>(a, d, br, s);
        stage_i32(s.rp0, tv.p[DRGNN_TI_ROWPTR0] + d.rowbase, d.N + 1);
        stage_i32(s.ix0, tv.p[DRGNN_TI_COL0] + d.e0, d.E);
        stage_i32(s.mp0, tv.p[DRGNN_TI_MPTR0] + d.rowbase, d.C + 1);
        stage_i32(s.mem0, tv.p[DRGNN_TI_MEM0] + d.n0, d.N);
        stage_i32(s.rp1, tv.p[DRGNN_TI_ROWPTR1] + d.rowbase, d.C + 1);
        stage_i32(s.ix1, tv.p[DRGNN_TI_COL1] + d.e0, d.E1);
        stage_i32(s.mp1, tv.p[DRGNN_TI_MPTR1] + d.rowbase, d.C1 + 1);
        stage_i32(s.mem1, tv.p[DRGNN_TI_MEM1] + d.n0, d.C);
        if (KIND == DRGNN_SGAT) {
            stage_f32(s.ew0, tv.w0 + d.e0, d.E);
            stage_f32(s.ew1, tv.w1 + d.e0, d.E1);
        }
    }
    BARRIER();

    // conv1 dense part:  U1 = X W1
    wg_gemm(d.N, DRGNN_H1, F, s.xs, F + 1, 1, s.wn1, DRGNN_W1LD, 1, s.u1, HC1, 1);
    if (KIND != DRGNN_GINET)
        wg_gemm(d.N, DRGNN_H1, F, s.xs, F + 1, 1, s.ws1, DRGNN_W1LD, 1, s.u1 + DRGNN_H1, HC1, 1);
    net_row_coefs<KIND>(d.N, s.rp0, s.ew0, s.dv0, s.sc0);
    net_row_coefs<KIND>(d.C, s.rp1, s.ew1, s.dv1, s.sc1);
    BARRIER();
    net_aggregate<KIND, DRGNN_H1>(d.N, s.rp0, s.ix0, s.ew0, s.dv0, s.sc0, s.u1, s.b1, s.z1);
    BARRIER();
    net_cluster_max<DRGNN_H1>(d.C, s.mp0, s.mem0, s.z1, s.xp, a.xp + nodeoff * DRGNN_H1,
                              a.arg0 + nodeoff * DRGNN_H1);
    BARRIER();
    // conv2 on the pooled graph
    wg_gemm(d.C, DRGNN_H2, DRGNN_H1, s.xp, DRGNN_H1, 1, s.wn2, DRGNN_W2LD, 1, s.u2, HC2, 1);
    if (KIND != DRGNN_GINET)
        wg_gemm(d.C, DRGNN_H2, DRGNN_H1, s.xp, DRGNN_H1, 1, s.ws2, DRGNN_W2LD, 1, s.u2 + DRGNN_H2, HC2, 1);
    BARRIER();
    net_aggregate<KIND, DRGNN_H2>(d.C, s.rp1, s.ix1, s.ew1, s.dv1, s.sc1, s.u2, s.b2, s.z2);
    BARRIER();
    net_cluster_max<DRGNN_H2>(d.C1, s.mp1, s.mem1, s.z2, s.p2, nullptr, a.arg1 + nodeoff * DRGNN_H2);
    BARRIER();
    // graph readout: mean over the depth-1 clusters (scatter_mean with count clamp)
    const int bad = tv.p[DRGNN_TI_ERR][0] | tv.p[DRGNN_TI_GSTAT][g] | tv.p[DRGNN_TI_GSTAT][a.n_graphs + g];
    const int width = DRGNN_H2 * a.net.n_branch;
    FOR_TID(c, DRGNN_H2) {
        float acc = 0.0f;
        for (int k = 0; k < d.C1; ++k) acc += s.p2[k * DRGNN_H2 + c];
        acc = acc / (float)(d.C1 > 0 ? d.C1 : 1);
        a.readout[(long)g * width + br * DRGNN_H2 + c] = bad ? DRGNN_NAN : acc;
    }
}

// ---------------------------------------------------------------------------------
// backward
// partial layout per workgroup (floats), K x H row-major blocks:
//   [dW1nbr F*16][dW1self F*16][db1 16][dW2nbr 16*32][dW2self 16*32][db2 32]
// ---------------------------------------------------------------------------------
HD int64_t net_partial_floats(int n_feat) {
    return 2LL * n_feat * DRGNN_H1 + DRGNN_H1 + 2LL * DRGNN_H1 * DRGNN_H2 + DRGNN_H2;
}

// dU[j, 0:H]   = sum over CSC entries t of column j : coef * dZ[row(t), :]
// dU[i, H:2H]  = sc[i] * dZ[i, :]
template <int KIND, int H, bool A16 = false, int LDU = 0, class IdxT = int>
DEV void net_aggregate_bwd(int n, const int* deg_rp, const int* cp, const IdxT* ridx, const IdxT* tslot,
                           const float* w, const float* dv, const float* sc, const float* dz,
                           float* du) {
    constexpr int HC = LDU ? LDU : ((KIND == DRGNN_GINET) ? H : 2 * H);
    constexpr int G = H / 4;
    FOR_TID(item, n * G) {
        const int j = item / G, c = (item % G) * 4;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        const int tlo = cp[j], thi = cp[j + 1];
        if (KIND == DRGNN_SGAT)
        for (int t = tlo; t < thi; t += 4) {      // batches of four independent chains, padded under a zero coefficient
            int ii[4];
            float cf[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int tt = (t + q < thi) ? t + q : thi - 1;
                ii[q] = ridx[tt];
                cf[q] = (t + q < thi) ? 1.0f : 0.0f;
                if (KIND == DRGNN_SGAT) cf[q] *= w[tslot[tt]] * dv[ii[q]];
                if (KIND == DRGNN_FOUT) cf[q] *= dv[ii[q]];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float* di = dz + ii[q] * H + c;
                float v0, v1, v2, v3;
                NET_LD4(A16, di, v0, v1, v2, v3);
                a0 = fmaf(cf[q], v0, a0); a1 = fmaf(cf[q], v1, a1);
                a2 = fmaf(cf[q], v2, a2); a3 = fmaf(cf[q], v3, a3);
            }
        }
        else
#pragma unroll 4
        for (int t = cp[j]; t < cp[j + 1]; ++t) {
            const int i = ridx[t];
            float cf = 1.0f;
            if (KIND == DRGNN_SGAT) cf = w[tslot[t]] * dv[i];
            if (KIND == DRGNN_FOUT) cf = dv[i];
            const float* di = dz + i * H + c;
            float v0, v1, v2, v3;
            NET_LD4(A16, di, v0, v1, v2, v3);
            a0 = fmaf(cf, v0, a0); a1 = fmaf(cf, v1, a1);
            a2 = fmaf(cf, v2, a2); a3 = fmaf(cf, v3, a3);
        }
        float* uj = du + j * HC + c;
        NET_ST4(A16, uj, a0, a1, a2, a3);
        if (KIND != DRGNN_GINET) {
            float s = sc[j];
            if (KIND == DRGNN_FOUT && deg_rp[j + 1] == deg_rp[j]) s = 0.0f;   // NaN row never wins a max
            const float* dj = dz + j * H + c;
            float d0, d1, d2, d3;
            NET_LD4(A16, dj, d0, d1, d2, d3);
            d0 *= s; d1 *= s; d2 *= s; d3 *= s;
            NET_ST4(A16, uj + H, d0, d1, d2, d3);
        }
    }
}

template <int KIND>
DEV void net_backward_graph(const NetArgs& a, int g, int br, float* scratch, int capN, int capE,
                            int capC) {
    constexpr int HC1 = (KIND == DRGNN_GINET) ? DRGNN_H1 : 2 * DRGNN_H1;
    constexpr int HC2 = (KIND == DRGNN_GINET) ? DRGNN_H2 : 2 * DRGNN_H2;
    const TopoView& tv = a.tv;
    const GraphDims d = net_dims(tv, g);
    const int F = a.net.n_feat;
    NetScratch s = net_carve(scratch, KIND, F, capN, capE, capC, 1);
    const long nodeoff = (long)br * a.n_nodes + d.n0;
    const int width = DRGNN_H2 * a.net.n_branch;
    const float* dr = a.hf.enabled ? nullptr : a.grad_readout + (long)g * width + br * DRGNN_H2;
    float* part = a.partials + ((long)g * a.net.n_branch + br) * a.n_partial;
    float* p_w1n = part;
    float* p_w1s = p_w1n + (long)F * DRGNN_H1;
    float* p_b1 = p_w1s + (long)F * DRGNN_H1;
    float* p_w2n = p_b1 + DRGNN_H1;
    float* p_w2s = p_w2n + DRGNN_H1 * DRGNN_H2;
    float* p_b2 = p_w2s + DRGNN_H1 * DRGNN_H2;

    if (a.step_inc != nullptr && g == 0 && br == 0) {
        FOR_TID(i, 1) { a.step_inc[0] = a.step_inc[0] + 1; }
    }
    // ---- stage: x tile, weights, transposed graphs, saved activations -------------------
    PHASE_MARK();
    const drgnn_conv_params& c1 = a.net.conv1[br];
    const drgnn_conv_params& c2 = a.net.conv2[br];
    const float* xg = a.x + (long)d.n0 * F;
    const bool burst = net_burst_ok(xg, F, d.N, d.E, d.C);
    const bool head_staged = burst && a.hf.enabled && a.hf.stage && a.hf.H * a.hf.R <= 8 * DRGNN_BCAP &&
                             a.hf.O * a.hf.H <= 2 * DRGNN_BCAP && a.hf.H <= DRGNN_BCAP;
    if (burst) {
        BurstX<4> bx;       burst_load_x(bx, xg, d.N, F);
        BurstW<1> bw1, bw2, bs1, bs2;
        burst_load_w(bw1, c1.w_nbr, c1.nbr_sk, c1.nbr_sh, F, DRGNN_H1);
        burst_load_w(bw2, c2.w_nbr, c2.nbr_sk, c2.nbr_sh, DRGNN_H1, DRGNN_H2);
        BurstW<8> bhw1;
        Burst<float, 1> bhb1, bhb2, bhx;
        Burst<float, 2> bhw2;
        if (head_staged) {
            burst_load_w(bhw1, a.hf.w1, a.hf.R, 1, a.hf.H, a.hf.R);
            burst_load(bhb1, a.hf.b1, a.hf.H);
            burst_load(bhw2, a.hf.w2, a.hf.O * a.hf.H);
            burst_load(bhb2, a.hf.b2, a.hf.O);
            burst_load(bhx, a.hf.readout + (long)g * a.hf.R, a.hf.R);
        }
        Burst<int, 1> bcp0, bcp1, bdg0, bdg1;
        Burst<int, 2> bix0, bix1, bts0, bts1;
        Burst<int, 4> ba0;
        Burst<int, 8> ba1;
        Burst<float, 4> bxp;
        Burst<float, 1> bdr, bb1, bb2;
        Burst<float, 2> bew0, bew1;
        burst_load(bcp0, tv.p[DRGNN_TI_COLPTR0] + d.rowbase, d.N + 1);
        burst_load(bix0, tv.p[DRGNN_TI_ROWIDX0] + d.e0, d.E);
        burst_load(bcp1, tv.p[DRGNN_TI_COLPTR1] + d.rowbase, d.C + 1);
        burst_load(bix1, tv.p[DRGNN_TI_ROWIDX1] + d.e0, d.E1);
        burst_load(ba0, (const int*)(a.arg0 + nodeoff * DRGNN_H1), d.C * DRGNN_H1);
        burst_load(ba1, (const int*)(a.arg1 + nodeoff * DRGNN_H2), d.C1 * DRGNN_H2);
        burst_load(bxp, (const float*)(a.xp + nodeoff * DRGNN_H1), d.C * DRGNN_H1);
        burst_load(bdr, dr, DRGNN_H2);
        if (KIND != DRGNN_GINET) {
            burst_load_w(bs1, c1.w_self, c1.self_sk, c1.self_sh, F, DRGNN_H1);
            burst_load_w(bs2, c2.w_self, c2.self_sk, c2.self_sh, DRGNN_H1, DRGNN_H2);
            burst_load(bdg0, tv.p[DRGNN_TI_ROWPTR0] + d.rowbase, d.N + 1);
            burst_load(bdg1, tv.p[DRGNN_TI_ROWPTR1] + d.rowbase, d.C + 1);
        }
        if (KIND == DRGNN_SGAT) {
            burst_load(bts0, tv.p[DRGNN_TI_TSLOT0] + d.e0, d.E);
            burst_load(bts1, tv.p[DRGNN_TI_TSLOT1] + d.e0, d.E1);
            burst_load(bew0, (const float*)(tv.w0 + d.e0), d.E);
            burst_load(bew1, (const float*)(tv.w1 + d.e0), d.E1);
        }
        burst_store_x(bx, s.xs);
        burst_store_w(bw1, s.wn1, DRGNN_W1LD);
        burst_store_w(bw2, s.wn2, DRGNN_W2LD);
        burst_store(bcp0, s.rp0); burst_store(bix0, s.ix0);
        burst_store(bcp1, s.rp1); burst_store(bix1, s.ix1);
        burst_store(ba0, s.a0); burst_store(ba1, s.a1);
        burst_store(bxp, s.xp); burst_store(bdr, s.misc);
        if (head_staged) {
            float* hw1 = s.end;
            float* hb1 = hw1 + (long)a.hf.H * (a.hf.R + 1);
            float* hw2 = hb1 + a.hf.H;
            float* hb2 = hw2 + (long)a.hf.O * a.hf.H;
            burst_store_w(bhw1, hw1, a.hf.R + 1);
            burst_store(bhb1, hb1); burst_store(bhw2, hw2); burst_store(bhb2, hb2);
            burst_store(bhx, hb2 + a.hf.O);
        }
        if (KIND != DRGNN_GINET) {
            burst_store_w(bs1, s.ws1, DRGNN_W1LD);
            burst_store_w(bs2, s.ws2, DRGNN_W2LD);
            burst_store(bdg0, s.dg0); burst_store(bdg1, s.dg1);
        }
        if (KIND == DRGNN_SGAT) {
            burst_store(bts0, s.ts0); burst_store(bts1, s.ts1);
            burst_store(bew0, s.ew0); burst_store(bew1, s.ew1);
        }
    } else {
        net_stage_common<KIND>(a, d, br, s);
        stage_i32(s.rp0, tv.p[DRGNN_TI_COLPTR0] + d.rowbase, d.N + 1);
        stage_i32(s.ix0, tv.p[DRGNN_TI_ROWIDX0] + d.e0, d.E);
        stage_i32(s.rp1, tv.p[DRGNN_TI_COLPTR1] + d.rowbase, d.C + 1);
        stage_i32(s.ix1, tv.p[DRGNN_TI_ROWIDX1] + d.e0, d.E1);
        if (KIND != DRGNN_GINET) {
            stage_i32(s.dg0, tv.p[DRGNN_TI_ROWPTR0] + d.rowbase, d.N + 1);
            stage_i32(s.dg1, tv.p[DRGNN_TI_ROWPTR1] + d.rowbase, d.C + 1);
        }
        if (KIND == DRGNN_SGAT) {
            stage_i32(s.ts0, tv.p[DRGNN_TI_TSLOT0] + d.e0, d.E);
            stage_i32(s.ts1, tv.p[DRGNN_TI_TSLOT1] + d.e0, d.E1);
            stage_f32(s.ew0, tv.w0 + d.e0, d.E);
            stage_f32(s.ew1, tv.w1 + d.e0, d.E1);
        }
        stage_i32(s.a0, a.arg0 + nodeoff * DRGNN_H1, d.C * DRGNN_H1);
        stage_i32(s.a1, a.arg1 + nodeoff * DRGNN_H2, d.C1 * DRGNN_H2);
        stage_f32(s.xp, a.xp + nodeoff * DRGNN_H1, d.C * DRGNN_H1);
        if (dr != nullptr) stage_f32(s.misc, dr, DRGNN_H2);
    }
    FOR_TID(item, d.C * DRGNN_H2) { s.z2[item] = 0.0f; }
    FOR_TID(item, d.N * DRGNN_H1) { s.z1[item] = 0.0f; }
    BARRIER();
    if (a.hf.enabled) {      // FC head + loss + their backward for this graph: d loss / d readout -> s.misc
        const HeadFused& hf = a.hf;
        if (head_staged) {
            float* hw1 = s.end;
            float* hb1 = hw1 + (long)hf.H * (hf.R + 1);
            float* hw2 = hb1 + hf.H;
            float* hb2 = hw2 + (long)hf.O * hf.H;
            float* hx = hb2 + hf.O;
            head_graph(hf, g, br, s.gp, s.misc, hw1, hf.R + 1, hb1, hw2, hb2, hx);
        } else {
            head_graph(hf, g, br, s.gp, s.misc, hf.w1, hf.R, hf.b1, hf.w2, hf.b2, hf.readout + (long)g * hf.R);
        }
    }

    // ---- depth-1 max + mean backward: dZ2 (relu mask folded into arg1 = -1) ------------
    net_row_coefs<KIND>(d.C, s.dg1, s.ew1, s.dv1, s.sc1);
    net_row_coefs<KIND>(d.N, s.dg0, s.ew0, s.dv0, s.sc0);
    {
        const float inv = 1.0f / (float)(d.C1 > 0 ? d.C1 : 1);
        FOR_TID(item, d.C1 * DRGNN_H2) {
            const int r = s.a1[item];
            const int c = item % DRGNN_H2;
            if (r >= 0) s.z2[(long)r * DRGNN_H2 + c] = s.misc[c] * inv;
        }
    }
    BARRIER();
    // ---- conv2 backward ------------------------------------------------------------
    net_aggregate_bwd<KIND, DRGNN_H2>(d.C, s.dg1, s.rp1, s.ix1, s.ts1, s.ew1, s.dv1, s.sc1, s.z2, s.u2);
    if (KIND != DRGNN_GINET) {
        FOR_TID(c, DRGNN_H2) {
            float acc = 0.0f;
            for (int r = 0; r < d.C; ++r) acc += s.z2[r * DRGNN_H2 + c];
            p_b2[c] = acc;
        }
    }
    BARRIER();
    // dW2 = XP^T dU2      (A(i=k16, k=r) = xp[r*16 + i])
    wg_gemm(DRGNN_H1, DRGNN_H2, d.C, s.xp, 1, DRGNN_H1, s.u2, HC2, 1, p_w2n, DRGNN_H2, 1);
    if (KIND != DRGNN_GINET)
        wg_gemm(DRGNN_H1, DRGNN_H2, d.C, s.xp, 1, DRGNN_H1, s.u2 + DRGNN_H2, HC2, 1, p_w2s, DRGNN_H2, 1);
    // dXP = dU2n W2n^T (+ dU2s W2s^T):  B(k=h, j=i16) = W2(i16, h) = wn2[i*33 + h]
    wg_gemm(d.C, DRGNN_H1, DRGNN_H2, s.u2, HC2, 1, s.wn2, 1, DRGNN_W2LD, s.dxp, DRGNN_H1, 1);
    if (KIND != DRGNN_GINET)
        wg_gemm(d.C, DRGNN_H1, DRGNN_H2, s.u2 + DRGNN_H2, HC2, 1, s.ws2, 1, DRGNN_W2LD, s.p2, DRGNN_H1, 1);
    BARRIER();
    // ---- depth-0 max backward: dZ1 ----------------------------------------------------
    FOR_TID(item, d.C * DRGNN_H1) {
        const int m = s.a0[item];
        const int c = item % DRGNN_H1;
        if (m >= 0) {
            float v = s.dxp[item];
            if (KIND != DRGNN_GINET) v += s.p2[item];
            s.z1[(long)m * DRGNN_H1 + c] = v;
        }
    }
    BARRIER();
    // ---- conv1 backward ------------------------------------------------------------
    net_aggregate_bwd<KIND, DRGNN_H1>(d.N, s.dg0, s.rp0, s.ix0, s.ts0, s.ew0, s.dv0, s.sc0, s.z1, s.u1);
    if (KIND != DRGNN_GINET) {
        FOR_TID(c, DRGNN_H1) {
            float acc = 0.0f;
            for (int i = 0; i < d.N; ++i) acc += s.z1[i * DRGNN_H1 + c];
            p_b1[c] = acc;
        }
    }
    BARRIER();
    // dW1 = X^T dU1      (A(i=f, k=node) = xs[node*(F+1) + f]); K = N_g is long: split it
    {
        const int mtiles = (F + 15) >> 4;
        int KS = imin(DRGNN_NWAVES / mtiles, 2048 / (F * DRGNN_H1));
        if (KS < 1) KS = 1;
        wg_gemm(F, DRGNN_H1, d.N, s.xs, 1, F + 1, s.u1, HC1, 1, p_w1n, DRGNN_H1, 1, KS, s.gp);
        if (KIND != DRGNN_GINET) {
            BARRIER();
            wg_gemm(F, DRGNN_H1, d.N, s.xs, 1, F + 1, s.u1 + DRGNN_H1, HC1, 1, p_w1s, DRGNN_H1, 1, KS, s.gp);
        }
    }
    if (a.grad_x != nullptr) {
        // dX = dU1n W1n^T (+ dU1s W1s^T), written per branch; branches are summed by the reducer
        float* gx = a.grad_x + nodeoff * F;
        BARRIER();
        wg_gemm(d.N, F, DRGNN_H1, s.u1, HC1, 1, s.wn1, 1, DRGNN_W1LD, s.xs, F + 1, 1);
        BARRIER();
        if (KIND != DRGNN_GINET) {
            FOR_TID(item, d.N * F) {
                const int i = item / F, f = item % F;
                const float* us = s.u1 + (long)i * HC1 + DRGNN_H1;
                float acc = s.xs[i * (F + 1) + f];
                for (int h = 0; h < DRGNN_H1; ++h) acc = fmaf(us[h], s.ws1[f * DRGNN_W1LD + h], acc);
                gx[item] = acc;
            }
        } else {
            FOR_TID(item, d.N * F) { gx[item] = s.xs[(item / F) * (F + 1) + item % F]; }
        }
    }
}
