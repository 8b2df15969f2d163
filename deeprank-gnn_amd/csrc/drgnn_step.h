// Fused training step of one (graph, branch) workgroup: body forward, FC head + loss, body
// backward in ONE launch, every intermediate kept in LDS.
//
// Same math as net_forward_graph + head + net_backward_graph (drgnn_net.h; reference
// ginet.py:103-139, sGAT.py:119-137, foutnet.py:108-124 and their autograd), but
//   * the graph is staged once (CSR and CSC of both levels, member lists, x tile, weights),
//   * pooled features / argmax indices never leave LDS (no xp / arg0 / arg1 round trip),
//   * the per-graph head needs the readout of BOTH branches of a GINet.  fc1 is linear, so each
//     branch workgroup multiplies ITS 32 readout columns with ITS column block of fc1.weight (the
//     only part of that matrix it ever touches: 16 KB in LDS, also used by the head's backward) and
//     the two workgroups of a graph exchange the H half-products through tagged 64-bit words in
//     global memory (one relaxed agent-scope atomic per value; tag = index of this step, so a
//     word is valid exactly when its tag matches -- no fence, no flag; inference launches, whose tag does
//     not change, have the reader clear the word it consumed),
//   * dW_fc1 = dhid^T readout is left to the update kernel (it only needs dhid [B,H] and the
//     readout [B,R]), so the head writes a compact slab  [dhid H][dW2 O*H][db2 O][loss][weight].
// Phases are separated by BARRIER(); the host emulation (tests) runs the two halves of the
// kernel as two passes over all workgroups (`part` 1 then 2) because it executes workgroups one
// after the other and cannot wait for a partner.
#ifndef DRGNN_STEP_H
#define DRGNN_STEP_H

#include "drgnn_net.h"

// ablation profiling (tools/ablate_step.sh): -DDRGNN_SKIP=k compiles the step kernel WITHOUT the work
// of phase k (barriers stay); the drop in kernel time is what that phase costs.  Never set in the product.
#ifndef DRGNN_SKIP
#define DRGNN_SKIP (-1)
#endif
#define PH(k) if (DRGNN_SKIP != (k))
// -DDRGNN_EXIT_AFTER=k: every workgroup returns after barrier k (cumulative timeline of the phases)
#ifndef DRGNN_EXIT_AFTER
#define DRGNN_EXIT_AFTER (-1)
#endif
// (the checksum over the whole scratch keeps every earlier LDS store alive in the truncated kernel)
#define EXIT_AFTER(k)                                                                          \
    do {                                                                                       \
        if (DRGNN_EXIT_AFTER == (k)) {                                                         \
            float acc_ = 0.0f;                                                                 \
            if ((k) > 0) { FOR_TID(i_, (int)(s.end - scratch)) { acc_ += scratch[i_]; } }      \
            if (acc_ == 12345.678f) a.hf.pred[0] = acc_;                                       \
            return;                                                                            \
        }                                                                                      \
    } while (0)

struct StepArgs {
    drgnn_net_desc net;
    const float* x;              // [Ntot, F]
    TopoView tv;
    int64_t n_nodes;
    int n_graphs;
    float* partials;             // [B*n_branch][P] conv weight-gradient slabs (net_partial_floats)
    int n_partial;
    HeadFused hf;                // hf.readout: [B][R] OUTPUT; hf.partials: [B][head_compact_floats]
    unsigned long long* xchg;    // [B][n_branch][H] tagged fc1 half-products (zero-initialised once by the owner)
    int xchg_stride;             // node-split layout (drgnn_step2.h): exchange words per graph (step2_xchg_words)
    int32_t* step2;              // [0] steps completed so far (read)   [1] index of this step (written)   [2] sticky fault bits
    // cached-topology mode: slot g of the launch is graph gather_ids[g] of the workspace `tv` describes (a whole
    // resident set, ws_graphs graphs); null: slot g = graph g of a per-mini-batch workspace
    const int32_t* gather_ids;
    int ws_graphs;
    // aggregation tiles of the workspace (DRGNN_TOPO_TILES): S [tile_nodes][F] | D [tile_nodes] | C [tile_nodes], node order
    const float* tiles;
    int64_t tile_nodes;
};

HD int64_t head_compact_floats(int R, int H, int O) { (void)R; return (int64_t)H + (int64_t)O * H + O + 2; }

#define STEP_XPLD (DRGNN_H1 + 4)     // pooled features: 20-float rows (conflict-free 128-bit row reads)
// the fc1 column block [H][STEP_WBLD] and, after the head, the partial tiles of step_gemm_tn (256 floats per
// (tile, K slice) unit) share one area: at least 8 units, all 16 when H makes it that large anyway
#define STEP_WBLD (DRGNN_H2 + 4)     // row stride of the fc1.weight column block in LDS (16-byte aligned rows)
HD int step_gp_words(int H) { return H * STEP_WBLD > 2048 ? H * STEP_WBLD : 2048; }
HD int step_pad4(int n) { return (n + 3) & ~3; }
HD int step_pad16(int n) { return (n + 15) & ~15; }

template <bool NARROW> struct StepIdx { typedef int type; };
template <> struct StepIdx<true> { typedef unsigned short type; };

// ---- scratch ---------------------------------------------------------------------------------
struct StepScratch {
    float* xs; float* w1t; float* ws1t; float* b1; float* w2t; float* w2n; float* b2;
    // sGAT / FoutNet conv2 as ONE product over the concatenated operand [S | T] (S = aggregated neighbours, T = scaled
    // self rows): wc2t[n][k] = [Wnbr ; Wself][k][n] (forward), wc2n[k][n] = the same matrix row-major (backward)
    float* wc2t; float* wc2n;
    float* ct0;                  // [capE] coefficient of every CSC0 entry: c_e d_row(e) (sGAT / FoutNet backward of conv1)
    // edge-indexed arrays (cx*, rx*, ts*): node ids / slot numbers of ONE graph.  32 bits each for GINet and
    // FoutNet; 16 bits for sGAT, whose per-edge weights and slot maps would not fit LDS otherwise (the narrow
    // loads cost GINet 1.6 us of 17.7, so it keeps the wide ones)
    int* rp0; int* cx0; float* ew0; int* cp0; int* rx0; int* ts0; int* mp0; int* mem0;
    int* rp1; int* cx1; float* ew1; int* cp1; int* rx1; int* ts1; int* mp1; int* mem1;
    short* a0; short* a1;        // argmax node ids as 16-bit (a graph in LDS has < 32768 nodes)
    float* u1; float* z1; float* dv0; float* sc0;
    float* xp; float* u2; float* z2; float* p2; float* dv1; float* sc1;
    float* gp; float* misc;
    float* xr; float* hid; float* dhid;
    float* wb; float* hb1; float* hw2; float* hb2;
    float* bsum;                 // [16 waves][32] wave partials of the bias-gradient column sums (sGAT / FoutNet)
    float* end;
};

// The head's small arrays and the weights come FIRST: their sizes depend on the head's widths and on the padded
// feature width only, so in the width-specialised kernels (both constants) their offsets fold into immediates instead
// of living in pinned VGPRs.
#define STEP_CARVE_LIST(X)                                                                     \
    X(misc, 128, 1)                                                                            \
    X(xr, R, 1)                                                                                \
    X(hid, H, 1)                                                                               \
    X(dhid, H, 1)                                                                              \
    X(hb1, H, 1)                                                                               \
    X(bsum, DRGNN_NWAVES * DRGNN_H2, !gin)                                                     \
    X(wb, step_gp_words((int)H), 1)                                                            \
    X(w1t, DRGNN_H1 * xld, 1)                                                                  \
    X(ws1t, DRGNN_H1 * xld, !gin)                                                              \
    X(b1, DRGNN_H1, !gin)                                                                      \
    X(w2t, DRGNN_H2 * STEP_XPLD, gin)                                                          \
    X(w2n, DRGNN_H1 * (DRGNN_H2 + 4), gin)                                                     \
    X(wc2t, DRGNN_H2 * (DRGNN_H2 + 4), !gin)                                                   \
    X(wc2n, DRGNN_H2 * (DRGNN_H2 + 4), !gin)                                                   \
    X(b2, DRGNN_H2, !gin)                                                                      \
    X(xs, (long)(capN + 4) * xld, 1)                                                           \
    X(rp0, capN + 1, 1)                                                                        \
    X(cx0, (sg ? (capE + 1) / 2 : capE), 1)                                                    \
    X(ew0, capE, sg)                                                                           \
    X(cp0, capN + 1, 1)                                                                        \
    X(rx0, (sg ? (capE + 1) / 2 : capE), 1)                                                    \
    X(ts0, (sg ? (capE + 1) / 2 : capE), sg)                                                   \
    X(ct0, capE, !gin)                                                                         \
    X(mp0, capC + 1, 1)                                                                        \
    X(mem0, capN, 1)                                                                           \
    X(rp1, capC + 1, 1)                                                                        \
    X(cx1, (sg ? (capE + 1) / 2 : capE), 1)                                                    \
    X(ew1, capE, sg)                                                                           \
    X(cp1, capC + 1, 1)                                                                        \
    X(rx1, (sg ? (capE + 1) / 2 : capE), 1)                                                    \
    X(ts1, (sg ? (capE + 1) / 2 : capE), sg)                                                   \
    X(mp1, capC + 1, 1)                                                                        \
    X(mem1, capC, 1)                                                                           \
    X(a0, ((long)capC * DRGNN_H1 + 1) / 2, 1)                                                    \
    X(a1, ((long)capC * DRGNN_H2 + 1) / 2, 1)                                                    \
    X(u1, (long)(capN + 4) * hc1, 1)                                                           \
    X(z1, (long)capN * DRGNN_H1, 1)                                                            \
    X(dv0, capN, !gin)                                                                         \
    X(sc0, capN, !gin)                                                                         \
    X(xp, (long)(capC + 4) * STEP_XPLD, 1)                                                     \
    X(u2, (long)(capC + 4) * (DRGNN_H2 + 4), 1)                                                \
    X(z2, (long)(capC + 4) * (DRGNN_H2 + 4), 1)                                                \
    X(p2, (gin ? ((long)capC * DRGNN_H2 > (long)(capC + 4) * STEP_XPLD ? (long)capC * DRGNN_H2 : (long)(capC + 4) * STEP_XPLD) \
               : (long)(capC + 4) * (DRGNN_H2 + 4)), 1)                                        \
    X(dv1, capC, !gin)                                                                         \
    X(sc1, capC, !gin)                                                                         \
    X(hw2, (long)O * H, 1)                                                                     \
    X(hb2, O, 1)

HD int64_t step_scratch_words(int kind, int64_t F, int64_t capN, int64_t capE, int64_t capC, int64_t R,
                              int64_t H, int64_t O) {
    const int64_t hc1 = (kind == DRGNN_GINET) ? DRGNN_H1 : 2 * DRGNN_H1;
    const int64_t hc2 = (kind == DRGNN_GINET) ? DRGNN_H2 : 2 * DRGNN_H2;
    const int sg = (kind == DRGNN_SGAT) ? 1 : 0;
    const int gin = (kind == DRGNN_GINET) ? 1 : 0;
    const int64_t xld = step_pad16((int)F) + 4;
    int64_t w = 0;
#define X(name, words, cond) w += (cond) ? (((int64_t)(words) + 3) & ~(int64_t)3) : 0;   /* 16-byte aligned arrays */
    STEP_CARVE_LIST(X)
#undef X
    return w + 16;
}

#ifndef DRGNN_EMU
typedef unsigned int drgnn_u2 __attribute__((ext_vector_type(2)));
#endif
#ifdef DRGNN_EMU
#define STEP_PIN(x) ((void)0)
#else
#define STEP_PIN(x) asm volatile("" : "+v"(x))
#endif
constexpr bool step_streq(const char* a, const char* b) { return *a == *b && (*a == 0 || step_streq(a + 1, b + 1)); }
constexpr bool step_cold_array(const char* n) {
    return step_streq(n, "ew0") || step_streq(n, "ts0") || step_streq(n, "ew1") || step_streq(n, "ts1") ||
           step_streq(n, "dv0") || step_streq(n, "sc0") || step_streq(n, "dv1") || step_streq(n, "sc1") ||
           step_streq(n, "b1") || step_streq(n, "b2") || step_streq(n, "bsum");
}
DEV StepScratch step_carve(float* base, int kind, int F, int capN, int capE, int capC, int R, int H, int O) {
    const int hc1 = (kind == DRGNN_GINET) ? DRGNN_H1 : 2 * DRGNN_H1;
    const int hc2 = (kind == DRGNN_GINET) ? DRGNN_H2 : 2 * DRGNN_H2;
    const int sg = (kind == DRGNN_SGAT) ? 1 : 0;
    const int gin = (kind == DRGNN_GINET) ? 1 : 0;
    const int xld = step_pad16(F) + 4;
    StepScratch s;
    int o = 0;
    // Every array offset is pinned in a vector register once: there are too many of them for the
    // scalar file, and otherwise each phase of each wave recomputes its operands' offsets from the
    // capacities (measured: ~15% of the kernel).
    // Arrays only the sGAT / FoutNet variants use in one or two phases (edge weights, transposed slots, degree and
    // scale vectors, biases) are NOT pinned: their offset is one add away from the pinned offset in front of them (the
    // running offset restarts from every pinned value), and those kernels sit at the VGPR limit.
#define X(name, words, cond)                                                          \
    { int off = o; if (!step_cold_array(#name)) { STEP_PIN(off); } s.name = (decltype(s.name))(base + off);          \
      o = off + ((cond) ? (int)(((long)(words) + 3) & ~3L) : 0); }
    STEP_CARVE_LIST(X)
#undef X
    s.end = base + o;
    // fc1's column block is dead after the head's backward; the K-split products that follow (dW2, dW1)
    // keep their partial tiles there
    s.gp = s.wb;
    return s;
}

// ---- dense products of the step kernel ----------------------------------------------------------
// Layout rules that let the MFMA loops run without lane predicates:
//   * row-major operands have 16-byte aligned rows (stride % 4 == 0) and their K extent is zero padded
//     to a multiple of 16 (step_gemm_nn) -- a lane fetches 4 consecutive k with ONE 128-bit LDS read
//     and feeds them to 4 MFMA steps (the k order inside a 16-chunk is permuted the same way for A
//     and B, which does not change the sum's terms);
//   * node-major operands (K = node index) keep rows [K, pad4(K)) readable and ZERO (step_gemm_tn).
// Rows past M of a last tile are computed from whatever LDS holds and their stores discarded.

// C[M x 16*NT] (row stride ldc) = A[M x K] * Bt^T,  A rows of stride lda, Bt[n][k] rows of stride ldbt
// RELU: C = relu(...) with NaN passing through, like torch
#ifdef DRGNN_EMU
template <bool RELU = false>
DEV void step_gemm_nn(int M, int NT, int K, const float* A, int lda, const float* Bt, int ldbt, float* C, int ldc,
                      int* dummy, const float* bias = nullptr, const float* nan_rows = nullptr, int wave_shift = 0) {
    (void)dummy; (void)wave_shift;
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < 16 * NT; ++j) {
            float acc = 0.0f;
            for (int k = 0; k < K; ++k) acc = fmaf(A[i * lda + k], Bt[j * ldbt + k], acc);
            if (bias) acc += bias[j];
            if (nan_rows && nan_rows[i] == 0.0f) acc = DRGNN_NAN;
            if (RELU) acc = (acc < 0.0f) ? 0.0f : acc;
            C[i * ldc + j] = acc;
        }
}
#else
// wave_shift: tile unit u goes to wave (u + wave_shift) mod 16 -- callers that issue two products in one phase start the
// second one where the first one's units end, so that all 16 waves get tiles
template <bool RELU = false>
DEV void step_gemm_nn(int M, int NT, int K, const float* A, int lda, const float* Bt, int ldbt, float* C, int ldc,
                      int* dummy, const float* bias = nullptr, const float* nan_rows = nullptr, int wave_shift = 0) {
    // nan_rows: rows i with nan_rows[i] == 0 are written as NaN (FoutLayer's mean over an empty neighbourhood)
    const int wave = (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) - wave_shift) & (DRGNN_NWAVES - 1);
    const int lane = threadIdx.x & 63;
    const int lr = lane & 15, lq = lane >> 4;
    const int units = ((M + 15) >> 4) * NT;
    for (int u = wave; u < units; u += DRGNN_NWAVES) {
        const int ti = (NT == 1) ? u : (u >> 1), tj = (NT == 1) ? 0 : (u & 1);      // NT is 1 or 2
        const float* ap = A + (ti * 16 + lr) * lda + 4 * lq;
        const float* bp = Bt + (tj * 16 + lr) * ldbt + 4 * lq;
        drgnn_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < K; k0 += 32) {
            const drgnn_f4 a0 = *(const drgnn_f4*)(ap + k0), b0 = *(const drgnn_f4*)(bp + k0);
            const bool two = k0 + 16 < K;
            drgnn_f4 a1 = a0, b1 = b0;
            if (two) { a1 = *(const drgnn_f4*)(ap + k0 + 16); b1 = *(const drgnn_f4*)(bp + k0 + 16); }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], b0[j], acc, 0, 0, 0);
            if (two) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], b1[j], acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ci = ti * 16 + lq * 4 + r;
            float* p = (ci < M) ? C + ci * ldc + tj * 16 + lr : (float*)dummy + lane;
            float v = acc[r];
            if (bias) v += bias[tj * 16 + lr];
            if (nan_rows && ci < M && nan_rows[ci] == 0.0f) v = DRGNN_NAN;
            if (RELU) v = (v < 0.0f) ? 0.0f : v;
            *p = v;
        }
    }
}
#endif

// Two products sharing the A operand: C[:, 0:16] = A Bt0^T, C[:, 16:32] = A Bt1^T (sGAT / FoutNet conv1: neighbour and self
// weights).  One unit = one 16-row tile with BOTH column tiles: the A fragments are fetched once.
#ifdef DRGNN_EMU
DEV void step_gemm_nn_dual(int M, int K, const float* A, int lda, const float* Bt0, const float* Bt1, int ldbt, float* C,
                           int ldc, int* dummy) {
    step_gemm_nn(M, 1, K, A, lda, Bt0, ldbt, C, ldc, dummy);
    step_gemm_nn(M, 1, K, A, lda, Bt1, ldbt, C + 16, ldc, dummy);
}
#else
DEV void step_gemm_nn_dual(int M, int K, const float* A, int lda, const float* Bt0, const float* Bt1, int ldbt, float* C,
                           int ldc, int* dummy) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int lr = lane & 15, lq = lane >> 4;
    const int units = (M + 15) >> 4;
    for (int ti = wave; ti < units; ti += DRGNN_NWAVES) {
        const float* ap = A + (ti * 16 + lr) * lda + 4 * lq;
        const float* bp0 = Bt0 + lr * ldbt + 4 * lq;
        const float* bp1 = Bt1 + lr * ldbt + 4 * lq;
        drgnn_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < K; k0 += 16) {
            const drgnn_f4 a = *(const drgnn_f4*)(ap + k0);
            const drgnn_f4 b0 = *(const drgnn_f4*)(bp0 + k0), b1 = *(const drgnn_f4*)(bp1 + k0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b0[j], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b1[j], acc1, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ci = ti * 16 + lq * 4 + r;
            float* p = (ci < M) ? C + ci * ldc + lr : (float*)dummy + lane;
            float* q = (ci < M) ? C + ci * ldc + 16 + lr : (float*)dummy + lane;
            *p = acc0[r];
            *q = acc1[r];
        }
    }
}
#endif

// C[Mrows <= 16*MT x 16*NT] (global, row stride ldc) = sum_k A[k][i] * B[k][j],  A rows of stride lda (K of them),
// B rows of stride ldb.  K is cut in KS slices (one (tile, slice) unit per wave); the partial tiles go to
// `part` ([KS * MT * NT][64 lanes][4]) and are summed in slice order.  Contains one workgroup barrier;
// callers put another one before reusing `part`.
#ifdef DRGNN_EMU
DEV void step_gemm_tn(int MT, int NT, int K, const float* A, int lda, const float* B, int ldb, int KS, float* part,
                      float* C, int ldc, int Mrows, int stage = 0, int wave_shift = 0) {
    (void)KS; (void)part; (void)MT; (void)wave_shift;
    if (stage == 2) return;      // (emulation: stage 1 forms the whole product)
    for (int i = 0; i < Mrows; ++i)
        for (int j = 0; j < 16 * NT; ++j) {
            float acc = 0.0f;
            for (int k = 0; k < K; ++k) acc = fmaf(A[k * lda + i], B[k * ldb + j], acc);
            C[i * ldc + j] = acc;
        }
}
DEV void step_gemm_tn_bufa(int MT, int K, const float* A, int lda, int a_bytes, const float* B, int ldb, int KS, float* part,
                           float* C, int ldc, int Mrows, int stage = 0, int wave_shift = 0) {
    (void)a_bytes; (void)wave_shift;
    step_gemm_tn(MT, 1, K, A, lda, B, ldb, KS, part, C, ldc, Mrows, stage);
}
// two products sharing the A operand in one pass: B holds [B0 | B1] side by side (16*NTH columns each), the results
// go to C and C + chalf (sGAT / FoutNet: neighbour and self weight gradients)
DEV void step_gemm_tn_pair(int MT, int NTH, int K, const float* A, int lda, const float* B, int ldb, int KS, float* part,
                           float* C, int chalf, int ldc, int Mrows) {
    step_gemm_tn(MT, NTH, K, A, lda, B, ldb, KS, part, C, ldc, Mrows);
    step_gemm_tn(MT, NTH, K, A, lda, B + 16 * NTH, ldb, KS, part, C + chalf, ldc, Mrows);
}
#else
// MT/NT compile-time (0: run-time value in mt_rt / nt_rt), KS a power of two: no integer division left
// NTH: column tiles per output block (NT = NTH: one block at C; NT = 2 * NTH: second block at C + chalf)
// stage: 0 = the whole product (contains a workgroup barrier); 1 = only the partial tiles, 2 = only their sum and the
// stores -- the caller's own phase barrier lies between the two, so the product adds no barrier to the chain
// BUFA: the A operand is read straight from GLOBAL memory through a buffer descriptor over its a_bytes (rows past the end
// read as zero, like the zero rows an LDS operand keeps): for callers whose LDS copy of A is gone by then
template <int MTC, int NTC, int NTH = 0, bool BUFA = false>
DEV void step_gemm_tn_t(int mt_rt, int nt_rt, int K, const float* A, int lda, const float* B, int ldb, int KS,
                        float* part, float* C, int ldc, int Mrows, int chalf = 0, int stage = 0, int a_bytes = 0,
                        int wave_shift = 0) {
    const int MT = MTC ? MTC : mt_rt, NT = NTC ? NTC : nt_rt;
    const int wave = (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) - wave_shift) & (DRGNN_NWAVES - 1);
    const int lane = threadIdx.x & 63;
    const int lr = lane & 15, lq = lane >> 4;
    const int K4 = step_pad4(K);
    const int ks_log = 31 - __builtin_clz((unsigned)KS);
    const int kslice = (((K4 >> 2) + KS - 1) >> ks_log) << 2;
    const int tiles = MT * NT, units = tiles * KS;
    if (stage != 2)
    for (int u = wave; u < units; u += DRGNN_NWAVES) {
        const int ks = u / tiles, t = u - ks * tiles;
        const int ti = t / NT, tj = t - ti * NT;
        const int kbeg = ks * kslice, kend = imin(K4, kbeg + kslice);
        drgnn_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* ap = A + (kbeg + lq) * lda + ti * 16 + lr;
        const float* bp = B + (kbeg + lq) * ldb + tj * 16 + lr;
        const __amdgpu_buffer_rsrc_t arsrc = buf_rsrc(A, BUFA ? a_bytes : 0);
        int aoff = ((kbeg + lq) * lda + ti * 16 + lr) * 4;
        for (int k0 = kbeg; k0 < kend; k0 += 32) {
            float a[8], b[8];
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) {      // unconditional: rows past kend exist in LDS and are not used
                if (BUFA) a[s2] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(arsrc, aoff + 16 * s2 * lda, 0, 0));
                else a[s2] = ap[4 * s2 * lda];
                b[s2] = bp[4 * s2 * ldb];
            }
            aoff += 128 * lda;
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2)
                if (k0 + 4 * s2 < kend) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s2], b[s2], acc, 0, 0, 0);
            ap += 32 * lda;
            bp += 32 * ldb;
        }
        *(drgnn_f4*)(part + (u * 64 + lane) * 4) = drgnn_f4{acc[0], acc[1], acc[2], acc[3]};
    }
    if (stage == 1) return;
    if (stage == 0) __syncthreads();
    for (int e = threadIdx.x; e < tiles * 64; e += DRGNN_NTHREADS) {
        const int t = e >> 6, l = e & 63;
        const int ti = t / NT, tj = t - ti * NT;
        drgnn_f4 sum = {0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < KS; ++ks) {
            const drgnn_f4 v = *(const drgnn_f4*)(part + ((ks * tiles + t) * 64 + l) * 4);
            sum[0] += v[0]; sum[1] += v[1]; sum[2] += v[2]; sum[3] += v[3];
        }
        const int row = ti * 16 + (l >> 4) * 4;
        float* c = (NTH == 0) ? C + row * ldc + tj * 16 + (l & 15)
                              : C + (tj / NTH) * chalf + row * ldc + (tj % NTH) * 16 + (l & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) if (row + r < Mrows) c[r * ldc] = sum[r];
    }
}
DEV void step_gemm_tn_pair(int MT, int NTH, int K, const float* A, int lda, const float* B, int ldb, int KS, float* part,
                           float* C, int chalf, int ldc, int Mrows) {
    KS = 1 << (31 - __builtin_clz((unsigned)(KS > 0 ? KS : 1)));       // round down to a power of two
    if (NTH == 2 && MT == 1) step_gemm_tn_t<1, 4, 2>(1, 4, K, A, lda, B, ldb, KS, part, C, ldc, Mrows, chalf);
    else if (NTH == 1 && MT == 2) step_gemm_tn_t<2, 2, 1>(2, 2, K, A, lda, B, ldb, KS, part, C, ldc, Mrows, chalf);
    else if (NTH == 1) step_gemm_tn_t<0, 2, 1>(MT, 2, K, A, lda, B, ldb, KS, part, C, ldc, Mrows, chalf);
    else step_gemm_tn_t<0, 4, 2>(MT, 4, K, A, lda, B, ldb, KS, part, C, ldc, Mrows, chalf);
}
// A read from global memory (a_bytes bytes, rows of lda floats), NT = 1
DEV void step_gemm_tn_bufa(int MT, int K, const float* A, int lda, int a_bytes, const float* B, int ldb, int KS, float* part,
                           float* C, int ldc, int Mrows, int stage = 0, int wave_shift = 0) {
    KS = 1 << (31 - __builtin_clz((unsigned)(KS > 0 ? KS : 1)));
    if (MT == 2) step_gemm_tn_t<2, 1, 0, true>(2, 1, K, A, lda, B, ldb, KS, part, C, ldc, Mrows, 0, stage, a_bytes, wave_shift);
    else step_gemm_tn_t<0, 1, 0, true>(MT, 1, K, A, lda, B, ldb, KS, part, C, ldc, Mrows, 0, stage, a_bytes, wave_shift);
}
DEV void step_gemm_tn(int MT, int NT, int K, const float* A, int lda, const float* B, int ldb, int KS, float* part,
                      float* C, int ldc, int Mrows, int stage = 0, int wave_shift = 0) {
    KS = 1 << (31 - __builtin_clz((unsigned)(KS > 0 ? KS : 1)));       // round down to a power of two
    if (MT == 1 && NT == 2) step_gemm_tn_t<1, 2>(1, 2, K, A, lda, B, ldb, KS, part, C, ldc, Mrows, 0, stage, 0, wave_shift);
    else if (MT == 2 && NT == 1) step_gemm_tn_t<2, 1>(2, 1, K, A, lda, B, ldb, KS, part, C, ldc, Mrows, 0, stage);
    else if (MT == 1 && NT == 1) step_gemm_tn_t<1, 1>(1, 1, K, A, lda, B, ldb, KS, part, C, ldc, Mrows, 0, stage);
    else step_gemm_tn_t<0, 0>(MT, NT, K, A, lda, B, ldb, KS, part, C, ldc, Mrows, 0, stage);
}
#endif

// ---- GINet's second convolution, aggregation first ------------------------------------------------
// relu(A (XP W2)) = relu((A XP) W2): summing the 16-wide pooled rows BEFORE the dense product halves the
// bytes the LDS gathers move (64 instead of 128 per edge), forward and backward alike.
// dst[i][0:16] = sum over CSR row i of src[col][0:16]; rows of LD floats, 4 lanes per row
template <int LD, class IdxT>
DEV void step_gather_rows(int n, const int* rp, const IdxT* col, const float* src, float* dst) {
#ifdef DRGNN_EMU
    FOR_TID(item, n * 4) {
        const int i = item >> 2, c = (item & 3) * 4;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        const int lo = rp[i], hi = rp[i + 1];
        for (int k = lo; k < hi; ++k) {
            float v0, v1, v2, v3;
            NET_LD4(true, src + col[k] * LD + c, v0, v1, v2, v3);
            a0 += v0; a1 += v1; a2 += v2; a3 += v3;
        }
        NET_ST4(true, dst + i * LD + c, a0, a1, a2, a3);
    }
#else
    // 16 lanes per pooled node, as in step_gather_scatter below: 4 channel groups x 4 interleaved slices of the
    // row's entry list (few, long rows), slice sums combined in fixed order by two DPP steps
    const int items = ((n * 16) + 63) & ~63;
    for (int item = threadIdx.x; item < items; item += DRGNN_NTHREADS) {
        const int i = item >> 4, sl = (item >> 2) & 3, c = (item & 3) * 4;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (i < n) {
            const int lo = rp[i], hi = rp[i + 1];
            for (int k = lo + sl; k < hi; k += 4) {
                const drgnn_f4 v = *(const drgnn_f4*)(src + ROW24(col[k], LD) + c);
                a0 += v[0]; a1 += v[1]; a2 += v[2]; a3 += v[3];
            }
        }
        a0 += dpp_take<0x128>(a0); a1 += dpp_take<0x128>(a1); a2 += dpp_take<0x128>(a2); a3 += dpp_take<0x128>(a3);
        a0 += dpp_take<0x124>(a0); a1 += dpp_take<0x124>(a1); a2 += dpp_take<0x124>(a2); a3 += dpp_take<0x124>(a3);
        if (sl == 0 && i < n) *(drgnn_f4*)(dst + i * LD + c) = drgnn_f4{a0, a1, a2, a3};
    }
#endif
}
// backward of conv1's aggregation for sGAT / FoutNet with the per-entry coefficients precomputed (ct[t]):
// dU[j, 0:16] = sum_t ct[t] dZ[row(t), :],  dU[j, 16:32] = s_j dZ[j, :]   (rows of 32 floats; dZ rows of 16)
template <int KIND, class IdxT>
DEV void step_aggregate_bwd_ct(int n, const int* deg_rp, const int* cp, const IdxT* ridx, const float* ct, const float* sc,
                               const float* dz, float* du) {
    constexpr int H = DRGNN_H1, HC = 2 * DRGNN_H1;
    FOR_TID(item, n * 4) {
        const int j = item >> 2, c = (item & 3) * 4;
        const int lo = cp[j], hi = cp[j + 1];
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (KIND == DRGNN_SGAT)
        for (int t = lo; t < hi; t += 4) {      // batches of four independent chains, padded under a zero coefficient
            int ii[4];
            float cf[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int tt = (t + q < hi) ? t + q : hi - 1;
                ii[q] = ridx[tt];
                cf[q] = (t + q < hi) ? ct[tt] : 0.0f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v0, v1, v2, v3;
                NET_LD4(true, dz + ii[q] * H + c, v0, v1, v2, v3);
                a0 = fmaf(cf[q], v0, a0); a1 = fmaf(cf[q], v1, a1); a2 = fmaf(cf[q], v2, a2); a3 = fmaf(cf[q], v3, a3);
            }
        }
        else
#pragma unroll 4
        for (int t = lo; t < hi; ++t) {
            const float cf = ct[t];
            float v0, v1, v2, v3;
            NET_LD4(true, dz + ridx[t] * H + c, v0, v1, v2, v3);
            a0 = fmaf(cf, v0, a0); a1 = fmaf(cf, v1, a1); a2 = fmaf(cf, v2, a2); a3 = fmaf(cf, v3, a3);
        }
        float* uj = du + j * HC + c;
        NET_ST4(true, uj, a0, a1, a2, a3);
        float sv = sc[j];
        if (KIND == DRGNN_FOUT && deg_rp[j + 1] == deg_rp[j]) sv = 0.0f;   // NaN row never wins a max
        float d0, d1, d2, d3;
        NET_LD4(true, dz + j * H + c, d0, d1, d2, d3);
        d0 *= sv; d1 *= sv; d2 *= sv; d3 *= sv;
        NET_ST4(true, uj + H, d0, d1, d2, d3);
    }
}

// ---- sGAT / FoutNet second convolution, aggregation first (same idea as GINet's) -------------------------------
//   z_i = s_i (xp_i Wself) + d_i sum_k c_k (xp_col(k) Wnbr) + b  =  [S_i | T_i] [Wnbr ; Wself] + b,
//   S_i = d_i sum_k c_k xp_col(k)   (16-wide gather instead of a 32-wide one),   T_i = s_i xp_i
// so the layer is ONE gather of pooled rows and ONE dense product over K = 32, forward and backward.
// FoutNet's NaN row of a node without out-edges (mean of an empty slice) is NOT materialised in [S | T] (a NaN operand
// would poison the weight-gradient product, NaN x 0): S_i = 0 there and the product's epilogue writes the NaN row of Z2
// (rows whose d_i is 0).
// Forward gather: ts[i] = [S_i | T_i] (rows of LDT floats), coefficients d_i / s_i filed in dv / sc for the backward.
// 16 lanes per pooled node: 4 channel groups x 4 interleaved slices of the entry list, combined by two DPP steps.
template <int KIND, int LDX, int LDT, class IdxT>
DEV void step_pooled_gather(int n, const int* rp, const IdxT* col, const float* w, float* dv, float* sc,
                            const float* xp, float* ts) {
#ifdef DRGNN_EMU
    FOR_TID(item, n * 4) {
        const int i = item >> 2, c = (item & 3) * 4;
        const int lo = rp[i], hi = rp[i + 1], deg = hi - lo;
        float a[4] = {0.f, 0.f, 0.f, 0.f}, asum = 0.0f;
        for (int k = lo; k < hi; ++k) {
            const float cf = (KIND == DRGNN_SGAT) ? w[k] : 1.0f;
            asum += cf;
            for (int q = 0; q < 4; ++q) a[q] = fmaf(cf, xp[col[k] * LDX + c + q], a[q]);
        }
        float d, sv;
        if (KIND == DRGNN_SGAT) { d = 1.0f / (float)(deg > 0 ? deg : 1); sv = asum * d; }
        else { d = deg > 0 ? 1.0f / (float)deg : 0.0f; sv = 1.0f; }
        if (c == 0) { dv[i] = d; sc[i] = sv; }
        for (int q = 0; q < 4; ++q) {
            float v = a[q] * d;
            ts[i * LDT + c + q] = v;
            ts[i * LDT + DRGNN_H1 + c + q] = sv * xp[i * LDX + c + q];
        }
    }
#else
    const int items = ((n * 16) + 63) & ~63;
    for (int item = threadIdx.x; item < items; item += DRGNN_NTHREADS) {
        const int i = item >> 4, sl = (item >> 2) & 3, c = (item & 3) * 4;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, asum = 0.f;
        int lo = 0, hi = 0;
        if (i < n) {
            lo = rp[i]; hi = rp[i + 1];
            for (int k = lo + sl; k < hi; k += 4) {
                const drgnn_f4 v = *(const drgnn_f4*)(xp + ROW24(col[k], LDX) + c);
                float cf = 1.0f;
                if (KIND == DRGNN_SGAT) { cf = w[k]; asum += cf; }
                a0 = fmaf(cf, v[0], a0); a1 = fmaf(cf, v[1], a1); a2 = fmaf(cf, v[2], a2); a3 = fmaf(cf, v[3], a3);
            }
        }
        a0 += dpp_take<0x128>(a0); a1 += dpp_take<0x128>(a1); a2 += dpp_take<0x128>(a2); a3 += dpp_take<0x128>(a3);
        a0 += dpp_take<0x124>(a0); a1 += dpp_take<0x124>(a1); a2 += dpp_take<0x124>(a2); a3 += dpp_take<0x124>(a3);
        if (KIND == DRGNN_SGAT) { asum += dpp_take<0x128>(asum); asum += dpp_take<0x124>(asum); }
        if (sl == 0 && i < n) {
            const int deg = hi - lo;
            float d, sv;
            if (KIND == DRGNN_SGAT) { d = 1.0f / (float)(deg > 0 ? deg : 1); sv = asum * d; }
            else { d = deg > 0 ? 1.0f / (float)deg : 0.0f; sv = 1.0f; }
            if (c == 0) { dv[i] = d; sc[i] = sv; }
            drgnn_f4 S = {a0 * d, a1 * d, a2 * d, a3 * d};
            const drgnn_f4 x = *(const drgnn_f4*)(xp + i * LDX + c);
            *(drgnn_f4*)(ts + i * LDT + c) = S;
            *(drgnn_f4*)(ts + i * LDT + DRGNN_H1 + c) = drgnn_f4{sv * x[0], sv * x[1], sv * x[2], sv * x[3]};
        }
    }
#endif
}
// Backward of the same: d xp_j = s_j dT_j + sum over CSC entries t of column j : c_t d_row(t) dS_row(t), with
// dts[i] = [dS_i | dT_i] (rows of LDT floats), scattered straight through the depth-0 argmax into dZ1 (row stride 16).
// deg_rp: CSR row pointers (a FoutNet node without out-edges produced NaN and never won a max: its self path is masked).
template <int KIND, int LDT, class IdxT>
DEV void step_pooled_gather_bwd(int n, const int* deg_rp, const int* cp, const IdxT* ridx, const IdxT* tslot,
                                const float* w, const float* dv, const float* sc, const float* dts, const short* arg,
                                float* dz) {
#ifdef DRGNN_EMU
    FOR_TID(item, n * 4) {
        const int j = item >> 2, c = (item & 3) * 4;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int t = cp[j]; t < cp[j + 1]; ++t) {
            const int i = ridx[t];
            float cf = dv[i];
            if (KIND == DRGNN_SGAT) cf *= w[tslot[t]];
            for (int q = 0; q < 4; ++q) acc[q] = fmaf(cf, dts[i * LDT + c + q], acc[q]);
        }
        float sv = sc[j];
        if (KIND == DRGNN_FOUT && deg_rp[j + 1] == deg_rp[j]) sv = 0.0f;
        for (int q = 0; q < 4; ++q) {
            const int m = arg[j * DRGNN_H1 + c + q];
            if (m >= 0) dz[m * DRGNN_H1 + c + q] = fmaf(sv, dts[j * LDT + DRGNN_H1 + c + q], acc[q]);
        }
    }
#else
    const int items = ((n * 16) + 63) & ~63;
    for (int item = threadIdx.x; item < items; item += DRGNN_NTHREADS) {
        const int j = item >> 4, sl = (item >> 2) & 3, c = (item & 3) * 4;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (j < n) {
            const int lo = cp[j], hi = cp[j + 1];
            for (int t = lo + sl; t < hi; t += 4) {
                const int i = ridx[t];
                float cf = dv[i];
                if (KIND == DRGNN_SGAT) cf *= w[tslot[t]];
                const drgnn_f4 v = *(const drgnn_f4*)(dts + i * LDT + c);
                a0 = fmaf(cf, v[0], a0); a1 = fmaf(cf, v[1], a1); a2 = fmaf(cf, v[2], a2); a3 = fmaf(cf, v[3], a3);
            }
        }
        a0 += dpp_take<0x128>(a0); a1 += dpp_take<0x128>(a1); a2 += dpp_take<0x128>(a2); a3 += dpp_take<0x128>(a3);
        a0 += dpp_take<0x124>(a0); a1 += dpp_take<0x124>(a1); a2 += dpp_take<0x124>(a2); a3 += dpp_take<0x124>(a3);
        if (sl == 0 && j < n) {
            float sv = sc[j];
            if (KIND == DRGNN_FOUT && deg_rp[j + 1] == deg_rp[j]) sv = 0.0f;
            const drgnn_f4 dt = *(const drgnn_f4*)(dts + j * LDT + DRGNN_H1 + c);
            const float acc[4] = {fmaf(sv, dt[0], a0), fmaf(sv, dt[1], a1), fmaf(sv, dt[2], a2), fmaf(sv, dt[3], a3)};
            const drgnn_u2 packed = *(const drgnn_u2*)(arg + j * DRGNN_H1 + c);      // (one 64-bit read: see step_gather_scatter)
            const int m4[4] = {(short)(packed[0] & 0xffffu), (short)(packed[0] >> 16), (short)(packed[1] & 0xffffu), (short)(packed[1] >> 16)};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (m4[q] >= 0) dz[m4[q] * DRGNN_H1 + c + q] = acc[q];
        }
    }
#endif
}

// the transposed sum (CSC: column j gathers the rows of its entries), scattered straight through the
// depth-0 argmax into dZ1 (row stride 16): the pooling backward needs no pass of its own
template <int LD, class IdxT>
DEV void step_gather_scatter(int n, const int* cp, const IdxT* ridx, const float* src, const short* arg, float* dz) {
#ifdef DRGNN_EMU
    FOR_TID(item, n * 4) {
        const int j = item >> 2, c = (item & 3) * 4;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int t = cp[j]; t < cp[j + 1]; ++t) {
            float v0, v1, v2, v3;
            NET_LD4(true, src + ridx[t] * LD + c, v0, v1, v2, v3);
            acc[0] += v0; acc[1] += v1; acc[2] += v2; acc[3] += v3;
        }
        for (int q = 0; q < 4; ++q) {
            const int m = arg[j * DRGNN_H1 + c + q];
            if (m >= 0) dz[m * DRGNN_H1 + c + q] = acc[q];
        }
    }
#else
    // 16 lanes per pooled node: 4 channel groups x 4 interleaved slices of its entry list (the pooled graph
    // has few, long rows: one lane per (node, group) left 3/4 of the workgroup idle behind ~15-deep
    // dependent gathers); the 4 slice sums are combined in fixed order with two DPP steps
    const int items = ((n * 16) + 63) & ~63;
    for (int item = threadIdx.x; item < items; item += DRGNN_NTHREADS) {
        const int j = item >> 4, sl = (item >> 2) & 3, c = (item & 3) * 4;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (j < n) {
            const int lo = cp[j], hi = cp[j + 1];
            for (int t = lo + sl; t < hi; t += 4) {
                const drgnn_f4 v = *(const drgnn_f4*)(src + ROW24(ridx[t], LD) + c);
                a0 += v[0]; a1 += v[1]; a2 += v[2]; a3 += v[3];
            }
        }
        // the 4 slices of a node sit 4 lanes apart inside a 16-lane row: rotate by 8, then by 4
        a0 += dpp_take<0x128>(a0); a1 += dpp_take<0x128>(a1); a2 += dpp_take<0x128>(a2); a3 += dpp_take<0x128>(a3);
        a0 += dpp_take<0x124>(a0); a1 += dpp_take<0x124>(a1); a2 += dpp_take<0x124>(a2); a3 += dpp_take<0x124>(a3);
        if (sl == 0 && j < n) {
            const float acc[4] = {a0, a1, a2, a3};
            // the four argmax entries of this lane in ONE 64-bit read (8-byte aligned: c is a multiple of 4): four 16-bit reads
            // made four dependent LDS round trips in front of four stores
            const drgnn_u2 packed = *(const drgnn_u2*)(arg + j * DRGNN_H1 + c);
            const int m4[4] = {(short)(packed[0] & 0xffffu), (short)(packed[0] >> 16), (short)(packed[1] & 0xffffu), (short)(packed[1] >> 16)};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (m4[q] >= 0) dz[m4[q] * DRGNN_H1 + c + q] = acc[q];
        }
    }
#endif
}

// Bias gradient = column sums of dZ [n x H] (dense rows of H floats).  Two stages around a barrier the caller has
// anyway: (1) every lane sums a float4 column group over its share of the rows and the lanes of a wave that hold the same
// group are combined by lane exchanges -> one partial row per wave in `wpart` [16][H];
// (2) after the barrier H lanes add the 16 wave rows in wave order.  Fixed order -> bit-reproducible.
template <int H, int LD = H>
DEV void step_colsum_partial(int n, const float* dz, float* wpart) {
#ifdef DRGNN_EMU
    for (int c = 0; c < H; ++c) {
        float acc = 0.0f;
        for (int r = 0; r < n; ++r) acc += dz[r * LD + c];
        wpart[c] = acc;
    }
#else
    // the 64 / G lanes of a wave that share a column group are CONSECUTIVE (8 for H = 32, 16 for H = 16): their partial
    // sums meet on the DPP path (3 - 4 adds per component) instead of 3 - 4 ds_bpermute round trips per component
    constexpr int G = H / 4;                      // column groups (float4 each)
    constexpr int LPG = 64 / G;                   // lanes per column group inside a wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cg = lane / LPG, j = lane % LPG;
    drgnn_f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int r = wave * LPG + j; r < n; r += DRGNN_NWAVES * LPG) {
        const drgnn_f4 v = *(const drgnn_f4*)(dz + r * LD + 4 * cg);
        acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = (LPG == 8) ? lanes8_sum(acc[q]) : lanes16_sum(acc[q]);
    if (j == 0) *(drgnn_f4*)(wpart + wave * H + 4 * cg) = acc;
#endif
}
template <int H>
DEV void step_colsum_finish(const float* wpart, float* out) {
#ifdef DRGNN_EMU
    for (int c = 0; c < H; ++c) out[c] = wpart[c];
#else
    if ((int)threadIdx.x < H) {
        float acc = 0.0f;
#pragma unroll
        for (int w = 0; w < DRGNN_NWAVES; ++w) acc += wpart[w * H + threadIdx.x];
        out[threadIdx.x] = acc;
    }
#endif
}

// per-graph scalars of the loss, fetched during staging:  misc = [bad (int)][y or class id][wy][denom]
#define STEP_M_BAD 0
#define STEP_M_Y 1
#define STEP_M_WY 2
#define STEP_M_DENOM 3

// Depth-1 max-pool with argmax (first maximum in ascending member order, NaN never wins, empty cluster -> 0,
// arg = -1 where no gradient can flow) fused with the graph readout = mean over the depth-1 clusters.
// SKIP0: rows of nodes without out-edges (rp[m+1] == rp[m]) count as NaN, i.e. never win (FoutNet)
// pub: (GINet) this branch's DRGNN_H2 exchange words -- every readout value is published to the partner branch's workgroup
// the moment it exists (tag = index of this step), so that it travels while both workgroups pass the phase's barrier
DEV void xchg_publish(unsigned long long* slot, uint32_t tag, float v);
// a1ld: layout of the argmax array.  0: arg[k][32] (cluster-major: every kernel family but one); > 0: arg[c][a1ld] (column-major,
// a1ld >= C1: drgnn_step3.h -- its readers walk the clusters of ONE column with consecutive lanes, which in the cluster-major
// layout lands 16 lanes on two LDS banks, profiles/r05_lds_conflicts.txt)
template <int LDZ, bool SKIP0 = false>
DEV void step_pool_readout(int C1, const int* mp, const int* mem, const float* z, short* arg, const float* misc,
                           float* xr, float* g_readout, const int* rp = nullptr, unsigned long long* pub = nullptr,
                           uint32_t tag = 0u, int a1ld = 0) {
    int bad; memcpy(&bad, &misc[STEP_M_BAD], 4);
    const float inv = 1.0f / (float)(C1 > 0 ? C1 : 1);
#ifdef DRGNN_EMU
    for (int c = 0; c < DRGNN_H2; ++c) {
        float acc = 0.0f;
        for (int k = 0; k < C1; ++k) {
            float best = DRGNN_NEG_INF;
            int am = -1;
            for (int p = mp[k]; p < mp[k + 1]; ++p) {
                const int m = mem[p];
                if (SKIP0 && rp[m + 1] == rp[m]) continue;
                const float v = z[m * LDZ + c];
                if (v > best) { best = v; am = m; }
            }
            if (am < 0) best = 0.0f;
            arg[k * DRGNN_H2 + c] = (short)((best > 0.0f) ? am : -1);
            acc += best;
        }
        acc *= inv;
        if (bad) acc = DRGNN_NAN;
        xr[c] = acc;
        g_readout[c] = acc;
        if (pub) xchg_publish(pub + c, tag, acc);
    }
#else
    for (int t = threadIdx.x; t < DRGNN_H2 * 16; t += DRGNN_NTHREADS) {      // 512 lanes: whole waves
        const int c = t >> 4, kk = t & 15;
        float acc = 0.0f;
        for (int k = kk; k < C1; k += 16) {
            float best = DRGNN_NEG_INF;
            int am = -1;
            // members in batches of four independent (member -> value) chains; a short last batch repeats the cluster's last
            // member, which cannot win again (strict >): same maximum, same first-maximum argmax -- without the 1 - 3 serial
            // LDS round trips of a remainder loop (SYN's depth-1 clusters have 3 members: all remainder)
            const int plo = mp[k], phi = mp[k + 1];
            for (int p = plo; p < phi; p += 4) {
                int mm[4];
                float vv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) mm[j] = mem[(p + j < phi) ? p + j : phi - 1];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    vv[j] = z[ROW24(mm[j], LDZ) + c];
                    if (SKIP0 && rp[mm[j] + 1] == rp[mm[j]]) vv[j] = DRGNN_NAN;      // (NaN never wins)
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (vv[j] > best) { best = vv[j]; am = mm[j]; }
            }
            if (am < 0) best = 0.0f;
            arg[a1ld ? c * a1ld + k : k * DRGNN_H2 + c] = (short)((best > 0.0f) ? am : -1);
            acc += best;
        }
        acc = lanes16_sum(acc) * inv;
        if (bad) acc = DRGNN_NAN;
        if (kk == 0) {
            if (pub) xchg_publish(pub + c, tag, acc);      // first: the store that has the farthest to go
            xr[c] = acc; g_readout[c] = acc;
        }
    }
#endif
}

// ---- readout exchange between the branch workgroups of a graph -----------------------------------
#ifdef DRGNN_EMU
DEV void xchg_publish(unsigned long long* slot, uint32_t tag, float v) {
    uint32_t bits; memcpy(&bits, &v, 4);
    *slot = ((unsigned long long)tag << 32) | bits;
}
DEV float xchg_wait(unsigned long long* slot, uint32_t tag, int32_t* fault) {      // emulation: the partner's pass 1 is complete
    (void)tag; (void)fault;
    const uint32_t bits = (uint32_t)*slot;
    float v; memcpy(&v, &bits, 4);
    return v;
}
#else
DEV void xchg_publish(unsigned long long* slot, uint32_t tag, float v) {
    const unsigned long long w = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
    __hip_atomic_store(slot, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Spins until the partner has published its value of THIS step.  The branch workgroups of a graph are 8 block ids apart
// (same XCD); up to 64 graphs the launch fits the device in one wave and the partner is resident.  Larger batches rely
// on the dispatcher handing out blocks in id order (it does; HIP does not promise it): the partner then becomes resident
// as soon as any earlier workgroup retires, long before this wait's bound (~0.3 s of the 100 MHz wall clock).  On expiry
// the value is NaN (a NaN loss instead of a hung queue) AND bit DRGNN_FAULT_EXCHANGE is raised in the sticky fault word
// step2[2], which the trainers check once per epoch.
DEV float xchg_wait(unsigned long long* slot, uint32_t tag, int32_t* fault) {
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        const unsigned long long w = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(w >> 32) == tag) return __uint_as_float((uint32_t)w);
        if (wall_clock64() - t0 > 30000000ull) { atomicOr(fault, DRGNN_FAULT_EXCHANGE); return DRGNN_NAN; }
        __builtin_amdgcn_s_sleep(1);
    }
}
// the same in two halves: the first poll is REQUESTED early (its L2 round trip overlaps the caller's own work) ...
DEV unsigned long long xchg_peek(const unsigned long long* slot) {
    return __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ... and completed here: spins only if that first answer was not this step's word yet
DEV float xchg_finish(unsigned long long* slot, unsigned long long w, uint32_t tag, int32_t* fault) {
    if ((uint32_t)(w >> 32) == tag) return __uint_as_float((uint32_t)w);
    return xchg_wait(slot, tag, fault);
}
#endif

// ---- head pieces -----------------------------------------------------------------------------
// Column block of fc1.weight owned by this branch: wb[h][c] = W1[h][br*32 + c]  (LDS, rows of
// STEP_WBLD floats).  Loaded as float4 (8 lanes cover the 128 contiguous bytes of a row).
#define STEP_WB_J 4        // float4 per lane: H * 8 <= STEP_WB_J * 1024  (H <= 512)
#ifdef DRGNN_EMU
template <int WJ> struct WBlockRegs { int dummy; };
template <int WJ> DEV void step_wblock_load(WBlockRegs<WJ>&, const HeadFused&, int) {}
template <int WJ> DEV void step_wblock_store(const WBlockRegs<WJ>&, const HeadFused& hf, int br, float* wb) {
    for (int h = 0; h < hf.H; ++h)
        for (int c = 0; c < DRGNN_H2; ++c) wb[h * STEP_WBLD + c] = hf.w1[(long)h * hf.R + br * DRGNN_H2 + c];
}
#else
// WJ: float4 per lane; 1 covers H <= 128 (the reference heads -- what the width-specialised kernels are launched
// for), STEP_WB_J the general case.  Twelve VGPRs apart, which is what the sGAT / FoutNet kernels spill otherwise.
template <int WJ> struct WBlockRegs { drgnn_f4 v[WJ * DRGNN_BSCALE]; };
template <int WJ> DEV void step_wblock_load(WBlockRegs<WJ>& wr, const HeadFused& hf, int br) {
    const bool vec = ((((uintptr_t)hf.w1) & 15) == 0);       // R = 32*n_branch floats: rows stay 16-byte aligned
#pragma unroll
    for (int j = 0; j < WJ * DRGNN_BSCALE; ++j) {
        const int t = threadIdx.x + j * DRGNN_NTHREADS;
        const int h = t >> 3, q = t & 7;
        drgnn_f4 v = {0.f, 0.f, 0.f, 0.f};
        if (h < hf.H) {
            const float* src = hf.w1 + (long)h * hf.R + br * DRGNN_H2 + 4 * q;
            if (vec) v = *(const drgnn_f4*)src;
            else { v[0] = src[0]; v[1] = src[1]; v[2] = src[2]; v[3] = src[3]; }
        }
        wr.v[j] = v;
    }
}
template <int WJ> DEV void step_wblock_store(const WBlockRegs<WJ>& wr, const HeadFused& hf, int br, float* wb) {
    (void)br;
#pragma unroll
    for (int j = 0; j < WJ * DRGNN_BSCALE; ++j) {
        const int t = threadIdx.x + j * DRGNN_NTHREADS;
        const int h = t >> 3, q = t & 7;
        if (h < hf.H) *(drgnn_f4*)(wb + h * STEP_WBLD + 4 * q) = wr.v[j];
    }
}
#endif

// hid = dropout(relu(b1 + P0 + P1)),  P_br = W1[:, br*32:(br+1)*32] readout_br.  8 lanes per hidden unit, DPP sum inside the
// lane group.  GINet: BOTH branch workgroups of a graph evaluate the whole of fc1 -- the own half from the column block in
// LDS (`wb`, which the head's backward needs anyway) and the own readout, the partner's half from the partner's column
// block held in REGISTERS since the second burst (`wo`) and the partner's readout, which the partner published value by
// value at the end of its pooling phase (step_pool_readout) -- one hand-off of 32 values that is already in flight while
// this workgroup passes the barrier and forms its own half, instead of 128 half products exchanged afterwards.
// `xg`: the [n_branch][DRGNN_H2] exchange words of graph g.  Lanes 0..31 of EVERY wave poll the partner's 32 words (a wave
// cannot learn them from another wave without a barrier) and hand them to the lane groups with four lane reads.
// `part` as in net_step_graph: 2 = from the wait on (emulation passes).
// HC: the head's width as a compile-time constant (0: taken from the descriptor).
template <int HC, int WJ>
DEV void step_head_fc1_t(const HeadFused& hf, int g, int br, int nb, const float* wb, const WBlockRegs<WJ>& wo,
                         const float* b1, const float* xr, float* hid, unsigned long long* xg, uint32_t tag, uint32_t step,
                         uint32_t thresh, float keep_scale, int32_t* fault) {
    const int H = HC ? HC : hf.H;
#ifdef DRGNN_EMU
    (void)wo;
    float xo[DRGNN_H2];
    for (int c = 0; c < DRGNN_H2; ++c) xo[c] = (nb > 1) ? xchg_wait(xg + (long)(1 - br) * DRGNN_H2 + c, tag, fault) : 0.0f;
    for (int h = 0; h < H; ++h) {
        float p = 0.0f, po = 0.0f;
        for (int c = 0; c < DRGNN_H2; ++c) p = fmaf(wb[h * STEP_WBLD + c], xr[c], p);
        if (nb > 1)
            for (int c = 0; c < DRGNN_H2; ++c) po = fmaf(hf.w1[(long)h * hf.R + (1 - br) * DRGNN_H2 + c], xo[c], po);
        float v = (nb > 1) ? ((br == 0) ? p + po : po + p) : p;       // P0 + P1 in both workgroups
        v += b1[h];
        v = v > 0.0f ? v : 0.0f;
        if (thresh) v = drgnn_keep(hf, step, g, H, h, thresh) ? v * keep_scale : 0.0f;
        hid[h] = v;
    }
#else
    const int lane = threadIdx.x & 63;
    unsigned long long* const slot = xg + (long)(1 - br) * DRGNN_H2 + (lane & (DRGNN_H2 - 1));
    unsigned long long w0 = 0ull;
    if (nb > 1 && lane < DRGNN_H2) w0 = xchg_peek(slot);        // in flight behind the own half below
    const int items = (H * 8 + 63) & ~63;         // whole waves: the lane-group sums need every lane
    drgnn_f4 xo = {0.f, 0.f, 0.f, 0.f};
    bool have_other = false;
    int j = 0;
    for (int t = threadIdx.x; t < items; t += DRGNN_NTHREADS, ++j) {
        const int h = t >> 3, q = t & 7;
        float acc = 0.0f;
        if (h < H) {
            const drgnn_f4 w = *(const drgnn_f4*)(wb + h * STEP_WBLD + 4 * q);
            const drgnn_f4 x = *(const drgnn_f4*)(xr + 4 * q);
            acc = fmaf(w[0], x[0], fmaf(w[1], x[1], fmaf(w[2], x[2], w[3] * x[3])));
        }
        acc = lanes8_sum(acc);
        float v = acc;
        if (nb > 1) {
            if (!have_other) {       // (wave-uniform) the partner's readout: lanes 0..31 poll, every lane takes its float4
                float pv = 0.0f;
                PH(7) { if (lane < DRGNN_H2) pv = xchg_finish(slot, w0, tag, fault); }
                const int src = (lane & 7) * 4;
                xo[0] = __shfl(pv, src, 64); xo[1] = __shfl(pv, src + 1, 64);
                xo[2] = __shfl(pv, src + 2, 64); xo[3] = __shfl(pv, src + 3, 64);
                have_other = true;
            }
            float other = 0.0f;
            if (h < H) {
                const drgnn_f4 w = wo.v[j < WJ * DRGNN_BSCALE ? j : 0];
                other = fmaf(w[0], xo[0], fmaf(w[1], xo[1], fmaf(w[2], xo[2], w[3] * xo[3])));
            }
            other = lanes8_sum(other);
            v = (br == 0) ? acc + other : other + acc;       // P0 + P1 in both workgroups
        }
        if (q == 0 && h < H) {
            v += b1[h];
            v = v > 0.0f ? v : 0.0f;
            if (thresh) v = drgnn_keep(hf, step, g, H, h, thresh) ? v * keep_scale : 0.0f;
            hid[h] = v;
        }
    }
#endif
}
// WREF: the fc1 width of the reference net of this kind (128 for GINet, 64 for sGAT / FoutNet) -- the one width a
// kernel carries a specialised copy for besides the generic routines (every copy is instruction-cache footprint)
// ONLY: the host has checked H == WREF for this launch (width-specialised kernels): no generic copy at all
template <int WREF, bool ONLY, int WJ>
DEV void step_head_fc1(const HeadFused& hf, int g, int br, int nb, const float* wb, const WBlockRegs<WJ>& wo, const float* b1,
                       const float* xr, float* hid, unsigned long long* xg, uint32_t tag, uint32_t step,
                       uint32_t thresh, float keep_scale, int32_t* fault) {
    if (ONLY || hf.H == WREF) step_head_fc1_t<WREF, WJ>(hf, g, br, nb, wb, wo, b1, xr, hid, xg, tag, step, thresh, keep_scale, fault);
    else step_head_fc1_t<0, WJ>(hf, g, br, nb, wb, wo, b1, xr, hid, xg, tag, step, thresh, keep_scale, fault);
}


// outs = W2 hid + b2, loss, d loss / d outs, then dhid = relu'/dropout' (W2^T douts).  Device: every
// wave evaluates outs redundantly (lane o keeps outs[o] / douts[o]) so that no barrier separates
// them from their consumers.  Branch 0 writes predictions and the head slab.
template <int HC, int OC>
DEV void step_head_loss_t(const HeadFused& hf, int g, int br, const float* hid, const float* w2, const float* b2,
                          const float* misc, float keep_scale, float* dhid, float* p_dhid, float* p_hw2,
                          float* p_hb2, float* p_loss) {
    const int H = HC ? HC : hf.H, O = OC ? OC : hf.O;
    if (__builtin_expect(!hf.train, 0)) {            // inference: predictions only
        if (br == 0) {
            FOR_TID(o, O) {
                float acc = b2[o];
                for (int h = 0; h < H; ++h) acc = fmaf(hid[h], w2[o * H + h], acc);
                if (hf.sigmoid && hf.task != DRGNN_TASK_CLASS) acc = drgnn_sigmoid(acc);
                hf.pred[(long)g * O + o] = acc;
            }
        }
        return;
    }
    // (DRGNN_TASK_GRAD: d loss / d pred comes from the caller's autograd -- misc[STEP_M_Y] holds it when O == 1, the row
    // hf.y_reg[g * O ..] otherwise -- and the loss slot is written as 0; regression in every other respect)
    const bool ext = hf.task == DRGNN_TASK_GRAD;
    const bool sig = hf.sigmoid && hf.task != DRGNN_TASK_CLASS;
    const float denom = misc[STEP_M_DENOM], wy = misc[STEP_M_WY];
#ifdef DRGNN_EMU
    float outs[DRGNN_MAX_OUT], douts[DRGNN_MAX_OUT];
    for (int o = 0; o < O; ++o) {
        float acc = 0.0f;
        for (int h = 0; h < H; ++h) acc = fmaf(hid[h], w2[o * H + h], acc);
        outs[o] = acc + b2[o];
        if (sig) outs[o] = drgnn_sigmoid(outs[o]);
    }
    float loss = 0.0f, wsum = 1.0f;
    if (ext) {
        for (int o = 0; o < O; ++o) douts[o] = hf.y_reg[(long)g * O + o] * (sig ? outs[o] * (1.0f - outs[o]) : 1.0f);
    } else if (hf.task == DRGNN_TASK_REG) {
        const float inv = 1.0f / (float)(hf.B * O);
        for (int o = 0; o < O; ++o) {
            const float d = outs[o] - misc[STEP_M_Y];
            loss += d * d * inv;
            douts[o] = 2.0f * d * inv * (sig ? outs[o] * (1.0f - outs[o]) : 1.0f);
        }
    } else {
        int yc; memcpy(&yc, &misc[STEP_M_Y], 4);
        float mx = outs[0];
        for (int o = 1; o < O; ++o) mx = outs[o] > mx ? outs[o] : mx;
        float se = 0.0f;
        for (int o = 0; o < O; ++o) se += expf(outs[o] - mx);
        const float lse = logf(se) + mx;
        loss = wy * (lse - outs[yc]) / denom;
        for (int o = 0; o < O; ++o) douts[o] = wy * (expf(outs[o] - lse) - (o == yc ? 1.0f : 0.0f)) / denom;
        wsum = wy;
    }
    if (br == 0) {
        for (int o = 0; o < O; ++o) { hf.pred[(long)g * O + o] = outs[o]; p_hb2[o] = douts[o]; }
        p_loss[0] = loss; p_loss[1] = wsum;
    }
    for (int h = 0; h < H; ++h) {
        float acc = 0.0f;
        const float hv = hid[h];
        for (int o = 0; o < O; ++o) {
            acc = fmaf(douts[o], w2[o * H + h], acc);
            if (br == 0) p_hw2[(long)o * H + h] = douts[o] * hv;
        }
        const float dh = (hv != 0.0f) ? acc * keep_scale : 0.0f;
        dhid[h] = dh;
        if (br == 0) p_dhid[h] = dh;
    }
#else
    // only the waves that own a hidden unit below (wave 0 also writes the predictions) need the outputs: the
    // others would just repeat the same ~150 instructions on the same SIMDs
    if ((int)(threadIdx.x & ~63u) >= H && threadIdx.x >= 64) return;
    const int lane = threadIdx.x & 63;
    if (HC != 0 && HC <= 128 && OC == 1 && hf.task != DRGNN_TASK_CLASS) {
        // The reference heads (one output, MSE): this phase is ONE dependent chain in one or two waves while fourteen wait,
        // so every LDS operand is requested up front (one round trip instead of six), the wave sum runs once (the loss of a
        // single output needs none) and no value travels through a lane read: out and d loss / d out are wave-uniform.
        // Same operations in the same order as the general path below: bit-identical results.
        constexpr int NH = (HC > 0) ? (HC + 63) / 64 : 1;      // (HC == 0 never takes this path)
        float hv[NH], wv[NH];
#pragma unroll
        for (int j = 0; j < NH; ++j) { hv[j] = hid[lane + 64 * j]; wv[j] = w2[lane + 64 * j]; }
        const float bias = b2[0], yv = misc[STEP_M_Y];
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < NH; ++j) acc = fmaf(hv[j], wv[j], acc);
        float out = lanes64_sum(acc) + bias;
        if (sig) out = drgnn_sigmoid(out);
        const float inv = 1.0f / (float)(hf.B * O);
        const float d = out - yv;
        const float dout = (ext ? yv : 2.0f * d * inv) * (sig ? out * (1.0f - out) : 1.0f);
        const int wv_id = (int)(threadIdx.x >> 6);
        float hme = hv[0], wme = wv[0];
#pragma unroll
        for (int j = 1; j < NH; ++j) { if (wv_id == j) { hme = hv[j]; wme = wv[j]; } }
        const int h = (int)threadIdx.x;                       // < HC: the waves past H have returned
        const float dh = (hme != 0.0f) ? fmaf(dout, wme, 0.0f) * keep_scale : 0.0f;     // relu' and dropout mask
        dhid[h] = dh;
        if (br == 0) {
            p_hw2[h] = dout * hme;
            p_dhid[h] = dh;
            if (threadIdx.x == 0) {
                hf.pred[(long)g] = out;
                p_hb2[0] = dout;
                p_loss[0] = ext ? 0.0f : (d * d * inv);
                p_loss[1] = 1.0f;
            }
        }
        return;
    }
    float my_out = 0.0f;
    for (int o = 0; o < O; ++o) {
        float acc = 0.0f;
        for (int h = lane; h < H; h += 64) acc = fmaf(hid[h], w2[o * H + h], acc);
        acc = lanes64_sum(acc) + b2[o];
        if (lane == o) my_out = acc;
    }
    if (sig) my_out = drgnn_sigmoid(my_out);
    float my_dout = 0.0f, loss, wsum = 1.0f;
    if (ext) {
        loss = 0.0f;
        my_dout = lane < O ? hf.y_reg[(long)g * O + lane] * (sig ? my_out * (1.0f - my_out) : 1.0f) : 0.0f;
    } else if (__builtin_expect(hf.task == DRGNN_TASK_REG, 1)) {      // (layout hint: the exp / log code of the other branch goes out of line)
        const float inv = 1.0f / (float)(hf.B * O);
        const float d = my_out - misc[STEP_M_Y];
        loss = lanes64_sum(lane < O ? d * d * inv : 0.0f);
        my_dout = lane < O ? 2.0f * d * inv * (sig ? my_out * (1.0f - my_out) : 1.0f) : 0.0f;
    } else {
        const int yc = __builtin_amdgcn_readfirstlane(__float_as_int(misc[STEP_M_Y]));
        const float mx = lanes64_max(lane < O ? my_out : DRGNN_NEG_INF);
        const float se = lanes64_sum(lane < O ? expf(my_out - mx) : 0.0f);
        const float lse = logf(se) + mx;
        loss = wy * (lse - lane_get(my_out, yc)) / denom;
        my_dout = lane < O ? wy * (expf(my_out - lse) - (lane == yc ? 1.0f : 0.0f)) / denom : 0.0f;
        wsum = wy;
    }
    if (br == 0 && (int)threadIdx.x < O) {
        hf.pred[(long)g * O + threadIdx.x] = my_out;
        p_hb2[threadIdx.x] = my_dout;
    }
    if (br == 0 && threadIdx.x == 0) { p_loss[0] = loss; p_loss[1] = wsum; }
    for (int h0 = 0; h0 < H; h0 += DRGNN_NTHREADS) {      // uniform trip count: lane_get below is wave-wide
        const int h = h0 + (int)threadIdx.x;
        const bool ok = h < H;
        const float hv = ok ? hid[h] : 0.0f;
        float acc = 0.0f;
        for (int o = 0; o < O; ++o) {
            const float dout = lane_get(my_dout, o);
            if (ok) {
                acc = fmaf(dout, w2[o * H + h], acc);
                if (br == 0) p_hw2[(long)o * H + h] = dout * hv;
            }
        }
        if (ok) {
            const float dh = (hv != 0.0f) ? acc * keep_scale : 0.0f;     // relu' and dropout mask
            dhid[h] = dh;
            if (br == 0) p_dhid[h] = dh;
        }
    }
#endif
}
template <int WREF, bool ONLY>
DEV void step_head_loss(const HeadFused& hf, int g, int br, const float* hid, const float* w2, const float* b2,
                        const float* misc, float keep_scale, float* dhid, float* p_dhid, float* p_hw2,
                        float* p_hb2, float* p_loss) {
    if (__builtin_expect(hf.H == WREF && hf.O == 1, 1))
        step_head_loss_t<WREF, 1>(hf, g, br, hid, w2, b2, misc, keep_scale, dhid, p_dhid, p_hw2, p_hb2, p_loss);
    else
        step_head_loss_t<0, 0>(hf, g, br, hid, w2, b2, misc, keep_scale, dhid, p_dhid, p_hw2, p_hb2, p_loss);
}

// d readout (this branch's 32 columns) = dhid wb, scattered straight into dZ2 through the depth-1
// argmax (mean over the C1 clusters -> factor inv)
template <int HC>
DEV void step_head_dreadout_t(const HeadFused& hf, const float* wb, const float* dhid, const short* a1, int C1,
                              float* z2, int ldz) {
    const int H = HC ? HC : hf.H;
    const float inv = 1.0f / (float)(C1 > 0 ? C1 : 1);
#ifdef DRGNN_EMU
    for (int c = 0; c < DRGNN_H2; ++c) {
        float acc = 0.0f;
        for (int h = 0; h < H; ++h) acc = fmaf(dhid[h], wb[h * STEP_WBLD + c], acc);
        for (int k = 0; k < C1; ++k) {
            const int r = a1[k * DRGNN_H2 + c];
            if (r >= 0) z2[r * ldz + c] = acc * inv;
        }
    }
#else
    for (int t = threadIdx.x; t < DRGNN_H2 * 32; t += DRGNN_NTHREADS) {
        const int c = t >> 5, q = t & 31;
        float acc = 0.0f;
        for (int h = q; h < H; h += 32) acc = fmaf(dhid[h], wb[h * STEP_WBLD + c], acc);
        const float v = lanes32_sum(acc) * inv;
        for (int k = q; k < C1; k += 32) {
            const int r = a1[k * DRGNN_H2 + c];
            if (r >= 0) z2[ROW24(r, ldz) + c] = v;
        }
    }
#endif
}
template <int WREF, bool ONLY>
DEV void step_head_dreadout(const HeadFused& hf, const float* wb, const float* dhid, const short* a1, int C1,
                            float* z2, int ldz) {
    if (ONLY || hf.H == WREF) step_head_dreadout_t<WREF>(hf, wb, dhid, a1, C1, z2, ldz);
    else step_head_dreadout_t<0>(hf, wb, dhid, a1, C1, z2, ldz);
}

// strided [K,H] weight -> transposed dense rows dst[h*ld + k]
DEV void step_stage_wt(float* dst, int ld, const float* src, long sk, long sh, int K, int H) {
    FOR_TID(e, K * H) {
        const int k = e / H, h = e % H;
        dst[h * ld + k] = src[(long)k * sk + (long)h * sh];
    }
}
DEV void step_copy_i32(int* dst, const int32_t* src, int n) { FOR_TID(i, n) { dst[i] = src[i]; } }
template <bool NARROW> DEV void step_copy_idx(int* dst, const int32_t* src, int n) {
    if (NARROW) { unsigned short* d16 = (unsigned short*)dst; FOR_TID(i, n) { d16[i] = (unsigned short)src[i]; } }
    else { FOR_TID(i, n) { dst[i] = src[i]; } }
}
DEV void step_copy_f32(float* dst, const float* src, int n) { FOR_TID(i, n) { dst[i] = src[i]; } }

// host side of the same conditions (net_burst_ok + the head's), from the batch-wide bounds
static inline bool step_burst_guaranteed(int kind, const float* x, int F, int capN, int capE, int capC, int H, int O) {
    if (H != ((kind == DRGNN_GINET) ? 128 : 64)) return false;      // the specialised kernels carry the reference head width only
    return ((((uintptr_t)x) & 15) == 0) && (F % 4 == 0) && (F * DRGNN_H1 <= DRGNN_BCAP) && ((long)capN * F <= 16L * DRGNN_BCAP) &&
           (capN + 1 <= DRGNN_BCAP) && (capE <= 2 * DRGNN_BCAP) && (capC * DRGNN_H1 <= 4 * DRGNN_BCAP) &&
           O * H <= 2 * DRGNN_BCAP && H * 8 <= STEP_WB_J * DRGNN_BCAP;
}

// `part`: 0 = whole step (device), 1 = up to the readout publication, 2 = from the head on
// XF: padded feature width F16 as a compile-time constant (16/32/48/64), 0 = taken from the descriptor.
// The strides of the x tile and of conv1's weights and the K loop of conv1's products hang on it;
// with it known the kernel is ~8% faster, so the common widths are instantiated.
// `late`: the graph's offsets and sizes (d_in.n0 / N / e0 / E) came with the launch arguments (host-known) and its
// device-computed counts (clusters of both depths, pooled edges) are still IN FLIGHT in cnt_c / cnt_e1 / cnt_c1: the
// prologue then issues its loads with host-known bounds and resolves the counts afterwards -- one dependent memory round
// trip less at the start of every workgroup (sizes -> arrays becomes a single wave of loads).
// CLS: capacity class of the LDS layout.  0: laid out for the run-time capacities (capN, capE, capC = the maxima of the
// batch, exact fit: that is what lets 200-node graphs into 160 KB at all).  1: the fixed layout STEP_CLS_N / _E / _C -- the
// largest graph shape all three kinds fit at feature widths up to 32 (the aggregation-first kernels: up to 48) -- with every array offset an immediate instead of
// ~40 pinned registers and run-time address arithmetic: 0.2 - 0.4 us per step.  The host takes it whenever the
// batch's maxima lie inside the class (train_step_impl); LDS is one workgroup per CU either way.
#define STEP_CLS_N 200
#define STEP_CLS_E 1024
#define STEP_CLS_C 52
template <int KIND, int XF, bool GATHER = false, int CLS = 0>
DEV void net_step_graph(const StepArgs& a, const GraphDims& d_in, int g, int gi, int br, float* scratch, int capN,
                        int capE, int capC, int part, bool late = false, int cnt_c = 0, int cnt_e1 = 0, int cnt_c1 = 0) {
    if (CLS == 1) { capN = STEP_CLS_N; capE = STEP_CLS_E; capC = STEP_CLS_C; }
    GraphDims d = d_in;
    // bounds of the prologue's loads of the pooled level: the true counts, or (late) what the host knows they cannot
    // exceed while staying inside this graph's workspace segment and the LDS arrays
    const int bC = late ? imin(d.N, capC) : d.C, bE1 = late ? d.E : d.E1, bC1 = late ? imin(d.N, capC) : d.C1;
#ifdef DRGNN_EMU
    // the emulation runs a workgroup in two passes (part 1 / part 2): the counts are plain values there, resolve at once
    if (late) { d.C = imin(cnt_c, capC); d.E1 = imin(cnt_e1, d.E); d.C1 = imin(cnt_c1, capC); }
#endif
    constexpr int HC1 = (KIND == DRGNN_GINET) ? DRGNN_H1 : 2 * DRGNN_H1;
    constexpr int HC2 = (KIND == DRGNN_GINET) ? DRGNN_H2 : 2 * DRGNN_H2;
    const TopoView& tv = a.tv;
    const HeadFused& hf = a.hf;
    // the branch count (hence the readout width) follows from the kind of net
    constexpr int nb = (KIND == DRGNN_GINET) ? 2 : 1;
    constexpr int R = DRGNN_H2 * nb;
    constexpr int WREF = (KIND == DRGNN_GINET) ? 128 : 64;       // ginet.py:136 / sGAT.py:134, foutnet.py:121
    const int F = a.net.n_feat;
    const int H = hf.H, O = hf.O;
    const int F16 = XF ? XF : step_pad16(F), XLD = F16 + 4;
    constexpr int W2NLD = DRGNN_H2 + 4;
    constexpr int TSLD = DRGNN_H2 + 4;                 // rows of the [S | T] operand of sGAT / FoutNet's second convolution
    constexpr bool GIN = (KIND == DRGNN_GINET);
    constexpr bool NARROW = (KIND == DRGNN_SGAT);
    constexpr bool LATE3 = !GIN;                      // backward-only index arrays staged by a third, later burst
    typedef typename StepIdx<NARROW>::type EIdx;      // element type of the edge-indexed LDS arrays
    constexpr int Z2LD = DRGNN_H2 + 4;      // Z2 / dZ2 rows feed dense products (16-byte aligned, conflict-free 128-bit reads)
    StepScratch s = step_carve(scratch, KIND, (XF != 0) ? XF : F, capN, capE, capC, R, (XF != 0) ? WREF : H, O);
    EXIT_AFTER(0);
    WBlockRegs<(XF != 0) ? 1 : STEP_WB_J> wreg;      // XF != 0: H is the reference width (step_burst_guaranteed)
    // GINet: the PARTNER branch's column block of fc1, kept in registers from the second burst to the head (both branch
    // workgroups of a graph evaluate the whole of fc1, see step_head_fc1_t)
    WBlockRegs<(XF != 0) ? 1 : STEP_WB_J> wother;
    int* const dummy = (int*)(s.misc + 64);      // 64 words that absorb discarded lanes' LDS stores
    const uint32_t done = (uint32_t)a.step2[0];
    const uint32_t tag = done + 1u;
    const drgnn_conv_params& c1 = a.net.conv1[br];
    const drgnn_conv_params& c2 = a.net.conv2[br];
    const float* b1 = s.hb1;
    const float* w2 = s.hw2;
    const float* b2 = s.hb2;

    if (part != 2) {
        // ---- one burst of independent loads: everything this graph needs -> LDS ------------
        PHASE_MARK();
        const float* xg = a.x + (long)d.n0 * F;
        // The width-specialised kernels (XF != 0) are only launched when the HOST has established that every graph of
        // the batch takes the register-burst prologue and that the head has the reference width (step_burst_guaranteed):
        // the plain per-array loops and the generic head routines are compiled out of them -- the kernel image is about
        // twice the instruction cache and every kilobyte of it shows.  The generic kernel decides per graph.
        const bool burst = (XF != 0) ? true
                                     : (net_burst_ok(xg, F, d.N, d.E, late ? capC : d.C) && O * H <= 2 * DRGNN_BCAP && H * 8 <= STEP_WB_J * DRGNN_BCAP);
        if (late && !burst) {      // the plain staging loops need the true counts at once
            d.C = WG_UNIFORM(cnt_c); d.E1 = WG_UNIFORM(cnt_e1); d.C1 = WG_UNIFORM(cnt_c1);
        }
        // Burst registers live across the first barrier: the x tile and the conv1 weights are
        // written to LDS at once, conv1's dense product starts, and everything else (index arrays, conv2
        // and head weights) is written to LDS after it -- their memory time hides behind the MFMAs.
        BurstX<4> bx;
        BurstW<1> bw1, bw2, bs1, bs2;
        // The small arrays (offset tables, index lists, head / bias vectors) are staged ONE ARRAY PER WAVE (rt.h,
        // StageJob): stage_job(burst, wave) names wave `wave`'s array of a burst -- at most 16 jobs per burst.
        //   burst 1 (requested with the x tile, filed before conv1's product): what conv1's aggregation and pooling read
        //   burst 2 (requested before conv1's product, filed after the aggregation): the pooled level, the head, and for
        //           GINet the CSC arrays of the backward pass
        //   burst 3 (sGAT / FoutNet: requested with the cluster max, filed after conv2's product): backward-only arrays
        WaveStage wst;
#ifndef DRGNN_EMU
        const int my_wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#endif
        auto stage_job = [&](int burst, int w) -> StageJob {
            const int32_t* const* P = tv.p;
            const int nar = NARROW ? 1 : 0;
            StageJob j = {nullptr, 0, nullptr, 0};
            switch (burst * 16 + w) {
            case 16 + 0: j = StageJob{P[DRGNN_TI_ROWPTR0] + d.rowbase, d.N + 1, s.rp0, 0}; break;
            case 16 + 1: j = stage_half(StageJob{P[DRGNN_TI_COL0] + d.e0, d.E, s.cx0, nar}, 0); break;
            case 16 + 2: j = stage_half(StageJob{P[DRGNN_TI_COL0] + d.e0, d.E, s.cx0, nar}, 1); break;
            case 16 + 3: j = StageJob{P[DRGNN_TI_MPTR0] + d.rowbase, bC + 1, s.mp0, 0}; break;
            case 16 + 4: j = StageJob{P[DRGNN_TI_MEM0] + d.n0, d.N, s.mem0, 0}; break;
            case 16 + 5: if (!GIN) j = StageJob{c1.bias, DRGNN_H1, s.b1, 0}; break;
            case 16 + 6: if (KIND == DRGNN_SGAT) j = stage_half(StageJob{tv.w0 + d.e0, d.E, s.ew0, 0}, 0); break;
            case 16 + 7: if (KIND == DRGNN_SGAT) j = stage_half(StageJob{tv.w0 + d.e0, d.E, s.ew0, 0}, 1); break;

            case 32 + 0: j = StageJob{P[DRGNN_TI_ROWPTR1] + d.rowbase, bC + 1, s.rp1, 0}; break;
            case 32 + 1: j = stage_half(StageJob{P[DRGNN_TI_COL1] + d.e0, bE1, s.cx1, nar}, 0); break;
            case 32 + 2: j = stage_half(StageJob{P[DRGNN_TI_COL1] + d.e0, bE1, s.cx1, nar}, 1); break;
            case 32 + 3: j = StageJob{P[DRGNN_TI_MPTR1] + d.rowbase, bC1 + 1, s.mp1, 0}; break;
            case 32 + 4: j = StageJob{P[DRGNN_TI_MEM1] + d.n0, bC, s.mem1, 0}; break;
            case 32 + 5: j = StageJob{hf.b1, H, s.hb1, 0}; break;
            case 32 + 6: j = stage_half(StageJob{hf.w2, O * H, s.hw2, 0}, 0); break;
            case 32 + 7: j = stage_half(StageJob{hf.w2, O * H, s.hw2, 0}, 1); break;
            case 32 + 8: j = StageJob{hf.b2, O, s.hb2, 0}; break;
            // (waves 9 .. 14 of burst 2: GINet's backward-only CSC arrays; the other nets' bias / pooled edge weights --
            // their backward-only arrays are burst 3)
            case 32 + 9:
                if (GIN) j = StageJob{P[DRGNN_TI_COLPTR0] + d.rowbase, d.N + 1, s.cp0, 0};
                else j = StageJob{c2.bias, DRGNN_H2, s.b2, 0};
                break;
            case 32 + 10:
                if (GIN) j = stage_half(StageJob{P[DRGNN_TI_ROWIDX0] + d.e0, d.E, s.rx0, nar}, 0);
                else if (KIND == DRGNN_SGAT) j = stage_half(StageJob{tv.w1 + d.e0, bE1, s.ew1, 0}, 0);
                break;
            case 32 + 11:
                if (GIN) j = stage_half(StageJob{P[DRGNN_TI_ROWIDX0] + d.e0, d.E, s.rx0, nar}, 1);
                else if (KIND == DRGNN_SGAT) j = stage_half(StageJob{tv.w1 + d.e0, bE1, s.ew1, 0}, 1);
                break;
            case 32 + 12: if (GIN) j = StageJob{P[DRGNN_TI_COLPTR1] + d.rowbase, bC + 1, s.cp1, 0}; break;
            case 32 + 13: if (GIN) j = stage_half(StageJob{P[DRGNN_TI_ROWIDX1] + d.e0, bE1, s.rx1, nar}, 0); break;
            case 32 + 14: if (GIN) j = stage_half(StageJob{P[DRGNN_TI_ROWIDX1] + d.e0, bE1, s.rx1, nar}, 1); break;

            case 48 + 0: if (!GIN) j = StageJob{P[DRGNN_TI_COLPTR0] + d.rowbase, d.N + 1, s.cp0, 0}; break;
            case 48 + 1: if (!GIN) j = stage_half(StageJob{P[DRGNN_TI_ROWIDX0] + d.e0, d.E, s.rx0, nar}, 0); break;
            case 48 + 2: if (!GIN) j = stage_half(StageJob{P[DRGNN_TI_ROWIDX0] + d.e0, d.E, s.rx0, nar}, 1); break;
            case 48 + 3: if (!GIN) j = StageJob{P[DRGNN_TI_COLPTR1] + d.rowbase, bC + 1, s.cp1, 0}; break;
            case 48 + 4: if (!GIN) j = stage_half(StageJob{P[DRGNN_TI_ROWIDX1] + d.e0, bE1, s.rx1, nar}, 0); break;
            case 48 + 5: if (!GIN) j = stage_half(StageJob{P[DRGNN_TI_ROWIDX1] + d.e0, bE1, s.rx1, nar}, 1); break;
            case 48 + 6: if (KIND == DRGNN_SGAT) j = stage_half(StageJob{P[DRGNN_TI_TSLOT0] + d.e0, d.E, s.ts0, nar}, 0); break;
            case 48 + 7: if (KIND == DRGNN_SGAT) j = stage_half(StageJob{P[DRGNN_TI_TSLOT0] + d.e0, d.E, s.ts0, nar}, 1); break;
            case 48 + 8: if (KIND == DRGNN_SGAT) j = stage_half(StageJob{P[DRGNN_TI_TSLOT1] + d.e0, bE1, s.ts1, nar}, 0); break;
            case 48 + 9: if (KIND == DRGNN_SGAT) j = stage_half(StageJob{P[DRGNN_TI_TSLOT1] + d.e0, bE1, s.ts1, nar}, 1); break;
            default: break;
            }
            return j;
        };
#ifdef DRGNN_EMU
        auto stage_request = [&](int burst) { (void)burst; };
        auto stage_file = [&](int burst) { for (int w = 0; w < 16; ++w) stage_copy(stage_job(burst, w)); };
#else
        auto stage_request = [&](int burst) { wstage_load(wst, stage_job(burst, my_wave)); };
        auto stage_file = [&](int burst) { (void)burst; wstage_store(wst); };
#endif
#ifndef DRGNN_EMU
        {   // fetch every workspace pointer in one go: otherwise each array's staging starts with its own
            // kernarg read + wait
            const int32_t* const* P = tv.p;
            asm volatile("" :: "s"(P[DRGNN_TI_ROWPTR0]), "s"(P[DRGNN_TI_COL0]), "s"(P[DRGNN_TI_COLPTR0]),
                         "s"(P[DRGNN_TI_ROWIDX0]), "s"(P[DRGNN_TI_MPTR0]), "s"(P[DRGNN_TI_MEM0]));
            asm volatile("" :: "s"(P[DRGNN_TI_ROWPTR1]), "s"(P[DRGNN_TI_COL1]), "s"(P[DRGNN_TI_COLPTR1]),
                         "s"(P[DRGNN_TI_ROWIDX1]), "s"(P[DRGNN_TI_MPTR1]), "s"(P[DRGNN_TI_MEM1]));
        }
#endif
        // per-graph scalars of the readout / loss phases: requested here so that their latency hides in
        // the burst (every lane asks for the same words; lane 0 files them in LDS after conv1's product)
        // Nothing here may make the compiler WAIT for a load before the burst below is issued (a wait = one more dependent
        // memory round trip in front of the x tile): the regression target travels as raw bits, untouched until lane 0
        // files it; the classification branch (class weight looked up through the label) completes its own loads inside
        // the branch, so no pending load of ITS registers reaches the join.
        int m_bad = 0, m_y = 0;
        float m_wy = 1.0f, m_denom = 1.0f;
#ifndef DRGNN_EMU
        if (my_wave == 0)      // lane 0 files them: the other 15 waves have no use for the five loads
#endif
        {
            // gi: this graph's number in the workspace (= g unless the launch gathers from a cached set)
            m_bad = tv.p[DRGNN_TI_ERR][0] | tv.p[DRGNN_TI_GSTAT][gi] | tv.p[DRGNN_TI_GSTAT][(GATHER ? a.ws_graphs : a.n_graphs) + gi];
            if (__builtin_expect(hf.train && hf.task == DRGNN_TASK_REG, 1)) {
#ifdef DRGNN_EMU
                const float y = hf.y_reg[gi];
                memcpy(&m_y, &y, 4);
#else
                m_y = __builtin_nontemporal_load((const int*)hf.y_reg + gi);
#endif
            } else if (hf.train) {
                m_y = (int)hf.y_cls[gi];
                m_wy = hf.class_w ? hf.class_w[m_y] : 1.0f;
                // CrossEntropyLoss(weight): mean over the sum of the targets' weights
#ifdef DRGNN_EMU
                m_denom = 0.0f;
                for (int q = 0; q < hf.B; ++q) m_denom += hf.class_w ? hf.class_w[hf.y_cls[GATHER ? a.gather_ids[q] : q]] : 1.0f;
#else
                m_denom = (float)hf.B;
                if (hf.class_w && threadIdx.x < 64) {
                    float part_sum = 0.0f;
                    for (int q = threadIdx.x; q < hf.B; q += 64) part_sum += hf.class_w[hf.y_cls[GATHER ? a.gather_ids[q] : q]];
                    m_denom = lanes64_sum(part_sum);
                }
                // complete this branch's loads here (see above)
                m_y = __builtin_amdgcn_readfirstlane(m_y);
                m_wy = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(m_wy)));
                m_bad = __builtin_amdgcn_readfirstlane(m_bad);
#endif
            }
        }
        if (burst) {
            // first burst: only what conv1 needs (x tile, its weights, CSR0, depth-0 member lists)
            burst_load_x(bx, xg, (DRGNN_SKIP == 20) ? 0 : d.N, F);
            burst_load_w(bw1, c1.w_nbr, c1.nbr_sk, c1.nbr_sh, F, DRGNN_H1);
            if (KIND != DRGNN_GINET) burst_load_w(bs1, c1.w_self, c1.self_sk, c1.self_sh, F, DRGNN_H1);
            stage_request(1);
            burst_store_x4(bx, s.xs, XLD);
            burst_store_wt(bw1, s.w1t, XLD);
            if (KIND != DRGNN_GINET) burst_store_wt(bs1, s.ws1t, XLD);
        } else {
            FOR_TID(e, d.N * F) { s.xs[(e / F) * XLD + e % F] = xg[e]; }
            step_stage_wt(s.w1t, XLD, c1.w_nbr, c1.nbr_sk, c1.nbr_sh, F, DRGNN_H1);
            if (GIN) {
                step_stage_wt(s.w2t, STEP_XPLD, c2.w_nbr, c2.nbr_sk, c2.nbr_sh, DRGNN_H1, DRGNN_H2);
                stage_weight(s.w2n, W2NLD, c2.w_nbr, c2.nbr_sk, c2.nbr_sh, DRGNN_H1, DRGNN_H2);
            } else {
                step_stage_wt(s.wc2t, TSLD, c2.w_nbr, c2.nbr_sk, c2.nbr_sh, DRGNN_H1, DRGNN_H2);
                stage_weight(s.wc2n, TSLD, c2.w_nbr, c2.nbr_sk, c2.nbr_sh, DRGNN_H1, DRGNN_H2);
            }
            step_copy_i32(s.rp0, tv.p[DRGNN_TI_ROWPTR0] + d.rowbase, d.N + 1);
            step_copy_idx<NARROW>(s.cx0, tv.p[DRGNN_TI_COL0] + d.e0, d.E);
            step_copy_i32(s.cp0, tv.p[DRGNN_TI_COLPTR0] + d.rowbase, d.N + 1);
            step_copy_idx<NARROW>(s.rx0, tv.p[DRGNN_TI_ROWIDX0] + d.e0, d.E);
            step_copy_i32(s.mp0, tv.p[DRGNN_TI_MPTR0] + d.rowbase, d.C + 1);
            step_copy_i32(s.mem0, tv.p[DRGNN_TI_MEM0] + d.n0, d.N);
            step_copy_i32(s.rp1, tv.p[DRGNN_TI_ROWPTR1] + d.rowbase, d.C + 1);
            step_copy_idx<NARROW>(s.cx1, tv.p[DRGNN_TI_COL1] + d.e0, d.E1);
            step_copy_i32(s.cp1, tv.p[DRGNN_TI_COLPTR1] + d.rowbase, d.C + 1);
            step_copy_idx<NARROW>(s.rx1, tv.p[DRGNN_TI_ROWIDX1] + d.e0, d.E1);
            step_copy_i32(s.mp1, tv.p[DRGNN_TI_MPTR1] + d.rowbase, d.C1 + 1);
            step_copy_i32(s.mem1, tv.p[DRGNN_TI_MEM1] + d.n0, d.C);
            FOR_TID(e, H * DRGNN_H2) {
                s.wb[(e / DRGNN_H2) * STEP_WBLD + e % DRGNN_H2] = hf.w1[(long)(e / DRGNN_H2) * R + br * DRGNN_H2 + e % DRGNN_H2];
            }
            if (nb > 1) step_wblock_load(wother, hf, 1 - br);
            step_copy_f32(s.hb1, hf.b1, H);
            step_copy_f32(s.hw2, hf.w2, O * H);
            step_copy_f32(s.hb2, hf.b2, O);
            if (KIND != DRGNN_GINET) {
                step_stage_wt(s.ws1t, XLD, c1.w_self, c1.self_sk, c1.self_sh, F, DRGNN_H1);
                step_stage_wt(s.wc2t + DRGNN_H1, TSLD, c2.w_self, c2.self_sk, c2.self_sh, DRGNN_H1, DRGNN_H2);
                stage_weight(s.wc2n + DRGNN_H1 * TSLD, TSLD, c2.w_self, c2.self_sk, c2.self_sh, DRGNN_H1, DRGNN_H2);
                step_copy_f32(s.b1, c1.bias, DRGNN_H1);
                step_copy_f32(s.b2, c2.bias, DRGNN_H2);
            }
            if (KIND == DRGNN_SGAT) {
                step_copy_f32(s.ew0, tv.w0 + d.e0, d.E);
                step_copy_f32(s.ew1, tv.w1 + d.e0, d.E1);
                step_copy_idx<NARROW>(s.ts0, tv.p[DRGNN_TI_TSLOT0] + d.e0, d.E);
                step_copy_idx<NARROW>(s.ts1, tv.p[DRGNN_TI_TSLOT1] + d.e0, d.E1);
            }
        }
        // zero padding the predicate-free products rely on: x rows [N, pad4(N)), and (F % 16 != 0) the
        // k columns [F, F16) of the x tile and of the transposed conv1 weights
        FOR_TID(e, (step_pad4(d.N) - d.N) * XLD) { s.xs[d.N * XLD + e] = 0.0f; }
        if (F16 > F) {
            const int padc = F16 - F;
            FOR_TID(e, d.N * padc) { s.xs[(e / padc) * XLD + F + e % padc] = 0.0f; }
            FOR_TID(e, DRGNN_H1 * padc) {
                s.w1t[(e / padc) * XLD + F + e % padc] = 0.0f;
                if (KIND != DRGNN_GINET) s.ws1t[(e / padc) * XLD + F + e % padc] = 0.0f;
            }
        }
        BARRIER();
        EXIT_AFTER(1);
        if (late) {      // the counts have landed with the first burst
            d.C = WG_UNIFORM(cnt_c); d.E1 = WG_UNIFORM(cnt_e1); d.C1 = WG_UNIFORM(cnt_c1);
            if (d.C > capC || d.E1 > d.E || d.C1 > capC) {      // malformed input (flagged by the builder): stay inside LDS, poison
                d.C = imin(d.C, capC); d.E1 = imin(d.E1, d.E); d.C1 = imin(d.C1, capC);
                m_bad |= 1;
            }
        }

        // ---- forward ------------------------------------------------------------------
        if (burst) {
            // burst 1 has landed with the x tile: file it (its registers serve burst 2), then request burst 2, in
            // flight behind conv1's product and aggregation: everything the later phases use
            stage_file(1);
            burst_load_w(bw2, c2.w_nbr, c2.nbr_sk, c2.nbr_sh, DRGNN_H1, DRGNN_H2);
            step_wblock_load(wreg, hf, br);
            if (nb > 1) step_wblock_load(wother, hf, 1 - br);
            if (KIND != DRGNN_GINET) burst_load_w(bs2, c2.w_self, c2.self_sk, c2.self_sh, DRGNN_H1, DRGNN_H2);
            stage_request(2);
        }
        if (KIND == DRGNN_GINET) {
            PH(1) step_gemm_nn(d.N, 1, F16, s.xs, XLD, s.w1t, XLD, s.u1, HC1, dummy);
        } else {
            PH(1) step_gemm_nn_dual(d.N, F16, s.xs, XLD, s.w1t, s.ws1t, XLD, s.u1, HC1, dummy);
        }
        FOR_TID(e, (step_pad4(d.C) - d.C) * STEP_XPLD) { s.xp[d.C * STEP_XPLD + e] = 0.0f; }
        // per-graph scalars of the readout / loss phases (fetched with the burst, see above)
        FOR_TID(i, 1) {
#ifdef DRGNN_EMU
            memcpy(&s.misc[STEP_M_BAD], &m_bad, 4);
            memcpy(&s.misc[STEP_M_Y], &m_y, 4);
#else
            ((int*)s.misc)[STEP_M_BAD] = m_bad;       // whole words: a byte-wise copy makes the compiler take the value
            ((int*)s.misc)[STEP_M_Y] = m_y;           // apart where it is LOADED (= a wait in the prologue)
#endif
            s.misc[STEP_M_WY] = m_wy;
            s.misc[STEP_M_DENOM] = m_denom;
        }
        BARRIER();
        EXIT_AFTER(2);
        // (the per-row coefficients of sGAT / FoutNet are formed inside the aggregation from the row's own entries)
        PH(2) net_aggregate<KIND, DRGNN_H1, true, 0, EIdx, true>(d.N, s.rp0, (const EIdx*)s.cx0, s.ew0, s.dv0, s.sc0, s.u1, s.b1, s.z1);
        if (burst) {      // the second burst has landed by now: file it in LDS
            if (GIN) {
                burst_store_wt(bw2, s.w2t, STEP_XPLD);
                burst_store_w(bw2, s.w2n, W2NLD);
            } else {
                burst_store_wt(bw2, s.wc2t, TSLD);                         // wc2t[n][k] = Wnbr[k][n]
                burst_store_w(bw2, s.wc2n, TSLD);                          // wc2n[k][n] = Wnbr[k][n]
            }
            step_wblock_store(wreg, hf, br, s.wb);
            if (KIND != DRGNN_GINET) {
                burst_store_wt(bs2, s.wc2t + DRGNN_H1, TSLD);              // wc2t[n][16 + k] = Wself[k][n]
                burst_store_w(bs2, s.wc2n + DRGNN_H1 * TSLD, TSLD);       // wc2n[16 + k][n] = Wself[k][n]
            }
            stage_file(2);
        }
        BARRIER();
        EXIT_AFTER(3);
        if (burst && LATE3) {
            // third burst: the arrays only the BACKWARD pass reads (CSC of both levels, sGAT's transposed slot maps).
            // Requested here, filed two phases later: their registers are not alive during the crowded second burst
            stage_request(3);
        }
        PH(3) net_cluster_max<DRGNN_H1, STEP_XPLD, short>(d.C, s.mp0, s.mem0, s.z1, s.xp, nullptr, s.a0);

        BARRIER();
        EXIT_AFTER(4);
        if (GIN) {      // S = A XP (16-wide gather), kept in the u2 area with rows of STEP_XPLD floats
            PH(4) step_gather_rows<STEP_XPLD, EIdx>(d.C, s.rp1, (const EIdx*)s.cx1, s.xp, s.u2);
            FOR_TID(e, (step_pad4(d.C) - d.C) * STEP_XPLD) { s.u2[d.C * STEP_XPLD + e] = 0.0f; }
        } else {        // [S | T]: aggregated neighbours and scaled self rows of the pooled features (u2 area, rows of TSLD)
            PH(4) step_pooled_gather<KIND, STEP_XPLD, TSLD, EIdx>(d.C, s.rp1, (const EIdx*)s.cx1, s.ew1, s.dv1, s.sc1, s.xp, s.u2);
            FOR_TID(e, (step_pad4(d.C) - d.C) * TSLD) { s.u2[d.C * TSLD + e] = 0.0f; }
        }
        FOR_TID(item, d.N * DRGNN_H1) { s.z1[item] = 0.0f; }      // Z1 is consumed: becomes dZ1
        // node rows [n, pad4(n)) of the backward products' K operands: zero (never written otherwise)
        FOR_TID(e, (step_pad4(d.N) - d.N) * HC1) { s.u1[d.N * HC1 + e] = 0.0f; }
        BARRIER();
        EXIT_AFTER(5);
        if (GIN) {      // Z2 = relu(S W2)
            PH(5) step_gemm_nn<true>(d.C, 2, DRGNN_H1, s.u2, STEP_XPLD, s.w2t, STEP_XPLD, s.z2, Z2LD, dummy);
        } else {        // Z2 = relu([S | T] [Wnbr ; Wself] + b): one product over K = 32
            PH(5) step_gemm_nn<true>(d.C, 2, DRGNN_H2, s.u2, TSLD, s.wc2t, TSLD, s.z2, Z2LD, dummy, s.b2,
                                     (KIND == DRGNN_FOUT) ? s.dv1 : nullptr);      // dv == 0 <=> no out-edges
        }
        if (burst && LATE3) stage_file(3);
        BARRIER();
        EXIT_AFTER(6);
        // depth-1 cluster max (+ argmax) and the graph readout (mean over those clusters) in one phase: 16 lanes
        // per channel share the clusters k = lane, lane+16, ..; their partial sums meet in a 16-lane DPP sum
        PH(6) step_pool_readout<Z2LD>(d.C1, s.mp1, s.mem1, s.z2, s.a1, s.misc, s.xr,
                                      const_cast<float*>(hf.readout) + (long)g * R + br * DRGNN_H2, nullptr,
                                      (nb > 1) ? a.xchg + (long)g * nb * H + br * DRGNN_H2 : nullptr, tag);
        if (!GIN && hf.train) {
            // coefficient of every entry of the TRANSPOSED level-0 aggregation, once (one item per entry, no dependent
            // chain; the CSC arrays have just been filed): the backward gather then reads (row, coefficient) pairs like GINet's reads rows
            FOR_TID(t, d.E) {
                const int i = ((const EIdx*)s.rx0)[t];
                float cf = s.dv0[i];
                if (KIND == DRGNN_SGAT) cf *= s.ew0[((const EIdx*)s.ts0)[t]];
                s.ct0[t] = cf;
            }
        }
        BARRIER();
        EXIT_AFTER(8);
    }

    // ---- FC head + loss + their backward ---------------------------------------------------
    const float keep_scale = (hf.p_drop > 0.0f) ? 1.0f / (1.0f - hf.p_drop) : 1.0f;
    const double pt = (double)hf.p_drop * 4294967296.0;
    const uint32_t thresh = (hf.p_drop > 0.0f) ? (uint32_t)(pt > 4294967295.0 ? 4294967295.0 : pt) : 0u;
    float* hp = hf.partials + (long)g * head_compact_floats(R, H, O);
    float* p_dhid = hp;
    float* p_hw2 = p_dhid + H;
    float* p_hb2 = p_hw2 + (long)O * H;
    float* p_loss = p_hb2 + O;
    if (hf.train && part != 2 && g == 0 && br == 0) { FOR_TID(i, 1) { a.step2[1] = (int32_t)tag; } }     // Adam's step index
    if (part == 1) return;      // (emulation: the readout is published, the partner's pass 1 completes before pass 2 starts)
    FOR_TID(item, step_pad4(d.C) * Z2LD) { s.z2[item] = 0.0f; }      // Z2 is consumed: becomes dZ2 (+ zero K padding)
    // fc1 on [own readout | the partner's readout, published at the end of its pooling phase], hid
    PH(8) step_head_fc1<WREF, (XF != 0)>(hf, g, br, nb, s.wb, wother, b1, s.xr, s.hid, a.xchg + (long)g * nb * H, tag, done, thresh,
                        keep_scale, a.step2 + 2);
    BARRIER();
    EXIT_AFTER(9);
    if (nb > 1 && !hf.train) {
        // inference launches all carry the same tag (the step counter does not move): the reader clears the words it
        // consumed (all 16 waves have, past the barrier), so that the next launch cannot pick up this one's values.
        // Training launches skip it -- their tag changes every step
        FOR_TID(c, DRGNN_H2) {
#ifdef DRGNN_EMU
            a.xchg[(long)g * nb * H + (1 - br) * DRGNN_H2 + c] = 0ull;
#else
            __hip_atomic_store(a.xchg + (long)g * nb * H + (1 - br) * DRGNN_H2 + c, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        }
    }
    // (loss and d readout as ONE phase -- every wave forming the loss redundantly, dhid recomputed inside the d-readout items --
    // was measured: 2.0 us for the merged phase against 0.7 + 0.85 us for the two; not kept)
    PH(9) step_head_loss<WREF, (XF != 0)>(hf, g, br, s.hid, w2, b2, s.misc, keep_scale, s.dhid, p_dhid, p_hw2, p_hb2, p_loss);
    if (!hf.train) return;
    BARRIER();
    EXIT_AFTER(10);
    PH(10) step_head_dreadout<WREF, (XF != 0)>(hf, s.wb, s.dhid, s.a1, d.C1, s.z2, Z2LD);
    BARRIER();
    EXIT_AFTER(11);

    // ---- backward body ---------------------------------------------------------------------
    float* part_w = a.partials + ((long)g * nb + br) * a.n_partial;
    float* p_w1n = part_w;
    float* p_w1s = p_w1n + (long)F * DRGNN_H1;
    float* p_b1 = p_w1s + (long)F * DRGNN_H1;
    float* p_w2n = p_b1 + DRGNN_H1;
    float* p_w2s = p_w2n + DRGNN_H1 * DRGNN_H2;
    float* p_b2 = p_w2s + DRGNN_H1 * DRGNN_H2;
    const int gp_units = step_gp_words(H) / 256;                 // (tile, K slice) units the partial-tile area holds
    const int KS2 = imin(DRGNN_NWAVES / 2, gp_units / 2);        // dW2: 2 tiles
    if (GIN) {
        // dS = dZ2 W2^T (into the p2 area, rows of STEP_XPLD floats);  dW2 = S^T dZ2 (K = pooled nodes)
        PH(11) step_gemm_nn(d.C, 1, DRGNN_H2, s.z2, Z2LD, s.w2n, W2NLD, s.p2, STEP_XPLD, dummy);
        // (the partial tiles here, their sum behind the phase's own barrier: the product adds no barrier to the chain)
        PH(12) step_gemm_tn(1, 2, d.C, s.u2, STEP_XPLD, s.z2, Z2LD, KS2, s.gp, p_w2n, DRGNN_H2, DRGNN_H1, 1);
        BARRIER();
        EXIT_AFTER(12);
        PH(12) step_gemm_tn(1, 2, d.C, s.u2, STEP_XPLD, s.z2, Z2LD, KS2, s.gp, p_w2n, DRGNN_H2, DRGNN_H1, 2);
        // dXP = A^T dS, scattered through the depth-0 argmax into dZ1
        PH(13) step_gather_scatter<STEP_XPLD, EIdx>(d.C, s.cp1, (const EIdx*)s.rx1, s.p2, s.a0, s.z1);
        BARRIER();
        EXIT_AFTER(14);
    } else {
    // d[S | T] = dZ2 [Wnbr ; Wself]^T (into the p2 area, rows of TSLD);  d[Wnbr ; Wself] = [S | T]^T dZ2 (K = pooled nodes):
    // the slab holds dW2nbr and dW2self back to back, i.e. exactly the 32 x 32 result
    PH(11) step_gemm_nn(d.C, 2, DRGNN_H2, s.z2, Z2LD, s.wc2n, TSLD, s.p2, TSLD, dummy);
    PH(12) step_gemm_tn(2, 2, d.C, s.u2, TSLD, s.z2, Z2LD, imin(DRGNN_NWAVES / 4, gp_units / 4), s.gp, p_w2n, DRGNN_H2,
                        2 * DRGNN_H1, 1);
    step_colsum_partial<DRGNN_H2, Z2LD>(d.C, s.z2, s.bsum);     // db2, stage 1
    BARRIER();
    EXIT_AFTER(12);
    PH(12) step_gemm_tn(2, 2, d.C, s.u2, TSLD, s.z2, Z2LD, imin(DRGNN_NWAVES / 4, gp_units / 4), s.gp, p_w2n, DRGNN_H2,
                        2 * DRGNN_H1, 2);
    step_colsum_finish<DRGNN_H2>(s.bsum, p_b2);
    // dXP = s dT + (d c A)^T dS, scattered through the depth-0 argmax into dZ1
    PH(13) step_pooled_gather_bwd<KIND, TSLD, EIdx>(d.C, s.rp1, s.cp1, (const EIdx*)s.rx1, (const EIdx*)s.ts1, s.ew1, s.dv1,
                                                    s.sc1, s.p2, s.a0, s.z1);
    BARRIER();
    EXIT_AFTER(14);
    }
    if (GIN) {
        PH(15) net_aggregate_bwd<KIND, DRGNN_H1, true, 0, EIdx>(d.N, s.rp0, s.cp0, (const EIdx*)s.rx0, (const EIdx*)s.ts0, s.ew0, s.dv0, s.sc0, s.z1, s.u1);
    } else {
        PH(15) step_aggregate_bwd_ct<KIND, EIdx>(d.N, s.rp0, s.cp0, (const EIdx*)s.rx0, s.ct0, s.sc0, s.z1, s.u1);
    }
    if (KIND != DRGNN_GINET) step_colsum_partial<DRGNN_H1>(d.N, s.z1, s.bsum);     // db1, stage 1
    BARRIER();
    EXIT_AFTER(15);
    if (KIND != DRGNN_GINET) step_colsum_finish<DRGNN_H1>(s.bsum, p_b1);
    {   // dW1 = X^T dU1: K = nodes of the graph, split in slices over the waves
        const int mtiles = F16 >> 4;
        int KS = imin(DRGNN_NWAVES / mtiles, gp_units / mtiles);
        if (KS < 1) KS = 1;
        if (KIND == DRGNN_GINET) {
            PH(16) step_gemm_tn(mtiles, 1, d.N, s.xs, XLD, s.u1, HC1, KS, s.gp, p_w1n, DRGNN_H1, F);
        } else if (2 * mtiles <= gp_units) {      // [dU1n | dU1s] in one pass
            int KSP = imin(DRGNN_NWAVES / (2 * mtiles), gp_units / (2 * mtiles));
            if (KSP < 1) KSP = 1;
            PH(16) step_gemm_tn_pair(mtiles, 1, d.N, s.xs, XLD, s.u1, HC1, KSP, s.gp, p_w1n, F * DRGNN_H1, DRGNN_H1, F);
        } else {                                   // very wide inputs: the partial-tile area holds one product at a time
            step_gemm_tn(mtiles, 1, d.N, s.xs, XLD, s.u1, HC1, KS, s.gp, p_w1n, DRGNN_H1, F);
            BARRIER();
            step_gemm_tn(mtiles, 1, d.N, s.xs, XLD, s.u1 + DRGNN_H1, HC1, KS, s.gp, p_w1s, DRGNN_H1, F);
        }
    }
}

#endif
