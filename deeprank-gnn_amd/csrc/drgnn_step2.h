// drgnn_step2.h -- fused training step of the single-branch nets (sGAT / FoutNet), AGGREGATION FIRST and NODE SPLIT:
// one or TWO workgroups per graph.
//
// Why: at batch 64 a single-branch net used 64 of the 256 CUs for step work, and its dense
// phases are bound by ONE CU's fp32 MFMA rate, its gather phases by one CU's LDS rate.  Two changes of formulation make a
// graph divisible between two workgroups with three small hand-offs and NO recomputation:
//
//  (1) conv1 aggregates FIRST (as conv2 already does, drgnn_step.h):
//          z_i = relu( G_i Wn + s_i x_i Ws + b ),   G_i = d_i sum_{e: row = i} c_e x_col(e)          (sGAT.py:62-93, foutnet.py:56-82)
//      G is a gather of x rows of the node's neighbours, so a workgroup needs the x tile (an input) but nothing the partner
//      computes.  And because x is a leaf (no d loss / d x), the backward of conv1 needs NO transposed aggregation at all:
//          dWn = G^T dZ1,   dWs = X^T (s . dZ1),   db = colsum dZ1
//      -- CSC0, its slot map and the per-entry coefficient array disappear from the step (and from LDS).
//      G itself depends on the inputs only (edge_attr is data, sGAT.py:76): the topology builder forms the raw sums S_i =
//      sum_e c_e x_col(e), D_i = d_i and C_i = s_i (DRGNN_TOPO_TILES, include/drgnn.h) in the workgroups co-launched with the
//      PREVIOUS step (cached topology: once per graph), and this kernel's prologue loads the S and x rows of its nodes, filing
//      row i at its hierarchical position (IHORD); conv1 is then Z1 = relu(D . (S Wn) + C . (x Ws) + b) and its backward
//      dWn = S^T (D . dZ1), dWs = X^T (C . dZ1).
//  (2) rows are kept in the HIERARCHICAL node order the topology builder emits (DRGNN_TI_HORD / HMP0 / HSPLIT): members of a
//      depth-0 cluster are consecutive rows, the depth-0 clusters of a depth-1 cluster consecutive runs.  Both poolings are
//      maxima over CONTIGUOUS rows (no member lists), and a prefix of the depth-1 clusters -- the split point the builder
//      chose, closest to half the nodes -- is a prefix of the rows: half h owns whole depth-1 clusters, hence whole depth-0
//      clusters, hence both poolings are local to a workgroup.
//
// What crosses between the two workgroups of a graph (tagged 64-bit words, the protocol of GINet's readout exchange:
// one relaxed agent-scope atomic per value, tag = index of this step, valid exactly when the tag matches):
//      forward   pooled features xp of the own depth-0 clusters      (C_h x 16 values)  -> the pooled gather reads all of xp
//      forward   partial readout sums over the own depth-1 clusters  (32 values)        -> both evaluate the head redundantly
//      backward  d_i dS_i of the own pooled rows                      (C_h x 16 values)  -> the transposed pooled gather
// Every value is published by the lane that forms it, the receiver's first poll is requested BEFORE the phase barrier.
// The two workgroups are launched only while every workgroup of the launch is resident (train_step_impl), 8 block ids apart
// (same XCD, like GINet's branch workgroups).  SPLIT = 1 is the same kernel with one workgroup owning everything (no
// exchange): the layout beyond the resident batch size.
//
// Gradient slabs: one per (graph, half); the update kernel sums 2 B slabs (fixed order).  Half 0 writes the head slab,
// the predictions and the readout.  GPU only: the host emulation keeps stepping these nets through drgnn_step.h.
// Instantiated per padded feature width 16 / 32 / 48 / 64 (any feature count up to 64: padded tile rows, and with F % 4 != 0 a
// padded copy of x in the tiles), SPLIT 1 / 2 for training, SPLIT 1 for inference launches (TRAIN = false).
#ifndef DRGNN_STEP2_H
#define DRGNN_STEP2_H

#include "drgnn_step.h"

#ifndef DRGNN_EMU

struct Step2Scratch {
    float* misc; float* xr; float* hid; float* dhid; float* hb1; float* bsum; float* wb;
    float* w1t; float* ws1t; float* b1; float* wc2t; float* wc2n; float* b2;
    float* xs;
    int* hord;
    int* hmp; int* cid; int* mp1;
    int* rp1; int* cx1; float* ew1; int* cp1; int* rx1; int* ts1;
    short* a0; short* a1;
    float* G; float* z1; float* dv0; float* sc0;
    float* xp; float* dsf; float* u2; float* z2; float* dt; float* dv1; float* sc1;
    float* hw2; float* hb2;
    float* end; float* gp;
};

#define STEP2_TSLD (DRGNN_H2 + 4)
// Z1 (conv1's activations, [capN + 4][16]) is dead once the depth-0 cluster max is formed; [S | T] (u2) and Z2 / dZ2 (z2),
// [capC + 4][36] each, are first written one barrier later: they share one region (13 KB at SYN size -- what lets 200-node
// graphs with 48 features into the 160 KB)
#define STEP2_Z1U_WORDS(capN, capC) \
    ((long)((capN) + 4) * DRGNN_H1 > 2L * ((capC) + 4) * STEP2_TSLD ? (long)((capN) + 4) * DRGNN_H1 : 2L * ((capC) + 4) * STEP2_TSLD)
#define STEP2_CARVE_LIST(X)                                                                    \
    X(misc, 128, 1)                                                                            \
    X(xr, 2 * DRGNN_H2, 1)                                                                     \
    X(hid, H, 1)                                                                               \
    X(dhid, H, 1)                                                                              \
    X(hb1, H, 1)                                                                               \
    X(bsum, DRGNN_NWAVES * DRGNN_H2, 1)                                                        \
    X(wb, step_gp_words((int)H), 1)                                                            \
    X(w1t, DRGNN_H1 * xld, 1)                                                                  \
    X(ws1t, DRGNN_H1 * xld, 1)                                                                 \
    X(b1, DRGNN_H1, 1)                                                                         \
    X(wc2t, DRGNN_H2 * STEP2_TSLD, 1)                                                          \
    X(wc2n, DRGNN_H2 * STEP2_TSLD, 1)                                                          \
    X(b2, DRGNN_H2, 1)                                                                         \
    X(xs, (long)(capN + 4) * xld, !xg)                                                         \
    X(hord, capN, xg)                                                                          \
    X(hmp, capC + 1, 1)                                                                        \
    X(cid, capC, 1)                                                                            \
    X(mp1, capC + 1, 1)                                                                        \
    X(rp1, capC + 1, 1)                                                                        \
    X(cx1, (sg ? (capE + 1) / 2 : capE), 1)                                                    \
    X(ew1, capE, sg)                                                                           \
    X(cp1, capC + 1, 1)                                                                        \
    X(rx1, (sg ? (capE + 1) / 2 : capE), 1)                                                    \
    X(ts1, (sg ? (capE + 1) / 2 : capE), sg)                                                   \
    X(a0, ((long)capC * DRGNN_H1 + 1) / 2, 1)                                                  \
    X(a1, ((long)capC * DRGNN_H2 + 1) / 2, 1)                                                  \
    X(G, (long)(capN + 4) * xld, xg < 2)                                                       \
    X(z1, STEP2_Z1U_WORDS(capN, capC), 1)                                                      \
    X(dv0, capN + 4, 1)                                                                        \
    X(sc0, capN + 4, 1)                                                                        \
    X(xp, (long)(capC + 4) * STEP_XPLD, 1)                                                     \
    X(dsf, (long)(capC + 4) * STEP_XPLD, 1)                                                    \
    X(dt, (long)(capC + 4) * STEP_XPLD, 1)                                                     \
    X(dv1, capC + 4, 1)                                                                        \
    X(sc1, capC + 4, 1)                                                                        \
    X(hw2, (long)O * H, 1)                                                                     \
    X(hb2, O, 1)

#endif  // !DRGNN_EMU

// (host + device; also compiled by the emulation build, whose plan function must answer "never" consistently)
// xg: the FROM-MEMORY forms (net_step2_graph<..., XG = 1 | 2>, run-time layout).  1: the x rows are not staged (the self
// product and the sparse dWs read them from memory: L2), the hierarchical order is (position -> node) instead; 2: neither are
// the S rows of the tiles (conv1's neighbour product and dWn read them the same way)
HD int64_t step2_scratch_words(int kind, int64_t F, int64_t capN, int64_t capE, int64_t capC, int64_t H, int64_t O, int xg = 0) {
    const int sg = (kind == DRGNN_SGAT) ? 1 : 0;
    const int64_t xld = step_pad16((int)F) + 4;
    int64_t w = 0;
#ifndef DRGNN_EMU
#define X(name, words, cond) w += (cond) ? (((int64_t)(words) + 3) & ~(int64_t)3) : 0;
    STEP2_CARVE_LIST(X)
#undef X
#else
    (void)sg; (void)xld; (void)capN; (void)capE; (void)capC; (void)H; (void)O; (void)xg;
    w = (int64_t)1 << 40;      // the emulation build has no node-split kernels
#endif
    return w + 16;
}
// exchange words (uint64) one graph of the split layout needs: [2 halves][capC x 16] pooled features, the same for d dS,
// [2][32] partial readouts
HD int64_t step2_xchg_words(int64_t capC) { return 4 * capC * DRGNN_H1 + 2 * DRGNN_H2; }

#ifndef DRGNN_EMU

// ---- prologue: the S rows of the graph (node order, global) -> G rows at their hierarchical positions -------------------------
// The position of every row travels with the burst: lane l's j-th float4 belongs to node row (l + j * threads) / (F / 4).
template <int J> struct BurstRowMap { int r[J * DRGNN_BSCALE]; };
template <int J> DEV void burst_load_rowmap(BurstRowMap<J>& m, const BurstX<J>& b, const int32_t* map, int nrows) {
    const FastDiv fd = fastdiv_make(b.F >> 2);
#pragma unroll
    for (int j = 0; j < J * DRGNN_BSCALE; ++j) {
        if (j > 0 && b.n4 <= j * DRGNN_NTHREADS) break;
        const int q = threadIdx.x + j * DRGNN_NTHREADS;
        int v = (q < b.n4) ? map[fastdiv(fd, q)] : 0;
        m.r[j] = ((unsigned)v < (unsigned)nrows) ? v : 0;      // (a position outside the graph can only come from a workspace the builder refused: stay inside LDS)
    }
}
// rows whose position lies in [base, base + count) go to LDS row (position - base); the others are not this workgroup's
template <int J> DEV void burst_store_x4_rows(const BurstX<J>& b, const BurstRowMap<J>& m, float* dst, int ld, int base = 0,
                                              int count = 0x7fffffff) {
    const FastDiv fd = fastdiv_make(b.F >> 2);
#pragma unroll
    for (int j = 0; j < J * DRGNN_BSCALE; ++j) {
        if (j > 0 && b.n4 <= j * DRGNN_NTHREADS) break;
        const int q = threadIdx.x + j * DRGNN_NTHREADS;
        if (q < b.n4 && (unsigned)(m.r[j] - base) < (unsigned)count) {
            const int row = fastdiv(fd, q);
            *(drgnn_f4*)(dst + (m.r[j] - base) * ld + 4 * fastmod(fd, q, row)) = b.v[j];
        }
    }
}

// SPIN: the run-time layout's offsets pinned in SCALAR registers.  sGAT's training kernels with the run-time layout sat at the
// 128-VGPR limit of a 16-wave workgroup with 31 - 39 registers spilled to scratch memory; with the ~36 offsets in SGPRs (some of
// which the compiler parks in lanes of a VGPR: no memory) they need 94 - 97 and spill nothing: batch 64, 32 features 21.05 ->
// 19.70 us per step rebuilt, 20.65 -> 19.32 cached, 48 features 23.0 -> 21.07 (profiles/r05_sgat_spin_ab.txt).  NOT for the
// one-workgroup launches that carry the one-role weighted builder (family 5 of drgnn_step_af.h): the scalar registers are what
// that builder chain lives on, 24.0 -> 24.75 us at batch 128; and not for the one-workgroup launches on a cached workspace
// either (31 - 35 spilled, yet 19.4 -> 19.9 us at batch 128 with the scalar pin): the two-workgroup launches only.
#ifdef DRGNN_EMU
#define STEP2_PIN(x) ((void)0)
#else
#define STEP2_PIN(x) do { if (SPIN) { asm volatile("" : "+s"(x)); } else { STEP_PIN(x); } } while (0)
#endif
template <int CLS, bool SPIN = false, int XG = 0>
DEV Step2Scratch step2_carve(float* base, int kind, int F, int capN, int capE, int capC, int H, int O) {
    const int sg = (kind == DRGNN_SGAT) ? 1 : 0;
    constexpr int xg = XG;
    const int xld = step_pad16(F) + 4;
    Step2Scratch s;
    int o = 0;
    // run-time capacities: every offset pinned in a register once (drgnn_step.h, step_carve); capacity class: immediates
#define X(name, words, cond)                                                          \
    { int off = o; if (CLS == 0) { STEP2_PIN(off); } s.name = (decltype(s.name))(base + off);         \
      o = off + ((cond) ? (int)(((long)(words) + 3) & ~3L) : 0); }
    STEP2_CARVE_LIST(X)
#undef X
    s.end = base + o;
    s.gp = s.wb;      // fc1's weights are dead after d readout: the K-split products keep their partial tiles there
    s.u2 = s.z1;      // (see STEP2_Z1U_WORDS)
    s.z2 = s.z1 + (long)(capC + 4) * STEP2_TSLD;
    return s;
}

// ---- phase B: Z1 = relu(D . (S Wn) + C . (X Ws) + b) over the own rows (S, X rows at their local positions) ---------------
// XG: the x rows come from memory (xs = the graph's rows in NODE order, `xtf` floats apart; hord: position -> node)
template <int KIND, int XF, int XG = 0>
DEV void step2_conv1(int n, int nmax, const float* G, const float* xs, const float* w1t, const float* ws1t, const float* b1,
                     const float* dv, const float* sc, float* z1, int* dummy, const int* hord = nullptr, int xtf = 0) {
    constexpr int XLD = XF + 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int lr = lane & 15, lq = lane >> 4;
    const int units = (n + 15) >> 4;
    for (int ti = wave; ti < units; ti += DRGNN_NWAVES) {
        const int prow = ti * 16 + lr;
        const int row = prow < nmax ? prow : nmax - 1;      // rows past the own range: any valid row (results discarded)
        // (XG == 2: G = the graph's S rows in memory, laid out like xs)
        const float* ag = (XG == 2) ? G + (long)hord[row] * xtf + 4 * lq : G + row * XLD + 4 * lq;
        const float* ax = XG ? xs + (long)hord[row] * xtf + 4 * lq : xs + row * XLD + 4 * lq;
        drgnn_f4 xv[XF / 16], gv[XF / 16];
        if (XG) {      // (all of the row's chunks requested at once; chunks past the row's end -- padded widths -- are zero)
#pragma unroll
            for (int k0 = 0; k0 < XF; k0 += 16)
                xv[k0 / 16] = (k0 + 4 * lq < xtf) ? *(const drgnn_f4*)(ax + k0) : drgnn_f4{0.f, 0.f, 0.f, 0.f};
        }
        if (XG == 2) {
#pragma unroll
            for (int k0 = 0; k0 < XF; k0 += 16)
                gv[k0 / 16] = (k0 + 4 * lq < xtf) ? *(const drgnn_f4*)(ag + k0) : drgnn_f4{0.f, 0.f, 0.f, 0.f};
        }
        const float* bn = w1t + lr * XLD + 4 * lq;
        const float* bs = ws1t + lr * XLD + 4 * lq;
        drgnn_f32x4 accn = {0.f, 0.f, 0.f, 0.f}, accs = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k0 = 0; k0 < XF; k0 += 16) {
            const drgnn_f4 a = (XG == 2) ? gv[k0 / 16] : *(const drgnn_f4*)(ag + k0), x = XG ? xv[k0 / 16] : *(const drgnn_f4*)(ax + k0);
            const drgnn_f4 wn = *(const drgnn_f4*)(bn + k0), ws = *(const drgnn_f4*)(bs + k0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                accn = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], wn[j], accn, 0, 0, 0);
                accs = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j], ws[j], accs, 0, 0, 0);
            }
        }
        const float bias = b1[lr];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ci = ti * 16 + lq * 4 + r;
            const bool ok = ci < n;
            const float s = ok ? sc[ci] : 0.0f, d = ok ? dv[ci] : 0.0f;
            float v = fmaf(s, accs[r], d * accn[r]) + bias;
            if (KIND == DRGNN_FOUT && ok && d == 0.0f) v = DRGNN_NAN;      // mean over an empty neighbourhood
            v = (v < 0.0f) ? 0.0f : v;                                      // relu that lets NaN through
            float* p = ok ? z1 + ci * DRGNN_H1 + lr : (float*)dummy + lane;
            *p = v;
        }
    }
}

// ---- phase C: depth-0 cluster max over CONTIGUOUS rows (+ argmax = own row, -1 where no gradient flows), published -------
DEV void step2_cluster_max(int nc, const int* hmp, int qbase, int nbase, const int* cid, const float* z, float* xp, short* a0,
                           unsigned long long* pub, uint32_t tag) {
    FOR_TID(item, nc * DRGNN_H1) {
        const int r = item >> 4, c = item & 15;
        const int plo = hmp[qbase + r] - nbase, phi = hmp[qbase + r + 1] - nbase;
        float best = DRGNN_NEG_INF;
        int arg = -1;
        for (int p = plo; p < phi; p += 4) {
            int mm[4];
            float vv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) mm[j] = (p + j < phi) ? p + j : phi - 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) vv[j] = z[mm[j] * DRGNN_H1 + c];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (vv[j] > best) { best = vv[j]; arg = mm[j]; }
        }
        if (arg < 0) best = 0.0f;
        if (pub) xchg_publish(pub + item, tag, best);      // first: the store that has the farthest to go
        xp[ROW24(cid[qbase + r], STEP_XPLD) + c] = best;
        a0[item] = (short)((best > 0.0f) ? arg : -1);
    }
}

// bounded wait for a partner's word (as xchg_wait, fault bit DRGNN_FAULT_SPLIT)
DEV float step2_wait(unsigned long long* slot, unsigned long long w, uint32_t tag, int32_t* fault) {
    if ((uint32_t)(w >> 32) == tag) return __uint_as_float((uint32_t)w);
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        w = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(w >> 32) == tag) return __uint_as_float((uint32_t)w);
        if (wall_clock64() - t0 > 30000000ull) { atomicOr(fault, DRGNN_FAULT_SPLIT); return DRGNN_NAN; }
        __builtin_amdgcn_s_sleep(1);
    }
}
// the partner's rows of a [clusters x 16] hand-off -> dst[cid[r]] (rows of STEP_XPLD floats); `w0`: this lane's first poll,
// requested before the barrier (item = threadIdx.x)
DEV void step2_receive(int nc, const int* cid, int qbase, unsigned long long* slots, unsigned long long w0, uint32_t tag,
                       int32_t* fault, float* dst) {
    for (int item = threadIdx.x; item < nc * DRGNN_H1; item += DRGNN_NTHREADS) {
        const int r = item >> 4, c = item & 15;
        const unsigned long long w = (item == (int)threadIdx.x) ? w0 : xchg_peek(slots + item);
        dst[ROW24(cid[qbase + r], STEP_XPLD) + c] = step2_wait(slots + item, w, tag, fault);
    }
}

// ---- phase E: [S | T] of the own pooled rows (row q <-> pooled node cid[q]); 16 lanes per row as step_pooled_gather -----
template <int KIND, class IdxT>
DEV void step2_pooled_gather(int n, const int* cid, int qbase, const int* rp, const IdxT* col, const float* w, float* dv,
                             float* sc, const float* xp, float* ts) {
    constexpr int LDX = STEP_XPLD, LDT = STEP2_TSLD;
    const int items = ((n * 16) + 63) & ~63;
    for (int item = threadIdx.x; item < items; item += DRGNN_NTHREADS) {
        const int q = item >> 4, sl = (item >> 2) & 3, c = (item & 3) * 4;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, asum = 0.f;
        int lo = 0, hi = 0, i = 0;
        if (q < n) {
            i = cid[qbase + q];
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
        if (sl == 0 && q < n) {
            const int deg = hi - lo;
            float d, sv;
            if (KIND == DRGNN_SGAT) { d = 1.0f / (float)(deg > 0 ? deg : 1); sv = asum * d; }
            else { d = deg > 0 ? 1.0f / (float)deg : 0.0f; sv = 1.0f; }
            if (c == 0) { dv[q] = d; sc[q] = sv; }
            const drgnn_f4 x = *(const drgnn_f4*)(xp + ROW24(i, LDX) + c);
            *(drgnn_f4*)(ts + q * LDT + c) = drgnn_f4{a0 * d, a1 * d, a2 * d, a3 * d};
            *(drgnn_f4*)(ts + q * LDT + DRGNN_H1 + c) = drgnn_f4{sv * x[0], sv * x[1], sv * x[2], sv * x[3]};
        }
    }
}

// ---- phase G: depth-1 max over contiguous rows of Z2 (+ argmax) and the PARTIAL readout sum over the own clusters --------
DEV void step2_pool_readout(int nk, const int* mp, int kbase, int qbase, const float* z, short* a1, float* xr_part,
                            unsigned long long* pub, uint32_t tag) {
    constexpr int LDZ = STEP2_TSLD;
    for (int t = threadIdx.x; t < DRGNN_H2 * 16; t += DRGNN_NTHREADS) {      // 512 lanes: whole waves
        const int c = t >> 4, kk = t & 15;
        float acc = 0.0f;
        for (int k = kk; k < nk; k += 16) {
            float best = DRGNN_NEG_INF;
            int am = -1;
            const int plo = mp[kbase + k] - qbase, phi = mp[kbase + k + 1] - qbase;
            for (int p = plo; p < phi; p += 4) {
                int mm[4];
                float vv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) mm[j] = (p + j < phi) ? p + j : phi - 1;
#pragma unroll
                for (int j = 0; j < 4; ++j) vv[j] = z[ROW24(mm[j], LDZ) + c];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (vv[j] > best) { best = vv[j]; am = mm[j]; }
            }
            if (am < 0) best = 0.0f;
            a1[k * DRGNN_H2 + c] = (short)((best > 0.0f) ? am : -1);
            acc += best;
        }
        acc = lanes16_sum(acc);
        if (kk == 0) {
            if (pub) xchg_publish(pub + c, tag, acc);
            xr_part[c] = acc;
        }
    }
}

// ---- phase H: readout = (partial of half 0 + partial of half 1) / C1, fc1, relu, dropout -> hid (both halves alike) -----
// fc1 is linear: each half forms W1 (its own partial readout) while the partner's 32 partial sums travel (requested first,
// consumed after the own product), then W1 (the partner's partial); the two are added as half 0 + half 1 in BOTH workgroups.
template <int HC, int SPLIT>
DEV void step2_head_fc1(const HeadFused& hf, int g, int half, const float* wb, const float* b1, float* xr, float* hid,
                        unsigned long long* ro_other, uint32_t tag, uint32_t step, uint32_t thresh, float keep_scale,
                        float inv, int bad, int32_t* fault, float* g_readout) {
    static_assert(HC > 0 && HC * 8 <= DRGNN_NTHREADS, "one pass: 8 lanes per hidden unit");
    constexpr int H = HC;
    if ((int)(threadIdx.x & ~63u) >= H * 8) return;      // waves without hidden units
    const int lane = threadIdx.x & 63;
    const int t = threadIdx.x, h = t >> 3, q = t & 7;
    unsigned long long w0 = 0ull;
    if (SPLIT == 2 && lane < DRGNN_H2) w0 = xchg_peek(ro_other + lane);
    const float own = (lane < DRGNN_H2) ? xr[DRGNN_H2 + lane] : 0.0f;
    const drgnn_f4 w = *(const drgnn_f4*)(wb + h * STEP_WBLD + 4 * q);
    float acc = fmaf(w[0], __shfl(own, 4 * q, 64), fmaf(w[1], __shfl(own, 4 * q + 1, 64),
                fmaf(w[2], __shfl(own, 4 * q + 2, 64), w[3] * __shfl(own, 4 * q + 3, 64))));
    acc = lanes8_sum(acc);
    float tot = own;
    if (SPLIT == 2) {
        float pv = 0.0f;
        if (lane < DRGNN_H2) pv = step2_wait(ro_other + lane, w0, tag, fault);
        float acco = fmaf(w[0], __shfl(pv, 4 * q, 64), fmaf(w[1], __shfl(pv, 4 * q + 1, 64),
                     fmaf(w[2], __shfl(pv, 4 * q + 2, 64), w[3] * __shfl(pv, 4 * q + 3, 64))));
        acco = lanes8_sum(acco);
        acc = (half == 0) ? acc + acco : acco + acc;
        tot = (half == 0) ? own + pv : pv + own;
    }
    if (threadIdx.x < DRGNN_H2) {
        float r = tot * inv;
        if (bad) r = DRGNN_NAN;
        xr[lane] = r;
        if (half == 0) g_readout[lane] = r;
    }
    if (q == 0) {
        float v = fmaf(acc, inv, b1[h]);
        if (bad) v = DRGNN_NAN;
        v = v > 0.0f ? v : 0.0f;
        if (thresh) v = drgnn_keep(hf, step, g, H, h, thresh) ? v * keep_scale : 0.0f;
        hid[h] = v;
    }
}

// ---- phase J: d readout = dhid W1, scattered through the depth-1 argmax of the OWN clusters into dZ2 (factor 1 / C1), and
// the weight / bias gradients of conv2, which are SPARSE sums (dZ2 is non-zero only in the rows that won a depth-1 cluster,
// every entry of column c equal to v_c):   d[Wnbr ; Wself][f][c] = v_c sum_k [S | T][a1[k][c]][f],   db2[c] = v_c #winners.
// 32 lanes per column: 8 float4 groups of the 32 [S | T] columns x 4 slices of the clusters (drgnn_step3.h has GINet's)
template <int HC>
DEV void step2_head_dreadout(const HeadFused& hf, const float* wb, const float* dhid, const short* a1, int nk, float inv,
                             const float* st, float* z2, float* g_dw2, float* g_db2) {
    const int H = HC ? HC : hf.H;
    for (int t = threadIdx.x; t < DRGNN_H2 * 32; t += DRGNN_NTHREADS) {
        const int c = t >> 5, q = t & 31;
        float acc = 0.0f;
        for (int h = q; h < H; h += 32) acc = fmaf(dhid[h], wb[h * STEP_WBLD + c], acc);
        const float v = lanes32_sum(acc) * inv;
        for (int k = q; k < nk; k += 32) {
            const int r = a1[k * DRGNN_H2 + c];
            if (r >= 0) z2[ROW24(r, STEP2_TSLD) + c] = v;
        }
        const int sl = q & 3, f4 = q >> 2;
        drgnn_f4 sum = {0.f, 0.f, 0.f, 0.f};
        float cnt = 0.0f;
        for (int k = sl; k < nk; k += 4) {
            const int r = a1[k * DRGNN_H2 + c];
            if (r >= 0) {
                const drgnn_f4 row = *(const drgnn_f4*)(st + ROW24(r, STEP2_TSLD) + 4 * f4);
                sum[0] += row[0]; sum[1] += row[1]; sum[2] += row[2]; sum[3] += row[3];
                cnt += 1.0f;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { sum[i] += dpp_take<0xB1>(sum[i]); sum[i] += dpp_take<0x4E>(sum[i]); }
        cnt += dpp_take<0xB1>(cnt); cnt += dpp_take<0x4E>(cnt);
        if (sl == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) g_dw2[(4 * f4 + i) * DRGNN_H2 + c] = v * sum[i];
            if (f4 == 0) g_db2[c] = v * cnt;
        }
    }
}

// ---- phase K: d[S | T] = dZ2 [Wnbr ; Wself]^T over the own rows.  dS goes, pre-multiplied by the row's d_i, to the full
// dS array (row = pooled node id) AND to the partner; dT stays local (rows = own positions) ------------------------------
DEV void step2_gemm_dst(int M, const float* dz, const float* wn, const int* cid, int qbase, const float* dv, float* dsf,
                        float* dt, unsigned long long* pub, uint32_t tag, int* dummy) {
    constexpr int LDA = STEP2_TSLD, LDB = STEP2_TSLD;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int lr = lane & 15, lq = lane >> 4;
    const int units = ((M + 15) >> 4) * 2;
    for (int u = wave; u < units; u += DRGNN_NWAVES) {
        const int ti = u >> 1, tj = u & 1;
        const float* ap = dz + (ti * 16 + lr) * LDA + 4 * lq;
        const float* bp = wn + (tj * 16 + lr) * LDB + 4 * lq;
        const drgnn_f4 a0 = *(const drgnn_f4*)ap, b0 = *(const drgnn_f4*)bp;
        const drgnn_f4 a1 = *(const drgnn_f4*)(ap + 16), b1 = *(const drgnn_f4*)(bp + 16);
        drgnn_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], b0[j], acc, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], b1[j], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ci = ti * 16 + lq * 4 + r;
            if (ci < M) {
                if (tj == 0) {
                    const float v = acc[r] * dv[ci];
                    if (pub) xchg_publish(pub + ci * DRGNN_H1 + lr, tag, v);
                    dsf[ROW24(cid[qbase + ci], STEP_XPLD) + lr] = v;
                } else {
                    dt[ci * STEP_XPLD + lr] = acc[r];
                }
            }
        }
    }
    (void)dummy;
}

// ---- phase M: d xp_j = s_j dT_j + sum over CSC1 entries t of column j of c_t (d dS)_row(t), own pooled rows: dense rows
// dxp[q] (the weight gradients of conv1 read them through the depth-0 argmax, step2_dw1_sparse)
template <int KIND, class IdxT>
DEV void step2_pooled_gather_bwd(int n, const int* cid, int qbase, const int* cp, const IdxT* ridx, const IdxT* tslot,
                                 const float* w, const float* dv, const float* sc, const float* dsf, const float* dt,
                                 float* dxp) {
    const int items = ((n * 16) + 63) & ~63;
    for (int item = threadIdx.x; item < items; item += DRGNN_NTHREADS) {
        const int q = item >> 4, sl = (item >> 2) & 3, c = (item & 3) * 4;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (q < n) {
            const int j = cid[qbase + q];
            const int lo = cp[j], hi = cp[j + 1];
            for (int t = lo + sl; t < hi; t += 4) {
                const drgnn_f4 v = *(const drgnn_f4*)(dsf + ROW24(ridx[t], STEP_XPLD) + c);
                float cf = 1.0f;
                if (KIND == DRGNN_SGAT) cf = w[tslot[t]];
                a0 = fmaf(cf, v[0], a0); a1 = fmaf(cf, v[1], a1); a2 = fmaf(cf, v[2], a2); a3 = fmaf(cf, v[3], a3);
            }
        }
        a0 += dpp_take<0x128>(a0); a1 += dpp_take<0x128>(a1); a2 += dpp_take<0x128>(a2); a3 += dpp_take<0x128>(a3);
        a0 += dpp_take<0x124>(a0); a1 += dpp_take<0x124>(a1); a2 += dpp_take<0x124>(a2); a3 += dpp_take<0x124>(a3);
        if (sl == 0 && q < n) {
            float sv = sc[q];
            if (KIND == DRGNN_FOUT && dv[q] == 0.0f) sv = 0.0f;      // its NaN row never won a max
            const drgnn_f4 d4 = *(const drgnn_f4*)(dt + q * STEP_XPLD + c);
            *(drgnn_f4*)(dxp + q * STEP_XPLD + c) = drgnn_f4{fmaf(sv, d4[0], a0), fmaf(sv, d4[1], a1), fmaf(sv, d4[2], a2), fmaf(sv, d4[3], a3)};
        }
    }
}
// ---- phase N: the weight / bias gradients of conv1 through the depth-0 argmax (dZ1 is non-zero only in the rows p = a0[q][h]
// that won a depth-0 cluster):  dWn[f][h] = sum_q D_p dxp[q][h] S[p][f],  dWs[f][h] = sum_q C_p dxp[q][h] X[p][f],
// db1[h] = sum_q dxp[q][h].  Wave = channel h; in a wave NCH feature chunks (float4) x NSL slices of the own pooled rows
// (consecutive lanes: the slice sums meet in DPP adds).  16-wide: 4 chunks x 16 slices, 32-wide: 8 x 8, 64-wide: 16 x 4;
// 48-wide: the sixteen chunk slots of the 64-wide form, twelve in use
template <int XF> struct Dw1Shape {
    static constexpr int NCH = (XF == 16) ? 4 : (XF == 32) ? 8 : 16;
    static constexpr int NSL = 64 / NCH;
};
template <int NSL> DEV float step_slices_sum(float v) {
    static_assert(NSL == 4 || NSL == 8 || NSL == 16, "aligned groups of 4 / 8 / 16 consecutive lanes");
    v += dpp_take<0xB1>(v);     // quad_perm [1,0,3,2]
    v += dpp_take<0x4E>(v);     // quad_perm [2,3,0,1]
    if (NSL >= 8) v += dpp_take<0x141>(v);    // row_half_mirror
    if (NSL >= 16) v += dpp_take<0x140>(v);   // row_mirror
    return v;
}
template <int XF, int XG = 0>
DEV void step2_dw1_sparse(int Ch, const short* a0, const float* dxp, const float* G, const float* xs, const float* dv,
                          const float* sc, float* g_dwn, float* g_dws, float* g_db1, int F, const int* hord = nullptr, int xtf = 0) {
    constexpr int XLD = XF + 4, NSL = Dw1Shape<XF>::NSL;
    const int h = threadIdx.x >> 6, fc = (threadIdx.x & 63) / NSL, sl = threadIdx.x & (NSL - 1);
    const bool live = 4 * fc < XF;
    drgnn_f4 an = {0.f, 0.f, 0.f, 0.f}, as = {0.f, 0.f, 0.f, 0.f};
    float bsum = 0.0f;
    for (int q = sl; q < Ch; q += 2 * NSL) {      // two pooled rows per trip in flight
        int arg[2];
        float d[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int qq = q + NSL * u;
            arg[u] = (qq < Ch && live) ? (int)a0[qq * DRGNN_H1 + h] : -1;
            d[u] = (qq < Ch) ? dxp[qq * STEP_XPLD + h] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (arg[u] >= 0) {
                const float dn = d[u] * dv[arg[u]], ds = d[u] * sc[arg[u]];
                const drgnn_f4 g = (XG < 2) ? *(const drgnn_f4*)(G + ROW24(arg[u], XLD) + 4 * fc)
                                   : (4 * fc < xtf) ? *(const drgnn_f4*)(G + (long)hord[arg[u]] * xtf + 4 * fc) : drgnn_f4{0.f, 0.f, 0.f, 0.f};
                const drgnn_f4 x = !XG ? *(const drgnn_f4*)(xs + ROW24(arg[u], XLD) + 4 * fc)
                                   : (4 * fc < xtf) ? *(const drgnn_f4*)(xs + (long)hord[arg[u]] * xtf + 4 * fc) : drgnn_f4{0.f, 0.f, 0.f, 0.f};
                an[0] = fmaf(dn, g[0], an[0]); an[1] = fmaf(dn, g[1], an[1]); an[2] = fmaf(dn, g[2], an[2]); an[3] = fmaf(dn, g[3], an[3]);
                as[0] = fmaf(ds, x[0], as[0]); as[1] = fmaf(ds, x[1], as[1]); as[2] = fmaf(ds, x[2], as[2]); as[3] = fmaf(ds, x[3], as[3]);
                bsum += d[u];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { an[i] = step_slices_sum<NSL>(an[i]); as[i] = step_slices_sum<NSL>(as[i]); }
    bsum = step_slices_sum<NSL>(bsum);
    if (sl == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (4 * fc + i < F) { g_dwn[(4 * fc + i) * DRGNN_H1 + h] = an[i]; g_dws[(4 * fc + i) * DRGNN_H1 + h] = as[i]; }
        if (fc == 0) g_db1[h] = bsum;
    }
}


// =========================================================================================================================
// XF: padded feature width (16 / 32 / 48 / 64; the host has checked step_burst_guaranteed: register-burst prologue, reference
// head width); CLS as in drgnn_step.h; SPLIT: workgroups per graph; half: which one.  `late` as in net_step_graph: sizes and
// offsets came with the launch arguments, the device-computed counts (clusters, pooled edges, split point) are in flight.
// TRAIN = false: the inference launch (forward + head, predictions only; one workgroup per graph).
// XG: x rows read from memory instead of staged in LDS -- what lets 200-node graphs with up to 64 features into the 160 KiB
// (133 KB instead of 189 KB at SYN size) and, at the narrower widths, graphs of 260 - 350 nodes; the host takes it only when
// the staged form does not fit.  XG = 2 (32-, 48- and 64-wide): the S rows of the tiles stay in memory as well.
template <int KIND, int XF, bool GATHER, int CLS, int SPLIT, bool TRAIN = true, int XG = 0>
DEV void net_step2_graph(const StepArgs& a, const GraphDims& d_in, int g, int gi, int half, float* scratch, int capN, int capE,
                         int capC, bool late, int cnt_c, int cnt_e1, int cnt_c1, int hs_k, int hs_q, int hs_n) {
    static_assert(XF == 16 || XF == 32 || XF == 48 || XF == 64, "width-specialised kernels only");
    static_assert(KIND != DRGNN_GINET, "single-branch nets");
    static_assert(TRAIN || SPLIT == 1, "inference launches run one workgroup per graph");
    if (CLS == 1) { capN = STEP_CLS_N; capE = STEP_CLS_E; capC = STEP_CLS_C; }
    GraphDims d = d_in;
    const int bC = late ? imin(d.N, capC) : d.C, bE1 = late ? d.E : d.E1, bC1 = late ? imin(d.N, capC) : d.C1;
    const TopoView& tv = a.tv;
    const HeadFused& hf = a.hf;
    constexpr int R = DRGNN_H2, WREF = 64;
    constexpr int XLD = XF + 4;
    constexpr bool NARROW = (KIND == DRGNN_SGAT);
    typedef typename StepIdx<NARROW>::type EIdx;
    const int F = a.net.n_feat;
    const int O = hf.O;
    constexpr bool SPIN = KIND == DRGNN_SGAT && TRAIN && SPLIT == 2;      // (see step2_carve)
    static_assert(!XG || CLS == 0, "the x-from-memory form has the run-time layout");
    Step2Scratch s = step2_carve<CLS, SPIN, XG>(scratch, KIND, XF, capN, capE, capC, WREF, O);
    EXIT_AFTER(0);
    WBlockRegs<1> wreg;
    int* const dummy = (int*)(s.misc + 64);
    const uint32_t done = (uint32_t)a.step2[0];
    const uint32_t tag = done + 1u;
    const drgnn_conv_params& c1 = a.net.conv1[0];
    const drgnn_conv_params& c2 = a.net.conv2[0];
    // exchange areas of this graph
    unsigned long long* const xg = a.xchg + (long)g * a.xchg_stride;
    const int xcap = (a.xchg_stride - 2 * DRGNN_H2) / (4 * DRGNN_H1);
    unsigned long long* const x_xp_own = xg + (long)half * xcap * DRGNN_H1;
    unsigned long long* const x_xp_oth = xg + (long)(1 - half) * xcap * DRGNN_H1;
    unsigned long long* const x_ds_own = xg + (long)(2 + half) * xcap * DRGNN_H1;
    unsigned long long* const x_ds_oth = xg + (long)(3 - half) * xcap * DRGNN_H1;
    unsigned long long* const x_ro_own = xg + (long)4 * xcap * DRGNN_H1 + half * DRGNN_H2;
    unsigned long long* const x_ro_oth = xg + (long)4 * xcap * DRGNN_H1 + (1 - half) * DRGNN_H2;
    int32_t* const fault = a.step2 + 2;

    // ---- prologue: one burst of independent loads ----------------------------------------------------------------------
    PHASE_MARK();
    // float4 per lane of a row tile (capacity class: STEP_CLS_N rows of XF floats over 1024 lanes -- 2 at 32, 3 at 48 features)
    constexpr int BJ = (CLS == 1) ? (STEP_CLS_N * (XF / 4) + DRGNN_BCAP - 1) / DRGNN_BCAP : 4;
    static_assert(BJ >= 1 && BJ <= 4, "a class row tile is one burst");
    // the graph's x rows and aggregation tiles (node order).  Rows of the tiles are TF = pad4(F) floats long (zero padded by the
    // builder); with F % 4 != 0 the x rows come from the tiles' padded copy instead of the (unaligned) input
    const int TF = (F + 3) & ~3;
    const float* xgl = (F & 3) ? a.tiles + drgnn_tiles_x_off(a.tile_nodes, TF) + (long)d.n0 * TF : a.x + (long)d.n0 * F;
    const float* sgl = a.tiles + (long)d.n0 * TF;
    const float* tdg = a.tiles + a.tile_nodes * TF + d.n0;
    const float* tcg = tdg + a.tile_nodes;
    BurstX<BJ> bx, bsum;
    BurstRowMap<BJ> brow;
    BurstW<1> bw1, bs1, bw2, bs2;
    WaveStage wst, wst2;
    const int my_wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    auto stage_job = [&](int burst, int w) -> StageJob {
        const int32_t* const* P = tv.p;
        const int nar = NARROW ? 1 : 0;
        StageJob j = {nullptr, 0, nullptr, 0};
        switch (burst * 16 + w) {
        case 16 + 0: j = StageJob{P[DRGNN_TI_HMP0] + d.rowbase, bC + 1, s.hmp, 0}; break;
        case 16 + 1: j = StageJob{P[DRGNN_TI_MEM1] + d.n0, bC, s.cid, 0}; break;
        case 16 + 2: j = StageJob{P[DRGNN_TI_MPTR1] + d.rowbase, bC1 + 1, s.mp1, 0}; break;
        case 16 + 3: j = StageJob{c1.bias, DRGNN_H1, s.b1, 0}; break;
        case 16 + 4: j = StageJob{P[DRGNN_TI_ROWPTR1] + d.rowbase, bC + 1, s.rp1, 0}; break;
        case 16 + 5: j = stage_half(StageJob{P[DRGNN_TI_COL1] + d.e0, bE1, s.cx1, nar}, 0); break;
        case 16 + 6: j = stage_half(StageJob{P[DRGNN_TI_COL1] + d.e0, bE1, s.cx1, nar}, 1); break;
        case 16 + 7: j = StageJob{hf.b1, WREF, s.hb1, 0}; break;
        case 16 + 8: j = StageJob{hf.w2, O * WREF, s.hw2, 0}; break;
        case 16 + 9: j = StageJob{hf.b2, O, s.hb2, 0}; break;
        case 16 + 10: j = StageJob{c2.bias, DRGNN_H2, s.b2, 0}; break;
        case 16 + 11: if (TRAIN) j = StageJob{P[DRGNN_TI_COLPTR1] + d.rowbase, bC + 1, s.cp1, 0}; break;
        case 16 + 12: if (TRAIN) j = stage_half(StageJob{P[DRGNN_TI_ROWIDX1] + d.e0, bE1, s.rx1, nar}, 0); break;
        case 16 + 13: if (TRAIN) j = stage_half(StageJob{P[DRGNN_TI_ROWIDX1] + d.e0, bE1, s.rx1, nar}, 1); break;
        case 16 + 14: if (XG) j = StageJob{P[DRGNN_TI_HORD] + d.n0, d.N, s.hord, 0}; break;

        case 32 + 0: if (KIND == DRGNN_SGAT) j = stage_half(StageJob{tv.w1 + d.e0, bE1, s.ew1, 0}, 0); break;
        case 32 + 1: if (KIND == DRGNN_SGAT) j = stage_half(StageJob{tv.w1 + d.e0, bE1, s.ew1, 0}, 1); break;
        case 32 + 2: if (KIND == DRGNN_SGAT && TRAIN) j = stage_half(StageJob{P[DRGNN_TI_TSLOT1] + d.e0, bE1, s.ts1, nar}, 0); break;
        case 32 + 3: if (KIND == DRGNN_SGAT && TRAIN) j = stage_half(StageJob{P[DRGNN_TI_TSLOT1] + d.e0, bE1, s.ts1, nar}, 1); break;
        default: break;
        }
        return j;
    };
    {   // every workspace pointer in one batch of scalar loads
        const int32_t* const* P = tv.p;
        asm volatile("" :: "s"(P[DRGNN_TI_IHORD]), "s"(P[DRGNN_TI_HMP0]), "s"(P[DRGNN_TI_MEM1]), "s"(P[DRGNN_TI_MPTR1]));
        asm volatile("" :: "s"(P[DRGNN_TI_ROWPTR1]), "s"(P[DRGNN_TI_COL1]), "s"(P[DRGNN_TI_COLPTR1]), "s"(P[DRGNN_TI_ROWIDX1]));
    }
    int m_bad = 0, m_y = 0;
    float m_wy = 1.0f, m_denom = 1.0f;
    if (my_wave == 0) {
        m_bad = tv.p[DRGNN_TI_ERR][0] | tv.p[DRGNN_TI_GSTAT][gi] | tv.p[DRGNN_TI_GSTAT][(GATHER ? a.ws_graphs : a.n_graphs) + gi];
        if (!TRAIN) {
        } else if (__builtin_expect(hf.task != DRGNN_TASK_CLASS, 1)) {      // (DRGNN_TASK_GRAD: d loss / d pred_g, O == 1; gi == g there)
            m_y = __builtin_nontemporal_load((const int*)hf.y_reg + gi);
        } else {
            m_y = (int)hf.y_cls[gi];
            m_wy = hf.class_w ? hf.class_w[m_y] : 1.0f;
            m_denom = (float)hf.B;
            if (hf.class_w && threadIdx.x < 64) {
                float part_sum = 0.0f;
                for (int q = threadIdx.x; q < hf.B; q += 64) part_sum += hf.class_w[hf.y_cls[GATHER ? a.gather_ids[q] : q]];
                m_denom = lanes64_sum(part_sum);
            }
            m_y = __builtin_amdgcn_readfirstlane(m_y);
            m_wy = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(m_wy)));
            m_bad = __builtin_amdgcn_readfirstlane(m_bad);
        }
    }
    if (XG < 2) burst_load_x(bsum, sgl, d.N, TF);
    if (!XG) burst_load_x(bx, xgl, d.N, TF);
    if (XG < 2) burst_load_rowmap(brow, bsum, tv.p[DRGNN_TI_IHORD] + d.n0, d.N);      // (the S and the x tile have the same geometry)
    // per-node coefficients D, C and the node's position (one node per lane: d.N <= threads, step_burst_guaranteed)
    float n_d = 0.0f, n_c = 0.0f;
    int n_pos = -1;
    if ((int)threadIdx.x < d.N) { n_d = tdg[threadIdx.x]; n_c = tcg[threadIdx.x]; n_pos = tv.p[DRGNN_TI_IHORD][d.n0 + threadIdx.x]; }
    burst_load_w(bw1, c1.w_nbr, c1.nbr_sk, c1.nbr_sh, F, DRGNN_H1);
    burst_load_w(bs1, c1.w_self, c1.self_sk, c1.self_sh, F, DRGNN_H1);
    wstage_load(wst, stage_job(1, my_wave));
    burst_load_w(bw2, c2.w_nbr, c2.nbr_sk, c2.nbr_sh, DRGNN_H1, DRGNN_H2);
    burst_load_w(bs2, c2.w_self, c2.self_sk, c2.self_sh, DRGNN_H1, DRGNN_H2);
    step_wblock_load(wreg, hf, 0);
    wstage_load(wst2, stage_job(2, my_wave));
    // the device-computed counts and the split point have been in flight since the kernel's first instructions
    if (late) {
        d.C = WG_UNIFORM(cnt_c); d.E1 = WG_UNIFORM(cnt_e1); d.C1 = WG_UNIFORM(cnt_c1);
        hs_k = WG_UNIFORM(hs_k); hs_q = WG_UNIFORM(hs_q); hs_n = WG_UNIFORM(hs_n);
    }
    int bad_shape = 0;
    if (d.C > capC || d.E1 > d.E || d.C1 > capC || hs_k < 0 || hs_k > d.C1 || hs_q < 0 || hs_q > d.C || hs_n < 0 || hs_n > d.N) {
        // malformed input (flagged by the builder): stay inside LDS, poison
        d.C = imin(d.C, capC); d.E1 = imin(d.E1, d.E); d.C1 = imin(d.C1, capC);
        hs_k = imin(imax(hs_k, 0), d.C1); hs_q = imin(imax(hs_q, 0), d.C); hs_n = imin(imax(hs_n, 0), d.N);
        bad_shape = 1;
    }
    // ---- ownership: half 0 = the depth-1 clusters [0, k), i.e. pooled rows [0, q) and node positions [0, n) ----------------
    const int kbase = (SPLIT == 2 && half == 1) ? hs_k : 0;
    const int qbase = (SPLIT == 2 && half == 1) ? hs_q : 0;
    const int nbase = (SPLIT == 2 && half == 1) ? hs_n : 0;
    const int Kh = (SPLIT == 2) ? (half == 0 ? hs_k : d.C1 - hs_k) : d.C1;
    const int Ch = (SPLIT == 2) ? (half == 0 ? hs_q : d.C - hs_q) : d.C;
    const int Nh = (SPLIT == 2) ? (half == 0 ? hs_n : d.N - hs_n) : d.N;
    const int Co = d.C - Ch, qbase_o = (SPLIT == 2 && half == 0) ? hs_q : 0;      // the partner's pooled rows
    const int nmax = imax(Nh, 1);
    // the S and x rows of the OWN positions -> LDS row (position - nbase); D, C likewise
    if (XG < 2) burst_store_x4_rows(bsum, brow, s.G, XLD, nbase, Nh);
    if (!XG) burst_store_x4_rows(bx, brow, s.xs, XLD, nbase, Nh);
    if ((unsigned)(n_pos - nbase) < (unsigned)Nh) { s.dv0[n_pos - nbase] = n_d; s.sc0[n_pos - nbase] = n_c; }
    FOR_TID(e, step_pad4(Nh) - Nh) { s.dv0[Nh + e] = 0.0f; s.sc0[Nh + e] = 0.0f; }      // (coefficients of the K padding rows)
    burst_store_wt(bw1, s.w1t, XLD);
    burst_store_wt(bs1, s.ws1t, XLD);
    wstage_store(wst);
    if (XF > TF) {     // zero padding of the k columns [TF, XF) of the row tiles ...
        const int padg = XF - TF;
        FOR_TID(e, Nh * padg) {
            if (!XG) { s.xs[(e / padg) * XLD + TF + e % padg] = 0.0f; }
            if (XG < 2) { s.G[(e / padg) * XLD + TF + e % padg] = 0.0f; }
        }
    }
    if (XF > F) {      // ... and [F, XF) of the weights
        const int padc = XF - F;
        FOR_TID(e, DRGNN_H1 * padc) { s.w1t[(e / padc) * XLD + F + e % padc] = 0.0f; s.ws1t[(e / padc) * XLD + F + e % padc] = 0.0f; }
    }
    FOR_TID(i, 1) {
        ((int*)s.misc)[STEP_M_BAD] = m_bad | bad_shape;
        ((int*)s.misc)[STEP_M_Y] = m_y;
        s.misc[STEP_M_WY] = m_wy;
        s.misc[STEP_M_DENOM] = m_denom;
    }
    BARRIER();
    EXIT_AFTER(1);
    EXIT_AFTER(2);
    // ---- B: conv1's product ------------------------------------------------------------------------------------------------
    PH(2) step2_conv1<KIND, XF, XG>(Nh, nmax, (XG == 2) ? sgl : s.G, XG ? xgl : s.xs, s.w1t, s.ws1t, s.b1, s.dv0, s.sc0, s.z1, dummy, XG ? s.hord + nbase : nullptr, TF);
    FOR_TID(e, (step_pad4(d.C) - d.C) * STEP_XPLD) { s.xp[d.C * STEP_XPLD + e] = 0.0f; }
    BARRIER();
    EXIT_AFTER(3);
    // ---- C: depth-0 cluster max over contiguous rows, published to the partner; the second burst is filed -----------------
    PH(3) step2_cluster_max(Ch, s.hmp, qbase, nbase, s.cid, s.z1, s.xp, s.a0, (SPLIT == 2) ? x_xp_own : nullptr, tag);
    burst_store_wt(bw2, s.wc2t, STEP2_TSLD);
    if (TRAIN) burst_store_w(bw2, s.wc2n, STEP2_TSLD);
    burst_store_wt(bs2, s.wc2t + DRGNN_H1, STEP2_TSLD);
    if (TRAIN) burst_store_w(bs2, s.wc2n + DRGNN_H1 * STEP2_TSLD, STEP2_TSLD);
    step_wblock_store(wreg, hf, 0, s.wb);
    wstage_store(wst2);
    unsigned long long w_first = 0ull;
    if (SPLIT == 2 && (int)threadIdx.x < Co * DRGNN_H1) w_first = xchg_peek(x_xp_oth + threadIdx.x);
    BARRIER();
    EXIT_AFTER(4);
    if (SPLIT == 2) {
        // ---- D: the partner's pooled features -> xp ------------------------------------------------------------------------
        PH(4) step2_receive(Co, s.cid, qbase_o, x_xp_oth, w_first, tag, fault, s.xp);
        BARRIER();
    }
    EXIT_AFTER(5);
    // ---- E: [S | T] of the own pooled rows ---------------------------------------------------------------------------------
    PH(5) step2_pooled_gather<KIND, EIdx>(Ch, s.cid, qbase, s.rp1, (const EIdx*)s.cx1, s.ew1, s.dv1, s.sc1, s.xp, s.u2);
    FOR_TID(e, (step_pad4(Ch) - Ch) * STEP2_TSLD) { s.u2[Ch * STEP2_TSLD + e] = 0.0f; }
    BARRIER();
    EXIT_AFTER(6);
    // ---- F: Z2 = relu([S | T] [Wnbr ; Wself] + b) ---------------------------------------------------------------------------
    PH(6) step_gemm_nn<true>(Ch, 2, DRGNN_H2, s.u2, STEP2_TSLD, s.wc2t, STEP2_TSLD, s.z2, STEP2_TSLD, dummy, s.b2,
                             (KIND == DRGNN_FOUT) ? s.dv1 : nullptr);
    BARRIER();
    EXIT_AFTER(7);
    // ---- G: depth-1 max + partial readout of the own depth-1 clusters --------------------------------------------------------
    PH(7) step2_pool_readout(Kh, s.mp1, kbase, qbase, s.z2, s.a1, s.xr + DRGNN_H2, (SPLIT == 2) ? x_ro_own : nullptr, tag);
    BARRIER();
    EXIT_AFTER(8);

    // ---- FC head + loss + their backward (both halves alike; half 0 writes) -----------------------------------------------
    const float keep_scale = (hf.p_drop > 0.0f) ? 1.0f / (1.0f - hf.p_drop) : 1.0f;
    const double pt = (double)hf.p_drop * 4294967296.0;
    const uint32_t thresh = (hf.p_drop > 0.0f) ? (uint32_t)(pt > 4294967295.0 ? 4294967295.0 : pt) : 0u;
    float* hp = hf.partials + (long)g * head_compact_floats(R, WREF, O);
    float* p_dhid = hp;
    float* p_hw2 = p_dhid + WREF;
    float* p_hb2 = p_hw2 + (long)O * WREF;
    float* p_loss = p_hb2 + O;
    if (TRAIN && g == 0 && half == 0) { FOR_TID(i, 1) { a.step2[1] = (int32_t)tag; } }     // Adam's step index
    if (TRAIN) { FOR_TID(item, step_pad4(Ch) * STEP2_TSLD) { s.z2[item] = 0.0f; } }      // Z2 is consumed: becomes dZ2 (+ zero K padding)
    const float inv_c1 = 1.0f / (float)(d.C1 > 0 ? d.C1 : 1);
    int bad; memcpy(&bad, &s.misc[STEP_M_BAD], 4);
    PH(8) step2_head_fc1<WREF, SPLIT>(hf, g, half, s.wb, s.hb1, s.xr, s.hid, x_ro_oth, tag, done, thresh, keep_scale, inv_c1,
                                      bad, fault, const_cast<float*>(hf.readout) + (long)g * R);
    BARRIER();
    EXIT_AFTER(9);
    PH(9) step_head_loss<WREF, true>(hf, g, half, s.hid, s.hw2, s.hb2, s.misc, keep_scale, s.dhid, p_dhid, p_hw2, p_hb2, p_loss);
    if (!TRAIN) return;
    BARRIER();
    EXIT_AFTER(10);
    float* part_w = a.partials + ((long)g * SPLIT + half) * a.n_partial;
    float* p_w1n = part_w;
    float* p_b1 = p_w1n + 2L * F * DRGNN_H1;
    float* p_w2n = p_b1 + DRGNN_H1;
    float* p_b2 = p_w2n + 2 * DRGNN_H1 * DRGNN_H2;
    // ---- J: d readout -> dZ2, d[Wnbr ; Wself] and db2 (sparse sums) -------------------------------------------------------------
    PH(10) step2_head_dreadout<WREF>(hf, s.wb, s.dhid, s.a1, Kh, inv_c1, s.u2, s.z2, p_w2n, p_b2);
    BARRIER();
    EXIT_AFTER(11);

    // ---- backward body -----------------------------------------------------------------------------------------------------
    // ---- K: d[S | T] (dS published, pre-multiplied by d_i) ---------------------------------------------------------------------
    PH(11) step2_gemm_dst(Ch, s.z2, s.wc2n, s.cid, qbase, s.dv1, s.dsf, s.dt, (SPLIT == 2) ? x_ds_own : nullptr, tag, dummy);
    if (SPLIT == 2) { w_first = ((int)threadIdx.x < Co * DRGNN_H1) ? xchg_peek(x_ds_oth + threadIdx.x) : 0ull; }
    BARRIER();
    EXIT_AFTER(12);
    if (SPLIT == 2) {
        // ---- L: the partner's d dS ---------------------------------------------------------------------------------------------
        PH(4) step2_receive(Co, s.cid, qbase_o, x_ds_oth, w_first, tag, fault, s.dsf);
        BARRIER();
    }
    EXIT_AFTER(13);
    // ---- M: d xp of the own pooled rows (dense rows; xp's own rows are dead: kept apart in u2's place? no -- in dt's rows) ----
    PH(13) step2_pooled_gather_bwd<KIND, EIdx>(Ch, s.cid, qbase, s.cp1, (const EIdx*)s.rx1, (const EIdx*)s.ts1, s.ew1, s.dv1, s.sc1,
                                               s.dsf, s.dt, s.z2);
    BARRIER();
    EXIT_AFTER(14);
    // ---- N: dWn, dWs, db1 through the depth-0 argmax ------------------------------------------------------------------------------
    PH(16) step2_dw1_sparse<XF, XG>(Ch, s.a0, s.z2, (XG == 2) ? sgl : s.G, XG ? xgl : s.xs, s.dv0, s.sc0, p_w1n, p_w1n + (long)F * DRGNN_H1, p_b1, F,
                                    XG ? s.hord + nbase : nullptr, TF);
}

#endif  // !DRGNN_EMU
#endif
