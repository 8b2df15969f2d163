// drgnn_step3.h -- fused training step of GINet (one workgroup per (graph, branch)), AGGREGATION FIRST at BOTH levels, node
// rows in the hierarchical order of the topology builder.
//
// Why: GINetConvLayer's attention is identically 1 (ginet.py:50-73), so
//          z_i = relu( sum_{e: row = i} W x_col(e) ) = relu( G_i W ),     G = A X
// and x is a leaf (no d loss / d x): the backward of conv1 is ONE product  dW1 = G^T dZ1 -- the transposed aggregation of the
// product-first form (CSC0, its staging and its 1 us phase) does not exist.  With the rows of G / Z1 kept in the hierarchical
// order (DRGNN_TI_HORD / HMP0: the members of a depth-0 cluster are CONSECUTIVE rows) the cluster maximum runs over contiguous
// rows without member lists.  The pooled level (conv2 aggregation first, pooling + readout, head with the partner branch's
// readout, d readout, dS / dW2, transposed pooled gather through the depth-0 argmax) is drgnn_step.h's, row = pooled node id.
//
// The aggregation itself is not formed here: G = A X depends on the inputs only, so the topology builder forms it together with
// the topology (DRGNN_TOPO_TILES: S rows in node order + the inverse hierarchical order; include/drgnn.h) -- in the workgroups
// co-launched with the PREVIOUS step, or once per graph in cached-topology mode -- and this kernel's prologue loads the S rows
// of the graph, filing row i at its hierarchical position.
// Per branch workgroup: 12 barrier-separated phases (drgnn_step.h: 16), LDS at SYN size 77 KB (136 KB), staged index words
// per graph 2 500 (6 300).  Same sums as the reference up to the association (G W instead of the per-edge products): parity
// 1e-4 like every other kernel (tests/test_gpu_fused_fullsize.py, test_gpu_width_classes.py: the reference's goldens directly).
// GPU only: the host emulation steps GINet through drgnn_step.h.  Instantiated per padded feature width 16 / 32 / 48 / 64
// (any feature count up to 64: the tiles' rows are padded), for training and for inference launches (TRAIN = false: forward +
// head), with one workgroup per (graph, branch) or both branches in one workgroup (net_step3_graph_both); launched by
// train_step_impl whenever the workspace holds the hierarchical order and the tiles and the head is the reference's
// (step_pick); everything else keeps the drgnn_step.h / drgnn_step1.h kernels.
#ifndef DRGNN_STEP3_H
#define DRGNN_STEP3_H

#include "drgnn_step2.h"
#include <type_traits>

#ifndef DRGNN_EMU
struct Step3Scratch {
    float* misc; float* xr; float* hid; float* dhid; float* hb1; float* wb;
    float* w1t; float* w2t; float* w2n;
    int* hmp;
    int* rp1; int* cx1; int* cp1; int* rx1; int* mp1; int* mem1;
    short* a0; short* a1;
    float* G; float* z1;
    float* xp; float* u2; float* z2; float* p2;
    float* hw2; float* hb2;
    float* end; float* gp;
};
#define STEP3_Z2LD (DRGNN_H2 + 4)
// the depth-1 argmax is kept COLUMN-major here: a1[c][STEP3_A1LD(capC)] (an even stride: columns start on word boundaries)
#define STEP3_A1LD(capC) (((capC) + 1) & ~1)
#define STEP3_CARVE_LIST(X)                                                                    \
    X(misc, 128)                                                                               \
    X(xr, 2 * DRGNN_H2)                                                                        \
    X(hid, H)                                                                                  \
    X(dhid, H)                                                                                 \
    X(hb1, H)                                                                                  \
    X(wb, step_gp_words((int)H))                                                               \
    X(w1t, DRGNN_H1 * xld)                                                                     \
    X(w2t, DRGNN_H2 * STEP_XPLD)                                                               \
    X(w2n, DRGNN_H1 * (DRGNN_H2 + 4))                                                          \
    X(hmp, capC + 1)                                                                           \
    X(rp1, capC + 1)                                                                           \
    X(cx1, capE)                                                                               \
    X(cp1, capC + 1)                                                                           \
    X(rx1, capE)                                                                               \
    X(mp1, capC + 1)                                                                           \
    X(mem1, capC)                                                                              \
    X(a0, ((long)STEP3_A1LD(capC) * DRGNN_H1 + 1) / 2)                                         \
    X(a1, ((long)STEP3_A1LD(capC) * DRGNN_H2 + 1) / 2)                                         \
    X(G, (long)(capN + 4) * xld)                                                               \
    X(z1, (long)(capN + 4) * DRGNN_H1)                                                         \
    X(xp, (long)(capC + 4) * STEP_XPLD)                                                        \
    X(u2, (long)(capC + 4) * STEP_XPLD)                                                        \
    X(z2, (long)(capC + 4) * STEP3_Z2LD)                                                       \
    X(p2, (long)(capC + 4) * STEP_XPLD)                                                        \
    X(hw2, (long)O * H)                                                                        \
    X(hb2, O)
#endif  // !DRGNN_EMU

// (host + device; the emulation build answers "never")
HD int64_t step3_scratch_words(int64_t F, int64_t capN, int64_t capE, int64_t capC, int64_t H, int64_t O) {
    const int64_t xld = step_pad16((int)F) + 4;
    int64_t w = 0;
#ifndef DRGNN_EMU
#define X(name, words) w += (((int64_t)(words) + 3) & ~(int64_t)3);
    STEP3_CARVE_LIST(X)
#undef X
#else
    (void)xld; (void)capN; (void)capE; (void)capC; (void)H; (void)O;
    w = (int64_t)1 << 40;
#endif
    return w + 16;
}

#ifndef DRGNN_EMU
template <int CLS>
DEV Step3Scratch step3_carve(float* base, int F, int capN, int capE, int capC, int H, int O) {
    const int xld = step_pad16(F) + 4;
    Step3Scratch s;
    int o = 0;
#define X(name, words)                                                                          \
    { int off = o; if (CLS == 0) { STEP_PIN(off); } s.name = (decltype(s.name))(base + off);    \
      o = off + (int)(((long)(words) + 3) & ~3L); }
    STEP3_CARVE_LIST(X)
#undef X
    s.end = base + o;
    s.gp = s.wb;      // fc1's column block is dead after d readout: the K-split products keep their partial tiles there
    return s;
}

// ---- phase C: depth-0 cluster max over CONTIGUOUS rows; results filed under the pooled node id cid[q] --------------------
// (argmax = row position of the winner, -1 where no gradient flows; kept COLUMN-major: a0[c][a0ld], as the depth-1 argmax)
DEV void step3_cluster_max(int nc, const int* hmp, const int* cid, const float* z, float* xp, short* a0, int a0ld) {
    FOR_TID(item, nc * DRGNN_H1) {
        const int q = item >> 4, c = item & 15;
        const int plo = hmp[q], phi = hmp[q + 1];
        const int j = cid[q];
        float best = DRGNN_NEG_INF;
        int arg = -1;
        for (int p = plo; p < phi; p += 4) {
            int mm[4];
            float vv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) mm[t] = (p + t < phi) ? p + t : phi - 1;
#pragma unroll
            for (int t = 0; t < 4; ++t) vv[t] = z[mm[t] * DRGNN_H1 + c];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (vv[t] > best) { best = vv[t]; arg = mm[t]; }
        }
        if (arg < 0) best = 0.0f;
        xp[ROW24(j, STEP_XPLD) + c] = best;
        a0[c * a0ld + j] = (short)((best > 0.0f) ? arg : -1);
    }
}

// ---- sparse weight gradients -----------------------------------------------------------------------------------------------
// The gradients that reach a convolution through a max-pool are SPARSE: dZ2 is non-zero only in the rows that won a depth-1
// cluster (C1 x 32 entries of C x 32, all of a column equal to that column's d readout), dZ1 only in the rows that won a
// depth-0 cluster (C x 16 of N x 16).  The weight gradients are therefore short sums of gathered operand rows,
//      dW2[f][c] = v_c  sum_k S2[a1[k][c]][f]              dW1[f][h] = sum_j dXP[j][h] G[a0[j][h]][f]
// formed by lane groups in registers and stored at once -- no dense K = rows product, no partial tiles through LDS, no
// barrier of their own, no zeroed dZ1 array and no scatter into it (skip-phase builds: dW2 0.74 us, dW1 1.31 us as products).
// d readout (this branch's 32 columns) scattered through the depth-1 argmax into dZ2 (for dS) + dW2; 32 lanes per column:
// 4 float4 groups of the 16 S2 columns x 8 slices of the clusters
template <int HC>
DEV void step3_dreadout_dw2(const float* wb, const float* dhid, const short* a1, int a1ld, int C1, const float* s2, float* z2,
                            float* g_dw2) {
    const float inv = 1.0f / (float)(C1 > 0 ? C1 : 1);
    for (int t = threadIdx.x; t < DRGNN_H2 * 32; t += DRGNN_NTHREADS) {
        const int c = t >> 5, q = t & 31;
        float acc = 0.0f;
#pragma unroll
        for (int h = q; h < HC; h += 32) acc = fmaf(dhid[h], wb[h * STEP_WBLD + c], acc);
        const float v = lanes32_sum(acc) * inv;
        const short* a1c = a1 + c * a1ld;
        for (int k = q; k < C1; k += 32) {
            const int r = a1c[k];
            if (r >= 0) z2[ROW24(r, STEP3_Z2LD) + c] = v;
        }
        const int sl = q & 7, f4 = q >> 3;
        drgnn_f4 sum = {0.f, 0.f, 0.f, 0.f};
        for (int k = sl; k < C1; k += 8) {
            const int r = a1c[k];
            if (r >= 0) {
                const drgnn_f4 row = *(const drgnn_f4*)(s2 + ROW24(r, STEP_XPLD) + 4 * f4);
                sum[0] += row[0]; sum[1] += row[1]; sum[2] += row[2]; sum[3] += row[3];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) sum[i] = lanes8_sum(sum[i]);
        if (sl == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) g_dw2[(4 * f4 + i) * DRGNN_H2 + c] = v * sum[i];
        }
    }
}
// dXP[j][0:16] = sum over the CSC1 column of j of dS rows (dense rows of LD floats; 16 lanes per pooled node as step_gather_rows)
template <int LD>
DEV void step3_gather_dxp(int n, const int* cp, const int* ridx, const float* src, float* dst) {
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
        a0 += dpp_take<0x128>(a0); a1 += dpp_take<0x128>(a1); a2 += dpp_take<0x128>(a2); a3 += dpp_take<0x128>(a3);
        a0 += dpp_take<0x124>(a0); a1 += dpp_take<0x124>(a1); a2 += dpp_take<0x124>(a2); a3 += dpp_take<0x124>(a3);
        if (sl == 0 && j < n) *(drgnn_f4*)(dst + j * LD + c) = drgnn_f4{a0, a1, a2, a3};
    }
}
// dW1[f][h] = sum over the pooled nodes j of dXP[j][h] G[a0[j][h]][f].  Wave = channel h; in a wave NCH feature chunks (float4)
// x NSL slices of the pooled nodes (consecutive lanes: the slice sums meet in DPP adds; Dw1Shape, drgnn_step2.h)
// SG: G = the graph's S rows in memory (node order, `gtf` floats apart), hord: row position -> node
template <int XF, bool SG = false>
DEV void step3_dw1_sparse(int C, const short* a0, int a0ld, const float* dxp, const float* G, float* g_dw1, int F,
                          const int* hord = nullptr, int gtf = 0) {
    constexpr int XLD = XF + 4, NSL = Dw1Shape<XF>::NSL;
    const int h = threadIdx.x >> 6, fc = (threadIdx.x & 63) / NSL, sl = threadIdx.x & (NSL - 1);
    const bool live = 4 * fc < XF;
    drgnn_f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int j = sl; j < C; j += 4 * NSL) {      // four pooled nodes per trip in flight
        int arg[4];
        float d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int jj = j + NSL * u;
            arg[u] = (jj < C && live) ? (int)a0[h * a0ld + jj] : -1;
            d[u] = (jj < C) ? dxp[jj * STEP_XPLD + h] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (arg[u] >= 0) {
                // (SG: chunks past the row's end -- padded widths -- hold nothing: their sums are never stored, F <= gtf)
                const drgnn_f4 g = !SG ? *(const drgnn_f4*)(G + ROW24(arg[u], XLD) + 4 * fc)
                                       : (4 * fc < gtf ? *(const drgnn_f4*)(G + (long)hord[arg[u]] * gtf + 4 * fc) : drgnn_f4{0.f, 0.f, 0.f, 0.f});
                acc[0] = fmaf(d[u], g[0], acc[0]); acc[1] = fmaf(d[u], g[1], acc[1]);
                acc[2] = fmaf(d[u], g[2], acc[2]); acc[3] = fmaf(d[u], g[3], acc[3]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = step_slices_sum<NSL>(acc[i]);
    if (sl == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (4 * fc + i < F) g_dw1[(4 * fc + i) * DRGNN_H1 + h] = acc[i];
    }
}

// Z1 = relu(S W1) with the S rows read from MEMORY (the S-from-memory form): row position r of the hierarchical order is node
// hord[r] of the graph's rows `sgl` (gtf floats apart).  The fragments, the order of the k chunks and of the matrix instructions
// are step_gemm_nn's: the same bits as the staged form.  All of a row's chunks are requested at once.
template <int XF>
DEV void step3_conv1_sg(int n, const float* sgl, int gtf, const int* hord, const float* w1t, float* z1, int* dummy) {
    constexpr int XLD = XF + 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int lr = lane & 15, lq = lane >> 4;
    const int units = (n + 15) >> 4;
    for (int ti = wave; ti < units; ti += DRGNN_NWAVES) {
        const int prow = ti * 16 + lr;
        const int row = prow < n ? prow : n - 1;          // rows past the graph: any valid row (results discarded)
        const float* ag = sgl + (long)hord[row] * gtf + 4 * lq;
        drgnn_f4 av[XF / 16];
#pragma unroll
        for (int k0 = 0; k0 < XF; k0 += 16)
            av[k0 / 16] = (k0 + 4 * lq < gtf) ? *(const drgnn_f4*)(ag + k0) : drgnn_f4{0.f, 0.f, 0.f, 0.f};
        const float* bp = w1t + lr * XLD + 4 * lq;
        drgnn_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k0 = 0; k0 < XF; k0 += 16) {
            const drgnn_f4 a = av[k0 / 16], b = *(const drgnn_f4*)(bp + k0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ci = ti * 16 + lq * 4 + r;
            float* p = (ci < n) ? z1 + ci * DRGNN_H1 + lr : (float*)dummy + lane;
            const float v = acc[r];
            *p = (v < 0.0f) ? 0.0f : v;
        }
    }
}

// =========================================================================================================================
// XF: padded feature width (the host has checked step_burst_guaranteed and the reference head width 128); CLS as in
// drgnn_step.h; `late` as in net_step_graph.  TRAIN = false: the inference launch (forward + head, predictions only: the
// arrays only the backward reads are not staged, the kernel returns behind the head).
template <int XF, bool GATHER, int CLS, bool TRAIN = true>
DEV void net_step3_graph(const StepArgs& a, const GraphDims& d_in, int g, int gi, int br, float* scratch, int capN, int capE,
                         int capC, bool late, int cnt_c, int cnt_e1, int cnt_c1) {
    static_assert(XF == 16 || XF == 32 || XF == 48 || XF == 64, "width-specialised kernels only");
    if (CLS == 1) { capN = STEP_CLS_N; capE = STEP_CLS_E; capC = STEP_CLS_C; }
    GraphDims d = d_in;
    const int bC = late ? imin(d.N, capC) : d.C, bE1 = late ? d.E : d.E1, bC1 = late ? imin(d.N, capC) : d.C1;
    const TopoView& tv = a.tv;
    const HeadFused& hf = a.hf;
    constexpr int nb = 2, R = 2 * DRGNN_H2, WREF = 128;
    constexpr int XLD = XF + 4, Z2LD = STEP3_Z2LD, W2NLD = DRGNN_H2 + 4;
    const int F = a.net.n_feat;
    const int O = hf.O;
    Step3Scratch s = step3_carve<CLS>(scratch, XF, capN, capE, capC, WREF, O);
    EXIT_AFTER(0);
    WBlockRegs<1> wreg, wother;
    int* const dummy = (int*)(s.misc + 64);
    const uint32_t done = (uint32_t)a.step2[0];
    const uint32_t tag = done + 1u;
    const drgnn_conv_params& c1 = a.net.conv1[br];
    const drgnn_conv_params& c2 = a.net.conv2[br];

    // ---- prologue: everything this graph needs, two register bursts requested back to back --------------------------------
    PHASE_MARK();
    const int TF = (F + 3) & ~3;                       // row length of the tiles (padded with zeros by the builder)
    const float* sgl = a.tiles + (long)d.n0 * TF;     // the graph's S rows (node order)
    BurstX<4> bx;
    BurstRowMap<4> brow;
    BurstW<1> bw1, bw2;
    WaveStage wst;
    const int my_wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    auto stage_job = [&](int burst, int w) -> StageJob {
        const int32_t* const* P = tv.p;
        StageJob j = {nullptr, 0, nullptr, 0};
        switch (burst * 16 + w) {
        // (thirteen small arrays: ONE burst, one array per wave)
        case 16 + 0: j = StageJob{P[DRGNN_TI_HMP0] + d.rowbase, bC + 1, s.hmp, 0}; break;
        case 16 + 1: j = StageJob{P[DRGNN_TI_MEM1] + d.n0, bC, s.mem1, 0}; break;
        case 16 + 2: j = StageJob{P[DRGNN_TI_MPTR1] + d.rowbase, bC1 + 1, s.mp1, 0}; break;
        case 16 + 3: j = StageJob{hf.b1, WREF, s.hb1, 0}; break;
        case 16 + 4: j = stage_half(StageJob{hf.w2, O * WREF, s.hw2, 0}, 0); break;
        case 16 + 5: j = stage_half(StageJob{hf.w2, O * WREF, s.hw2, 0}, 1); break;
        case 16 + 6: j = StageJob{hf.b2, O, s.hb2, 0}; break;
        case 16 + 7: j = StageJob{P[DRGNN_TI_ROWPTR1] + d.rowbase, bC + 1, s.rp1, 0}; break;
        case 16 + 8: j = stage_half(StageJob{P[DRGNN_TI_COL1] + d.e0, bE1, s.cx1, 0}, 0); break;
        case 16 + 9: j = stage_half(StageJob{P[DRGNN_TI_COL1] + d.e0, bE1, s.cx1, 0}, 1); break;
        case 16 + 10: if (TRAIN) j = StageJob{P[DRGNN_TI_COLPTR1] + d.rowbase, bC + 1, s.cp1, 0}; break;
        case 16 + 11: if (TRAIN) j = stage_half(StageJob{P[DRGNN_TI_ROWIDX1] + d.e0, bE1, s.rx1, 0}, 0); break;
        case 16 + 12: if (TRAIN) j = stage_half(StageJob{P[DRGNN_TI_ROWIDX1] + d.e0, bE1, s.rx1, 0}, 1); break;
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
    burst_load_x(bx, sgl, d.N, TF);
    burst_load_rowmap(brow, bx, tv.p[DRGNN_TI_IHORD] + d.n0, d.N);
    burst_load_w(bw1, c1.w_nbr, c1.nbr_sk, c1.nbr_sh, F, DRGNN_H1);
    wstage_load(wst, stage_job(1, my_wave));
    burst_load_w(bw2, c2.w_nbr, c2.nbr_sk, c2.nbr_sh, DRGNN_H1, DRGNN_H2);
    step_wblock_load(wreg, hf, br);
    step_wblock_load(wother, hf, 1 - br);
    burst_store_x4_rows(bx, brow, s.G, XLD);
    burst_store_wt(bw1, s.w1t, XLD);
    wstage_store(wst);
    // zero padding the predicate-free products rely on: G rows [N, pad4(N)) and (F < XF) the k columns [F, XF)
    FOR_TID(e, (step_pad4(d.N) - d.N) * XLD) { s.G[d.N * XLD + e] = 0.0f; }
    if (XF > TF) {
        const int padg = XF - TF;
        FOR_TID(e, d.N * padg) { s.G[(e / padg) * XLD + TF + e % padg] = 0.0f; }
    }
    if (XF > F) {
        const int padc = XF - F;
        FOR_TID(e, DRGNN_H1 * padc) { s.w1t[(e / padc) * XLD + F + e % padc] = 0.0f; }
    }
    FOR_TID(i, 1) {
        ((int*)s.misc)[STEP_M_BAD] = m_bad;
        ((int*)s.misc)[STEP_M_Y] = m_y;
        s.misc[STEP_M_WY] = m_wy;
        s.misc[STEP_M_DENOM] = m_denom;
    }
    BARRIER();
    EXIT_AFTER(1);
    if (late) { d.C = WG_UNIFORM(cnt_c); d.E1 = WG_UNIFORM(cnt_e1); d.C1 = WG_UNIFORM(cnt_c1); }
    int bad_shape = 0;
    if (d.C > capC || d.E1 > d.E || d.C1 > capC) {      // malformed input (flagged by the builder): stay inside LDS, poison
        d.C = imin(d.C, capC); d.E1 = imin(d.E1, d.E); d.C1 = imin(d.C1, capC);
        bad_shape = 1;
    }

    EXIT_AFTER(2);
    // ---- B: Z1 = relu(G W1); the second burst is filed ---------------------------------------------------------------------
    PH(2) step_gemm_nn<true>(d.N, 1, XF, s.G, XLD, s.w1t, XLD, s.z1, DRGNN_H1, dummy);
    burst_store_wt(bw2, s.w2t, STEP_XPLD);
    if (TRAIN) burst_store_w(bw2, s.w2n, W2NLD);
    step_wblock_store(wreg, hf, br, s.wb);
    FOR_TID(e, (step_pad4(d.C) - d.C) * STEP_XPLD) { s.xp[d.C * STEP_XPLD + e] = 0.0f; }
    if (bad_shape) { FOR_TID(i, 1) { ((int*)s.misc)[STEP_M_BAD] = 1; } }
    BARRIER();
    EXIT_AFTER(3);
    // ---- C: depth-0 cluster max over contiguous rows -----------------------------------------------------------------------
    PH(3) step3_cluster_max(d.C, s.hmp, s.mem1, s.z1, s.xp, s.a0, STEP3_A1LD(capC));
    BARRIER();
    EXIT_AFTER(4);
    // ---- E: S = A1 XP (16-wide gather of pooled rows) -----------------------------------------------------------------------
    PH(4) step_gather_rows<STEP_XPLD, int>(d.C, s.rp1, s.cx1, s.xp, s.u2);
    FOR_TID(e, (step_pad4(d.C) - d.C) * STEP_XPLD) { s.u2[d.C * STEP_XPLD + e] = 0.0f; }
    BARRIER();
    EXIT_AFTER(5);
    // ---- F: Z2 = relu(S W2) ---------------------------------------------------------------------------------------------------
    PH(5) step_gemm_nn<true>(d.C, 2, DRGNN_H1, s.u2, STEP_XPLD, s.w2t, STEP_XPLD, s.z2, Z2LD, dummy);
    BARRIER();
    EXIT_AFTER(6);
    // ---- G: depth-1 max + readout, published to the partner branch ------------------------------------------------------------
    PH(6) step_pool_readout<Z2LD>(d.C1, s.mp1, s.mem1, s.z2, s.a1, s.misc, s.xr,
                                  const_cast<float*>(hf.readout) + (long)g * R + br * DRGNN_H2, nullptr,
                                  a.xchg + (long)g * nb * WREF + br * DRGNN_H2, tag, STEP3_A1LD(capC));
    BARRIER();
    EXIT_AFTER(8);

    // ---- FC head + loss + their backward -----------------------------------------------------------------------------------
    const float keep_scale = (hf.p_drop > 0.0f) ? 1.0f / (1.0f - hf.p_drop) : 1.0f;
    const double pt = (double)hf.p_drop * 4294967296.0;
    const uint32_t thresh = (hf.p_drop > 0.0f) ? (uint32_t)(pt > 4294967295.0 ? 4294967295.0 : pt) : 0u;
    float* hp = hf.partials + (long)g * head_compact_floats(R, WREF, O);
    float* p_dhid = hp;
    float* p_hw2 = p_dhid + WREF;
    float* p_hb2 = p_hw2 + (long)O * WREF;
    float* p_loss = p_hb2 + O;
    if (TRAIN && g == 0 && br == 0) { FOR_TID(i, 1) { a.step2[1] = (int32_t)tag; } }     // Adam's step index
    if (TRAIN) { FOR_TID(item, step_pad4(d.C) * Z2LD) { s.z2[item] = 0.0f; } }      // Z2 is consumed: becomes dZ2 (+ zero K padding)
    PH(8) step_head_fc1<WREF, true>(hf, g, br, nb, s.wb, wother, s.hb1, s.xr, s.hid, a.xchg + (long)g * nb * WREF, tag, done,
                                    thresh, keep_scale, a.step2 + 2);
    BARRIER();
    EXIT_AFTER(9);
    if (!TRAIN) {
        // inference launches all carry the same tag (the step counter does not move): the reader clears the words it
        // consumed (all 16 waves have, past the barrier), so that the next launch cannot pick up this one's values
        FOR_TID(c, DRGNN_H2) {
            __hip_atomic_store(a.xchg + (long)g * nb * WREF + (1 - br) * DRGNN_H2 + c, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    PH(9) step_head_loss<WREF, true>(hf, g, br, s.hid, s.hw2, s.hb2, s.misc, keep_scale, s.dhid, p_dhid, p_hw2, p_hb2, p_loss);
    if (!TRAIN) return;
    BARRIER();
    EXIT_AFTER(10);
    float* part_w = a.partials + ((long)g * nb + br) * a.n_partial;
    float* p_w1n = part_w;
    float* p_w2n = p_w1n + 2L * F * DRGNN_H1 + DRGNN_H1;
    // d readout scattered into dZ2 + dW2 (sparse: see step3_dreadout_dw2)
    PH(10) step3_dreadout_dw2<WREF>(s.wb, s.dhid, s.a1, STEP3_A1LD(capC), d.C1, s.u2, s.z2, p_w2n);
    BARRIER();
    EXIT_AFTER(11);

    // ---- backward body -----------------------------------------------------------------------------------------------------
    // dS = dZ2 W2^T (rows of STEP_XPLD floats)
    PH(11) step_gemm_nn(d.C, 1, DRGNN_H2, s.z2, Z2LD, s.w2n, W2NLD, s.p2, STEP_XPLD, dummy);
    BARRIER();
    EXIT_AFTER(12);
    // dXP = A1^T dS (dense rows; xp is dead)
    PH(13) step3_gather_dxp<STEP_XPLD>(d.C, s.cp1, s.rx1, s.p2, s.xp);
    BARRIER();
    EXIT_AFTER(14);
    // dW1 through the depth-0 argmax (sparse)
    PH(16) step3_dw1_sparse<XF>(d.C, s.a0, STEP3_A1LD(capC), s.xp, s.G, p_w1n, F);
}


// =========================================================================================================================
// The same step with BOTH branches of a graph in ONE workgroup, one after the other: the launch layout beyond the resident
// batch size (2 B + builder workgroups > CUs), where a workgroup must never wait for a partner (drgnn_step1.h's role for the
// product-first kernels).  The branches convolve over the same edge_index (ginet.py:101-128), so G = A X -- the S rows of the
// tiles -- is SHARED: loaded once, multiplied with either branch's W1, and the K operand of both dW1.  Per-branch state kept
// from the forward to the backward: the argmax of both depths, S2 = A1 XP (dW2's operand) and dZ2; Z1 / XP / dS are
// reused.  fc1's column block of branch 0 sits in LDS, branch 1's in registers until branch 0's d readout is done.
// 25 barrier-separated phases, 101 KB of LDS at SYN size.
struct Step3BScratch {
    float* misc; float* xr; float* hid; float* dhid; float* hb1; float* wb;
    float* w1t[2]; float* w2t[2]; float* w2n[2];
    int* hmp;
    int* rp1; int* cx1; int* cp1; int* rx1; int* mp1; int* mem1;
    short* a0[2]; short* a1[2];
    float* G; float* z1;
    float* xp; float* u2[2]; float* z2[2]; float* p2;
    float* hw2; float* hb2;
    int* hord;
    float* end; float* gp;
};
// sg: the S-FROM-MEMORY form (net_step3_graph_both<..., SG = true>): the S rows of the tiles are not staged -- conv1's product
// and dW1 read them from memory through the hierarchical order (hord: position -> node), 4 bytes of LDS per node instead of
// 4 (XF + 4): graphs beyond the LDS budget of the staged form (200 - 270 nodes, by width) up to what the builder forms tiles
// for (drgnn_topology_tiles_ok)
#define STEP3B_CARVE_LIST(X)                                                                   \
    X(misc, 128)                                                                               \
    X(xr, 2 * DRGNN_H2)                                                                        \
    X(hid, H)                                                                                  \
    X(dhid, H)                                                                                 \
    X(hb1, H)                                                                                  \
    X(wb, step_gp_words((int)H))                                                               \
    X(w1t[0], DRGNN_H1 * xld)                                                                  \
    X(w1t[1], DRGNN_H1 * xld)                                                                  \
    X(w2t[0], DRGNN_H2 * STEP_XPLD)                                                            \
    X(w2t[1], DRGNN_H2 * STEP_XPLD)                                                            \
    X(w2n[0], DRGNN_H1 * (DRGNN_H2 + 4))                                                       \
    X(w2n[1], DRGNN_H1 * (DRGNN_H2 + 4))                                                       \
    X(hmp, capC + 1)                                                                           \
    X(rp1, capC + 1)                                                                           \
    X(cx1, capE)                                                                               \
    X(cp1, capC + 1)                                                                           \
    X(rx1, capE)                                                                               \
    X(mp1, capC + 1)                                                                           \
    X(mem1, capC)                                                                              \
    X(a0[0], ((long)STEP3_A1LD(capC) * DRGNN_H1 + 1) / 2)                                      \
    X(a0[1], ((long)STEP3_A1LD(capC) * DRGNN_H1 + 1) / 2)                                      \
    X(a1[0], ((long)STEP3_A1LD(capC) * DRGNN_H2 + 1) / 2)                                      \
    X(a1[1], ((long)STEP3_A1LD(capC) * DRGNN_H2 + 1) / 2)                                      \
    X(G, sg ? 4 : (long)(capN + 4) * xld)                                                      \
    X(z1, (long)(capN + 4) * DRGNN_H1)                                                         \
    X(xp, (long)(capC + 4) * STEP_XPLD)                                                        \
    X(u2[0], (long)(capC + 4) * STEP_XPLD)                                                     \
    X(z2[0], (long)(capC + 4) * STEP3_Z2LD)                                                    \
    X(u2[1], (long)(capC + 4) * STEP_XPLD)      /* (u2[1], z2[1], p2 in a row: the side-by-side form's second Z1) */ \
    X(z2[1], (long)(capC + 4) * STEP3_Z2LD)                                                    \
    X(p2, (long)(capC + 4) * STEP_XPLD)                                                        \
    X(hw2, (long)O * H)                                                                        \
    X(hb2, O)                                                                                  \
    X(hord, sg ? capN : 0)
#endif  // !DRGNN_EMU

HD int64_t step3b_scratch_words(int64_t F, int64_t capN, int64_t capE, int64_t capC, int64_t H, int64_t O, int sg = 0) {
    const int64_t xld = step_pad16((int)F) + 4;
    int64_t w = 0;
#ifndef DRGNN_EMU
#define X(name, words) w += (((int64_t)(words) + 3) & ~(int64_t)3);
    STEP3B_CARVE_LIST(X)
#undef X
#else
    (void)xld; (void)capN; (void)capE; (void)capC; (void)H; (void)O; (void)sg;
    w = (int64_t)1 << 40;
#endif
    return w + 16;
}

// (extra words of the side-by-side form below: per-branch Z1, XP, dS)
#define STEP3B_DUAL(XF, CLS, TRAIN) (((XF) == 32 || (XF) == 48) && (CLS) == 1)      // (training and inference instances)
HD int64_t step3b_dual_extra_words(int64_t capN, int64_t capC) {
    (void)capN;
    return 2 * ((((capC + 4) * STEP_XPLD) + 3) & ~(int64_t)3);
}

#ifndef DRGNN_EMU
// =========================================================================================================================
// The one-workgroup step with the two branches SIDE BY SIDE: waves 0 - 7 work branch 0 off, waves 8 - 15 branch 1, through the
// same barrier-separated phases (13 instead of 21 behind the prologue).  Every phase behind conv1's product is over <= 52 pooled
// rows -- a dependent chain of LDS round trips that a few lanes of a few waves wait for --, so two of them at once cost what one
// costs; conv1's product (13 row tiles per branch) takes two trips of 8 waves instead of one of 16, as before for both branches.
// Z1, XP and dS, which the branch-after-branch form reuses, exist per branch here: XP and dS behind the list's arrays (+ 9 KB at
// the capacity class's shape: 140.4 KB for the 32-wide class kernel, 155.5 KB for the 48-wide one), branch 1's Z1 in the place of
// its own [S2 | Z2] and of dS -- dead after the depth-0 cluster max, one barrier before S2 is first written.  Every sum is formed by the same lanes in the same order as in
// net_step3_graph_both: bit-identical results (test_capacity_class_kernels_equal_the_runtime_layout_bit_for_bit steps the two
// against each other).  The head and d readout stay as they are (fc1's second column block lives in registers).
#define STEP3D_NT (DRGNN_NTHREADS / 2)
#define STEP3D_NW (DRGNN_NWAVES / 2)
template <bool RELU>
DEV void step3d_gemm_nn(int vwave, int M, int NT, int K, const float* A, int lda, const float* Bt, int ldbt, float* C, int ldc,
                        int* dummy) {
    const int lane = threadIdx.x & 63;
    const int lr = lane & 15, lq = lane >> 4;
    const int units = ((M + 15) >> 4) * NT;
    for (int u = vwave; u < units; u += STEP3D_NW) {
        const int ti = (NT == 1) ? u : (u >> 1), tj = (NT == 1) ? 0 : (u & 1);
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
            if (RELU) v = (v < 0.0f) ? 0.0f : v;
            *p = v;
        }
    }
}
DEV void step3d_cluster_max(int tid, int nc, const int* hmp, const int* cid, const float* z, float* xp, short* a0, int a0ld) {
    _Pragma("nounroll") for (int item = tid; item < nc * DRGNN_H1; item += STEP3D_NT) {
        const int q = item >> 4, c = item & 15;
        const int plo = hmp[q], phi = hmp[q + 1];
        const int j = cid[q];
        float best = DRGNN_NEG_INF;
        int arg = -1;
        for (int p = plo; p < phi; p += 4) {
            int mm[4];
            float vv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) mm[t] = (p + t < phi) ? p + t : phi - 1;
#pragma unroll
            for (int t = 0; t < 4; ++t) vv[t] = z[mm[t] * DRGNN_H1 + c];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (vv[t] > best) { best = vv[t]; arg = mm[t]; }
        }
        if (arg < 0) best = 0.0f;
        xp[ROW24(j, STEP_XPLD) + c] = best;
        a0[c * a0ld + j] = (short)((best > 0.0f) ? arg : -1);
    }
}
// (step_gather_rows / step3_gather_dxp: one routine, the index arrays are the caller's)
template <int LD>
DEV void step3d_gather16(int tid, int n, const int* ptr, const int* idx, const float* src, float* dst) {
    const int items = ((n * 16) + 63) & ~63;
    for (int item = tid; item < items; item += STEP3D_NT) {
        const int i = item >> 4, sl = (item >> 2) & 3, c = (item & 3) * 4;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (i < n) {
            const int lo = ptr[i], hi = ptr[i + 1];
            for (int k = lo + sl; k < hi; k += 4) {
                const drgnn_f4 v = *(const drgnn_f4*)(src + ROW24(idx[k], LD) + c);
                a0 += v[0]; a1 += v[1]; a2 += v[2]; a3 += v[3];
            }
        }
        a0 += dpp_take<0x128>(a0); a1 += dpp_take<0x128>(a1); a2 += dpp_take<0x128>(a2); a3 += dpp_take<0x128>(a3);
        a0 += dpp_take<0x124>(a0); a1 += dpp_take<0x124>(a1); a2 += dpp_take<0x124>(a2); a3 += dpp_take<0x124>(a3);
        if (sl == 0 && i < n) *(drgnn_f4*)(dst + i * LD + c) = drgnn_f4{a0, a1, a2, a3};
    }
}
// depth-1 max-pool + readout of ONE branch by 512 lanes (step_pool_readout's own shape: DRGNN_H2 x 16 lanes), column-major argmax
template <int LDZ>
DEV void step3d_pool_readout(int tid, int C1, const int* mp, const int* mem, const float* z, short* arg, const float* misc,
                             float* xr, float* g_readout, int a1ld) {
    int bad; memcpy(&bad, &misc[STEP_M_BAD], 4);
    const float inv = 1.0f / (float)(C1 > 0 ? C1 : 1);
    static_assert(DRGNN_H2 * 16 == STEP3D_NT, "one trip of a half workgroup");
    const int c = tid >> 4, kk = tid & 15;
    float acc = 0.0f;
    for (int k = kk; k < C1; k += 16) {
        float best = DRGNN_NEG_INF;
        int am = -1;
        const int plo = mp[k], phi = mp[k + 1];
        for (int p = plo; p < phi; p += 4) {
            int mm[4];
            float vv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) mm[j] = mem[(p + j < phi) ? p + j : phi - 1];
#pragma unroll
            for (int j = 0; j < 4; ++j) vv[j] = z[ROW24(mm[j], LDZ) + c];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (vv[j] > best) { best = vv[j]; am = mm[j]; }
        }
        if (am < 0) best = 0.0f;
        arg[c * a1ld + k] = (short)((best > 0.0f) ? am : -1);
        acc += best;
    }
    acc = lanes16_sum(acc) * inv;
    if (bad) acc = DRGNN_NAN;
    if (kk == 0) { xr[c] = acc; g_readout[c] = acc; }
}
// step3_dw1_sparse with the sixteen channels on eight waves (two channels per wave, one after the other)
template <int XF>
DEV void step3d_dw1_sparse(int vwave, int C, const short* a0, int a0ld, const float* dxp, const float* G, float* g_dw1, int F) {
    constexpr int XLD = XF + 4, NSL = Dw1Shape<XF>::NSL;
    const int fc = (threadIdx.x & 63) / NSL, sl = threadIdx.x & (NSL - 1);
    const bool live = 4 * fc < XF;
#pragma unroll 1
    for (int h = vwave; h < DRGNN_H1; h += STEP3D_NW) {
        drgnn_f4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int j = sl; j < C; j += 4 * NSL) {
            int arg[4];
            float d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int jj = j + NSL * u;
                arg[u] = (jj < C && live) ? (int)a0[h * a0ld + jj] : -1;
                d[u] = (jj < C) ? dxp[jj * STEP_XPLD + h] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (arg[u] >= 0) {
                    const drgnn_f4 g = *(const drgnn_f4*)(G + ROW24(arg[u], XLD) + 4 * fc);
                    acc[0] = fmaf(d[u], g[0], acc[0]); acc[1] = fmaf(d[u], g[1], acc[1]);
                    acc[2] = fmaf(d[u], g[2], acc[2]); acc[3] = fmaf(d[u], g[3], acc[3]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = step_slices_sum<NSL>(acc[i]);
        if (sl == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (4 * fc + i < F) g_dw1[(4 * fc + i) * DRGNN_H1 + h] = acc[i];
        }
    }
}

template <int CLS, bool SG = false>
DEV Step3BScratch step3b_carve(float* base, int F, int capN, int capE, int capC, int H, int O) {
    const int xld = step_pad16(F) + 4;
    constexpr int sg = SG ? 1 : 0;
    Step3BScratch s;
    int o = 0;
#define X(name, words)                                                                          \
    { int off = o; if (CLS == 0) { STEP_PIN(off); }                                             \
      s.name = (typename std::remove_reference<decltype(s.name)>::type)(base + off);            \
      o = off + (int)(((long)(words) + 3) & ~3L); }
    STEP3B_CARVE_LIST(X)
#undef X
    s.end = base + o;
    s.gp = s.wb;
    return s;
}

// hid = dropout(relu(b1 + W1[:, 0:32] readout_0 + W1[:, 32:64] readout_1)): branch 0's column block from LDS, branch 1's from
// registers (the lane's float4 of its hidden unit's row, step_wblock_load's layout); 8 lanes per hidden unit
template <int HC>
DEV void step3b_head_fc1(const HeadFused& hf, int g, const float* wb, const WBlockRegs<1>& w1r, const float* b1, const float* xr,
                         float* hid, uint32_t step, uint32_t thresh, float keep_scale) {
    static_assert(HC * 8 <= DRGNN_NTHREADS, "one pass: 8 lanes per hidden unit");
    if ((int)(threadIdx.x & ~63u) >= HC * 8) return;
    const int t = threadIdx.x, h = t >> 3, q = t & 7;
    const drgnn_f4 w0 = *(const drgnn_f4*)(wb + h * STEP_WBLD + 4 * q);
    const drgnn_f4 x0 = *(const drgnn_f4*)(xr + 4 * q), x1 = *(const drgnn_f4*)(xr + DRGNN_H2 + 4 * q);
    const drgnn_f4 w1 = w1r.v[0];
    float p0 = fmaf(w0[0], x0[0], fmaf(w0[1], x0[1], fmaf(w0[2], x0[2], w0[3] * x0[3])));
    float p1 = fmaf(w1[0], x1[0], fmaf(w1[1], x1[1], fmaf(w1[2], x1[2], w1[3] * x1[3])));
    p0 = lanes8_sum(p0);
    p1 = lanes8_sum(p1);
    if (q == 0) {
        float v = (p0 + p1) + b1[h];
        v = v > 0.0f ? v : 0.0f;
        if (thresh) v = drgnn_keep(hf, step, g, HC, h, thresh) ? v * keep_scale : 0.0f;
        hid[h] = v;
    }
}

// DUAL: the side-by-side form (see above); the caller has laid step3b_dual_extra_words more LDS out behind the arrays of the list
// SG: the S-from-memory form (see STEP3B_CARVE_LIST): run-time layout, branch after branch
template <int XF, bool GATHER, int CLS, bool TRAIN = true, bool DUAL = false, bool SG = false>
DEV void net_step3_graph_both(const StepArgs& a, const GraphDims& d_in, int g, int gi, float* scratch, int capN, int capE,
                              int capC, bool late, int cnt_c, int cnt_e1, int cnt_c1) {
    static_assert(XF == 16 || XF == 32 || XF == 48 || XF == 64, "width-specialised kernels only");
    if (CLS == 1) { capN = STEP_CLS_N; capE = STEP_CLS_E; capC = STEP_CLS_C; }
    GraphDims d = d_in;
    const int bC = late ? imin(d.N, capC) : d.C, bE1 = late ? d.E : d.E1, bC1 = late ? imin(d.N, capC) : d.C1;
    const TopoView& tv = a.tv;
    const HeadFused& hf = a.hf;
    constexpr int nb = 2, R = 2 * DRGNN_H2, WREF = 128;
    constexpr int XLD = XF + 4, Z2LD = STEP3_Z2LD, W2NLD = DRGNN_H2 + 4;
    const int F = a.net.n_feat;
    const int O = hf.O;
    static_assert(!SG || (CLS == 0 && !DUAL), "the S-from-memory form has the run-time layout");
    Step3BScratch s = step3b_carve<CLS, SG>(scratch, XF, capN, capE, capC, WREF, O);
    WBlockRegs<1> wreg, wother;
    int* const dummy = (int*)(s.misc + 64);
    const uint32_t done = (uint32_t)a.step2[0];
    const uint32_t tag = done + 1u;

    // ---- prologue ----------------------------------------------------------------------------------------------------------
    PHASE_MARK();
    const int TF = (F + 3) & ~3;
    const float* sgl = a.tiles + (long)d.n0 * TF;
    BurstX<4> bx;
    BurstRowMap<4> brow;
    BurstW<1> bw1[2], bw2[2];
    WaveStage wst;
    const int my_wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    auto stage_job = [&](int w) -> StageJob {
        const int32_t* const* P = tv.p;
        StageJob j = {nullptr, 0, nullptr, 0};
        switch (w) {
        case 0: j = StageJob{P[DRGNN_TI_HMP0] + d.rowbase, bC + 1, s.hmp, 0}; break;
        case 1: j = StageJob{P[DRGNN_TI_MEM1] + d.n0, bC, s.mem1, 0}; break;
        case 2: j = StageJob{P[DRGNN_TI_MPTR1] + d.rowbase, bC1 + 1, s.mp1, 0}; break;
        case 3: j = StageJob{hf.b1, WREF, s.hb1, 0}; break;
        case 4: j = stage_half(StageJob{hf.w2, O * WREF, s.hw2, 0}, 0); break;
        case 5: j = stage_half(StageJob{hf.w2, O * WREF, s.hw2, 0}, 1); break;
        case 6: j = StageJob{hf.b2, O, s.hb2, 0}; break;
        case 7: j = StageJob{P[DRGNN_TI_ROWPTR1] + d.rowbase, bC + 1, s.rp1, 0}; break;
        case 8: j = stage_half(StageJob{P[DRGNN_TI_COL1] + d.e0, bE1, s.cx1, 0}, 0); break;
        case 9: j = stage_half(StageJob{P[DRGNN_TI_COL1] + d.e0, bE1, s.cx1, 0}, 1); break;
        case 10: if (TRAIN) j = StageJob{P[DRGNN_TI_COLPTR1] + d.rowbase, bC + 1, s.cp1, 0}; break;
        case 11: if (TRAIN) j = stage_half(StageJob{P[DRGNN_TI_ROWIDX1] + d.e0, bE1, s.rx1, 0}, 0); break;
        case 12: if (TRAIN) j = stage_half(StageJob{P[DRGNN_TI_ROWIDX1] + d.e0, bE1, s.rx1, 0}, 1); break;
        case 13: if (SG) j = StageJob{P[DRGNN_TI_HORD] + d.n0, d.N, s.hord, 0}; break;
        default: break;
        }
        return j;
    };
    {
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
    if (!SG) {
        burst_load_x(bx, sgl, d.N, TF);
        burst_load_rowmap(brow, bx, tv.p[DRGNN_TI_IHORD] + d.n0, d.N);
    }
#pragma unroll
    for (int br = 0; br < 2; ++br) {
        const drgnn_conv_params& c1 = a.net.conv1[br];
        const drgnn_conv_params& c2 = a.net.conv2[br];
        burst_load_w(bw1[br], c1.w_nbr, c1.nbr_sk, c1.nbr_sh, F, DRGNN_H1);
        burst_load_w(bw2[br], c2.w_nbr, c2.nbr_sk, c2.nbr_sh, DRGNN_H1, DRGNN_H2);
    }
    wstage_load(wst, stage_job(my_wave));
    step_wblock_load(wreg, hf, 0);
    step_wblock_load(wother, hf, 1);
    if (!SG) burst_store_x4_rows(bx, brow, s.G, XLD);
#pragma unroll
    for (int br = 0; br < 2; ++br) {
        burst_store_wt(bw1[br], s.w1t[br], XLD);
        burst_store_wt(bw2[br], s.w2t[br], STEP_XPLD);
        if (TRAIN) burst_store_w(bw2[br], s.w2n[br], W2NLD);
    }
    step_wblock_store(wreg, hf, 0, s.wb);
    wstage_store(wst);
    if (!SG) {
        FOR_TID(e, (step_pad4(d.N) - d.N) * XLD) { s.G[d.N * XLD + e] = 0.0f; }
        if (XF > TF) {
            const int padg = XF - TF;
            FOR_TID(e, d.N * padg) { s.G[(e / padg) * XLD + TF + e % padg] = 0.0f; }
        }
    }
    if (XF > F) {
        const int padc = XF - F;
        FOR_TID(e, DRGNN_H1 * padc) { s.w1t[0][(e / padc) * XLD + F + e % padc] = 0.0f; s.w1t[1][(e / padc) * XLD + F + e % padc] = 0.0f; }
    }
    BARRIER();
    if (late) { d.C = WG_UNIFORM(cnt_c); d.E1 = WG_UNIFORM(cnt_e1); d.C1 = WG_UNIFORM(cnt_c1); }
    if (d.C > capC || d.E1 > d.E || d.C1 > capC) {      // malformed input (flagged by the builder): stay inside LDS, poison
        d.C = imin(d.C, capC); d.E1 = imin(d.E1, d.E); d.C1 = imin(d.C1, capC);
        m_bad |= 1;
    }
    FOR_TID(i, 1) {
        ((int*)s.misc)[STEP_M_BAD] = m_bad;
        ((int*)s.misc)[STEP_M_Y] = m_y;
        s.misc[STEP_M_WY] = m_wy;
        s.misc[STEP_M_DENOM] = m_denom;
    }
    FOR_TID(e, (step_pad4(d.C) - d.C) * STEP_XPLD) { s.xp[d.C * STEP_XPLD + e] = 0.0f; }

    // ---- the side-by-side form: the half workgroup's branch, its lane / wave numbers inside the half, its own Z1 / XP / dS ------
    const int hbr = DUAL ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 9)) : 0;
    const int htid = threadIdx.x & (STEP3D_NT - 1), hwave = my_wave & (STEP3D_NW - 1);
    static_assert(!DUAL || CLS == 1, "the side-by-side form is laid out for the capacity class");
    static_assert((STEP_CLS_N + 4) * DRGNN_H1 <= (STEP_CLS_C + 4) * (2 * STEP_XPLD + STEP3_Z2LD), "branch 1's Z1 fits [S2 | Z2 | dS]");
    float* const dz1 = hbr ? s.u2[1] : s.z1;
    float* const dxp = hbr ? s.end : s.xp;
    float* const dp2 = hbr ? s.end + (((capC + 4) * STEP_XPLD + 3) & ~3) : s.p2;
    float* const dw1t = hbr ? s.w1t[1] : s.w1t[0];
    float* const dw2t = hbr ? s.w2t[1] : s.w2t[0];
    float* const dw2n = hbr ? s.w2n[1] : s.w2n[0];
    float* const du2 = hbr ? s.u2[1] : s.u2[0];
    float* const dz2 = hbr ? s.z2[1] : s.z2[0];
    short* const da0 = hbr ? s.a0[1] : s.a0[0];
    short* const da1 = hbr ? s.a1[1] : s.a1[0];
    if (DUAL) {
        // (branch 1's XP rows past C: zero K padding of its pooled product, as the shared array's above)
        if (hbr) { _Pragma("nounroll") for (int e = htid; e < (step_pad4(d.C) - d.C) * STEP_XPLD; e += STEP3D_NT) dxp[d.C * STEP_XPLD + e] = 0.0f; }
        step3d_gemm_nn<true>(hwave, d.N, 1, XF, s.G, XLD, dw1t, XLD, dz1, DRGNN_H1, dummy);
        BARRIER();
        step3d_cluster_max(htid, d.C, s.hmp, s.mem1, dz1, dxp, da0, STEP3_A1LD(capC));
        BARRIER();
        step3d_gather16<STEP_XPLD>(htid, d.C, s.rp1, s.cx1, dxp, du2);
        _Pragma("nounroll") for (int e = htid; e < (step_pad4(d.C) - d.C) * STEP_XPLD; e += STEP3D_NT) du2[d.C * STEP_XPLD + e] = 0.0f;
        BARRIER();
        step3d_gemm_nn<true>(hwave, d.C, 2, DRGNN_H1, du2, STEP_XPLD, dw2t, STEP_XPLD, dz2, Z2LD, dummy);
        BARRIER();
        step3d_pool_readout<Z2LD>(htid, d.C1, s.mp1, s.mem1, dz2, da1, s.misc, s.xr + hbr * DRGNN_H2,
                                  const_cast<float*>(hf.readout) + (long)g * R + hbr * DRGNN_H2, STEP3_A1LD(capC));
        BARRIER();
    }
    // ---- forward, branch after branch (misc is read in the readout phases: four barriers away) -------------------------------
    // (the per-branch arrays are picked by selects on constant indices: a run-time index into the struct's pointer arrays
    // would put them in scratch memory)
#pragma unroll 1
    for (int br = 0; br < (DUAL ? 0 : 2); ++br) {
        const bool b1 = br != 0;
        float* const w1t = b1 ? s.w1t[1] : s.w1t[0];
        float* const w2t = b1 ? s.w2t[1] : s.w2t[0];
        float* const u2 = b1 ? s.u2[1] : s.u2[0];
        float* const z2 = b1 ? s.z2[1] : s.z2[0];
        short* const a0 = b1 ? s.a0[1] : s.a0[0];
        short* const a1 = b1 ? s.a1[1] : s.a1[0];
        if (SG) step3_conv1_sg<XF>(d.N, sgl, TF, s.hord, w1t, s.z1, dummy);
        else step_gemm_nn<true>(d.N, 1, XF, s.G, XLD, w1t, XLD, s.z1, DRGNN_H1, dummy);
        BARRIER();
        step3_cluster_max(d.C, s.hmp, s.mem1, s.z1, s.xp, a0, STEP3_A1LD(capC));
        BARRIER();
        step_gather_rows<STEP_XPLD, int>(d.C, s.rp1, s.cx1, s.xp, u2);
        FOR_TID(e, (step_pad4(d.C) - d.C) * STEP_XPLD) { u2[d.C * STEP_XPLD + e] = 0.0f; }
        BARRIER();
        step_gemm_nn<true>(d.C, 2, DRGNN_H1, u2, STEP_XPLD, w2t, STEP_XPLD, z2, Z2LD, dummy);
        BARRIER();
        step_pool_readout<Z2LD>(d.C1, s.mp1, s.mem1, z2, a1, s.misc, s.xr + br * DRGNN_H2,
                                const_cast<float*>(hf.readout) + (long)g * R + br * DRGNN_H2, nullptr, nullptr, 0u, STEP3_A1LD(capC));
        BARRIER();
    }

    // ---- FC head + loss + their backward -----------------------------------------------------------------------------------
    const float keep_scale = (hf.p_drop > 0.0f) ? 1.0f / (1.0f - hf.p_drop) : 1.0f;
    const double pt = (double)hf.p_drop * 4294967296.0;
    const uint32_t thresh = (hf.p_drop > 0.0f) ? (uint32_t)(pt > 4294967295.0 ? 4294967295.0 : pt) : 0u;
    float* hp = hf.partials + (long)g * head_compact_floats(R, WREF, O);
    float* p_dhid = hp;
    float* p_hw2 = p_dhid + WREF;
    float* p_hb2 = p_hw2 + (long)O * WREF;
    float* p_loss = p_hb2 + O;
    if (TRAIN && g == 0) { FOR_TID(i, 1) { a.step2[1] = (int32_t)tag; } }     // Adam's step index
    if (TRAIN) { FOR_TID(item, step_pad4(d.C) * Z2LD) { s.z2[0][item] = 0.0f; s.z2[1][item] = 0.0f; } }      // become dZ2 (+ zero K padding)
    step3b_head_fc1<WREF>(hf, g, s.wb, wother, s.hb1, s.xr, s.hid, done, thresh, keep_scale);
    BARRIER();
    step_head_loss<WREF, true>(hf, g, 0, s.hid, s.hw2, s.hb2, s.misc, keep_scale, s.dhid, p_dhid, p_hw2, p_hb2, p_loss);
    if (!TRAIN) return;
    BARRIER();
    // d readout scattered into dZ2 + dW2 (sparse sums: step3_dreadout_dw2), branch after branch
    float* const p_w1n0 = a.partials + ((long)g * nb) * a.n_partial;
    float* const p_w1n1 = p_w1n0 + a.n_partial;
    const long w2off = 2L * F * DRGNN_H1 + DRGNN_H1;
    step3_dreadout_dw2<WREF>(s.wb, s.dhid, s.a1[0], STEP3_A1LD(capC), d.C1, s.u2[0], s.z2[0], p_w1n0 + w2off);
    BARRIER();
    step_wblock_store(wother, hf, 1, s.wb);      // branch 1's column block of fc1 takes branch 0's place
    BARRIER();
    step3_dreadout_dw2<WREF>(s.wb, s.dhid, s.a1[1], STEP3_A1LD(capC), d.C1, s.u2[1], s.z2[1], p_w1n1 + w2off);
    BARRIER();

    if (DUAL) {
        step3d_gemm_nn<false>(hwave, d.C, 1, DRGNN_H2, dz2, Z2LD, dw2n, W2NLD, dp2, STEP_XPLD, dummy);
        BARRIER();
        step3d_gather16<STEP_XPLD>(htid, d.C, s.cp1, s.rx1, dp2, dxp);
        BARRIER();
        step3d_dw1_sparse<XF>(hwave, d.C, da0, STEP3_A1LD(capC), dxp, s.G, hbr ? p_w1n1 : p_w1n0, F);
        return;
    }
    // ---- backward body, branch after branch: dS = dZ2 W2^T, dXP through CSC1 (dense rows), dW1 through the depth-0 argmax ---
#pragma unroll 1
    for (int br = 0; br < 2; ++br) {
        const bool b1 = br != 0;
        float* const w2n = b1 ? s.w2n[1] : s.w2n[0];
        float* const z2 = b1 ? s.z2[1] : s.z2[0];
        short* const a0 = b1 ? s.a0[1] : s.a0[0];
        step_gemm_nn(d.C, 1, DRGNN_H2, z2, Z2LD, w2n, W2NLD, s.p2, STEP_XPLD, dummy);
        BARRIER();
        step3_gather_dxp<STEP_XPLD>(d.C, s.cp1, s.rx1, s.p2, s.xp);
        BARRIER();
        // (the next branch's first two phases leave xp, a0, G alone)
        if (SG) step3_dw1_sparse<XF, true>(d.C, a0, STEP3_A1LD(capC), s.xp, sgl, b1 ? p_w1n1 : p_w1n0, F, s.hord, TF);
        else step3_dw1_sparse<XF>(d.C, a0, STEP3_A1LD(capC), s.xp, s.G, b1 ? p_w1n1 : p_w1n0, F);
    }
}

#endif  // !DRGNN_EMU
#endif
