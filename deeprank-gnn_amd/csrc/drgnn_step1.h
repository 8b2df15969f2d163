// GINet training step with BOTH branches of a graph in ONE workgroup, one after the other.
//
// Why: the two-workgroup step (drgnn_step.h) lets the branch workgroups of a graph wait for each other's readout.  That
// is only sound while every workgroup of the launch is resident (2 B + builder workgroups <= CUs: one workgroup per CU
// at these LDS sizes) -- HIP promises nothing about dispatch order (MI355X_MICROARCH.md, "Workgroup dispatch").  Beyond
// that the step runs HERE: no cross-workgroup wait exists, so placement cannot matter; the x tile, CSR / CSC of both
// levels and the member lists are staged ONCE for both branches (they convolve over the same edge_index: ginet.py:101-128
// clones `data`), the head needs no exchange, and a graph costs one dispatch instead of two -- at batch sizes where the
// device is CU-bound anyway this is also the cheaper schedule per graph.
//
// Same phase routines as the two-workgroup kernel (same arithmetic per branch; the K splits of the two weight-gradient
// products differ, so results agree to rounding, not bit for bit).  Reference: ginet.py:99-141 + its autograd.
//
// LDS plan (SYN: 149 KB): what a branch must keep from its forward pass to its backward pass is small -- the depth-0 /
// depth-1 argmax (a0, a1) and S = A XP -- and exists twice; everything else (u1, z1, xp, z2) is reused by the second
// branch.  fc1's column block is in LDS one branch at a time (the other block waits in registers); the partial tiles of
// the K-split products live in arrays that are dead at that point (dW2: u1; dW1: z1 + z2).
#ifndef DRGNN_STEP1_H
#define DRGNN_STEP1_H

#include "drgnn_step.h"

struct Step1Scratch {
    float* misc; float* xr; float* hid; float* dhid; float* hb1; float* hp0; float* wb;
    float* w1t0; float* w1t1; float* w2t0; float* w2t1; float* w2n0; float* w2n1;
    float* xs;
    int* rp0; int* cx0; int* cp0; int* rx0; int* mp0; int* mem0;
    int* rp1; int* cx1; int* cp1; int* rx1; int* mp1; int* mem1;
    short* a00; short* a01; short* a10; short* a11;      // a<depth><branch>
    float* u1; float* z1; float* z2; float* xp; float* sg0; float* sg1;
    float* u1b; float* z1b; float* z2b; float* xpb;      // PAIRED: branch 1's copies (branch 0 uses u1 / z1 / z2 / xp)
    float* hw2; float* hb2;
    float* end;
};

HD int64_t step1_u1_words(int64_t capN) { const int64_t w = (capN + 4) * DRGNN_H1; return w > 512 ? w : 512; }
// z1 (+ z2 behind it) holds the partial tiles of dW1 = X^T dU1: one 256-word unit per 16-row tile of the F16 input columns at least
HD int64_t step1_z1_words(int64_t capN, int64_t f16) {
    const int64_t w = capN * DRGNN_H1, m = 256 * (f16 / 16);
    return w > m ? w : m;
}

// PAIRED (both branches share every phase): u1 / z2 / xp exist per branch; the two z1 arrays live in the x tile's area, which
// is dead between conv1's products and the end (dW1 reads x from global memory there)
HD int64_t step1_xs_words(int64_t capN, int64_t xld, int paired) {
    // paired: room for the two Z1 arrays and, later, for one 256-word partial tile per 16 input columns and branch at least
    const int64_t w = (capN + 4) * xld, z = 2 * ((capN * DRGNN_H1 + 3) & ~(int64_t)3), m = 2 * 256 * ((xld - 4) / 16);
    if (!paired) return w;
    return w > z ? (w > m ? w : m) : (z > m ? z : m);
}
#define STEP1_CARVE_LIST(X)                                                                    \
    X(misc, 128)                                                                               \
    X(xr, 2 * DRGNN_H2)                                                                        \
    X(hid, H)                                                                                  \
    X(dhid, H)                                                                                 \
    X(hb1, H)                                                                                  \
    X(hp0, H)                                                                                  \
    X(wb, (long)H * STEP_WBLD)                                                                 \
    X(w1t0, DRGNN_H1 * xld)                                                                  \
    X(w1t1, DRGNN_H1 * xld)                                                                  \
    X(w2t0, DRGNN_H2 * STEP_XPLD)                                                            \
    X(w2t1, DRGNN_H2 * STEP_XPLD)                                                            \
    X(w2n0, DRGNN_H1 * (DRGNN_H2 + 4))                                                       \
    X(w2n1, DRGNN_H1 * (DRGNN_H2 + 4))                                                       \
    X(xs, step1_xs_words(capN, xld, paired))                                                   \
    X(rp0, capN + 1)                                                                           \
    X(cx0, capE)                                                                               \
    X(cp0, capN + 1)                                                                           \
    X(rx0, capE)                                                                               \
    X(mp0, capC + 1)                                                                           \
    X(mem0, capN)                                                                              \
    X(rp1, capC + 1)                                                                           \
    X(cx1, capE)                                                                               \
    X(cp1, capC + 1)                                                                           \
    X(rx1, capE)                                                                               \
    X(mp1, capC + 1)                                                                           \
    X(mem1, capC)                                                                              \
    X(a00, ((long)capC * DRGNN_H1 + 1) / 2)                                                  \
    X(a01, ((long)capC * DRGNN_H1 + 1) / 2)                                                  \
    X(a10, ((long)capC * DRGNN_H2 + 1) / 2)                                                  \
    X(a11, ((long)capC * DRGNN_H2 + 1) / 2)                                                  \
    X(u1, step1_u1_words(capN))                                                                \
    X(z1, paired ? 0 : step1_z1_words(capN, f16))                                              \
    X(z2, (long)(capC + 4) * (DRGNN_H2 + 4))                                                   \
    X(xp, (long)(capC + 4) * STEP_XPLD)                                                        \
    X(u1b, paired ? step1_u1_words(capN) : 0)                                                  \
    X(z1b, 0)                                                                                  \
    X(z2b, paired ? (long)(capC + 4) * (DRGNN_H2 + 4) : 0)                                     \
    X(xpb, paired ? (long)(capC + 4) * STEP_XPLD : 0)                                          \
    X(sg0, (long)(capC + 4) * STEP_XPLD)                                                     \
    X(sg1, (long)(capC + 4) * STEP_XPLD)                                                     \
    X(hw2, (long)O * H)                                                                        \
    X(hb2, O)

HD int64_t step1_scratch_words(int64_t F, int64_t capN, int64_t capE, int64_t capC, int64_t H, int64_t O, int paired = 0) {
    const int64_t f16 = step_pad16((int)F), xld = f16 + 4;
    int64_t w = 0;
#define X(name, words) w += (((int64_t)(words) + 3) & ~(int64_t)3);   /* 16-byte aligned arrays */
    STEP1_CARVE_LIST(X)
#undef X
    return w + 16;
}

DEV Step1Scratch step1_carve(float* base, int F, int capN, int capE, int capC, int H, int O, int paired = 0) {
    const int f16 = step_pad16(F), xld = f16 + 4;
    Step1Scratch s;
    int o = 0;
#define X(name, words) { int off = o; STEP_PIN(off); s.name = (decltype(s.name))(base + off); o = off + (int)(((long)(words) + 3) & ~3L); }
    STEP1_CARVE_LIST(X)
#undef X
    s.end = base + o;
    if (paired) {      // the two dZ1 / Z1 arrays inside the x tile's area ((capN + 4) * xld >= 2 * capN * 16 words)
        s.z1 = s.xs;
        s.z1b = s.xs + (((long)capN * DRGNN_H1 + 3) & ~3L);
    }
    return s;
}

// half product of fc1 with ONE branch's readout: out[h] = sum_c wb[h][c] xr[c]   (8 lanes per hidden unit, DPP sum)
template <int HC>
DEV void step1_fc1_half(int Hrt, const float* wb, const float* xr, float* out) {
    const int H = HC ? HC : Hrt;
#ifdef DRGNN_EMU
    for (int h = 0; h < H; ++h) {
        float p = 0.0f;
        for (int c = 0; c < DRGNN_H2; ++c) p = fmaf(wb[h * STEP_WBLD + c], xr[c], p);
        out[h] = p;
    }
#else
    const int items = (H * 8 + 63) & ~63;
    for (int t = threadIdx.x; t < items; t += DRGNN_NTHREADS) {
        const int h = t >> 3, q = t & 7;
        float acc = 0.0f;
        if (h < H) {
            const drgnn_f4 w = *(const drgnn_f4*)(wb + h * STEP_WBLD + 4 * q);
            const drgnn_f4 x = *(const drgnn_f4*)(xr + 4 * q);
            acc = fmaf(w[0], x[0], fmaf(w[1], x[1], fmaf(w[2], x[2], w[3] * x[3])));
        }
        acc = lanes8_sum(acc);
        if (q == 0 && h < H) out[h] = acc;
    }
#endif
}
// hid = dropout(relu(b1 + P0 + P1)): P0 from `p0` (filed after branch 0's readout), P1 formed here from fc1's second column
// block in `wb` and branch 1's readout -- the same sums as the two-workgroup kernel's fc1, in the same order
template <int HC>
DEV void step1_fc1_finish(const HeadFused& hf, int g, const float* wb, const float* b1, const float* xr1, const float* p0,
                          float* hid, uint32_t step, uint32_t thresh, float keep_scale) {
    const int H = HC ? HC : hf.H;
#ifdef DRGNN_EMU
    for (int h = 0; h < H; ++h) {
        float p = 0.0f;
        for (int c = 0; c < DRGNN_H2; ++c) p = fmaf(wb[h * STEP_WBLD + c], xr1[c], p);
        float v = p0[h] + p;
        v += b1[h];
        v = v > 0.0f ? v : 0.0f;
        if (thresh) v = drgnn_keep(hf, step, g, H, h, thresh) ? v * keep_scale : 0.0f;
        hid[h] = v;
    }
#else
    const int items = (H * 8 + 63) & ~63;
    for (int t = threadIdx.x; t < items; t += DRGNN_NTHREADS) {
        const int h = t >> 3, q = t & 7;
        float acc = 0.0f;
        if (h < H) {
            const drgnn_f4 w = *(const drgnn_f4*)(wb + h * STEP_WBLD + 4 * q);
            const drgnn_f4 x = *(const drgnn_f4*)(xr1 + 4 * q);
            acc = fmaf(w[0], x[0], fmaf(w[1], x[1], fmaf(w[2], x[2], w[3] * x[3])));
        }
        acc = lanes8_sum(acc);
        if (q == 0 && h < H) {
            float v = p0[h] + acc;
            v += b1[h];
            v = v > 0.0f ? v : 0.0f;
            if (thresh) v = drgnn_keep(hf, step, g, H, h, thresh) ? v * keep_scale : 0.0f;
            hid[h] = v;
        }
    }
#endif
}

// XF / GATHER / late: as net_step_graph (drgnn_step.h)
// PAIRED: every phase works on BOTH branches (routine of branch 0, routine of branch 1, then the barrier): half the barriers
// and twice the independent work between them -- the form taken whenever its LDS plan fits (153 KB at SYN size); the
// branch-after-branch form (141 KB) is the fallback for larger graphs
// CLS: capacity class of the LDS layout (drgnn_step.h: net_step_graph)
template <int XF, bool GATHER = false, bool PAIRED = false, int CLS = 0>
DEV void net_step_graph_both(const StepArgs& a, const GraphDims& d_in, int g, int gi, float* scratch, int capN, int capE,
                             int capC, bool late = false, int cnt_c = 0, int cnt_e1 = 0, int cnt_c1 = 0) {
    if (CLS == 1) { capN = STEP_CLS_N; capE = STEP_CLS_E; capC = STEP_CLS_C; }
    GraphDims d = d_in;
    const int bC = late ? imin(d.N, capC) : d.C, bE1 = late ? d.E : d.E1, bC1 = late ? imin(d.N, capC) : d.C1;
#ifdef DRGNN_EMU
    if (late) { d.C = imin(cnt_c, capC); d.E1 = imin(cnt_e1, d.E); d.C1 = imin(cnt_c1, capC); }
#endif
    constexpr int KIND = DRGNN_GINET;
    constexpr int HC1 = DRGNN_H1;
    constexpr int R = 2 * DRGNN_H2;
    constexpr int WREF = 128;                          // ginet.py:136
    constexpr int W2NLD = DRGNN_H2 + 4, Z2LD = DRGNN_H2 + 4;
    typedef int EIdx;
    const TopoView& tv = a.tv;
    const HeadFused& hf = a.hf;
    const int F = a.net.n_feat;
    const int H = hf.H, O = hf.O;
    const int F16 = XF ? XF : step_pad16(F), XLD = F16 + 4;
    Step1Scratch s = step1_carve(scratch, (XF != 0) ? XF : F, capN, capE, capC, (XF != 0) ? WREF : H, O, PAIRED ? 1 : 0);
    WBlockRegs<(XF != 0) ? 1 : STEP_WB_J> wreg0, wreg1;     // fc1's two column blocks: in LDS one at a time
    int* const dummy = (int*)(s.misc + 64);
    const uint32_t done = (uint32_t)a.step2[0];
    const uint32_t tag = done + 1u;
    const float* b1 = s.hb1;
    const float* w2 = s.hw2;
    const float* b2 = s.hb2;

    // ---- staging: once for both branches ---------------------------------------------------------
    const float* xg = a.x + (long)d.n0 * F;
    const bool burst = (XF != 0) ? true
                                 : (net_burst_ok(xg, F, d.N, d.E, late ? capC : d.C) && O * H <= 2 * DRGNN_BCAP && H * 8 <= STEP_WB_J * DRGNN_BCAP);
    if (late && !burst) { d.C = WG_UNIFORM(cnt_c); d.E1 = WG_UNIFORM(cnt_e1); d.C1 = WG_UNIFORM(cnt_c1); }
    BurstX<4> bx;
    BurstW<1> bw10, bw11, bw20, bw21;
    WaveStage wst;
#ifndef DRGNN_EMU
    const int my_wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#endif
    auto stage_job = [&](int burst_no, int w) -> StageJob {
        const int32_t* const* P = tv.p;
        StageJob j = {nullptr, 0, nullptr, 0};
        switch (burst_no * 16 + w) {
        case 16 + 0: j = StageJob{P[DRGNN_TI_ROWPTR0] + d.rowbase, d.N + 1, s.rp0, 0}; break;
        case 16 + 1: j = stage_half(StageJob{P[DRGNN_TI_COL0] + d.e0, d.E, s.cx0, 0}, 0); break;
        case 16 + 2: j = stage_half(StageJob{P[DRGNN_TI_COL0] + d.e0, d.E, s.cx0, 0}, 1); break;
        case 16 + 3: j = StageJob{P[DRGNN_TI_MPTR0] + d.rowbase, bC + 1, s.mp0, 0}; break;
        case 16 + 4: j = StageJob{P[DRGNN_TI_MEM0] + d.n0, d.N, s.mem0, 0}; break;
        case 32 + 0: j = StageJob{P[DRGNN_TI_ROWPTR1] + d.rowbase, bC + 1, s.rp1, 0}; break;
        case 32 + 1: j = stage_half(StageJob{P[DRGNN_TI_COL1] + d.e0, bE1, s.cx1, 0}, 0); break;
        case 32 + 2: j = stage_half(StageJob{P[DRGNN_TI_COL1] + d.e0, bE1, s.cx1, 0}, 1); break;
        case 32 + 3: j = StageJob{P[DRGNN_TI_MPTR1] + d.rowbase, bC1 + 1, s.mp1, 0}; break;
        case 32 + 4: j = StageJob{P[DRGNN_TI_MEM1] + d.n0, bC, s.mem1, 0}; break;
        case 32 + 5: j = StageJob{hf.b1, H, s.hb1, 0}; break;
        case 32 + 6: j = stage_half(StageJob{hf.w2, O * H, s.hw2, 0}, 0); break;
        case 32 + 7: j = stage_half(StageJob{hf.w2, O * H, s.hw2, 0}, 1); break;
        case 32 + 8: j = StageJob{hf.b2, O, s.hb2, 0}; break;
        case 32 + 9: j = StageJob{P[DRGNN_TI_COLPTR0] + d.rowbase, d.N + 1, s.cp0, 0}; break;
        case 32 + 10: j = stage_half(StageJob{P[DRGNN_TI_ROWIDX0] + d.e0, d.E, s.rx0, 0}, 0); break;
        case 32 + 11: j = stage_half(StageJob{P[DRGNN_TI_ROWIDX0] + d.e0, d.E, s.rx0, 0}, 1); break;
        case 32 + 12: j = StageJob{P[DRGNN_TI_COLPTR1] + d.rowbase, bC + 1, s.cp1, 0}; break;
        case 32 + 13: j = stage_half(StageJob{P[DRGNN_TI_ROWIDX1] + d.e0, bE1, s.rx1, 0}, 0); break;
        case 32 + 14: j = stage_half(StageJob{P[DRGNN_TI_ROWIDX1] + d.e0, bE1, s.rx1, 0}, 1); break;
        default: break;
        }
        return j;
    };
#ifdef DRGNN_EMU
    auto stage_request = [&](int burst_no) { (void)burst_no; };
    auto stage_file = [&](int burst_no) { for (int w = 0; w < 16; ++w) stage_copy(stage_job(burst_no, w)); };
#else
    auto stage_request = [&](int burst_no) { wstage_load(wst, stage_job(burst_no, my_wave)); };
    auto stage_file = [&](int burst_no) { (void)burst_no; wstage_store(wst); };
#endif
    // per-graph scalars of the readout / loss phases (as in net_step_graph)
    int m_bad = 0, m_y = 0;
    float m_wy = 1.0f, m_denom = 1.0f;
#ifndef DRGNN_EMU
    if (my_wave == 0)
#endif
    {
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
            m_y = __builtin_amdgcn_readfirstlane(m_y);
            m_wy = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(m_wy)));
            m_bad = __builtin_amdgcn_readfirstlane(m_bad);
#endif
        }
    }
    if (burst) {
        burst_load_x(bx, xg, d.N, F);
        burst_load_w(bw10, a.net.conv1[0].w_nbr, a.net.conv1[0].nbr_sk, a.net.conv1[0].nbr_sh, F, DRGNN_H1);
        burst_load_w(bw11, a.net.conv1[1].w_nbr, a.net.conv1[1].nbr_sk, a.net.conv1[1].nbr_sh, F, DRGNN_H1);
        stage_request(1);
        burst_store_x4(bx, s.xs, XLD);
        burst_store_wt(bw10, s.w1t0, XLD);
        burst_store_wt(bw11, s.w1t1, XLD);
    } else {
        FOR_TID(e, d.N * F) { s.xs[(e / F) * XLD + e % F] = xg[e]; }
        for (int br = 0; br < 2; ++br) {
            const drgnn_conv_params& c1 = a.net.conv1[br];
            const drgnn_conv_params& c2 = a.net.conv2[br];
            step_stage_wt((br ? s.w1t1 : s.w1t0), XLD, c1.w_nbr, c1.nbr_sk, c1.nbr_sh, F, DRGNN_H1);
            step_stage_wt((br ? s.w2t1 : s.w2t0), STEP_XPLD, c2.w_nbr, c2.nbr_sk, c2.nbr_sh, DRGNN_H1, DRGNN_H2);
            stage_weight((br ? s.w2n1 : s.w2n0), W2NLD, c2.w_nbr, c2.nbr_sk, c2.nbr_sh, DRGNN_H1, DRGNN_H2);
        }
        step_wblock_load(wreg0, hf, 0);
        step_wblock_load(wreg1, hf, 1);
        step_copy_i32(s.rp0, tv.p[DRGNN_TI_ROWPTR0] + d.rowbase, d.N + 1);
        step_copy_i32(s.cx0, tv.p[DRGNN_TI_COL0] + d.e0, d.E);
        step_copy_i32(s.cp0, tv.p[DRGNN_TI_COLPTR0] + d.rowbase, d.N + 1);
        step_copy_i32(s.rx0, tv.p[DRGNN_TI_ROWIDX0] + d.e0, d.E);
        step_copy_i32(s.mp0, tv.p[DRGNN_TI_MPTR0] + d.rowbase, d.C + 1);
        step_copy_i32(s.mem0, tv.p[DRGNN_TI_MEM0] + d.n0, d.N);
        step_copy_i32(s.rp1, tv.p[DRGNN_TI_ROWPTR1] + d.rowbase, d.C + 1);
        step_copy_i32(s.cx1, tv.p[DRGNN_TI_COL1] + d.e0, d.E1);
        step_copy_i32(s.cp1, tv.p[DRGNN_TI_COLPTR1] + d.rowbase, d.C + 1);
        step_copy_i32(s.rx1, tv.p[DRGNN_TI_ROWIDX1] + d.e0, d.E1);
        step_copy_i32(s.mp1, tv.p[DRGNN_TI_MPTR1] + d.rowbase, d.C1 + 1);
        step_copy_i32(s.mem1, tv.p[DRGNN_TI_MEM1] + d.n0, d.C);
        step_copy_f32(s.hb1, hf.b1, H);
        step_copy_f32(s.hw2, hf.w2, O * H);
        step_copy_f32(s.hb2, hf.b2, O);
    }
    FOR_TID(e, (step_pad4(d.N) - d.N) * XLD) { s.xs[d.N * XLD + e] = 0.0f; }
    if (F16 > F) {
        const int padc = F16 - F;
        FOR_TID(e, d.N * padc) { s.xs[(e / padc) * XLD + F + e % padc] = 0.0f; }
        FOR_TID(e, DRGNN_H1 * padc) {
            s.w1t0[(e / padc) * XLD + F + e % padc] = 0.0f;
            s.w1t1[(e / padc) * XLD + F + e % padc] = 0.0f;
        }
    }
    BARRIER();
    if (late) {
        d.C = WG_UNIFORM(cnt_c); d.E1 = WG_UNIFORM(cnt_e1); d.C1 = WG_UNIFORM(cnt_c1);
        if (d.C > capC || d.E1 > d.E || d.C1 > capC) {
            d.C = imin(d.C, capC); d.E1 = imin(d.E1, d.E); d.C1 = imin(d.C1, capC);
            m_bad |= 1;
        }
    }

    const float keep_scale = (hf.p_drop > 0.0f) ? 1.0f / (1.0f - hf.p_drop) : 1.0f;
    const double pt = (double)hf.p_drop * 4294967296.0;
    const uint32_t thresh = (hf.p_drop > 0.0f) ? (uint32_t)(pt > 4294967295.0 ? 4294967295.0 : pt) : 0u;
    float* hp = hf.partials + (long)g * head_compact_floats(R, H, O);
    float* p_dhid = hp;
    float* p_hw2 = p_dhid + H;
    float* p_hb2 = p_hw2 + (long)O * H;
    float* p_loss = p_hb2 + O;

    if (PAIRED) {
        // ================= both branches in every phase =================================================
        if (burst) {
            stage_file(1);
            burst_load_w(bw20, a.net.conv2[0].w_nbr, a.net.conv2[0].nbr_sk, a.net.conv2[0].nbr_sh, DRGNN_H1, DRGNN_H2);
            burst_load_w(bw21, a.net.conv2[1].w_nbr, a.net.conv2[1].nbr_sk, a.net.conv2[1].nbr_sh, DRGNN_H1, DRGNN_H2);
            step_wblock_load(wreg0, hf, 0);
            step_wblock_load(wreg1, hf, 1);
            stage_request(2);
        }
        step_gemm_nn(d.N, 1, F16, s.xs, XLD, s.w1t0, XLD, s.u1, HC1, dummy);
        step_gemm_nn(d.N, 1, F16, s.xs, XLD, s.w1t1, XLD, s.u1b, HC1, dummy, nullptr, nullptr, (d.N + 15) >> 4);
        FOR_TID(e, (step_pad4(d.C) - d.C) * STEP_XPLD) { s.xp[d.C * STEP_XPLD + e] = 0.0f; s.xpb[d.C * STEP_XPLD + e] = 0.0f; }
        FOR_TID(i, 1) {
#ifdef DRGNN_EMU
            memcpy(&s.misc[STEP_M_BAD], &m_bad, 4);
            memcpy(&s.misc[STEP_M_Y], &m_y, 4);
#else
            ((int*)s.misc)[STEP_M_BAD] = m_bad;
            ((int*)s.misc)[STEP_M_Y] = m_y;
#endif
            s.misc[STEP_M_WY] = m_wy;
            s.misc[STEP_M_DENOM] = m_denom;
        }
        BARRIER();
        // (the x tile is dead from here on: Z1 of both branches takes its place)
        net_aggregate<KIND, DRGNN_H1, true, 0, EIdx, true>(d.N, s.rp0, (const EIdx*)s.cx0, nullptr, nullptr, nullptr, s.u1, nullptr, s.z1);
        net_aggregate<KIND, DRGNN_H1, true, 0, EIdx, true>(d.N, s.rp0, (const EIdx*)s.cx0, nullptr, nullptr, nullptr, s.u1b, nullptr, s.z1b);
        if (burst) {
            burst_store_wt(bw20, s.w2t0, STEP_XPLD);
            burst_store_w(bw20, s.w2n0, W2NLD);
            burst_store_wt(bw21, s.w2t1, STEP_XPLD);
            burst_store_w(bw21, s.w2n1, W2NLD);
            stage_file(2);
        }
        step_wblock_store(wreg0, hf, 0, s.wb);      // fc1's block of branch 0 (block 1 follows once block 0 has served)
        BARRIER();
        net_cluster_max<DRGNN_H1, STEP_XPLD, short>(d.C, s.mp0, s.mem0, s.z1, s.xp, nullptr, s.a00);
        net_cluster_max<DRGNN_H1, STEP_XPLD, short>(d.C, s.mp0, s.mem0, s.z1b, s.xpb, nullptr, s.a01);
        BARRIER();
        step_gather_rows<STEP_XPLD, EIdx>(d.C, s.rp1, (const EIdx*)s.cx1, s.xp, s.sg0);
        step_gather_rows<STEP_XPLD, EIdx>(d.C, s.rp1, (const EIdx*)s.cx1, s.xpb, s.sg1);
        FOR_TID(e, (step_pad4(d.C) - d.C) * STEP_XPLD) { s.sg0[d.C * STEP_XPLD + e] = 0.0f; s.sg1[d.C * STEP_XPLD + e] = 0.0f; }
        FOR_TID(item, d.N * DRGNN_H1) { s.z1[item] = 0.0f; s.z1b[item] = 0.0f; }      // Z1 is consumed: becomes dZ1
        BARRIER();
        step_gemm_nn<true>(d.C, 2, DRGNN_H1, s.sg0, STEP_XPLD, s.w2t0, STEP_XPLD, s.z2, Z2LD, dummy);
        step_gemm_nn<true>(d.C, 2, DRGNN_H1, s.sg1, STEP_XPLD, s.w2t1, STEP_XPLD, s.z2b, Z2LD, dummy, nullptr, nullptr, 2 * ((d.C + 15) >> 4));
        BARRIER();
        step_pool_readout<Z2LD>(d.C1, s.mp1, s.mem1, s.z2, s.a10, s.misc, s.xr, const_cast<float*>(hf.readout) + (long)g * R);
        step_pool_readout<Z2LD>(d.C1, s.mp1, s.mem1, s.z2b, s.a11, s.misc, s.xr + DRGNN_H2,
                                const_cast<float*>(hf.readout) + (long)g * R + DRGNN_H2);
        BARRIER();
        // ---- head: fc1's half with block 0, block 1 into LDS, the other half + hid, loss, d readout of both branches
        if (hf.train && g == 0) { FOR_TID(i, 1) { a.step2[1] = (int32_t)tag; } }     // Adam's step index
        FOR_TID(item, step_pad4(d.C) * Z2LD) { s.z2[item] = 0.0f; s.z2b[item] = 0.0f; }      // Z2 -> dZ2 (+ zero K padding)
        if (XF != 0 || hf.H == WREF) step1_fc1_half<WREF>(H, s.wb, s.xr, s.hp0);
        else step1_fc1_half<0>(H, s.wb, s.xr, s.hp0);
        BARRIER();
        step_wblock_store(wreg1, hf, 1, s.wb);
        BARRIER();
        if (XF != 0 || hf.H == WREF) step1_fc1_finish<WREF>(hf, g, s.wb, b1, s.xr + DRGNN_H2, s.hp0, s.hid, done, thresh, keep_scale);
        else step1_fc1_finish<0>(hf, g, s.wb, b1, s.xr + DRGNN_H2, s.hp0, s.hid, done, thresh, keep_scale);
        BARRIER();
        step_head_loss<WREF, (XF != 0)>(hf, g, 0, s.hid, w2, b2, s.misc, keep_scale, s.dhid, p_dhid, p_hw2, p_hb2, p_loss);
        if (!hf.train) return;
        BARRIER();
        step_head_dreadout<WREF, (XF != 0)>(hf, s.wb, s.dhid, s.a11, d.C1, s.z2b, Z2LD);      // block 1 is in LDS
        BARRIER();
        step_wblock_store(wreg0, hf, 0, s.wb);
        BARRIER();
        step_head_dreadout<WREF, (XF != 0)>(hf, s.wb, s.dhid, s.a10, d.C1, s.z2, Z2LD);
        BARRIER();
        // ---- backward body, both branches per phase ---------------------------------------------------
        float* part0 = a.partials + ((long)g * 2 + 0) * a.n_partial;
        float* part1 = a.partials + ((long)g * 2 + 1) * a.n_partial;
        const long o_w2n = 2L * F * DRGNN_H1 + DRGNN_H1;
        const int u1_units = (int)(step1_u1_words(capN) / 256);
        const int KS2 = imin(DRGNN_NWAVES / 4, u1_units / 2);        // two products of 2 tiles each share the 16 waves
        // dS = dZ2 W2^T (into the xp areas);  dW2 = S^T dZ2: partial tiles (in u1 / u1b) here, their sum behind the barrier
        step_gemm_nn(d.C, 1, DRGNN_H2, s.z2, Z2LD, s.w2n0, W2NLD, s.xp, STEP_XPLD, dummy);
        step_gemm_nn(d.C, 1, DRGNN_H2, s.z2b, Z2LD, s.w2n1, W2NLD, s.xpb, STEP_XPLD, dummy, nullptr, nullptr, (d.C + 15) >> 4);
        step_gemm_tn(1, 2, d.C, s.sg0, STEP_XPLD, s.z2, Z2LD, KS2, s.u1, part0 + o_w2n, DRGNN_H2, DRGNN_H1, 1);
        step_gemm_tn(1, 2, d.C, s.sg1, STEP_XPLD, s.z2b, Z2LD, KS2, s.u1b, part1 + o_w2n, DRGNN_H2, DRGNN_H1, 1, DRGNN_NWAVES / 2);
        BARRIER();
        step_gemm_tn(1, 2, d.C, s.sg0, STEP_XPLD, s.z2, Z2LD, KS2, s.u1, part0 + o_w2n, DRGNN_H2, DRGNN_H1, 2);
        step_gemm_tn(1, 2, d.C, s.sg1, STEP_XPLD, s.z2b, Z2LD, KS2, s.u1b, part1 + o_w2n, DRGNN_H2, DRGNN_H1, 2);
        step_gather_scatter<STEP_XPLD, EIdx>(d.C, s.cp1, (const EIdx*)s.rx1, s.xp, s.a00, s.z1);
        step_gather_scatter<STEP_XPLD, EIdx>(d.C, s.cp1, (const EIdx*)s.rx1, s.xpb, s.a01, s.z1b);
        BARRIER();
        // (the partial tiles are summed: u1 / u1b take dU1; their K padding rows are zeroed in the same pass)
        net_aggregate_bwd<KIND, DRGNN_H1, true, 0, EIdx>(d.N, s.rp0, s.cp0, (const EIdx*)s.rx0, nullptr, nullptr, nullptr, nullptr, s.z1, s.u1);
        net_aggregate_bwd<KIND, DRGNN_H1, true, 0, EIdx>(d.N, s.rp0, s.cp0, (const EIdx*)s.rx0, nullptr, nullptr, nullptr, nullptr, s.z1b, s.u1b);
        FOR_TID(e, (step_pad4(d.N) - d.N) * HC1) { s.u1[d.N * HC1 + e] = 0.0f; s.u1b[d.N * HC1 + e] = 0.0f; }
        BARRIER();
        {   // dW1 = X^T dU1, X read from global memory (its LDS tile made room for Z1); partial tiles in the dead Z1 area
            const int mtiles = F16 >> 4;
            const int z_units = (int)(step1_xs_words(capN, XLD, 1) / 256 / 2);      // per branch
            int KS = imin(imax(DRGNN_NWAVES / (2 * mtiles), 1), z_units / mtiles);
            if (KS < 1) KS = 1;
            float* pt0 = s.xs;
            float* pt1 = s.xs + (long)z_units * 256;
            step_gemm_tn_bufa(mtiles, d.N, xg, F, d.N * F * 4, s.u1, HC1, KS, pt0, part0, DRGNN_H1, F, 1);
            step_gemm_tn_bufa(mtiles, d.N, xg, F, d.N * F * 4, s.u1b, HC1, KS, pt1, part1, DRGNN_H1, F, 1, DRGNN_NWAVES / 2);
            BARRIER();
            step_gemm_tn_bufa(mtiles, d.N, xg, F, d.N * F * 4, s.u1, HC1, KS, pt0, part0, DRGNN_H1, F, 2);
            step_gemm_tn_bufa(mtiles, d.N, xg, F, d.N * F * 4, s.u1b, HC1, KS, pt1, part1, DRGNN_H1, F, 2);
        }
        return;
    }

    // ---- forward, branch 0 then branch 1 -----------------------------------------------------------
    for (int br = 0; br < 2; ++br) {
        if (br == 0 && burst) {
            stage_file(1);
            burst_load_w(bw20, a.net.conv2[0].w_nbr, a.net.conv2[0].nbr_sk, a.net.conv2[0].nbr_sh, DRGNN_H1, DRGNN_H2);
            burst_load_w(bw21, a.net.conv2[1].w_nbr, a.net.conv2[1].nbr_sk, a.net.conv2[1].nbr_sh, DRGNN_H1, DRGNN_H2);
            step_wblock_load(wreg0, hf, 0);
            step_wblock_load(wreg1, hf, 1);
            stage_request(2);
        }
        if (br == 1) {      // fc1's half product with branch 0's readout, while the block of branch 0 is in LDS
            if (XF != 0 || hf.H == WREF) step1_fc1_half<WREF>(H, s.wb, s.xr, s.hp0);
            else step1_fc1_half<0>(H, s.wb, s.xr, s.hp0);
        }
        step_gemm_nn(d.N, 1, F16, s.xs, XLD, (br ? s.w1t1 : s.w1t0), XLD, s.u1, HC1, dummy);
        FOR_TID(e, (step_pad4(d.C) - d.C) * STEP_XPLD) { s.xp[d.C * STEP_XPLD + e] = 0.0f; }
        if (br == 0) {
            FOR_TID(i, 1) {
#ifdef DRGNN_EMU
                memcpy(&s.misc[STEP_M_BAD], &m_bad, 4);
                memcpy(&s.misc[STEP_M_Y], &m_y, 4);
#else
                ((int*)s.misc)[STEP_M_BAD] = m_bad;
                ((int*)s.misc)[STEP_M_Y] = m_y;
#endif
                s.misc[STEP_M_WY] = m_wy;
                s.misc[STEP_M_DENOM] = m_denom;
            }
        }
        BARRIER();
        net_aggregate<KIND, DRGNN_H1, true, 0, EIdx, true>(d.N, s.rp0, (const EIdx*)s.cx0, nullptr, nullptr, nullptr, s.u1, nullptr, s.z1);
        if (br == 0 && burst) {
            burst_store_wt(bw20, s.w2t0, STEP_XPLD);
            burst_store_w(bw20, s.w2n0, W2NLD);
            burst_store_wt(bw21, s.w2t1, STEP_XPLD);
            burst_store_w(bw21, s.w2n1, W2NLD);
            stage_file(2);
        }
        // fc1's column block of THIS branch into LDS (branch 0: first time; branch 1: block 0 has served above)
        if (br == 0) step_wblock_store(wreg0, hf, 0, s.wb);
        else step_wblock_store(wreg1, hf, 1, s.wb);
        BARRIER();
        net_cluster_max<DRGNN_H1, STEP_XPLD, short>(d.C, s.mp0, s.mem0, s.z1, s.xp, nullptr, (br ? s.a01 : s.a00));
        BARRIER();
        step_gather_rows<STEP_XPLD, EIdx>(d.C, s.rp1, (const EIdx*)s.cx1, s.xp, (br ? s.sg1 : s.sg0));
        FOR_TID(e, (step_pad4(d.C) - d.C) * STEP_XPLD) { (br ? s.sg1 : s.sg0)[d.C * STEP_XPLD + e] = 0.0f; }
        FOR_TID(item, d.N * DRGNN_H1) { s.z1[item] = 0.0f; }      // Z1 is consumed: becomes dZ1
        BARRIER();
        step_gemm_nn<true>(d.C, 2, DRGNN_H1, (br ? s.sg1 : s.sg0), STEP_XPLD, (br ? s.w2t1 : s.w2t0), STEP_XPLD, s.z2, Z2LD, dummy);
        BARRIER();
        step_pool_readout<Z2LD>(d.C1, s.mp1, s.mem1, s.z2, (br ? s.a11 : s.a10), s.misc, s.xr + br * DRGNN_H2,
                                const_cast<float*>(hf.readout) + (long)g * R + br * DRGNN_H2);
        BARRIER();
    }

    // ---- FC head + loss: no exchange, both readouts are here ---------------------------------------
    if (hf.train && g == 0) { FOR_TID(i, 1) { a.step2[1] = (int32_t)tag; } }     // Adam's step index
    FOR_TID(item, step_pad4(d.C) * Z2LD) { s.z2[item] = 0.0f; }      // Z2 is consumed: becomes dZ2 (+ zero K padding)
    if (XF != 0 || hf.H == WREF) step1_fc1_finish<WREF>(hf, g, s.wb, b1, s.xr + DRGNN_H2, s.hp0, s.hid, done, thresh, keep_scale);
    else step1_fc1_finish<0>(hf, g, s.wb, b1, s.xr + DRGNN_H2, s.hp0, s.hid, done, thresh, keep_scale);
    BARRIER();
    // loss, its gradient, dhid (LDS: both branches' d readout read it) and the head slab
    step_head_loss<WREF, (XF != 0)>(hf, g, 0, s.hid, w2, b2, s.misc, keep_scale, s.dhid, p_dhid, p_hw2, p_hb2, p_loss);
    if (!hf.train) return;
    BARRIER();
    // d readout of branch 1 (whose column block of fc1 is the one in LDS), scattered into dZ2
    step_head_dreadout<WREF, (XF != 0)>(hf, s.wb, s.dhid, s.a11, d.C1, s.z2, Z2LD);
    BARRIER();

    // ---- backward, branch 1 then branch 0 ----------------------------------------------------------
    for (int pass = 0; pass < 2; ++pass) {
        const int br = 1 - pass;
        float* part_w = a.partials + ((long)g * 2 + br) * a.n_partial;
        float* p_w1n = part_w;
        float* p_w2n = part_w + 2L * F * DRGNN_H1 + DRGNN_H1;
        const int u1_units = (int)(step1_u1_words(capN) / 256);
        const int z_units = (int)((step1_z1_words(capN, F16) + (long)(capC + 4) * Z2LD) / 256);
        if (pass == 1) {
            // turnaround: dZ1 / dZ2 zeroed again (they held branch 1's gradients and partial tiles), fc1's block 0 back in
            // LDS, then branch 0's d readout
            FOR_TID(item, d.N * DRGNN_H1) { s.z1[item] = 0.0f; }
            FOR_TID(item, step_pad4(d.C) * Z2LD) { s.z2[item] = 0.0f; }
            step_wblock_store(wreg0, hf, 0, s.wb);
            BARRIER();
            step_head_dreadout<WREF, (XF != 0)>(hf, s.wb, s.dhid, s.a10, d.C1, s.z2, Z2LD);
            BARRIER();
        }
        // dS = dZ2 W2^T (into the xp area, rows of STEP_XPLD floats);  dW2 = S^T dZ2 (K = pooled nodes; partial tiles in u1)
        step_gemm_nn(d.C, 1, DRGNN_H2, s.z2, Z2LD, (br ? s.w2n1 : s.w2n0), W2NLD, s.xp, STEP_XPLD, dummy);
        step_gemm_tn(1, 2, d.C, (br ? s.sg1 : s.sg0), STEP_XPLD, s.z2, Z2LD, imin(DRGNN_NWAVES / 2, u1_units / 2), s.u1, p_w2n, DRGNN_H2, DRGNN_H1);
        BARRIER();      // (u1's K padding rows are zeroed below, behind the sum of the partial tiles the product kept there)
        // dXP = A^T dS, scattered through the depth-0 argmax into dZ1; u1's K padding rows (partial tiles were there) zero again
        step_gather_scatter<STEP_XPLD, EIdx>(d.C, s.cp1, (const EIdx*)s.rx1, s.xp, (br ? s.a01 : s.a00), s.z1);
        FOR_TID(e, (step_pad4(d.N) - d.N) * HC1) { s.u1[d.N * HC1 + e] = 0.0f; }
        BARRIER();
        net_aggregate_bwd<KIND, DRGNN_H1, true, 0, EIdx>(d.N, s.rp0, s.cp0, (const EIdx*)s.rx0, nullptr, nullptr, nullptr, nullptr, s.z1, s.u1);
        BARRIER();
        {   // dW1 = X^T dU1: K = nodes of the graph, split in slices over the waves (partial tiles in z1 + z2)
            const int mtiles = F16 >> 4;
            int KS = imin(DRGNN_NWAVES / mtiles, z_units / mtiles);
            if (KS < 1) KS = 1;
            step_gemm_tn(mtiles, 1, d.N, s.xs, XLD, s.u1, HC1, KS, s.z1, p_w1n, DRGNN_H1, F);
        }
        if (pass == 0) BARRIER();
    }
}

#endif
