// drgnn_head.h -- the dense head of the three nets plus loss, fused forward+backward, and the
// optimiser update: what the reference trainer runs around the message-passing body for
// every mini-batch (NeuralNet.py:489-506: model(...) tail, loss, loss.backward(), Adam step).
//
//   head   (ginet.py:136-139; sGAT.py:134-135; foutnet.py:121-122)
//          hid  = dropout(relu(readout W1^T + b1), p)        W1 [H,R]  (R = 64 | 32, H = 128 | 64)
//          pred = hid W2^T + b2                                W2 [O,H]
//   loss   MSE on pred.reshape(-1) (regression) or (weighted) cross entropy (classification),
//          mean reduction                                      (NeuralNet.py:239-263)
//   update Adam, torch defaults betas (0.9, 0.999), eps 1e-8, no weight decay
//          (NeuralNet.py:183-184)
//
// One workgroup handles a tile of 64 graphs: readout tile and W1 live in LDS (padded rows),
// the two [64 x H] products run on the f32 MFMA, fc2 / loss are a few hundred FMAs on VALU.
// Weight-gradient contributions go to a per-workgroup partial slab
//   [dW1 H*R][db1 H][dW2 O*H][db2 O][loss_sum][weight_sum]
// which the reducer sums in fixed order (deterministic).
#pragma once
#include "drgnn_net.h"

#define DRGNN_HEAD_TILE_SMALL 16     // graphs per workgroup for small batches (latency)
#define DRGNN_HEAD_TILE_LARGE 64     // ... for large batches (fewer partial slabs)
#define DRGNN_HEAD_TMP 4096          // LDS floats for split sums / K-split GEMM partials

HD int head_tile(int64_t n_graphs) { return n_graphs <= 512 ? DRGNN_HEAD_TILE_SMALL : DRGNN_HEAD_TILE_LARGE; }

struct HeadArgs {
    const float* readout;     // [B, R]
    const float* y_reg;       // [B]            (regression)
    const int64_t* y_cls;     // [B] class index (classification)
    const float* class_w;     // [O] or null
    const float* w1; const float* b1;   // [H,R], [H]
    const float* w2; const float* b2;   // [O,H], [O]
    const int32_t* step;      // device step counter (dropout stream id)
    float* pred;              // [B, O]
    float* grad_readout;      // [B, R] or null (inference)
    float* partials;          // [n_wg][P] or null
    int B, R, H, O, T;        // T = graphs per workgroup
    int task;
    int train;                // apply dropout, compute gradients
    int sigmoid;              // regression: pred = sigmoid(output) before the loss
    float p_drop;
    uint32_t seed;
};

HD int64_t head_lds_words(int R, int H, int O, int T) {
    return (int64_t)T * (R + 1) + (int64_t)H * (R + 1) + (int64_t)T * (H + 1) + H + (int64_t)O * H + O +
           2 * T * DRGNN_MAX_OUT + 2 * T + DRGNN_HEAD_TMP + 64;
}

DEV void head_block(const HeadArgs& a, int blk, float* lds) {
    const int R = a.R, H = a.H, O = a.O, T = a.T;
    const int g0 = blk * T;
    const int G = imin(T, a.B - g0);
    const int ldx = R + 1, ldh = H + 1;
    float* xs = lds;                                   // [T][R+1]
    float* w1p = xs + T * ldx;                         // [H][R+1]
    float* hid = w1p + (long)H * ldx;                  // [T][H+1]
    float* b1s = hid + T * ldh;                        // [H]
    float* w2s = b1s + H;                              // [O][H]
    float* b2s = w2s + (long)O * H;                    // [O]
    float* outs = b2s + O;                             // [T][MAX_OUT]   pred tile
    float* douts = outs + T * DRGNN_MAX_OUT;           // [T][MAX_OUT]   d loss / d pred
    float* red = douts + T * DRGNN_MAX_OUT;            // [T][2] per-graph (loss, weight)
    float* tmp = red + 2 * T;                          // [DRGNN_HEAD_TMP]
    const uint32_t step = a.step ? (uint32_t)a.step[0] : 0u;
    const FastDiv dH = fastdiv_make(H), dO = fastdiv_make(O), dR = fastdiv_make(R);
    PHASE_MARK();

    // ---- stage (all loads in flight together when the sizes allow) ------------------------
    if (H * R <= 8 * DRGNN_BCAP && T * R <= 4 * DRGNN_BCAP && O * H <= 2 * DRGNN_BCAP &&
        H <= DRGNN_BCAP) {
        BurstW<8> bw1;  burst_load_w(bw1, a.w1, R, 1, H, R);
        BurstW<4> bx;   burst_load_w(bx, a.readout + (long)g0 * R, R, 1, G, R);
        Burst<float, 1> bb1, bb2;  burst_load(bb1, a.b1, H);  burst_load(bb2, a.b2, O);
        Burst<float, 2> bw2;       burst_load(bw2, a.w2, O * H);
        burst_store_w(bw1, w1p, ldx);
        burst_store_w(bx, xs, ldx);
        burst_store(bb1, b1s); burst_store(bb2, b2s); burst_store(bw2, w2s);
    } else {
        FOR_TID(e, G * R) { const int g = fastdiv(dR, e); xs[g * ldx + fastmod(dR, e, g)] = a.readout[(long)g0 * R + e]; }
        FOR_TID(e, H * R) { const int h = fastdiv(dR, e); w1p[h * ldx + fastmod(dR, e, h)] = a.w1[e]; }
        FOR_TID(h, H) { b1s[h] = a.b1[h]; }
        FOR_TID(e, O * H) { w2s[e] = a.w2[e]; }
        FOR_TID(o, O) { b2s[o] = a.b2[o]; }
    }
    FOR_TID(e, (T - G) * R) { const int g = fastdiv(dR, e); xs[(G + g) * ldx + fastmod(dR, e, g)] = 0.0f; }   // rows beyond the batch
    BARRIER();
    // hid = X W1^T            B(k=r, j=h) = w1p[h*ldx + r]
    wg_gemm(T, H, R, xs, ldx, 1, w1p, 1, ldx, hid, ldh, 1);
    BARRIER();
    {
        const float keep_scale = (a.train && a.p_drop > 0.0f) ? 1.0f / (1.0f - a.p_drop) : 1.0f;
        const double pt = (double)a.p_drop * 4294967296.0;
        const uint32_t thresh = (a.train && a.p_drop > 0.0f) ? (uint32_t)(pt > 4294967295.0 ? 4294967295.0 : pt) : 0u;
        FOR_TID(e, T * H) {
            const int g = fastdiv(dH, e), h = fastmod(dH, e, g);
            float v = hid[g * ldh + h] + b1s[h];
            v = v > 0.0f ? v : 0.0f;
            if (thresh) {
                const uint32_t u = drgnn_hash(a.seed, step, (uint32_t)((g0 + g) * H + h));
                v = (u >= thresh) ? v * keep_scale : 0.0f;
            }
            hid[g * ldh + h] = v;
        }
    }
    BARRIER();
    // fc2: 8 interleaved partial dot products per output, combined in fixed order
    FOR_TID(e, T * O * 8) {
        const int q = e & 7, go = e >> 3;
        const int g = fastdiv(dO, go), o = fastmod(dO, go, g);
        float acc = 0.0f;
        for (int h = q; h < H; h += 8) acc = fmaf(hid[g * ldh + h], w2s[o * H + h], acc);
        tmp[e] = acc;
    }
    BARRIER();
    FOR_TID(go, T * O) {
        const int g = fastdiv(dO, go), o = fastmod(dO, go, g);
        float acc = b2s[o];
        for (int q = 0; q < 8; ++q) acc += tmp[go * 8 + q];
        if (a.sigmoid && a.task == DRGNN_TASK_REG) acc = drgnn_sigmoid(acc);
        outs[g * DRGNN_MAX_OUT + o] = acc;
        if (g < G) a.pred[(long)(g0 + g) * O + o] = acc;
    }
    BARRIER();
    if (!a.train || a.grad_readout == nullptr) return;

    // ---- loss and d loss / d pred (mean reduction over the WHOLE batch) --------------------
    FOR_TID(g, T) {
        float loss = 0.0f, wsum = 0.0f;
        if (g < G) {
            if (a.task == DRGNN_TASK_REG) {
                // MSELoss()(pred.reshape(-1), y): mean over B*O elements (O == 1 in the reference)
                const float inv = 1.0f / (float)(a.B * O);
                for (int o = 0; o < O; ++o) {
                    const float ov = outs[g * DRGNN_MAX_OUT + o];
                    const float d = ov - a.y_reg[g0 + g];
                    loss += d * d * inv;
                    douts[g * DRGNN_MAX_OUT + o] = 2.0f * d * inv * (a.sigmoid ? ov * (1.0f - ov) : 1.0f);
                }
                wsum = 1.0f;
            } else {
                // CrossEntropyLoss(weight, reduction='mean'): sum_g w[y_g] * nll_g / sum_g w[y_g]
                float denom = 0.0f;
                for (int q = 0; q < a.B; ++q) denom += a.class_w ? a.class_w[a.y_cls[q]] : 1.0f;
                const int yc = (int)a.y_cls[g0 + g];
                const float wy = a.class_w ? a.class_w[yc] : 1.0f;
                float mx = outs[g * DRGNN_MAX_OUT];
                for (int o = 1; o < O; ++o) mx = outs[g * DRGNN_MAX_OUT + o] > mx ? outs[g * DRGNN_MAX_OUT + o] : mx;
                float se = 0.0f;
                for (int o = 0; o < O; ++o) se += expf(outs[g * DRGNN_MAX_OUT + o] - mx);
                const float lse = logf(se) + mx;
                loss = wy * (lse - outs[g * DRGNN_MAX_OUT + yc]) / denom;
                for (int o = 0; o < O; ++o) {
                    const float p = expf(outs[g * DRGNN_MAX_OUT + o] - lse);
                    douts[g * DRGNN_MAX_OUT + o] = wy * (p - (o == yc ? 1.0f : 0.0f)) / denom;
                }
                wsum = wy;
            }
        } else {
            for (int o = 0; o < O; ++o) douts[g * DRGNN_MAX_OUT + o] = 0.0f;
        }
        red[2 * g] = loss;
        red[2 * g + 1] = wsum;
    }
    BARRIER();
    float* part = a.partials + (long)blk * head_partial_floats(R, H, O);
    float* p_w1 = part;
    float* p_b1 = p_w1 + (long)H * R;
    float* p_w2 = p_b1 + H;
    float* p_b2 = p_w2 + (long)O * H;
    float* p_loss = p_b2 + O;
    // dW2[o,h] = sum_g dout[g,o] hid[g,h];  db2;  loss partial
    FOR_TID(e, O * H) {
        const int o = fastdiv(dH, e), h = fastmod(dH, e, o);
        float acc = 0.0f;
        for (int g = 0; g < T; ++g) acc = fmaf(douts[g * DRGNN_MAX_OUT + o], hid[g * ldh + h], acc);
        p_w2[e] = acc;
    }
    FOR_TID(o, O) {
        float acc = 0.0f;
        for (int g = 0; g < T; ++g) acc += douts[g * DRGNN_MAX_OUT + o];
        p_b2[o] = acc;
    }
    FOR_TID(i, 1) {
        float l = 0.0f, w = 0.0f;
        for (int g = 0; g < T; ++g) { l += red[2 * g]; w += red[2 * g + 1]; }
        p_loss[0] = l;
        p_loss[1] = w;
    }
    BARRIER();
    // d hid (in place): (dout W2) * relu' * dropout mask -- both folded into "hid != 0"
    {
        const float keep_scale = (a.p_drop > 0.0f) ? 1.0f / (1.0f - a.p_drop) : 1.0f;
        FOR_TID(e, T * H) {
            const int g = fastdiv(dH, e), h = fastmod(dH, e, g);
            float acc = 0.0f;
            for (int o = 0; o < O; ++o) acc = fmaf(douts[g * DRGNN_MAX_OUT + o], w2s[o * H + h], acc);
            hid[g * ldh + h] = (hid[g * ldh + h] != 0.0f) ? acc * keep_scale : 0.0f;
        }
    }
    BARRIER();
    // dW1 = dhid^T X   (A(i=h,k=g) = hid[g*ldh + h];  B(k=g,j=r) = xs[g*ldx + r])
    wg_gemm(H, R, T, hid, 1, ldh, xs, ldx, 1, p_w1, R, 1);
    FOR_TID(h, H) {
        float acc = 0.0f;
        for (int g = 0; g < T; ++g) acc += hid[g * ldh + h];
        p_b1[h] = acc;
    }
    // d readout = dhid W1   (B(k=h, j=r) = w1p[h*ldx + r]); rows beyond the batch are skipped.
    // K = H is long and there are few output tiles: split K over the idle waves
    {
        const int tiles = ((G + 15) >> 4) * ((R + 15) >> 4);
        int KS = imin(DRGNN_NWAVES / imax(tiles, 1), DRGNN_HEAD_TMP / imax(G * R, 1));
        KS = imax(1, imin(KS, 8));
        wg_gemm(G, R, H, hid, ldh, 1, w1p, ldx, 1, a.grad_readout + (long)g0 * R, R, 1, KS, tmp);
    }
}

// ---- reduction of the head partials + Adam -----------------------------------------------
struct HeadReduceArgs {
    const float* partials;    // [n_wg][P]   n_wg = ceil(B / head_tile(B))
    int n_wg, P;              // P = head_partial_floats
    float* grad;              // contiguous [H*R + H + O*H + O] block of the flat gradient
    float* loss;              // scalar out
    int32_t* step;            // incremented once per step by the reducer
};

DEV void head_reduce_item(const HeadReduceArgs& a, int item) {
    const int n_grad = a.P - 2;
    if (item < n_grad) {
        float acc = 0.0f;
        for (int w = 0; w < a.n_wg; ++w) acc += a.partials[(long)w * a.P + item];
        a.grad[item] = acc;
    } else if (item == n_grad) {
        float acc = 0.0f;
        for (int w = 0; w < a.n_wg; ++w) acc += a.partials[(long)w * a.P + n_grad];
        if (a.loss) a.loss[0] = acc;
        if (a.step) a.step[0] = a.step[0] + 1;
    }
}

struct AdamArgs {
    float* param; const float* grad; float* exp_avg; float* exp_avg_sq;
    const int32_t* step;      // already counts this step (>= 1)
    int64_t n;
    float lr, beta1, beta2, eps, weight_decay;
};

// torch.optim.Adam (no amsgrad, maximize=False): identical operation order to the reference
// implementation (_single_tensor_adam): denom = sqrt(v)/sqrt(bc2) + eps; p -= (lr/bc1) * m/denom
DEV double adam_ipow(double b, int t) {
    double r = 1.0;
    for (unsigned e = (unsigned)(t > 0 ? t : 0); e; e >>= 1) {
        if (e & 1u) r *= b;
        b *= b;
    }
    return r;
}
// adam_item split in two so that a caller can have the element's state and the step-dependent scalars in flight
// while it is still computing the gradient (same arithmetic, same order -> same bits)
struct AdamPre { float p, m, v, step_size, sqrt_bc2; bool ok; };
// the step-dependent scalars: ~35 dependent double-precision operations behind the load of the step index
DEV void adam_bias_scalars(const AdamArgs& a, float& step_size, float& sqrt_bc2) {
    // bias corrections in double, as the Python-side scalars of torch's reference path; beta^t by
    // repeated squaring (t is an integer): a libm pow() in double costs more than the rest of the launch
    const int t = a.step[0];
    const double bc1 = 1.0 - adam_ipow((double)a.beta1, t);
    const double bc2 = 1.0 - adam_ipow((double)a.beta2, t);
    step_size = (float)((double)a.lr / bc1);
    sqrt_bc2 = (float)sqrt(bc2);
}
// the element's state only (k_update: another wave forms the scalars meanwhile and hands them over through LDS)
DEV AdamPre adam_prefetch_state(const AdamArgs& a, int64_t i) {
    AdamPre r;
    r.ok = i >= 0 && i < a.n;
    r.p = r.m = r.v = 0.0f; r.step_size = 0.0f; r.sqrt_bc2 = 1.0f;
    if (!r.ok) return r;
    r.p = a.param[i];
    r.m = a.exp_avg[i];
    r.v = a.exp_avg_sq[i];
    return r;
}
DEV AdamPre adam_prefetch(const AdamArgs& a, int64_t i) {
    AdamPre r = adam_prefetch_state(a, i);
    if (r.ok) adam_bias_scalars(a, r.step_size, r.sqrt_bc2);
    return r;
}
DEV void adam_apply(const AdamArgs& a, int64_t i, float g, const AdamPre& r) {
#ifndef DRGNN_EMU
#pragma clang fp contract(off)      // every product and sum rounded on its own: the same bits wherever this is inlined
#endif
    if (!r.ok) return;
    if (a.weight_decay != 0.0f) g = fmaf(a.weight_decay, r.p, g);
    const float m = r.m + (g - r.m) * (1.0f - a.beta1);          // lerp
    const float v = a.beta2 * r.v + (1.0f - a.beta2) * g * g;
    a.exp_avg[i] = m;
    a.exp_avg_sq[i] = v;
    const float denom = sqrtf(v) / r.sqrt_bc2 + a.eps;
    a.param[i] = r.p - r.step_size * (m / denom);
}
DEV void adam_item(const AdamArgs& a, int64_t i) {
    if (i >= a.n) return;
    const float g = a.grad[i];
    const AdamPre r = adam_prefetch(a, i);
    adam_apply(a, i, g, r);          // ONE statement of the arithmetic for every caller: identical bits
}
