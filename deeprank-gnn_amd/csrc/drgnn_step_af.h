// drgnn_step_af.h -- which __global__ instance of the AGGREGATION-FIRST step families (drgnn_step2.h: sGAT / FoutNet;
// drgnn_step3.h: GINet) a launch takes.  The families are instantiated per padded feature width 16 / 32 / 48 / 64,
// {per-mini-batch, cached whole-set workspace}, {training, inference}; training launches of the 32- and 48-wide kernels also with the
// capacity-class LDS layout (CLS = 1, drgnn_step.h), the single-branch nets with one or two workgroups per graph.
//
// One lookup function per (family, width).  In the library build (Makefile: DRGNN_SPLIT_TU) each is DEFINED in the translation
// unit that thereby instantiates the kernels it names (drgnn_step_tu.hip with -DDRGNN_AF_FAM=<family> -DDRGNN_AF_W=<width>) and
// only declared everywhere else, so the kernels are compiled once, in parallel, and no list of them has to be kept in two
// places; a single-unit build (profiling / ablation variants) defines all of them in drgnn_capi.hip.
#ifndef DRGNN_STEP_AF_H
#define DRGNN_STEP_AF_H
#ifndef DRGNN_EMU

typedef void (*drgnn_step_kernel_t)(StepCoLaunch);

#define DRGNN_AF_GINET_TWO 1      // k_step3_co_topo: one workgroup per (graph, branch)
#define DRGNN_AF_GINET_ONE 2      // k_step3b_co_topo: both branches of a graph in one workgroup
#define DRGNN_AF_SGAT 3           // k_step2_co_topo<DRGNN_SGAT>
#define DRGNN_AF_FOUT 4           // k_step2_co_topo<DRGNN_FOUT>
#define DRGNN_AF_SGAT_WHOLE 5     // k_step2_co_topo<DRGNN_SGAT, ., false, ., 1, true>: a unit of its own, see af_pick_single
#define DRGNN_AF_SGAT_XG 6        // k_step2_co_topo<DRGNN_SGAT, ., ., 0, ., ., true>: x rows read from memory (graphs beyond the staged form's LDS)
#define DRGNN_AF_FOUT_XG 7        // ... of FoutNet
#define DRGNN_AF_GINET_SG 8       // k_step3b_co_topo<., ., 0, ., true>: S rows read from memory (graphs beyond the staged form's LDS)

// (cls: 1 = capacity-class layout, honoured for the 32- and 48-wide kernels only, training and inference launches -- the host
// asks for nothing else; 48: the feature count of the reference's shipped regression models)
template <int XF> drgnn_step_kernel_t af_pick_ginet_two(bool gather, int cls, bool train) {
    constexpr int C1 = (XF == 32 || XF == 48) ? 1 : 0;
    if (!train && cls && C1) return gather ? k_step3_co_topo<XF, true, C1, false> : k_step3_co_topo<XF, false, C1, false>;
    if (!train) return gather ? k_step3_co_topo<XF, true, 0, false> : k_step3_co_topo<XF, false, 0, false>;
    if (cls && C1) return gather ? k_step3_co_topo<XF, true, C1, true> : k_step3_co_topo<XF, false, C1, true>;
    return gather ? k_step3_co_topo<XF, true, 0, true> : k_step3_co_topo<XF, false, 0, true>;
}
template <int XF> drgnn_step_kernel_t af_pick_ginet_one(bool gather, int cls, bool train) {
    constexpr int C1 = (XF == 32 || XF == 48) ? 1 : 0;
    if (!train && cls && C1) return gather ? k_step3b_co_topo<XF, true, C1, false> : k_step3b_co_topo<XF, false, C1, false>;
    if (!train) return gather ? k_step3b_co_topo<XF, true, 0, false> : k_step3b_co_topo<XF, false, 0, false>;
    if (cls && C1) return gather ? k_step3b_co_topo<XF, true, C1, true> : k_step3b_co_topo<XF, false, C1, true>;
    return gather ? k_step3b_co_topo<XF, true, 0, true> : k_step3b_co_topo<XF, false, 0, true>;
}
template <int XF> drgnn_step_kernel_t af_pick_ginet_sg(bool gather, bool train) {
    if (!train) return gather ? k_step3b_co_topo<XF, true, 0, false, true> : k_step3b_co_topo<XF, false, 0, false, true>;
    return gather ? k_step3b_co_topo<XF, true, 0, true, true> : k_step3b_co_topo<XF, false, 0, true, true>;
}
// sGAT's training launches with one workgroup per graph on a per-mini-batch workspace are the ones whose co-launched builder --
// one workgroup per graph working BOTH chains off, with edge weights -- bounds the launch (batch 128 and beyond, topology
// rebuilt).  The single-branch units are compiled at -Os (Makefile), which suits the step's phases and costs that builder chain
// 1 us (profiles/r05_ab_opt_level.txt): these instances live in a unit of their own, compiled at -O3.
// (48 features: the class instance of THIS launch is slower than the run-time layout -- 27.45 against 26.77 us per step at batch
// 128, profiles/r05_cls48_ab.txt -- although it does not spill: the run-time layout with the class's capacities steps those)
template <int XF> drgnn_step_kernel_t af_pick_sgat_whole(int cls) {
    constexpr int C1 = (XF == 32) ? 1 : 0;
    if (cls && C1) return k_step2_co_topo<DRGNN_SGAT, XF, false, C1, 1, true>;
    return k_step2_co_topo<DRGNN_SGAT, XF, false, 0, 1, true>;
}
template <int XF> drgnn_step_kernel_t af_sgat_whole(int cls);      // (defined per width below: af_sgat_whole_<W>)
// split: workgroups per graph (2: training launches only)
template <int KIND, int XF> drgnn_step_kernel_t af_pick_single(bool gather, int cls, int split, bool train) {
    constexpr int C1 = (XF == 32 || XF == 48) ? 1 : 0;
    if (!train && cls && C1) return gather ? k_step2_co_topo<KIND, XF, true, C1, 1, false> : k_step2_co_topo<KIND, XF, false, C1, 1, false>;
    if (!train) return gather ? k_step2_co_topo<KIND, XF, true, 0, 1, false> : k_step2_co_topo<KIND, XF, false, 0, 1, false>;
    if (split == 2) {
        if (cls && C1) return gather ? k_step2_co_topo<KIND, XF, true, C1, 2, true> : k_step2_co_topo<KIND, XF, false, C1, 2, true>;
        return gather ? k_step2_co_topo<KIND, XF, true, 0, 2, true> : k_step2_co_topo<KIND, XF, false, 0, 2, true>;
    }
    if constexpr (KIND == DRGNN_SGAT) {
        if (!gather) return af_sgat_whole<XF>(cls);
        if (cls && C1) return k_step2_co_topo<KIND, XF, true, C1, 1, true>;
        return k_step2_co_topo<KIND, XF, true, 0, 1, true>;
    } else {
        if (cls && C1) return gather ? k_step2_co_topo<KIND, XF, true, C1, 1, true> : k_step2_co_topo<KIND, XF, false, C1, 1, true>;
        return gather ? k_step2_co_topo<KIND, XF, true, 0, 1, true> : k_step2_co_topo<KIND, XF, false, 0, 1, true>;
    }
}

// the from-memory forms (run-time LDS layout only): graphs whose S AND x tiles do not fit the 160 KiB -- level 1: the x rows
// stay in memory; level 2 (32-, 48- and 64-wide: the widths whose S tile fills the LDS before the edge arrays do): the S rows too
template <int KIND, int XF, int LV> drgnn_step_kernel_t af_pick_single_xg_level(bool gather, int split, bool train) {
    if (!train) return gather ? k_step2_co_topo<KIND, XF, true, 0, 1, false, LV> : k_step2_co_topo<KIND, XF, false, 0, 1, false, LV>;
    if (split == 2) return gather ? k_step2_co_topo<KIND, XF, true, 0, 2, true, LV> : k_step2_co_topo<KIND, XF, false, 0, 2, true, LV>;
    return gather ? k_step2_co_topo<KIND, XF, true, 0, 1, true, LV> : k_step2_co_topo<KIND, XF, false, 0, 1, true, LV>;
}
template <int KIND, int XF> drgnn_step_kernel_t af_pick_single_xg(bool gather, int split, bool train, int level) {
    if constexpr (XF >= 32) { if (level == 2) return af_pick_single_xg_level<KIND, XF, 2>(gather, split, train); }
    return level == 1 ? af_pick_single_xg_level<KIND, XF, 1>(gather, split, train) : nullptr;
}
#define DRGNN_AF_DEFINE_SGAT_XG(W) DRGNN_AF_DEFINE_SGAT_XG_X(W)
#define DRGNN_AF_DEFINE_FOUT_XG(W) DRGNN_AF_DEFINE_FOUT_XG_X(W)
#define DRGNN_AF_DEFINE_SGAT_XG_X(W) \
    drgnn_step_kernel_t af_sgat_xg_##W(bool gather, int split, bool train, int level) { return af_pick_single_xg<DRGNN_SGAT, W>(gather, split, train, level); }
#define DRGNN_AF_DEFINE_FOUT_XG_X(W) \
    drgnn_step_kernel_t af_fout_xg_##W(bool gather, int split, bool train, int level) { return af_pick_single_xg<DRGNN_FOUT, W>(gather, split, train, level); }

#define DRGNN_AF_DECLARE(W)                                                                     \
    drgnn_step_kernel_t af_ginet_two_##W(bool gather, int cls, bool train);                    \
    drgnn_step_kernel_t af_ginet_one_##W(bool gather, int cls, bool train);                    \
    drgnn_step_kernel_t af_ginet_sg_##W(bool gather, bool train);                              \
    drgnn_step_kernel_t af_sgat_xg_##W(bool gather, int split, bool train, int level);         \
    drgnn_step_kernel_t af_fout_xg_##W(bool gather, int split, bool train, int level);         \
    drgnn_step_kernel_t af_sgat_##W(bool gather, int cls, int split, bool train);              \
    drgnn_step_kernel_t af_fout_##W(bool gather, int cls, int split, bool train);              \
    drgnn_step_kernel_t af_sgat_whole_##W(int cls);                                            \
    template <> inline drgnn_step_kernel_t af_sgat_whole<W>(int cls) { return af_sgat_whole_##W(cls); }
DRGNN_AF_DECLARE(16) DRGNN_AF_DECLARE(32) DRGNN_AF_DECLARE(48) DRGNN_AF_DECLARE(64)
#undef DRGNN_AF_DECLARE

// (two levels: the width may itself be a macro -- the translation units pass DRGNN_AF_W)
#define DRGNN_AF_DEFINE_GINET_TWO(W) DRGNN_AF_DEFINE_GINET_TWO_X(W)
#define DRGNN_AF_DEFINE_GINET_ONE(W) DRGNN_AF_DEFINE_GINET_ONE_X(W)
#define DRGNN_AF_DEFINE_GINET_SG(W) DRGNN_AF_DEFINE_GINET_SG_X(W)
#define DRGNN_AF_DEFINE_GINET_SG_X(W) \
    drgnn_step_kernel_t af_ginet_sg_##W(bool gather, bool train) { return af_pick_ginet_sg<W>(gather, train); }
#define DRGNN_AF_DEFINE_SGAT(W) DRGNN_AF_DEFINE_SGAT_X(W)
#define DRGNN_AF_DEFINE_FOUT(W) DRGNN_AF_DEFINE_FOUT_X(W)
#define DRGNN_AF_DEFINE_SGAT_WHOLE(W) DRGNN_AF_DEFINE_SGAT_WHOLE_X(W)
#define DRGNN_AF_DEFINE_SGAT_WHOLE_X(W) \
    drgnn_step_kernel_t af_sgat_whole_##W(int cls) { return af_pick_sgat_whole<W>(cls); }
#define DRGNN_AF_DEFINE_GINET_TWO_X(W) \
    drgnn_step_kernel_t af_ginet_two_##W(bool gather, int cls, bool train) { return af_pick_ginet_two<W>(gather, cls, train); }
#define DRGNN_AF_DEFINE_GINET_ONE_X(W) \
    drgnn_step_kernel_t af_ginet_one_##W(bool gather, int cls, bool train) { return af_pick_ginet_one<W>(gather, cls, train); }
#define DRGNN_AF_DEFINE_SGAT_X(W)                                                    \
    drgnn_step_kernel_t af_sgat_##W(bool gather, int cls, int split, bool train) {   \
        return af_pick_single<DRGNN_SGAT, W>(gather, cls, split, train);              \
    }
#define DRGNN_AF_DEFINE_FOUT_X(W)                                                    \
    drgnn_step_kernel_t af_fout_##W(bool gather, int cls, int split, bool train) {   \
        return af_pick_single<DRGNN_FOUT, W>(gather, cls, split, train);              \
    }
#define DRGNN_AF_FOR_WIDTHS(X) X(16) X(32) X(48) X(64)

#if defined(DRGNN_KERNELS_MAIN)
#if !defined(DRGNN_SPLIT_TU)
DRGNN_AF_FOR_WIDTHS(DRGNN_AF_DEFINE_GINET_TWO)
DRGNN_AF_FOR_WIDTHS(DRGNN_AF_DEFINE_GINET_ONE)
DRGNN_AF_FOR_WIDTHS(DRGNN_AF_DEFINE_GINET_SG)
DRGNN_AF_FOR_WIDTHS(DRGNN_AF_DEFINE_SGAT)
DRGNN_AF_FOR_WIDTHS(DRGNN_AF_DEFINE_FOUT)
DRGNN_AF_FOR_WIDTHS(DRGNN_AF_DEFINE_SGAT_WHOLE)
DRGNN_AF_FOR_WIDTHS(DRGNN_AF_DEFINE_SGAT_XG)
DRGNN_AF_FOR_WIDTHS(DRGNN_AF_DEFINE_FOUT_XG)
#endif
// family: DRGNN_AF_*; width: 16 / 32 / 48 / 64.  nullptr: no such instance
// level: 1 / 2 of the from-memory families (DRGNN_AF_SGAT_XG / _FOUT_XG)
static drgnn_step_kernel_t af_step_kernel(int family, int width, bool gather, int cls, int split, bool train, int level = 0) {
#define DRGNN_AF_CASE(W)                                                                       \
    case W:                                                                                     \
        switch (family) {                                                                       \
        case DRGNN_AF_GINET_TWO: return af_ginet_two_##W(gather, cls, train);                   \
        case DRGNN_AF_GINET_ONE: return af_ginet_one_##W(gather, cls, train);                   \
        case DRGNN_AF_GINET_SG: return af_ginet_sg_##W(gather, train);                          \
        case DRGNN_AF_SGAT_XG: return af_sgat_xg_##W(gather, split, train, level);              \
        case DRGNN_AF_FOUT_XG: return af_fout_xg_##W(gather, split, train, level);              \
        case DRGNN_AF_SGAT: return af_sgat_##W(gather, cls, split, train);                      \
        case DRGNN_AF_FOUT: return af_fout_##W(gather, cls, split, train);                      \
        default: return nullptr;                                                                \
        }
    switch (width) {
        DRGNN_AF_FOR_WIDTHS(DRGNN_AF_CASE)
    default: return nullptr;
    }
#undef DRGNN_AF_CASE
}
#endif  // DRGNN_KERNELS_MAIN

#endif  // !DRGNN_EMU
#endif
