// drgnn_step_tu.hip -- explicit instantiations of the fused step kernel for ONE kind of net (-DDRGNN_TU_KIND=0|1|2; 3 = the one-workgroup GINet step):
// five feature widths x {per-mini-batch workspace, cached whole-set workspace}.  See drgnn_kernels.h.
#include "drgnn_kernels.h"
#ifndef DRGNN_TU_KIND
#error "compile with -DDRGNN_TU_KIND=<kind>"
#endif
#define DRGNN_STEP_INST(K, XF)                                              \
    template __global__ void k_step_co_topo<K, XF, false>(StepCoLaunch);    \
    template __global__ void k_step_co_topo<K, XF, true>(StepCoLaunch);
#if DRGNN_TU_KIND == 3
// the one-workgroup-per-graph GINet step (drgnn_step1.h)
#define DRGNN_STEP1_INST(K, XF)                                                    \
    template __global__ void k_step1_co_topo<XF, false, false>(StepCoLaunch);      \
    template __global__ void k_step1_co_topo<XF, true, false>(StepCoLaunch);
DRGNN_STEP_FOR_WIDTHS(DRGNN_STEP1_INST, 0)
#elif DRGNN_TU_KIND == 4
// ... its form with both branches in every phase (generic and 32-wide)
template __global__ void k_step1_co_topo<0, false, true>(StepCoLaunch);
template __global__ void k_step1_co_topo<0, true, true>(StepCoLaunch);
template __global__ void k_step1_co_topo<32, false, true>(StepCoLaunch);
template __global__ void k_step1_co_topo<32, true, true>(StepCoLaunch);
// ... and the 32-wide paired form with the capacity-class LDS layout
template __global__ void k_step1_co_topo<32, false, true, 1>(StepCoLaunch);
template __global__ void k_step1_co_topo<32, true, true, 1>(StepCoLaunch);
#elif DRGNN_TU_KIND == 5 || DRGNN_TU_KIND == 6
// the node-split, aggregation-first step of sGAT (5) / FoutNet (6) (drgnn_step2.h): 32-wide, {mini-batch, cached} x
// {run-time, capacity-class layout} x {one, two workgroups per graph}
#define DRGNN_STEP2_K (DRGNN_TU_KIND == 5 ? DRGNN_SGAT : DRGNN_FOUT)
template __global__ void k_step2_co_topo<DRGNN_STEP2_K, 32, false, 0, 1>(StepCoLaunch);
template __global__ void k_step2_co_topo<DRGNN_STEP2_K, 32, true, 0, 1>(StepCoLaunch);
template __global__ void k_step2_co_topo<DRGNN_STEP2_K, 32, false, 1, 1>(StepCoLaunch);
template __global__ void k_step2_co_topo<DRGNN_STEP2_K, 32, true, 1, 1>(StepCoLaunch);
template __global__ void k_step2_co_topo<DRGNN_STEP2_K, 32, false, 0, 2>(StepCoLaunch);
template __global__ void k_step2_co_topo<DRGNN_STEP2_K, 32, true, 0, 2>(StepCoLaunch);
template __global__ void k_step2_co_topo<DRGNN_STEP2_K, 32, false, 1, 2>(StepCoLaunch);
template __global__ void k_step2_co_topo<DRGNN_STEP2_K, 32, true, 1, 2>(StepCoLaunch);
#elif DRGNN_TU_KIND == 7
// the aggregation-first GINet step (drgnn_step3.h): 32-wide, {mini-batch, cached} x {run-time, capacity-class layout}
template __global__ void k_step3_co_topo<32, false, 0>(StepCoLaunch);
template __global__ void k_step3_co_topo<32, true, 0>(StepCoLaunch);
template __global__ void k_step3_co_topo<32, false, 1>(StepCoLaunch);
template __global__ void k_step3_co_topo<32, true, 1>(StepCoLaunch);
#elif DRGNN_TU_KIND == 8
// ... with both branches of a graph in one workgroup (net_step3_graph_both)
template __global__ void k_step3b_co_topo<32, false, 0>(StepCoLaunch);
template __global__ void k_step3b_co_topo<32, true, 0>(StepCoLaunch);
template __global__ void k_step3b_co_topo<32, false, 1>(StepCoLaunch);
template __global__ void k_step3b_co_topo<32, true, 1>(StepCoLaunch);
#else
DRGNN_STEP_FOR_WIDTHS(DRGNN_STEP_INST, DRGNN_TU_KIND)
// ... and the 32-wide kernels with the capacity-class LDS layout (net_step_graph: CLS = 1)
template __global__ void k_step_co_topo<DRGNN_TU_KIND, 32, false, 1>(StepCoLaunch);
template __global__ void k_step_co_topo<DRGNN_TU_KIND, 32, true, 1>(StepCoLaunch);
#endif
