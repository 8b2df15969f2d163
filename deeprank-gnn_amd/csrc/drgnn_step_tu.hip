// drgnn_step_tu.hip -- one translation unit of the fused step kernels' instantiations.
//   -DDRGNN_AF_FAM=<1..5> -DDRGNN_AF_W=<16|32|48|64>: one (family, width) of the aggregation-first kernels (drgnn_step_af.h:
//       the unit defines that family's kernel lookup, which instantiates the kernels);
//   -DDRGNN_TU_KIND=0|1|2: the product-first kernels of one kind of net (drgnn_step.h; five feature widths x {per-mini-batch
//       workspace, cached whole-set workspace}); 3 / 4: the one-workgroup product-first GINet step (drgnn_step1.h).
#include "drgnn_kernels.h"
#if defined(DRGNN_AF_FAM)
#if DRGNN_AF_FAM == DRGNN_AF_GINET_TWO
DRGNN_AF_DEFINE_GINET_TWO(DRGNN_AF_W)
#elif DRGNN_AF_FAM == DRGNN_AF_GINET_ONE
DRGNN_AF_DEFINE_GINET_ONE(DRGNN_AF_W)
#elif DRGNN_AF_FAM == DRGNN_AF_SGAT
DRGNN_AF_DEFINE_SGAT(DRGNN_AF_W)
#elif DRGNN_AF_FAM == DRGNN_AF_FOUT
DRGNN_AF_DEFINE_FOUT(DRGNN_AF_W)
#elif DRGNN_AF_FAM == DRGNN_AF_SGAT_WHOLE
DRGNN_AF_DEFINE_SGAT_WHOLE(DRGNN_AF_W)
#else
#error "DRGNN_AF_FAM: 1 .. 5"
#endif
#else
#ifndef DRGNN_TU_KIND
#error "compile with -DDRGNN_TU_KIND=<kind> or -DDRGNN_AF_FAM=<family> -DDRGNN_AF_W=<width>"
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
#else
DRGNN_STEP_FOR_WIDTHS(DRGNN_STEP_INST, DRGNN_TU_KIND)
// ... and the 32-wide kernels with the capacity-class LDS layout (net_step_graph: CLS = 1)
template __global__ void k_step_co_topo<DRGNN_TU_KIND, 32, false, 1>(StepCoLaunch);
template __global__ void k_step_co_topo<DRGNN_TU_KIND, 32, true, 1>(StepCoLaunch);
#endif
#endif  // DRGNN_AF_FAM
