// drgnn_step_tu.hip -- one translation unit of the fused step kernels' instantiations.
//   -DDRGNN_AF_FAM=<1..8> -DDRGNN_AF_W=<16|32|48|64>: one (family, width) of the aggregation-first kernels (drgnn_step_af.h:
//       the unit defines that family's kernel lookup, which instantiates the kernels);
// (The product-first kernels of rounds 2 - 3, drgnn_step.h / drgnn_step1.h, are no longer instantiated for the device.)
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
#elif DRGNN_AF_FAM == DRGNN_AF_SGAT_XG
DRGNN_AF_DEFINE_SGAT_XG(DRGNN_AF_W)
#elif DRGNN_AF_FAM == DRGNN_AF_FOUT_XG
DRGNN_AF_DEFINE_FOUT_XG(DRGNN_AF_W)
#elif DRGNN_AF_FAM == DRGNN_AF_GINET_SG
DRGNN_AF_DEFINE_GINET_SG(DRGNN_AF_W)
#else
#error "DRGNN_AF_FAM: 1 .. 8"
#endif
#else
#error "compile with -DDRGNN_AF_FAM=<family> -DDRGNN_AF_W=<width>"
#endif  // DRGNN_AF_FAM
