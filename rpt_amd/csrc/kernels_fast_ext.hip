// Fast build of the kernels with the extended shape set (MonomialSurface): contraction allowed.
#define RPT_NS rpt_fast_ext
#define RPT_EXT_SHAPES 1
#include "kernels.inc"
