// Fast build of the kernels: FMA contraction allowed (-ffp-contract=fast).
#define RPT_NS rpt_fast
#include "kernels.inc"
