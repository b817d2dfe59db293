// Parity build of the kernels with the extended shape set (MonomialSurface): -ffp-contract=off.
#define RPT_NS rpt_strict_ext
#define RPT_EXT_SHAPES 1
#include "kernels.inc"
