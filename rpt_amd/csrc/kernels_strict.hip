// Parity build of the kernels: compiled with -ffp-contract=off (see __graft_entry__.build).
#define RPT_NS rpt_strict
#include "kernels.inc"
