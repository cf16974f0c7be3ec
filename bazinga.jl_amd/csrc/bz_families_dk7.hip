// one-pass iteration kernels of the oracle families with D class 7 (FAM_D_*, bz_kernels.h)
#define BZ_FAMILY_DK 7
#include "bz_families.inc"
