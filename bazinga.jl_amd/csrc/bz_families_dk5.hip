// one-pass iteration kernels of the oracle families with D class 5 (FAM_D_*, bz_kernels.h)
#define BZ_FAMILY_DK 5
#include "bz_families.inc"
