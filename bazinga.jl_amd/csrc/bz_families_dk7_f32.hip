// one-pass iteration kernels of the oracle families with D class 7 (FAM_D_*, bz_kernels.h), float
#define BZ_FAMILY_DK 7
#define BZ_FAMILY_T float
#include "bz_families.inc"
