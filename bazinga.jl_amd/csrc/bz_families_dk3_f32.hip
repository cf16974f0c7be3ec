// one-pass iteration kernels of the oracle families with D class 3 (FAM_D_*, bz_kernels.h), float
#define BZ_FAMILY_DK 3
#define BZ_FAMILY_T float
#include "bz_families.inc"
