// one-pass iteration kernels of the oracle families with D class 4 (FAM_D_*, bz_kernels.h), float
#define BZ_FAMILY_DK 4
#define BZ_FAMILY_T float
#include "bz_families.inc"
