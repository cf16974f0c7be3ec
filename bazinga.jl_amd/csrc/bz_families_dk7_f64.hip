// one-pass iteration kernels of the oracle families with D class 7 (FAM_D_*, bz_kernels.h), double
#define BZ_FAMILY_DK 7
#define BZ_FAMILY_T double
#include "bz_families.inc"
