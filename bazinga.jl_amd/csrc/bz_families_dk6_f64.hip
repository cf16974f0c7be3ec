// one-pass iteration kernels of the oracle families with D class 6 (FAM_D_*, bz_kernels.h), double
#define BZ_FAMILY_DK 6
#define BZ_FAMILY_T double
#include "bz_families.inc"
