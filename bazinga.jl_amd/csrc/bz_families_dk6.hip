// one-pass iteration kernels of the oracle families with D class 6 (FAM_D_*, bz_kernels.h)
#define BZ_FAMILY_DK 6
#include "bz_families.inc"
