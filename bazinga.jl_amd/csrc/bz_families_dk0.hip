// one-pass iteration kernels of the oracle families with D class 0 (FAM_D_*, bz_kernels.h)
#define BZ_FAMILY_DK 0
#include "bz_families.inc"
