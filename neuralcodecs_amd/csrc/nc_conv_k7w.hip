// Instantiates the wide k=7 variants: 8 waves / 512 columns per workgroup, one workgroup per CU.
#include "nc_conv_kernel.hip.h"
NC_INSTANTIATE_CONV_WIDE(7, 8, 9)
