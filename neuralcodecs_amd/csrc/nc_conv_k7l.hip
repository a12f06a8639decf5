// Instantiates the light k=7 variants: reduction block of 4 input channels, 3 workgroups per CU.
#include "nc_conv_kernel.hip.h"
NC_INSTANTIATE_CONV_LIGHT(7, 4, 5)
