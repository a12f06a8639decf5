// Instantiates the distributed-staging k=7 variants.
#include "nc_conv_kernel.hip.h"
NC_INSTANTIATE_CONV_DIST(7, 8, 10)
