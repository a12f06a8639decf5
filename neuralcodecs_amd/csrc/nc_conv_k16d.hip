// Distributed-staging variants of the k = 16 strided convolution and the k = 2 sub-pixel up-convolution (tiny grids, see nc_conv.hip).
#include "nc_conv_kernel.hip.h"
NC_INSTANTIATE_CONV_DIST_SMALL(dist_k16, 16, 2, 18, false)
NC_INSTANTIATE_CONV_DIST_SMALL(dist_sub_k2, 2, 16, 20, true)
