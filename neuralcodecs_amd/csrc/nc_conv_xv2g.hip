// XV-only instances of the two-tap sub-pixel up-convolutions for any stride (multiply-shift row map; nc_conv_kernel.hip.h "XVK").
#include "nc_conv_kernel.hip.h"
NC_INSTANTIATE_CONV_XV(xv_subg_k2, 2, 16, 20, false, 2)
