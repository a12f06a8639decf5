// XV-only instances of the two-tap sub-pixel up-convolutions (power-of-two strides; nc_conv_kernel.hip.h "XVK").
#include "nc_conv_kernel.hip.h"
NC_INSTANTIATE_CONV_XV(xv_sub_k2, 2, 16, 20, false, 1)
