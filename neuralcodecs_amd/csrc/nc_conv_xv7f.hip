// XV-only instances of the fused residual unit (k = 7 + Snake + 1x1 + skip in one launch; nc_conv_kernel.hip.h "XVK").
#include "nc_conv_kernel.hip.h"
NC_INSTANTIATE_CONV_XV(xv_fused_k7, 7, 8, 10, true, 0)
