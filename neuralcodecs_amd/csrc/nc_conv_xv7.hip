// XV-only instances of the k = 7 residual-unit convolution (nc_conv_kernel.hip.h "XVK"): vectorised one-run staging, legacy modes compiled out.
#include "nc_conv_kernel.hip.h"
NC_INSTANTIATE_CONV_XV(xv_k7, 7, 8, 10, false, 0)
