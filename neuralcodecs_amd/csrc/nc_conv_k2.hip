// Instantiates the implicit-GEMM convolution for taps-per-phase K=2 (reduction block of 16 input channels,
// up to 20 prefetched window words per lane).
#include "nc_conv_kernel.hip.h"
NC_INSTANTIATE_CONV_K(2, 16, 20)
NC_INSTANTIATE_CONV_NARROW(2, 16, 20)
NC_INSTANTIATE_CONV_SUB(2, 16, 20)
NC_INSTANTIATE_CONV_SUB_NARROW(2, 16, 20)
