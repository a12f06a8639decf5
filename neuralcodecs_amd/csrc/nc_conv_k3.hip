// Instantiates the implicit-GEMM convolution for taps-per-phase K=3 (reduction block of 16 input channels,
// up to 20 prefetched window words per lane).
#include "nc_conv_kernel.hip.h"
NC_INSTANTIATE_CONV_K(3, 16, 20)
NC_INSTANTIATE_CONV_NARROW(3, 16, 20)
NC_INSTANTIATE_CONV_SLIM(3, 8, 10)
