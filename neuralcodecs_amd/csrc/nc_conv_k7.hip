// Instantiates the implicit-GEMM convolution for taps-per-phase K=7 (reduction block of 8 input channels,
// up to 10 prefetched window words per lane).
#include "nc_conv_kernel.hip.h"
NC_INSTANTIATE_CONV_K(7, 8, 10)

NC_INSTANTIATE_CONV_NARROW(7, 8, 10)
NC_INSTANTIATE_CONV_SLIM(7, 4, 5)
