// Instantiates the implicit-GEMM convolution for taps-per-phase K=16 (reduction block of 2 input channels,
// up to 18 prefetched window words per lane).
#include "nc_conv_kernel.hip.h"
NC_INSTANTIATE_CONV_K(16, 2, 18)
NC_INSTANTIATE_CONV_NARROW(16, 2, 18)
