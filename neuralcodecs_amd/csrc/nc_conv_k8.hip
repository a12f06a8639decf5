// Instantiates the implicit-GEMM convolution for taps-per-phase K=8 (reduction block of 4 input channels,
// up to 18 prefetched window words per lane).
#include "nc_conv_kernel.hip.h"
NC_INSTANTIATE_CONV_K(8, 4, 18)
