// Instantiates the implicit-GEMM convolution for taps-per-phase K=6 (reduction block of 5 input channels,
// up to 18 prefetched window words per lane).
#include "nc_conv_kernel.hip.h"
NC_INSTANTIATE_CONV_K(6, 5, 18)
