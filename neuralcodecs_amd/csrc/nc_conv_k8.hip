// Instantiates the implicit-GEMM convolution for taps-per-phase K=8 (reduction block of 8 input channels).
#include "nc_conv_kernel.hip.h"
NC_INSTANTIATE_CONV_K(8, 8)
