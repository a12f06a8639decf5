// Instantiates the implicit-GEMM convolution for taps-per-phase K=1 (reduction block of 32 input channels).
#include "nc_conv_kernel.hip.h"
NC_INSTANTIATE_CONV_K(1, 32)
