// Instantiates the implicit-GEMM convolution for taps-per-phase K=16 (reduction block of 4 input channels).
#include "nc_conv_kernel.hip.h"
NC_INSTANTIATE_CONV_K(16, 4)
