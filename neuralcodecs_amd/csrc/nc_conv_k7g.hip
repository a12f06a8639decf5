// Instantiates the wide fused residual-unit kernels (C = 192 / 256: whole-channel tiles, W1 streamed; ResidualUnit.cs:24-59).
#include "nc_conv_kernel.hip.h"
NC_INSTANTIATE_CONV_FUSED_WIDE(7, 4, 5)
