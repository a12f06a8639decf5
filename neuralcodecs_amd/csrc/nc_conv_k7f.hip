// Instantiates the fused residual-unit kernels: k=7 dilated conv + Snake + 1x1 conv + skip (ResidualUnit.cs:24-59).
#include "nc_conv_kernel.hip.h"
NC_INSTANTIATE_CONV_FUSED(7, 8, 10)
