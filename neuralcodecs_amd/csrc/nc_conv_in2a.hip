// Two-input (shortcut + branch) variants of the Encodec input mode: strided down-convolutions k = 4 / 8 (SEANetEncoder.cs ratios 2 / 4).
#include "nc_conv_kernel.hip.h"
NC_INSTANTIATE_CONV_IN2(in2_k4, 4, 8, 18, false)
NC_INSTANTIATE_CONV_IN2(in2_k8, 8, 4, 18, false)
