// Two-input (shortcut + branch) variants of the Encodec input mode: strided down-convolutions k = 10 / 16 (SEANetEncoder.cs ratios 5 / 8).
#include "nc_conv_kernel.hip.h"
NC_INSTANTIATE_CONV_IN2(in2_k10, 10, 3, 18, false)
NC_INSTANTIATE_CONV_IN2(in2_k16, 16, 2, 18, false)
