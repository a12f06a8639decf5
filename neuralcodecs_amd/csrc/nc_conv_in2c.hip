// Two-input (shortcut + branch) variants of the Encodec input mode: the up-convolutions (two taps per phase; per-phase and sub-pixel forms,
// SEANetDecoder.cs ratios 8 / 5 / 4 / 2).
#include "nc_conv_kernel.hip.h"
NC_INSTANTIATE_CONV_IN2(in2_k2, 2, 16, 20, false)
NC_INSTANTIATE_CONV_IN2(in2_sub_k2, 2, 16, 20, true)
