// Instantiates the sub-pixel transposed convolution for strides that are not a power of two (two taps per phase, SUB == 2).
#include "nc_conv_kernel.hip.h"
NC_INSTANTIATE_CONV_SUBG(2, 16, 20)
