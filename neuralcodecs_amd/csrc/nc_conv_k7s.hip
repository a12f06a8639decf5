// Instantiates the wave-specialised k=7 variants (4 consumer + 2 producer waves; 20 window items per producer wave).
#include "nc_conv_kernel.hip.h"
NC_INSTANTIATE_CONV_SPEC(7, 8, 20)
