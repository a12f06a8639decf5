// DUO instances of the k = 7 residual-unit convolution (nc_conv_kernel.hip.h "DUO"): two tiles per workgroup of 8 wavefronts, the second
// sub-workgroup half a reduction block behind the first, XV-only staging.
#include "nc_conv_kernel.hip.h"
NC_INSTANTIATE_CONV_DUO(duo_k7, 7, 8, 10, 0)
