// Instantiates the implicit-GEMM convolution for taps-per-phase K=7 (reduction block of 8 input channels,
// up to 10 prefetched window words per lane).
#include "nc_conv_kernel.hip.h"
NC_INSTANTIATE_CONV_K(7, 8, 10)

#ifdef NC_DBG_TRACE
namespace nc {
__device__ unsigned long long nc_dbg_buf[NC_TR_WAVES * NC_TR_STAMPS];
__device__ unsigned int nc_dbg_count;
}
using nc::nc_dbg_buf;
using nc::nc_dbg_count;
extern "C" __attribute__((visibility("default"))) int nc_dbg_trace_read(unsigned long long* dst, unsigned int* count, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(dst, HIP_SYMBOL(nc_dbg_buf), sizeof(unsigned long long) * NC_TR_WAVES * NC_TR_STAMPS);
    hipMemcpyFromSymbol(count, HIP_SYMBOL(nc_dbg_count), sizeof(unsigned int));
    if (reset) { unsigned int z = 0; hipMemcpyToSymbol(HIP_SYMBOL(nc_dbg_count), &z, sizeof(z)); }
    return 0;
}
#endif
NC_INSTANTIATE_CONV_NARROW(7, 8, 10)
NC_INSTANTIATE_CONV_SLIM(7, 4, 5)
