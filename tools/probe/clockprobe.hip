// Core-clock probe: one wave spins for `spin_us` microseconds and reports how many shader-clock ticks (s_memtime) passed per
// 100 MHz wall tick (s_memrealtime).  Run concurrently with the codec's kernels (its own stream) to read the clock the matrix
// cores actually run at under that load -- the denominator of any "fraction of MFMA peak" statement.
#include <hip/hip_runtime.h>
#include <cstdint>
extern "C" __global__ void clock_probe_kernel(uint64_t* out, uint64_t spin_ticks) {
    const uint64_t w0 = wall_clock64(), c0 = clock64();
    uint64_t w = w0;
    while (w - w0 < spin_ticks) { __builtin_amdgcn_s_sleep(8); w = wall_clock64(); }
    const uint64_t c1 = clock64();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w - w0; }
}
extern "C" __attribute__((visibility("default"))) int clock_probe_launch(uint64_t* dev_out, uint64_t spin_ticks, void* stream) {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dev_out, spin_ticks);
    return (int)hipGetLastError();
}
