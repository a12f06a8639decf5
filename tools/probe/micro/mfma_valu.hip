// Micro-benchmark behind DESIGN 8 (round 4): what an instruction issued between matrix-core instructions costs on gfx950.
// One iteration = 6 independent v_mfma_f32_32x32x2_f32 (the k = 7 template's step at 96 x 64 per wave) + N extra instructions of one
// kind spread between them: vector ALU (v_fma_f32), LDS read (ds_read_b32) or scalar ALU (s_add_u32).  256 workgroups (one per CU) of
// 4 or 8 waves = 1 or 2 waves per SIMD; reports the wall-clock rate of the whole launch (the clock is warmed up first).
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value mfma_valu.hip -o mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int KIND, int N>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    __shared__ float lds[1024];
    lds[threadIdx.x] = threadIdx.x;
    lds[threadIdx.x + 512] = threadIdx.x;
    __syncthreads();
    f32x16 acc[6];
    for (int i = 0; i < 6; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + a, v[8] = {a, b, a + 1, b + 1, a + 2, b + 2, a + 3, b + 3};
    const unsigned la = (threadIdx.x & 255) * 4;
    unsigned sa = 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < (N * (i + 1)) / 6 - (N * i) / 6; ++j) {
                if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(i + j) & 7]) : "v"(b), "v"(a));
                if constexpr (KIND == 1) asm volatile("ds_read_b32 %0, %1" : "=v"(v[(i + j) & 7]) : "v"(la) : "memory");
                if constexpr (KIND == 2) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sa) : : "scc");
            }
        }
        if constexpr (KIND == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = (float)sa;
    for (int i = 0; i < 6; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
static float* g_out;
template <int KIND, int N>
static double run(int waves_per_simd, int iters) {
    const int blocks = 256, threads = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<KIND, N>), dim3(blocks), dim3(threads), 0, 0, g_out, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return (double)blocks * (threads / 64) * iters * 6 * 4096.0 / (ms * 1e-3) / 1e12;
}
template <int KIND>
static void sweep(const char* name, int w) {
    const int it = 20000;
    const double t0 = run<KIND, 0>(w, it);
    printf("%-10s %d wave(s)/SIMD  TFLOP/s at N = 0 / 3 / 6 / 12 / 24 extra per 6 MFMA: %6.1f %6.1f %6.1f %6.1f %6.1f\n", name, w, t0, run<KIND, 3>(w, it),
           run<KIND, 6>(w, it), run<KIND, 12>(w, it), run<KIND, 24>(w, it));
}
int main() {
    hipMalloc(&g_out, (size_t)256 * 512 * 4);
    for (int i = 0; i < 12; ++i) run<0, 0>(2, 20000);   // ~80 ms of matrix work: the clock governor settles
    for (int w = 1; w <= 2; ++w) { sweep<0>("vector ALU", w); sweep<1>("LDS read", w); sweep<2>("scalar ALU", w); }
    return 0;
}
