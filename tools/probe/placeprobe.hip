// Build: hipcc --offload-arch=gfx950 -O2 -o tools/probe/placeprobe tools/probe/placeprobe.hip   (the binary is git-ignored; it travels with gpurun)
// Workgroup placement probe: every workgroup of a grid records (XCC_ID, HW_ID) and spins for `spin_ticks` of the 100 MHz wall clock, so that
// all workgroups of the grid are resident at once.  Answers: how many workgroups share a CU at a given grid size / LDS request?
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ void place_kernel(uint32_t* out, uint64_t spin_ticks) {
    extern __shared__ float lds[];
    const uint64_t w0 = wall_clock64();
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; lds[0] = 1.0f; }
    while (wall_clock64() - w0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
}
int main(int argc, char** argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 256;
    const int lds = argc > 2 ? atoi(argv[2]) : 68608;
    hipFuncSetAttribute((const void*)place_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int a = 3; a < argc; ++a) {
        const int grid = atoi(argv[a]);
        uint32_t* d;
        hipMalloc(&d, (size_t)grid * 8);
        hipLaunchKernelGGL(place_kernel, dim3(grid), dim3(threads), lds, 0, d, (uint64_t)20000);   // 200 us
        std::vector<uint32_t> h((size_t)grid * 2);
        hipMemcpy(h.data(), d, (size_t)grid * 8, hipMemcpyDeviceToHost);
        std::map<uint32_t, int> per_cu, per_xcc;
        for (int b = 0; b < grid; ++b) {
            const uint32_t xcc = h[2 * b] & 0xf, hw = h[2 * b + 1];
            const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            per_cu[(xcc << 16) | (se << 8) | (sh << 4) | cu]++;
            per_xcc[xcc]++;
        }
        int hist[8] = {0};
        for (auto& kv : per_cu) hist[kv.second < 7 ? kv.second : 7]++;
        printf("threads %d lds %d grid %d: CUs used %zu; CUs with 1/2/3/4+ workgroups: %d %d %d %d; per XCC:", threads, lds, grid, per_cu.size(), hist[1], hist[2],
               hist[3], hist[4] + hist[5] + hist[6] + hist[7]);
        for (auto& kv : per_xcc) printf(" %d", kv.second);
        printf("\n");
        if (a == 3) {
            printf("   block->(xcc,se,sh,cu) of the first 40 blocks:");
            for (int b = 0; b < 40 && b < grid; ++b) printf(" %u:%u.%u.%u", h[2 * b] & 0xf, (h[2 * b + 1] >> 13) & 7, (h[2 * b + 1] >> 12) & 1, (h[2 * b + 1] >> 8) & 0xf);
            printf("\n");
        }
        hipFree(d);
    }
    return 0;
}
