// GroupNorm(1,C) statistics in the canonical block order (NormConv1d.cs:155; oracle/c/nc_ref_encodec.c gn_sums / block_sums).
//
// The normalised tensor is viewed as the matrix the matrix-core kernels emit it from -- rows R = c*sub + (t % sub), columns q = t / sub
// (sub = 1, or the stride of a sub-pixel transposed convolution) -- cut into 32x32 blocks.  Inside a block, slot i = 32*h + c adds the
// 16 elements of column c in rows 8*j + 4*h + k (ascending) in binary64 from +0, and the 64 slot sums meet in the xor butterfly
// 1,2,4,8,16,32 (p_i <- p_i + p_{i^off}): exactly the accumulator layout of v_mfma_f32_32x32x2_f32 (lane = slot, register r = row
// (r&3) + 8*(r>>2) + 4*h), so the convolution kernels reduce each block they have just computed in registers and write ONE (S1, S2)
// pair per block -- the tensor is never read back for its statistics.  The butterfly runs low offsets first so that every level is a
// register-to-register lane exchange: quad permutes (1, 2), half-row / row mirrors (4, 8: the lanes of a quad / half row already hold
// equal sums, so the mirror image IS the xor partner's value), v_permlane16_swap / v_permlane32_swap (16, 32) -- no LDS crossbar.
// gn_final_kernel adds the block sums of a sample (64 strided slots + the same butterfly).
#pragma once
#include <hip/hip_runtime.h>

namespace nc {

template <int CTRL>
__device__ __forceinline__ double nc_gn_dpp(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_mov_dpp((int)(unsigned)(unsigned long long)b, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_mov_dpp((int)(unsigned)((unsigned long long)b >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
// v[i] + v[i ^ 16] (SWAP32 = false) / v[i] + v[i ^ 32] (true): the swap leaves (own, partner) in some order in every lane
template <bool SWAP32>
__device__ __forceinline__ double nc_gn_swap_add(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
    unsigned l0, l1, h0, h1;
    if constexpr (SWAP32) {
        auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        l0 = rl[0]; l1 = rl[1]; h0 = rh[0]; h1 = rh[1];
    } else {
        auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        l0 = rl[0]; l1 = rl[1]; h0 = rh[0]; h1 = rh[1];
    }
    const double a = __longlong_as_double((long long)(((unsigned long long)h0 << 32) | l0));
    const double c = __longlong_as_double((long long)(((unsigned long long)h1 << 32) | l1));
    return a + c;
}
// levels 1, 2, 4, 8 of the butterfly (inside a 16-lane row)
__device__ __forceinline__ void nc_gn_butterfly_row(double& s1, double& s2) {
    s1 += nc_gn_dpp<0xB1>(s1); s2 += nc_gn_dpp<0xB1>(s2);     // quad_perm [1,0,3,2]: lane ^ 1
    s1 += nc_gn_dpp<0x4E>(s1); s2 += nc_gn_dpp<0x4E>(s2);     // quad_perm [2,3,0,1]: lane ^ 2
    s1 += nc_gn_dpp<0x141>(s1); s2 += nc_gn_dpp<0x141>(s2);   // row_half_mirror: the other quad of the half row (== lane ^ 4 here)
    s1 += nc_gn_dpp<0x140>(s1); s2 += nc_gn_dpp<0x140>(s2);   // row_mirror: the other half row (== lane ^ 8 here)
}
__device__ __forceinline__ void nc_gn_butterfly(double& s1, double& s2) {
    nc_gn_butterfly_row(s1, s2);
    s1 = nc_gn_swap_add<false>(s1); s2 = nc_gn_swap_add<false>(s2);
    s1 = nc_gn_swap_add<true>(s1); s2 = nc_gn_swap_add<true>(s2);
}

// slot sum over the 16 registers of one accumulator tile; bias[r] is added first (the stored value); FULL: no element predicate
template <bool FULL>
__device__ __forceinline__ void nc_gn_slot_sums(const float (&v)[16], unsigned ok_mask, double& s1, double& s2) {
    s1 = 0.0;
    s2 = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const double d = (FULL || ((ok_mask >> r) & 1u)) ? (double)v[r] : 0.0;
        s1 += d;
        s2 = __builtin_fma(d, d, s2);   // (d*d is exact in binary64: the fused form rounds once, like mul + add)
    }
}

}  // namespace nc
