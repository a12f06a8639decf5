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

// ---- finishing the statistics inside the producing launch ------------------------------------------------------------------------------
// Every workgroup of a launch belongs to ONE sample; after its block sums are stored it arrives on the sample's counter, and the
// LAST workgroup to arrive adds the block sums of the sample (gn_final's order: 64 strided slots + butterfly) and writes (mean, rstd)
// -- no follow-up launch.  Hand-off in the placement-independent form of cdna_hip_programming.md G16 / R1 (the form the persistent LSTM
// uses): the block sums are stored WRITE-THROUGH (relaxed agent-scope stores, sc1), each wave drains them (vmcnt(0)) before the
// workgroup barrier, ONE relaxed agent-scope fetch-add per workgroup publishes the arrival, and the last arriver reads the sums with
// agent-scope (sc1) loads, which bypass its CU's L1.  The counter is reset by the last arriver (nothing touches it again before the
// next launch on this stream), so it never needs a memset.  The (mean, rstd) pair is consumed by the NEXT kernel: a plain store.
typedef __attribute__((address_space(1))) unsigned nc_gn_gu32;
__device__ __forceinline__ void nc_gn_store_partial(double* q, double s1, double s2) {
    __hip_atomic_store(q, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 1, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The last arriver's part: the sample's n block sums -> (mean, rstd), counter back to zero.  Called by one whole wavefront.
__device__ __forceinline__ void nc_gn_finish_sample(const double* part, unsigned* counter, float* stats, int n, double count);
// Called by EVERY thread of the workgroup (it holds barriers), after the writer lanes have issued nc_gn_store_partial.
// part: the sample's block sums [n][2]; n_wg: workgroups of this launch that belong to the sample; count = C*T elements.
__device__ __forceinline__ void nc_gn_arrive_and_finish(const double* part, unsigned* counter, float* stats, int n, unsigned n_wg, double count) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's block sums have reached memory
    __syncthreads();
    if (threadIdx.x >= 64) return;                     // the first wavefront arrives for the workgroup (and finishes, if it is the last)
    unsigned prev = 0;
    if (threadIdx.x == 0) prev = __hip_atomic_fetch_add((nc_gn_gu32*)counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    prev = __builtin_amdgcn_readfirstlane(prev);
    if (prev + 1 != n_wg) return;
    nc_gn_finish_sample(part, counter, stats, n, count);
}
// Flattened column axis: a workgroup's tile covers up to 4 consecutive samples and has written inc[m] block sums of sample m (of the
// samples part0 / counter0 / stats0 point at); arrivals are counted in BLOCK SUMS, a sample is complete at n of them.
__device__ __forceinline__ void nc_gn_arrive_blocks(const double* part0, unsigned* counter0, float* stats0, int n, double count, unsigned inc0, unsigned inc1,
                                                    unsigned inc2, unsigned inc3) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    const unsigned inc = lane == 0 ? inc0 : lane == 1 ? inc1 : lane == 2 ? inc2 : lane == 3 ? inc3 : 0u;
    unsigned prev = 0;
    if (inc) prev = __hip_atomic_fetch_add((nc_gn_gu32*)(counter0 + lane), inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long done = __builtin_amdgcn_ballot_w64(inc != 0 && prev + inc == (unsigned)n);
    while (done) {   // (wave-uniform)
        const int m = __builtin_ctzll(done);
        done &= done - 1;
        nc_gn_finish_sample(part0 + (size_t)m * n * 2, counter0 + m, stats0 + 2 * m, n, count);
    }
}
__device__ __forceinline__ void nc_gn_finish_sample(const double* part, unsigned* counter, float* stats, int n, double count) {
    const int lane = threadIdx.x & 63;
    // count < 0 (host: NC_SYNC_ACQUIRE=1, gn_count_arg in nc_conv.h): an agent-scope acquire fence between the arrival that made this wave
    // the last one and its reads of the block sums -- the textbook form of the hand-off.  The default relies on the reads being
    // agent-scope (sc1) loads of write-through stores that were drained before the producers' arrivals (validated: parity sweeps, the
    // fallback-switch suite runs both forms); the fence costs the last arriver ~1.7 us per sample and launch.
    if (count < 0.0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        count = -count;
    }
    double s1 = 0.0, s2 = 0.0;
    for (int k0 = 0; k0 < n; k0 += 64 * 4) {
        double a[4], c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + lane + 64 * u;
            a[u] = k < n ? __hip_atomic_load(part + 2 * (size_t)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
            c[u] = k < n ? __hip_atomic_load(part + 2 * (size_t)k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { s1 += a[u]; s2 += c[u]; }
    }
    nc_gn_butterfly(s1, s2);
    if (lane == 0) {
        const double mu = s1 / count;
        double var = s2 / count - mu * mu;
        if (var < 0.0) var = 0.0;
        stats[0] = (float)mu;
        stats[1] = (float)(1.0 / sqrt(var + 1e-5));
        __hip_atomic_store((nc_gn_gu32*)counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace nc
