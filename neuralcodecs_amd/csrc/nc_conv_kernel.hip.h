// Device code of the implicit-GEMM convolution (included by the per-K instantiation units).
#pragma once
#include <hip/hip_runtime.h>

#include "nc_conv.h"
#include "nc_math.h"

namespace nc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Block = 4 waves (256 threads).  Wave w owns all BM = 32*TM output channels of the tile and the
// 32*TN columns [w*32*TN, (w+1)*32*TN).  Per reduction block of CB input channels the block stages
//   As[kk][BM]   packed weights (linear copy from the pre-packed global image)
//   Xs[ci][...]  the input window of the tile (Snake applied on the way in), de-interleaved by
//                stride phase so the 32 lanes of an MFMA B-fragment always read consecutive words
// and then issues KB/2 steps of v_mfma_f32_32x32x2_f32 per accumulator:
//   lane l supplies A[row = l&31][kk = 2*kp + (l>>5)] and B[kk][col = l&31].
template <int TM, int TN, int K, int CB>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvArgs p) {
    constexpr int BM = 32 * TM;
    constexpr int BNW = 32 * TN;
    constexpr int BN = 4 * BNW;
    constexpr int KB = CB * K;
    constexpr int KP = KB / 2;
    static_assert(KB % 2 == 0, "reduction block must hold an even number of kk");
    static_assert((KB * BM) % 4 == 0, "A tile must be float4-copyable");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Xs = smem + KB * BM;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hi = lane >> 5;

    // ---- XCD-aware block -> tile map: block id b runs on XCD b%8 (observed; speed only).  Give each
    // XCD a contiguous range of the (phase, co_tile, clip, t_tile) order so the blocks resident on
    // one XCD share a weight panel in that XCD's L2.
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    int lin;
    {
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int t_tile = lin % p.n_t_tiles;
    lin /= p.n_t_tiles;
    const int b = lin % p.B;
    lin /= p.B;
    const int co_tile = lin % p.n_co_tiles;
    const int phase = lin / p.n_co_tiles;

    const int col0 = t_tile * BN;
    const int s = p.stride;
    const int xs0 = col0 * s - p.pad - p.xneg;  // global x position of window slot 0

    // per-lane LDS offsets of the B fragments: slot of (ci_local, k) for kk = 2*kp + hi
    int xo[KP];
#pragma unroll
    for (int kp = 0; kp < KP; ++kp) {
        const int kk = 2 * kp + hi;
        const int ci = kk / K, k = kk - ci * K;
        const int q = k * p.dil + p.xneg;
        int o = ci * p.xrow;
        if (s == 1) o += q;
        else o += (q % s) * p.xwp + q / s;
        xo[kp] = o + wave * BNW + l31;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const float* wbase = p.w + (int64_t)phase * p.w_phase_stride + (int64_t)co_tile * p.n_cb * (KB * BM);
    const float* xb = p.x + (int64_t)b * p.x_bstride;

    for (int cb = 0; cb < p.n_cb; ++cb) {
        __syncthreads();
        // ---- stage weights (linear, 16 B per lane)
        {
            const float4* ag = reinterpret_cast<const float4*>(wbase + (int64_t)cb * (KB * BM));
            float4* as4 = reinterpret_cast<float4*>(As);
#pragma unroll
            for (int i = tid; i < KB * BM / 4; i += 256) as4[i] = ag[i];
        }
        // ---- stage the input window (zero outside the clip, Snake fused)
#pragma unroll 1
        for (int c = 0; c < CB; ++c) {
            const int ci = cb * CB + c;
            const bool cok = ci < p.Cin;
            const float* xr = xb + (int64_t)ci * p.x_cstride;
            const float al = (p.alpha_in != nullptr && cok) ? p.alpha_in[ci] : 0.0f;
            float* xd = Xs + c * p.xrow;
            for (int j = tid; j < p.xw; j += 256) {
                const int gp = xs0 + j;
                float v = 0.0f;
                if (cok && gp >= 0 && gp < p.x_len) v = xr[gp];
                if (p.alpha_in != nullptr) v = nc_snakef(v, al);
                const int slot = (s == 1) ? j : (j % s) * p.xwp + j / s;
                xd[slot] = v;
            }
        }
        __syncthreads();
        // ---- matrix-core steps, ascending kk
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) {
            float a[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[(2 * kp + hi) * BM + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = Xs[xo[kp] + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bv[j], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: D[row = (r&3) + 8*(r>>2) + 4*hi][col = l31]
    const int64_t ybase = (int64_t)b * p.y_bstride;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            const int co = co_tile * BM + row;
            if (co >= p.Cout) continue;
            const float bias = p.bias ? p.bias[co] : 0.0f;
            const float ao = p.alpha_out ? p.alpha_out[co] : 0.0f;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = col0 + wave * BNW + j * 32 + l31;
                const int t = col * p.y_tstride + p.y_toff + phase;
                if (col >= p.n_cols || t < 0 || t >= p.Tout) continue;
                const int64_t o = ybase + (int64_t)co * p.y_cstride + t;
                float v = acc[i][j][r] + bias;
                if (p.res) v = v + p.res[o];
                if (p.alpha_out) v = nc_snakef(v, ao);
                if (p.epi & EPI_TANH) v = nc_tanhf(v);
                if (p.epi & EPI_RVQ) {
                    p.rvq_zq[o] = p.rvq_zq[o] + v;
                    if (p.rvq_res) p.rvq_res[o] = p.rvq_res[o] - v;
                } else {
                    p.y[o] = v;
                }
            }
        }
    }
}

typedef void (*conv_kernel_fn)(const ConvArgs);

template <int TM, int TN, int K, int CB>
inline conv_kernel_fn get_conv_kernel() {
    return &conv_mfma_kernel<TM, TN, K, CB>;
}

}  // namespace nc

// Instantiation helper: one translation unit per K registers its 8 (TM,TN) variants.
#define NC_INSTANTIATE_CONV_K(KVAL, CBVAL)                                                                 \
    namespace nc {                                                                                         \
    conv_kernel_fn conv_kernel_table_k##KVAL(int TM, int TN) {                                             \
        switch (TM * 10 + TN) {                                                                            \
            case 11: return get_conv_kernel<1, 1, KVAL, CBVAL>();                                          \
            case 12: return get_conv_kernel<1, 2, KVAL, CBVAL>();                                          \
            case 21: return get_conv_kernel<2, 1, KVAL, CBVAL>();                                          \
            case 22: return get_conv_kernel<2, 2, KVAL, CBVAL>();                                          \
            case 31: return get_conv_kernel<3, 1, KVAL, CBVAL>();                                          \
            case 32: return get_conv_kernel<3, 2, KVAL, CBVAL>();                                          \
            case 41: return get_conv_kernel<4, 1, KVAL, CBVAL>();                                          \
            case 42: return get_conv_kernel<4, 2, KVAL, CBVAL>();                                          \
        }                                                                                                  \
        return nullptr;                                                                                    \
    }                                                                                                      \
    }
