// Device code of the implicit-GEMM convolution (included by the per-K instantiation units).
#pragma once
#include <hip/hip_runtime.h>

#include "nc_conv.h"
#include "nc_math.h"

#include <type_traits>
#include <utility>

namespace nc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int N, class F, int... I>
__device__ __forceinline__ void nc_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
// f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N-1>{}), in order
template <int N, class F>
__device__ __forceinline__ void nc_static_for(F&& f) {
    nc_static_for_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}
typedef __attribute__((address_space(1))) const void* nc_gptr;
typedef __attribute__((address_space(3))) void* nc_lptr;

// Block = 4 waves (256 threads).  Wave w owns all BM = 32*TM output channels of the tile and the
// 32*TN columns [w*32*TN, (w+1)*32*TN).  The reduction runs over blocks of CB input channels
// (KB = CB*K flattened kk = ci*K + k, ascending: the canonical chain order).  Per block the
// workgroup holds in LDS, double-buffered,
//   As[kk][BM]   packed weights (a linear copy of the pre-packed global image)
//   Xs[ci][...]  the input window of the tile (Snake applied on the way in), de-interleaved by
//                stride phase so the 32 lanes of an MFMA B-fragment always read consecutive words
// Software pipeline, one barrier per reduction block: the KB/2 matrix-core steps of block cb are cut
// into NSEG segments; the global loads of block cb+1 are issued in NSEG-1 groups, group g at the head
// of segment g, and group g is transformed (Snake) and written to the other LDS buffer at the head of
// segment g+1 -- so a load has a whole segment of MFMA time to land, only 1/(NSEG-1) of the staging
// registers are live at once, and the VALU work sits between matrix-core segments.
//   MFMA step kp: lane l supplies A[row = l&31][kk = 2*kp + (l>>5)] and B[kk][col = l&31].
template <int TM, int TN, int K, int CB, int NX, bool FUSE = false>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvArgs p) {
    constexpr int BM = 32 * TM;
    constexpr int BNW = 32 * TN;
    constexpr int BN = 4 * BNW;
    constexpr int KB = CB * K;
    constexpr int KP = KB / 2;
    constexpr int A_FLOATS = KB * BM;
    constexpr int A_VEC = A_FLOATS / 4;            // float4 words in the weight tile
    constexpr int NA = (A_VEC + 255) / 256;        // float4 copies per thread
    constexpr int NSEG = KP >= 16 ? 4 : 2;
    constexpr int NG = NSEG - 1;
    constexpr int GA = (NA + NG - 1) / NG;         // per-group register footprint
    constexpr int GX = (NX + NG - 1) / NG;
    static_assert(KB % 2 == 0, "reduction block must hold an even number of kk");
    static_assert(A_FLOATS % 4 == 0, "A tile must be 16-byte granular");

    extern __shared__ __attribute__((aligned(16))) float smem[];

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

    const f32x4* const wbase =
        reinterpret_cast<const f32x4*>(p.w + (int64_t)phase * p.w_phase_stride + (int64_t)co_tile * p.n_cb * A_FLOATS);
    const float* const xb = p.x + (int64_t)b * p.x_bstride;

    float* const As0 = smem;
    float* const Xs0 = smem + 2 * A_FLOATS;

    // kernel arguments used inside the pipeline, pinned in registers
    const int nchunk = p.nchunk, chunk_magic = p.chunk_magic, stride_magic = p.stride_magic;
    const int xwp = p.xwp, xrow = p.xrow, xbuf = p.xbuf, Cin = p.Cin, x_len = p.x_len, n_cb = p.n_cb;
    const unsigned x_cstride = (unsigned)p.x_cstride;
    const float* const alpha_in = p.alpha_in;
    int tap[K];  // window slot of tap k (wave-uniform scalars)
#pragma unroll
    for (int k = 0; k < K; ++k) tap[k] = p.tapoff[k];
    // Snake alphas of all input channels, staged once per block behind the tile buffers
    float2* const Al = reinterpret_cast<float2*>(Xs0 + 2 * xbuf);   // (alpha, 1/alpha) per input channel
    if (alpha_in != nullptr)
        for (int i = tid; i < n_cb * CB; i += 256) {
            const float al = alpha_in[min(i, Cin - 1)];
            Al[i] = make_float2(al, nc_snake_inv(al));
        }
    __syncthreads();

    // ---- staging (branch-free; every address is clamped into the tensor) ------------------------
    // weights: thread t copies float4 words t + 256*n; input window: wave-level items of 64 consecutive
    // window slots of one channel, item -> wave item%4; items past n_items land in pad rows.
    f32x4 ra[GA];
    float rx[GX];
    auto issue_group = [&](int cbn, auto gtag) __attribute__((always_inline)) {
        constexpr int g = decltype(gtag)::value;
        const f32x4* src = wbase + (size_t)cbn * A_VEC;
        nc_static_for<GA>([&](auto ut) __attribute__((always_inline)) {
            constexpr int u = decltype(ut)::value, n = g * GA + u;
            if constexpr (n < NA) {
                const unsigned idx = (unsigned)(tid + 256 * n);
                ra[u] = src[(A_VEC % 256 == 0) ? idx : min(idx, (unsigned)(A_VEC - 1))];
            }
        });
        nc_static_for<GX>([&](auto ut) __attribute__((always_inline)) {
            constexpr int u = decltype(ut)::value, i = g * GX + u;
            if constexpr (i < NX) {
                const int item = wave + 4 * i;
                const int c = (item * chunk_magic) >> 20;
                const int ci = min(cbn * CB + c, Cin - 1);
                const int gp = xs0 + (item - c * nchunk) * 64 + lane;
                rx[u] = xb[(unsigned)ci * x_cstride + (unsigned)min(max(gp, 0), x_len - 1)];
            }
        });
    };
    auto store_group = [&](int cbn, float* Ad, float* Xd, auto gtag, auto snake_tag) __attribute__((always_inline)) {
        constexpr int g = decltype(gtag)::value;
        constexpr bool SNAKE = decltype(snake_tag)::value;
        nc_static_for<GA>([&](auto ut) __attribute__((always_inline)) {
            constexpr int u = decltype(ut)::value, n = g * GA + u;
            if constexpr (n < NA) {
                const int idx = tid + 256 * n;
                if ((A_VEC % 256 == 0) || idx < A_VEC) reinterpret_cast<f32x4*>(Ad)[idx] = ra[u];
            }
        });
        float2 al[GX];
        if (SNAKE) {
            nc_static_for<GX>([&](auto ut) __attribute__((always_inline)) {
                constexpr int u = decltype(ut)::value, i = g * GX + u;
                if constexpr (i < NX) {
                    const int item = wave + 4 * i;
                    al[u] = Al[cbn * CB + ((item * chunk_magic) >> 20)];
                }
            });
        }
        float v[GX];
        int off[GX];
        nc_static_for<GX>([&](auto ut) __attribute__((always_inline)) {
            constexpr int u = decltype(ut)::value, i = g * GX + u;
            if constexpr (i < NX) {
                const int item = wave + 4 * i;
                const int c = (item * chunk_magic) >> 20;
                const int ci = cbn * CB + c;
                const int j = (item - c * nchunk) * 64 + lane;
                const int gp = xs0 + j;
                const bool ok = (ci < Cin) & (gp >= 0) & (gp < x_len);  // slots past xw / items past n_items are never read
                v[u] = ok ? rx[u] : 0.0f;
                if (SNAKE) v[u] = nc_snakef(v[u], al[u].x, al[u].y);
                off[u] = item * 64 + lane;
                if (s != 1) {
                    const int q = (j * stride_magic) >> 20;
                    off[u] = c * xrow + (j - q * s) * xwp + q;
                }
            }
        });
        nc_static_for<GX>([&](auto ut) __attribute__((always_inline)) {
            constexpr int u = decltype(ut)::value, i = g * GX + u;
            if constexpr (i < NX) Xd[off[u]] = v[u];
        });
    };
    auto store_group_any = [&](int cbn, float* Ad, float* Xd, auto gtag) __attribute__((always_inline)) {
        if (alpha_in != nullptr) store_group(cbn, Ad, Xd, gtag, std::true_type{});
        else store_group(cbn, Ad, Xd, gtag, std::false_type{});
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // ---- prologue: stage reduction block 0
    nc_static_for<NG>([&](auto g) __attribute__((always_inline)) {
        issue_group(0, g);
        store_group_any(0, As0, Xs0, g);
    });
    __syncthreads();

    const int a_lane = hi * BM + l31;
    const int x_lane = wave * BNW + l31;

    float fa[2][TM], fb[2][TN];
    auto load_frag = [&](const float* Ac, const float* Xc, auto kp_tag) __attribute__((always_inline)) {
        constexpr int kp = decltype(kp_tag)::value;
        constexpr int c0 = (2 * kp) / K, k0 = (2 * kp) % K, c1 = (2 * kp + 1) / K, k1 = (2 * kp + 1) % K;
        const int o0 = c0 * xrow + tap[k0], o1 = c1 * xrow + tap[k1];
        const int o = hi ? o1 : o0;
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[kp & 1][i] = Ac[2 * kp * BM + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[kp & 1][j] = Xc[o + j * 32];
    };

    for (int cb = 0; cb < n_cb; ++cb) {
        const int cur = cb & 1;
        const float* Ac = As0 + cur * A_FLOATS + a_lane;
        const float* Xc = Xs0 + cur * xbuf + x_lane;
        float* const An = As0 + (cur ^ 1) * A_FLOATS;
        float* const Xn = Xs0 + (cur ^ 1) * xbuf;
        const bool more = cb + 1 < n_cb;
        nc_static_for<NSEG>([&](auto seg_tag) __attribute__((always_inline)) {
            constexpr int seg = decltype(seg_tag)::value;
#if !defined(NC_ABL_NOSTAGE)
            if (more) {
                if constexpr (seg >= 1) store_group_any(cb + 1, An, Xn, std::integral_constant<int, (seg >= 1 ? seg - 1 : 0)>{});
                if constexpr (seg < NG) issue_group(cb + 1, seg_tag);
            }
#endif
            // ---- matrix-core steps of this segment, ascending kk; fragments of step kp+1 are read before
            //      the MFMAs of step kp are issued (register double buffer fa/fb)
            constexpr int kp_lo = seg * KP / NSEG, kp_hi = (seg + 1) * KP / NSEG;
            if constexpr (seg == 0) load_frag(Ac, Xc, std::integral_constant<int, 0>{});
#ifdef NC_EXP_SETPRIO
            __builtin_amdgcn_s_setprio(1);
#endif
            nc_static_for<kp_hi - kp_lo>([&](auto d) __attribute__((always_inline)) {
                constexpr int kp = kp_lo + decltype(d)::value;
#if !defined(NC_ABL_NOFRAG)
                if constexpr (kp + 1 < KP) load_frag(Ac, Xc, std::integral_constant<int, kp + 1>{});
#endif
#ifdef NC_EXP_SCHED
                __builtin_amdgcn_sched_barrier(0);   // keep the fragment reads of step kp+1 ahead of the MFMAs of step kp
#endif
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kp & 1][i], fb[kp & 1][j], acc[i][j], 0, 0, 0);
            });
#ifdef NC_EXP_SETPRIO
            __builtin_amdgcn_s_setprio(0);
#endif
        });
#if !defined(NC_ABL_NOBAR)
        __syncthreads();
#endif
    }

    // ---- epilogue: D[row = (r&3) + 8*(r>>2) + 4*hi][col = l31]
    const int64_t ybase = (int64_t)b * p.y_bstride;
    // store one 32-row block `ib` of the tile: v[j][r] + bias (+ residual) (Snake) (tanh) -> y / RVQ accumulate.
    // Rows go out in quads: all global reads of a quad (residual / RVQ operands) are issued before its first store, so the quad
    // costs one memory round trip -- loads interleaved with possibly-aliasing stores would serialise into one round trip per value.
    auto emit_rows = [&](int ib, const f32x16 (&v)[TN], const float* bias_p, const float* ao_p, const float* res_p) __attribute__((always_inline)) {
        const bool rvq = (p.epi & EPI_RVQ) != 0, noise = (p.epi & EPI_NOISE) != 0;
        nc_static_for<4>([&](auto qt) __attribute__((always_inline)) {
            constexpr int rq = decltype(qt)::value;
            float rv[4][TN], zv[4][TN], sv[4][TN];
            int64_t off[4][TN];
            bool ok[4][TN];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int row = ib * 32 + rr + 8 * rq + 4 * hi;     // D row of register r = 4*rq + rr
                const int co = co_tile * BM + row;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = col0 + wave * BNW + j * 32 + l31;
                    const int t = col * p.y_tstride + p.y_toff + phase;
                    ok[rr][j] = (co < p.Cout) & (col < p.n_cols) & (t >= 0) & (t < p.Tout);
                    off[rr][j] = ybase + (int64_t)co * p.y_cstride + t;
                    rv[rr][j] = (ok[rr][j] && res_p) ? res_p[off[rr][j]] : 0.0f;
                    zv[rr][j] = (ok[rr][j] && rvq) ? p.rvq_zq[off[rr][j]] : 0.0f;
                    sv[rr][j] = (ok[rr][j] && rvq && p.rvq_res) ? p.rvq_res[off[rr][j]] : 0.0f;
                }
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int row = ib * 32 + rr + 8 * rq + 4 * hi;
                const int co = min(co_tile * BM + row, p.Cout - 1);
                const float bias = bias_p ? bias_p[co] : 0.0f;
                const float ao = ao_p ? ao_p[co] : 0.0f;
                const float ao_inv = nc_snake_inv(ao);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (!ok[rr][j]) continue;
                    const int64_t o = off[rr][j];
                    float val = v[j][4 * rq + rr] + bias;
                    if (noise) {
                        const int col = col0 + wave * BNW + j * 32 + l31;
                        const int t = col * p.y_tstride + p.y_toff + phase;
                        val = rv[rr][j] + p.noise[(int64_t)b * p.noise_bstride + t] * val;
                    } else if (res_p) {
                        val = val + rv[rr][j];
                    }
                    if (ao_p) val = nc_snakef(val, ao, ao_inv);
                    if (p.epi & EPI_TANH) val = nc_tanhf(val);
                    if (rvq) {
                        p.rvq_zq[o] = zv[rr][j] + val;
                        if (p.rvq_res) p.rvq_res[o] = sv[rr][j] - val;
                    } else {
                        p.y[o] = val;
                    }
                }
            }
        });
    };

    if constexpr (!FUSE) {
#pragma unroll
        for (int i = 0; i < TM; ++i) emit_rows(i, acc[i], p.bias, p.alpha_out, p.res);
    } else {
        // ---- fused ResidualUnit tail (ResidualUnit.cs:29-34,50-59): this block holds ALL channels of
        //      h_pre = conv7(snake(x)) for its columns (n_co_tiles == 1, Cin == Cout == BM), so
        //          y = x + W1 . snake_a2(h_pre + b7) + b1
        //      is finished here: the activation never leaves the registers.  The 1x1 contraction runs on the
        //      matrix cores with B fragments taken from the accumulators: v_permlane32_swap turns the pair of
        //      D registers holding rows (2m, 2m+1) of one lane half into the B operands of k-steps m and m+2.
        static_assert(!FUSE || K == 7, "the fused tail belongs to the k=7 residual-unit convolution");
        // 1) h = snake(h_pre + b7, a2), in place
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const float bias = p.bias[row], ao = p.alpha_out[row];
                const float ao_inv = nc_snake_inv(ao);
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j][r] = nc_snakef(acc[i][j][r] + bias, ao, ao_inv);
            }
        // 2) accumulator layout -> B-operand layout, in place
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int qe = 0; qe < 8; ++qe) {
                    const int r0 = 4 * (qe >> 1) + 2 * (qe & 1);
                    auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i][j][r0]), __float_as_uint(acc[i][j][r0 + 1]),
                                                               false, false);
                    acc[i][j][r0] = __uint_as_float(sw[0]);      // rows (2m, 2m+1) of the low half : k-step m = 4q+e
                    acc[i][j][r0 + 1] = __uint_as_float(sw[1]);  // rows of the high half           : k-step m+2
                }
        // 3) W1 (packed [row block][ci][32 rows]) -> LDS; the main loop's last barrier freed the tile buffers
        {
            const f32x4* w2 = reinterpret_cast<const f32x4*>(p.w2);
            f32x4* dst = reinterpret_cast<f32x4*>(smem);
#pragma unroll 4
            for (int idx = tid; idx < BM * BM / 4; idx += 256) dst[idx] = w2[idx];
        }
        __syncthreads();
        // 4) y = W1 . h + b1 + x, one 32-row block at a time (ascending ci chain, like the stand-alone 1x1 kernel)
        nc_static_for<TM>([&](auto i2t) __attribute__((always_inline)) {
            constexpr int i2 = decltype(i2t)::value;
            f32x16 acc2[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[j][r] = 0.0f;
            const float* W1s = smem + i2 * BM * 32 + hi * 32 + l31;
            nc_static_for<BM / 2>([&](auto kpt) __attribute__((always_inline)) {
                constexpr int kp = decltype(kpt)::value;
                constexpr int i = kp / 16, m = kp % 16;
                constexpr int reg = 4 * (m >> 2) + 2 * (m & 1) + ((m >> 1) & 1);
                const float a = W1s[2 * kp * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) acc2[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, acc[i][j][reg], acc2[j], 0, 0, 0);
            });
            emit_rows(i2, acc2, p.bias2, p.alpha_out2, p.res);
        });
    }
}

typedef void (*conv_kernel_fn)(const ConvArgs);

template <int TM, int TN, int K, int CB, int NX, bool FUSE = false>
inline conv_kernel_fn get_conv_kernel() {
    return &conv_mfma_kernel<TM, TN, K, CB, NX, FUSE>;
}

}  // namespace nc

// Instantiation helper: one translation unit per K registers its 8 (TM,TN) variants.
#define NC_INSTANTIATE_CONV_K(KVAL, CBVAL, NXVAL)                                                          \
    namespace nc {                                                                                         \
    conv_kernel_fn conv_kernel_table_k##KVAL(int TM, int TN) {                                             \
        switch (TM * 10 + TN) {                                                                            \
            case 11: return get_conv_kernel<1, 1, KVAL, CBVAL, NXVAL>();                                   \
            case 12: return get_conv_kernel<1, 2, KVAL, CBVAL, NXVAL>();                                   \
            case 21: return get_conv_kernel<2, 1, KVAL, CBVAL, NXVAL>();                                   \
            case 22: return get_conv_kernel<2, 2, KVAL, CBVAL, NXVAL>();                                   \
            case 31: return get_conv_kernel<3, 1, KVAL, CBVAL, NXVAL>();                                   \
            case 32: return get_conv_kernel<3, 2, KVAL, CBVAL, NXVAL>();                                   \
            case 41: return get_conv_kernel<4, 1, KVAL, CBVAL, NXVAL>();                                   \
            case 42: return get_conv_kernel<4, 2, KVAL, CBVAL, NXVAL>();                                   \
        }                                                                                                  \
        return nullptr;                                                                                    \
    }                                                                                                      \
    int conv_kernel_cb_k##KVAL() { return CBVAL; }                                                         \
    int conv_kernel_nx_k##KVAL() { return NXVAL; }                                                         \
    }

// Fused residual-unit variants (k=7 conv + Snake + 1x1 conv + skip in one launch): Cin == Cout == 32*TM.
#define NC_INSTANTIATE_CONV_FUSED(KVAL, CBVAL, NXVAL)                                                      \
    namespace nc {                                                                                         \
    conv_kernel_fn conv_kernel_table_fused_k##KVAL(int TM, int TN) {                                       \
        switch (TM * 10 + TN) {                                                                            \
            case 21: return get_conv_kernel<2, 1, KVAL, CBVAL, NXVAL, true>();                             \
            case 22: return get_conv_kernel<2, 2, KVAL, CBVAL, NXVAL, true>();                             \
            case 31: return get_conv_kernel<3, 1, KVAL, CBVAL, NXVAL, true>();                             \
            case 32: return get_conv_kernel<3, 2, KVAL, CBVAL, NXVAL, true>();                             \
            case 41: return get_conv_kernel<4, 1, KVAL, CBVAL, NXVAL, true>();                             \
            case 42: return get_conv_kernel<4, 2, KVAL, CBVAL, NXVAL, true>();                             \
        }                                                                                                  \
        return nullptr;                                                                                    \
    }                                                                                                      \
    }
