// The first down-convolution of the Encodec 48 kHz encoder in streaming form (SEANetEncoder.cs: [ResnetBlock, ELU, SConv1d(C -> 2C, k = 4,
// stride 2)]; SConv1d.cs:144-173: non-causal reflect pad 1 + 1):
//     y = conv_{k4,s2}( pad( ELU( GN_s(s) + GN_y(y_branch) ) ) )
// where (s, y_branch) are the two pending outputs of the residual block in front of it (SEANetResnetBlock.cs:72-85).  32 -> 64 channels at
// 48000 -> 24000 steps x 32 clips: 12.6 GFLOP over 590 MB -- on the HBM side of the machine balance; the windowed template's two-input
// instance staged every element through LDS behind a barrier per 8 channels and ran at 2.1 TB/s (276 us).  Here, as in conv3_stream_kernel
// / res_a_kernel, the B fragments never touch LDS: a lane owns two adjacent INPUT columns (2t, 2t+1) of its channel row for BOTH operands
// (one 8-byte load each per channel pair), normalises, adds and activates once per element in registers, and the four taps of output
// column t -- x[2t-1], x[2t], x[2t+1], x[2t+2] -- are its own two values, the left lane's second and the right lane's first (DPP shifts;
// one halo value per 64-column span; reflect as an in-lane fix).  kk = ci*4 + k ascending -- the canonical chain: a channel pair (c0 =
// lanes 0-31, c1 = lanes 32-63) feeds four matrix-core steps
//     (c0,k0 | c0,k1)   (c0,k2 | c0,k3)   (c1,k0 | c1,k1)   (c1,k2 | c1,k3)
// so each half needs two values of the other half's channel per step pair: four v_permlane32_swap per channel pair.  The weight image is
// the one ConvLayer::build packs for the layer ([n_cb][8 channels x 4 taps][64 rows]); it streams through LDS double-buffered.  Epilogue:
// GroupNorm block sums in the canonical order with the in-launch finish, bias, stores.  Bit-identical to the windowed two-input launch
// (NC_NO_DOWN2=1 runs that; tests/test_encodec_gpu.py holds both to the C oracle).
#include <type_traits>
#include <utility>

#include "nc_conv.h"
#include "nc_frag.h"
#include "nc_gn.h"
#include "nc_math.h"

namespace nc {

typedef float d2_f32x16 __attribute__((ext_vector_type(16)));
typedef float d2_f32x4 __attribute__((ext_vector_type(4)));
typedef float d2_f32x2 __attribute__((ext_vector_type(2)));

template <int N, class F, int... I>
__device__ __forceinline__ void d2_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void d2_static_for(F&& f) {
    d2_static_for_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}
__device__ __forceinline__ float d2_from_left(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float d2_from_right(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ float d2_other_half(float v, int hi) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(hi ? r[0] : r[1]);
}

// TM: 32-row tiles of the output (Cout / 32, one row tile per workgroup set: Cout <= 128).  XV2: rows 8-byte aligned at even columns.
template <int TM, bool XV2>
__global__ __launch_bounds__(256, 4) void down2_kernel(const Down2Args p) {
    constexpr int CB = 8, K = 4, BM = 32 * TM;
    constexpr int A_FLOATS = CB * K * BM, A_VEC = A_FLOATS / 4, NA = (A_VEC + 255) / 256;
    constexpr int PF = 4;                          // channel pairs in flight (= the pairs of a block)

    __shared__ __attribute__((aligned(16))) float As[2][A_FLOATS];
    __shared__ float Ep[BM];
    __shared__ float4 Gt[128];                     // (gamma_a, beta_a, gamma_b, beta_b) per input channel (Cin <= 128)

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = gridDim.x, bid = blockIdx.x;
    int lin;
    {
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int t_tile = __builtin_amdgcn_readfirstlane(lin % p.n_t_tiles);
    const int b = __builtin_amdgcn_readfirstlane(lin / p.n_t_tiles);
    const int T = p.T, Tout = p.Tout, n_cb = p.n_cb, Cin = p.Cin;
    const bool gn_in = p.stats_a != nullptr;
    for (int i = tid; i < BM; i += 256) Ep[i] = p.bias ? p.bias[min(i, p.Cout - 1)] : 0.0f;
    float mu_a = 0.0f, rs_a = 1.0f, mu_b = 0.0f, rs_b = 1.0f;
    if (gn_in) {
        mu_a = p.stats_a[2 * b]; rs_a = p.stats_a[2 * b + 1];
        mu_b = p.stats_b[2 * b]; rs_b = p.stats_b[2 * b + 1];
        for (int i = tid; i < n_cb * CB; i += 256) {
            const int c = min(i, Cin - 1);
            Gt[i] = make_float4(p.gamma_a[c], p.beta_a[c], p.gamma_b[c], p.beta_b[c]);
        }
    }
    const unsigned x_cstride = (unsigned)p.x_cstride;
    const int ocol0 = t_tile * 128 + wave * 32;                    // first OUTPUT column of this wave's span
    const int ocol = ocol0 + l31;                                  // this lane's output column
    const int col0 = 2 * ocol0;                                    // first input column of the span (64 of them)
    const int col = col0 + 2 * l31;                                // this lane's two input columns: col, col + 1
    const int colc = min(col, T - 2);
    const int hcol = min(max(l31 < 16 ? col0 - 1 : col0 + 64, 0), T - 1);
    const float* const xa = p.xa + (int64_t)b * p.x_bstride;
    const float* const xb = p.xb + (int64_t)b * p.x_bstride;
    const unsigned x_lane_off = (unsigned)hi * x_cstride + (unsigned)colc;
    const unsigned h_lane_off = (unsigned)hi * x_cstride + (unsigned)hcol;
    const d2_f32x4* const wbase = reinterpret_cast<const d2_f32x4*>(p.w);
    const bool first_col = col == 0, last_col = col + 2 == T;     // reflect: x[-1] = x[1], x[T] = x[T-2]
    const bool lane_first = l31 == 0, lane_last = l31 == 31;

    d2_f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    d2_f32x2 qa[PF], qb[PF];
    float ha[PF], hb[PF];
    const int last_pair = Cin / 2 - 1;
    auto load_pair = [&](int g, d2_f32x2& va, d2_f32x2& vb, float& h_a, float& h_b) __attribute__((always_inline)) {
        const size_t ro = (size_t)(2 * min(g, last_pair)) * x_cstride;
        if constexpr (XV2) {
            va = *reinterpret_cast<const d2_f32x2*>(xa + ro + x_lane_off);
            vb = *reinterpret_cast<const d2_f32x2*>(xb + ro + x_lane_off);
        } else {
            va[0] = xa[ro + x_lane_off]; va[1] = xa[ro + x_lane_off + 1];
            vb[0] = xb[ro + x_lane_off]; vb[1] = xb[ro + x_lane_off + 1];
        }
        h_a = xa[ro + h_lane_off];
        h_b = xb[ro + h_lane_off];
    };
#pragma unroll
    for (int u = 0; u < PF; ++u) load_pair(u, qa[u], qb[u], ha[u], hb[u]);

    d2_f32x4 ra[NA];
#pragma unroll
    for (int n = 0; n < NA; ++n) {
        const int idx = tid + 256 * n;
        if ((A_VEC % 256 == 0) || idx < A_VEC) reinterpret_cast<d2_f32x4*>(As[0])[idx] = wbase[idx];
    }
    __syncthreads();

    for (int cb = 0; cb < n_cb; ++cb) {
        const int cur = cb & 1;
        const bool more = cb + 1 < n_cb;
        if (more) {
            const d2_f32x4* src = wbase + (size_t)(cb + 1) * A_VEC;
#pragma unroll
            for (int n = 0; n < NA; ++n) ra[n] = src[(A_VEC % 256 == 0) ? (unsigned)(tid + 256 * n) : min((unsigned)(tid + 256 * n), (unsigned)(A_VEC - 1))];
        }
        const float* Ac = As[cur] + hi * BM + nc_a_lane_off<TM>(l31);
        // The staged value: GN_a(a) + GN_b(b), then ELU (pad_act_kernel's arithmetic: normalise each operand, add, activate), evaluated on
        // packed pairs -- the lane's two columns of a channel pair, and the halo values of TWO channel pairs together.
        d2_static_for<CB / 4>([&](auto dt) __attribute__((always_inline)) {
            constexpr int du = decltype(dt)::value;
            const int g0 = cb * (CB / 2) + 2 * du;
            const float4 gt0 = gn_in ? Gt[2 * g0 + hi] : make_float4(1.0f, 0.0f, 1.0f, 0.0f);
            const float4 gt1 = gn_in ? Gt[2 * (g0 + 1) + hi] : make_float4(1.0f, 0.0f, 1.0f, 0.0f);
            nc_f2 hva = {ha[(2 * du) % PF], ha[(2 * du + 1) % PF]}, hvb = {hb[(2 * du) % PF], hb[(2 * du + 1) % PF]};
            if (gn_in) {
                hva = ((hva - mu_a) * rs_a) * (nc_f2){gt0.x, gt1.x} + (nc_f2){gt0.y, gt1.y};
                hvb = ((hvb - mu_b) * rs_b) * (nc_f2){gt0.z, gt1.z} + (nc_f2){gt0.w, gt1.w};
            }
            const nc_f2 hvv = nc_eluf2(hva + hvb);
            d2_static_for<2>([&](auto pt) __attribute__((always_inline)) {
                constexpr int pr = 2 * du + decltype(pt)::value;
                const int g = cb * (CB / 2) + pr;
                const float4 gt = decltype(pt)::value ? gt1 : gt0;
                nc_f2 va = qa[pr % PF], vb = qb[pr % PF];
                load_pair(g + PF, qa[pr % PF], qb[pr % PF], ha[pr % PF], hb[pr % PF]);
                if (gn_in) {
                    va = ((va - mu_a) * rs_a) * gt.x + gt.y;
                    vb = ((vb - mu_b) * rs_b) * gt.z + gt.w;
                }
                const nc_f2 ev = nc_eluf2(va + vb);
                const float a = ev[0], bb = ev[1], hv = hvv[decltype(pt)::value];
                float aL = d2_from_left(bb), bR = d2_from_right(a);
                aL = lane_first ? hv : aL;
                bR = lane_last ? hv : bR;
                aL = first_col ? bb : aL;                          // reflect pad (SConv1d.cs:258-274)
                bR = last_col ? a : bR;
                // the other half's channel: taps of the same output column
                const float xaL = d2_other_half(aL, hi), xa_ = d2_other_half(a, hi), xbb = d2_other_half(bb, hi), xbR = d2_other_half(bR, hi);
                // step 0: (c0,k0 | c0,k1)   step 1: (c0,k2 | c0,k3)   step 2: (c1,k0 | c1,k1)   step 3: (c1,k2 | c1,k3)
                const float s0 = hi ? xa_ : aL;
                const float s1 = hi ? xbR : bb;
                const float s2 = hi ? a : xaL;
                const float s3 = hi ? bR : xbb;
                float fa[TM];
                nc_load_a_frag<TM>(Ac + 2 * (4 * pr) * BM, l31, fa);
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], s0, acc[i], 0, 0, 0);
                nc_load_a_frag<TM>(Ac + 2 * (4 * pr + 1) * BM, l31, fa);
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], s1, acc[i], 0, 0, 0);
                nc_load_a_frag<TM>(Ac + 2 * (4 * pr + 2) * BM, l31, fa);
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], s2, acc[i], 0, 0, 0);
                nc_load_a_frag<TM>(Ac + 2 * (4 * pr + 3) * BM, l31, fa);
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], s3, acc[i], 0, 0, 0);
            });
        });
        if (more) {
#pragma unroll
            for (int n = 0; n < NA; ++n) {
                const int idx = tid + 256 * n;
                if ((A_VEC % 256 == 0) || idx < A_VEC) reinterpret_cast<d2_f32x4*>(As[cur ^ 1])[idx] = ra[n];
            }
        }
        __syncthreads();
    }

    // ---- epilogue: D[row = (r&3) + 8*(r>>2) + 4*hi][column l31]: one 32x32 block per row tile and wave
    const bool colok = ocol < Tout;
    if (p.gn_part != nullptr) {
        double* const gp = p.gn_part + (int64_t)b * p.gn_nrb * p.gn_ncb * 2;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float vv[16];
            unsigned okm16 = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int R = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                vv[r] = acc[i][r] + Ep[R];
                if (colok && R < p.Cout) okm16 |= 1u << r;
            }
            double s1, s2;
            nc_gn_slot_sums<false>(vv, okm16, s1, s2);
            nc_gn_butterfly(s1, s2);
            const int cbk = ocol0 >> 5;
            if (lane == 0 && i < p.gn_nrb && cbk < p.gn_ncb) nc_gn_store_partial(gp + ((int64_t)i * p.gn_ncb + cbk) * 2, s1, s2);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (p.gn_count != nullptr)
            nc_gn_arrive_and_finish(gp, p.gn_count + b, p.gn_stats + 2 * b, p.gn_nrb * p.gn_ncb, (unsigned)p.n_t_tiles, p.gn_n);
    }
    if (!colok) return;
    float* const yt = p.y + (int64_t)b * p.y_bstride + (unsigned)(4 * hi) * (unsigned)p.y_cstride + (unsigned)ocol;
    const int rows_left = p.Cout - 4 * hi;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int R = i * 32 + (r & 3) + 8 * (r >> 2);
            if (R >= rows_left) continue;
            yt[(size_t)R * (unsigned)p.y_cstride] = acc[i][r] + Ep[R + 4 * hi];
        }
}

bool launch_down2(const Down2Args& a, int TM, bool aligned, hipStream_t stream) {
    void (*fn)(const Down2Args) = nullptr;
    if (TM == 2) fn = aligned ? &down2_kernel<2, true> : &down2_kernel<2, false>;
    if (!fn) return false;
    hipLaunchKernelGGL(fn, dim3((unsigned)((int64_t)a.B * a.n_t_tiles)), dim3(256), 0, stream, a);
    NC_HIP(hipGetLastError());
    return true;
}

}  // namespace nc
