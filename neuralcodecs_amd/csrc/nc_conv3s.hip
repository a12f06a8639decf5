// Streaming k=3 convolution of the Encodec residual branches on the fp32 matrix cores (SEANetResnetBlock.cs:53-70: ELU -> SConv1d(C -> C/2,
// k=3) with the pending GroupNorm of the block input; SConv1d.cs:144-173: non-causal reflect pad 1 + 1).
//
// These layers are thin and long (32 -> 16 channels at 48000 steps x 32 clips): 2*Cin*Cout*3 flops per 4*(Cin + Cout) bytes, far on the
// HBM side of the machine balance.  The windowed template stages every reduction block through LDS behind a workgroup barrier and was
// latency-bound at 1.5 TB/s (199 us where the 1x1 convolution of the same tensor streams in 66 us).  Here, as in conv1x1_kernel, the B
// fragments never touch LDS: a lane owns two adjacent columns of its channel row (one 8-byte load per channel pair, PF pairs in
// flight), applies the pending GroupNorm + ELU ONCE per element in registers, and builds the three taps from its own values, its
// lane neighbours' (wavefront shifts: column t-1 is the left neighbour's second value, t+1 the right neighbour's first) and one halo
// value per 64-column span.  MFMA step kp pairs kk = 2kp (lanes 0-31) with 2kp+1 (lanes 32-63), kk = ci*3 + k ascending -- the
// canonical chain -- so a channel pair (c0 = lanes 0-31, c1 = lanes 32-63) feeds three steps:
//     (c0,k0 | c0,k1)   (c0,k2 | c1,k0)   (c1,k1 | c1,k2)
// i.e. each half needs the CENTRE tap of the other half's channel once: one v_permlane32_swap per value.  Reflect padding touches only
// the first / last column of a clip and is an in-lane fix (x[-1] = x[1], x[T] = x[T-2]; T is even, so both live in the same lane).
// Only the weight tile (shared by the four waves) goes through LDS, double-buffered, one barrier per 16 input channels.
#include <type_traits>
#include <utility>

#include "nc_conv.h"
#include "nc_frag.h"
#include "nc_gn.h"
#include "nc_math.h"

namespace nc {

typedef float c3_f32x16 __attribute__((ext_vector_type(16)));
typedef float c3_f32x4 __attribute__((ext_vector_type(4)));
typedef float c3_f32x2 __attribute__((ext_vector_type(2)));

template <int N, class F, int... I>
__device__ __forceinline__ void c3_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void c3_static_for(F&& f) {
    c3_static_for_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// lane i <- lane i-1 / lane i+1 of the wavefront (DPP wave_shr:1 / wave_shl:1; the lanes shifted in at the ends are fixed by the caller)
__device__ __forceinline__ float c3_from_left(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float c3_from_right(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false));
}
// the value the same lane of the OTHER half holds
__device__ __forceinline__ float c3_other_half(float v, int hi) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(hi ? r[0] : r[1]);
}

constexpr int conv3s_occupancy(int TM) { return TM == 1 ? 5 : TM == 2 ? 4 : 2; }

// XV2: the rows are 8-byte aligned at even columns (one 8-byte load per lane and channel); otherwise two 4-byte loads (a row view that
// starts at an odd sample: the trimmed output of an odd-padded transposed convolution, SConvTranspose1d.cs:159-171)
template <int TM, bool XV2>
__global__ __launch_bounds__(256, conv3s_occupancy(TM)) void conv3_stream_kernel(const ConvArgs p) {
    constexpr int TN = 2, CB = 16, K = 3;
    constexpr int BM = 32 * TM, BNW = 32 * TN, BN = 4 * BNW;
    constexpr int KB = CB * K;                       // 24 matrix-core steps per reduction block = 8 channel pairs x 3
    constexpr int A_FLOATS = KB * BM, A_VEC = A_FLOATS / 4, NA = (A_VEC + 255) / 256;
    constexpr int PF = 4;                            // channel pairs in flight (divides the 8 pairs of a block)

    __shared__ __attribute__((aligned(16))) float As[2][A_FLOATS];
    __shared__ float Ep[BM];
    __shared__ float2 Gt[512];                       // (gamma, beta) of the pending GroupNorm per input channel (Cin <= 512)

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = gridDim.x, bid = blockIdx.x;
    int lin;
    {
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int co_in = __builtin_amdgcn_readfirstlane(lin % p.co_group);
    lin /= p.co_group;
    const int t_tile = __builtin_amdgcn_readfirstlane(lin % p.n_t_tiles);
    lin /= p.n_t_tiles;
    const int b = __builtin_amdgcn_readfirstlane(lin % p.B);
    const int co_tile = __builtin_amdgcn_readfirstlane((lin / p.B) * p.co_group + co_in);
    const int T = p.Tout, n_cb = p.n_cb, Cin = p.Cin;
    const int in_mode = p.in_mode;
    for (int i = tid; i < BM; i += 256) Ep[i] = p.bias ? p.bias[min(co_tile * BM + i, p.Cout - 1)] : 0.0f;
    float in_mu = 0.0f, in_rs = 1.0f;
    if (in_mode & 1) {
        in_mu = p.in_stats[2 * b];
        in_rs = p.in_stats[2 * b + 1];
        for (int i = tid; i < n_cb * CB; i += 256) Gt[i] = make_float2(p.in_gamma[min(i, Cin - 1)], p.in_beta[min(i, Cin - 1)]);
    }
    const unsigned x_cstride = (unsigned)p.x_cstride;
    const int col0 = t_tile * BN + wave * BNW;                     // first column of this wave's 64-column span
    const int col = col0 + TN * l31;                               // this lane's two columns: col, col + 1
    const int colc = min(col, T - TN);
    const int hcol = min(max(l31 < 16 ? col0 - 1 : col0 + BNW, 0), T - 1);   // halo column: left of the span (lanes 0-15) / right of it
    const float* const xb = p.x + (int64_t)b * p.x_bstride;
    const unsigned x_lane_off = (unsigned)hi * x_cstride + (unsigned)colc;
    const unsigned h_lane_off = (unsigned)hi * x_cstride + (unsigned)hcol;
    const c3_f32x4* const wbase = reinterpret_cast<const c3_f32x4*>(p.w + (int64_t)co_tile * n_cb * A_FLOATS);
    const bool first_col = col == 0, last_col = col + TN == T;    // reflect: x[-1] = x[1] (this lane's second value), x[T] = x[T-2] (its first)
    const bool lane_first = l31 == 0, lane_last = l31 == 31;

    c3_f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // ring of channel pairs: pair g = channels 2g (lanes 0-31) and 2g+1 (lanes 32-63); rows past Cin meet zero weights
    c3_f32x2 bq[PF];
    float hq[PF];
    const int last_pair = Cin / 2 - 1;
    auto load_pair = [&](int g, c3_f32x2& v, float& h) __attribute__((always_inline)) {
        const float* row = xb + (size_t)(2 * min(g, last_pair)) * x_cstride;
        if constexpr (XV2) {
            v = *reinterpret_cast<const c3_f32x2*>(row + x_lane_off);
        } else {
            v[0] = row[x_lane_off];
            v[1] = row[x_lane_off + 1];
        }
        h = row[h_lane_off];
    };
#pragma unroll
    for (int u = 0; u < PF; ++u) load_pair(u, bq[u], hq[u]);

    c3_f32x4 ra[NA];
#pragma unroll
    for (int n = 0; n < NA; ++n) {
        const int idx = tid + 256 * n;
        if ((A_VEC % 256 == 0) || idx < A_VEC) reinterpret_cast<c3_f32x4*>(As[0])[idx] = wbase[idx];
    }
    __syncthreads();

    auto act = [&](float t, float2 gb) __attribute__((always_inline)) -> float {
        if (in_mode & 1) t = ((t - in_mu) * in_rs) * gb.x + gb.y;    // GroupNorm(1,C) apply (NormConv1d.cs:155)
        if (in_mode & 2) t = nc_eluf(t);   // (a branch-free ELU measured slower here: 138 -> 145 us on the 32 -> 16 layer)
        return t;
    };

    for (int cb = 0; cb < n_cb; ++cb) {
        const int cur = cb & 1;
        const bool more = cb + 1 < n_cb;
        if (more) {
            const c3_f32x4* src = wbase + (size_t)(cb + 1) * A_VEC;
#pragma unroll
            for (int n = 0; n < NA; ++n) {
                const unsigned idx = (unsigned)(tid + 256 * n);
                ra[n] = src[(A_VEC % 256 == 0) ? idx : min(idx, (unsigned)(A_VEC - 1))];
            }
        }
        const float* Ac = As[cur] + hi * BM + nc_a_lane_off<TM>(l31);
        c3_static_for<CB / 2>([&](auto pt) __attribute__((always_inline)) {
            constexpr int pr = decltype(pt)::value;           // channel pair within the block
            const int g = cb * (CB / 2) + pr;
            const float2 gb = (in_mode & 1) ? Gt[2 * g + hi] : make_float2(1.0f, 0.0f);
            const c3_f32x2 raw = bq[pr % PF];
            const float hraw = hq[pr % PF];
            load_pair(g + PF, bq[pr % PF], hq[pr % PF]);     // unconditional (clamped)
            const float a = act(raw[0], gb), bb = act(raw[1], gb), hv = act(hraw, gb);
            // neighbours: column col-1 = the left lane's second value (the span's halo for its first lane), col+2 = the right lane's first
            float aL = c3_from_left(bb), bR = c3_from_right(a);
            aL = lane_first ? hv : aL;
            bR = lane_last ? hv : bR;
            aL = first_col ? bb : aL;                         // reflect pad (SConv1d.cs:258-274): x[-1] = x[1]
            bR = last_col ? a : bR;                           //                                    x[T]  = x[T-2]
            const float ax = c3_other_half(a, hi), bx = c3_other_half(bb, hi);
            // step 0: (c0,k0 | c0,k1)   step 1: (c0,k2 | c1,k0)   step 2: (c1,k1 | c1,k2); per output column j: taps (L, C, R)
            const float s0[2] = {hi ? ax : aL, hi ? bx : a};
            const float s1[2] = {hi ? aL : bb, hi ? a : bR};
            const float s2[2] = {hi ? bb : ax, hi ? bR : bx};
            float fa[TM];
            nc_load_a_frag<TM>(Ac + 2 * (3 * pr) * BM, l31, fa);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], s0[0], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], s0[1], acc[i][1], 0, 0, 0);
            }
            nc_load_a_frag<TM>(Ac + 2 * (3 * pr + 1) * BM, l31, fa);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], s1[0], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], s1[1], acc[i][1], 0, 0, 0);
            }
            nc_load_a_frag<TM>(Ac + 2 * (3 * pr + 2) * BM, l31, fa);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], s2[0], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], s2[1], acc[i][1], 0, 0, 0);
            }
        });
        if (more) {
#pragma unroll
            for (int n = 0; n < NA; ++n) {
                const int idx = tid + 256 * n;
                if ((A_VEC % 256 == 0) || idx < A_VEC) reinterpret_cast<c3_f32x4*>(As[cur ^ 1])[idx] = ra[n];
            }
        }
        __syncthreads();
    }

    // ---- epilogue: D[row = (r&3) + 8*(r>>2) + 4*hi] for this lane's two columns: GroupNorm block sums of the output (nc_gn.h; the
    //      column of (lane, j) is 2*l31 + j, as in conv1x1_kernel), then bias + store
    const int rows_total = p.Cout - co_tile * BM;
    if (p.gn_part != nullptr) {
        double* const gp = p.gn_part + (int64_t)b * p.gn_nrb * p.gn_ncb * 2;
        const bool colok = col < T;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            double a1[TN], a2[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float vv[16];
                unsigned okm16 = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int R = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    vv[r] = acc[i][j][r] + Ep[R];
                    if (colok && R < rows_total) okm16 |= 1u << r;
                }
                nc_gn_slot_sums<false>(vv, okm16, a1[j], a2[j]);
            }
            double s1 = a1[0] + a1[1], s2 = a2[0] + a2[1];
            nc_gn_butterfly_row(s1, s2);
            s1 = nc_gn_swap_add<true>(s1);
            s2 = nc_gn_swap_add<true>(s2);
            const int rbk = co_tile * TM + i, cbk = (col0 >> 5) + (l31 >> 4);
            if ((lane & 47) == 0 && rbk < p.gn_nrb && cbk < p.gn_ncb) nc_gn_store_partial(gp + ((int64_t)rbk * p.gn_ncb + cbk) * 2, s1, s2);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (p.gn_count != nullptr)
            nc_gn_arrive_and_finish(gp, p.gn_count + b, p.gn_stats + 2 * b, p.gn_nrb * p.gn_ncb, (unsigned)(p.n_co_tiles * p.n_t_tiles), p.gn_n);
    }
    if (col >= T) return;
    const int64_t tile_base = (int64_t)b * p.y_bstride + (int64_t)co_tile * BM * p.y_cstride;
    const unsigned cstride = (unsigned)p.y_cstride;
    float* const yt = p.y + tile_base + (unsigned)(4 * hi) * cstride + (unsigned)col;
    const int rows_left = rows_total - 4 * hi;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int R = i * 32 + (r & 3) + 8 * (r >> 2);
            if (R >= rows_left) continue;
            const float bias = Ep[R + 4 * hi];
            const c3_f32x2 v = {acc[i][0][r] + bias, acc[i][1][r] + bias};
            *reinterpret_cast<c3_f32x2*>(yt + (size_t)R * cstride) = v;
        }
}

typedef void (*conv_kernel_fn)(const ConvArgs);
conv_kernel_fn conv3_stream_kernel_table(int TM, bool aligned) {
    switch (TM) {
        case 1: return aligned ? &conv3_stream_kernel<1, true> : &conv3_stream_kernel<1, false>;
        case 2: return aligned ? &conv3_stream_kernel<2, true> : &conv3_stream_kernel<2, false>;
        case 4: return aligned ? &conv3_stream_kernel<4, true> : &conv3_stream_kernel<4, false>;
    }
    return nullptr;
}

}  // namespace nc
