// The third down-convolution of the Encodec 48 kHz encoder in streaming form (SEANetEncoder.cs: [ResnetBlock, ELU, SConv1d(C -> 2C, k = 10,
// stride 5)]; SConv1d.cs:144-173: non-causal reflect pad 3 + 2):
//     y = conv_{k10,s5}( pad( ELU( GN_s(s) + GN_y(y_branch) ) ) )          128 -> 256 channels, 6000 -> 1200 steps x 32 clips, 25.2 GFLOP
// Until round 6: a summed / activated copy (pad_act_kernel, 57 us) + the windowed template on 608 workgroups (404 us: 1.2 rounds of the chip).
// Same streaming form as down5_kernel (nc_down4.hip): a lane owns FIVE adjacent input columns (5t .. 5t+4; rows are only 4-byte aligned at
// that pitch: five dword loads per operand) of its channel row, normalises + adds + activates once per element in registers; the ten taps
// of output column t -- x[5t-3 .. 5t+6] -- are the left lane's last three values, its own five and the right lane's first two (DPP shifts;
// a halo triple per 32-column span; reflect as in-lane fixes).  kk = ci*10 + k ascending: channel c feeds five matrix-core steps (k even |
// k odd); with c0 on lanes 0-31 and c1 on lanes 32-63 one v_permlane32_swap per step pair yields both B operands.  The 256 output rows are
// two row tiles of 128 (TM = 4) -- one workgroup each over the same columns, neighbours in the tile map (the second reads the operands out
// of L2) -- because 256 rows x 32 columns of accumulators (128 registers) would leave no room for three waves per SIMD, which this grid needs
// (2432 waves: 1.19 rounds at two per SIMD, one round at three).  The weight image is ConvLayer::build's: inside a row tile its rows are
// simply kk = c*10 + k, so ANY channel range is contiguous -- four channels (two pairs, 20 KB) per barrier, double-buffered.
// Bit-identical to pad_act_kernel + the windowed launch (NC_NO_DOWN5=1 runs those; tests/test_encodec_gpu.py holds both to the C oracle).
#include <type_traits>
#include <utility>

#include "nc_conv.h"
#include "nc_frag.h"
#include "nc_gn.h"
#include "nc_math.h"

namespace nc {

typedef float d5_f32x16 __attribute__((ext_vector_type(16)));
typedef float d5_f32x4 __attribute__((ext_vector_type(4)));
typedef float d5_f32x2 __attribute__((ext_vector_type(2)));

template <int N, class F, int... I>
__device__ __forceinline__ void d5_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void d5_static_for(F&& f) {
    d5_static_for_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}
__device__ __forceinline__ float d5_from_left(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float d5_from_right(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false));
}
// (even, odd) tap of this lane's channel -> the two B operands of the step pair: b0 = (c0 even | c0 odd), b1 = (c1 even | c1 odd)
__device__ __forceinline__ void d5_step_operands(float even, float odd, float& b0, float& b1) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(even), __float_as_uint(odd), false, false);
    b0 = __uint_as_float(r[0]);   // `even` with its upper half replaced by the lower half of `odd`
    b1 = __uint_as_float(r[1]);   // `odd` with its lower half replaced by the upper half of `even`
}

template <int TM>
__global__ __launch_bounds__(256, 3) void down5_kernel(const Down2Args p) {
    constexpr int CB = 4, K = 10, BM = 32 * TM;    // four input channels (two pairs) per barrier
    constexpr int A_FLOATS = CB * K * BM, A_VEC = A_FLOATS / 4, NA = (A_VEC + 255) / 256;
    constexpr int PF = 2;                          // channel pairs in flight (= the pairs of a stage: the ring slot of a pair is its index in the stage)

    __shared__ __attribute__((aligned(16))) float As[2][A_FLOATS];
    __shared__ float Ep[BM];
    __shared__ float4 Gt[128];

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = gridDim.x, bid = blockIdx.x;
    int lin;
    {
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int co_tile = __builtin_amdgcn_readfirstlane(lin % p.n_co_tiles);      // (the row tiles of a column tile are neighbours in the launch order)
    lin /= p.n_co_tiles;
    const int t_tile = __builtin_amdgcn_readfirstlane(lin % p.n_t_tiles);
    const int b = __builtin_amdgcn_readfirstlane(lin / p.n_t_tiles);
    const int T = p.T, Tout = p.Tout, n_cb = p.n_cb, Cin = p.Cin;
    const bool gn_in = p.stats_a != nullptr;
    for (int i = tid; i < BM; i += 256) Ep[i] = p.bias ? p.bias[min(co_tile * BM + i, p.Cout - 1)] : 0.0f;
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
    const int ocol = ocol0 + l31;
    const int col0 = 5 * ocol0, col = 5 * ocol;                    // input columns col .. col + 4
    const int colc = min(col, T - 5);
    const int hcol = min(max(l31 < 16 ? col0 - 3 : col0 + 160, 0), T - 3);   // halo TRIPLE: the three columns left of the span (lanes 0-15) / the (two) right of it
    const float* const xa = p.xa + (int64_t)b * p.x_bstride;
    const float* const xb = p.xb + (int64_t)b * p.x_bstride;
    const unsigned x_lane_off = (unsigned)hi * x_cstride + (unsigned)colc;
    const unsigned h_lane_off = (unsigned)hi * x_cstride + (unsigned)hcol;
    const d5_f32x4* const wbase = reinterpret_cast<const d5_f32x4*>(p.w + (int64_t)co_tile * p.w_co_stride);
    const bool first_col = col == 0, last_col = col + 5 == T;     // reflect: x[-q] = x[q]; x[T] = x[T-2], x[T+1] = x[T-3]
    const bool lane_first = l31 == 0, lane_last = l31 == 31;

    d5_f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    struct Five { float v[5]; };
    struct Three { float v[3]; };
    Five qa[PF], qb[PF];
    Three ha[PF], hb[PF];
    const int last_pair = Cin / 2 - 1;
    auto load_pair = [&](int g, Five& va, Five& vb, Three& h_a, Three& h_b) __attribute__((always_inline)) {
        const size_t ro = (size_t)(2 * min(g, last_pair)) * x_cstride;
#pragma unroll
        for (int i = 0; i < 5; ++i) { va.v[i] = xa[ro + x_lane_off + i]; vb.v[i] = xb[ro + x_lane_off + i]; }
#pragma unroll
        for (int i = 0; i < 3; ++i) { h_a.v[i] = xa[ro + h_lane_off + i]; h_b.v[i] = xb[ro + h_lane_off + i]; }
    };
#pragma unroll
    for (int u = 0; u < PF; ++u) load_pair(u, qa[u], qb[u], ha[u], hb[u]);

    d5_f32x4 ra[NA];
#pragma unroll
    for (int n = 0; n < NA; ++n) reinterpret_cast<d5_f32x4*>(As[0])[tid + 256 * n] = wbase[tid + 256 * n];   // (A_VEC = 5 * 256: every thread, every pass)
    __syncthreads();

    // the staged value: GN_a(a) + GN_b(b), then ELU (pad_act_kernel's arithmetic: normalise each operand, add, activate)
    auto act = [&](float va, float vb, float4 g) __attribute__((always_inline)) -> float {
        float v = va, w = vb;
        if (gn_in) {
            v = ((v - mu_a) * rs_a) * g.x + g.y;
            w = ((w - mu_b) * rs_b) * g.z + g.w;
        }
        v = v + w;
        return nc_eluf(v);
    };

    for (int cb = 0; cb < n_cb; ++cb) {
        const int cur = cb & 1;
        const bool more = cb + 1 < n_cb;
        if (more) {
            const d5_f32x4* src = wbase + (size_t)(cb + 1) * A_VEC;
#pragma unroll
            for (int n = 0; n < NA; ++n) ra[n] = src[tid + 256 * n];
        }
        const float* Ac = As[cur] + hi * BM + nc_a_lane_off<TM>(l31);
        d5_static_for<CB / 2>([&](auto pt) __attribute__((always_inline)) {
            constexpr int pr = decltype(pt)::value;                 // channel pair within the block: channels 2 pr (c0), 2 pr + 1 (c1)
            const int g = cb * (CB / 2) + pr;
            const float4 gt = gn_in ? Gt[2 * g + hi] : make_float4(1.0f, 0.0f, 1.0f, 0.0f);
            // (two pairs per stage and PF = 2: the ring slot of pair g is pr)
            const Five va = qa[pr], vb = qb[pr];
            const Three h_a = ha[pr], h_b = hb[pr];
            load_pair(g + PF, qa[pr], qb[pr], ha[pr], hb[pr]);
            float x[5], h[3];
#pragma unroll
            for (int i = 0; i < 5; ++i) x[i] = act(va.v[i], vb.v[i], gt);
#pragma unroll
            for (int i = 0; i < 3; ++i) h[i] = act(h_a.v[i], h_b.v[i], gt);
            float L2 = d5_from_left(x[2]), L3 = d5_from_left(x[3]), L4 = d5_from_left(x[4]), R0 = d5_from_right(x[0]), R1 = d5_from_right(x[1]);
            L2 = lane_first ? h[0] : L2;  L3 = lane_first ? h[1] : L3;  L4 = lane_first ? h[2] : L4;
            R0 = lane_last ? h[0] : R0;   R1 = lane_last ? h[1] : R1;
            L2 = first_col ? x[3] : L2;   L3 = first_col ? x[2] : L3;   L4 = first_col ? x[1] : L4;   // reflect pad (SConv1d.cs:258-274): x[-3], x[-2], x[-1]
            R0 = last_col ? x[3] : R0;    R1 = last_col ? x[2] : R1;                                  //                                    x[T] = x[T-2], x[T+1] = x[T-3]
            float b0[5], b1[5];
            d5_step_operands(L2, L3, b0[0], b1[0]);
            d5_step_operands(L4, x[0], b0[1], b1[1]);
            d5_step_operands(x[1], x[2], b0[2], b1[2]);
            d5_step_operands(x[3], x[4], b0[3], b1[3]);
            d5_step_operands(R0, R1, b0[4], b1[4]);
            // channel c0 = 2 pr of the stage: kk = 10 (2 pr) + k -> steps 10 pr .. 10 pr + 4; channel c1: steps 10 pr + 5 .. 10 pr + 9
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                float fa[TM];
                nc_load_a_frag<TM>(Ac + 2 * (10 * pr + s) * BM, l31, fa);
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], b0[s], acc[i], 0, 0, 0);
            }
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                float fa[TM];
                nc_load_a_frag<TM>(Ac + 2 * (10 * pr + 5 + s) * BM, l31, fa);
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], b1[s], acc[i], 0, 0, 0);
            }
        });
        if (more) {
#pragma unroll
            for (int n = 0; n < NA; ++n) reinterpret_cast<d5_f32x4*>(As[cur ^ 1])[tid + 256 * n] = ra[n];
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
                if (colok && co_tile * BM + R < p.Cout) okm16 |= 1u << r;
            }
            double s1, s2;
            nc_gn_slot_sums<false>(vv, okm16, s1, s2);
            nc_gn_butterfly(s1, s2);
            const int cbk = ocol0 >> 5;
            const int rbk = co_tile * TM + i;
            if (lane == 0 && rbk < p.gn_nrb && cbk < p.gn_ncb) nc_gn_store_partial(gp + ((int64_t)rbk * p.gn_ncb + cbk) * 2, s1, s2);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (p.gn_count != nullptr)
            nc_gn_arrive_and_finish(gp, p.gn_count + b, p.gn_stats + 2 * b, p.gn_nrb * p.gn_ncb, (unsigned)(p.n_t_tiles * p.n_co_tiles), p.gn_n);
    }
    if (!colok) return;
    float* const yt = p.y + (int64_t)b * p.y_bstride + (unsigned)(co_tile * BM + 4 * hi) * (unsigned)p.y_cstride + (unsigned)ocol;
    const int rows_left = p.Cout - co_tile * BM - 4 * hi;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int R = i * 32 + (r & 3) + 8 * (r >> 2);
            if (R >= rows_left) continue;
            yt[(size_t)R * (unsigned)p.y_cstride] = acc[i][r] + Ep[R + 4 * hi];
        }
}

bool launch_down5(const Down2Args& a, int TM, hipStream_t stream) {
    if (TM != 4) return false;
    hipLaunchKernelGGL(down5_kernel<4>, dim3((unsigned)((int64_t)a.B * a.n_t_tiles * a.n_co_tiles)), dim3(256), 0, stream, a);
    NC_HIP(hipGetLastError());
    return true;
}

}  // namespace nc
