// The second down-convolution of the Encodec 48 kHz encoder in streaming form (SEANetEncoder.cs: [ResnetBlock, ELU, SConv1d(C -> 2C, k = 8,
// stride 4)]; SConv1d.cs:144-173: non-causal reflect pad 2 + 2):
//     y = conv_{k8,s4}( pad( ELU( GN_s(s) + GN_y(y_branch) ) ) )            64 -> 128 channels, 24000 -> 6000 steps x 32 clips, 25.2 GFLOP
// Until round 6 this layer was a summed / activated copy (pad_act_kernel, 590 MB at 5.4 TB/s: 118 us) followed by the windowed template on the
// copy (299 us): with 128 output rows the two-input staging mode re-staged the window per row tile and lost.  The streaming form
// has no window at all: a lane owns FOUR adjacent input columns (4t .. 4t+3: one 16-byte load per operand and channel pair) of its channel
// row, normalises + adds + activates once per element in registers, and the eight taps of output column t -- x[4t-2 .. 4t+5] -- are the
// left lane's last two values, its own four and the right lane's first two (DPP shifts; one 8-byte halo pair per 32-column span; reflect
// as in-lane fixes).  kk = ci*8 + k ascending -- the canonical chain: channel c feeds four matrix-core steps (k even | k odd); with channel
// c0 on lanes 0-31 and c1 on lanes 32-63 ONE v_permlane32_swap(tap 2s, tap 2s+1) per step pair yields both B operands -- its first result
// is (c0 tap 2s | c0 tap 2s+1), its second (c1 tap 2s | c1 tap 2s+1) -- no selects.  All 128 output rows in one workgroup (TM = 4, 32
// columns per wave), so the activation work is done once.  Weight image: ConvLayer::build's ([n_cb][4 channels x 8 taps][128 rows]),
// double-buffered through LDS.  Epilogue: canonical GroupNorm block sums with the in-launch finish, bias, stores.
// Bit-identical to pad_act_kernel + the windowed launch (NC_NO_DOWN4=1 runs those; tests/test_encodec_gpu.py holds both to the C oracle).
#include <type_traits>
#include <utility>

#include "nc_conv.h"
#include "nc_frag.h"
#include "nc_gn.h"
#include "nc_math.h"

namespace nc {

typedef float d4_f32x16 __attribute__((ext_vector_type(16)));
typedef float d4_f32x4 __attribute__((ext_vector_type(4)));
typedef float d4_f32x2 __attribute__((ext_vector_type(2)));

template <int N, class F, int... I>
__device__ __forceinline__ void d4_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void d4_static_for(F&& f) {
    d4_static_for_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}
__device__ __forceinline__ float d4_from_left(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float d4_from_right(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false));
}
// (even, odd) tap of this lane's channel -> the two B operands of the step pair: b0 = (c0 even | c0 odd), b1 = (c1 even | c1 odd)
__device__ __forceinline__ void d4_step_operands(float even, float odd, float& b0, float& b1) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(even), __float_as_uint(odd), false, false);
    b0 = __uint_as_float(r[0]);   // `even` with its upper half replaced by the lower half of `odd`
    b1 = __uint_as_float(r[1]);   // `odd` with its lower half replaced by the upper half of `even`
}

template <int TM>
__global__ __launch_bounds__(256, 2) void down4_kernel(const Down2Args p) {
    // (CB = 8 input channels per barrier = TWO reduction blocks of the packed image, which lays the blocks of a row tile end to end; one block
    //  per barrier -- two channel pairs -- measured the same: 315 vs 310 us, the barrier is not what this kernel waits for)
    constexpr int CB = 8, K = 8, BM = 32 * TM;
    constexpr int A_FLOATS = CB * K * BM, A_VEC = A_FLOATS / 4, NA = (A_VEC + 255) / 256;
    constexpr int PF = 4;                          // channel pairs in flight

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
    const int ocol = ocol0 + l31;
    const int col0 = 4 * ocol0, col = 4 * ocol;                    // input columns col .. col + 3
    const int colc = min(col, T - 4);
    const int hcol = min(max(l31 < 16 ? col0 - 2 : col0 + 128, 0), T - 2);   // halo PAIR: left of the span (lanes 0-15) / right of it
    const float* const xa = p.xa + (int64_t)b * p.x_bstride;
    const float* const xb = p.xb + (int64_t)b * p.x_bstride;
    const unsigned x_lane_off = (unsigned)hi * x_cstride + (unsigned)colc;
    const unsigned h_lane_off = (unsigned)hi * x_cstride + (unsigned)hcol;
    const d4_f32x4* const wbase = reinterpret_cast<const d4_f32x4*>(p.w);
    const bool first_col = col == 0, last_col = col + 4 == T;     // reflect: x[-1] = x[1], x[-2] = x[2]; x[T] = x[T-2], x[T+1] = x[T-3]
    const bool lane_first = l31 == 0, lane_last = l31 == 31;

    d4_f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    d4_f32x4 qa[PF], qb[PF];
    d4_f32x2 ha[PF], hb[PF];
    const int last_pair = Cin / 2 - 1;
    auto load_pair = [&](int g, d4_f32x4& va, d4_f32x4& vb, d4_f32x2& h_a, d4_f32x2& h_b) __attribute__((always_inline)) {
        const size_t ro = (size_t)(2 * min(g, last_pair)) * x_cstride;
        va = *reinterpret_cast<const d4_f32x4*>(xa + ro + x_lane_off);
        vb = *reinterpret_cast<const d4_f32x4*>(xb + ro + x_lane_off);
        h_a = *reinterpret_cast<const d4_f32x2*>(xa + ro + h_lane_off);
        h_b = *reinterpret_cast<const d4_f32x2*>(xb + ro + h_lane_off);
    };
#pragma unroll
    for (int u = 0; u < PF; ++u) load_pair(u, qa[u], qb[u], ha[u], hb[u]);

    d4_f32x4 ra[NA];
#pragma unroll
    for (int n = 0; n < NA; ++n) reinterpret_cast<d4_f32x4*>(As[0])[tid + 256 * n] = wbase[tid + 256 * n];
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
            const d4_f32x4* src = wbase + (size_t)(cb + 1) * A_VEC;
#pragma unroll
            for (int n = 0; n < NA; ++n) ra[n] = src[tid + 256 * n];
        }
        const float* Ac = As[cur] + hi * BM + nc_a_lane_off<TM>(l31);
        d4_static_for<CB / 2>([&](auto pt) __attribute__((always_inline)) {
            constexpr int pr = decltype(pt)::value;                 // channel pair within the block: channels 2 pr (c0), 2 pr + 1 (c1)
            const int g = cb * (CB / 2) + pr;
            const float4 gt = gn_in ? Gt[2 * g + hi] : make_float4(1.0f, 0.0f, 1.0f, 0.0f);
            // (four pairs per block and PF = 4: the ring slot of pair g is pr)
            const d4_f32x4 va = qa[pr], vb = qb[pr];
            const d4_f32x2 h_a = ha[pr], h_b = hb[pr];
            load_pair(g + PF, qa[pr], qb[pr], ha[pr], hb[pr]);
            const float x0 = act(va[0], vb[0], gt), x1 = act(va[1], vb[1], gt), x2 = act(va[2], vb[2], gt), x3 = act(va[3], vb[3], gt);
            const float h0 = act(h_a[0], h_b[0], gt), h1 = act(h_a[1], h_b[1], gt);
            float L2 = d4_from_left(x2), L3 = d4_from_left(x3), R0 = d4_from_right(x0), R1 = d4_from_right(x1);
            L2 = lane_first ? h0 : L2;  L3 = lane_first ? h1 : L3;
            R0 = lane_last ? h0 : R0;   R1 = lane_last ? h1 : R1;
            L2 = first_col ? x2 : L2;   L3 = first_col ? x1 : L3;      // reflect pad (SConv1d.cs:258-274): x[-2] = x[2], x[-1] = x[1]
            R0 = last_col ? x2 : R0;    R1 = last_col ? x1 : R1;       //                                    x[T] = x[T-2], x[T+1] = x[T-3]
            float b0[4], b1[4];
            d4_step_operands(L2, L3, b0[0], b1[0]);
            d4_step_operands(x0, x1, b0[1], b1[1]);
            d4_step_operands(x2, x3, b0[2], b1[2]);
            d4_step_operands(R0, R1, b0[3], b1[3]);
            // channel c0: kk = 8 (2 pr) + k -> steps 8 pr .. 8 pr + 3 of the block; channel c1: steps 8 pr + 4 .. 8 pr + 7
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float fa[TM];
                nc_load_a_frag<TM>(Ac + 2 * (8 * pr + s) * BM, l31, fa);
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], b0[s], acc[i], 0, 0, 0);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float fa[TM];
                nc_load_a_frag<TM>(Ac + 2 * (8 * pr + 4 + s) * BM, l31, fa);
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], b1[s], acc[i], 0, 0, 0);
            }
        });
        if (more) {
#pragma unroll
            for (int n = 0; n < NA; ++n) reinterpret_cast<d4_f32x4*>(As[cur ^ 1])[tid + 256 * n] = ra[n];
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

bool launch_down4(const Down2Args& a, int TM, hipStream_t stream) {
    if (TM != 4) return false;
    hipLaunchKernelGGL(down4_kernel<4>, dim3((unsigned)((int64_t)a.B * a.n_t_tiles)), dim3(256), 0, stream, a);
    NC_HIP(hipGetLastError());
    return true;
}

}  // namespace nc
