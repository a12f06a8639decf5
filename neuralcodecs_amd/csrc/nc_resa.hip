// First pass of a SEANetResnetBlock in ONE launch (SEANetResnetBlock.cs:53-85, trueSkip = false):
//     s = shortcut(x)      = SConv1d(C -> C, k = 1)                 on the block input with its pending GroupNorm applied
//     h = conv1(ELU(x))    = SConv1d(C -> C/2, k = 3, reflect 1+1)  on the same input, activated
// Both are thin, long, HBM-bound layers of the 48 kHz model's outer stages (C = 32 at 48000 steps, C = 64 at 24000: 4 * C * T bytes in,
// 4 * 1.5 * C * T out per clip); as two launches the block input is read twice and the k = 3 launch alone ran at 2.1 TB/s.  Here a lane
// owns two adjacent columns of its channel row exactly as in conv3_stream_kernel (nc_conv3s.hip): ONE 8-byte load per channel pair, the
// pending GroupNorm applied once per element in registers; the normalised value feeds the shortcut's matrix-core step (kk = ci: lanes
// 0-31 hold channel 2g, lanes 32-63 channel 2g+1 -- the canonical chain, no exchange), its ELU the three tap steps of the branch (taps from
// the lane neighbours by DPP shifts + one halo value per 64-column span; kk = ci*3 + k; reflect pad as an in-lane fix).  Both weight tiles
// stream through LDS double-buffered, one barrier per 16 input channels -- the images are the ones ConvLayer::build packs for the two
// layers (single row tile each: C <= 64).  Epilogue twice: GroupNorm block sums in the canonical order (nc_gn.h) with the in-launch finish
// on the output's own counters, bias, 8-byte stores.  The second pass of the block (conv2, k = 1 on h) stays conv1x1_kernel.
// Arithmetic: operation for operation that of conv1x1_kernel (shortcut) and conv3_stream_kernel (branch): results are bit-identical to
// the two-launch path (tests/test_encodec_gpu.py runs both and the C oracle).
#include <type_traits>
#include <utility>

#include "nc_conv.h"
#include "nc_frag.h"
#include "nc_gn.h"
#include "nc_math.h"

namespace nc {

typedef float ra_f32x16 __attribute__((ext_vector_type(16)));
typedef float ra_f32x4 __attribute__((ext_vector_type(4)));
typedef float ra_f32x2 __attribute__((ext_vector_type(2)));

template <int N, class F, int... I>
__device__ __forceinline__ void ra_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void ra_static_for(F&& f) {
    ra_static_for_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}
__device__ __forceinline__ float ra_from_left(float v) {    // lane i <- lane i-1 (DPP wave_shr:1)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float ra_from_right(float v) {   // lane i <- lane i+1 (DPP wave_shl:1)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ float ra_other_half(float v, int hi) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(hi ? r[0] : r[1]);
}

// GroupNorm block sums of one output (rows = TM tiles of 32, this wave's 64 columns) + the in-launch finish; every thread calls it
template <int TM>
__device__ __forceinline__ void ra_gn_out(const ra_f32x16 (&acc)[TM][2], const float* Ep, int rows_total, bool colok, int col0, int l31, int hi, int lane,
                                          double* gp, int nrb, int ncb, unsigned* count, float* stats, unsigned n_wg, double n) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        double a1[2], a2[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
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
        const int cbk = (col0 >> 5) + (l31 >> 4);
        if ((lane & 47) == 0 && i < nrb && cbk < ncb) nc_gn_store_partial(gp + ((int64_t)i * ncb + cbk) * 2, s1, s2);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (count != nullptr) nc_gn_arrive_and_finish(gp, count, stats, nrb * ncb, n_wg, n);
}

template <int TM>
__device__ __forceinline__ void ra_store(const ra_f32x16 (&acc)[TM][2], const float* Ep, int rows_total, float* yt, unsigned cstride, int hi) {
    const int rows_left = rows_total - 4 * hi;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int R = i * 32 + (r & 3) + 8 * (r >> 2);
            if (R >= rows_left) continue;
            const float bias = Ep[R + 4 * hi];
            const ra_f32x2 v = {acc[i][0][r] + bias, acc[i][1][r] + bias};
            *reinterpret_cast<ra_f32x2*>(yt + (size_t)R * cstride) = v;
        }
}

// TMS: 32-row tiles of the shortcut (C / 32); the branch has one (C / 2 <= 32 rows).  XV2: rows 8-byte aligned at even columns.
template <int TMS, bool XV2>
__global__ __launch_bounds__(256, TMS == 1 ? 4 : 3) void res_a_kernel(const ResAArgs p) {
    constexpr int CB = 16, BNW = 64, BN = 256;
    constexpr int BMS = 32 * TMS, BMB = 32;
    constexpr int AB_FLOATS = CB * 3 * BMB, AB_VEC = AB_FLOATS / 4, NAB = (AB_VEC + 255) / 256;   // 384 vectors: 2 passes
    constexpr int AS_FLOATS = CB * BMS, AS_VEC = AS_FLOATS / 4;                                    // 128 / 256 vectors: 1 pass
    // (PF: channel pairs in flight; 8 measured the same as 4)


    constexpr int PF = 4;


    __shared__ __attribute__((aligned(16))) float Asb[2][AB_FLOATS];
    __shared__ __attribute__((aligned(16))) float Ass[2][AS_FLOATS];
    __shared__ float Eps[BMS];
    __shared__ float Epb[BMB];
    __shared__ float2 Gt[64];

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
    const int T = p.T, n_cb = p.n_cb, Cin = p.Cin;
    const bool gn_in = p.in_stats != nullptr;
    for (int i = tid; i < BMS; i += 256) Eps[i] = p.bias_s ? p.bias_s[min(i, p.Cs - 1)] : 0.0f;
    for (int i = tid; i < BMB; i += 256) Epb[i] = p.bias_b ? p.bias_b[min(i, p.Cb - 1)] : 0.0f;
    float in_mu = 0.0f, in_rs = 1.0f;
    if (gn_in) {
        in_mu = p.in_stats[2 * b];
        in_rs = p.in_stats[2 * b + 1];
        for (int i = tid; i < n_cb * CB; i += 256) Gt[i] = make_float2(p.in_gamma[min(i, Cin - 1)], p.in_beta[min(i, Cin - 1)]);
    }
    const unsigned x_cstride = (unsigned)p.x_cstride;
    const int col0 = t_tile * BN + wave * BNW;
    const int col = col0 + 2 * l31;
    const int colc = min(col, T - 2);
    const float* const xb = p.x + (int64_t)b * p.x_bstride;
    const unsigned x_lane_off = (unsigned)hi * x_cstride + (unsigned)colc;
    const ra_f32x4* const wb_base = reinterpret_cast<const ra_f32x4*>(p.w_b);
    const ra_f32x4* const ws_base = reinterpret_cast<const ra_f32x4*>(p.w_s);
    const bool first_col = col == 0, last_col = col + 2 == T;
    const bool lane_first = l31 == 0, lane_last = l31 == 31;
    const int hsel_addr = 4 * (32 * hi + (lane_last ? 2 : 0));   // ds_bpermute byte address of this lane's halo value for pair 0

    ra_f32x16 accs[TMS][2], accb[1][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            accb[0][j][r] = 0.0f;
#pragma unroll
            for (int i = 0; i < TMS; ++i) accs[i][j][r] = 0.0f;
        }

    ra_f32x2 bq[PF];
    const int last_pair = Cin / 2 - 1;
    auto load_pair = [&](int g, ra_f32x2& v) __attribute__((always_inline)) {
        const float* row = xb + (size_t)(2 * min(g, last_pair)) * x_cstride;
        if constexpr (XV2) {
            v = *reinterpret_cast<const ra_f32x2*>(row + x_lane_off);
        } else {
            v[0] = row[x_lane_off];
            v[1] = row[x_lane_off + 1];
        }
    };
#pragma unroll
    for (int u = 0; u < PF; ++u) load_pair(u, bq[u]);
    // Halo values of a WHOLE reduction block in one go: the two columns next to the span (col0 - 1, col0 + 64) x 8 channel pairs x 2 channels
    // = 32 values, one per lane group -- lane (hi, l31) fetches channel 2 (8 cb + (l31 >> 2)) + hi at the left (bit 1 of l31 clear) / right
    // column -- so ONE GroupNorm apply + ELU per block serves all eight pairs (it was one per pair, for two useful lanes each); pair pr picks
    // its four values with v_readlane / v_writelane.
    const int hb_pair = l31 >> 2;
    const int hb_col = min(max((l31 & 2) ? col0 + BNW : col0 - 1, 0), T - 1);
    auto load_halo_block = [&](int cbi) __attribute__((always_inline)) -> float {
        const int g = min(cbi * (CB / 2) + hb_pair, last_pair);
        return xb[(size_t)(2 * g + hi) * x_cstride + (unsigned)hb_col];
    };
    float hb_next = load_halo_block(0);

    ra_f32x4 rab[NAB], ras;
#pragma unroll
    for (int n = 0; n < NAB; ++n) {
        const int idx = tid + 256 * n;
        if (idx < AB_VEC) reinterpret_cast<ra_f32x4*>(Asb[0])[idx] = wb_base[idx];
    }
    if (tid < AS_VEC) reinterpret_cast<ra_f32x4*>(Ass[0])[tid] = ws_base[tid];
    __syncthreads();

    for (int cb = 0; cb < n_cb; ++cb) {
        const int cur = cb & 1;
        const bool more = cb + 1 < n_cb;
        if (more) {
            const ra_f32x4* srcb = wb_base + (size_t)(cb + 1) * AB_VEC;
#pragma unroll
            for (int n = 0; n < NAB; ++n) rab[n] = srcb[min((unsigned)(tid + 256 * n), (unsigned)(AB_VEC - 1))];
            ras = (ws_base + (size_t)(cb + 1) * AS_VEC)[min((unsigned)tid, (unsigned)(AS_VEC - 1))];
        }
        const float* Acb = Asb[cur] + hi * BMB + nc_a_lane_off<1>(l31);
        const float* Acs = Ass[cur] + hi * BMS + nc_a_lane_off<TMS>(l31);
        float hvb = hb_next;
        hb_next = load_halo_block(cb + 1);                     // (clamped past the last block)
        if (gn_in) {
            const float2 gbh = Gt[2 * min(cb * (CB / 2) + hb_pair, last_pair) + hi];
            hvb = ((hvb - in_mu) * in_rs) * gbh.x + gbh.y;
        }
        hvb = nc_eluf(hvb);
        ra_static_for<CB / 2>([&](auto pt) __attribute__((always_inline)) {
            constexpr int pr = decltype(pt)::value;
            const int g = cb * (CB / 2) + pr;
            const float2 gb = gn_in ? Gt[2 * g + hi] : make_float2(1.0f, 0.0f);
            const ra_f32x2 raw = bq[pr % PF];
            load_pair(g + PF, bq[pr % PF]);
            float na = raw[0], nb = raw[1];
            if (gn_in) {                                       // GroupNorm(1,C) apply (NormConv1d.cs:155)
                na = ((na - in_mu) * in_rs) * gb.x + gb.y;
                nb = ((nb - in_mu) * in_rs) * gb.x + gb.y;
            }
            // shortcut: kk = ci, the un-activated value of this lane's own channel
            {
                float fs[TMS];
                nc_load_a_frag<TMS>(Acs + 2 * pr * BMS, l31, fs);
#pragma unroll
                for (int i = 0; i < TMS; ++i) {
                    accs[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fs[i], na, accs[i][0], 0, 0, 0);
                    accs[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fs[i], nb, accs[i][1], 0, 0, 0);
                }
            }
            // branch: ELU, then the three taps (conv3_stream_kernel's step layout)





            const float a = nc_eluf(na), bb = nc_eluf(nb);
            float aL = ra_from_left(bb), bR = ra_from_right(a);
            // the span's halo for the first / last lane of either half: lanes 4 pr (+32) hold the left values, 4 pr + 2 (+32) the right ones
            const float hs = __int_as_float(__builtin_amdgcn_ds_bpermute(hsel_addr + 16 * pr, __float_as_int(hvb)));
            aL = lane_first ? hs : aL;
            bR = lane_last ? hs : bR;
            aL = first_col ? bb : aL;                          // reflect pad (SConv1d.cs:258-274): x[-1] = x[1]
            bR = last_col ? a : bR;                            //                                    x[T]  = x[T-2]
            const float ax = ra_other_half(a, hi), bx = ra_other_half(bb, hi);
            const float s0[2] = {hi ? ax : aL, hi ? bx : a};
            const float s1[2] = {hi ? aL : bb, hi ? a : bR};
            const float s2[2] = {hi ? bb : ax, hi ? bR : bx};
            float fa[1];
            nc_load_a_frag<1>(Acb + 2 * (3 * pr) * BMB, l31, fa);
            accb[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0], s0[0], accb[0][0], 0, 0, 0);
            accb[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0], s0[1], accb[0][1], 0, 0, 0);
            nc_load_a_frag<1>(Acb + 2 * (3 * pr + 1) * BMB, l31, fa);
            accb[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0], s1[0], accb[0][0], 0, 0, 0);
            accb[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0], s1[1], accb[0][1], 0, 0, 0);
            nc_load_a_frag<1>(Acb + 2 * (3 * pr + 2) * BMB, l31, fa);
            accb[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0], s2[0], accb[0][0], 0, 0, 0);
            accb[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0], s2[1], accb[0][1], 0, 0, 0);
        });
        if (more) {
#pragma unroll
            for (int n = 0; n < NAB; ++n) {
                const int idx = tid + 256 * n;
                if (idx < AB_VEC) reinterpret_cast<ra_f32x4*>(Asb[cur ^ 1])[idx] = rab[n];
            }
            if (tid < AS_VEC) reinterpret_cast<ra_f32x4*>(Ass[cur ^ 1])[tid] = ras;
        }
        __syncthreads();
    }

    const bool colok = col < T;

    if (p.gn_part_s != nullptr) {
        ra_gn_out<TMS>(accs, Eps, p.Cs, colok, col0, l31, hi, lane, p.gn_part_s + (int64_t)b * p.gn_nrb_s * p.gn_ncb * 2, p.gn_nrb_s, p.gn_ncb,
                       p.gn_count_s ? p.gn_count_s + b : nullptr, p.gn_stats_s + 2 * b, (unsigned)p.n_t_tiles, p.gn_n_s);
        ra_gn_out<1>(accb, Epb, p.Cb, colok, col0, l31, hi, lane, p.gn_part_b + (int64_t)b * p.gn_nrb_b * p.gn_ncb * 2, p.gn_nrb_b, p.gn_ncb,
                     p.gn_count_b ? p.gn_count_b + b : nullptr, p.gn_stats_b + 2 * b, (unsigned)p.n_t_tiles, p.gn_n_b);
    }

    if (!colok) return;
    ra_store<TMS>(accs, Eps, p.Cs, p.ys + (int64_t)b * p.ys_bstride + (unsigned)(4 * hi) * (unsigned)p.ys_cstride + (unsigned)col, (unsigned)p.ys_cstride, hi);
    ra_store<1>(accb, Epb, p.Cb, p.yb + (int64_t)b * p.yb_bstride + (unsigned)(4 * hi) * (unsigned)p.yb_cstride + (unsigned)col, (unsigned)p.yb_cstride, hi);
}

bool launch_res_a(const ResAArgs& a, int TMS, bool aligned, hipStream_t stream) {
    void (*fn)(const ResAArgs) = nullptr;
    if (TMS == 1) fn = aligned ? &res_a_kernel<1, true> : &res_a_kernel<1, false>;
    else if (TMS == 2) fn = aligned ? &res_a_kernel<2, true> : &res_a_kernel<2, false>;
    if (!fn) return false;
    hipLaunchKernelGGL(fn, dim3((unsigned)((int64_t)a.B * a.n_t_tiles)), dim3(256), 0, stream, a);
    NC_HIP(hipGetLastError());
    return true;
}

}  // namespace nc
