// The last up-convolution of the Encodec 48 kHz decoder in streaming form (SEANetDecoder.cs: [ELU, SConvTranspose1d(C -> C/2, k = 4,
// stride 2), ResnetBlock]; SConvTranspose1d.cs:116-139: GroupNorm over the UNTRIMMED output, then the trim):
//     y[co, 2q + p] = sum_ci  x[ci, q] w[ci, co, p] + x[ci, q-1] w[ci, co, p + 2],     x = ELU( GN_s(s) + GN_y(y_branch) ),  q = 0 .. L
// in the sub-pixel form of the template (rows R = co*2 + p, two taps per phase, kk = ci*2 + j with tap j at x[q - j]: the canonical
// chain): 64 -> 32 channels at 24000 -> 48002 steps x 32 clips, 12.6 GFLOP over 590 MB.  The windowed two-input instance staged every
// element through LDS and ran at 1.9 TB/s (306 us).  Here, as in down2_kernel, a lane owns two adjacent input columns (2l, 2l+1) of its
// channel row for BOTH operands, normalises + adds + activates once per element in registers, and the two taps of a column are its own
// value and its left neighbour's (DPP shift; one halo value per 64-column span; x[-1] = x[L] = 0: the transposed convolution's zero
// extension of the ACTIVATED tensor).  A channel pair (c0 = lanes 0-31, c1 = lanes 32-63) feeds two matrix-core steps
//     (c0,x[q] | c0,x[q-1])   (c1,x[q] | c1,x[q-1])
// with three v_permlane32_swap.  Weight image: the one ConvLayer::build packs for the layer (sub-pixel rows, [n_cb][16 channels x 2
// taps][64 rows]), double-buffered through LDS.  Epilogue: GroupNorm block sums in the canonical order over the (rows R, columns q) view
// (conv_gn_sub = 2) with the in-launch finish, bias, and per output channel FOUR consecutive samples per lane (two 8-byte stores).
// Bit-identical to the windowed two-input launch (NC_NO_UP2=1 runs that; tests/test_encodec_gpu.py holds both to the C oracle).
//
// S = 4 (the up-convolution before it: 128 -> 64 channels, k = 8, stride 4, 6000 -> 24004 steps): the same two taps per phase with 256
// sub-pixel rows = two row tiles of 128 (TM = 4), one workgroup each over the same columns (adjacent in the tile map: the second reads the
// operands out of L2); a lane stores FOUR consecutive samples per output channel and column (16-byte stores).  It replaces a summed copy
// (pad_act_kernel, 56 us) + the windowed instance (378 us).  NC_NO_UP4=1 runs those.
#include <type_traits>
#include <utility>

#include "nc_conv.h"
#include "nc_frag.h"
#include "nc_gn.h"
#include "nc_math.h"

namespace nc {

typedef float u2_f32x16 __attribute__((ext_vector_type(16)));
typedef float u2_f32x4 __attribute__((ext_vector_type(4)));
typedef float u2_f32x2 __attribute__((ext_vector_type(2)));

template <int N, class F, int... I>
__device__ __forceinline__ void u2_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void u2_static_for(F&& f) {
    u2_static_for_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}
__device__ __forceinline__ float u2_from_left(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float u2_other_half(float v, int hi) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(hi ? r[0] : r[1]);
}

// TM: 32-row tiles of the sub-pixel rows (2 * Cout / 32; one row tile per workgroup: 2 * Cout <= 128).  XV2: rows 8-byte aligned at even columns.
template <int TM, int S, bool XV2>
__global__ __launch_bounds__(256, TM <= 2 ? 3 : 2) void up2_kernel(const Up2Args p) {
    constexpr int CB = 16, KT = 2, BM = 32 * TM;
    constexpr int A_FLOATS = CB * KT * BM, A_VEC = A_FLOATS / 4, NA = (A_VEC + 255) / 256;
    constexpr int PF = 4;

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
    const int co_tile = __builtin_amdgcn_readfirstlane(lin % p.n_co_tiles);      // (the row tiles of a column tile are neighbours in the launch order)
    lin /= p.n_co_tiles;
    const int t_tile = __builtin_amdgcn_readfirstlane(lin % p.n_t_tiles);
    const int b = __builtin_amdgcn_readfirstlane(lin / p.n_t_tiles);
    const int L = p.L, n_cb = p.n_cb, Cin = p.Cin;
    const int ncols = L + 1;                                       // columns q = 0 .. L of the sub-pixel view
    const bool gn_in = p.stats_a != nullptr;
    // bias of row R = co*S + phase is bias[co]
    for (int i = tid; i < BM; i += 256) Ep[i] = p.bias ? p.bias[min((co_tile * BM + i) / S, p.Cout - 1)] : 0.0f;
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
    const int col0 = t_tile * 256 + wave * 64;                     // first column of this wave's 64-column span
    const int col = col0 + 2 * l31;                                // this lane's two columns: col, col + 1
    const int colc = min(col, L - 2);                              // (L even, >= 4: the clamped pair is in bounds; values past L are masked)
    const int hcol = min(max(col0 - 1, 0), L - 1);                 // halo: the column left of the span
    const float* const xa = p.xa + (int64_t)b * p.x_bstride;
    const float* const xb = p.xb + (int64_t)b * p.x_bstride;
    const unsigned x_lane_off = (unsigned)hi * x_cstride + (unsigned)colc;
    const unsigned h_lane_off = (unsigned)hi * x_cstride + (unsigned)hcol;
    const u2_f32x4* const wbase = reinterpret_cast<const u2_f32x4*>(p.w) + (size_t)co_tile * n_cb * A_VEC;
    const bool lane_first = l31 == 0;
    const bool ok0 = col < L, ok1 = col + 1 < L;                   // x[q] = 0 for q >= L (zero extension of the activated tensor)
    const bool halo_ok = col0 >= 1 && col0 - 1 < L;                // x[-1] = 0

    u2_f32x16 acc[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    u2_f32x2 qa[PF], qb[PF];
    float ha[PF], hb[PF];
    const int last_pair = Cin / 2 - 1;
    auto load_pair = [&](int g, u2_f32x2& va, u2_f32x2& vb, float& h_a, float& h_b) __attribute__((always_inline)) {
        const size_t ro = (size_t)(2 * min(g, last_pair)) * x_cstride;
        if constexpr (XV2) {
            va = *reinterpret_cast<const u2_f32x2*>(xa + ro + x_lane_off);
            vb = *reinterpret_cast<const u2_f32x2*>(xb + ro + x_lane_off);
        } else {
            va[0] = xa[ro + x_lane_off]; va[1] = xa[ro + x_lane_off + 1];
            vb[0] = xb[ro + x_lane_off]; vb[1] = xb[ro + x_lane_off + 1];
        }
        h_a = xa[ro + h_lane_off];
        h_b = xb[ro + h_lane_off];
    };
#pragma unroll
    for (int u = 0; u < PF; ++u) load_pair(u, qa[u], qb[u], ha[u], hb[u]);

    u2_f32x4 ra[NA];
#pragma unroll
    for (int n = 0; n < NA; ++n) {
        const int idx = tid + 256 * n;
        if ((A_VEC % 256 == 0) || idx < A_VEC) reinterpret_cast<u2_f32x4*>(As[0])[idx] = wbase[idx];
    }
    __syncthreads();

    // the staged value: GN_a(a) + GN_b(b), then ELU (pad_act_kernel's arithmetic)
    auto act = [&](float va, float vb, float4 g) __attribute__((always_inline)) -> float {
        float v = va, w = vb;
        if (gn_in) {
            v = ((v - mu_a) * rs_a) * g.x + g.y;
            w = ((w - mu_b) * rs_b) * g.z + g.w;
        }
        v = v + w;
        return p.elu ? nc_eluf(v) : v;
    };

    for (int cb = 0; cb < n_cb; ++cb) {
        const int cur = cb & 1;
        const bool more = cb + 1 < n_cb;
        if (more) {
            const u2_f32x4* src = wbase + (size_t)(cb + 1) * A_VEC;
#pragma unroll
            for (int n = 0; n < NA; ++n) ra[n] = src[(A_VEC % 256 == 0) ? (unsigned)(tid + 256 * n) : min((unsigned)(tid + 256 * n), (unsigned)(A_VEC - 1))];
        }
        const float* Ac = As[cur] + hi * BM + nc_a_lane_off<TM>(l31);
        u2_static_for<CB / 2>([&](auto pt) __attribute__((always_inline)) {
            constexpr int pr = decltype(pt)::value;
            const int g = cb * (CB / 2) + pr;
            const float4 gt = gn_in ? Gt[2 * g + hi] : make_float4(1.0f, 0.0f, 1.0f, 0.0f);
            const u2_f32x2 rawa = qa[pr % PF], rawb = qb[pr % PF];
            const float hra = ha[pr % PF], hrb = hb[pr % PF];
            load_pair(g + PF, qa[pr % PF], qb[pr % PF], ha[pr % PF], hb[pr % PF]);
            float a = act(rawa[0], rawb[0], gt), bb = act(rawa[1], rawb[1], gt), hv = act(hra, hrb, gt);
            a = ok0 ? a : 0.0f;
            bb = ok1 ? bb : 0.0f;
            hv = halo_ok ? hv : 0.0f;
            float aL = u2_from_left(bb);                       // x[col - 1]: the left lane's second value (the span's halo for its first lane)
            aL = lane_first ? hv : aL;
            const float xaL = u2_other_half(aL, hi), xa_ = u2_other_half(a, hi), xbb = u2_other_half(bb, hi);
            // step 0: (c0,x[q] | c0,x[q-1])   step 1: (c1,x[q] | c1,x[q-1]);  column j = 0: q = col, j = 1: q = col + 1
            const float s0[2] = {hi ? xaL : a, hi ? xa_ : bb};
            const float s1[2] = {hi ? aL : xa_, hi ? a : xbb};
            float fa[TM];
            nc_load_a_frag<TM>(Ac + 2 * (2 * pr) * BM, l31, fa);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], s0[0], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], s0[1], acc[i][1], 0, 0, 0);
            }
            nc_load_a_frag<TM>(Ac + 2 * (2 * pr + 1) * BM, l31, fa);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], s1[0], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], s1[1], acc[i][1], 0, 0, 0);
            }
        });
        if (more) {
#pragma unroll
            for (int n = 0; n < NA; ++n) {
                const int idx = tid + 256 * n;
                if ((A_VEC % 256 == 0) || idx < A_VEC) reinterpret_cast<u2_f32x4*>(As[cur ^ 1])[idx] = ra[n];
            }
        }
        __syncthreads();
    }

    // ---- epilogue: D[row R = (r&3) + 8*(r>>2) + 4*hi (+32 i)][column col + j]; R = co*2 + phase
    const int rows_total = S * p.Cout - co_tile * BM;
    if (p.gn_part != nullptr) {
        double* const gp = p.gn_part + (int64_t)b * p.gn_nrb * p.gn_ncb * 2;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            double a1[2], a2[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float vv[16];
                unsigned okm16 = 0;
                const bool colok = col + j < ncols;
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
            const int rbk = co_tile * TM + i;
            if ((lane & 47) == 0 && rbk < p.gn_nrb && cbk < p.gn_ncb) nc_gn_store_partial(gp + ((int64_t)rbk * p.gn_ncb + cbk) * 2, s1, s2);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (p.gn_count != nullptr)
            nc_gn_arrive_and_finish(gp, p.gn_count + b, p.gn_stats + 2 * b, p.gn_nrb * p.gn_ncb, (unsigned)(p.n_t_tiles * p.n_co_tiles), p.gn_n);
    }
    if (col >= ncols) return;
    // output channel co = R / S holds S consecutive registers = its S phases: samples S col .. S col + 2 S - 1 of its row for the lane's two columns
    float* const yb = p.y + (int64_t)b * p.y_bstride + (unsigned)(S * col);
    const bool second = col + 1 < ncols;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; r += S) {
            const int R = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;            // a multiple of S: phase 0 of channel (co_tile * BM + R) / S
            if (R >= rows_total) continue;
            float* yr = yb + (size_t)((co_tile * BM + R) / S) * (unsigned)p.y_cstride;
            const float b0 = Ep[R];
            if constexpr (S == 2) {
                const u2_f32x2 v0 = {acc[i][0][r] + b0, acc[i][0][r + 1] + b0};
                *reinterpret_cast<u2_f32x2*>(yr) = v0;
                if (second) {
                    const u2_f32x2 v1 = {acc[i][1][r] + b0, acc[i][1][r + 1] + b0};
                    *reinterpret_cast<u2_f32x2*>(yr + 2) = v1;
                }
            } else {
                const u2_f32x4 v0 = {acc[i][0][r] + b0, acc[i][0][r + 1] + b0, acc[i][0][r + 2] + b0, acc[i][0][r + 3] + b0};
                *reinterpret_cast<u2_f32x4*>(yr) = v0;
                if (second) {
                    const u2_f32x4 v1 = {acc[i][1][r] + b0, acc[i][1][r + 1] + b0, acc[i][1][r + 2] + b0, acc[i][1][r + 3] + b0};
                    *reinterpret_cast<u2_f32x4*>(yr + 4) = v1;
                }
            }
        }
}

bool launch_up2(const Up2Args& a, int TM, int S, bool aligned, hipStream_t stream) {
    void (*fn)(const Up2Args) = nullptr;
    if (TM == 2 && S == 2) fn = aligned ? &up2_kernel<2, 2, true> : &up2_kernel<2, 2, false>;
    else if (TM == 4 && S == 4) fn = aligned ? &up2_kernel<4, 4, true> : &up2_kernel<4, 4, false>;
    if (!fn) return false;
    hipLaunchKernelGGL(fn, dim3((unsigned)((int64_t)a.B * a.n_t_tiles * a.n_co_tiles)), dim3(256), 0, stream, a);
    NC_HIP(hipGetLastError());
    return true;
}

}  // namespace nc
