// Thin-output convolution: Cout <= 4 (the decoders' PCM head: Conv1d(C, 1 or 2, k=7) + tanh, Decoder.cs:44-46, Modules/SNAC/Decoder.cs,
// SEANetDecoder final conv).  On the matrix-core template such a layer fills 1-2 of 32 tile rows and runs at its instruction rate;
// the work is really one pass over the input tensor (HBM-bound: 4*Cin bytes per output step), so it gets a streaming kernel:
// one workgroup = 1024 output steps of one clip, CC input channels at a time staged in LDS (coalesced 16-byte reads, zero padding by
// predicate), each thread 4 consecutive outputs x all output channels with the weights in scalar registers.
// Arithmetic per output = the canonical chain: fmaf over kk = ci*K + k ascending from +0, then + bias, then tanh.
#include <cstdlib>
#include <type_traits>

#include "nc_conv.h"
#include "nc_gn.h"
#include "nc_math.h"

namespace nc {

typedef float thin_f32x4 __attribute__((ext_vector_type(4)));

constexpr int THIN_TILE = 1024;   // output steps per workgroup
constexpr int THIN_CC = 8;        // input channels staged per round

// VEC: the window starts at a 16-byte boundary of the row (sh = 0..3 slots before position t0 - pad) and is read with 16-byte
// vector loads (2 per thread and channel instead of 5 dword reads); needs 16-byte aligned rows whose length is a multiple of 4.
template <int COUT, int K, bool VEC>
__global__ __launch_bounds__(256) void conv_thin_kernel(const float* __restrict__ x, int64_t x_bstride, int64_t x_cstride, int Cin, int x_len,
                                                        const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y,
                                                        int64_t y_bstride, int64_t y_cstride, int Tout, int pad, int dil, int n_t_tiles, int tanh_out) {
    extern __shared__ __attribute__((aligned(16))) float xs[];   // [CC][row], row = THIN_TILE + (K-1)*dil rounded up to 4
    const int tid = threadIdx.x;
    const int b = blockIdx.x / n_t_tiles, tt = blockIdx.x - b * n_t_tiles;
    const int t0 = tt * THIN_TILE;
    const int halo = (K - 1) * dil;
    const int row = (THIN_TILE + halo + 3 + 4) & ~3;
    const float* xb = x + (int64_t)b * x_bstride;
    const int sh = VEC ? ((4 - (pad & 3)) & 3) : 0;
    const int g0 = t0 - pad - sh;         // input position of window slot 0 (VEC: a multiple of 4)

    float acc[COUT][4];
#pragma unroll
    for (int c = 0; c < COUT; ++c)
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[c][o] = 0.0f;

    constexpr int NJ = 5;                 // window slots per thread per channel row (row <= 1280)
    // VEC: the reads of round r+1 are issued before the arithmetic of round r and land under it (registers carry them across)
    constexpr int NQ = 2;
    thin_f32x4 rq[VEC ? THIN_CC : 1][NQ];
    const int nwords = row >> 2, xw4 = x_len >> 2;
    auto issue_round = [&](int c0) __attribute__((always_inline)) {
        if constexpr (VEC) {
            // 16-byte words: word tid and (lanes 0..3: the halo) word 256 + tid of every channel row, all in flight together
#pragma unroll
            for (int c = 0; c < THIN_CC; ++c) {
                const thin_f32x4* xr = reinterpret_cast<const thin_f32x4*>(xb + (int64_t)min(c0 + c, Cin - 1) * x_cstride);
#pragma unroll
                for (int u = 0; u < NQ; ++u) {
                    const int wq = (g0 >> 2) + tid + 256 * u;        // g0 is a multiple of 4 (arithmetic shift: floor)
                    rq[c][u] = thin_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                    if (u == 0 || tid + 256 * u < nwords) rq[c][u] = xr[min(max(wq, 0), xw4 - 1)];   // clamped: always in bounds
                }
            }
        }
    };
    issue_round(0);
    for (int c0 = 0; c0 < Cin; c0 += THIN_CC) {
        const int nc = min(THIN_CC, Cin - c0);
        if constexpr (VEC) {
#pragma unroll
            for (int c = 0; c < THIN_CC; ++c)
#pragma unroll
                for (int u = 0; u < NQ; ++u) {
                    const int wq = (g0 >> 2) + tid + 256 * u;
                    if (wq < 0 || wq >= xw4) rq[c][u] = thin_f32x4{0.0f, 0.0f, 0.0f, 0.0f};   // (x_len % 4 == 0: words are all-in or all-out)
                }
            __syncthreads();   // the previous round's arithmetic has finished reading the window
#pragma unroll
            for (int c = 0; c < THIN_CC; ++c)
#pragma unroll
                for (int u = 0; u < NQ; ++u) {
                    const int j = tid + 256 * u;
                    if (j < nwords) reinterpret_cast<thin_f32x4*>(xs + c * row)[j] = rq[c][u];
                }
            __syncthreads();
            if (c0 + THIN_CC < Cin) issue_round(c0 + THIN_CC);
        } else {
        // all reads of the round are issued before the first LDS store (a store between them would serialise the round trips)
        float r[THIN_CC][NJ];
#pragma unroll
        for (int c = 0; c < THIN_CC; ++c) {
            const float* xr = xb + (int64_t)min(c0 + c, Cin - 1) * x_cstride;
#pragma unroll
            for (int u = 0; u < NJ; ++u) r[c][u] = xr[min(max(g0 + tid + 256 * u, 0), x_len - 1)];   // clamped: always in bounds
        }
        // (opaque uses after the last read keep every read unconditional and in flight together; the zero padding is a select)
#pragma unroll
        for (int c = 0; c < THIN_CC; ++c)
#pragma unroll
            for (int u = 0; u < NJ; ++u) {
                asm volatile("" : "+v"(r[c][u]));
                const int g = g0 + tid + 256 * u;
                r[c][u] = (g >= 0 && g < x_len) ? r[c][u] : 0.0f;
            }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < THIN_CC; ++c)
#pragma unroll
            for (int u = 0; u < NJ; ++u) {
                const int j = tid + 256 * u;
                if (j < row) xs[c * row + j] = r[c][u];
            }
        __syncthreads();
        }
        for (int c = 0; c < nc; ++c) {
            const float* xl = xs + c * row + 4 * tid;
            float wv[COUT][K];
#pragma unroll
            for (int co = 0; co < COUT; ++co)
#pragma unroll
                for (int k = 0; k < K; ++k) wv[co][k] = w[((int64_t)co * Cin + (c0 + c)) * K + k];   // uniform: scalar loads
            if (dil == 1) {
                // the 4 outputs of this thread read window slots [4*tid, 4*tid + K + 3): whole 16-byte words
                constexpr int NV = (K + 3 + 3 + (VEC ? 3 : 0)) / 4;   // (VEC: up to 3 slots of alignment shift)
                float xw[4 * NV];
#pragma unroll
                for (int q = 0; q < NV; ++q) {
                    const thin_f32x4 v = *reinterpret_cast<const thin_f32x4*>(xl + 4 * q);
                    xw[4 * q] = v[0]; xw[4 * q + 1] = v[1]; xw[4 * q + 2] = v[2]; xw[4 * q + 3] = v[3];
                }
#pragma unroll
                for (int k = 0; k < K; ++k)
#pragma unroll
                    for (int co = 0; co < COUT; ++co)
#pragma unroll
                        for (int o = 0; o < 4; ++o) acc[co][o] = fmaf(wv[co][k], xw[k + o + sh], acc[co][o]);
            } else {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    float xv[4];
#pragma unroll
                    for (int o = 0; o < 4; ++o) xv[o] = xl[k * dil + o + sh];
#pragma unroll
                    for (int co = 0; co < COUT; ++co)
#pragma unroll
                        for (int o = 0; o < 4; ++o) acc[co][o] = fmaf(wv[co][k], xv[o], acc[co][o]);
                }
            }
        }
    }
    const int t = t0 + 4 * tid;
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        const float bv = bias ? bias[co] : 0.0f;
        float* yr = y + (int64_t)b * y_bstride + (int64_t)co * y_cstride;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            float v = acc[co][o] + bv;
            if (tanh_out) v = nc_tanhf(v);
            if (t + o < Tout) yr[t + o] = v;
        }
    }
}

template <int COUT>
static bool launch_thin_k(int K, bool vec, dim3 grid, size_t lds, hipStream_t s, const float* x, int64_t xb, int64_t xc, int Cin, int x_len, const float* w,
                          const float* bias, float* y, int64_t yb, int64_t yc, int Tout, int pad, int dil, int ntt, int tanh_out) {
    switch (K) {
        case 7:
            if (vec) hipLaunchKernelGGL((conv_thin_kernel<COUT, 7, true>), grid, dim3(256), lds, s, x, xb, xc, Cin, x_len, w, bias, y, yb, yc, Tout, pad, dil, ntt, tanh_out);
            else hipLaunchKernelGGL((conv_thin_kernel<COUT, 7, false>), grid, dim3(256), lds, s, x, xb, xc, Cin, x_len, w, bias, y, yb, yc, Tout, pad, dil, ntt, tanh_out);
            return true;
        case 3: hipLaunchKernelGGL((conv_thin_kernel<COUT, 3, false>), grid, dim3(256), lds, s, x, xb, xc, Cin, x_len, w, bias, y, yb, yc, Tout, pad, dil, ntt, tanh_out); return true;
        case 1: hipLaunchKernelGGL((conv_thin_kernel<COUT, 1, false>), grid, dim3(256), lds, s, x, xb, xc, Cin, x_len, w, bias, y, yb, yc, Tout, pad, dil, ntt, tanh_out); return true;
    }
    return false;
}

// dense weights [Cout][Cin][K]; returns false when the shape has no instantiation (the caller takes the generic template)
bool launch_conv_thin(const float* x, int64_t x_bstride, int64_t x_cstride, int Cin, int x_len, const float* w_dense, const float* bias, float* y,
                      int64_t y_bstride, int64_t y_cstride, int B, int Cout, int K, int pad, int dil, int64_t Tout, bool tanh_out, hipStream_t s) {
    if (Cout < 1 || Cout > 2 || (K != 7 && K != 3 && K != 1) || Tout <= 0) return false;
    const int halo = (K - 1) * dil;
    const int row = (THIN_TILE + halo + 3 + 3 + 4) & ~3;   // + up to 3 slots of alignment shift + one spare 16-byte word for the whole-word reads
    if (row > 1280) return false;
    // 16-byte window reads: k = 7, dilation 1 (the PCM heads), rows that start on 16-byte boundaries and hold whole 16-byte words
    static const bool no_vec = env_present("NC_THIN_NO_VEC");
    const bool vec = !no_vec && K == 7 && dil == 1 && (x_len & 3) == 0 && x_len >= 4 && (x_bstride & 3) == 0 && (x_cstride & 3) == 0 &&
                     (reinterpret_cast<uintptr_t>(x) & 15) == 0 && row <= 4 * 512;
    const size_t lds = sizeof(float) * (size_t)THIN_CC * row;
    if (lds > 64 * 1024) return false;
    const int ntt = (int)((Tout + THIN_TILE - 1) / THIN_TILE);
    const dim3 grid((unsigned)(B * ntt));
    bool ok = Cout == 1 ? launch_thin_k<1>(K, vec, grid, lds, s, x, x_bstride, x_cstride, Cin, x_len, w_dense, bias, y, y_bstride, y_cstride, (int)Tout, pad, dil, ntt, tanh_out)
                        : launch_thin_k<2>(K, vec, grid, lds, s, x, x_bstride, x_cstride, Cin, x_len, w_dense, bias, y, y_bstride, y_cstride, (int)Tout, pad, dil, ntt, tanh_out);
    if (ok) NC_HIP(hipGetLastError());
    return ok;
}

// Thin-output convolution in the Encodec input mode (the decoder's last SConv1d(32 -> 2, k = 7), SEANetDecoder.cs:140-148, fed by the sum
// of a residual block's shortcut and branch, each a raw conv output with a pending GroupNorm; SConv1d.cs:144-173 reflect pad 3 + 3;
// NormConv1d.cs:155 GroupNorm(1, 2) over its own output).  Everything the summed / activated / padded copy and the stand-alone statistics
// pass did happens here: a window slot reads sample q = reflect(j - left) of both operands, normalises each with its own statistics,
// adds, applies the ELU (the zero extension stays zero) and goes to LDS; the outputs' GroupNorm block sums (nc_gn.h: the two live rows
// of a 32-row block, 32-column blocks = 8 consecutive threads) are reduced in registers and finished in the launch.
constexpr int THIN_INM_CC = 4;
template <int COUT>
__global__ __launch_bounds__(256) void conv_thin_inm_kernel(const ThinInmArgs a) {
    constexpr int K = 7, NJ = 5, ROW = (THIN_TILE + K - 1 + 3 + 4) & ~3;
    __shared__ __attribute__((aligned(16))) float xs[THIN_INM_CC * ROW];
    const int tid = threadIdx.x;
    const int b = blockIdx.x / a.n_t_tiles, tt = blockIdx.x - b * a.n_t_tiles;
    const int t0 = tt * THIN_TILE;
    const float* xa = a.xa + (int64_t)b * a.x_bstride;
    const float* xb = a.xb2 ? a.xb2 + (int64_t)b * a.x_bstride : nullptr;
    const bool gn = a.stats_a != nullptr;
    const float mu_a = gn ? a.stats_a[2 * b] : 0.0f, rs_a = gn ? a.stats_a[2 * b + 1] : 1.0f;
    const float mu_b = (gn && xb) ? a.stats_b[2 * b] : 0.0f, rs_b = (gn && xb) ? a.stats_b[2 * b + 1] : 1.0f;
    // this thread's window slots: padded positions t0 + tid + 256*u -> source sample (SConv1d.Pad1d as an index map)
    int qs[NJ];
    bool ok[NJ];
#pragma unroll
    for (int u = 0; u < NJ; ++u) {
        const int j = t0 + tid + 256 * u;
        int q = j - a.left;
        q = q < 0 ? -q : q;
        if (q >= a.Lz) q = 2 * (a.Lz - 1) - q;
        ok[u] = j < a.Lp && q >= 0 && q < a.L;
        qs[u] = min(max(q, 0), a.L - 1);
    }
    float acc[COUT][4];
#pragma unroll
    for (int c = 0; c < COUT; ++c)
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[c][o] = 0.0f;
    for (int c0 = 0; c0 < a.Cin; c0 += THIN_INM_CC) {
        const int nc = min(THIN_INM_CC, a.Cin - c0);
        float ra[THIN_INM_CC][NJ], rb[THIN_INM_CC][NJ];
#pragma unroll
        for (int c = 0; c < THIN_INM_CC; ++c) {
            const int64_t ro = (int64_t)min(c0 + c, a.Cin - 1) * a.x_cstride;
#pragma unroll
            for (int u = 0; u < NJ; ++u) {
                ra[c][u] = xa[ro + qs[u]];
                rb[c][u] = xb ? xb[ro + qs[u]] : 0.0f;
            }
        }
#pragma unroll
        for (int c = 0; c < THIN_INM_CC; ++c) {
            const int ci = min(c0 + c, a.Cin - 1);
            const float ga = gn ? a.gamma_a[ci] : 1.0f, ba = gn ? a.beta_a[ci] : 0.0f;
            const float gb = (gn && xb) ? a.gamma_b[ci] : 1.0f, bb = (gn && xb) ? a.beta_b[ci] : 0.0f;
#pragma unroll
            for (int u = 0; u < NJ; ++u) {
                float v = ra[c][u];
                if (gn) v = ((v - mu_a) * rs_a) * ga + ba;
                if (xb) {
                    float w2 = rb[c][u];
                    if (gn) w2 = ((w2 - mu_b) * rs_b) * gb + bb;
                    v = v + w2;
                }
                if (a.elu) v = nc_eluf(v);
                ra[c][u] = ok[u] ? v : 0.0f;
            }
        }
        __syncthreads();   // the previous round's arithmetic has finished reading the window
#pragma unroll
        for (int c = 0; c < THIN_INM_CC; ++c)
#pragma unroll
            for (int u = 0; u < NJ; ++u) {
                const int j = tid + 256 * u;
                if (j < ROW) xs[c * ROW + j] = ra[c][u];
            }
        __syncthreads();
        for (int c = 0; c < nc; ++c) {
            const float* xl = xs + c * ROW + 4 * tid;
            float wv[COUT][K];
#pragma unroll
            for (int co = 0; co < COUT; ++co)
#pragma unroll
                for (int k = 0; k < K; ++k) wv[co][k] = a.w[((int64_t)co * a.Cin + (c0 + c)) * K + k];   // uniform: scalar loads
            constexpr int NV = (K + 3 + 3) / 4;
            float xw[4 * NV];
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const thin_f32x4 v = *reinterpret_cast<const thin_f32x4*>(xl + 4 * q);
                xw[4 * q] = v[0]; xw[4 * q + 1] = v[1]; xw[4 * q + 2] = v[2]; xw[4 * q + 3] = v[3];
            }
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int co = 0; co < COUT; ++co)
#pragma unroll
                    for (int o = 0; o < 4; ++o) acc[co][o] = fmaf(wv[co][k], xw[k + o], acc[co][o]);
        }
    }
    const int t = t0 + 4 * tid;
    float yv[COUT][4];
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        const float bv = a.bias ? a.bias[co] : 0.0f;
#pragma unroll
        for (int o = 0; o < 4; ++o) yv[co][o] = acc[co][o] + bv;
    }
    if (a.gn_part != nullptr) {
        // block sums: slot (h = 0, column c) adds rows 0 .. COUT-1 of column c in binary64 from +0; butterfly 1, 2 in this thread's 4
        // columns, 4 / 8 / 16 over the 8 threads of a 32-column block (the h = 1 half of the block holds no rows: + 0)
        double p1[4], p2[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int co = 0; co < COUT; ++co) {
                const double d = (t + o < a.Tout) ? (double)yv[co][o] : 0.0;
                s1 += d;
                s2 = __builtin_fma(d, d, s2);
            }
            p1[o] = s1; p2[o] = s2;
        }
        double s1 = (p1[0] + p1[1]) + (p1[2] + p1[3]), s2 = (p2[0] + p2[1]) + (p2[2] + p2[3]);
        s1 += nc_gn_dpp<0xB1>(s1); s2 += nc_gn_dpp<0xB1>(s2);
        s1 += nc_gn_dpp<0x4E>(s1); s2 += nc_gn_dpp<0x4E>(s2);
        s1 += nc_gn_dpp<0x141>(s1); s2 += nc_gn_dpp<0x141>(s2);
        double* const gp = a.gn_part + (int64_t)b * a.gn_ncb * 2;
        const int cbk = t >> 5;
        if ((tid & 7) == 0 && cbk < a.gn_ncb) nc_gn_store_partial(gp + (int64_t)cbk * 2, s1, s2);
        if (a.gn_count != nullptr) nc_gn_arrive_and_finish(gp, a.gn_count + b, a.gn_stats + 2 * b, a.gn_ncb, (unsigned)a.n_t_tiles, a.gn_n);
    }
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        float* yr = a.y + (int64_t)b * a.y_bstride + (int64_t)co * a.y_cstride;
#pragma unroll
        for (int o = 0; o < 4; ++o)
            if (t + o < a.Tout) yr[t + o] = yv[co][o];
    }
}

bool launch_conv_thin_inm(const ThinInmArgs& a, int B, int Cout, hipStream_t s) {
    if (Cout < 1 || Cout > 2 || a.Tout <= 0) return false;
    ThinInmArgs k = a;
    k.n_t_tiles = (int)((a.Tout + THIN_TILE - 1) / THIN_TILE);
    const dim3 grid((unsigned)(B * k.n_t_tiles));
    if (Cout == 1) hipLaunchKernelGGL(conv_thin_inm_kernel<1>, grid, dim3(256), 0, s, k);
    else hipLaunchKernelGGL(conv_thin_inm_kernel<2>, grid, dim3(256), 0, s, k);
    NC_HIP(hipGetLastError());
    return true;
}

// Thin-INPUT convolution: the encoders' stem Conv1d(1, C, k=7, pad 3) (Encoder.cs:31, Modules/SNAC/Encoder.cs:33).  On the matrix-core
// template the single input channel is padded to a reduction block of 8; the work is really one streaming STORE of C rows
// (HBM-bound: 4*Cout bytes per input step).  One workgroup = 1024 steps of one clip: each thread keeps its K+3 input samples in
// registers (zero padding and DAC.Preprocess's right pad by predicate), walks the output channels with the K weights + bias of a
// channel read from LDS (one 32-byte row per channel), and writes 4 consecutive samples per channel: a wavefront's store is 1 KB of
// one output row.  Arithmetic per output = the canonical chain: fmaf over k ascending from +0, then + bias (then the next layer's
// Snake when the consumer fused it into this store).
template <int K>
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ x, int64_t x_bstride, int x_len, const float* __restrict__ w,
                                                        const float* __restrict__ bias, const float* __restrict__ alpha_out, float* __restrict__ y,
                                                        int64_t y_bstride, int64_t y_cstride, int Cout, int Tout, int pad, int n_t_tiles, int vec_ok) {
    extern __shared__ __attribute__((aligned(16))) float ws[];   // [Cout][8]: K weights, bias | then [Cout][2]: alpha, 1/alpha
    const int tid = threadIdx.x;
    const int b = blockIdx.x / n_t_tiles, tt = blockIdx.x - b * n_t_tiles;
    const int t = tt * THIN_TILE + 4 * tid;
    const float* xb = x + (int64_t)b * x_bstride;
    float xw[K + 3];
#pragma unroll
    for (int i = 0; i < K + 3; ++i) {
        const int g = t - pad + i;
        const float v = xb[min(max(g, 0), x_len - 1)];
        xw[i] = (g >= 0 && g < x_len) ? v : 0.0f;
    }
    for (int i = tid; i < Cout * 8; i += 256) {
        const int co = i >> 3, k = i & 7;
        ws[i] = k < K ? w[co * K + k] : (k == 7 && bias ? bias[co] : 0.0f);
    }
    float* al = ws + Cout * 8;
    if (alpha_out)
        for (int i = tid; i < Cout; i += 256) { const float a = alpha_out[i]; al[2 * i] = a; al[2 * i + 1] = nc_snake_inv(a); }
    __syncthreads();
    if (t >= Tout) return;
    float* yr = y + (int64_t)b * y_bstride + t;
    for (int co = 0; co < Cout; ++co) {
        const thin_f32x4 w0 = *reinterpret_cast<const thin_f32x4*>(ws + co * 8), w1 = *reinterpret_cast<const thin_f32x4*>(ws + co * 8 + 4);
        const float wk[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
        thin_f32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float a = 0.0f;
#pragma unroll
            for (int k = 0; k < K; ++k) a = fmaf(wk[k], xw[k + q], a);
            a = a + wk[7];
            o[q] = a;
        }
        if (alpha_out) {
            const float a = al[2 * co], ai = al[2 * co + 1];
#pragma unroll
            for (int q = 0; q < 4; q += 2) {
                const nc_f2 r = nc_snakef2(nc_f2{o[q], o[q + 1]}, nc_f2{a, a}, nc_f2{ai, ai});   // two values per packed instruction (nc_math.h)
                o[q] = r[0];
                o[q + 1] = r[1];
            }
        }
        float* yp = yr + (int64_t)co * y_cstride;
        if (vec_ok && t + 3 < Tout) {
            *reinterpret_cast<thin_f32x4*>(yp) = o;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (t + q < Tout) yp[q] = o[q];
        }
    }
}

// Cin == 1, stride 1, dilation 1, K <= 7: dense weights [Cout][1][K]; returns false when the shape has no instantiation
bool launch_conv_stem(const float* x, int64_t x_bstride, int x_len, const float* w_dense, const float* bias, const float* alpha_out, float* y,
                      int64_t y_bstride, int64_t y_cstride, int B, int Cout, int K, int pad, int64_t Tout, hipStream_t s) {
    if (K != 7 || Cout < 1 || Cout > 1024 || Tout <= 0 || x_len <= 0) return false;
    const int ntt = (int)((Tout + THIN_TILE - 1) / THIN_TILE);
    const size_t lds = sizeof(float) * (size_t)Cout * 10;
    const int vec_ok = ((reinterpret_cast<uintptr_t>(y) & 15) == 0 && (y_bstride & 3) == 0 && (y_cstride & 3) == 0) ? 1 : 0;
    hipLaunchKernelGGL((stem_conv_kernel<7>), dim3((unsigned)(B * ntt)), dim3(256), lds, s, x, x_bstride, x_len, w_dense, bias, alpha_out, y, y_bstride,
                       y_cstride, Cout, (int)Tout, pad, ntt, vec_ok);
    NC_HIP(hipGetLastError());
    return true;
}

}  // namespace nc
