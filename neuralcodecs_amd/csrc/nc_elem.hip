// HBM-bound kernels of the codec path that are not dense contractions: depthwise k-tap convolution (SNAC), average pooling and
// repeat-interleave RVQ update (SNAC multi-rate quantizer), LayerNorm + windowed rotary attention (SNAC LocalMHA), noise source.
//
// Reference call sites (under NeuralCodecs.Torch/):
//   depthwise conv   Modules/SNAC/ResidualUnit.cs:32 (groups == dim), Encoder.cs:55-62, Decoder.cs:45
//   avg_pool1d       Modules/SNAC/VectorQuantizer.cs:88 ; repeat_interleave  VectorQuantizer.cs:100, ResidualVectorQuantizer.cs:120
//   LayerNorm / SDPA Modules/SNAC/LocalMHA.cs:85,105 ; rotary  RotaryEmbedding.cs:46-68 ; randn  NoiseBlock.cs:41
// Arithmetic is the canonical arithmetic of DESIGN.md (same sequences as oracle/c/nc_ref_snac.c).
#include <cstdlib>

#include "nc_elem.h"
#include "nc_math.h"

namespace nc {

// ---------------------------------------------------------------------------------------------- depthwise conv
// y[b,c,t] = (chain_k w[c,k] * snake_in(x[b,c,t + k*dil - pad])) + bias[c]  [-> snake_out]
// One block = one (clip, channel) row segment of DW_TT outputs; the activated input window lives in LDS.
constexpr int DW_TT = 2048;
constexpr int DW_MAXK = 7;

__global__ __launch_bounds__(256) void dwconv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, const float* __restrict__ alpha_in,
                                                     const float* __restrict__ alpha_out, float* __restrict__ y, int C, int T,
                                                     int K, int dil, int pad) {
    extern __shared__ float win[];
    const int tile = blockIdx.x, c = blockIdx.y, b = blockIdx.z;
    const int t0 = tile * DW_TT;
    const int span = DW_TT + (K - 1) * dil;
    const float* xr = x + ((int64_t)b * C + c) * T;
    const float ai = alpha_in ? alpha_in[c] : 0.0f;
    const float ai_inv = nc_snake_inv(ai);
    // window -> LDS: all reads of the window are in flight before the first LDS store (a read/store pair per iteration costs one
    // global round trip each); addresses clamped into the row, zero padding by select
    constexpr int NJ = (DW_TT + 6 * 9 + 255) / 256 + 1;   // slots per thread for K <= 7, dil <= 9 ... generic spans take more passes
    for (int j0 = 0; j0 < span; j0 += NJ * 256) {
        float r[NJ];
#pragma unroll
        for (int u = 0; u < NJ; ++u) r[u] = xr[min(max(t0 - pad + j0 + (int)threadIdx.x + 256 * u, 0), T - 1)];
#pragma unroll
        for (int u = 0; u < NJ; ++u) {
            asm volatile("" : "+v"(r[u]));
            const int gp = t0 - pad + j0 + (int)threadIdx.x + 256 * u;
            r[u] = (gp >= 0 && gp < T) ? r[u] : 0.0f;
        }
        if (alpha_in) {   // Snake on the way in, two values per packed instruction (nc_math.h)
#pragma unroll
            for (int u = 0; u + 1 < NJ; u += 2) nc_snake_pair(r[u], r[u + 1], ai, ai_inv, ai, ai_inv);
            if (NJ & 1) r[NJ - 1] = nc_snakef(r[NJ - 1], ai, ai_inv);
        }
#pragma unroll
        for (int u = 0; u < NJ; ++u) {
            const int j = j0 + threadIdx.x + 256 * u;
            if (j < span) win[j] = r[u];
        }
    }
    __syncthreads();
    float wk[DW_MAXK];
#pragma unroll
    for (int k = 0; k < DW_MAXK; ++k) wk[k] = k < K ? w[c * K + k] : 0.0f;
    const float bv = bias ? bias[c] : 0.0f;
    const float ao = alpha_out ? alpha_out[c] : 0.0f;
    const float ao_inv = nc_snake_inv(ao);
    float* yr = y + ((int64_t)b * C + c) * T;
    float ov[DW_TT / 256];
#pragma unroll
    for (int i = 0; i < DW_TT / 256; ++i) {
        const int lt = threadIdx.x + 256 * i;
        float a = 0.0f;
#pragma unroll
        for (int k = 0; k < DW_MAXK; ++k)
            if (k < K) a = nc_fma(wk[k], win[lt + k * dil], a);
        ov[i] = a + bv;
    }
    if (alpha_out) {
#pragma unroll
        for (int i = 0; i < DW_TT / 256; i += 2) nc_snake_pair(ov[i], ov[i + 1], ao, ao_inv, ao, ao_inv);
    }
#pragma unroll
    for (int i = 0; i < DW_TT / 256; ++i) {
        const int t = t0 + (int)threadIdx.x + 256 * i;
        if (t < T) yr[t] = ov[i];
    }
}

// Vector form for the residual units' k = 7 depthwise convolutions (dilation 1 / 3 / 9, "same" padding 3*dil; rows 16-byte aligned,
// T a multiple of 4): the window starts at a 16-byte boundary of the row (SH slots before position t0 - pad), is read with 16-byte
// loads, activated two values per packed instruction and stored with ds_write_b128; a thread computes two groups of 4 consecutive
// outputs (group stride 1024: the b128 LDS reads of a wavefront are 16 bytes apart, conflict-free) and writes 16-byte stores.
// Same arithmetic per output as dwconv_kernel: fma chain over k ascending from +0, + bias, Snake.
typedef float dw_f32x4 __attribute__((ext_vector_type(4)));
template <int DIL>
__global__ __launch_bounds__(256) void dwconv_vec_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                         const float* __restrict__ alpha_in, const float* __restrict__ alpha_out,
                                                         float* __restrict__ y, int C, int T) {
    constexpr int K = 7, PAD = 3 * DIL, SH = (4 - (PAD & 3)) & 3;
    constexpr int SPAN = DW_TT + (K - 1) * DIL + SH, NWORDS = (SPAN + 3) / 4;        // window slots / 16-byte words
    constexpr int NW = (4 + (K - 1) * DIL + SH + 3) / 4;                              // words a group of 4 outputs reads
    __shared__ __attribute__((aligned(16))) float win[NWORDS * 4 + 4];
    const int tid = threadIdx.x, tile = blockIdx.x, c = blockIdx.y, b = blockIdx.z;
    const int t0 = tile * DW_TT, g0 = t0 - PAD - SH;                                  // input position of window slot 0 (a multiple of 4)
    const float* xr = x + ((int64_t)b * C + c) * T;
    const int xw4 = T >> 2;
    const float ai = alpha_in ? alpha_in[c] : 0.0f, ai_inv = nc_snake_inv(ai);
    constexpr int NL = (NWORDS + 255) / 256;
    dw_f32x4 r[NL];
#pragma unroll
    for (int u = 0; u < NL; ++u) {
        const int wq = (g0 >> 2) + tid + 256 * u;
        r[u] = reinterpret_cast<const dw_f32x4*>(xr)[min(max(wq, 0), xw4 - 1)];       // clamped: always in bounds
    }
#pragma unroll
    for (int u = 0; u < NL; ++u) {
        const int wq = (g0 >> 2) + tid + 256 * u;
        if (wq < 0 || wq >= xw4) r[u] = dw_f32x4{0.0f, 0.0f, 0.0f, 0.0f};             // (T % 4 == 0: a word is all in or all out)
        if (alpha_in) {
            const nc_f2 lo = nc_snakef2_m(nc_f2{r[u][0], r[u][1]}, nc_f2{ai, ai}, nc_f2{ai_inv, ai_inv});
            const nc_f2 hi = nc_snakef2_m(nc_f2{r[u][2], r[u][3]}, nc_f2{ai, ai}, nc_f2{ai_inv, ai_inv});
            r[u] = dw_f32x4{lo[0], lo[1], hi[0], hi[1]};
        }
        if (tid + 256 * u < NWORDS) reinterpret_cast<dw_f32x4*>(win)[tid + 256 * u] = r[u];
    }
    __syncthreads();
    float wk[K];
#pragma unroll
    for (int k = 0; k < K; ++k) wk[k] = w[c * K + k];
    const float bv = bias ? bias[c] : 0.0f;
    const float ao = alpha_out ? alpha_out[c] : 0.0f, ao_inv = nc_snake_inv(ao);
    float* yr = y + ((int64_t)b * C + c) * T;
#pragma unroll
    for (int grp = 0; grp < 2; ++grp) {
        const int lt = grp * 1024 + 4 * tid;                                          // first of this group's 4 outputs (tile-local)
        float wv[4 * NW];
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            const dw_f32x4 v = reinterpret_cast<const dw_f32x4*>(win)[(lt >> 2) + q];
            wv[4 * q] = v[0]; wv[4 * q + 1] = v[1]; wv[4 * q + 2] = v[2]; wv[4 * q + 3] = v[3];
        }
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float a = 0.0f;
#pragma unroll
            for (int k = 0; k < K; ++k) a = nc_fma(wk[k], wv[i + SH + k * DIL], a);
            o[i] = a + bv;
        }
        if (alpha_out) {
            const nc_f2 lo = nc_snakef2_m(nc_f2{o[0], o[1]}, nc_f2{ao, ao}, nc_f2{ao_inv, ao_inv});
            const nc_f2 hi = nc_snakef2_m(nc_f2{o[2], o[3]}, nc_f2{ao, ao}, nc_f2{ao_inv, ao_inv});
            o[0] = lo[0]; o[1] = lo[1]; o[2] = hi[0]; o[3] = hi[1];
        }
        const int t = t0 + lt;
        if (t < T) *reinterpret_cast<dw_f32x4*>(yr + t) = dw_f32x4{o[0], o[1], o[2], o[3]};   // (T % 4 == 0: a group is all in or all out)
    }
}

void launch_dwconv(const DwConvLayer& L, const float* x, const float* alpha_in, const float* alpha_out, float* y, int B, int64_t T,
                   hipStream_t s, Profiler* prof) {
    if (L.K > DW_MAXK) fail(NC_EUNSUPPORTED, "depthwise kernel size %d > %d", L.K, DW_MAXK);
    {
        static const bool no_vec = env_flag("NC_DW_NO_VEC");
        const bool al = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
        if (!no_vec && L.K == 7 && L.pad == 3 * L.dil && (L.dil == 1 || L.dil == 3 || L.dil == 9) && (T & 3) == 0 && T >= 4 && al) {
            dim3 grid((unsigned)((T + DW_TT - 1) / DW_TT), (unsigned)L.C, (unsigned)B);
            if (prof && prof->on) prof->begin(s, NC_KC_DWCONV, 2.0 * L.K * L.C * (double)T * B, 8.0 * L.C * (double)T * B);
            const float* bp = L.has_bias ? L.bias.as<float>() : nullptr;
            if (L.dil == 1) hipLaunchKernelGGL(dwconv_vec_kernel<1>, grid, dim3(256), 0, s, x, L.w.as<float>(), bp, alpha_in, alpha_out, y, L.C, (int)T);
            else if (L.dil == 3) hipLaunchKernelGGL(dwconv_vec_kernel<3>, grid, dim3(256), 0, s, x, L.w.as<float>(), bp, alpha_in, alpha_out, y, L.C, (int)T);
            else hipLaunchKernelGGL(dwconv_vec_kernel<9>, grid, dim3(256), 0, s, x, L.w.as<float>(), bp, alpha_in, alpha_out, y, L.C, (int)T);
            NC_HIP(hipGetLastError());
            if (prof && prof->on) prof->end(s);
            return;
        }
    }
    const size_t lds = sizeof(float) * (DW_TT + (L.K - 1) * L.dil);
    dim3 grid((unsigned)((T + DW_TT - 1) / DW_TT), (unsigned)L.C, (unsigned)B);
    if (prof && prof->on) prof->begin(s, NC_KC_DWCONV, 2.0 * L.K * L.C * (double)T * B, 8.0 * L.C * (double)T * B);
    hipLaunchKernelGGL(dwconv_kernel, grid, dim3(256), lds, s, x, L.w.as<float>(), L.has_bias ? L.bias.as<float>() : nullptr,
                       alpha_in, alpha_out, y, L.C, (int)T, L.K, L.dil, L.pad);
    NC_HIP(hipGetLastError());
    if (prof && prof->on) prof->end(s);
}

void DwConvLayer::build(const float* dense_w, const float* bias_h, int C_, int K_, int pad_, int dil_) {
    C = C_; K = K_; pad = pad_; dil = dil_;
    w.reserve(sizeof(float) * C * K);
    NC_HIP(hipMemcpy(w.p, dense_w, sizeof(float) * C * K, hipMemcpyHostToDevice));
    has_bias = bias_h != nullptr;
    if (has_bias) {
        bias.reserve(sizeof(float) * C);
        NC_HIP(hipMemcpy(bias.p, bias_h, sizeof(float) * C, hipMemcpyHostToDevice));
    }
}

// ---------------------------------------------------------------------------------------------- avg_pool1d(s)
__global__ void avg_pool_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t rows, int64_t T, int s) {
    const int64_t Ts = T / s;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * Ts) return;
    const int64_t r = i / Ts, t = i - r * Ts;
    const float* xp = x + r * T + t * s;
    float a = xp[0];
    for (int j = 1; j < s; ++j) a = a + xp[j];
    y[i] = a / (float)s;
}
void launch_avg_pool(const float* x, float* y, int64_t rows, int64_t T, int s, hipStream_t st, Profiler* prof) {
    const int64_t n = rows * (T / s);
    ProfScope ps(prof, st, NC_KC_ELEM, (double)rows * T, 4.0 * ((double)rows * T + (double)n));
    hipLaunchKernelGGL(avg_pool_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, y, rows, T, s);
    NC_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ zq (+)= repeat_interleave(q, s) ; residual -= ...
__global__ void rvq_update_kernel(const float* __restrict__ q, float* __restrict__ zq, float* __restrict__ residual, int64_t rows,
                                  int64_t T, int s, int first) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * T) return;
    const int64_t r = i / T, t = i - r * T;
    const float qv = q[r * (T / s) + t / s];
    zq[i] = first ? qv : zq[i] + qv;
    if (residual) residual[i] = residual[i] - qv;
}
void launch_rvq_update(const float* q, float* zq, float* residual, int64_t rows, int64_t T, int s, bool first, hipStream_t st,
                       Profiler* prof) {
    const int64_t n = rows * T;
    ProfScope ps(prof, st, NC_KC_ELEM, (residual ? 2.0 : 1.0) * n, 4.0 * ((double)n / s + (first ? 1.0 : 2.0) * n + (residual ? 2.0 * n : 0.0)));
    hipLaunchKernelGGL(rvq_update_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, q, zq, residual, rows, T, s,
                       first ? 1 : 0);
    NC_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------- LayerNorm over C
// x, y in [B,C,T] layout; one thread per (b,t), sequential binary64 sums over c (the canonical order).
__global__ void layernorm_ct_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                    float* __restrict__ y, int B, int C, int64_t T) {
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= (int64_t)B * T) return;
    const int64_t b = i / T, t = i - b * T;
    const float* xp = x + b * C * T + t;
    // Channel loops run in batches of 32 reads issued together (same accumulation order as a plain loop): the kernel has only B*T
    // threads (4608 at the C5 share) and is bound by its memory round trips, so each thread keeps 32 in flight; in the output loop
    // this also keeps the reads of a batch ahead of its stores -- a read behind a store waits for the store's acknowledgement.
    constexpr int U = 32;
    double s1 = 0.0;
    for (int c0 = 0; c0 < C; c0 += U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = xp[(int64_t)min(c0 + u, C - 1) * T];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (c0 + u < C) s1 += (double)v[u];
    }
    const double mu = s1 / C;
    double s2 = 0.0;
    for (int c0 = 0; c0 < C; c0 += U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = xp[(int64_t)min(c0 + u, C - 1) * T];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (c0 + u < C) {
                const double d = (double)v[u] - mu;
                s2 += d * d;
            }
    }
    const float r = (float)(1.0 / sqrt(s2 / C + 1e-5));
    const float muf = (float)mu;
    float* yp = y + b * C * T + t;
    for (int c0 = 0; c0 < C; c0 += U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = xp[(int64_t)min(c0 + u, C - 1) * T];
#pragma unroll
        for (int u = 0; u < U; ++u) asm volatile("" : "+v"(v[u]));
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (c0 + u < C) yp[(int64_t)(c0 + u) * T] = ((v[u] - muf) * r) * gamma[c0 + u] + beta[c0 + u];
    }
}
// Tile form (round 4): the kernel above has B*T threads that each walk C strided words three times -- 144 dependent memory round trips,
// 224 us for a 28 MB tensor.  Here a workgroup stages a [C][TT] tile (TT consecutive steps of one clip) in LDS with every load in
// flight at once, TT lanes walk their column out of LDS in the SAME sequential binary64 order (channels ascending: the canonical
// sums; the dependent chain of 2 C double additions is what is left of the run time), and all threads normalise and store.
template <int TT>
__global__ __launch_bounds__(256) void layernorm_tile_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ y, int C, int64_t T) {
    extern __shared__ __attribute__((aligned(16))) float ln_lds[];   // [C][TT] | mean[TT] | rstd[TT]
    float* tile = ln_lds;
    float* stat = ln_lds + (size_t)C * TT;
    const int b = blockIdx.y;
    const int64_t t0 = (int64_t)blockIdx.x * TT;
    const int col = threadIdx.x % TT, r0 = threadIdx.x / TT;
    constexpr int RP = 256 / TT;                                       // rows per pass
    const int64_t tc = t0 + col < T ? t0 + col : T - 1;
    const float* xp = x + (int64_t)b * C * T + tc;
    constexpr int U = 16;
    for (int c0 = r0; c0 < C; c0 += RP * U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = xp[(int64_t)min(c0 + u * RP, C - 1) * T];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (c0 + u * RP < C) tile[(c0 + u * RP) * TT + col] = v[u];
    }
    __syncthreads();
    if (threadIdx.x < TT) {
        const float* cp = tile + threadIdx.x;
        double s1 = 0.0;
        for (int c = 0; c < C; ++c) s1 += (double)cp[c * TT];
        const double mu = s1 / C;
        double s2 = 0.0;
        for (int c = 0; c < C; ++c) {
            const double d = (double)cp[c * TT] - mu;
            s2 += d * d;
        }
        stat[threadIdx.x] = (float)mu;
        stat[TT + threadIdx.x] = (float)(1.0 / sqrt(s2 / C + 1e-5));
    }
    __syncthreads();
    if (t0 + col >= T) return;
    const float muf = stat[col], r = stat[TT + col];
    float* yp = y + (int64_t)b * C * T + t0 + col;
    for (int c = r0; c < C; c += RP) yp[(int64_t)c * T] = ((tile[c * TT + col] - muf) * r) * gamma[c] + beta[c];
}

void launch_layernorm_ct(const float* x, const float* gamma, const float* beta, float* y, int B, int C, int64_t T, hipStream_t st,
                         Profiler* prof) {
    const int64_t n = (int64_t)B * T;
    ProfScope ps(prof, st, NC_KC_NORM, 8.0 * C * (double)n, 8.0 * C * (double)n);
    static const int tile_env = (int)env_int("NC_LN_TILE", -1);   // 0 = the per-column kernel, 8 / 16 = tile width
    const int tt = tile_env >= 0 ? tile_env : ((size_t)C * 16 * 4 <= 48 * 1024 ? 16 : 8);
    if ((tt == 8 || tt == 16) && (size_t)C * tt * 4 + 2 * tt * 4 <= 96 * 1024) {
        const size_t lds = (size_t)C * tt * 4 + 2 * tt * 4;
        const dim3 grid((unsigned)((T + tt - 1) / tt), (unsigned)B);
        if (tt == 16) {
            ensure_dynamic_lds((const void*)layernorm_tile_kernel<16>, lds);
            hipLaunchKernelGGL(layernorm_tile_kernel<16>, grid, dim3(256), lds, st, x, gamma, beta, y, C, T);
        } else {
            ensure_dynamic_lds((const void*)layernorm_tile_kernel<8>, lds);
            hipLaunchKernelGGL(layernorm_tile_kernel<8>, grid, dim3(256), lds, st, x, gamma, beta, y, C, T);
        }
        NC_HIP(hipGetLastError());
        return;
    }
    hipLaunchKernelGGL(layernorm_ct_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, x, gamma, beta, y, B, C, T);
    NC_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------- windowed rotary attention
// qkv [B,3C,T] (channel = part*C + head*64 + d) -> out [B,C,T]; one block per (window, head, clip), thread i = query i.
// cos/sin tables [W][64] come from the host (binary64 libm, rounded once).
constexpr int ATT_D = 64, ATT_W = 32;
__global__ __launch_bounds__(64) void local_attn_kernel(const float* __restrict__ qkv, const float* __restrict__ cs,
                                                        const float* __restrict__ sn, float* __restrict__ out, int C, int64_t T,
                                                        int W) {
    __shared__ float ks[ATT_W][ATT_D + 1], vs[ATT_W][ATT_D + 1];
    const int wdx = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int i = threadIdx.x;
    const float* base = qkv + (int64_t)b * 3 * C * T;
    const int64_t t = (int64_t)wdx * W + i;
    float q[ATT_D];
    if (i < W) {
        // 16 feature rows at a time: the 5 x 16 strided reads of a batch (q, k, their rotated partners, v) are all in flight before
        // the batch's arithmetic and LDS stores (one memory round trip per batch instead of one per feature)
#pragma unroll
        for (int d0 = 0; d0 < ATT_D; d0 += 16) {
            float qv[16], qr[16], kv[16], kr[16], vv[16], cc[16], ss[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int d = d0 + u, dr = d < 32 ? d + 32 : d - 32;
                qv[u] = base[(int64_t)(h * 64 + d) * T + t];
                qr[u] = base[(int64_t)(h * 64 + dr) * T + t];
                kv[u] = base[(int64_t)(C + h * 64 + d) * T + t];
                kr[u] = base[(int64_t)(C + h * 64 + dr) * T + t];
                vv[u] = base[(int64_t)(2 * C + h * 64 + d) * T + t];
                cc[u] = cs[i * 64 + d];
                ss[u] = sn[i * 64 + d];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int d = d0 + u;
                q[d] = (qv[u] * cc[u]) + ((d < 32 ? -qr[u] : qr[u]) * ss[u]);
                ks[i][d] = (kv[u] * cc[u]) + ((d < 32 ? -kr[u] : kr[u]) * ss[u]);
                vs[i][d] = vv[u];
            }
        }
    }
    __syncthreads();
    if (i >= W) return;
    float s[ATT_W];
    float mx = -__builtin_inff();
#pragma unroll
    for (int j = 0; j < ATT_W; ++j) {
        if (j < W) {
            float a = 0.0f;
#pragma unroll
            for (int d = 0; d < ATT_D; ++d) a = nc_fma(q[d], ks[j][d], a);
            s[j] = a * 0.125f;
            if (s[j] > mx) mx = s[j];
        }
    }
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < ATT_W; ++j)
        if (j < W) {
            s[j] = nc_expf(s[j] - mx);
            sum = sum + s[j];
        }
#pragma unroll
    for (int j = 0; j < ATT_W; ++j)
        if (j < W) s[j] = s[j] / sum;
    float* op = out + ((int64_t)b * C + h * 64) * T + t;
#pragma unroll 2
    for (int d0 = 0; d0 < ATT_D; d0 += 8) {
        float a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a[u] = 0.0f;
#pragma unroll
            for (int j = 0; j < ATT_W; ++j)
                if (j < W) a[u] = nc_fma(s[j], vs[j][d0 + u], a[u]);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) op[(int64_t)(d0 + u) * T] = a[u];
    }
}
// The same arithmetic on the matrix cores for the window the presets use (W = 32, head dim 64; round 4).  The kernel above keeps one
// query per lane and reads K / V from LDS one word per fma: 4096 LDS round trips per lane, 256 VGPRs, one wave per SIMD -- 361 us per
// launch on a 28 MB tensor.  Here one wavefront owns a (window, head, clip): q and k arrive from global memory ALREADY in the operand
// layout of v_mfma_f32_32x32x2_f32 (lane l: position l & 31, feature 2s + (l >> 5) at step s -- two 128-byte row segments per load), the
// rotation partner of feature d is the same lane's register of step s ^ 16, and
//   scores   S[i][j] = chain_d(q'[i][d], k'[j][d])      32 instructions, d ascending = the canonical fma chain
//   softmax  through LDS in the canonical order (row max, canonical exp, SEQUENTIAL sum over j, one division per element)
//   output   O[i][d] = chain_j(p[i][j], v[j][d])         2 x 16 instructions, j ascending
// are the identical operation sequences (bit-exact: tests/test_snac_gpu.py 44 kHz + LocalMHA vs the oracle).
typedef float attn_f32x16 __attribute__((ext_vector_type(16)));
typedef float attn_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void local_attn_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ cs,
                                                             const float* __restrict__ sn, float* __restrict__ out, int C, int64_t T) {
    constexpr int W = 32, D = 64, LD = 33;
    __shared__ float Ss[W * LD];      // scores -> exponentials -> probabilities, [query][key]
    __shared__ float Vs[D * LD];      // [feature][key]
    const int wdx = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int l = threadIdx.x, i = l & 31, hi = l >> 5;
    const float* base = qkv + (int64_t)b * 3 * C * T + (int64_t)wdx * W + i;
    const float* qp = base + (int64_t)(h * D + hi) * T;
    const float* kp = base + (int64_t)(C + h * D + hi) * T;
    const float* vp = base + (int64_t)(2 * C + h * D + hi) * T;
    float qv[32], kv[32];
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        qv[s] = qp[(int64_t)(2 * s) * T];
        kv[s] = kp[(int64_t)(2 * s) * T];
    }
#pragma unroll
    for (int s = 0; s < 32; ++s) Vs[(2 * s + hi) * LD + i] = vp[(int64_t)(2 * s) * T];
    attn_f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        const int d = 2 * s + hi;                                  // s < 16 <=> d < 32; the partner feature d +- 32 is step s ^ 16
        const float cc = cs[i * D + d], ss = sn[i * D + d];
        const float qr = s < 16 ? -qv[s ^ 16] : qv[s ^ 16], kr = s < 16 ? -kv[s ^ 16] : kv[s ^ 16];
        const float qq = (qv[s] * cc) + (qr * ss);                 // RotaryEmbedding.cs:46-68
        const float kk = (kv[s] * cc) + (kr * ss);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qq, kk, acc, 0, 0, 0);
    }
    // accumulator: key j = i (column), queries (v & 3) + 8 * (v >> 2) + 4 * hi
#pragma unroll
    for (int v = 0; v < 16; ++v) Ss[((v & 3) + 8 * (v >> 2) + 4 * hi) * LD + i] = acc[v] * 0.125f;
    // row i: lane (i, hi) owns keys 16 hi .. 16 hi + 15
    float* row = Ss + i * LD + 16 * hi;
    float e[16];
    float mx = -__builtin_inff();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        e[j] = row[j];
        if (e[j] > mx) mx = e[j];
    }
    {
        const float other = __shfl_xor(mx, 32);
        if (other > mx) mx = other;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        e[j] = nc_expf(e[j] - mx);
        row[j] = e[j];
    }
    float sum = 0.0f;                                              // the canonical SEQUENTIAL sum over the 32 keys (both halves compute it)
    {
        const float* r0 = Ss + i * LD;
#pragma unroll
        for (int j = 0; j < W; ++j) sum = sum + r0[j];
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) row[j] = e[j] / sum;
    attn_f32x16 o0, o1;
#pragma unroll
    for (int v = 0; v < 16; ++v) { o0[v] = 0.0f; o1[v] = 0.0f; }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const float p = Ss[i * LD + 2 * s + hi];                   // A[query i][key 2s + hi]
        o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(p, Vs[i * LD + 2 * s + hi], o0, 0, 0, 0);          // B[key][feature i]
        o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(p, Vs[(32 + i) * LD + 2 * s + hi], o1, 0, 0, 0);   // features 32 + i
    }
    float* op = out + ((int64_t)b * C + h * D + i) * T + (int64_t)wdx * W + 4 * hi;   // feature i (column), queries 8 g + 4 hi + 0..3
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const attn_f32x4 a0 = {o0[4 * g], o0[4 * g + 1], o0[4 * g + 2], o0[4 * g + 3]};
        const attn_f32x4 a1 = {o1[4 * g], o1[4 * g + 1], o1[4 * g + 2], o1[4 * g + 3]};
        *reinterpret_cast<attn_f32x4*>(op + 8 * g) = a0;
        *reinterpret_cast<attn_f32x4*>(op + (int64_t)32 * T + 8 * g) = a1;
    }
}

void launch_local_attn(const float* qkv, const float* cs, const float* sn, float* out, int B, int C, int64_t T, int W,
                       hipStream_t st, Profiler* prof) {
    if (W > ATT_W || W <= 0 || T % W != 0 || C % 64 != 0) fail(NC_EUNSUPPORTED, "local attention: window %d / dim %d not supported", W, C);
    ProfScope ps(prof, st, NC_KC_ATTN, 4.0 * W * C * (double)T * B, 16.0 * C * (double)T * B);
    static const bool no_mfma = env_flag("NC_ATTN_NO_MFMA");
    if (W == 32 && !no_mfma && T % 4 == 0) {
        hipLaunchKernelGGL(local_attn_mfma_kernel, dim3((unsigned)(T / W), (unsigned)(C / 64), (unsigned)B), dim3(64), 0, st, qkv, cs, sn, out, C, T);
        NC_HIP(hipGetLastError());
        return;
    }
    hipLaunchKernelGGL(local_attn_kernel, dim3((unsigned)(T / W), (unsigned)(C / 64), (unsigned)B), dim3(64), 0, st, qkv, cs, sn, out,
                       C, T, W);
    NC_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------- N(0,1) source (NoiseBlock)
// The reference draws torch.randn at inference (NoiseBlock.cs:41): not reproducible by construction.  When the caller does not
// inject the noise we draw it from a counter-based generator (SplitMix64 -> Box-Muller), deterministic in (seed, element index).
__device__ inline uint64_t nc_splitmix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ void randn_kernel(float* __restrict__ out, int64_t n, uint64_t seed) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t r = nc_splitmix(seed * 0xD1342543DE82EF95ull + (uint64_t)i);
    const float u1 = ((float)((r >> 40) + 1)) * (1.0f / 16777217.0f);  // (0,1]
    const float u2 = ((float)((r >> 16) & 0xFFFFFF)) * (1.0f / 16777216.0f);
    out[i] = sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}
void launch_randn(float* out, int64_t n, uint64_t seed, hipStream_t st) {
    hipLaunchKernelGGL(randn_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out, n, seed);
    NC_HIP(hipGetLastError());
}

}  // namespace nc
