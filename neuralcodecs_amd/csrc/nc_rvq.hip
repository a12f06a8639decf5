// Residual-VQ codebook kernels: nearest-neighbour lookup (argmin) and code gather.
//
// Reference: Modules/DAC/VectorQuantizer.cs:99-125 (DecodeLatents: dist = |e|^2 + |c|^2 - 2 e.c on
// UN-normalised vectors, deviation D1; argmin(1) = lowest index on ties) and :135-142 (DecodeCode).
// Same code path serves SNAC (Modules/SNAC/VectorQuantizer.cs:115-141, 4096 codes).
//
// Layout: the whole codebook lives in LDS transposed ([D][N], 32 KB for DAC, 128 KB for SNAC) so the
// 64 lanes of a wavefront read 64 consecutive codes of one dimension per ds_read_b32 (conflict-free);
// each lane scans N/64 codes, then a wavefront DPP/shuffle min-reduction over (distance, index) picks
// the winner with the first-index tie-break.  HBM traffic per frame is D floats in, D floats + one
// int64 out: the kernel is bound by LDS/VALU rate, not by HBM.
#include <cstdlib>

#include "nc_math.h"
#include "nc_model.h"

namespace nc {

constexpr int VQ_MAX_D = 16;
constexpr int VQ_FRAMES_PER_WAVE = 4;   // 16 frames per workgroup: enough workgroups to spread a ~3 k-frame launch over the chip

// DD > 0: the codebook dimension as a compile-time constant (DAC / SNAC: 8) -- with a run-time D every tap of the distance chain was a
// predicated branch (16 per code: 112 us for ONE workgroup of SNAC's 4096-entry codebook; 75-110 us per launch at any size).  FPW: frames
// per wavefront (4: 16 frames per workgroup, enough workgroups for a ~3 k-frame launch; 1 for launches of a few dozen frames).
template <int DD, int FPW>
__global__ __launch_bounds__(256) void vq_argmin_kernel(const float* __restrict__ cbT, const float* __restrict__ c2,
                                                        const float* __restrict__ cb_rm, int N, int Drt, const float* z_e,
                                                        int64_t ze_bstride, int B, int64_t T, int64_t* codes,
                                                        int64_t codes_bstride, float* st) {
    constexpr int MAXD = DD > 0 ? DD : VQ_MAX_D;
    const int D = DD > 0 ? DD : Drt;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_cb = smem;          // [D][N]
    float* s_c2 = smem + D * N;  // [N]
    const int tid = threadIdx.x;
    // codebook -> LDS, 16-byte words, 8 reads per thread in flight before their LDS stores (a read/store pair per iteration would
    // cost one global round trip per word); cbT and c2 are contiguous float arrays of D*N and N entries, N a multiple of 4
    {
        typedef float vq_f32x4 __attribute__((ext_vector_type(4)));
        const vq_f32x4* src = reinterpret_cast<const vq_f32x4*>(cbT);
        vq_f32x4* dst = reinterpret_cast<vq_f32x4*>(s_cb);
        const int nv = D * N / 4;
        for (int i0 = 0; i0 < nv; i0 += 8 * 256) {
            vq_f32x4 r[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) r[u] = src[min(i0 + tid + 256 * u, nv - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + tid + 256 * u < nv) dst[i0 + tid + 256 * u] = r[u];
        }
        for (int i = tid; i < N; i += 256) s_c2[i] = c2[i];
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int64_t total = (int64_t)B * T;
    const int64_t f0 = ((int64_t)blockIdx.x * 4 + wave) * FPW;
    // all frames of the wave are fetched before any is searched: one global round trip instead of one per frame
    float e[FPW][MAXD];
#pragma unroll
    for (int fi = 0; fi < FPW; ++fi) {
        const int64_t f = min(f0 + fi, total - 1);
        const int64_t b = f / T, t = f - b * T;
        const float* zp = z_e + b * ze_bstride + t;
#pragma unroll
        for (int d = 0; d < MAXD; ++d) e[fi][d] = d < D ? zp[(int64_t)d * T] : 0.0f;
    }
    int win[FPW];
#pragma unroll
    for (int fi = 0; fi < FPW; ++fi) {
        float e2 = 0.0f;
#pragma unroll
        for (int d = 0; d < MAXD; ++d)
            if (d < D) e2 = nc_fma(e[fi][d], e[fi][d], e2);
        float best = __builtin_inff();
        int bi = 0x7fffffff;
#pragma unroll 4
        for (int n = lane; n < N; n += 64) {
            float cr = 0.0f;
#pragma unroll
            for (int d = 0; d < MAXD; ++d)
                if (d < D) cr = nc_fma(e[fi][d], s_cb[d * N + n], cr);
            const float dist = (e2 + s_c2[n]) - 2.0f * cr;
            if (nc_argmin_scan(dist, best)) {
                best = dist;
                bi = n;
            }
        }
        // wavefront min-reduction on (dist, index), lowest index wins ties
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float od = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (nc_argmin_before(od, oi, best, bi)) {
                best = od;
                bi = oi;
            }
        }
        if (bi == 0x7fffffff) bi = 0;  // a row of +inf only: every distance equal, ATen returns index 0
        win[fi] = bi;
    }
    // epilogue: codes and the straight-through values, restated literally (VectorQuantizer.cs:81): e + (q - e)
#pragma unroll
    for (int fi = 0; fi < FPW; ++fi) {
        const int64_t f = f0 + fi;
        if (f >= total) break;
        const int64_t b = f / T, t = f - b * T;
        if (lane == 0) codes[b * codes_bstride + t] = (int64_t)win[fi];
        if (lane < D) {
            const float q = cb_rm[(int64_t)win[fi] * D + lane];
            float ev = 0.0f;
#pragma unroll
            for (int d = 0; d < MAXD; ++d)
                if (d == lane) ev = e[fi][d];
            st[(b * D + lane) * T + t] = ev + (q - ev);
        }
    }
}

__global__ __launch_bounds__(256) void vq_gather_kernel(const float* __restrict__ cb_rm, int N, int D, const int64_t* codes,
                                                        int64_t codes_bstride, int B, int64_t T, float* out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)B * D * T;
    if (i >= total) return;
    const int64_t t = i % T;
    const int d = (int)((i / T) % D);
    const int64_t b = i / (T * D);
    int64_t c = codes[b * codes_bstride + t];
    if (c < 0) c = 0;
    if (c >= N) c = N - 1;
    out[i] = cb_rm[c * D + d];
}

void Codebook::build(const float* h, int N_, int D_) {
    N = N_;
    D = D_;
    std::vector<float> t((size_t)N * D), n2(N);
    for (int n = 0; n < N; ++n) {
        float a = 0.0f;
        for (int d = 0; d < D; ++d) {
            t[(size_t)d * N + n] = h[(size_t)n * D + d];
            a = __builtin_fmaf(h[(size_t)n * D + d], h[(size_t)n * D + d], a);
        }
        n2[n] = a;
    }
    cbT.reserve(t.size() * 4);
    cb.reserve(t.size() * 4);
    c2.reserve(n2.size() * 4);
    NC_HIP(hipMemcpy(cbT.p, t.data(), t.size() * 4, hipMemcpyHostToDevice));
    NC_HIP(hipMemcpy(cb.p, h, t.size() * 4, hipMemcpyHostToDevice));
    NC_HIP(hipMemcpy(c2.p, n2.data(), n2.size() * 4, hipMemcpyHostToDevice));
}


void launch_vq_argmin(const Codebook& cb, const float* z_e, int64_t ze_bstride, int B, int64_t T, int64_t* codes,
                      int64_t codes_bstride, float* st, hipStream_t s, Profiler* prof) {
    if (cb.D > VQ_MAX_D) fail(NC_EUNSUPPORTED, "codebook_dim %d > %d", cb.D, VQ_MAX_D);
    if ((cb.D * cb.N) % 4 != 0) fail(NC_EUNSUPPORTED, "codebook of %d x %d entries is not a whole number of 16-byte words", cb.N, cb.D);
    const size_t lds = sizeof(float) * ((size_t)cb.D * cb.N + cb.N);
    if (lds > 160 * 1024) fail(NC_EUNSUPPORTED, "codebook of %d x %d does not fit LDS", cb.N, cb.D);
    const int64_t total = (int64_t)B * T;
    const int fpw = total <= 256 ? 1 : VQ_FRAMES_PER_WAVE;   // a few dozen frames: one per wavefront, so that they spread over more CUs
    const int64_t per_block = 4 * fpw;
    const int64_t grid = (total + per_block - 1) / per_block;
    if (prof && prof->on) prof->begin(s, NC_KC_RVQ, 3.0 * 2.0 * cb.D * cb.N * (double)total, 4.0 * total * (2.0 * cb.D + 2));
    auto launch = [&](auto kern) {
        ensure_dynamic_lds((const void*)kern, 160 * 1024);
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, s, cb.cbT.as<float>(), cb.c2.as<float>(), cb.cb.as<float>(), cb.N, cb.D, z_e,
                           ze_bstride, B, T, codes, codes_bstride, st);
    };
    if (cb.D == 8) {
        if (fpw == 1) launch(vq_argmin_kernel<8, 1>);
        else launch(vq_argmin_kernel<8, VQ_FRAMES_PER_WAVE>);
    } else {
        if (fpw == 1) launch(vq_argmin_kernel<0, 1>);
        else launch(vq_argmin_kernel<0, VQ_FRAMES_PER_WAVE>);
    }
    NC_HIP(hipGetLastError());
    if (prof && prof->on) prof->end(s);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Stage-fused DAC quantizer: in_proj -> nearest code -> out_proj -> (zq += q, residual -= q) for ALL n_q stages in one launch
// (ResidualVectorQuantizer.cs:54-103 over VectorQuantizer.cs:67-125).  Every frame is independent, so a workgroup owns 16 frames and
// walks the stages with the residual block [latent][16] in LDS and its share of zq (4 channels x 16 frames per thread) in registers;
// the stage's codebook ([D][N] + norms) and in_proj weights are staged into LDS once per stage.  Same arithmetic, operation for
// operation, as the stage-by-stage launches (skinny_proj_kernel / vq_argmin_kernel / the 1x1 template with the RVQ epilogue):
//   z_e[d]  = (fma chain over c ascending of W_in[d][c] * r[c], from +0) + b_in[d]
//   idx     = argmin_n (|z_e|^2 + |c_n|^2) - 2 (z_e . c_n), chains over d ascending, lowest index on ties
//   st[d]   = z_e[d] + (c_idx[d] - z_e[d])
//   q[c]    = (fma chain over d ascending of W_out[c][d] * st[d], from +0) + b_out[c];   zq[c] = zq[c] + q[c];   r[c] = r[c] - q[c]
// 36 launches (4 per stage) and the round trips of residual / zq through HBM between them become one launch.
constexpr int RF = 16;     // frames per workgroup
constexpr int RP = RF + 1; // row pitch of the residual block in LDS: the update phase walks the channels across the lanes -- at pitch 16 every
                           // lane of a wave met one of TWO banks (32-way conflicts on its 128 reads and writes per thread and stage)
constexpr int RD = 8;      // codebook dimension of this instantiation
typedef float rvq_f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const float* rvq_gp;
typedef __attribute__((address_space(1))) const rvq_f32x4* rvq_gp4;
struct RvqFusedArgs {
    const RvqStage* stages;
    const float* residual;
    float* zq;
    float* latents;
    int64_t* codes;
    int n_q, L, N, B;
    int64_t T;
};
__global__ __launch_bounds__(256) void dac_rvq_fused_kernel(const RvqFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int L = a.L, N = a.N;
    float* rs = sm;                       // [L][RP] residual block (RF frames + one pad word per channel row)
    float* s_cb = rs + L * RP;            // [RD][N]
    float* s_c2 = s_cb + RD * N;          // [N]
    float* s_w = s_c2 + N;                // [L][RD] in_proj weight (transposed)
    float* s_ze = s_w + L * RD;           // [RF][RD]
    float* s_st = s_ze + RF * RD;         // [RF][RD]
    int* s_bt = reinterpret_cast<int*>(s_st + RF * RD);   // [RF][2]: clip, frame within the clip (clip = -1: past the end)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t T = a.T, total = (int64_t)a.B * T, f0 = (int64_t)blockIdx.x * RF;
    if (tid < RF) {
        const int64_t fr = f0 + tid;
        const int64_t b = fr < total ? fr / T : -1;
        s_bt[2 * tid] = (int)b;
        s_bt[2 * tid + 1] = (int)(fr < total ? fr - b * T : 0);
    }
    __syncthreads();
    for (int i = tid; i < L * RF; i += 256) {
        const int c = i / RF, f = i - c * RF;
        const int b = s_bt[2 * f], t = s_bt[2 * f + 1];
        rs[c * RP + f] = b >= 0 ? a.residual[((int64_t)b * L + c) * T + t] : 0.0f;
    }
    const int NJ = L / 256;               // channels per thread in the out_proj / update phase (host: L % 256 == 0, NJ <= 4)
    float zqr[4][RF];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int f = 0; f < RF; ++f) zqr[j][f] = 0.0f;
    for (int q = 0; q < a.n_q; ++q) {
        const RvqStage sg = a.stages[q];
        __syncthreads();                  // the previous stage has finished with the codebook / weight images and updated rs
        {
            const rvq_gp4 c4 = (rvq_gp4)sg.cbT, w4 = (rvq_gp4)sg.w_inT, n4 = (rvq_gp4)sg.c2;
            // every read of the stage's images (68 KB) is issued before the first LDS store: three copy loops in sequence were three L2 round
            // trips.  (One stage AHEAD in registers -- read under the previous stage, stored behind its last barrier -- measured slower: 198 -> 240 us.)
            constexpr int NCW = RD * 1024 / 4 / 256, NWW = 1024 * RD / 4 / 256;   // words per thread at the largest N / L the launcher admits
            rvq_f32x4 cw[NCW], ww[NWW], nw;
            const int ncb = RD * N / 4, nww = L * RD / 4;
#pragma unroll
            for (int u = 0; u < NCW; ++u) cw[u] = c4[min(tid + 256 * u, ncb - 1)];
#pragma unroll
            for (int u = 0; u < NWW; ++u) ww[u] = w4[min(tid + 256 * u, nww - 1)];
            nw = n4[min(tid, N / 4 - 1)];
            rvq_f32x4* d4 = reinterpret_cast<rvq_f32x4*>(s_cb);
#pragma unroll
            for (int u = 0; u < NCW; ++u)
                if (tid + 256 * u < ncb) d4[tid + 256 * u] = cw[u];
            d4 = reinterpret_cast<rvq_f32x4*>(s_w);
#pragma unroll
            for (int u = 0; u < NWW; ++u)
                if (tid + 256 * u < nww) d4[tid + 256 * u] = ww[u];
            if (tid < N / 4) reinterpret_cast<rvq_f32x4*>(s_c2)[tid] = nw;
        }
        // out_proj rows of this thread's channels: read now, used after the search (their latency hides under the in_proj chains)
        rvq_f32x4 wo[4][2];
        float bo4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = min(tid + 256 * j, L - 1);
            wo[j][0] = ((rvq_gp4)sg.w_out)[2 * c];
            wo[j][1] = ((rvq_gp4)sg.w_out)[2 * c + 1];
            bo4[j] = ((rvq_gp)sg.b_out)[c];
        }
        __syncthreads();
        if (tid < RF * RD) {              // in_proj: thread = (frame, d), one chain over the latent channels
            const int f = tid / RD, d = tid - f * RD;
            const float b_in = ((rvq_gp)sg.b_in)[d];
            float acc = 0.0f;
            // ONE chain over c ascending (the canonical order); the LDS operands of 16 steps are read ahead of their 16 dependent fmas, the
            // next 16 while those issue (left to `#pragma unroll 16` every fma waited for its own pair of reads)
            float wv[2][16], rv[2][16];
#pragma unroll
            for (int u = 0; u < 16; ++u) { wv[0][u] = s_w[u * RD + d]; rv[0][u] = rs[u * RP + f]; }
#pragma unroll 1
            for (int c0 = 0; c0 < L; c0 += 32) {
#pragma unroll
                for (int u = 0; u < 16; ++u) { wv[1][u] = s_w[(c0 + 16 + u) * RD + d]; rv[1][u] = rs[(c0 + 16 + u) * RP + f]; }
#pragma unroll
                for (int u = 0; u < 16; ++u) acc = nc_fma(wv[0][u], rv[0][u], acc);
                const int cn = min(c0 + 32, L - 16);
#pragma unroll
                for (int u = 0; u < 16; ++u) { wv[0][u] = s_w[(cn + u) * RD + d]; rv[0][u] = rs[(cn + u) * RP + f]; }
#pragma unroll
                for (int u = 0; u < 16; ++u) acc = nc_fma(wv[1][u], rv[1][u], acc);
            }
            const float ze = acc + b_in;
            s_ze[tid] = ze;
            const int b = s_bt[2 * f], t = s_bt[2 * f + 1];
            if (b >= 0) a.latents[((int64_t)b * a.n_q * RD + (int64_t)q * RD + d) * T + t] = ze;
        }
        __syncthreads();
        {   // nearest code: wave = 4 frames, lane scans N/64 codes (vq_argmin_kernel's arithmetic).  The four frames of a wave go through
            // ONE pass over the codebook (a code's 8 LDS words feed four chains; frame after frame the loop was four dependent scans)
            constexpr int FW = RF / 4;
            float e[FW][RD], e2[FW], best[FW];
            int bi[FW];
#pragma unroll
            for (int fi = 0; fi < FW; ++fi) {
                const int f = wave * FW + fi;
#pragma unroll
                for (int d = 0; d < RD; ++d) e[fi][d] = s_ze[f * RD + d];
                e2[fi] = 0.0f;
#pragma unroll
                for (int d = 0; d < RD; ++d) e2[fi] = nc_fma(e[fi][d], e[fi][d], e2[fi]);
                best[fi] = __builtin_inff();
                bi[fi] = 0x7fffffff;
            }
            for (int n = lane; n < N; n += 64) {
                float cv[RD];
#pragma unroll
                for (int d = 0; d < RD; ++d) cv[d] = s_cb[d * N + n];
                const float cn = s_c2[n];
#pragma unroll
                for (int fi = 0; fi < FW; ++fi) {
                    float cr = 0.0f;
#pragma unroll
                    for (int d = 0; d < RD; ++d) cr = nc_fma(e[fi][d], cv[d], cr);
                    const float dist = (e2[fi] + cn) - 2.0f * cr;
                    if (nc_argmin_scan(dist, best[fi])) { best[fi] = dist; bi[fi] = n; }
                }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
                for (int fi = 0; fi < FW; ++fi) {
                    const float od = __shfl_xor(best[fi], off, 64);
                    const int oi = __shfl_xor(bi[fi], off, 64);
                    if (nc_argmin_before(od, oi, best[fi], bi[fi])) { best[fi] = od; bi[fi] = oi; }
                }
            }
#pragma unroll
            for (int fi = 0; fi < FW; ++fi) {
                const int f = wave * FW + fi;
                if (bi[fi] == 0x7fffffff) bi[fi] = 0;
                const int b = s_bt[2 * f], t = s_bt[2 * f + 1];
                if (lane == 0 && b >= 0) a.codes[((int64_t)b * a.n_q + q) * T + t] = (int64_t)bi[fi];
                if (lane < RD) {
                    const float qv = s_cb[lane * N + bi[fi]];   // (the stage's codebook is in LDS, transposed: the same words as sg.cb[bi][lane]
                                                                //  without an L2 round trip per frame)
                    const float ev = s_ze[f * RD + lane];
                    s_st[f * RD + lane] = ev + (qv - ev);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {           // out_proj + zq += q ; residual -= q: thread = 1 channel per pass x all frames
            if (j < NJ) {
                const int c = tid + 256 * j;
                const float wv[RD] = {wo[j][0][0], wo[j][0][1], wo[j][0][2], wo[j][0][3], wo[j][1][0], wo[j][1][1], wo[j][1][2], wo[j][1][3]};
#pragma unroll
                for (int f = 0; f < RF; ++f) {
                    float acc = 0.0f;
#pragma unroll
                    for (int d = 0; d < RD; ++d) acc = nc_fma(wv[d], s_st[f * RD + d], acc);
                    const float y = acc + bo4[j];
                    zqr[j][f] = zqr[j][f] + y;
                    rs[c * RP + f] = rs[c * RP + f] - y;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j < NJ) {
            const int c = tid + 256 * j;
#pragma unroll
            for (int f = 0; f < RF; ++f) {
                const int b = s_bt[2 * f], t = s_bt[2 * f + 1];
                if (b >= 0) a.zq[((int64_t)b * L + c) * T + t] = zqr[j][f];
            }
        }
    }
}

bool launch_dac_rvq_fused(const RvqStage* stages_dev, int n_q, int L, int D, int N, const float* residual, int B, int64_t T, int64_t* codes,
                          float* zq, float* latents, hipStream_t s, Profiler* prof) {
    static const bool off = env_present("NC_DAC_RVQ_STAGEWISE");
    if (off || D != RD || L % 256 != 0 || L > 1024 || N % 64 != 0 || N > 1024 || n_q <= 0) return false;
    const size_t lds = sizeof(float) * ((size_t)L * RP + (size_t)RD * N + N + (size_t)L * RD + 2 * RF * RD + 2 * RF);
    if (lds > 160 * 1024) return false;
    ensure_dynamic_lds((const void*)dac_rvq_fused_kernel, 160 * 1024);
    const int64_t total = (int64_t)B * T;
    RvqFusedArgs a{};
    a.stages = stages_dev; a.residual = residual; a.zq = zq; a.latents = latents; a.codes = codes;
    a.n_q = n_q; a.L = L; a.N = N; a.B = B; a.T = T;
    if (prof && prof->on)
        prof->begin(s, NC_KC_RVQ, (3.0 * 2.0 * D * N + 4.0 * D * L) * (double)total * n_q, 4.0 * total * (2.0 * L + (double)n_q * (D + 2)));
    hipLaunchKernelGGL(dac_rvq_fused_kernel, dim3((unsigned)((total + RF - 1) / RF)), dim3(256), lds, s, a);
    NC_HIP(hipGetLastError());
    if (prof && prof->on) prof->end(s);
    return true;
}

void launch_vq_gather(const Codebook& cb, const int64_t* codes, int64_t codes_bstride, int B, int64_t T, float* out, hipStream_t s,
                      Profiler* prof) {
    const int64_t total = (int64_t)B * cb.D * T;
    if (prof && prof->on) prof->begin(s, NC_KC_RVQ, 0.0, 4.0 * total + 8.0 * B * T);
    hipLaunchKernelGGL(vq_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, cb.cb.as<float>(), cb.N, cb.D,
                       codes, codes_bstride, B, T, out);
    NC_HIP(hipGetLastError());
    if (prof && prof->on) prof->end(s);
}

}  // namespace nc
