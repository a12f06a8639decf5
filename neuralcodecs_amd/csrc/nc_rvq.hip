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
#include "nc_math.h"
#include "nc_model.h"

namespace nc {

constexpr int VQ_MAX_D = 16;
constexpr int VQ_FRAMES_PER_WAVE = 4;   // 16 frames per workgroup: enough workgroups to spread a ~3 k-frame launch over the chip

__global__ __launch_bounds__(256) void vq_argmin_kernel(const float* __restrict__ cbT, const float* __restrict__ c2,
                                                        const float* __restrict__ cb_rm, int N, int D, const float* z_e,
                                                        int64_t ze_bstride, int B, int64_t T, int64_t* codes,
                                                        int64_t codes_bstride, float* st) {
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
    const int64_t f0 = ((int64_t)blockIdx.x * 4 + wave) * VQ_FRAMES_PER_WAVE;
    // all frames of the wave are fetched before any is searched: one global round trip instead of one per frame
    float e[VQ_FRAMES_PER_WAVE][VQ_MAX_D];
#pragma unroll
    for (int fi = 0; fi < VQ_FRAMES_PER_WAVE; ++fi) {
        const int64_t f = min(f0 + fi, total - 1);
        const int64_t b = f / T, t = f - b * T;
        const float* zp = z_e + b * ze_bstride + t;
#pragma unroll
        for (int d = 0; d < VQ_MAX_D; ++d) e[fi][d] = d < D ? zp[(int64_t)d * T] : 0.0f;
    }
    int win[VQ_FRAMES_PER_WAVE];
#pragma unroll
    for (int fi = 0; fi < VQ_FRAMES_PER_WAVE; ++fi) {
        float e2 = 0.0f;
#pragma unroll
        for (int d = 0; d < VQ_MAX_D; ++d)
            if (d < D) e2 = nc_fma(e[fi][d], e[fi][d], e2);
        float best = __builtin_inff();
        int bi = 0x7fffffff;
        for (int n = lane; n < N; n += 64) {
            float cr = 0.0f;
#pragma unroll
            for (int d = 0; d < VQ_MAX_D; ++d)
                if (d < D) cr = nc_fma(e[fi][d], s_cb[d * N + n], cr);
            const float dist = (e2 + s_c2[n]) - 2.0f * cr;
            if (dist < best) {
                best = dist;
                bi = n;
            }
        }
        // wavefront min-reduction on (dist, index), lowest index wins ties
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float od = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (od < best || (od == best && oi < bi)) {
                best = od;
                bi = oi;
            }
        }
        if (bi == 0x7fffffff) bi = 0;  // all-NaN row: ATen returns an index as well; pick 0
        win[fi] = bi;
    }
    // epilogue: codes and the straight-through values, restated literally (VectorQuantizer.cs:81): e + (q - e)
#pragma unroll
    for (int fi = 0; fi < VQ_FRAMES_PER_WAVE; ++fi) {
        const int64_t f = f0 + fi;
        if (f >= total) break;
        const int64_t b = f / T, t = f - b * T;
        if (lane == 0) codes[b * codes_bstride + t] = (int64_t)win[fi];
        if (lane < D) {
            const float q = cb_rm[(int64_t)win[fi] * D + lane];
            float ev = 0.0f;
#pragma unroll
            for (int d = 0; d < VQ_MAX_D; ++d)
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
    ensure_dynamic_lds((const void*)vq_argmin_kernel, 160 * 1024);
    const int64_t total = (int64_t)B * T;
    const int64_t per_block = 4 * VQ_FRAMES_PER_WAVE;
    const int64_t grid = (total + per_block - 1) / per_block;
    if (prof && prof->on) prof->begin(s, NC_KC_RVQ, 3.0 * 2.0 * cb.D * cb.N * (double)total, 4.0 * total * (2.0 * cb.D + 2));
    hipLaunchKernelGGL(vq_argmin_kernel, dim3((unsigned)grid), dim3(256), lds, s, cb.cbT.as<float>(), cb.c2.as<float>(),
                       cb.cb.as<float>(), cb.N, cb.D, z_e, ze_bstride, B, T, codes, codes_bstride, st);
    NC_HIP(hipGetLastError());
    if (prof && prof->on) prof->end(s);
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
