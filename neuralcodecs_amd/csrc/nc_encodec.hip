// Encodec on the engine: SEANet encoder / decoder, LSTM, Euclidean RVQ, RMS scale and overlap-add.
//
// Reference call stacks restated as kernel launches:
//   Encodec.Encode  Models/Encodec.cs:259-285 -> EncodeFrame :457-489 -> SEANetEncoder.cs:37-148 -> ResidualVectorQuantizer.cs:133-157
//   Encodec.Decode  Models/Encodec.cs:213-235 -> DecodeFrame :436-455 -> SEANetDecoder.cs:40-153 -> DSP.LinearOverlapAdd
// Every SConv1d (SConv1d.cs:144-173) is  [pad_act kernel: GroupNorm-apply of the producer + ELU + asymmetric reflect pad, incl. the
// small-input path D9]  ->  [implicit-GEMM conv kernel on the padded tensor]  ->  [GroupNorm statistics kernels]; the normalisation of
// a conv output is applied by its consumer, so each activation is written once raw and once padded.  Dense contractions (convs,
// LSTM input projections) run on the matrix-core conv template; the recurrent part of the LSTM is one launch per time step.
// Arithmetic is the canonical arithmetic of DESIGN.md (same sequences as oracle/c/nc_ref_encodec.c).
#include <cmath>
#include <cstdlib>
#include <mutex>

#include "nc_gn.h"
#include "nc_math.h"
#include "nc_model.h"
#include "nc_lstm.h"

namespace nc {

constexpr int GN_CHUNK = 256;    // RMS-scale chunks

// ---------------------------------------------------------------------------------------------- kernels
struct ActView {          // [B,C,L] view of a raw conv output with its pending GroupNorm (applied by the consumer)
    const float* p;
    int64_t rs, off;      // row stride (elements), offset of logical sample 0 in a row (conv-transpose trim)
    const float* stats;   // [B][2] = (mean, rstd); null: no normalisation
    const float* gamma;
    const float* beta;
};


__device__ __forceinline__ float act_value(const ActView& v, int64_t b, int c, int C, int64_t q) {
    float x = v.p[(b * C + c) * v.rs + v.off + q];
    if (v.stats) x = ((x - v.stats[2 * b]) * v.stats[2 * b + 1]) * v.gamma[c] + v.beta[c];
    return x;
}

// dst[b,c,j] = pad(elu?(GN(a) [+ GN(b2)]))[j],  j in [0,Lp): reflect over the zero-extended row of length Lz (SConv1d.cs:258-274).
// Grid = (row segments of 1024, channels, samples): no index division, the per-(b,c) operands are scalars, 4 elements per thread
// with all reads ahead of the stores.
__global__ __launch_bounds__(256) void pad_act_kernel(ActView a, ActView b2, int has_b, int elu, float* __restrict__ dst, int C, int64_t L, int64_t Lz,
                                                      int64_t left, int64_t Lp) {
    const int c = blockIdx.y;
    const int64_t b = blockIdx.z;
    const float* ra = a.p + (b * C + c) * a.rs + a.off;
    const float* rb = b2.p + (b * C + c) * b2.rs + b2.off;
    const bool na = a.stats != nullptr, nb = has_b && b2.stats != nullptr;
    const float mu_a = na ? a.stats[2 * b] : 0.0f, r_a = na ? a.stats[2 * b + 1] : 1.0f, g_a = na ? a.gamma[c] : 1.0f, be_a = na ? a.beta[c] : 0.0f;
    const float mu_b = nb ? b2.stats[2 * b] : 0.0f, r_b = nb ? b2.stats[2 * b + 1] : 1.0f, g_b = nb ? b2.gamma[c] : 1.0f, be_b = nb ? b2.beta[c] : 0.0f;
    float* out = dst + (b * C + c) * Lp;
    float va[4], vb[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t j = (int64_t)blockIdx.x * 1024 + u * 256 + threadIdx.x;
        int64_t q = j - left;
        if (q < 0) q = -q;
        if (q >= Lz) q = 2 * (Lz - 1) - q;
        ok[u] = j < Lp && q >= 0 && q < L;
        const int64_t qa = ok[u] ? q : 0;
        va[u] = ra[qa];
        vb[u] = has_b ? rb[qa] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t j = (int64_t)blockIdx.x * 1024 + u * 256 + threadIdx.x;
        float v = va[u];
        if (na) v = ((v - mu_a) * r_a) * g_a + be_a;
        if (has_b) {
            float w = vb[u];
            if (nb) w = ((w - mu_b) * r_b) * g_b + be_b;
            v = v + w;
        }
        if (elu) v = nc_eluf(v);
        if (j < Lp) out[j] = ok[u] ? v : 0.0f;
    }
}

// Stand-alone GroupNorm block sums in the canonical order of nc_gn.h, for the outputs whose producing kernel cannot emit them from
// its epilogue (the streaming thin-output head, per-phase transposed launches): one wavefront per 32x32 block of the (rows = c*sub +
// t % sub, columns = t / sub) view of x [B][C][T]; lane (h, c) adds its 16 rows of column c, then the butterfly.
constexpr int GN_CBW = 4;   // column blocks per wavefront: their 64 row reads are all in flight before the first sum (memory-level parallelism)
__global__ __launch_bounds__(256) void gn_block_kernel(const float* __restrict__ x, double* __restrict__ part, int64_t B, int C, int64_t T, int sub,
                                                       int nrb, int ncb, int64_t rs /* row pitch (elements) */) {
    const int lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31;
    const int ncg = (ncb + GN_CBW - 1) / GN_CBW;                      // groups of column blocks per row block
    const int64_t total = B * nrb * ncg;
    const int64_t gi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gi >= total) return;
    const int64_t b = gi / ((int64_t)nrb * ncg), rem = gi - b * nrb * ncg;
    const int rb = (int)(rem / ncg), cg = (int)(rem - (int64_t)rb * ncg);
    const float* xb = x + b * C * rs;
    float v[GN_CBW][16];
    unsigned okm[GN_CBW];
#pragma unroll
    for (int u = 0; u < GN_CBW; ++u) {
        const int64_t q = (int64_t)(cg * GN_CBW + u) * 32 + c;
        okm[u] = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int R = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int co = sub == 1 ? R : R / sub;
            const int64_t t = sub == 1 ? q : q * sub + (R - co * sub);
            const bool ok = co < C && t < T;
            v[u][r] = xb[(int64_t)min(co, C - 1) * rs + min(t, T - 1)];   // branch-free: clamped address, value masked in the sum
            if (ok) okm[u] |= 1u << r;
        }
    }
#pragma unroll
    for (int u = 0; u < GN_CBW; ++u) {
        const int cb = cg * GN_CBW + u;
        double s1, s2;
        nc_gn_slot_sums<false>(v[u], okm[u], s1, s2);
        nc_gn_butterfly(s1, s2);
        if (lane == 0 && cb < ncb) {
            const int64_t bi = (b * nrb + rb) * ncb + cb;
            part[2 * bi] = s1;
            part[2 * bi + 1] = s2;
        }
    }
}
// one wavefront per sample: the n block sums of the sample by 64 strided slots (slot i: idx = i, i+64, ... ascending) + butterfly
// -> (mean, rstd); count = C*T elements
__global__ __launch_bounds__(64) void gn_final_kernel(const double* __restrict__ part, float* __restrict__ stats, int64_t n, double count) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const double* p = part + 2 * (int64_t)b * n;
    double s1 = 0.0, s2 = 0.0;
    for (int64_t k0 = 0; k0 < n; k0 += 64 * 8) {   // 8 independent 16-byte reads in flight per lane
        double a[8], c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t k = k0 + lane + 64 * u;
            const double2 v = k < n ? *reinterpret_cast<const double2*>(p + 2 * k) : double2{0.0, 0.0};
            a[u] = v.x; c[u] = v.y;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { s1 += a[u]; s2 += c[u]; }
    }
    nc_gn_butterfly(s1, s2);
    if (lane == 0) {
        const double mu = s1 / count;
        double var = s2 / count - mu * mu;
        if (var < 0.0) var = 0.0;
        stats[2 * b] = (float)mu;
        stats[2 * b + 1] = (float)(1.0 / sqrt(var + 1e-5));
    }
}

// RMS scale (Encodec.cs:469-480): chunk sums of fl32(mono^2), then scale = sqrtf((float)(S/L)) + 1e-8f
__global__ void rms_partial_kernel(const float* __restrict__ x, double* __restrict__ part, int B, int C, int64_t L, int nchunk) {
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= (int64_t)B * nchunk) return;
    const int64_t b = i / nchunk, ch = i - b * nchunk;
    const int64_t t0 = ch * GN_CHUNK, t1 = t0 + GN_CHUNK < L ? t0 + GN_CHUNK : L;
    double s = 0.0;
    for (int64_t t = t0; t < t1; ++t) {
        float a = x[(b * C) * L + t];
        for (int c = 1; c < C; ++c) a = a + x[(b * C + c) * L + t];
        const float m = a / (float)C;
        s += (double)(m * m);
    }
    part[i] = s;
}
__global__ void rms_final_kernel(const double* __restrict__ part, float* __restrict__ scale, int B, int64_t L, int nchunk) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    double s = 0.0;
    for (int k = 0; k < nchunk; ++k) s += part[(int64_t)b * nchunk + k];
    scale[b] = sqrtf((float)(s / (double)L)) + 1e-8f;
}
// The same scale in ONE launch (the default; NC_RMS_TWO_PASS=1 runs the two kernels above): a workgroup stages RMS_G chunks of a clip through
// LDS -- coalesced reads, fl32(mono^2) per sample computed in parallel -- and thread g adds chunk g's 256 values in ascending order in
// binary64 (the canonical chunk sum); the LAST workgroup of a clip to arrive (self-resetting counter, the hand-off of nc_gn.h: write-through
// partials, drain, barrier, one relaxed agent-scope fetch-add, agent-scope loads) adds the clip's chunk sums in ascending order and writes
// the scale.  Bit-identical to rms_partial_kernel + rms_final_kernel: 44 + 24 us -> one short launch on C3 (16 x 2 s).
constexpr int RMS_G = 16;
__global__ __launch_bounds__(256) void rms_scale_kernel(const float* __restrict__ x, double* __restrict__ part, unsigned* __restrict__ counter,
                                                        float* __restrict__ scale, int C, int64_t L, int nchunk, int nblk) {
    __shared__ float sq[RMS_G][GN_CHUNK + 1];
    __shared__ double fin[256];
    __shared__ int last;
    const int b = blockIdx.x / nblk, blk = blockIdx.x - b * nblk, tid = threadIdx.x;
    const int64_t t_base = (int64_t)blk * RMS_G * GN_CHUNK;
    const float* xb = x + (int64_t)b * C * L;
    const float invC = (float)C;
    for (int g0 = 0; g0 < RMS_G; g0 += 4) {                       // four chunk rows per pass: the reads of a pass are in flight together
        float a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t t = t_base + (int64_t)(g0 + u) * GN_CHUNK + tid;
            a[u] = t < L ? xb[t] : 0.0f;
        }
        for (int c = 1; c < C; ++c) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t t = t_base + (int64_t)(g0 + u) * GN_CHUNK + tid;
                const float v = t < L ? xb[(int64_t)c * L + t] : 0.0f;
                a[u] = a[u] + v;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float m = a[u] / invC;
            sq[g0 + u][tid] = m * m;
        }
    }
    __syncthreads();
    const int ch = blk * RMS_G + tid;
    if (tid < RMS_G && ch < nchunk) {
        const int64_t t0 = (int64_t)ch * GN_CHUNK;
        const int n = (int)((t0 + GN_CHUNK < L ? t0 + GN_CHUNK : L) - t0);
        double acc = 0.0;
        for (int i = 0; i < n; ++i) acc += (double)sq[tid][i];
        __hip_atomic_store(part + (int64_t)b * nchunk + ch, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) last = __hip_atomic_fetch_add(counter + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == (unsigned)nblk;
    __syncthreads();
    if (!last) return;
    double tot = 0.0;                                               // (ascending chunk order, one accumulator: rms_final_kernel's sum)
    for (int k0 = 0; k0 < nchunk; k0 += 256) {
        if (k0 + tid < nchunk) fin[tid] = __hip_atomic_load(part + (int64_t)b * nchunk + k0 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (tid == 0) {
            const int n = nchunk - k0 < 256 ? nchunk - k0 : 256;
            for (int i = 0; i < n; ++i) tot += fin[i];
        }
        __syncthreads();
    }
    if (tid == 0) {
        scale[b] = sqrtf((float)(tot / (double)L)) + 1e-8f;
        __hip_atomic_store(counter + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// y = x / scale[b]  (mode 0)   |   y = GN(x) * scale[b] (mode 1, scale nullable -> plain GN materialisation)
__global__ void scale_kernel(ActView a, const float* __restrict__ scale, int mode, float* __restrict__ y, int B, int C, int64_t L) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * C * L) return;
    const int64_t t = i % L, r = i / L;
    const int c = (int)(r % C);
    const int64_t b = r / C;
    float v = act_value(a, b, c, C, t);
    if (scale) v = mode == 0 ? v / scale[b] : v * scale[b];
    y[i] = v;
}

// One LSTM time step of one layer (SLSTM.cs:31,40-57; gate order i,f,g,o).  Block j = hidden unit, thread = clip.
//   gi   [B,4C,T]  input projection incl. b_ih (matrix-core 1x1 conv)         hprev/hnext, cst  [C][B] (unit-major)
//   out  [B,C,T]   h_t (+ skip[b,j,t] for the last layer: output.add(permuted), SLSTM.cs:50-53; elu_out: the ELU every consumer of an
//                  SLSTM applies first -- SEANetEncoder.cs / SEANetDecoder.cs: [.., SLSTM, ELU, conv] -- is applied here, once)
__global__ __launch_bounds__(64) void lstm_step_kernel(const float* __restrict__ gi, const float* __restrict__ whh,
                                                       const float* __restrict__ bhh, const float* __restrict__ hprev,
                                                       float* __restrict__ hnext, float* __restrict__ cst, const float* __restrict__ skip,
                                                       float* __restrict__ out, int B, int C, int64_t T, int64_t t, int elu_out) {
    extern __shared__ float wrow[];   // [4][C]
    const int j = blockIdx.x;
    for (int i = threadIdx.x; i < 4 * C; i += 64) wrow[i] = whh[(int64_t)((i / C) * C + j) * C + (i % C)];
    __syncthreads();
    const int b = blockIdx.y * 64 + threadIdx.x;
    if (b >= B) return;
    // recurrent contraction: four quarter chains combined as (q0 + q1) + (q2 + q3) (the canonical order: oracle slstm); one chain
    // when C is not a multiple of 4
    float r0, r1, r2, r3;
    {
        const int nq = (C % 4 == 0) ? 4 : 1, qlen = C / nq;
        float q0[4], q1[4], q2[4], q3[4];
        for (int s4 = 0; s4 < nq; ++s4) {
            float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, c3 = 0.0f;
            for (int k = s4 * qlen; k < (s4 + 1) * qlen; ++k) {
                const float h = hprev[(int64_t)k * B + b];
                c0 = nc_fma(wrow[k], h, c0);
                c1 = nc_fma(wrow[C + k], h, c1);
                c2 = nc_fma(wrow[2 * C + k], h, c2);
                c3 = nc_fma(wrow[3 * C + k], h, c3);
            }
            q0[s4] = c0; q1[s4] = c1; q2[s4] = c2; q3[s4] = c3;
        }
        if (nq == 4) {
            r0 = (q0[0] + q0[1]) + (q0[2] + q0[3]);
            r1 = (q1[0] + q1[1]) + (q1[2] + q1[3]);
            r2 = (q2[0] + q2[1]) + (q2[2] + q2[3]);
            r3 = (q3[0] + q3[1]) + (q3[2] + q3[3]);
        } else {
            r0 = q0[0]; r1 = q1[0]; r2 = q2[0]; r3 = q3[0];
        }
    }
    const float* g = gi + ((int64_t)b * 4 * C) * T + t;
    const float pi = g[(int64_t)j * T] + (r0 + bhh[j]);
    const float pf = g[(int64_t)(C + j) * T] + (r1 + bhh[C + j]);
    const float pg = g[(int64_t)(2 * C + j) * T] + (r2 + bhh[2 * C + j]);
    const float po = g[(int64_t)(3 * C + j) * T] + (r3 + bhh[3 * C + j]);
    const float ig = nc_sigmoidf(pi), fg = nc_sigmoidf(pf), gg = nc_tanhf(pg), og = nc_sigmoidf(po);
    const float cn = (fg * cst[(int64_t)j * B + b]) + (ig * gg);
    cst[(int64_t)j * B + b] = cn;
    const float h = og * nc_tanhf(cn);
    hnext[(int64_t)j * B + b] = h;
    const int64_t o = ((int64_t)b * C + j) * T + t;
    const float y = skip ? h + skip[o] : h;
    out[o] = elu_out ? nc_eluf(y) : y;
}

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* lstm_gptr;
typedef __attribute__((address_space(3))) void* lstm_lptr;

// Persistent LSTM layer: ALL T time steps of one layer in ONE launch (SLSTM.cs:31,40-57).
//   * W_hh stays in LDS for the whole sequence: a workgroup of 16 wavefronts owns 16 hidden units; wave (ub, q) holds the 16 gate
//     rows of 4 units (one 16-row matrix-core tile) for quarter q of the reduction: 128 KB per workgroup at C = 512 -- one workgroup
//     per CU, C/16 workgroups per 16-clip column tile, loaded once by LDS DMA.
//   * Per step the recurrent contraction runs as FOUR quarter chains (the canonical order of the oracle: (q0 + q1) + (q2 + q3)) walked
//     side by side by the four waves of a unit block: the dependent chain -- the critical path of the step -- is 32
//     v_mfma_f32_16x16x4_f32 instead of 128 (0.5 us instead of 2.1); the partial tiles meet in LDS and the q = 0 wave applies the gates
//     with the cell state in its registers.
//   * h_t of a column tile is exchanged between its C/16 workgroups through global memory with the placement-independent
//     release/acquire protocol of cdna_hip_programming.md G16 (recipe R1): write-through (sc1) payload stores -> vmcnt(0) ->
//     workgroup barrier -> one relaxed agent-scope flag store per workgroup; consumers poll the flags of their tile (one lane per
//     producer, relaxed), a workgroup barrier, and then read the h tile with relaxed AGENT-SCOPE (sc1) loads, which bypass the CU's L1
//     and observe the producers' write-through stores directly -- the shipped build has NO acquire fence (the "sc1 stores and sc1
//     loads both sides" form of MI355X_MICROARCH.md; ordering between the flag and the payload comes from the producer's vmcnt(0)
//     drain before its flag store and the consumer's barrier before its loads).  NC_SYNC_ACQUIRE=1 adds the acquire fence behind the poll
//     at run time (an operational fallback, exercised by tests/test_children_gpu.py); -DNC_LSTM_FENCE is the compile-time fence +
//     plain-load variant (`make CXXFLAGS+=-DNC_LSTM_FENCE`; no envmatrix row builds it).  Layout [unit][16 clips] = the B-fragment order: every operand load is a 256-byte row.  Double-buffered by step parity: a
//     workgroup can only publish h_{t+1} after every workgroup of the tile has published h_t, i.e. finished reading h_{t-1}.
//   * Every spin is bounded: a timeout sets *tmo and all workgroups leave (the host reports NC_EDEVICE at the next synchronise).
// Grid = (C/16, column tiles): at most 128 workgroups of 140 KB LDS per launch, all co-resident on the 256 CUs.
// (A granule form of the exchange -- 8-byte {tag, value} stores, data-is-flag -- measured slower: 18.1 vs 17.0 ms on C3.)
struct LstmSeqArgs {
    const float* gi;      // [N,4C,T] input projections incl. b_ih
    const float* whhp;    // [C/4 unit blocks][KS][64] A-fragment image of W_hh
    const float* bhh;     // [4C]
    const float* skip;    // nullable [N,C,T]: added to the last layer's output (SLSTM.cs:50-53)
    float* out;           // [N,C,T]
    int elu_out;          // ELU applied to the stored value (the activation in front of the consuming convolution)
    float* hx;            // [2][tiles][C][16] exchange buffers
    unsigned* flags;      // [tiles][C/16] steps published per workgroup (zeroed before the launch)
    unsigned* tmo;        // timeout word (zeroed at model creation)
    float* cstate;        // [tiles][C][16] cell state carried between the chunk launches of one layer
    // element strides of gi / out over (clip, channel row, step): [N][rows][T] = (rows*T, T, 1) for tensors that meet the convolution
    // stack, [rows][T][N] = (1, T*N, N) for the tensors between the layers of a pipelined call (a step range is then a contiguous
    // column range of one 2-D matrix, which is what the chunked input-projection GEMM of the next layer wants)
    int64_t gi_b, gi_c, gi_t, out_b, out_c, out_t;
    int N, C;
    int64_t T;
    int64_t t0, t1;       // this launch runs steps [t0, t1) of the T-step sequence (state of step t0-1 in hx / cstate / flags)
    int tile0;            // first column tile of this launch
    int acquire;          // NC_SYNC_ACQUIRE=1: agent-scope acquire fence behind the flag poll (the textbook hand-off; the default reads h with
                          // agent-scope loads of drained write-through stores instead: validated by the parity sweeps and run in both forms by
                          // tests/test_children_gpu.py)
};
typedef __attribute__((address_space(1))) unsigned lstm_gu32;
// UB = unit blocks (of 4 hidden units) per workgroup: 4 (16 wavefronts, C/16 workgroups per column tile) or 2 (8 wavefronts, C/8
// workgroups per tile: two chains per SIMD instead of four -- the matrix-core part of a step halves, twice the CUs take part)
// HT (round 6, the default; NC_LSTM_NO_HTILE=1 selects the form above): the h tile goes through LDS and the weights live in REGISTERS.
//   The four unit-block waves of a quarter need the same 8 KB of h_{t-1}; each fetched it for itself -- 128 KB of agent-scope reads per
//   workgroup and step, a good part of the step's ~1.5 us operand fetch at the CU's fabric port.  Now the 16 waves fetch the 32 KB tile
//   ONCE (8 coalesced dword reads per lane, the same agent-scope loads), park it in LDS, and every wave takes its 32 B fragments from
//   there; W_hh (a wave's share is QS x 64 words = 32 registers) stays in registers for the whole sequence instead of being re-read from
//   LDS every step.  One more workgroup barrier per step; same operands, same chains: bit-identical.
template <int KS, int UB = 4, bool HT = true>
__global__ __launch_bounds__(256 * UB, 1) void lstm_seq_kernel(const LstmSeqArgs a) {
    constexpr int QS = KS / 4;                                          // k-steps per quarter
    constexpr int CC = 4 * KS;                                          // hidden units (= a.C)
    extern __shared__ __attribute__((aligned(16))) float lstm_lds[];   // [4 UB waves][QS][64] weights (HT: [CC][16] h tile) | [3][UB][64] f32x4 partial tiles
    __shared__ int dead;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ubl = wave % UB, q = wave / UB;       // unit block within the workgroup, reduction quarter
    const int C = a.C, N = a.N;
    const int64_t T = a.T;
    const int nprod = C / (4 * UB);                 // workgroups per column tile
    const int ubw = blockIdx.x, tile = blockIdx.y;  // tile: local index within this launch
    const int ub = ubw * UB + ubl;                   // unit block of this wave: hidden units 4*ub .. 4*ub+3
    float* Aw = lstm_lds + wave * QS * 64;
    float* const Hs = lstm_lds;
    f32x4v* const part = reinterpret_cast<f32x4v*>(lstm_lds + (HT ? CC * 16 : 4 * UB * QS * 64));
    const float* wsrc = a.whhp + ((int64_t)ub * KS + q * QS) * 64;
    float aw[HT ? QS : 1];
    if constexpr (HT) {
#pragma unroll
        for (int i = 0; i < QS; ++i) aw[i] = wsrc[i * 64 + lane];
    } else {
#pragma unroll
        for (int i = 0; i < QS / 4; ++i)
            __builtin_amdgcn_global_load_lds((lstm_gptr)(wsrc + i * 256 + lane * 4), (lstm_lptr)(Aw + i * 256), 16, 0, 0);
    }
    if (threadIdx.x == 0) dead = 0;
    const int k4 = lane >> 4, cl = lane & 15;
    const int j = ub * 4 + k4;                      // this lane's hidden unit
    const int b = (a.tile0 + tile) * 16 + cl;       // this lane's clip
    const int bb = min(b, N - 1);
    const float* g = a.gi + (int64_t)bb * a.gi_b;
    const int64_t gc = a.gi_c, gt = a.gi_t;
    const float bh0 = a.bhh[j], bh1 = a.bhh[C + j], bh2 = a.bhh[2 * C + j], bh3 = a.bhh[3 * C + j];
    float* const hx0 = a.hx + ((int64_t)(0 * gridDim.y + tile) * C) * 16;
    float* const hx1 = a.hx + ((int64_t)(1 * gridDim.y + tile) * C) * 16;
    unsigned* const flags = a.flags + (int64_t)tile * nprod;
    const int64_t orow = ((int64_t)bb * C + j) * T;                       // skip tensor: always [N,C,T]
    float* const orow_out = a.out + (int64_t)bb * a.out_b + (int64_t)j * a.out_c;
    float* const cs_slot = a.cstate + ((int64_t)tile * C) * 16 + ub * 64 + lane;
    float cst = (q == 0 && a.t0 > 0) ? *cs_slot : 0.0f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // input-projection pre-activations (and the skip value) run one step ahead of their use (q = 0 waves only): they are reads,
    // issued before the step's stores, so they never queue behind a store acknowledgement
    float g0 = 0.0f, g1 = 0.0f, g2 = 0.0f, g3 = 0.0f, sk = 0.0f;
    if (q == 0) {
        const int64_t ts = a.t0;
        g0 = g[(int64_t)j * gc + ts * gt]; g1 = g[(int64_t)(C + j) * gc + ts * gt]; g2 = g[(int64_t)(2 * C + j) * gc + ts * gt]; g3 = g[(int64_t)(3 * C + j) * gc + ts * gt];
        if (a.skip) sk = a.skip[orow + ts];
    }
    for (int64_t t = a.t0; t < a.t1; ++t) {
        f32x4v acc = {0.0f, 0.0f, 0.0f, 0.0f};
        const float c0 = g0, c1 = g1, c2 = g2, c3 = g3, csk = sk;
        const int64_t tn = min(t + 1, T - 1);
        if (t > 0) {                                 // h_{-1} = 0: every quarter chain of step 0 is +0
            if (wave == 0) {
                const unsigned want = (unsigned)t;
                bool ok = false;
                for (unsigned spins = 0; spins < (1u << 21); ++spins) {
                    const unsigned v = lane < nprod ? __hip_atomic_load((lstm_gu32*)(flags + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : want;
                    if (__all(v >= want)) { ok = true; break; }
                    if ((spins & 1023) == 1023 && __hip_atomic_load((lstm_gu32*)a.tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                if (!ok && lane == 0) {
                    __hip_atomic_store((lstm_gu32*)a.tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    dead = 1;
                }
#ifdef NC_LSTM_FENCE
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#else
                if (a.acquire) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
            }
            __syncthreads();
            if (dead) return;
            float hb[QS];
            // the h operands are read with agent-scope (sc1) loads: they observe the producers' write-through stores directly, so the
            // consumer needs no acquire fence (an agent-scope L1 invalidation costs ~1.7 us, more than the loads themselves)
            if constexpr (HT) {
                constexpr int NT = 256 * UB, NL = (CC * 16 + NT - 1) / NT;
                const float* hsrc = (t & 1) ? hx0 : hx1;                 // h_{t-1} sits in buffer (t-1)&1: [unit][16 clips], CC * 16 words
                float hv[NL];
#pragma unroll
                for (int u = 0; u < NL; ++u) {
                    const int idx = min(u * NT + (int)threadIdx.x, CC * 16 - 1);
#ifdef NC_LSTM_FENCE
                    hv[u] = hsrc[idx];
#else
                    hv[u] = __hip_atomic_load(hsrc + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
                }
                if (q == 0) {
                    g0 = g[(int64_t)j * gc + tn * gt]; g1 = g[(int64_t)(C + j) * gc + tn * gt]; g2 = g[(int64_t)(2 * C + j) * gc + tn * gt]; g3 = g[(int64_t)(3 * C + j) * gc + tn * gt];
                    if (a.skip) sk = a.skip[orow + tn];
                }
#pragma unroll
                for (int u = 0; u < NL; ++u)
                    if (u * NT + (int)threadIdx.x < CC * 16) Hs[u * NT + threadIdx.x] = hv[u];
                __syncthreads();
#pragma unroll
                for (int i = 0; i < QS; ++i) hb[i] = Hs[(q * QS + i) * 64 + lane];
#pragma unroll
                for (int i = 0; i < QS; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[i], hb[i], acc, 0, 0, 0);
            } else {
                const float* hp = ((t & 1) ? hx0 : hx1) + (int64_t)q * QS * 64 + lane;   // h_{t-1} sits in buffer (t-1)&1
#pragma unroll
                for (int i = 0; i < QS; ++i) {
#ifdef NC_LSTM_FENCE
                    hb[i] = hp[i * 64];
#else
                    hb[i] = __hip_atomic_load(hp + i * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
                }
                if (q == 0) {
                    g0 = g[(int64_t)j * gc + tn * gt]; g1 = g[(int64_t)(C + j) * gc + tn * gt]; g2 = g[(int64_t)(2 * C + j) * gc + tn * gt]; g3 = g[(int64_t)(3 * C + j) * gc + tn * gt];
                    if (a.skip) sk = a.skip[orow + tn];
                }
#pragma unroll
                for (int i = 0; i < QS; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(Aw[i * 64 + lane], hb[i], acc, 0, 0, 0);
            }
            if (q > 0) part[((q - 1) * UB + ubl) * 64 + lane] = acc;
            __syncthreads();
            if (q == 0) {
                const f32x4v p1 = part[(0 * UB + ubl) * 64 + lane], p2 = part[(1 * UB + ubl) * 64 + lane], p3 = part[(2 * UB + ubl) * 64 + lane];
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = (acc[r] + p1[r]) + (p2[r] + p3[r]);
            }
        } else if (q == 0) {
            g0 = g[(int64_t)j * gc + tn * gt]; g1 = g[(int64_t)(C + j) * gc + tn * gt]; g2 = g[(int64_t)(2 * C + j) * gc + tn * gt]; g3 = g[(int64_t)(3 * C + j) * gc + tn * gt];
            if (a.skip) sk = a.skip[orow + tn];
        }
        if (q == 0) {
            const float pi = c0 + (acc[0] + bh0);
            const float pf = c1 + (acc[1] + bh1);
            const float pg = c2 + (acc[2] + bh2);
            const float po = c3 + (acc[3] + bh3);
            const float ig = nc_sigmoidf(pi), fg = nc_sigmoidf(pf), gg = nc_tanhf(pg), og = nc_sigmoidf(po);
            cst = (fg * cst) + (ig * gg);
            const float h = og * nc_tanhf(cst);
            if (t + 1 < T) {
                // publish h_t: write-through payload, drained per wave, then one flag store for the workgroup
                float* hq = ((t & 1) ? hx1 : hx0) + ub * 64 + lane;   // [unit j][clip cl] = j*16 + cl = ub*64 + lane
                __hip_atomic_store(hq, h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (b < N) {
                const float yo = a.skip ? h + csk : h;
                orow_out[t * a.out_t] = a.elu_out ? nc_eluf(yo) : yo;
            }
        }
        if (t + 1 < T) {
            __syncthreads();   // the four publishing waves have drained their stores (and the partial tiles are free again)
            if (threadIdx.x == 0) __hip_atomic_store((lstm_gu32*)(flags + ubw), (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (q == 0 && a.t1 < T) *cs_slot = cst;   // the next chunk launch of this layer resumes from here
}

// Euclidean codebook search, D <= 128 (EuclideanCodebook.cs:155-182): per frame dist_n = (|x|^2 + |e_n|^2) - 2*(x.e_n) with fma
// chains over d ascending, argmin with lowest-index ties; then residual -= embed[idx] (ResidualVectorQuantizer.cs:150-152).
// Block = EQ_F frames x 256 threads; thread n scans codes n, n+256, ...; codebook transposed [D][N] streams from L2.
constexpr int EQ_F = 8, EQ_MAXD = 128, EQ_NPT = 4;
__global__ __launch_bounds__(256) void euclid_vq_kernel(float* __restrict__ residual, const float* __restrict__ cbT,
                                                        const float* __restrict__ cb, const float* __restrict__ c2, int N, int D, int B,
                                                        int64_t T, int64_t* __restrict__ codes, int64_t codes_bstride) {
    __shared__ float es[EQ_F][EQ_MAXD];
    __shared__ float e2s[EQ_F];
    __shared__ float bd[EQ_F][4];
    __shared__ int bi[EQ_F][4];
    __shared__ int win[EQ_F];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t f0 = (int64_t)blockIdx.x * EQ_F, total = (int64_t)B * T;
    for (int i = tid; i < EQ_F * D; i += 256) {
        const int f = i / D, d = i - f * D;
        const int64_t fr = f0 + f;
        float v = 0.0f;
        if (fr < total) { const int64_t b = fr / T, t = fr - b * T; v = residual[(b * D + d) * T + t]; }
        es[f][d] = v;
    }
    __syncthreads();
    if (tid < EQ_F) {
        float a = 0.0f;
        for (int d = 0; d < D; ++d) a = nc_fma(es[tid][d], es[tid][d], a);
        e2s[tid] = a;
    }
    __syncthreads();
    float best[EQ_F];
    int besti[EQ_F];
#pragma unroll
    for (int f = 0; f < EQ_F; ++f) { best[f] = __builtin_inff(); besti[f] = 0x7fffffff; }
    for (int n0 = 0; n0 < N; n0 += 256 * EQ_NPT) {
        float cr[EQ_NPT][EQ_F];
#pragma unroll
        for (int u = 0; u < EQ_NPT; ++u)
#pragma unroll
            for (int f = 0; f < EQ_F; ++f) cr[u][f] = 0.0f;
        for (int d = 0; d < D; ++d) {
            float cv[EQ_NPT];
#pragma unroll
            for (int u = 0; u < EQ_NPT; ++u) {
                const int n = n0 + u * 256 + tid;
                cv[u] = n < N ? cbT[(int64_t)d * N + n] : 0.0f;
            }
#pragma unroll
            for (int f = 0; f < EQ_F; ++f) {
                const float ev = es[f][d];
#pragma unroll
                for (int u = 0; u < EQ_NPT; ++u) cr[u][f] = nc_fma(ev, cv[u], cr[u][f]);
            }
        }
#pragma unroll
        for (int u = 0; u < EQ_NPT; ++u) {
            const int n = n0 + u * 256 + tid;
            if (n < N) {
                const float cc = c2[n];
#pragma unroll
                for (int f = 0; f < EQ_F; ++f) {
                    const float dist = (e2s[f] + cc) - 2.0f * cr[u][f];
                    if (nc_argmin_scan(dist, best[f])) { best[f] = dist; besti[f] = n; }
                }
            }
        }
    }
#pragma unroll
    for (int f = 0; f < EQ_F; ++f) {
        float d0 = best[f];
        int i0 = besti[f];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float od = __shfl_xor(d0, off, 64);
            const int oi = __shfl_xor(i0, off, 64);
            if (nc_argmin_before(od, oi, d0, i0)) { d0 = od; i0 = oi; }
        }
        if (lane == 0) { bd[f][wave] = d0; bi[f][wave] = i0; }
    }
    __syncthreads();
    if (tid < EQ_F) {
        float d0 = bd[tid][0];
        int i0 = bi[tid][0];
        for (int w = 1; w < 4; ++w)
            if (nc_argmin_before(bd[tid][w], bi[tid][w], d0, i0)) { d0 = bd[tid][w]; i0 = bi[tid][w]; }
        if (i0 == 0x7fffffff) i0 = 0;
        win[tid] = i0;
        const int64_t fr = f0 + tid;
        if (fr < total) { const int64_t b = fr / T, t = fr - b * T; codes[b * codes_bstride + t] = (int64_t)i0; }
    }
    __syncthreads();
    for (int i = tid; i < EQ_F * D; i += 256) {
        const int f = i / D, d = i - f * D;
        const int64_t fr = f0 + f;
        if (fr < total) {
            const int64_t b = fr / T, t = fr - b * T;
            residual[(b * D + d) * T + t] = es[f][d] - cb[(int64_t)win[f] * D + d];
        }
    }
}

// All n_q stages of the Euclidean RVQ for a block of 32 frames in ONE launch, cross terms on the matrix cores
// (ResidualVectorQuantizer.cs:139-156 over EuclideanCodebook.cs:155-182).  Same arithmetic as euclid_vq_kernel, operation for
// operation: cr_n = fma chain over d ascending from +0 of e_d * c_{n,d} -- which is what a chain of v_mfma_f32_32x32x2_f32 over
// k = d computes for output (row n, column frame) -- then dist_n = (|e|^2 + |c_n|^2) - 2 * cr_n, argmin with the lowest index on
// ties, residual -= embed[idx].  Rows = codes (A fragments straight from the transposed codebook [D][N]: 32 consecutive codes per
// lane half, L2-resident), columns = frames (B fragments from the residual block in LDS, [d][frame]); wave w scans codes
// [w*N/4, (w+1)*N/4) 128 codes at a time (four independent accumulation chains; row l of tile i = code 4 l + i, so a lane's four A
// values per k are one 16-byte load), the reads two groups of steps ahead of the matrix cores.  The residual block never leaves LDS between
// the stages.  8 launches of 63-90 us (600 workgroups re-streaming the 512 KB codebook each) become one of ~0.2 ms on C3.
constexpr int EM_F = 32, EM_MAXD = 128;
typedef float em_f32x16 __attribute__((ext_vector_type(16)));
typedef float em_f32x4 __attribute__((ext_vector_type(4)));
// NWV wavefronts share a stage's codebook scan (N / NWV codes each: 8 waves halve the scan of a workgroup that sits alone on its CU)
template <int DD, int NWV = 4>
__global__ __launch_bounds__(64 * NWV) void euclid_rvq_mfma_kernel(const float* __restrict__ residual, const float* const* __restrict__ cbT_ptrs,
                                                              const float* const* __restrict__ cb_ptrs, const float* const* __restrict__ c2_ptrs,
                                                              int n_q, int N, int B, int64_t T, int64_t* __restrict__ codes,
                                                              int64_t codes_bstride) {
    constexpr int D = DD;
    __shared__ float es[EM_MAXD][EM_F + 1];   // residual block [d][frame]: lane (frame, k half) of a B fragment reads es[2kp + half][frame]; rows padded by one
                                              // word -- the residual update walks d across the lanes (unpadded: every lane of a wave on ONE bank)
    __shared__ float e2s[EM_F];
    __shared__ __attribute__((aligned(16))) float c2s[1024];   // |c_n|^2 of the stage (N <= 1024: the launcher routes larger codebooks to euclid_vq_kernel)
    __shared__ float bd[NWV][EM_F];
    __shared__ int bi[NWV][EM_F];
    __shared__ int win[EM_F];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int64_t f0 = (int64_t)blockIdx.x * EM_F, total = (int64_t)B * T;
    for (int i = tid; i < EM_F * D; i += 64 * NWV) {
        const int d = i >> 5, f = i & 31;
        const int64_t fr = f0 + f;
        float v = 0.0f;
        if (fr < total) { const int64_t b = fr / T, t = fr - b * T; v = residual[(b * D + d) * T + t]; }
        es[d][f] = v;
    }
    const int npw = N / NWV;                 // codes per wave (a multiple of 128)
    const int npass = npw / 128;
    // (the stage's pointers come out of a pointer table: say that they are global memory, or the reads are issued as flat loads, which
    // count against the LDS counter too and serialise with the B-fragment reads)
    typedef __attribute__((address_space(1))) const em_f32x4* em_gp4;
    typedef __attribute__((address_space(1))) const float* em_gp1;
    // 128 codes per pass as four row tiles; row l of tile i is code n0 + 4 l + i, so the four A values a lane needs for one k are four
    // consecutive codes of the transposed codebook: ONE 16-byte load, 512 contiguous bytes per lane half.  The reads run two groups of
    // G matrix-core steps ahead through a ring of FOUR register sets (a pass is 8 groups: every pass starts on set 0, so the ring runs
    // on across the passes AND the stages -- the first two groups of the next pass / the next stage's first pass are in flight under the
    // last two groups of this one, the argmin, the hand-off and the residual update; filled per pass, every pass and every stage began
    // with an exposed L2 round trip: 288 -> 251 us on C3's 150-workgroup grid together with the two changes below)
    constexpr int G = 8, NG = DD / 2 / G;    // matrix-core steps per group, groups per pass
    static_assert(NG % 4 == 0, "the four-set ring must start every pass on set 0");
    const int64_t kstride = (int64_t)2 * N / 4;                            // float4 words per MFMA step (two codebook rows)
    auto pass_ptr = [&](int q, int pass) __attribute__((always_inline)) -> em_gp4 {
        return (em_gp4)(cbT_ptrs[q] + (int64_t)hi * N + wave * npw + pass * 128 + 4 * l31);   // k = hi at kp = 0
    };
    em_f32x4 av[4][G];
    em_gp4 ap = pass_ptr(0, 0);
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int u = 0; u < G; ++u) av[g][u] = ap[(int64_t)(g * G + u) * kstride];
    // |c_n|^2 of a stage: in registers one stage ahead, in LDS for the stage's scans (the argmin read them from global memory per pass)
    constexpr int C2R = 1024 / (64 * NWV);
    float c2r[C2R];
#pragma unroll
    for (int i = 0; i < C2R; ++i) c2r[i] = ((em_gp1)c2_ptrs[0])[min(tid + i * 64 * NWV, N - 1)];
    for (int q = 0; q < n_q; ++q) {
        const float* __restrict__ cb = cb_ptrs[q];
        __syncthreads();                     // es holds the residual entering this stage
#pragma unroll
        for (int i = 0; i < C2R; ++i)
            if (tid + i * 64 * NWV < N) c2s[tid + i * 64 * NWV] = c2r[i];
        if (q + 1 < n_q) {
#pragma unroll
            for (int i = 0; i < C2R; ++i) c2r[i] = ((em_gp1)c2_ptrs[q + 1])[min(tid + i * 64 * NWV, N - 1)];
        }
        if (tid < EM_F) {
            float a = 0.0f;
            // |e|^2: ONE fma chain over d ascending (the canonical order), the LDS reads 16 at a time ahead of their 16 dependent fmas
            // (rolled, every fma waited for its own read: 2.0 us of a 27 us stage)
#pragma unroll 1
            for (int d0 = 0; d0 < D; d0 += 16) {
                float ev[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) ev[u] = es[d0 + u][tid];
#pragma unroll
                for (int u = 0; u < 16; ++u) a = nc_fma(ev[u], ev[u], a);
            }
            e2s[tid] = a;
        }
        __syncthreads();
        const float e2 = e2s[l31];
        float best = __builtin_inff();
        int besti = 0x7fffffff;
        for (int pass = 0; pass < npass; ++pass) {
            const int n0 = wave * npw + pass * 128;
            const bool last_pass = pass + 1 == npass;
            const bool has_next = !last_pass || q + 1 < n_q;
            const em_gp4 ap_next = last_pass ? pass_ptr(min(q + 1, n_q - 1), 0) : ap + 32;   // (+ 128 codes)
            em_f32x16 acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 2 < NG) {
#pragma unroll
                    for (int u = 0; u < G; ++u) av[(g + 2) % 4][u] = ap[(int64_t)((g + 2) * G + u) * kstride];
                } else if (has_next) {
#pragma unroll
                    for (int u = 0; u < G; ++u) av[(g + 2) % 4][u] = ap_next[(int64_t)((g + 2 - NG) * G + u) * kstride];
                }
                __builtin_amdgcn_sched_barrier(0);   // the reads of group g+2 stay ahead of the matrix-core steps of group g
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const float bv = es[2 * (g * G + u) + hi][l31];
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g % 4][u][i], bv, acc[i], 0, 0, 0);
                }
            }
            ap = ap_next;
            // D[row = (r & 3) + 8 (r >> 2) + 4 hi][column = l31].  A lane meets its codes in ASCENDING order (code = n0 + 4 row + i: rows ascend with r,
            // i is the inner loop, n0 ascends over the passes), so the ascending-scan form of ATen's order applies: an equal distance never
            // replaces the incumbent, a NaN takes over once.  (The lane halves and the waves are merged with the any-order form below.)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nb = n0 + 4 * ((r & 3) + 8 * (r >> 2) + 4 * hi);
                const em_f32x4 cc = *reinterpret_cast<const em_f32x4*>(c2s + nb);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float dist = (e2 + cc[i]) - 2.0f * acc[i][r];
                    if (nc_argmin_scan(dist, best)) { best = dist; besti = nb + i; }
                }
            }
        }
        {   // the two lane halves hold the same frame; then the four waves meet in LDS
            const float od = __shfl_xor(best, 32, 64);
            const int oi = __shfl_xor(besti, 32, 64);
            if (nc_argmin_before(od, oi, best, besti)) { best = od; besti = oi; }
            if (hi == 0) { bd[wave][l31] = best; bi[wave][l31] = besti; }
        }
        __syncthreads();
        if (tid < EM_F) {
            float d0 = bd[0][tid];
            int i0 = bi[0][tid];
            for (int w = 1; w < NWV; ++w)
                if (nc_argmin_before(bd[w][tid], bi[w][tid], d0, i0)) { d0 = bd[w][tid]; i0 = bi[w][tid]; }
            if (i0 == 0x7fffffff) i0 = 0;
            win[tid] = i0;
            const int64_t fr = f0 + tid;
            if (fr < total) { const int64_t b = fr / T, t = fr - b * T; codes[b * codes_bstride + (int64_t)q * T + t] = (int64_t)i0; }
        }
        __syncthreads();
        {   // residual -= embed[idx]: thread (d = tid % D, frames f = tid / D + (64 NWV / D) u) -- a frame's code vector is one coalesced 512-byte
            // read; all of a thread's reads are issued before the first is used (one L2 round trip per stage instead of one per element)
            constexpr int FS = 64 * NWV / D, NU = EM_F / FS;
            static_assert((64 * NWV) % D == 0 && EM_F % FS == 0, "update map");
            const int d = tid % D, fb = tid / D;
            float cv[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) cv[u] = ((em_gp1)cb)[(int64_t)win[fb + FS * u] * D + d];
#pragma unroll
            for (int u = 0; u < NU; ++u) es[d][fb + FS * u] = es[d][fb + FS * u] - cv[u];
        }
    }
}

// ResidualVectorQuantizer.Decode (:107-124): emb = ((0 + e_0[idx_0]) + e_1[idx_1]) + ...
__global__ void emb_sum_kernel(const int64_t* __restrict__ codes, const float* const* __restrict__ cbs, int n_q, int N, int D, int B,
                               int64_t T, float* __restrict__ emb) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * D * T) return;
    const int64_t t = i % T, r = i / T;
    const int d = (int)(r % D);
    const int64_t b = r / D;
    float a = 0.0f;
    for (int q = 0; q < n_q; ++q) {
        int64_t c = codes[(b * n_q + q) * T + t];
        if (c < 0) c = 0;
        if (c >= N) c = N - 1;
        a = a + cbs[q][c * D + d];
    }
    emb[i] = a;
}

// DSP.LinearOverlapAdd (AudioTensorDSP.cs:161-261): out[r,i] = (sum_f frame_f[r, i - f*stride] * w[i - f*stride]) / sw[i]
// Frame pointers / lengths come from device arrays (any number of segments); only the frames that can cover sample t are visited,
// in ascending order -- the same additions the all-frames loop performs.
__global__ void overlap_add_kernel(const float* const* __restrict__ fp, const int64_t* __restrict__ flen, int nfr, int64_t L0,
                                   const float* __restrict__ w, const float* __restrict__ sw, int64_t rows, int64_t stride, int64_t total,
                                   float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * total) return;
    const int64_t r = i / total, t = i - r * total;
    const int64_t f_lo = t >= L0 ? (t - L0) / stride + 1 : 0;
    const int64_t f_hi = min((int64_t)nfr - 1, t / stride);
    float a = 0.0f;
    for (int64_t f = f_lo; f <= f_hi; ++f) {
        const int64_t q = t - f * stride, len = flen[f];
        if (q >= 0 && q < len) a = a + fp[f][r * len + q] * w[q];
    }
    out[i] = a / sw[t];
}

// ---------------------------------------------------------------------------------------------- model
EncodecModel::EncodecModel(const nc_encodec_config& c) : cfg(c) {
    if (c.n_ratios <= 0 || c.n_ratios > 8) fail(NC_EINVAL, "ratios must hold 1..8 entries");
    if (c.channels <= 0 || c.channels > 2) fail(NC_EINVAL, "Invalid number of channels: %d", c.channels);          // Encodec.cs:267-270
    if (c.dimension <= 0 || c.dimension > EQ_MAXD || c.n_filters <= 0 || c.sample_rate <= 0 || c.codebook_size <= 0 || c.n_codebooks <= 0 ||
        c.n_codebooks > 64 || c.frame_rate <= 0 || c.lstm_layers < 0 || c.lstm_layers > 4 || c.compress <= 0)
        fail(NC_EINVAL, "Encodec config fields out of range");
    if (c.kernel_size != 7 || c.last_kernel_size != 7 || c.residual_kernel_size != 3)
        fail(NC_EUNSUPPORTED, "SEANet kernel sizes other than 7/7/3 are not instantiated");
    if (c.time_group_norm && c.causal) fail(NC_EINVAL, "GroupNorm doesn't support causal evaluation");               // NormConv1d.cs:143-147
    if ((c.segment_length > 0) != (c.segment_stride > 0)) fail(NC_EINVAL, "segment_length and segment_stride go together");
    hop = 1;
    for (int i = 0; i < c.n_ratios; ++i) {
        if (c.ratios[i] <= 0) fail(NC_EINVAL, "ratios must be positive");
        hop *= c.ratios[i];
    }
    set_bandwidth(c.bandwidth);
}

void EncodecModel::set_bandwidth(float bw) {
    // ResidualVectorQuantizer.Encode (:139-147): nQ = max(1, floor(bw*1000 / (log2(bins)*frameRate)))
    const double bw_per_q = std::log2((double)cfg.codebook_size) * cfg.frame_rate;
    int n = cfg.n_codebooks;
    if (bw > 0) n = (int)std::max(1.0, std::floor((double)bw * 1000.0 / bw_per_q));
    if (n > cfg.n_codebooks) fail(NC_EINVAL, "This model doesn't support the bandwidth %g kbps", (double)bw);          // Encodec.cs:411-416
    n_q = n;
    cfg.bandwidth = bw;
}

EncodecModel::Plan EncodecModel::plan_sconv(int64_t L, int k, int stride, int dil) const {
    Plan p;
    const int64_t eff = (int64_t)(k - 1) * dil + 1, pt = eff - stride;
    const float nf = (float)(L - eff + pt) / (float)stride + 1.0f;                      // SConv1d.cs:245-250 (float division)
    const int64_t ideal = ((int64_t)std::ceil(nf) - 1) * stride + (eff - pt);
    const int64_t extra = ideal - L;
    if (cfg.causal) { p.left = pt; p.right = extra; }
    else { const int64_t r = pt / 2; p.left = pt - r; p.right = r + extra; }
    const int64_t mx = std::max(p.left, p.right);
    p.Lz = L <= mx ? L + (mx - L + 1) : L;                                                 // SConv1d.cs:258-274 (D9)
    p.Lp = p.Lz + p.left + p.right;
    p.Lout = (p.Lp - eff) / stride + 1;
    return p;
}

int64_t EncodecModel::frames_for(int64_t L) const {
    L = plan_sconv(L, cfg.kernel_size, 1, 1).Lout;
    for (int i = cfg.n_ratios - 1; i >= 0; --i) {
        L = plan_sconv(L, cfg.residual_kernel_size, 1, 1).Lout;
        L = plan_sconv(L, 2 * cfg.ratios[i], cfg.ratios[i], 1).Lout;
    }
    return plan_sconv(L, cfg.last_kernel_size, 1, 1).Lout;
}
int64_t EncodecModel::decoded_for(int64_t Tz) const {
    int64_t L = plan_sconv(Tz, cfg.kernel_size, 1, 1).Lout;
    for (int i = 0; i < cfg.n_ratios; ++i) {
        L = L * cfg.ratios[i];
        L = plan_sconv(L, cfg.residual_kernel_size, 1, 1).Lout;
    }
    return plan_sconv(L, cfg.last_kernel_size, 1, 1).Lout;
}

std::vector<EncodecModel::Seg> EncodecModel::segments(int64_t T) const {
    std::vector<Seg> v;
    const int64_t seg = cfg.segment_length > 0 ? cfg.segment_length : T, stride = cfg.segment_stride > 0 ? cfg.segment_stride : T;
    for (int64_t off = 0; off < T; off += stride) {                                         // Encodec.cs:278-282
        Seg s;
        s.off = off;
        s.len = std::min(off + seg, T) - off;
        s.frames = frames_for(s.len);
        v.push_back(s);
    }
    return v;
}

static void upload(DevBuf& d, const float* h, size_t n) {
    d.reserve(n * sizeof(float));
    NC_HIP(hipMemcpy(d.p, h, n * sizeof(float), hipMemcpyHostToDevice));
}

void EncodecModel::load_sconv(const Blob& b, const std::string& key, SConv& L, int Cin, int Cout, int K, int stride, bool transposed) {
    const BlobTensor* w = b.find(key + ".conv.weight");
    const BlobTensor* bias = b.find(key + ".conv.bias");
    const int64_t d0 = transposed ? Cin : Cout, d1 = transposed ? Cout : Cin;
    std::vector<float> folded;
    const float* dense;
    if (w) {
        if (w->dims.size() != 3 || w->dims[0] != d0 || w->dims[1] != d1 || w->dims[2] != K) fail(NC_EINVAL, "%s.conv.weight has the wrong shape", key.c_str());
        dense = static_cast<const float*>(w->data);
    } else {
        const BlobTensor& v = b.get(key + ".conv.weight_v");
        const BlobTensor& g = b.get(key + ".conv.weight_g");
        if (v.dims.size() != 3 || v.dims[0] != d0 || v.dims[1] != d1 || v.dims[2] != K || g.numel() != d0)
            fail(NC_EINVAL, "%s.conv.weight_v/g have the wrong shape", key.c_str());
        folded.resize((size_t)v.numel());
        fold_weight_norm_snac(static_cast<const float*>(v.data), static_cast<const float*>(g.data), d0, d1 * K, folded.data());   // D3
        dense = folded.data();
    }
    if (bias && bias->numel() != Cout) fail(NC_EINVAL, "%s.conv.bias has the wrong length", key.c_str());
    L.K = K; L.stride = stride; L.Cin = Cin; L.Cout = Cout; L.transposed = transposed;
    L.conv.kclass = transposed ? NC_KC_CONV_UP : (stride > 1 ? NC_KC_CONV_DOWN : (K == 1 ? NC_KC_CONV_K1 : (Cin <= 2 ? NC_KC_STEM : (Cout <= 2 ? NC_KC_HEAD : NC_KC_CONV_MISC))));
    L.conv.build(dense, bias ? static_cast<const float*>(bias->data) : nullptr, Cin, Cout, K, stride, 0, 1, 0, transposed);
    if (cfg.time_group_norm) {
        const BlobTensor& gw = b.get(key + ".norm.weight");
        const BlobTensor& gb = b.get(key + ".norm.bias");
        if (gw.numel() != Cout || gb.numel() != Cout) fail(NC_EINVAL, "%s.norm has the wrong shape", key.c_str());
        upload(L.gamma, static_cast<const float*>(gw.data), Cout);
        upload(L.beta, static_cast<const float*>(gb.data), Cout);
    }
}

void EncodecModel::load_resblock(const Blob& b, const std::string& key, ResBlock& r, int dim) {
    const int h = dim / cfg.compress;
    load_sconv(b, key + ".block.1", r.c1, dim, h, cfg.residual_kernel_size, 1, false);
    load_sconv(b, key + ".block.3", r.c2, h, dim, 1, 1, false);
    load_sconv(b, key + ".shortcut", r.sc, dim, dim, 1, 1, false);
}

void EncodecModel::load_lstm(const Blob& b, const std::string& key, Lstm& l, int C) {
    l.C = C;
    l.layers.clear();
    for (int i = 0; i < cfg.lstm_layers; ++i) {
        l.layers.emplace_back(new LstmLayer());
        LstmLayer& y = *l.layers.back();
        char sfx[32];
        snprintf(sfx, sizeof sfx, "_l%d", i);
        const BlobTensor& wih = b.get(key + ".lstm.weight_ih" + sfx);
        const BlobTensor& whh = b.get(key + ".lstm.weight_hh" + sfx);
        const BlobTensor& bih = b.get(key + ".lstm.bias_ih" + sfx);
        const BlobTensor& bhh = b.get(key + ".lstm.bias_hh" + sfx);
        if (wih.numel() != (int64_t)4 * C * C || whh.numel() != (int64_t)4 * C * C || bih.numel() != 4 * C || bhh.numel() != 4 * C)
            fail(NC_EINVAL, "%s: LSTM tensors have the wrong shape", key.c_str());
        y.ih.kclass = NC_KC_CONV_K1;
        y.ih.build(static_cast<const float*>(wih.data), static_cast<const float*>(bih.data), C, 4 * C, 1, 1, 0, 1, 0, false);
        upload(y.whh, static_cast<const float*>(whh.data), (size_t)4 * C * C);
        if (C % 4 == 0) {   // A-fragment image for lstm_step_mfma_kernel: [unit block][k-step][lane], lane = (k4 << 4) | (unit << 2) | gate
            const float* w = static_cast<const float*>(whh.data);
            std::vector<float> pk((size_t)4 * C * C);
            const int KS = C / 4;
            for (int ub = 0; ub < C / 4; ++ub)
                for (int kp = 0; kp < KS; ++kp)
                    for (int l = 0; l < 64; ++l) {
                        const int k4 = l >> 4, r = l & 15, u = r >> 2, gate = r & 3;
                        pk[((size_t)ub * KS + kp) * 64 + l] = w[(size_t)(gate * C + ub * 4 + u) * C + 4 * kp + k4];
                    }
            upload(y.whhp, pk.data(), pk.size());
        }
        upload(y.bhh, static_cast<const float*>(bhh.data), (size_t)4 * C);
        upload(y.bih, static_cast<const float*>(bih.data), (size_t)4 * C);
#ifdef NC_EXPERIMENTS
        if (lstm2_supported(C)) {   // fragment images of nc_lstm.hip: W_hh of every layer (per-layer split kernel), W_ih of the upper one (fused kernel)
            std::vector<float> img((size_t)4 * C * C);
            lstm2_pack_image(static_cast<const float*>(whh.data), C, img.data());
            upload(y.w2hh, img.data(), img.size());
            if (i == 1 && cfg.lstm_layers == 2) {
                lstm2_pack_image(static_cast<const float*>(wih.data), C, img.data());
                upload(y.w2ih, img.data(), img.size());
            }
        }
#endif
    }
}

void EncodecModel::load(const Blob& b) {
    use_device();
    char nm[128];
    const int nf = cfg.n_filters;
    load_sconv(b, "encoder.layers.0", enc_in, cfg.channels, nf, cfg.kernel_size, 1, false);
    int n = 1, mult = 1;
    for (int i = 0; i < cfg.n_ratios; ++i) {
        const int r = cfg.ratios[cfg.n_ratios - 1 - i], d = mult * nf;
        snprintf(nm, sizeof nm, "encoder.layers.%d", n);
        load_resblock(b, nm, enc_res[i], d);
        snprintf(nm, sizeof nm, "encoder.layers.%d", n + 2);
        load_sconv(b, nm, enc_down[i], d, 2 * d, 2 * r, r, false);
        n += 3;
        mult *= 2;
    }
    snprintf(nm, sizeof nm, "encoder.layers.%d", n);
    load_lstm(b, nm, enc_lstm, mult * nf);
    snprintf(nm, sizeof nm, "encoder.layers.%d", n + 2);
    load_sconv(b, nm, enc_out, mult * nf, cfg.dimension, cfg.last_kernel_size, 1, false);
    books.clear();
    std::vector<const float*> ptrs;
    for (int i = 0; i < cfg.n_codebooks; ++i) {
        snprintf(nm, sizeof nm, "quantizer.layers.%d.codebook.embed", i);
        const BlobTensor& e = b.get(nm);
        if (e.dims.size() != 2 || e.dims[0] != cfg.codebook_size || e.dims[1] != cfg.dimension) fail(NC_EINVAL, "%s has the wrong shape", nm);
        books.emplace_back(new Codebook());
        books.back()->build(static_cast<const float*>(e.data), cfg.codebook_size, cfg.dimension);
        ptrs.push_back(books.back()->cb.as<float>());
    }
    book_ptrs.reserve(ptrs.size() * sizeof(float*));
    NC_HIP(hipMemcpy(book_ptrs.p, ptrs.data(), ptrs.size() * sizeof(float*), hipMemcpyHostToDevice));
    {   // transposed codebooks and squared norms of all stages, for the stage-fused RVQ kernel
        std::vector<const float*> pt, p2;
        for (auto& bk : books) { pt.push_back(bk->cbT.as<float>()); p2.push_back(bk->c2.as<float>()); }
        book_ptrsT.reserve(pt.size() * sizeof(float*));
        book_ptrs2.reserve(p2.size() * sizeof(float*));
        NC_HIP(hipMemcpy(book_ptrsT.p, pt.data(), pt.size() * sizeof(float*), hipMemcpyHostToDevice));
        NC_HIP(hipMemcpy(book_ptrs2.p, p2.data(), p2.size() * sizeof(float*), hipMemcpyHostToDevice));
    }
    load_sconv(b, "decoder.layers.0", dec_in, cfg.dimension, mult * nf, cfg.kernel_size, 1, false);
    load_lstm(b, "decoder.layers.1", dec_lstm, mult * nf);
    n = 2;
    for (int i = 0; i < cfg.n_ratios; ++i) {
        const int r = cfg.ratios[i], d = mult * nf;
        snprintf(nm, sizeof nm, "decoder.layers.%d", n + 1);
        load_sconv(b, nm, dec_up[i], d, d / 2, 2 * r, r, true);
        snprintf(nm, sizeof nm, "decoder.layers.%d", n + 2);
        load_resblock(b, nm, dec_res[i], d / 2);
        n += 3;
        mult /= 2;
    }
    snprintf(nm, sizeof nm, "decoder.layers.%d", n + 1);
    load_sconv(b, nm, dec_out, nf, cfg.channels, cfg.last_kernel_size, 1, false);
    gn_counters.reserve((size_t)3 * 2 * GN_MAX_SAMPLES * sizeof(unsigned));
    NC_HIP(hipMemset(gn_counters.p, 0, (size_t)3 * 2 * GN_MAX_SAMPLES * sizeof(unsigned)));
    if (!lstm_tmo_host) {
        void* hp = nullptr;
        NC_HIP(hipHostMalloc(&hp, 64, hipHostMallocMapped));
        std::memset(hp, 0, 64);
        lstm_tmo_host = static_cast<unsigned*>(hp);
        void* dp = nullptr;
        NC_HIP(hipHostGetDevicePointer(&dp, hp, 0));
        lstm_tmo_dev = static_cast<unsigned*>(dp);
    }
    if (!ev_fork) {
        for (int i = 0; i < 2; ++i) {
            NC_HIP(hipStreamCreateWithFlags(&side_stream[i], hipStreamNonBlocking));
            NC_HIP(hipEventCreateWithFlags(&ev_join[i], hipEventDisableTiming));
        }
        NC_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    }
    NC_HIP(hipDeviceSynchronize());
    loaded = true;
}

// ---- per-device ticket of the persistent LSTM sections (include/nc_mi355x.h "threading": distinct handles may run concurrently) --------
// A persistent launch needs ALL its workgroups resident (one per CU at ~140 KB of LDS); a handle budgets its own launches for that (<= 64
// workgroups per launch, two pipelined layers, a concurrent tail-segment group: <= 192 of the 256 CUs).  Two handles driven from two host
// threads on ONE device would double that: partially resident launches then spin for CUs that other partially resident launches hold, until
// the bounded spins time out and both handles drop to the step-wise kernels for good.  So the LSTM sections of DIFFERENT handles on one
// device run one after the other on the GPU: a section waits for the event recorded behind the previous section (of any handle) and
// records it again behind itself -- stream-ordered, no host wait; the mutex only keeps two host threads from interleaving their enqueues
// (held for the ~100 us a section takes to enqueue).  A device with one LSTM-running handle never waits.  Everything else of the two
// handles (convolutions, quantizer) still overlaps.  Other PROCESSES on the same device are outside its reach (the timeout path remains).
struct LstmTicket {
    std::mutex mu;
    hipEvent_t ev = nullptr;   // behind the last persistent section enqueued on this device
    int live = 0;              // handles on this device that have run a persistent section
    const void* last_owner = nullptr;   // the handle that recorded `ev`
};
static LstmTicket& lstm_ticket_of(int device) {
    // (leaked on purpose: handles destroyed during static destruction still find their ticket -- ADVICE r5)
    static std::mutex* m = new std::mutex();
    static std::map<int, LstmTicket*>* t = new std::map<int, LstmTicket*>();
    std::lock_guard<std::mutex> lk(*m);
    LstmTicket*& p = (*t)[device];
    if (!p) p = new LstmTicket();
    return *p;
}
namespace {
struct LstmSection {
    LstmTicket& t;
    EncodecModel& m;
    hipStream_t s;
    std::unique_lock<std::mutex> lk;
    int unwinding_at_entry;
    LstmSection(EncodecModel& model, hipStream_t stream)
        : t(model.lstm_ticket ? *model.lstm_ticket : lstm_ticket_of(model.device)), m(model), s(stream), lk(t.mu), unwinding_at_entry(std::uncaught_exceptions()) {
        if (!m.lstm_ticket) { m.lstm_ticket = &t; ++t.live; }
        if (!t.ev) NC_HIP(hipEventCreateWithFlags(&t.ev, hipEventDisableTiming));
        // (with more than one live handle EVERY section waits for the one before it, a handle's own included: the event is re-recorded by
        //  each section, so the chain main group -> side group -> next handle is what keeps a third party behind all of them)
        else if (t.live > 1) NC_HIP(hipStreamWaitEvent(s, t.ev, 0));
    }
    ~LstmSection() {
        if (!t.ev) return;
        if (std::uncaught_exceptions() > unwinding_at_entry) {
            // error path: the layer-pipelined form may have left persistent launches on the second stream that were never joined into `s`;
            // the ticket must not be handed on before they are done
            if (m.lstm_stream) (void)hipStreamSynchronize(m.lstm_stream);
            (void)hipStreamSynchronize(s);
        }
        (void)hipEventRecord(t.ev, s);
        t.last_owner = &m;
    }
};
}  // namespace

EncodecModel::~EncodecModel() {
    for (int i = 0; i < 2; ++i) {
        if (side_stream[i]) (void)hipStreamDestroy(side_stream[i]);
        if (ev_join[i]) (void)hipEventDestroy(ev_join[i]);
    }
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (lstm_stream) (void)hipStreamDestroy(lstm_stream);
    for (hipEvent_t e : lstm_events) (void)hipEventDestroy(e);
    for (hipEvent_t e : ola_ev)
        if (e) (void)hipEventDestroy(e);
    if (ola_pin) (void)hipHostFree(ola_pin);
    if (lstm_tmo_host) (void)hipHostFree(lstm_tmo_host);
    if (lstm_ticket) {
        std::lock_guard<std::mutex> lk(lstm_ticket->mu);
        --lstm_ticket->live;
        if (lstm_ticket->last_owner == this) lstm_ticket->last_owner = nullptr;
        if (lstm_ticket->live == 0 && lstm_ticket->ev) {   // (the last handle of the device: the event goes with it; the next first section makes a new one)
            (void)hipEventDestroy(lstm_ticket->ev);
            lstm_ticket->ev = nullptr;
        }
    }
}

// Called where the host knows the stream is idle (nc_codec_synchronize, the host-pointer entry points, nc_codec_check_errors) and at the
// start of every device-pointer call: a persistent LSTM launch that timed out invalidated the call it belonged to.  The handle then
// switches to the step-wise kernels (fresh launches need no co-residency), so the caller's retry -- or, for the host-pointer entry
// points, the engine's own -- succeeds.
void EncodecModel::check_async_errors() {
    if (!lstm_timed_out()) return;
    // the launches of the failed call that are still queued could raise the word again after it has been cleared: let them finish first
    // (every side stream of a call joins the handle's stream before the call returns, so this covers them)
    (void)hipStreamSynchronize(stream);
    *reinterpret_cast<volatile unsigned*>(lstm_tmo_host) = 0;
    lstm_force_stepwise = true;
    ++lstm_timeouts;
    fail(NC_EDEVICE, "persistent LSTM kernel: a workgroup exchange timed out (its workgroups were not co-resident); the results of that call are "
                     "invalid -- this handle now runs the step-wise LSTM kernels, repeat the call");
}

// Host-pointer entry points: a timeout left behind by an EARLIER device-pointer call is not this call's failure.  Take note of it
// (step-wise kernels from here on) and carry on; the caller of the earlier call learns of it through nc_codec_check_errors /
// nc_codec_synchronize as documented -- or not at all if it never asked, which is its business.
void EncodecModel::absorb_stale_timeout() {
    if (!lstm_timed_out()) return;
    (void)hipStreamSynchronize(stream);
    *reinterpret_cast<volatile unsigned*>(lstm_tmo_host) = 0;
    lstm_force_stepwise = true;
    ++lstm_timeouts;
}

// ---- launch helpers --------------------------------------------------------------------------------
float* EncodecModel::alloc(size_t n_floats) {
    if (pool_i == pool.size()) pool.emplace_back(new DevBuf());
    pool[pool_i]->reserve(n_floats * sizeof(float));
    return pool[pool_i++]->as<float>();
}

static ActView view_of(const EncodecModel::Act& a) {
    ActView v;
    v.p = a.p; v.rs = a.rs; v.off = a.off; v.stats = a.stats; v.gamma = a.gamma; v.beta = a.beta;
    return v;
}

float* EncodecModel::pad_act(const Act& a, const Act* b2, bool elu, int N, const Plan& pl) {
    float* dst = alloc((size_t)N * a.C * pl.Lp);
    const int64_t n = (int64_t)N * a.C * pl.Lp;
    ActView vb = b2 ? view_of(*b2) : view_of(a);
    (void)n;
    {
        ProfScope ps(&prof, stream, NC_KC_ELEM, 0.0, 4.0 * N * a.C * ((double)a.L * (b2 ? 2 : 1) + (double)pl.Lp));
        hipLaunchKernelGGL(pad_act_kernel, dim3((unsigned)((pl.Lp + 1023) / 1024), (unsigned)a.C, (unsigned)N), dim3(256), 0, stream, view_of(a), vb,
                           b2 ? 1 : 0, elu ? 1 : 0, dst, a.C, a.L, pl.Lz, pl.left, pl.Lp);
    }
    NC_HIP(hipGetLastError());
    return dst;
}

// GroupNorm(1,C) statistics of a raw conv output [N,C,L] (NormConv1d.cs:155).  gn_begin sizes the block-sum buffer and, when the
// producing launch can emit the sums from its epilogue, hands the buffer to it (ConvIO::gn_part); gn_end runs the stand-alone block
// pass otherwise and then the per-sample final: (mean, rstd) pairs the consumer applies while staging its input.
EncodecModel::GnJob EncodecModel::gn_begin(const ConvLayer& conv, ConvIO& io, int N, int C, int64_t L, int sub) {
    GnJob j;
    if (!cfg.time_group_norm) return j;
    j.on = true; j.sub = sub;
    j.nrb = (int)(((int64_t)C * sub + 31) / 32);
    j.ncb = (int)(((L + sub - 1) / sub + 31) / 32);
    j.part = reinterpret_cast<double*>(alloc((size_t)N * j.nrb * j.ncb * 4));
    j.stats = alloc((size_t)N * 2);
    if (conv_gn_fusable(conv, io, N)) {
        io.gn_part = j.part; io.gn_nrb = j.nrb; io.gn_ncb = j.ncb;
        j.fused = true;
        // finish inside the launch: the last workgroup of a sample to arrive writes (mean, rstd).  One self-resetting counter per
        // sample; the segment groups of a call run concurrently, so each has its own set.
        static const bool no_finish = env_flag("NC_NO_GN_FINISH");
        if (!no_finish && N <= GN_MAX_SAMPLES) {
            io.gn_count = gn_counters.as<unsigned>() + (size_t)cur_group * 2 * GN_MAX_SAMPLES;
            io.gn_stats = j.stats;
            io.gn_n = gn_count_arg((double)C * (double)L);
            j.finished = true;
        }
    }
    return j;
}
const float* EncodecModel::gn_end(const GnJob& j, const float* raw, int N, int C, int64_t L, int64_t rs) {
    if (rs <= 0) rs = L;
    if (!j.on) return nullptr;
    if (j.finished) return j.stats;
    const int64_t n = (int64_t)j.nrb * j.ncb;
    ProfScope ps(&prof, stream, NC_KC_NORM, 3.0 * N * C * (double)L, j.fused ? 16.0 * N * (double)n : 4.0 * N * C * (double)L);
    if (!j.fused)
        hipLaunchKernelGGL(gn_block_kernel, dim3((unsigned)(((int64_t)N * j.nrb * ((j.ncb + GN_CBW - 1) / GN_CBW) + 3) / 4)), dim3(256), 0, stream, raw, j.part,
                           (int64_t)N, C, L, j.sub, j.nrb, j.ncb, rs);
    hipLaunchKernelGGL(gn_final_kernel, dim3((unsigned)N), dim3(64), 0, stream, j.part, j.stats, n, (double)C * (double)L);
    NC_HIP(hipGetLastError());
    return j.stats;
}

// SConv1d.forward on an activated view: returns the raw conv output with its pending GroupNorm.
// Single-input layers run with the producer's pending GroupNorm, the ELU and the asymmetric reflect pad folded into the conv's
// tile load (ConvArgs::in_mode): no padded copy of the activation is ever written.  Two-input layers (shortcut + branch of a
// residual block) are summed, activated and padded by pad_act_kernel first.
static void fused_input(ConvIO& io, const EncodecModel::Act& a, bool elu, const EncodecModel::Plan* pl) {
    io.x = a.p + a.off; io.x_bstride = (int64_t)a.C * a.rs; io.x_cstride = a.rs;
    io.in_stats = a.stats; io.in_gamma = a.stats ? a.gamma : nullptr; io.in_beta = a.stats ? a.beta : nullptr;
    io.in_elu = elu;
    if (pl && (pl->left != 0 || pl->Lp != a.L)) {
        io.in_left = pl->left; io.in_Lz = pl->Lz; io.in_L = a.L;
        io.x_len = (int32_t)pl->Lp; io.Tin = pl->Lp;
    } else {
        io.x_len = (int32_t)a.L; io.Tin = a.L;
    }
}

// second operand of a two-input layer (the branch of a residual block beside its shortcut): same geometry, own pending GroupNorm
static void second_input(ConvIO& io, const EncodecModel::Act& b2) {
    io.x2 = b2.p + b2.off;
    io.in_stats2 = b2.stats; io.in_gamma2 = b2.stats ? b2.gamma : nullptr; io.in_beta2 = b2.stats ? b2.beta : nullptr;
}

EncodecModel::Act EncodecModel::sconv(SConv& L, const Act& a, const Act* b2, bool elu, int N) {
    const Plan pl = plan_sconv(a.L, L.K, L.stride, 1);
    static const bool no_fuse = env_flag("NC_ENCODEC_NO_FUSE");
    {   // the stride-2 / stride-4 down-convolutions behind the first two residual blocks: streaming two-input kernels (nc_down2.hip, nc_down4.hip)
        static const bool no_down2 = env_flag("NC_NO_DOWN2");
        static const bool no_down4 = env_flag("NC_NO_DOWN4");
        const int64_t T = a.L;
        const bool common = !no_fuse && b2 && elu && !cfg.causal && !L.transposed && !(L.Cin & 1) && L.Cin <= 128 && b2->C == a.C && b2->L == a.L &&
                            b2->rs == a.rs && (a.stats != nullptr) == (b2->stats != nullptr) && pl.Lz == T && (int64_t)(a.C + 1) * a.rs + T < ((int64_t)1 << 32) &&
                            (!cfg.time_group_norm || (N <= GN_MAX_SAMPLES && !env_flag("NC_NO_GN_FINISH")));
        auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
        const bool s2 = common && !no_down2 && L.K == 4 && L.stride == 2 && L.conv.cfg.TM == 2 && L.conv.cfg.CB == 8 && L.Cout == 64 && T >= 4 && !(T & 1) &&
                        pl.left == 1 && pl.right == 1 && pl.Lout == T / 2;
        const bool s4 = common && !no_down4 && L.K == 8 && L.stride == 4 && L.conv.cfg.TM == 4 && L.conv.cfg.CB == 4 && L.Cout == 128 && L.Cin % 8 == 0 && T >= 8 &&
                        !(T & 3) && pl.left == 2 && pl.right == 2 && pl.Lout == T / 4 && !(a.rs & 3) && al16(a.p + a.off) && al16(b2->p + b2->off);
        static const bool no_down5 = env_flag("NC_NO_DOWN5");
        const bool s5 = common && !no_down5 && L.K == 10 && L.stride == 5 && L.conv.cfg.TM == 4 && L.conv.cfg.CB == 3 && L.Cout == 256 && L.Cin % 4 == 0 && T >= 10 &&
                        T % 5 == 0 && pl.left == 3 && pl.right == 2 && pl.Lout == T / 5;
        if (s2 || s4 || s5) {
            Down2Args d{};
            d.xa = a.p + a.off; d.xb = b2->p + b2->off; d.x_bstride = (int64_t)a.C * a.rs; d.x_cstride = a.rs;
            d.Cin = a.C; d.T = (int)T; d.Tout = (int)pl.Lout;
            d.stats_a = a.stats; d.gamma_a = a.stats ? a.gamma : nullptr; d.beta_a = a.stats ? a.beta : nullptr;
            d.stats_b = b2->stats; d.gamma_b = b2->stats ? b2->gamma : nullptr; d.beta_b = b2->stats ? b2->beta : nullptr;
            d.w = L.conv.w.as<float>(); d.bias = L.conv.has_bias ? L.conv.bias.as<float>() : nullptr;
            float* y = alloc((size_t)N * L.Cout * pl.Lout);
            d.y = y; d.y_bstride = (int64_t)L.Cout * pl.Lout; d.y_cstride = pl.Lout; d.Cout = L.Cout;
            d.B = N; d.n_t_tiles = (int)((pl.Lout + 127) / 128); d.n_cb = (L.Cin + 7) / 8;   // (both kernels walk 8 input channels per barrier; the stride-4 one = two blocks of its CB = 4 image)
            d.n_co_tiles = 1; d.w_co_stride = 0;
            if (s5) {   // four channels per barrier; two row tiles of 128 whose images lie n_cb * KB * BM floats apart
                d.n_cb = L.Cin / 4; d.n_co_tiles = L.Cout / 128;
                d.w_co_stride = (int64_t)((L.Cin + 2) / 3) * 30 * 128;
            }
            float* st = nullptr;
            if (cfg.time_group_norm) {
                d.gn_nrb = L.Cout / 32; d.gn_ncb = (int)((pl.Lout + 31) / 32);
                d.gn_part = reinterpret_cast<double*>(alloc((size_t)N * d.gn_nrb * d.gn_ncb * 4));
                st = alloc((size_t)N * 2);
                d.gn_stats = st;
                d.gn_count = gn_counters.as<unsigned>() + (size_t)cur_group * 2 * GN_MAX_SAMPLES;
                d.gn_n = gn_count_arg((double)L.Cout * (double)pl.Lout);
            }
            auto al8 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 7) == 0; };
            const bool aligned = al8(d.xa) && al8(d.xb) && !(d.x_cstride & 1) && !(d.x_bstride & 1);
            {
                ProfScope ps(&prof, stream, L.conv.kclass, L.conv.flops(N, pl.Lp), 4.0 * N * (2.0 * a.C * (double)T + (double)L.Cout * pl.Lout));
                if (!(s5 ? launch_down5(d, 4, stream) : s4 ? launch_down4(d, 4, stream) : launch_down2(d, 2, aligned, stream)))
                    fail(NC_ESTATE, "internal: no streaming down-convolution instance");
            }
            Act o;
            o.p = y; o.C = L.Cout; o.L = pl.Lout; o.rs = pl.Lout; o.off = 0;
            o.stats = st; o.gamma = st ? L.gamma.as<float>() : nullptr; o.beta = st ? L.beta.as<float>() : nullptr;
            return o;
        }
    }
    float* y = nullptr;
    ConvIO io{};
    const bool two_in = b2 && !no_fuse && conv_in2_available(L.conv) && b2->C == a.C && b2->L == a.L && b2->rs == a.rs &&
                        (a.stats != nullptr) == (b2->stats != nullptr);
    if ((!b2 || two_in) && !no_fuse && pl.Lp < ((int64_t)1 << 30)) {
        fused_input(io, a, elu, &pl);
        if (two_in) second_input(io, *b2);
        y = alloc((size_t)N * L.Cout * pl.Lout);
    } else {
        const bool passthrough = !b2 && !elu && !a.stats && pl.left == 0 && pl.Lp == a.L && a.off == 0 && a.rs == a.L;
        const float* xin = passthrough ? a.p : pad_act(a, b2, elu, N, pl);
        y = alloc((size_t)N * L.Cout * pl.Lout);
        io.x = xin; io.x_bstride = (int64_t)L.Cin * pl.Lp; io.x_cstride = pl.Lp; io.x_len = (int32_t)pl.Lp; io.Tin = pl.Lp;
    }
    io.y = y; io.y_bstride = (int64_t)L.Cout * pl.Lout; io.y_cstride = pl.Lout;
    const GnJob gj = gn_begin(L.conv, io, N, L.Cout, pl.Lout, 1);
    launch_conv(L.conv, io, N, stream, &prof);
    Act o;
    o.p = y; o.C = L.Cout; o.L = pl.Lout; o.rs = pl.Lout; o.off = 0;
    o.stats = gn_end(gj, y, N, L.Cout, pl.Lout);
    o.gamma = o.stats ? L.gamma.as<float>() : nullptr;
    o.beta = o.stats ? L.beta.as<float>() : nullptr;
    return o;
}

// SConvTranspose1d.forward (SConvTranspose1d.cs:116-139): conv-transpose, GroupNorm over the UNTRIMMED output, then the trim
EncodecModel::Act EncodecModel::sconvT(SConv& L, const Act& a, const Act* b2, bool elu, int N) {
    static const bool no_fuse = env_flag("NC_ENCODEC_NO_FUSE");
    const int64_t Lfull = (a.L - 1) * L.stride + L.K;
    {   // the stride-2 / stride-4 up-convolutions in front of the last two residual blocks: streaming two-input kernel (nc_up2.hip)
        static const bool no_up2 = env_flag("NC_NO_UP2");
        static const bool no_up4 = env_flag("NC_NO_UP4");
        const int64_t T = a.L;
        const int S = L.stride;
        const bool common = !no_fuse && b2 && !cfg.causal && L.transposed && L.K == 2 * S && L.conv.sub_stride == S && L.conv.n_phase == 1 && L.conv.cfg.CB == 16 &&
                            !(L.Cin & 1) && L.Cin <= 128 && b2->C == a.C && b2->L == a.L && b2->rs == a.rs && (a.stats != nullptr) == (b2->stats != nullptr) &&
                            T >= 4 && !(T & 1) && (int64_t)(a.C + 1) * a.rs + T < ((int64_t)1 << 32) && (int64_t)L.Cout * Lfull < ((int64_t)1 << 31) &&
                            (!cfg.time_group_norm || (N <= GN_MAX_SAMPLES && !env_flag("NC_NO_GN_FINISH")));
        const bool u2 = common && !no_up2 && S == 2 && L.conv.cfg.TM == 2 && L.Cout == 32;
        const bool u4 = common && !no_up4 && S == 4 && L.conv.cfg.TM == 4 && L.Cout == 64;
        if (u2 || u4) {
            Up2Args d{};
            d.xa = a.p + a.off; d.xb = b2->p + b2->off; d.x_bstride = (int64_t)a.C * a.rs; d.x_cstride = a.rs;
            d.Cin = a.C; d.L = (int)T; d.elu = elu ? 1 : 0;
            d.stats_a = a.stats; d.gamma_a = a.stats ? a.gamma : nullptr; d.beta_a = a.stats ? a.beta : nullptr;
            d.stats_b = b2->stats; d.gamma_b = b2->stats ? b2->gamma : nullptr; d.beta_b = b2->stats ? b2->beta : nullptr;
            d.w = L.conv.w.as<float>(); d.bias = L.conv.has_bias ? L.conv.bias.as<float>() : nullptr;
            float* y = alloc((size_t)N * L.Cout * Lfull);
            d.y = y; d.y_bstride = (int64_t)L.Cout * Lfull; d.y_cstride = Lfull; d.Cout = L.Cout;
            d.B = N; d.n_t_tiles = (int)((T + 1 + 255) / 256); d.n_cb = (L.Cin + 15) / 16; d.n_co_tiles = S * L.Cout / (32 * L.conv.cfg.TM);
            float* st = nullptr;
            if (cfg.time_group_norm) {
                d.gn_nrb = S * L.Cout / 32; d.gn_ncb = (int)(((Lfull + S - 1) / S + 31) / 32);
                d.gn_part = reinterpret_cast<double*>(alloc((size_t)N * d.gn_nrb * d.gn_ncb * 4));
                st = alloc((size_t)N * 2);
                d.gn_stats = st;
                d.gn_count = gn_counters.as<unsigned>() + (size_t)cur_group * 2 * GN_MAX_SAMPLES;
                d.gn_n = gn_count_arg((double)L.Cout * (double)Lfull);
            }
            auto al8 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 7) == 0; };
            const bool aligned = al8(d.xa) && al8(d.xb) && !(d.x_cstride & 1) && !(d.x_bstride & 1);
            {
                ProfScope ps(&prof, stream, L.conv.kclass, L.conv.flops(N, T), 4.0 * N * (2.0 * a.C * (double)T + (double)L.Cout * Lfull));
                if (!launch_up2(d, L.conv.cfg.TM, S, aligned, stream)) fail(NC_ESTATE, "internal: no streaming up-convolution instance");
            }
            const int64_t pt = L.K - L.stride, right = pt / 2, left = pt - right;           // non-causal trim (SConvTranspose1d.cs:159-171)
            Act o;
            o.p = y; o.C = L.Cout; o.L = Lfull - left - right; o.rs = Lfull; o.off = left;
            o.stats = st; o.gamma = st ? L.gamma.as<float>() : nullptr; o.beta = st ? L.beta.as<float>() : nullptr;
            return o;
        }
    }
    ConvIO io{};
    const bool two_in = b2 && !no_fuse && conv_in2_available(L.conv) && b2->C == a.C && b2->L == a.L && b2->rs == a.rs &&
                        (a.stats != nullptr) == (b2->stats != nullptr);
    if ((!b2 || two_in) && !no_fuse) {
        fused_input(io, a, elu, nullptr);
        if (two_in) second_input(io, *b2);
    } else {
        Plan pl; pl.left = 0; pl.right = 0; pl.Lz = a.L; pl.Lp = a.L; pl.Lout = a.L;
        const float* xin = pad_act(a, b2, elu, N, pl);
        io.x = xin; io.x_bstride = (int64_t)L.Cin * a.L; io.x_cstride = a.L; io.x_len = (int32_t)a.L; io.Tin = a.L;
    }
    const int64_t pt = L.K - L.stride;
    int64_t right, left;
    if (cfg.causal) { right = pt; left = 0; }                                               // trimRightRatio = 1
    else { right = pt / 2; left = pt - right; }
    // The consumers read the TRIMMED view (rows start `left` samples in).  Where that start is not 8-byte aligned -- the stride-5 layer: left = 3,
    // odd row length 6005 -- the residual block behind it lost its aligned kernels (the C = 128 shortcut ran on the windowed template: 181 us
    // against 71 us for the same layer in the encoder).  Rows are therefore laid out at a pitch of whole 16 bytes and the buffer starts
    // `shift` samples in, so that every row of the trimmed view begins on a 16-byte boundary (NC_NO_UP_PITCH=1: dense rows, no shift).
    static const bool no_pitch = env_flag("NC_NO_UP_PITCH");
    const int64_t P = no_pitch ? Lfull : ((Lfull + 3) & ~(int64_t)3);
    const int64_t shift = no_pitch ? 0 : (4 - left % 4) % 4;
    float* y = alloc((size_t)N * L.Cout * P + 4) + shift;
    io.y = y; io.y_bstride = (int64_t)L.Cout * P; io.y_cstride = P;
    // (the block view of the statistics follows the layer geometry, not the kernel form: the same sums under NC_NO_SUBPIXEL)
    const GnJob gj = gn_begin(L.conv, io, N, L.Cout, Lfull, conv_gn_sub(L.K, L.stride, L.Cout, true));
    launch_conv(L.conv, io, N, stream, &prof);
    Act o;
    o.p = y; o.C = L.Cout; o.L = Lfull - left - right; o.rs = P; o.off = left;
    o.stats = gn_end(gj, y, N, L.Cout, Lfull, P);
    o.gamma = o.stats ? L.gamma.as<float>() : nullptr;
    o.beta = o.stats ? L.beta.as<float>() : nullptr;
    return o;
}

// The first pass of a residual block as ONE launch (nc_resa.hip): s = shortcut(x) and h = conv3(elu(x)) from a single read of x -- the
// thin outer stages of the 48 kHz model (C = 32 / 64: HBM-bound; the block input was read twice, the k = 3 launch alone ran at 2.1 TB/s).
// Same arithmetic as the two launches (NC_NO_RES_A=1 runs those).
bool EncodecModel::resblock_first_pass(ResBlock& r, const Act& x, int N, Act& s, Act& h) {
    static const bool off = env_flag("NC_NO_RES_A") || env_flag("NC_ENCODEC_NO_FUSE");
    const int C = x.C;
    const int64_t T = x.L;
    if (off || cfg.causal || (C != 32 && C != 64) || r.sc.K != 1 || r.c1.K != 3 || r.sc.Cin != C || r.sc.Cout != C || r.c1.Cin != C || r.c1.Cout != C / 2) return false;
    if (r.sc.conv.cfg.TM != C / 32 || r.sc.conv.cfg.CB != 16 || r.c1.conv.cfg.TM != 1 || r.c1.conv.cfg.CB != 16) return false;   // one row tile each
    if (T < 4 || (T & 1) || T >= ((int64_t)1 << 30) || (int64_t)(C + 1) * x.rs + T >= ((int64_t)1 << 32)) return false;
    const Plan pl = plan_sconv(T, 3, 1, 1);
    if (pl.left != 1 || pl.Lz != T || pl.Lp != T + 2 || pl.Lout != T) return false;   // (the non-causal reflect pad 1 + 1 the kernel folds into its lanes)
    const bool gn = cfg.time_group_norm;
    if (gn && (N > GN_MAX_SAMPLES || env_flag("NC_NO_GN_FINISH"))) return false;
    float* ys = alloc((size_t)N * C * T);
    float* yb = alloc((size_t)N * (C / 2) * T);
    ResAArgs a{};
    a.x = x.p + x.off; a.x_bstride = (int64_t)x.C * x.rs; a.x_cstride = x.rs; a.Cin = C; a.T = (int)T;
    a.in_stats = x.stats; a.in_gamma = x.stats ? x.gamma : nullptr; a.in_beta = x.stats ? x.beta : nullptr;
    a.w_s = r.sc.conv.w.as<float>(); a.bias_s = r.sc.conv.has_bias ? r.sc.conv.bias.as<float>() : nullptr;
    a.ys = ys; a.ys_bstride = (int64_t)C * T; a.ys_cstride = T; a.Cs = C;
    a.w_b = r.c1.conv.w.as<float>(); a.bias_b = r.c1.conv.has_bias ? r.c1.conv.bias.as<float>() : nullptr;
    a.yb = yb; a.yb_bstride = (int64_t)(C / 2) * T; a.yb_cstride = T; a.Cb = C / 2;
    a.B = N; a.n_t_tiles = (int)((T + 255) / 256); a.n_cb = C / 16;
    float* st_s = nullptr; float* st_b = nullptr;
    if (gn) {
        a.gn_nrb_s = C / 32; a.gn_nrb_b = 1; a.gn_ncb = (int)((T + 31) / 32);
        a.gn_part_s = reinterpret_cast<double*>(alloc((size_t)N * a.gn_nrb_s * a.gn_ncb * 4));
        a.gn_part_b = reinterpret_cast<double*>(alloc((size_t)N * a.gn_nrb_b * a.gn_ncb * 4));
        st_s = alloc((size_t)N * 2); st_b = alloc((size_t)N * 2);
        a.gn_stats_s = st_s; a.gn_stats_b = st_b;
        a.gn_count_s = gn_counters.as<unsigned>() + (size_t)cur_group * 2 * GN_MAX_SAMPLES;
        a.gn_count_b = a.gn_count_s + GN_MAX_SAMPLES;
        a.gn_n_s = gn_count_arg((double)C * (double)T); a.gn_n_b = gn_count_arg((double)(C / 2) * (double)T);
    }
    auto al8 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 7) == 0; };
    const bool aligned = al8(a.x) && !(a.x_cstride & 1) && !(a.x_bstride & 1);
    {
        // (one class for the launch: the pointwise one -- 2/5 of its flops, all of its input bytes)
        ProfScope ps(&prof, stream, NC_KC_CONV_K1, 2.0 * C * (C + 1.5 * C) * (double)T * N, 4.0 * N * (double)T * (C + C + C / 2));
        if (!launch_res_a(a, C / 32, aligned, stream)) fail(NC_ESTATE, "internal: no first-pass kernel for C = %d", C);
    }
    s.p = ys; s.C = C; s.L = T; s.rs = T; s.off = 0; s.stats = st_s; s.gamma = st_s ? r.sc.gamma.as<float>() : nullptr; s.beta = st_s ? r.sc.beta.as<float>() : nullptr;
    h.p = yb; h.C = C / 2; h.L = T; h.rs = T; h.off = 0; h.stats = st_b; h.gamma = st_b ? r.c1.gamma.as<float>() : nullptr; h.beta = st_b ? r.c1.beta.as<float>() : nullptr;
    return true;
}

// SEANetResnetBlock.forward (:72-85): shortcut(x) + conv1(elu(conv3(elu(x)))) -> the two pending views (s, y)
void EncodecModel::resblock(ResBlock& r, const Act& x, int N, Act& s, Act& y) {
    Act h;
    if (!resblock_first_pass(r, x, N, s, h)) {
        s = sconv(r.sc, x, nullptr, false, N);
        h = sconv(r.c1, x, nullptr, true, N);
    }
    y = sconv(r.c2, h, nullptr, true, N);
    // A row shorter than the k=3 pad takes SConv1d's small-input path (zero-extend, never trimmed: D9), so the block branch comes
    // out LONGER than the 1x1 shortcut and the reference's add() would broadcast.  Such degenerate segments are rejected.
    if (y.L != s.L) fail(NC_EINVAL, "segment too short: a residual block sees %lld samples", (long long)x.L);
}

// GroupNorm-apply of a view into a dense tensor (optionally x scale[b] / divided by scale[b])
float* EncodecModel::materialize(const Act& a, int N, const float* scale, int mode) {
    float* y = alloc((size_t)N * a.C * a.L);
    const int64_t n = (int64_t)N * a.C * a.L;
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, view_of(a), scale, mode, y, N, a.C, a.L);
    NC_HIP(hipGetLastError());
    return y;
}

// [N][C][T] -> [C][T][N]: the layout in which a range of steps is one contiguous column range (run_lstm's chunked input projections)
__global__ void nct_to_ctn_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C, int64_t T) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)N * C * T) return;
    const int n = (int)(i % N);
    const int64_t r = i / N, t = r % T, c = r / T;
    y[i] = x[((int64_t)n * C + c) * T + t];
}

// true when run_lstm should apply the consumer's ELU in its output store (NC_LSTM_NO_ELU=1: the consumer applies it while staging)
static bool lstm_applies_elu(const EncodecModel::Lstm& l) {
    static const bool off = env_flag("NC_LSTM_NO_ELU");
    return !off && !l.layers.empty();
}

// SLSTM.forward (SLSTM.cs:40-57) on a dense x [N,C,T]; returns lstm(x) + x (elu_out: ELU of it)
float* EncodecModel::run_lstm(Lstm& l, const float* x, int N, int64_t T, bool elu_out) {
    const int C = l.C;
    if (l.layers.empty()) return const_cast<float*>(x);
    static const bool stepwise_env = env_flag("NC_LSTM_STEPWISE");
    const int KS = C / 4;
    const int nl = (int)l.layers.size();
    // The persistent kernel needs every workgroup of a launch resident at once (one per CU at 140 KB of LDS): up to 64 per launch, two
    // launches in flight when the layers are pipelined, beside a concurrent segment group.  Devices (or partitions) that cannot hold
    // that, and handles that have seen a timeout, take the step-wise kernels.
    const size_t lds_need = (size_t)4 * KS * 64 * 4 + 3 * 4 * 64 * 16;
    const bool stepwise = stepwise_env || lstm_force_stepwise || cu_count < 192 || lds_per_cu < lds_need + 1024;
    auto ih_gemm = [&](LstmLayer& y, const float* in, float* gi, hipStream_t s) {
        ConvIO io{};   // W_ih * x_t + b_ih for all steps: one pointwise convolution over the [N,C,T] tensor
        io.x = in; io.x_bstride = (int64_t)C * T; io.x_cstride = T; io.x_len = (int32_t)T; io.Tin = T;
        io.y = gi; io.y_bstride = (int64_t)4 * C * T; io.y_cstride = T;
        launch_conv(y.ih, io, N, s, &prof);
    };
#ifdef NC_EXPERIMENTS   // measured-and-rejected: `make EXPERIMENTS=1` compiles nc_lstm.hip and these two branches
    // Fused two-layer launch (nc_lstm.hip, NC_LSTM_FUSED=1): every step of both layers in ONE persistent launch per pair of column tiles
    // -- no drain in the exchange (values validated against a sentinel), chain wavefronts that never store, no tensors between the
    // layers, no chunked projection GEMMs.  Bit-exact, but MEASURED SLOWER than the per-layer kernels below on two tiles (C3 9.5-9.8
    // against 9.05 ms; 16 clips at 24 kHz 5.4-5.9 against 5.67 ms): with C / 4 = 128 workgroups in every exchange a step is 9.8-12 us
    // (in-kernel trace, tools/probe/lstm2_trace.py: publish -> flags seen 2.6-3.4 us, operand loads 1.4-2.2 us, and the workgroups
    // drift 3.5-5 us apart inside a step) against 5.7-6.2 us for an exchange among 32.  Kept as a switch, not the default: DESIGN 8.
    static const bool fused_env = env_flag("NC_LSTM_FUSED");
    const bool per_layer_env = !fused_env;
    const int n_tiles2 = (N + 15) / 16;
    const size_t ex_floats = lstm2_exchange_floats(C, T, std::min(n_tiles2, 2));
    if (!stepwise && !per_layer_env && nl == 2 && lstm2_supported(C) && l.layers[1]->w2ih.p && cu_count >= C / 4 &&
        lds_per_cu >= lstm2_lds_bytes(C, std::min(n_tiles2, 2)) && ex_floats * 4 < ((size_t)1 << 31) && (int64_t)4 * C * T * N < ((int64_t)1 << 31)) {
        LstmSection section(*this, stream);
        {   // NC_LSTM_FAKE_TIMEOUT=1 (tests): the first persistent launch of the process is reported as timed out
            static bool fake = env_flag("NC_LSTM_FAKE_TIMEOUT");
            if (fake) { fake = false; *reinterpret_cast<volatile unsigned*>(lstm_tmo_host) = 1; }
        }
        // layer 0's input projections of ALL steps as one pointwise GEMM over the [C][T][N] view of x: gi0 [4C][T][N]
        float* xT = alloc((size_t)N * C * T);
        float* gi0 = alloc((size_t)N * 4 * C * T);
        float* out = alloc((size_t)N * C * T);
        {
            const int64_t n = (int64_t)N * C * T;
            hipLaunchKernelGGL(nct_to_ctn_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, xT, N, C, T);
            NC_HIP(hipGetLastError());
            ConvIO io{};
            io.x = xT; io.x_bstride = 0; io.x_cstride = T * N; io.x_len = (int32_t)(T * N); io.Tin = T * N;
            io.y = gi0; io.y_bstride = 0; io.y_cstride = T * N;
            launch_conv(l.layers[0]->ih, io, 1, stream, &prof);
        }
        for (int tl = 0; tl < n_tiles2; tl += 2) {
            const int nt = std::min(2, n_tiles2 - tl);
            const size_t exf = lstm2_exchange_floats(C, T, nt);
            float* S = alloc(exf);
            unsigned* flags = reinterpret_cast<unsigned*>(alloc((size_t)nt * 2 * (C / 4)));
            NC_HIP(hipMemsetAsync(S, 0xFF, exf * 4, stream));                       // LSTM2_SENTINEL in every word
            NC_HIP(hipMemsetAsync(flags, 0, (size_t)nt * 2 * (C / 4) * 4, stream));
            Lstm2Args a{};
            a.gi0 = gi0; a.whh0 = l.layers[0]->w2hh.as<float>(); a.wih1 = l.layers[1]->w2ih.as<float>(); a.whh1 = l.layers[1]->w2hh.as<float>();
            a.bhh0 = l.layers[0]->bhh.as<float>(); a.bih1 = l.layers[1]->bih.as<float>(); a.bhh1 = l.layers[1]->bhh.as<float>();
            a.skip = x; a.out = out; a.elu_out = elu_out ? 1 : 0; a.S = S; a.flags = flags; a.tmo = lstm_tmo_dev;
            a.N = N; a.C = C; a.T = T; a.tile0 = tl; a.tiles = nt;
            const double nn = (double)std::min(N - tl * 16, nt * 16) * (double)T;
            if (prof.on) prof.begin(stream, NC_KC_LSTM, 2.0 * 3 * 4 * C * C * nn, 4.0 * 8 * C * nn);   // W_hh0, W_ih1, W_hh1 contractions
            static const char* trace_path = env_str("NC_LSTM2_TRACE");   // diagnostic: stamps of 8 steps of the first full-size launch
            static bool traced = false;
            const size_t trace_words = (size_t)(C / 4) * nt * 8 * LSTM2_TRACE_STEPS * 4 + 4;   // + the clock probe
            if (trace_path && !traced && T >= LSTM2_TRACE_T0 + LSTM2_TRACE_STEPS) {
                a.trace = reinterpret_cast<unsigned long long*>(alloc(trace_words * 2));
                NC_HIP(hipMemsetAsync(a.trace, 0, trace_words * 8, stream));
            }
            lstm2_launch(a, stream);
            if (prof.on) prof.end(stream);
            if (a.trace) {
                traced = true;
                std::vector<unsigned long long> hbuf(trace_words);
                NC_HIP(hipStreamSynchronize(stream));
                NC_HIP(hipMemcpy(hbuf.data(), a.trace, trace_words * 8, hipMemcpyDeviceToHost));
                if (FILE* f = std::fopen(trace_path, "wb")) {
                    const int hdr[4] = {C / 4, nt, 8, LSTM2_TRACE_STEPS};
                    std::fwrite(hdr, sizeof(int), 4, f);
                    std::fwrite(hbuf.data(), 8, trace_words, f);
                    std::fclose(f);
                }
            }
        }
        return out;
    }
#endif  // NC_EXPERIMENTS
    if (!stepwise && C % 64 == 0 && (KS == 128 || KS == 16)) {
        LstmSection section(*this, stream);   // (see LstmTicket: sections of different handles on one device run one after the other)
        // Persistent layer kernel: a launch runs a range of steps for a group of column tiles (<= 64 co-resident workgroups, so the
        // two layers of a pipelined call plus a concurrent segment group still fit the chip's 256 CUs at one workgroup per CU).
        // Layer pipelining: layer l+1 at step t needs only h^l_t, so the sequence is cut into chunks and chunk k of layer l+1 (with its
        // input-projection GEMM) runs on a second stream while layer l runs chunk k+1: the dependent chain of a 2-layer LSTM shrinks
        // from 2T steps to about T + T/chunks.  The arithmetic is untouched -- the same kernel, resumed from carried (h, c) state.
        // The tensors BETWEEN the layers (h^l and the input projections of layer l+1) are laid out [rows][T][N]: the steps of a chunk
        // are then one contiguous column range of a 2-D matrix, so the chunk's GEMM is a one-"clip" pointwise convolution over
        // chunk*N columns with full 128-column tiles (cut out of [N,C,T], a chunk would fill a fifth of every tile: measured, the
        // per-chunk GEMMs then cost as much as the full one and 6 chunks made C3 2.2 ms slower).
        // (round 6: 6 chunks, was 4 -- re-swept after the projection GEMMs moved to the pointwise kernel: C3 8.26 -> 8.21 ms, Encodec 24 kHz x 16 clips
        //  5.51 -> 5.26 ms; 8 chunks: back to the 4-chunk times.  tools/probe/r6_chunks24.sh)
        static const int want_chunks = [] { const int v = (int)env_int("NC_LSTM_CHUNKS", 6); return v < 1 ? 1 : v; }();
        // chunk boundaries (even: chunk starts stay 8-byte aligned for the 1x1 path).  The layer above trails the layer below by its
        // LAST chunk (+ that chunk's input-projection GEMM), so the last chunk is short (T/8) and the others share the rest: with 4
        // chunks of 150 steps 44 / 44 / 44 / 18 instead of 38 / 38 / 38 / 36 (6 chunks: 28 / 28 / 28 / 28 / 20 / 18) -- the tail after layer 0 has finished shrinks from 36
        // steps to 18 without a single extra cross-stream event.
        std::vector<int64_t> cstart{0};
        if (nl >= 2 && want_chunks > 1 && T >= 32 && !on_side_group && (int64_t)4 * C * T * N < ((int64_t)1 << 31)) {
            static const bool even_chunks = env_flag("NC_LSTM_EVEN_CHUNKS");
            const int64_t last = even_chunks ? 0 : std::max<int64_t>(8, (T / 8) & ~(int64_t)1);
            const int nbig = even_chunks ? want_chunks : want_chunks - 1;
            int64_t big = ((((T - last) + nbig - 1) / nbig) + 1) & ~(int64_t)1;
            big = std::max<int64_t>(big, 8);
            for (int64_t t0 = big; t0 < ((T - last) & ~(int64_t)1); t0 += big) cstart.push_back(t0);
            const int64_t tail0 = (T - last) & ~(int64_t)1;   // an even start whatever the parity of T: the last chunk absorbs the odd step
            if (last > 0 && tail0 > cstart.back()) cstart.push_back(tail0);
        }
        cstart.push_back(T);
        const int nch = (int)cstart.size() - 1;
        const bool piped = nch > 1;
        // Unit blocks per workgroup: 2 at C = 512 with ONE column tile (64 workgroups of 8 wavefronts, 72 KB of LDS each -- the
        // matrix-core part of a step is 2 chains per SIMD instead of 4: 1.8 -> 0.9 us of the ~5.7; Encodec 24 kHz at 16 clips
        // 6.24 -> 6.13 ms), 4 otherwise: with two tiles the 2 x 64-producer exchange costs more than the chains save (C3 9.59 -> 9.70 ms).
        // NC_LSTM_UB=4 / 2 force one form.
        static const int ub_env = (int)env_int("NC_LSTM_UB", 0);
        const int n_tiles = (N + 15) / 16;
        const int UBW = (KS == 128 && (ub_env == 2 || (ub_env != 4 && n_tiles == 1))) ? 2 : 4;
        const int nprod = C / (4 * UBW), per_launch = std::max(1, (UBW == 2 ? 128 : 64) / nprod);
        const size_t lds = (size_t)4 * UBW * (KS / 4) * 64 * 4 + 3 * UBW * 64 * 16;
        unsigned* sync = lstm_tmo_dev;                                                 // timeout word (host-visible)
        {   // NC_LSTM_FAKE_TIMEOUT=1 (tests): the first persistent launch of the process is reported as timed out
            static bool fake = env_flag("NC_LSTM_FAKE_TIMEOUT");
            if (fake) { fake = false; *reinterpret_cast<volatile unsigned*>(lstm_tmo_host) = 1; }
        }
        // Role-split kernel (nc_lstm.hip lstm1_kernel, round 4, NC_LSTM_SPLIT=1): the same launches and chunk schedule, but load-only chain
        // waves, store-only gate waves and exchange regions that are never reused and validated by value -- no drain.  Bit-exact and
        // MEASURED SLOWER than lstm_seq_kernel (C3 10.1-10.2 against 9.07-9.15 ms on the same box, 16 clips at 24 kHz 6.75-6.9 against 5.6),
        // with the flags polled by a gate wave and by a (load-only) chain wave alike: without the drain the hint flags run ahead of
        // the payload, so operand loads are retried, and the LDS post / wait hops between chain, gate and polling waves cost more than
        // the two workgroup barriers they replace.  Not the default; DESIGN 8 round 4.
#ifdef NC_EXPERIMENTS
        static const bool want_split = env_flag("NC_LSTM_SPLIT");
        const bool split = want_split && lstm2_supported(C) && l.layers[0]->w2hh.p && lds_per_cu >= lstm1_lds_bytes(C) &&
                           (size_t)T * n_tiles * C * 16 * 4 < ((size_t)1 << 31);
#else
        constexpr bool split = false;
#endif
        std::vector<float*> gi(nl), out(nl), hx(nl), cs(nl);
        std::vector<unsigned*> flags(nl);
        for (int li = 0; li < nl; ++li) {
            gi[li] = alloc((size_t)N * 4 * C * T);
            out[li] = alloc((size_t)N * C * T);
            cs[li] = alloc((size_t)n_tiles * C * 16);
            if (split) {
                hx[li] = alloc((size_t)T * n_tiles * C * 16);                          // one exchange region per step
                NC_HIP(hipMemsetAsync(hx[li], 0xFF, (size_t)T * n_tiles * C * 16 * 4, stream));   // LSTM2_SENTINEL in every word
                flags[li] = reinterpret_cast<unsigned*>(alloc((size_t)n_tiles * (C / 4)));
                NC_HIP(hipMemsetAsync(flags[li], 0, (size_t)n_tiles * (C / 4) * 4, stream));
            } else {
                hx[li] = alloc((size_t)2 * n_tiles * C * 16);
                flags[li] = reinterpret_cast<unsigned*>(alloc((size_t)n_tiles * nprod));   // per call + layer: groups may run concurrently
                NC_HIP(hipMemsetAsync(flags[li], 0, (size_t)n_tiles * nprod * 4, stream));
            }
        }
        hipStream_t sA = stream, sB = stream;
        size_t ev_i = 0;
        auto next_event = [&]() {
            if (ev_i == lstm_events.size()) {
                hipEvent_t e;
                NC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                lstm_events.push_back(e);
            }
            return lstm_events[ev_i++];
        };
        auto layer_stream = [&](int li) { return (li & 1) ? sB : sA; };
        // tensor of layer li that meets the convolution stack ([N][rows][T]) or sits between two pipelined layers ([rows][T][N])
        auto between = [&](int li_out) { return piped && li_out + 1 < nl; };
        auto lstm_chunk = [&](int li, int64_t t0, int64_t t1) {
            LstmLayer& y = *l.layers[li];
            hipStream_t s = layer_stream(li);
            const bool last = li + 1 == nl;
            const double n = (double)N * (double)(t1 - t0);
            if (prof.on) prof.begin(s, NC_KC_LSTM, 2.0 * 4 * C * C * n, 4.0 * 6 * C * n);
#ifdef NC_EXPERIMENTS
            for (int tl = 0; split && tl < n_tiles; tl += 4) {                         // (C / 16 workgroups per tile: up to four tiles per launch)
                const int nt = std::min(4, n_tiles - tl);
                LstmSplitArgs a{};
                a.gi = gi[li]; a.w = y.w2hh.as<float>(); a.bhh = y.bhh.as<float>(); a.skip = last ? x : nullptr; a.out = out[li]; a.elu_out = (last && elu_out) ? 1 : 0;
                if (piped) { a.gi_b = 1; a.gi_c = T * N; a.gi_t = N; }
                else { a.gi_b = (int64_t)4 * C * T; a.gi_c = T; a.gi_t = 1; }
                if (between(li)) { a.out_b = 1; a.out_c = T * N; a.out_t = N; }
                else { a.out_b = (int64_t)C * T; a.out_c = T; a.out_t = 1; }
                a.S = hx[li]; a.flags = flags[li]; a.tmo = sync; a.cstate = cs[li];
                a.N = N; a.C = C; a.T = T; a.t0 = t0; a.t1 = t1; a.tile0 = tl; a.tiles_total = n_tiles;
                lstm1_launch(a, nt, s);
            }
#endif
            for (int tl = 0; !split && tl < n_tiles; tl += per_launch) {
                const int nt = std::min(per_launch, n_tiles - tl);
                LstmSeqArgs a{};
                a.gi = gi[li]; a.whhp = y.whhp.as<float>(); a.bhh = y.bhh.as<float>(); a.skip = last ? x : nullptr; a.out = out[li]; a.elu_out = (last && elu_out) ? 1 : 0;
                if (piped) { a.gi_b = 1; a.gi_c = T * N; a.gi_t = N; }
                else { a.gi_b = (int64_t)4 * C * T; a.gi_c = T; a.gi_t = 1; }
                if (between(li)) { a.out_b = 1; a.out_c = T * N; a.out_t = N; }
                else { a.out_b = (int64_t)C * T; a.out_c = T; a.out_t = 1; }
                a.hx = hx[li] + (size_t)2 * tl * C * 16;
                a.cstate = cs[li] + (size_t)tl * C * 16;
                a.flags = flags[li] + (size_t)tl * nprod; a.tmo = sync;
                a.N = N; a.C = C; a.T = T; a.t0 = t0; a.t1 = t1; a.tile0 = tl;
                static const bool sync_acquire = env_flag("NC_SYNC_ACQUIRE");
                a.acquire = sync_acquire ? 1 : 0;
                auto launch = [&](auto kern) {
                    ensure_dynamic_lds((const void*)kern, lds);
                    hipLaunchKernelGGL(kern, dim3((unsigned)nprod, (unsigned)nt), dim3(256 * UBW), lds, s, a);
                };
                static const bool no_htile = env_flag("NC_LSTM_NO_HTILE");   // every wave fetches its own h operands, W_hh in LDS (the round-2..5 form)
                if (no_htile) {
                    if (KS == 128 && UBW == 2) launch(lstm_seq_kernel<128, 2, false>);
                    else if (KS == 128) launch(lstm_seq_kernel<128, 4, false>);
                    else launch(lstm_seq_kernel<16, 4, false>);
                } else {
                    if (KS == 128 && UBW == 2) launch(lstm_seq_kernel<128, 2, true>);
                    else if (KS == 128) launch(lstm_seq_kernel<128, 4, true>);
                    else launch(lstm_seq_kernel<16, 4, true>);
                }
            }
            NC_HIP(hipGetLastError());
            if (prof.on) prof.end(s);
        };
        if (!piped) {
            for (int li = 0; li < nl; ++li) {
                ih_gemm(*l.layers[li], li ? out[li - 1] : x, gi[li], stream);
                lstm_chunk(li, 0, T);
            }
            return out[nl - 1];
        }
        if (!lstm_stream) NC_HIP(hipStreamCreateWithFlags(&lstm_stream, hipStreamNonBlocking));
        sB = lstm_stream;
        {
            hipEvent_t fork = next_event();
            NC_HIP(hipEventRecord(fork, sA));
            NC_HIP(hipStreamWaitEvent(sB, fork, 0));
        }
        // the input projections of steps [t0, t0+n) of a layer above the first: columns [t0*N, (t0+n)*N) of the [rows][T*N] matrices
        auto ih_gemm_chunk = [&](LstmLayer& y, const float* in, float* g, int64_t t0, int64_t n, hipStream_t s) {
            ConvIO io{};
            io.x = in + t0 * N; io.x_bstride = 0; io.x_cstride = T * N; io.x_len = (int32_t)(n * N); io.Tin = n * N;
            io.y = g + t0 * N; io.y_bstride = 0; io.y_cstride = T * N;
            launch_conv(y.ih, io, 1, s, &prof);
        };
        // Layer 0's input projections depend on x alone: x goes to the [C][T][N] layout once and its chunk GEMMs run ahead on the second
        // stream, chunk k releasing layer 0's chunk k (the recurrence starts after the first chunk's GEMM, not after the whole one).
        // The projection GEMM of chunk k of a layer ABOVE runs on the stream of the layer BELOW, right behind that layer's chunk k:
        // the layer above then runs its chunks back to back (on its own stream, GEMM and chunk alternating, it was the sum of both --
        // ~1.0 ms for 0.78 ms of recurrence -- and bounded the whole LSTM), while the layer below, which finishes a short last chunk
        // ahead anyway, absorbs the ~40 us per GEMM.  (A third stream for the GEMMs measured WORSE: the runtime multiplexes streams onto
        // 4 hardware queues, the extra stream shared a queue with the tail-segment group or the other layer and serialised with it,
        // +0.9 ms; with GPU_MAX_HW_QUEUES=8 every queue got slower gaps, +1.8 ms.)
        std::vector<hipEvent_t> ready(nch, nullptr);   // ready[k]: the input projections of chunk k of the current layer are complete
        {
            float* xT = alloc((size_t)N * C * T);
            const int64_t n = (int64_t)N * C * T;
            hipLaunchKernelGGL(nct_to_ctn_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, sB, x, xT, N, C, T);
            NC_HIP(hipGetLastError());
            for (int k = 0; k < nch; ++k) {
                ih_gemm_chunk(*l.layers[0], xT, gi[0], cstart[(size_t)k], cstart[(size_t)k + 1] - cstart[(size_t)k], sB);
                ready[k] = next_event();
                NC_HIP(hipEventRecord(ready[k], sB));
            }
        }
        for (int li = 0; li < nl; ++li) {
            hipStream_t s = layer_stream(li);
            for (int k = 0; k < nch; ++k) {
                NC_HIP(hipStreamWaitEvent(s, ready[k], 0));
                lstm_chunk(li, cstart[(size_t)k], cstart[(size_t)k + 1]);
                if (li + 1 < nl) {
                    ih_gemm_chunk(*l.layers[li + 1], out[li], gi[li + 1], cstart[(size_t)k], cstart[(size_t)k + 1] - cstart[(size_t)k], s);
                    ready[k] = next_event();
                    NC_HIP(hipEventRecord(ready[k], s));
                }
            }
        }
        hipEvent_t join = next_event();   // the handle's stream continues after everything issued on the second one
        NC_HIP(hipEventRecord(join, sB));
        NC_HIP(hipStreamWaitEvent(sA, join, 0));
        return out[nl - 1];
    }
    // generic fallback: one launch per time step (block = hidden unit with its 4 weight rows in LDS, thread = clip)
    const float* in = x;
    float* out = nullptr;
    for (int li = 0; li < nl; ++li) {
        LstmLayer& y = *l.layers[li];
        float* gi = alloc((size_t)N * 4 * C * T);
        ih_gemm(y, in, gi, stream);
        out = alloc((size_t)N * C * T);
        const bool last = li + 1 == nl;
        if (prof.on) prof.begin(stream, NC_KC_LSTM, 2.0 * 4 * C * C * (double)N * T, 4.0 * 6 * C * (double)N * T);
        float* h0 = alloc((size_t)C * N);
        float* h1 = alloc((size_t)C * N);
        float* cs = alloc((size_t)C * N);
        NC_HIP(hipMemsetAsync(h0, 0, (size_t)C * N * 4, stream));
        NC_HIP(hipMemsetAsync(cs, 0, (size_t)C * N * 4, stream));
        for (int64_t t = 0; t < T; ++t)
            hipLaunchKernelGGL(lstm_step_kernel, dim3((unsigned)C, (unsigned)((N + 63) / 64)), dim3(64), (size_t)4 * C * sizeof(float), stream,
                               gi, y.whh.as<float>(), y.bhh.as<float>(), (t & 1) ? h1 : h0, (t & 1) ? h0 : h1, cs, last ? x : nullptr, out, N, C, T, t, (last && elu_out) ? 1 : 0);
        NC_HIP(hipGetLastError());
        if (prof.on) prof.end(stream);
        in = out;
    }
    return out;
}

// EncodeFrame on N = clips of one segment length: x [N,channels,L] dense -> codes [N,n_q,T'] (+ scale [N], emb)
void EncodecModel::encode_batch(const float* x, int N, int64_t L, int64_t Tz, int64_t* codes, float* scale_out, float* emb_out) {
    Act cur;
    cur.p = x; cur.C = cfg.channels; cur.L = L; cur.rs = L; cur.off = 0; cur.stats = nullptr; cur.gamma = cur.beta = nullptr;
    if (cfg.normalize) {
        const int nchunk = (int)((L + GN_CHUNK - 1) / GN_CHUNK);
        double* part = reinterpret_cast<double*>(alloc((size_t)N * nchunk * 2));
        float* sc = scale_out ? scale_out : alloc((size_t)N);
        static const bool two_pass = env_flag("NC_RMS_TWO_PASS");
        if (!two_pass && N <= GN_MAX_SAMPLES) {
            const int nblk = (nchunk + RMS_G - 1) / RMS_G;
            ProfScope ps(&prof, stream, NC_KC_ELEM, 3.0 * N * cfg.channels * (double)L, 4.0 * N * cfg.channels * (double)L);
            hipLaunchKernelGGL(rms_scale_kernel, dim3((unsigned)((int64_t)N * nblk)), dim3(256), 0, stream, x, part,
                               gn_counters.as<unsigned>() + (size_t)cur_group * 2 * GN_MAX_SAMPLES, sc, cfg.channels, L, nchunk, nblk);
        } else {
            hipLaunchKernelGGL(rms_partial_kernel, dim3((unsigned)(((int64_t)N * nchunk + 63) / 64)), dim3(64), 0, stream, x, part, N, cfg.channels, L, nchunk);
            hipLaunchKernelGGL(rms_final_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, stream, part, sc, N, L, nchunk);
        }
        cur.p = materialize(cur, N, sc, 0);
    }
    cur = sconv(enc_in, cur, nullptr, false, N);
    for (int i = 0; i < cfg.n_ratios; ++i) {
        Act s, y;
        resblock(enc_res[i], cur, N, s, y);
        cur = sconv(enc_down[i], s, &y, true, N);
    }
    const float* xl = materialize(cur, N, nullptr, 0);
    Act a;
    // (the ELU in front of the last convolution is applied by the LSTM's output store: once per element instead of once per row tile
    //  of the consumer's staging, and the consumer runs as a plain convolution)
    const bool lstm_elu = lstm_applies_elu(enc_lstm);
    a.p = run_lstm(enc_lstm, xl, N, cur.L, lstm_elu); a.C = cur.C; a.L = cur.L; a.rs = cur.L; a.off = 0; a.stats = nullptr; a.gamma = a.beta = nullptr;
    Act e = sconv(enc_out, a, nullptr, !lstm_elu, N);
    if (e.L != Tz) fail(NC_ESTATE, "internal: encoder produced %lld frames, expected %lld", (long long)e.L, (long long)Tz);
    float* residual = materialize(e, N, nullptr, 0);
    const int D = cfg.dimension;
    if (emb_out) NC_HIP(hipMemcpyAsync(emb_out, residual, (size_t)N * D * Tz * 4, hipMemcpyDeviceToDevice, stream));
    const int64_t total = (int64_t)N * Tz;
    if (prof.on) prof.begin(stream, NC_KC_RVQ, 2.0 * D * cfg.codebook_size * (double)total * n_q, 0.0);   // the distance GEMM
    static const bool no_mfma_vq = env_present("NC_EUCLID_NO_MFMA");
    const int Nc = cfg.codebook_size;
    if (!no_mfma_vq && Nc % 512 == 0 && Nc <= 1024 && D == 128) {
        // all stages in one launch, cross terms on the matrix cores (the residual block stays in LDS between the stages)
        // (NC_RVQ_8WAVES=1: 8 wavefronts per workgroup, 128 codes each -- measured the same 236 us on C3's 150-workgroup grid as the
        // 4-wave form: the stage is bound by its serial phases and the codebook stream, not by the matrix-core chain)
#ifdef NC_EXPERIMENTS
        static const bool rvq8 = env_flag("NC_RVQ_8WAVES");
        const bool wide = rvq8 && Nc % 1024 == 0 && (total + EM_F - 1) / EM_F <= 256;
        if (wide)
            hipLaunchKernelGGL((euclid_rvq_mfma_kernel<128, 8>), dim3((unsigned)((total + EM_F - 1) / EM_F)), dim3(512), 0, stream, residual,
                               book_ptrsT.as<const float*>(), book_ptrs.as<const float*>(), book_ptrs2.as<const float*>(), n_q, Nc, N, Tz, codes,
                               (int64_t)n_q * Tz);
        else
#endif
        hipLaunchKernelGGL(euclid_rvq_mfma_kernel<128>, dim3((unsigned)((total + EM_F - 1) / EM_F)), dim3(256), 0, stream, residual,
                           book_ptrsT.as<const float*>(), book_ptrs.as<const float*>(), book_ptrs2.as<const float*>(), n_q, Nc, N, Tz, codes,
                           (int64_t)n_q * Tz);
    } else {
        for (int q = 0; q < n_q; ++q) {
            Codebook& cb = *books[q];
            hipLaunchKernelGGL(euclid_vq_kernel, dim3((unsigned)((total + EQ_F - 1) / EQ_F)), dim3(256), 0, stream, residual, cb.cbT.as<float>(),
                               cb.cb.as<float>(), cb.c2.as<float>(), cb.N, D, N, Tz, codes + (int64_t)q * Tz, (int64_t)n_q * Tz);
        }
    }
    NC_HIP(hipGetLastError());
    if (prof.on) prof.end(stream);
}

// DecodeFrame on N clips: codes [N,n_q,T'] -> out [N,channels,Lout] dense (x scale[n] when given)
float* EncodecModel::decode_batch(const int64_t* codes, int N, int nq, int64_t Tz, const float* scale, int64_t* Lout) {
    const int D = cfg.dimension;
    float* emb = alloc((size_t)N * D * Tz);
    {
        const int64_t n = (int64_t)N * D * Tz;
        hipLaunchKernelGGL(emb_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, codes, book_ptrs.as<const float*>(), nq,
                           cfg.codebook_size, D, N, Tz, emb);
        NC_HIP(hipGetLastError());
    }
    Act cur;
    cur.p = emb; cur.C = D; cur.L = Tz; cur.rs = Tz; cur.off = 0; cur.stats = nullptr; cur.gamma = cur.beta = nullptr;
    cur = sconv(dec_in, cur, nullptr, false, N);
    const float* xl = materialize(cur, N, nullptr, 0);
    Act a;
    const bool lstm_elu = lstm_applies_elu(dec_lstm);
    a.p = run_lstm(dec_lstm, xl, N, cur.L, lstm_elu); a.C = cur.C; a.L = cur.L; a.rs = cur.L; a.off = 0; a.stats = nullptr; a.gamma = a.beta = nullptr;
    Act s = a, y;
    bool dual = false;
    for (int i = 0; i < cfg.n_ratios; ++i) {
        Act u = sconvT(dec_up[i], s, dual ? &y : nullptr, !(i == 0 && lstm_elu), N);
        resblock(dec_res[i], u, N, s, y);
        dual = true;
    }
    Act o = sconv(dec_out, s, dual ? &y : nullptr, true, N);
    *Lout = o.L;
    return materialize(o, N, scale, 1);
}

void EncodecModel::encode_dev(const float* pcm, int B, int64_t T, int64_t* codes, float* scales, float* emb) {
    if (!loaded) fail(NC_ESTATE, "weights not loaded (call nc_codec_load_weights first)");
    if (!pcm || !codes) fail(NC_EINVAL, "pcm and codes must not be null");
    if (B <= 0 || T <= 0 || T > ((int64_t)1 << 30)) fail(NC_EINVAL, "B and T must be positive");
    use_device();
    check_async_errors();   // an earlier device-pointer call on this handle may have timed out unnoticed
    pool_i = 0;
    const std::vector<Seg> segs = segments(T);
    const int C = cfg.channels, D = cfg.dimension;
    int64_t code_off = 0, emb_off = 0;
    static const bool no_overlap = env_flag("NC_ENCODEC_NO_OVERLAP");
    hipStream_t const main_stream = stream;
    struct Restore { hipStream_t& s; hipStream_t v; ~Restore() { s = v; } } restore{stream, main_stream};   // also on an exception
    int n_groups = 0;
    bool side_used[2] = {false, false};
    NC_HIP(hipEventRecord(ev_fork, main_stream));
    // Consecutive segments of equal length run as ONE batch of G*B rows (every operator of the path is per sample: GroupNorm(1,C),
    // RMS scale, LSTM state, RVQ), segment-major -- so the batch's codes [G*B, n_q, T'] ARE the G frames' [B, n_q, T'] tensors laid
    // end to end, the layout the ABI emits.  Halves the number of dependent LSTM steps of a 2 s clip and doubles every grid.
    for (size_t f = 0; f < segs.size();) {
        size_t g = f + 1;
        while (g < segs.size() && segs[g].len == segs[f].len && (int64_t)(g - f + 1) * B <= 4096) ++g;
        const int G = (int)(g - f);
        const Seg& s = segs[f];
        const int side = (n_groups > 0 && !no_overlap) ? (n_groups - 1) % 2 : -1;   // groups after the first: side streams
        on_side_group = side >= 0;
        cur_group = side + 1;
        if (side >= 0) {
            stream = side_stream[side];
            if (!side_used[side]) NC_HIP(hipStreamWaitEvent(stream, ev_fork, 0));
            side_used[side] = true;
        }
        ++n_groups;
        float* x = alloc((size_t)G * B * C * s.len);
        for (int q = 0; q < G; ++q)   // slice segment f+q out of [B,C,T] into rows [q*B, (q+1)*B) of the dense [G*B,C,len] tensor
            NC_HIP(hipMemcpy2DAsync(x + (size_t)q * B * C * s.len, (size_t)s.len * 4, pcm + segs[f + q].off, (size_t)T * 4, (size_t)s.len * 4,
                                    (size_t)B * C, hipMemcpyDeviceToDevice, stream));
        float* sc = (cfg.normalize && scales) ? scales + (int64_t)f * B : nullptr;
        encode_batch(x, G * B, s.len, s.frames, codes + code_off, sc, emb ? emb + emb_off : nullptr);
        code_off += (int64_t)G * B * n_q * s.frames;
        emb_off += (int64_t)G * B * D * s.frames;
        stream = main_stream;
        on_side_group = false;
        cur_group = 0;
        f = g;
    }
    for (int i = 0; i < 2; ++i)
        if (side_used[i]) {
            NC_HIP(hipEventRecord(ev_join[i], side_stream[i]));
            NC_HIP(hipStreamWaitEvent(main_stream, ev_join[i], 0));
        }
}

void EncodecModel::decode_dev(const int64_t* codes, const float* scales, int B, int64_t T, int nq, float* pcm) {
    if (!loaded) fail(NC_ESTATE, "weights not loaded (call nc_codec_load_weights first)");
    if (!codes || !pcm) fail(NC_EINVAL, "codes and pcm must not be null");
    if (B <= 0 || T <= 0) fail(NC_EINVAL, "No frames provided to decode");                    // Encodec.cs:215-218
    if (nq <= 0 || nq > cfg.n_codebooks) fail(NC_EINVAL, "codes carry %d codebooks; the model has %d", nq, cfg.n_codebooks);
    if (cfg.normalize && !scales) fail(NC_EINVAL, "this model normalises frames: scales must be given");
    use_device();
    check_async_errors();
    pool_i = 0;
    const std::vector<Seg> segs = segments(T);
    const int C = cfg.channels;
    const int nfr = (int)segs.size();
    std::vector<const float*> fp((size_t)nfr);
    std::vector<int64_t> flen((size_t)nfr);
    int64_t code_off = 0;
    static const bool no_overlap = env_flag("NC_ENCODEC_NO_OVERLAP");
    hipStream_t const main_stream = stream;
    struct Restore { hipStream_t& s; hipStream_t v; ~Restore() { s = v; } } restore{stream, main_stream};   // also on an exception
    int n_groups = 0;
    bool side_used[2] = {false, false};
    NC_HIP(hipEventRecord(ev_fork, main_stream));
    for (size_t f = 0; f < segs.size();) {   // equal-length frames decode as one batch (see encode_dev)
        size_t g = f + 1;
        while (g < segs.size() && segs[g].frames == segs[f].frames && (int64_t)(g - f + 1) * B <= 4096) ++g;
        const int G = (int)(g - f);
        const int side = (n_groups > 0 && !no_overlap) ? (n_groups - 1) % 2 : -1;
        on_side_group = side >= 0;
        cur_group = side + 1;
        if (side >= 0) {
            stream = side_stream[side];
            if (!side_used[side]) NC_HIP(hipStreamWaitEvent(stream, ev_fork, 0));
            side_used[side] = true;
        }
        ++n_groups;
        int64_t Lo = 0;
        const float* out = decode_batch(codes + code_off, G * B, nq, segs[f].frames, cfg.normalize ? scales + (int64_t)f * B : nullptr, &Lo);
        for (int q = 0; q < G; ++q) {
            fp[f + q] = out + (size_t)q * B * C * Lo;
            flen[f + q] = Lo;
        }
        code_off += (int64_t)G * B * nq * segs[f].frames;
        stream = main_stream;
        on_side_group = false;
        cur_group = 0;
        f = g;
    }
    for (int i = 0; i < 2; ++i)
        if (side_used[i]) {
            NC_HIP(hipEventRecord(ev_join[i], side_stream[i]));
            NC_HIP(hipStreamWaitEvent(main_stream, ev_join[i], 0));
        }
    if (cfg.segment_length <= 0) {                                                           // single frame: DecodeFrame output as is
        NC_HIP(hipMemcpyAsync(pcm, fp[0], (size_t)B * C * flen[0] * 4, hipMemcpyDeviceToDevice, stream));
        return;
    }
    // triangular window + weight sum on the host (geometry only), AudioTensorDSP.cs:176-252
    const int64_t stride = cfg.segment_stride, total = stride * (nfr - 1) + flen[(size_t)nfr - 1], L0 = flen[0];
    for (int f = 0; f < nfr; ++f)
        if (flen[(size_t)f] > L0) fail(NC_EINVAL, "a later frame is longer than the first one");
    // window and weight sum: a function of the frame geometry alone -- computed once per geometry, kept on the device
    std::vector<int64_t> key{L0, stride, (int64_t)nfr};
    key.insert(key.end(), flen.begin(), flen.end());
    if (key != ola_key) {
        std::vector<float> w((size_t)L0), sw((size_t)total, 0.0f);
        for (int64_t i = 0; i < L0; ++i) {
            const float t = (float)((double)(i + 1) / (double)(L0 + 1));
            w[(size_t)i] = 0.5f - std::fabs(t - 0.5f);
        }
        for (int f = 0; f < nfr; ++f)
            for (int64_t i = 0; i < flen[(size_t)f]; ++i) sw[(size_t)(f * stride + i)] = sw[(size_t)(f * stride + i)] + w[(size_t)i];
        float mn = INFINITY;
        for (float v : sw) mn = std::min(mn, v);
        if (mn <= 1e-10f) for (float& v : sw) v = v + 1e-10f;
        NC_HIP(hipStreamSynchronize(stream));   // an earlier call may still read the old tables
        ola_w.reserve((size_t)L0 * 4);
        ola_sw.reserve((size_t)total * 4);
        NC_HIP(hipMemcpy(ola_w.p, w.data(), (size_t)L0 * 4, hipMemcpyHostToDevice));
        NC_HIP(hipMemcpy(ola_sw.p, sw.data(), (size_t)total * 4, hipMemcpyHostToDevice));
        ola_key = key;
    }
    const float* dw = ola_w.as<float>();
    const float* dsw = ola_sw.as<float>();
    // frame pointers / lengths of THIS call: pinned slot -> device arrays, asynchronously on the stream
    const size_t need = (size_t)nfr * (sizeof(float*) + sizeof(int64_t));
    if (need > ola_slot_bytes) {
        NC_HIP(hipStreamSynchronize(stream));
        if (ola_pin) NC_HIP(hipHostFree(ola_pin));
        ola_pin = nullptr;
        ola_slot_bytes = std::max<size_t>(1024, 2 * need);
        NC_HIP(hipHostMalloc(&ola_pin, 4 * ola_slot_bytes, hipHostMallocDefault));
        for (bool& u : ola_ev_used) u = false;
    }
    const int slot = ola_next;
    ola_next = (ola_next + 1) & 3;
    if (!ola_ev[slot]) NC_HIP(hipEventCreateWithFlags(&ola_ev[slot], hipEventDisableTiming));
    if (ola_ev_used[slot]) NC_HIP(hipEventSynchronize(ola_ev[slot]));   // the copy that last read this slot (4 calls ago) is done
    char* hs = static_cast<char*>(ola_pin) + (size_t)slot * ola_slot_bytes;
    std::memcpy(hs, fp.data(), (size_t)nfr * sizeof(float*));
    std::memcpy(hs + (size_t)nfr * sizeof(float*), flen.data(), (size_t)nfr * sizeof(int64_t));
    char* dslot = reinterpret_cast<char*>(alloc((need + 3) / 4 + 4));
    NC_HIP(hipMemcpyAsync(dslot, hs, need, hipMemcpyHostToDevice, stream));
    NC_HIP(hipEventRecord(ola_ev[slot], stream));
    ola_ev_used[slot] = true;
    const float** dfp = reinterpret_cast<const float**>(dslot);
    int64_t* dfl = reinterpret_cast<int64_t*>(dslot + (size_t)nfr * sizeof(float*));
    const int64_t n = (int64_t)B * C * total;
    hipLaunchKernelGGL(overlap_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dfp, dfl, nfr, L0, dw, dsw, (int64_t)B * C, stride,
                       total, pcm);
    NC_HIP(hipGetLastError());
}


void op_euclid_rvq(const float* residual_in, int B, int D, int64_t T, const float* books_host, int n_q, int N, int form, int64_t* codes_host,
                   float* residual_out) {
    std::vector<Codebook> books((size_t)n_q);
    std::vector<const float*> pT, pR, p2;
    for (int q = 0; q < n_q; ++q) {
        books[(size_t)q].build(books_host + (int64_t)q * N * D, N, D);
        pT.push_back(books[(size_t)q].cbT.as<float>()); pR.push_back(books[(size_t)q].cb.as<float>()); p2.push_back(books[(size_t)q].c2.as<float>());
    }
    DevBuf res, codes, dT, dR, d2;
    const size_t nb = (size_t)B * D * T * 4;
    res.reserve(nb); codes.reserve((size_t)B * n_q * T * 8);
    dT.reserve(pT.size() * sizeof(float*)); dR.reserve(pT.size() * sizeof(float*)); d2.reserve(pT.size() * sizeof(float*));
    NC_HIP(hipMemcpy(res.p, residual_in, nb, hipMemcpyHostToDevice));
    NC_HIP(hipMemcpy(dT.p, pT.data(), pT.size() * sizeof(float*), hipMemcpyHostToDevice));
    NC_HIP(hipMemcpy(dR.p, pR.data(), pR.size() * sizeof(float*), hipMemcpyHostToDevice));
    NC_HIP(hipMemcpy(d2.p, p2.data(), p2.size() * sizeof(float*), hipMemcpyHostToDevice));
    const int64_t total = (int64_t)B * T;
    if (form == 1) {
        if (D != 128 || N % 512 != 0 || N > 1024) fail(NC_EUNSUPPORTED, "the matrix-core Euclidean RVQ takes D == 128 and N = 512 or 1024");
        hipLaunchKernelGGL(euclid_rvq_mfma_kernel<128>, dim3((unsigned)((total + EM_F - 1) / EM_F)), dim3(256), 0, nullptr, res.as<float>(),
                           dT.as<const float*>(), dR.as<const float*>(), d2.as<const float*>(), n_q, N, B, T, codes.as<int64_t>(), (int64_t)n_q * T);
    } else {
        for (int q = 0; q < n_q; ++q)
            hipLaunchKernelGGL(euclid_vq_kernel, dim3((unsigned)((total + EQ_F - 1) / EQ_F)), dim3(256), 0, nullptr, res.as<float>(), pT[(size_t)q],
                               pR[(size_t)q], p2[(size_t)q], N, D, B, T, codes.as<int64_t>() + (int64_t)q * T, (int64_t)n_q * T);
    }
    NC_HIP(hipGetLastError());
    NC_HIP(hipDeviceSynchronize());
    NC_HIP(hipMemcpy(codes_host, codes.p, (size_t)B * n_q * T * 8, hipMemcpyDeviceToHost));
    if (residual_out) NC_HIP(hipMemcpy(residual_out, res.p, nb, hipMemcpyDeviceToHost));
    res.release(); codes.release(); dT.release(); dR.release(); d2.release();
    for (auto& b : books) { b.cbT.release(); b.cb.release(); b.c2.release(); }
}

}  // namespace nc
