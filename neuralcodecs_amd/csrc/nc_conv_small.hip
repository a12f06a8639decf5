// Short-row strided convolution on v_mfma_f32_16x16x4_f32: the deep down-convolutions of a ONE-clip (or few-clip) batch -- SNAC 24 kHz
// at one second: 384 -> 768, k = 16, stride 8 over 47 output frames (EncoderBlock.cs:42-47) -- are a skinny GEMM, 768 x 6144 weights
// against 47 columns.  On the 32x32x2 template such a launch is a handful of lone workgroups, each walking its reduction as ONE dependent
// chain of 3072 matrix-core instructions of 64 cycles (245 us measured, 8 % of the whole C1 step per layer).  The 16x16x4 instruction
// retires four reduction steps per 32 cycles -- the same canonical chain (bitwise a k-ordered fmaf chain, as in the persistent LSTM),
// four times shorter -- and 16-column tiles turn 47 frames into three workgroups per row block instead of one.
//   * workgroup = 4 wavefronts = 64 output rows x 16 output columns of one clip; wave w owns rows 16w .. 16w+15;
//   * weights never touch LDS: every wave streams its own A fragments from a packed image ([16-row tile][group of 4 steps][lane][4],
//     one 16-byte load per lane and group) through a ring of SMALL_PF groups in flight;
//   * the input window of the 16 columns (15*stride + (K-1)*dil + 1 samples per channel) is staged through LDS, SMALL_CB channels per
//     block, double-buffered, one barrier per block; zero padding by predicate;
//   * per output: fmaf over kk = ci*K + k ascending from +0 (four kk per instruction), then + bias -- the canonical chain.
// Forms: the rolled kernel (any eligible layer, 16-column tiles); the straight-line kernel `conv_small_unrolled_kernel<GPB, NB, TN, CB,
// NS, INM, GN>` -- 16- or 32-column tiles (TN), k = 16 / 10 / 8 / 6 / 4 strided layers with 8 channels per block, k = 7 / k = 3
// stride-1 layers with 16; INM: Encodec's pending GroupNorm + ELU applied while the window is written to LDS; GN: GroupNorm block
// sums of the output in the canonical order, finished in the launch.  The input may be SConv1d's reflect-padded view of an un-padded
// row (an index map); the epilogue adds the bias and, when asked, the consumer's Snake.
#include <cstdlib>
#include <type_traits>
#include <utility>
#include <vector>

#include "nc_conv.h"
#include "nc_gn.h"
#include "nc_math.h"

namespace nc {

typedef float small_f32x4 __attribute__((ext_vector_type(4)));

template <int N, class F, int... I>
__device__ __forceinline__ void nc_static_for_small_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void nc_static_for_small(F&& f) {
    nc_static_for_small_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

constexpr int SMALL_CB = 8;    // input channels per LDS block
constexpr int SMALL_PF = 12;   // weight groups (4 matrix-core steps each) in flight per wave

struct ConvSmallArgs {
    const float* x;
    int64_t x_bstride, x_cstride;
    int x_len;
    int in_left, in_Lz, in_L;   // in_L > 0: x is the UN-padded row of in_L samples and positions run over SConv1d's reflect-padded row of x_len
                                // samples (SConv1d.cs:258-274): position g reads sample |g - in_left| mirrored at in_Lz - 1, zero past in_L
    // Encodec input mode (the unrolled INM instances): the row is a raw conv output with a pending GroupNorm(1,C) -- (mean, rstd) per sample,
    // (gamma, beta) per channel, NormConv1d.cs:155 -- and / or a pending ELU, applied while the window is written to LDS (the template's
    // formula: ((x - mu) * rstd) * gamma + beta, then ELU, then the zero extension of the ACTIVATED row)
    const float* in_stats; const float* in_gamma; const float* in_beta; int in_elu;
    // GroupNorm block sums of the output (the GN instances, 32-column tiles): [B][gn_nrb][gn_ncb][2] partial sums in the canonical order
    // of nc_gn.h, finished in the launch by the sample's last workgroup when gn_count is given
    double* gn_part; int gn_nrb, gn_ncb; unsigned* gn_count; float* gn_stats; double gn_n;
    const float* wp;     // packed image, see pack_small()
    const float* bias;   // nullable
    const float* alpha_out;   // nullable: Snake of the consuming layer applied to the stored value (Snake1d.cs:40-63)
    float* y;
    int64_t y_bstride, y_cstride;
    int Cin, Cout, K, stride, pad, dil, Tout;
    int n_t_tiles, n_row_tiles;   // 16-column tiles per clip, 64-row tiles
    int W;                         // window samples per channel: 15*stride + (K-1)*dil + 1
    int groups;                    // Cin*K / 16: groups of 4 steps in the whole reduction
};

// value of the canonical chain -> stored value: + bias, then the consumer's Snake when the producer applies it
__device__ __forceinline__ float small_epilogue(const ConvSmallArgs& a, int row, float v) {
    v = v + (a.bias ? a.bias[row] : 0.0f);
    if (a.alpha_out) {
        const float al = a.alpha_out[row];
        v = nc_snakef(v, al, nc_snake_inv(al));
    }
    return v;
}

__global__ __launch_bounds__(256) void conv_small_kernel(const ConvSmallArgs a) {
    extern __shared__ __attribute__((aligned(16))) float xs[];   // [2][SMALL_CB][W]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    const int rt = bid % a.n_row_tiles; bid /= a.n_row_tiles;        // row tiles of one column tile are adjacent: they share the window in L2
    const int tt = bid % a.n_t_tiles;
    const int b = bid / a.n_t_tiles;
    const int K = a.K, W = a.W, dil = a.dil;
    const int col = lane & 15, kq = lane >> 4;
    const int g0 = tt * 16 * a.stride - a.pad;                       // input position of window slot 0
    const float* xb = a.x + (int64_t)b * a.x_bstride;
    // this wave's weight stream: 16-row tile (4*rt + wave), groups in reduction order
    const small_f32x4* wsrc = reinterpret_cast<const small_f32x4*>(a.wp) + ((int64_t)(4 * rt + wave) * a.groups) * 64 + lane;
    const int groups = a.groups;
    small_f32x4 ring[SMALL_PF];
#pragma unroll
    for (int i = 0; i < SMALL_PF; ++i) ring[i] = wsrc[(int64_t)min(i, groups - 1) * 64];
    // window staging: block cbk = channels [cbk*CB, cbk*CB + CB), slot i of the block = channel i / W, position i % W
    const int n_blocks = (a.Cin + SMALL_CB - 1) / SMALL_CB;
    const int n_slots = SMALL_CB * W;
    constexpr int NS = 6;                                            // slots per thread (CB * W <= 6 * 256)
    float rx[NS];
    auto issue = [&](int cbk) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int i = tid + 256 * u;
            const int c = i / W, j = i - c * W;
            const int ci = cbk * SMALL_CB + c;
            int g = g0 + j;
            bool ok = (i < n_slots) & (ci < a.Cin) & (g >= 0) & (g < a.x_len);
            if (a.in_L > 0) {
                int q = g - a.in_left;
                q = q < 0 ? -q : q;
                if (q >= a.in_Lz) q = 2 * (a.in_Lz - 1) - q;
                ok = ok & (q >= 0) & (q < a.in_L);
                g = q;
            }
            rx[u] = ok ? xb[(int64_t)ci * a.x_cstride + g] : 0.0f;
        }
    };
    auto store = [&](float* dst) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int i = tid + 256 * u;
            if (i < n_slots) dst[i] = rx[u];
        }
    };
    issue(0);
    store(xs);
    __syncthreads();
    small_f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    // This lane's reduction index inside a block is kk_local = 4*step + kq -> (channel c, tap k), i.e. window offset c*W + col*stride + k*dil.
    // (k, off) is the READ cursor: the four B values of a group are read one group ahead of the matrix-core steps that consume them.
    const int groups_per_block = SMALL_CB * K / 16;                  // (host: CB*K % 16 == 0, blocks are whole groups)
    int cbk = 0, gblk = 0;
    const float* xc = xs;
    if (n_blocks > 1) issue(1);
    int k = kq, off = col * a.stride + kq * dil;                     // (K >= 4: kq < K)
    auto read_group = [&](float (&bv)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bv[j] = xc[off];
            k += 4; off += 4 * dil;
            if (k >= K) { k -= K; off += W - K * dil; }              // next channel row of the block
        }
    };
    float bc[4], bn[4];
    read_group(bc);
    for (int gbase = 0; gbase < groups; gbase += SMALL_PF) {
        nc_static_for_small<SMALL_PF>([&](auto it) __attribute__((always_inline)) {
            constexpr int i = decltype(it)::value;
            if (gbase + i < groups) {                                // (wave-uniform)
                const small_f32x4 wv = ring[i];
                ring[i] = wsrc[(int64_t)min(gbase + i + SMALL_PF, groups - 1) * 64];
                const bool last_of_block = gblk + 1 == groups_per_block;
                if (!last_of_block) read_group(bn);                  // next group of this block: its reads land under this group's steps
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j], bc[j], acc, 0, 0, 0);
                if (last_of_block) {
                    ++cbk;
                    gblk = 0;
                    if (cbk < n_blocks) {                            // publish the staged block, start staging the one after
                        store(xs + (cbk & 1) * n_slots);
                        __syncthreads();
                        xc = xs + (cbk & 1) * n_slots;
                        if (cbk + 1 < n_blocks) issue(cbk + 1);
                        k = kq; off = col * a.stride + kq * dil;
                        read_group(bn);
                    }
                } else {
                    ++gblk;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) bc[j] = bn[j];
            }
        });
    }
    // D: lane holds rows 4*kq + r (r = 0..3) of its wave's 16-row tile, column col
    const int t = tt * 16 + col;
    if (t < a.Tout) {
        float* yb = a.y + (int64_t)b * a.y_bstride + t;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = (4 * rt + wave) * 16 + 4 * kq + r;
            if (row < a.Cout) yb[(int64_t)row * a.y_cstride] = small_epilogue(a, row, acc[r]);
        }
    }
}

// Straight-line form: NB blocks of GPB weight groups per loop iteration, the ring position of every group a compile-time constant.
// In the rolled kernel above the compiler cannot see, across its branches, how many weight loads were issued after a block's window
// loads, and drains the whole ring in front of every block barrier (s_waitcnt vmcnt(1): one memory latency per 8 channels -- 48 of
// them in the 384 -> 768 layer, 98 us).  Here the window loads of block n+1 are followed by exactly GPB ring loads before the barrier
// that needs them, the wait is vmcnt(GPB), and the weight stream never stops.  Needs n_blocks % NB == 0 (the host checks).
template <int GPB, int NB, int TN = 1, int CB = SMALL_CB, int NS = (TN == 1 ? 6 : TN == 2 ? 9 : 18), bool INM = false, bool GN = false>
__global__ __launch_bounds__(256) void conv_small_unrolled_kernel(const ConvSmallArgs a) {
    constexpr int BNC = 16 * TN;                                      // output columns per workgroup: TN column tiles share every A fragment
    constexpr int PF = GPB * NB;                                      // ring depth = groups per loop iteration
    extern __shared__ __attribute__((aligned(16))) float xs[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    const int rt = bid % a.n_row_tiles; bid /= a.n_row_tiles;
    const int tt = bid % a.n_t_tiles;
    const int b = bid / a.n_t_tiles;
    const int K = a.K, W = a.W, dil = a.dil;
    const int col = lane & 15, kq = lane >> 4;
    const int g0 = tt * BNC * a.stride - a.pad;
    const float* xb = a.x + (int64_t)b * a.x_bstride;
    const small_f32x4* wsrc = reinterpret_cast<const small_f32x4*>(a.wp) + ((int64_t)(4 * rt + wave) * a.groups) * 64 + lane;
    const int groups = a.groups;
    small_f32x4 ring[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) ring[i] = wsrc[(int64_t)min(i, groups - 1) * 64];
    const int n_blocks = a.Cin / CB;
    const int n_slots = CB * W;
    float rx[NS];
    // INM: (gamma, beta) of all input channels behind the two window buffers; the sample's (mean, rstd)
    float2* const Gt = reinterpret_cast<float2*>(xs + 2 * n_slots);
    float in_mu = 0.0f, in_rs = 1.0f;
    if constexpr (INM) {
        if (a.in_stats) {
            in_mu = a.in_stats[2 * b]; in_rs = a.in_stats[2 * b + 1];
            for (int i = tid; i < a.Cin; i += 256) Gt[i] = make_float2(a.in_gamma[i], a.in_beta[i]);
        }
        __syncthreads();
    }
    // loop-invariant part of the window reads: slot -> (channel of the block, position), predicate of the position
    int xo[NS], cu[INM ? NS : 1];
    unsigned okm = 0;
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        const int i = tid + 256 * u;
        const int c = i / W, j = i - c * W;
        if constexpr (INM) cu[u] = min(c, CB - 1);
        int g = g0 + j;
        bool ok = (i < n_slots) & (g >= 0) & (g < a.x_len);
        if (a.in_L > 0) {   // reflect-padded view of an un-padded row
            int q = g - a.in_left;
            q = q < 0 ? -q : q;
            if (q >= a.in_Lz) q = 2 * (a.in_Lz - 1) - q;
            ok = ok & (q >= 0) & (q < a.in_L);
            g = q;
        }
        if (ok) okm |= 1u << u;
        xo[u] = ok ? c * (int)a.x_cstride + g : 0;                   // (host: Cin * x_cstride fits 31 bits)
    }
    auto issue = [&](int cbk) __attribute__((always_inline)) {
        const float* xr = xb + (int64_t)cbk * CB * a.x_cstride;
#pragma unroll
        for (int u = 0; u < NS; ++u) rx[u] = xr[xo[u]];               // branch-free: masked slots read a valid word and are zeroed at the store
    };
    auto store = [&](float* dst, int cbk) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int i = tid + 256 * u;
            float v = rx[u];
            if constexpr (INM) {
                if (a.in_stats) {
                    const float2 gb = Gt[cbk * CB + cu[u]];
                    v = ((v - in_mu) * in_rs) * gb.x + gb.y;
                }
                if (a.in_elu) v = nc_eluf(v);
            }
            if (i < n_slots) dst[i] = ((okm >> u) & 1u) ? v : 0.0f;
        }
    };
    issue(0);
    store(xs, 0);
    __syncthreads();
    small_f32x4 acc[TN];
#pragma unroll
    for (int c = 0; c < TN; ++c) acc[c] = small_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const int off0 = col * a.stride + kq * dil, cts = 16 * a.stride;   // cts: window offset between adjacent column tiles
    for (int cb0 = 0; cb0 < n_blocks; cb0 += NB) {
        nc_static_for_small<NB>([&](auto bt) __attribute__((always_inline)) {
            constexpr int nb = decltype(bt)::value;
            const int cbk = cb0 + nb;
            const float* xc = xs + (cbk & 1) * n_slots;
            const int nxt = min(cbk + 1, n_blocks - 1);              // (the last block re-reads itself: branch-free, never stored)
            issue(nxt);
            int k = kq, off = off0;
            if constexpr (CB == 64) { k = 0; off = col * a.stride + kq * W; }   // (pointwise: lane group kq starts in channel row kq)
            else if (k >= K) { k -= K; off += W - K * dil; }         // (k = 3: the fourth lane group starts in the next channel row)
            float bv[2][4][TN];
            auto read_group = [&](float (&v)[4][TN]) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int c = 0; c < TN; ++c) v[j][c] = xc[off + c * cts];
                    if constexpr (CB == 64) {                        // the pointwise instance: every reduction index is its own channel row
                        off += 4 * W;
                    } else {
                        k += 4; off += 4 * dil;
                        if (k >= K) { k -= K; off += W - K * dil; }
                        if (k >= K) { k -= K; off += W - K * dil; }  // (second wrap: k = 3 only)
                    }
                }
            };
            read_group(bv[0]);
            nc_static_for_small<GPB>([&](auto gt) __attribute__((always_inline)) {
                constexpr int g = decltype(gt)::value, ri = nb * GPB + g;
                const small_f32x4 wv = ring[ri];
                ring[ri] = wsrc[(int64_t)min((cbk + NB) * GPB + g, groups - 1) * 64];
                if constexpr (g + 1 < GPB) read_group(bv[(g + 1) & 1]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int c = 0; c < TN; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j], bv[g & 1][j][c], acc[c], 0, 0, 0);
            });
            store(xs + ((cbk + 1) & 1) * n_slots, nxt);               // (after the last block: a dead store into the free buffer)
            __syncthreads();
        });
    }
    if constexpr (GN) {
        static_assert(!GN || TN == 2, "the GroupNorm epilogue reduces whole 32x32 blocks: 32-column tiles");
        // The 64 x 32 tile meets in LDS (the window buffers are free after the last barrier), and waves 0 / 1 re-read its two 32 x 32
        // blocks in the accumulator layout of v_mfma_f32_32x32x2_f32 -- lane = slot 32*h + c, register r = row (r&3) + 8*(r>>2) + 4*h --
        // i.e. the canonical order of nc_gn.h: the same sums, bit for bit, as the template's register reduction.
        float* const tile = xs;                                       // [64][33]
#pragma unroll
        for (int c = 0; c < TN; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lr = 16 * wave + 4 * kq + r, row = 64 * rt + lr;
                tile[lr * 33 + c * 16 + col] = acc[c][r] + ((a.bias && row < a.Cout) ? a.bias[row] : 0.0f);
            }
        __syncthreads();
        if (wave < 2) {
            const int h = lane >> 5, cc = lane & 31;
            const int t = tt * 32 + cc;
            float vv[16];
            unsigned okm16 = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h;
                vv[r] = tile[lr * 33 + cc];
                if ((64 * rt + lr < a.Cout) & (t < a.Tout)) okm16 |= 1u << r;
            }
            double s1, s2;
            nc_gn_slot_sums<false>(vv, okm16, s1, s2);
            nc_gn_butterfly(s1, s2);
            const int rbk = 2 * rt + wave, cbk = tt;
            if (lane == 0 && rbk < a.gn_nrb && cbk < a.gn_ncb)
                nc_gn_store_partial(a.gn_part + (((int64_t)b * a.gn_nrb + rbk) * a.gn_ncb + cbk) * 2, s1, s2);
        }
        if (a.gn_count != nullptr)
            nc_gn_arrive_and_finish(a.gn_part + (int64_t)b * a.gn_nrb * a.gn_ncb * 2, a.gn_count + b, a.gn_stats + 2 * b, a.gn_nrb * a.gn_ncb,
                                    (unsigned)(a.n_row_tiles * a.n_t_tiles), a.gn_n);
    }
#pragma unroll
    for (int c = 0; c < TN; ++c) {
        const int t = tt * BNC + c * 16 + col;
        if (t < a.Tout) {
            float* yb = a.y + (int64_t)b * a.y_bstride + t;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = (4 * rt + wave) * 16 + 4 * kq + r;
                if (row < a.Cout) yb[(int64_t)row * a.y_cstride] = small_epilogue(a, row, acc[c][r]);
            }
        }
    }
}

// packed image for conv_small_kernel: float index ((tile*groups + g)*64 + lane)*4 + j = W[row = 16*tile + (lane & 15)][kk = 16g + 4j + (lane >> 4)]
void pack_conv_small(const float* dense_w, int Cin, int Cout, int K, std::vector<float>& out) {
    const int tiles = ((Cout + 63) / 64) * 4, groups = Cin * K / 16;
    out.assign((size_t)tiles * groups * 256, 0.0f);
    for (int tile = 0; tile < tiles; ++tile)
        for (int g = 0; g < groups; ++g)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 4; ++j) {
                    const int row = 16 * tile + (l & 15), kk = 16 * g + 4 * j + (l >> 4);
                    if (row < Cout) out[(((size_t)tile * groups + g) * 64 + l) * 4 + j] = dense_w[(size_t)row * Cin * K + kk];   // [Cout][Cin][K]: kk = ci*K + k
                }
}

// k = 7, stride 1 (the 512 <-> 128 convolutions around Encodec's quantizer, DAC's 1024 -> 1536 decoder input): 16 channels per block
// ... and k = 3, stride 1 (DAC's 1024 -> 1024 encoder output convolution): the same 16-channel blocks, 3 groups each
// ... and the wide pointwise GEMMs over few columns (the chunked LSTM input projections, 512 -> 2048 over 44 steps x 32 rows): 64
// channels per block
static bool small_k1(int Cin, int Cout, int K, int stride) { return K == 1 && stride == 1 && Cin % 256 == 0 && Cout >= 1024; }
static bool small_k7(int Cin, int K, int stride, int dil) { return (K == 7 || K == 3) && stride == 1 && dil == 1 && Cin % 32 == 0; }

bool conv_small_eligible(int Cin, int Cout, int K, int stride, int dil, bool transposed) {
    if (!transposed && small_k1(Cin, Cout, K, stride)) return true;
    if (!transposed && Cout >= 64 && small_k7(Cin, K, stride, dil)) return true;
    if (transposed || K < 4 || stride < 2 || Cout < 64) return false;
    if ((Cin * K) % 16 != 0 || (SMALL_CB * K) % 16 != 0 || Cin % SMALL_CB != 0) return false;
    const int W = 15 * stride + (K - 1) * dil + 1;
    return SMALL_CB * W <= 6 * 256 && (size_t)2 * SMALL_CB * W * 4 <= 64 * 1024;
}

// widest column tile (in 16-column units) the instantiated kernels offer for this layer
int conv_small_max_tn(int Cin, int K, int stride, int dil) {
    if (K == 1 && stride == 1 && Cin % 256 == 0) return 2;
    if (small_k7(Cin, K, stride, dil)) return 2;
    const int W32 = 31 * stride + (K - 1) * dil + 1;
    const bool k16 = K * SMALL_CB / 16 == 8 && Cin % SMALL_CB == 0 && (Cin / SMALL_CB) % 2 == 0;
    return (k16 && SMALL_CB * W32 <= 9 * 256) ? 2 : 1;
}

// true when a GroupNorm-sum epilogue instance serves this layer (32-column tiles; inm: the input carries a pending GroupNorm / ELU)
bool conv_small_gn_available(int Cin, int K, int stride, int dil, bool inm) {
    if (conv_small_max_tn(Cin, K, stride, dil) < 2) return false;
    if (small_k7(Cin, K, stride, dil)) return K == 3 ? Cin <= 1024 : !inm;
    return !inm;   // k = 16
}

// true when the input-mode (pending GroupNorm / ELU) instances serve this layer
bool conv_small_inm_available(int Cin, int K, int stride, int dil) { return K == 3 && small_k7(Cin, K, stride, dil) && Cin <= 1024; }

bool launch_conv_small(const float* x, int64_t x_bstride, int64_t x_cstride, int x_len, int in_left, int in_Lz, int in_L, const float* in_stats,
                       const float* in_gamma, const float* in_beta, int in_elu, const ConvSmallGn* gn, const float* wp, const float* bias, const float* alpha_out, float* y,
                       int64_t y_bstride, int64_t y_cstride, int B, int Cin, int Cout, int K, int stride, int pad, int dil, int Tout, int want_tn, hipStream_t s) {
    ConvSmallArgs a{};
    a.x = x; a.x_bstride = x_bstride; a.x_cstride = x_cstride; a.x_len = x_len; a.wp = wp; a.bias = bias; a.alpha_out = alpha_out;
    a.in_left = in_left; a.in_Lz = in_Lz; a.in_L = in_L;
    a.in_stats = in_stats; a.in_gamma = in_gamma; a.in_beta = in_beta; a.in_elu = in_elu;
    if (gn && gn->part) {
        if (alpha_out || conv_small_max_tn(Cin, K, stride, dil) < 2) return false;
        a.gn_part = gn->part; a.gn_nrb = gn->nrb; a.gn_ncb = gn->ncb; a.gn_count = gn->count; a.gn_stats = gn->stats; a.gn_n = gn->n;
        want_tn = 2;
    }
    const bool with_gn = a.gn_part != nullptr;
    const bool inm = in_stats != nullptr || in_elu != 0;
    if (inm && !conv_small_inm_available(Cin, K, stride, dil)) return false;
    a.y = y; a.y_bstride = y_bstride; a.y_cstride = y_cstride;
    a.Cin = Cin; a.Cout = Cout; a.K = K; a.stride = stride; a.pad = pad; a.dil = dil; a.Tout = Tout;
    a.n_row_tiles = (Cout + 63) / 64;
    a.groups = Cin * K / 16;
    // 32-column workgroups (two column tiles per A fragment: half the weight traffic) where the caller asks for them -- launches with
    // thousands of 16-column tiles are bound by the weight stream out of L2; 16-column tiles for the latency-bound launches.  (A
    // 64-column form is instantiated behind NC_SMALL_TN=4: 429 registers, one wave per SIMD, slower.)
    static const int tn_env = (int)env_int("NC_SMALL_TN", 0);
    const int W64 = 63 * stride + (K - 1) * dil + 1;
    const bool tn2_fits = conv_small_max_tn(Cin, K, stride, dil) >= 2, tn4_fits = tn2_fits && SMALL_CB * W64 <= 18 * 256;
    const int TN = with_gn ? 2 : (tn4_fits && tn_env == 4 && !small_k7(Cin, K, stride, dil)) ? 4 : (tn2_fits && (tn_env == 2 || (tn_env == 0 && want_tn >= 2))) ? 2 : 1;
    a.n_t_tiles = (Tout + 16 * TN - 1) / (16 * TN);
    a.W = (16 * TN - 1) * stride + (K - 1) * dil + 1;
    const bool k7 = small_k7(Cin, K, stride, dil), k1 = small_k1(Cin, Cout, K, stride);
    size_t lds = (size_t)2 * (k1 ? 64 : k7 ? 16 : SMALL_CB) * a.W * sizeof(float) + (inm ? (size_t)Cin * sizeof(float2) : 0);
    if (with_gn) lds = std::max(lds, (size_t)64 * 33 * sizeof(float));
    const int64_t grid = (int64_t)B * a.n_t_tiles * a.n_row_tiles;
    if (grid <= 0 || grid > 0x7fffffff) return false;
    // the straight-line form where the block count is a multiple of its unroll; the rolled kernel otherwise
    static const bool rolled_only = env_flag("NC_SMALL_ROLLED");
    const int gpb = SMALL_CB * K / 16, n_blocks = Cin / SMALL_CB;
    const bool fits31 = (int64_t)Cin * x_cstride + x_len < ((int64_t)1 << 31);
    void (*fn)(const ConvSmallArgs) = conv_small_kernel;
    if (k1) {
        if (!fits31 || TN != 2 || with_gn || inm) return false;
        fn = conv_small_unrolled_kernel<4, 4, 2, 64, 8>;
    } else if (k7) {
        if (!fits31) return false;
        if (with_gn) {
            if (K == 7 && !inm) fn = conv_small_unrolled_kernel<7, 2, 2, 16, 3, false, true>;
            else if (K == 3) fn = conv_small_unrolled_kernel<3, 2, 2, 16, 3, true, true>;   // (the INM instance also runs plain rows: flags at run time)
            else return false;
        } else if (K == 7) fn = TN == 2 ? conv_small_unrolled_kernel<7, 2, 2, 16, 3> : conv_small_unrolled_kernel<7, 2, 1, 16, 2>;
        else if (inm) fn = TN == 2 ? conv_small_unrolled_kernel<3, 2, 2, 16, 3, true> : conv_small_unrolled_kernel<3, 2, 1, 16, 2, true>;
        else fn = TN == 2 ? conv_small_unrolled_kernel<3, 2, 2, 16, 3> : conv_small_unrolled_kernel<3, 2, 1, 16, 2>;
    } else if (TN == 4) {
        if (!fits31) return false;
        fn = conv_small_unrolled_kernel<8, 2, 4>;
    } else if (TN == 2) {
        if (!fits31 || (with_gn && inm)) return false;
        fn = with_gn ? conv_small_unrolled_kernel<8, 2, 2, SMALL_CB, 9, false, true> : conv_small_unrolled_kernel<8, 2, 2>;
    } else if (!rolled_only && fits31 && Cin % SMALL_CB == 0) {
        if (gpb == 8 && n_blocks % 2 == 0) fn = conv_small_unrolled_kernel<8, 2>;
        else if (gpb == 4 && n_blocks % 4 == 0) fn = conv_small_unrolled_kernel<4, 4>;
        else if (gpb == 5 && n_blocks % 4 == 0) fn = conv_small_unrolled_kernel<5, 4>;
        else if (gpb == 3 && n_blocks % 6 == 0) fn = conv_small_unrolled_kernel<3, 6>;
        else if (gpb == 2 && n_blocks % 8 == 0) fn = conv_small_unrolled_kernel<2, 8>;
    }
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(256), lds, s, a);
    NC_HIP(hipGetLastError());
    return true;
}

}  // namespace nc
