// Pointwise (k=1) convolution on the fp32 matrix cores with register-fed B fragments.
//
// A 1x1 conv is the plain GEMM  Y[Cout, T] = W[Cout, Cin] . X[Cin, T]  per clip (ResidualUnit.cs:33 / Modules/SNAC/ResidualUnit.cs:33
// second conv, NoiseBlock.cs:38, LocalMHA.cs:92,112 projections, SLSTM input projections).  Unlike the k>1 convolutions there is no tap
// reuse and no halo, and each wavefront's columns are private to it, so staging X through LDS buys nothing: the B fragment of MFMA step kp
// (lane l: X[ci = 2*kp + (l>>5)][col(l&31, j)]) is loaded straight from global memory, TN consecutive columns per lane in one
// 4*TN-byte load, through a rolling register pipeline PF steps deep.  Only the weight tile (shared by the 4 waves) goes through LDS,
// double-buffered, one barrier per CB input channels.  Column permutation: column of (lane, j) = col0 + wave*32*TN + TN*lane + j, so
// loads and stores are TN-wide vectors.  Accumulation order per output = ci ascending from +0, then + bias, then + residual: the
// canonical chain of DESIGN.md, bit-identical to the generic conv template and the oracle.
#include <algorithm>
#include <mutex>
#include <set>
#include <type_traits>
#include <utility>

#include "nc_conv.h"
#include "nc_math.h"
#include "nc_frag.h"
#include "nc_gn.h"

namespace nc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int N, class F, int... I>
__device__ __forceinline__ void nc_static_for_1x1_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void nc_static_for_1x1(F&& f) {
    nc_static_for_1x1_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// Waves per SIMD the register budget is held to (the layers this kernel serves are short reductions: occupancy hides their
// memory latency), matching the workgroups-per-CU table of the launch-time tile choice.
constexpr int conv1x1_occupancy(int TM) { return TM == 1 ? 6 : TM == 2 ? 5 : 3; }

// MODE: epilogue of this instantiation, chosen at launch (bit 0: residual, bit 1: Snake of the consuming layer, bit 2: noise
// injection y = res + noise * conv): one straight-line epilogue per kernel keeps the register budget.
// INM: Encodec input mode (ConvArgs::in_mode bits 0/1): the B fragments are raw conv outputs with a pending GroupNorm(1,C) and ELU,
// applied to the fragment registers right before they feed the matrix cores (a 1x1 has no padding, so no reflect map here).
template <int TM, int CB, int MODE, bool INM = false>
__global__ __launch_bounds__(256, conv1x1_occupancy(TM)) void conv1x1_kernel(const ConvArgs p) {
    constexpr int TN = 2;
    constexpr int BM = 32 * TM;
    constexpr int BNW = 32 * TN;
    constexpr int BN = 4 * BNW;
    constexpr int KP = CB / 2;
    constexpr int A_FLOATS = CB * BM;
    constexpr int A_VEC = A_FLOATS / 4;
    constexpr int NA = (A_VEC + 255) / 256;
    constexpr int PF = 4;                       // B prefetch distance in MFMA steps (divides KP); an 8-step ring measures the same
    static_assert(KP % PF == 0, "prefetch ring must divide the reduction block");

    __shared__ __attribute__((aligned(16))) float As[2][A_FLOATS];
    __shared__ float Ep[3 * BM];   // per-row epilogue operands of this tile: bias, Snake alpha, 1/alpha
    __shared__ float2 Gt[INM ? 512 : 1];   // INM: (gamma, beta) of the pending GroupNorm per input channel (Cin <= 512)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hi = lane >> 5;

    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    int lin;
    {
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int co_in = __builtin_amdgcn_readfirstlane(lin % p.co_group);   // row tiles in groups: see conv_mfma_kernel's tile map
    lin /= p.co_group;
    const int t_tile = __builtin_amdgcn_readfirstlane(lin % p.n_t_tiles);
    lin /= p.n_t_tiles;
    const int b = __builtin_amdgcn_readfirstlane(lin % p.B);
    const int co_tile = __builtin_amdgcn_readfirstlane((lin / p.B) * p.co_group + co_in);
    for (int i = tid; i < BM; i += 256) {
        const int co = min(co_tile * BM + i, p.Cout - 1);
        const float ao = p.alpha_out ? p.alpha_out[co] : 0.0f;
        Ep[i] = p.bias ? p.bias[co] : 0.0f;
        Ep[BM + i] = ao;
        Ep[2 * BM + i] = nc_snake_inv(ao);
    }

    const int T = p.Tout;
    const int n_cb = p.n_cb, Cin = p.Cin;
    const int in_mode = INM ? p.in_mode : 0;
    float in_mu = 0.0f, in_rs = 1.0f;
    if constexpr (INM) {
        if (in_mode & 1) {
            in_mu = p.in_stats[2 * b];
            in_rs = p.in_stats[2 * b + 1];
            for (int i = tid; i < n_cb * CB; i += 256) Gt[i] = make_float2(p.in_gamma[min(i, Cin - 1)], p.in_beta[min(i, Cin - 1)]);
        }
    }
    const unsigned x_cstride = (unsigned)p.x_cstride;
    const int col = t_tile * BN + wave * BNW + TN * l31;          // first of this lane's TN columns
    const int colc = min(col, T - TN);                             // clamped (T % TN == 0, T >= TN): loads stay inside the row
    const float* const xb = p.x + (int64_t)b * p.x_bstride;       // uniform; the lane part is a 32-bit offset
    // B reads are buffer loads: resource = this clip's rows (uniform), the lane part a constant 32-bit byte offset, the channel pair of
    // step g a SCALAR byte offset -- no vector address arithmetic per read (a vector instruction in the step is issued in the matrix
    // pipe's time).  The launcher admits the layer only when a clip's rows span < 4 GB.
    const unsigned x_lane_off = 4u * ((unsigned)hi * x_cstride + (unsigned)colc);   // bytes
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, (int)(4u * (unsigned)Cin * x_cstride), 0x00020000);
    const f32x4* const wbase = reinterpret_cast<const f32x4*>(p.w + (int64_t)co_tile * n_cb * A_FLOATS);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // B ring: step g (global MFMA step index) uses channel ci = 2g + hi
    f32x2 bq[PF];
    // step g reads channel rows 2g (lanes 0-31) and 2g+1 (lanes 32-63): the row pair is a scalar offset (clamped to the last channel
    // pair -- rows past Cin meet zero weights)
    const int last_pair = Cin / 2 - 1;                            // (the launcher admits even Cin only)
    const unsigned pair_bytes = 8u * x_cstride;
    auto load_b = [&](int g) __attribute__((always_inline)) {
        return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xrs, x_lane_off, (unsigned)min(g, last_pair) * pair_bytes, 0));
    };
#pragma unroll
    for (int u = 0; u < PF; ++u) bq[u] = load_b(u);

    // A tile 0
    f32x4 ra[NA];
#pragma unroll
    for (int n = 0; n < NA; ++n) {
        const int idx = tid + 256 * n;
        if ((A_VEC % 256 == 0) || idx < A_VEC) reinterpret_cast<f32x4*>(As[0])[idx] = wbase[idx];
    }
    __syncthreads();

    for (int cb = 0; cb < n_cb; ++cb) {
        const int cur = cb & 1;
        const bool more = cb + 1 < n_cb;
        if (more) {
            const f32x4* src = wbase + (size_t)(cb + 1) * A_VEC;
#pragma unroll
            for (int n = 0; n < NA; ++n) {
                const unsigned idx = (unsigned)(tid + 256 * n);
                ra[n] = src[(A_VEC % 256 == 0) ? idx : min(idx, (unsigned)(A_VEC - 1))];
            }
        }
        const float* Ac = As[cur] + hi * BM + nc_a_lane_off<TM>(l31);
        // A step = the A fragments of the NEXT step, its own matrix-core instructions, THEN the refill of the ring slot they just
        // consumed (in place) -- fenced with sched_barrier so the order survives the scheduler.  Written as "copy the slot, refill it,
        // multiply" the refill lands in a fresh register that must be copied back into the loop-carried slot before the back edge, and
        // the copy waits for the read it was meant to overlap (`s_waitcnt vmcnt(1)` two steps after the issue); left to itself the
        // scheduler also sinks the refills of a ring turn into one batch behind its last step and the wave then waits for the
        // batch's first read a few instructions after issuing it (found in the ISA, round 4; sched_group_barrier pins only some
        // instances).
        float a[TM];
        nc_load_a_frag<TM>(Ac, l31, a);
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) {
            float an[TM];
            if (kp + 1 < KP) nc_load_a_frag<TM>(Ac + 2 * (kp + 1) * BM, l31, an);   // the next step's A fragments, ahead of this step's multiplies
            __builtin_amdgcn_sched_barrier(0);
            f32x2 bv = bq[kp % PF];
            if constexpr (INM) {
                if (in_mode & 1) {
                    const float2 gb = Gt[2 * (cb * KP + kp) + hi];
                    bv[0] = ((bv[0] - in_mu) * in_rs) * gb.x + gb.y;
                    bv[1] = ((bv[1] - in_mu) * in_rs) * gb.x + gb.y;
                }
                if (in_mode & 2) {
                    bv[0] = nc_eluf(bv[0]);
                    bv[1] = nc_eluf(bv[1]);
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bv[0], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bv[1], acc[i][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            bq[kp % PF] = load_b(cb * KP + kp + PF);   // unconditional (clamped): the step is one basic block
            __builtin_amdgcn_sched_barrier(0);
            if (kp + 1 < KP) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = an[i];
            }
        }
        if (more) {
#pragma unroll
            for (int n = 0; n < NA; ++n) {
                const int idx = tid + 256 * n;
                if ((A_VEC % 256 == 0) || idx < A_VEC) reinterpret_cast<f32x4*>(As[cur ^ 1])[idx] = ra[n];
            }
        }
        __syncthreads();
    }

    // ---- epilogue: D[row = (r&3) + 8*(r>>2) + 4*hi] for this lane's TN consecutive columns.  Phase A issues every global read
    //      (residual tile, noise) and folds it into the accumulators; phase B only stores.  A global read issued after a store waits
    //      for all earlier stores to be acknowledged, so the per-row operands (bias, Snake alpha) come from the LDS table Ep.
    const int rows_total = p.Cout - co_tile * BM;   // uniform
    if constexpr (INM) {
        // GroupNorm(1,C) block sums of the output (nc_gn.h), emitted from the accumulators.  Column of (lane, j) = 2*l31 + j inside the
        // wave's 64 columns: lanes l31 < 16 hold the first 32-column block, l31 >= 16 the second, and the canonical column tree
        // (xor 16, 8, 4, 2, 1 of the column index) is lane xor 8, 4, 2, 1 followed by the in-lane sum over j.
        if (p.gn_part != nullptr) {
            double* const gp = p.gn_part + (int64_t)b * p.gn_nrb * p.gn_ncb * 2;
            const bool colok = col < T;   // (T is even: both columns of a lane are in or out together)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                double a1[TN], a2[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) {
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
                // column tree low bit first: bit 0 of the column is j (in-lane), bits 1..4 are lane bits 0..3, then the lane half
                double s1 = a1[0] + a1[1], s2 = a2[0] + a2[1];
                nc_gn_butterfly_row(s1, s2);
                s1 = nc_gn_swap_add<true>(s1);
                s2 = nc_gn_swap_add<true>(s2);
                const int rbk = co_tile * TM + i, cbk = ((t_tile * BN + wave * BNW) >> 5) + (l31 >> 4);
                if ((lane & 47) == 0 && rbk < p.gn_nrb && cbk < p.gn_ncb)   // lanes 0 and 16
                    nc_gn_store_partial(gp + ((int64_t)rbk * p.gn_ncb + cbk) * 2, s1, s2);
                __builtin_amdgcn_sched_barrier(0);   // one row block at a time (register budget of the 5-waves-per-SIMD instances)
            }
            if (p.gn_count != nullptr)
                nc_gn_arrive_and_finish(gp, p.gn_count + b, p.gn_stats + 2 * b, p.gn_nrb * p.gn_ncb, (unsigned)(p.n_co_tiles * p.n_t_tiles), p.gn_n);
        }
    }
    if (col >= T) return;
    const int64_t tile_base = (int64_t)b * p.y_bstride + (int64_t)co_tile * BM * p.y_cstride;
    const unsigned cstride = (unsigned)p.y_cstride;
    const unsigned lane_off = (unsigned)(4 * hi) * cstride + (unsigned)col;
    const int rows_left = rows_total - 4 * hi;
    // rows_tag: all BM rows of the tile are inside the tensor (row addresses are then uniform base + 32-bit lane offset)
    auto run_epilogue = [&](auto res_tag, auto snake_tag, auto noise_tag, auto rows_tag) __attribute__((always_inline)) {
        constexpr bool RES = decltype(res_tag)::value, SNAKE = decltype(snake_tag)::value, NOISE = decltype(noise_tag)::value;
        f32x2 nz = {0.0f, 0.0f};
        if constexpr (NOISE) nz = *reinterpret_cast<const f32x2*>(p.noise + (int64_t)b * p.noise_bstride + col);
        const float* const rt = p.res + tile_base;
        // residual rows arrive a quad (4 rows) at a time, one quad ahead of the arithmetic, no store in between
        f32x2 rs[2][4];
        unsigned loff = lane_off;
        auto load_quad = [&](int q, f32x2 (&dst)[4]) __attribute__((always_inline)) {   // q = 4*i + rq
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int R = (q >> 2) * 32 + rr + 8 * (q & 3);
                dst[rr] = *reinterpret_cast<const f32x2*>(rt + (size_t)R * cstride + loff);
            }
        };
        if constexpr (RES) load_quad(0, rs[0]);
        nc_static_for_1x1<4 * TM>([&](auto qt) __attribute__((always_inline)) {
            constexpr int q = decltype(qt)::value, i = q >> 2, rq = q & 3;
            if constexpr (RES && q + 1 < 4 * TM) {
                // the reads of quad q+1 wait (data dependence through the offset) for the arithmetic of quad q-1: two quads in
                // flight, instead of the whole tile's reads hoisted to the top at 2 registers each
                if constexpr (q >= 1) asm volatile("" : "+v"(loff) : "v"(acc[(q - 1) >> 2][0][4 * ((q - 1) & 3) + 3]));
                load_quad(q + 1, rs[(q + 1) & 1]);
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int r = 4 * rq + rr;
                const float bias = Ep[i * 32 + rr + 8 * rq + 4 * hi];
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    float v = acc[i][j][r] + bias;
                    if constexpr (NOISE) v = rs[q & 1][rr][j] + nz[j] * v;
                    else if constexpr (RES) v = v + rs[q & 1][rr][j];
                    acc[i][j][r] = v;
                }
            }
        });
        float* const yt = p.y + tile_base;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int R = i * 32 + (r & 3) + 8 * (r >> 2);
                f32x2 v = {acc[i][0][r], acc[i][1][r]};
                if constexpr (SNAKE) {   // Snake of the consuming layer, fused into the store
                    const float ao = Ep[BM + R + 4 * hi], ao_inv = Ep[2 * BM + R + 4 * hi];
                    v = nc_snakef2(v, nc_f2{ao, ao}, nc_f2{ao_inv, ao_inv});   // two values per packed instruction (nc_math.h)
                }
                float* rowp = yt + (size_t)R * cstride;
                *reinterpret_cast<f32x2*>(rowp + lane_off) = v;
            }
    };
    using res_t = std::integral_constant<bool, (MODE & 5) != 0>;
    using snake_t = std::integral_constant<bool, (MODE & 2) != 0>;
    using noise_t = std::integral_constant<bool, (MODE & 4) != 0>;
    if (rows_total >= BM) {
        run_epilogue(res_t{}, snake_t{}, noise_t{}, std::true_type{});
        return;
    }
    // Row-partial tile (Cout not a multiple of BM: the narrow layers): rows go out in quads, the residual reads of a quad issued
    // before its first store (one memory round trip per quad).
    {
        constexpr bool RES = res_t::value, SNAKE = snake_t::value, NOISE = noise_t::value;
        f32x2 nz = {0.0f, 0.0f};
        if constexpr (NOISE) nz = *reinterpret_cast<const f32x2*>(p.noise + (int64_t)b * p.noise_bstride + col);
        const float* const rt = p.res + tile_base;
        float* const yt = p.y + tile_base;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                f32x2 rs[4];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int R = i * 32 + rr + 8 * rq;
                    rs[rr] = (RES && R < rows_left) ? *reinterpret_cast<const f32x2*>(rt + (size_t)R * cstride + lane_off) : f32x2{0.0f, 0.0f};
                }
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int R = i * 32 + rr + 8 * rq;
                    if (R >= rows_left) continue;
                    const int r = 4 * rq + rr;
                    const float bias = Ep[R + 4 * hi];
                    f32x2 v = {acc[i][0][r] + bias, acc[i][1][r] + bias};
                    if constexpr (NOISE) {
                        v[0] = rs[rr][0] + nz[0] * v[0];
                        v[1] = rs[rr][1] + nz[1] * v[1];
                    } else if constexpr (RES) {
                        v[0] = v[0] + rs[rr][0];
                        v[1] = v[1] + rs[rr][1];
                    }
                    if constexpr (SNAKE) {
                        const float ao = Ep[BM + R + 4 * hi], ao_inv = Ep[2 * BM + R + 4 * hi];
                        v = nc_snakef2(v, nc_f2{ao, ao}, nc_f2{ao_inv, ao_inv});   // two values per packed instruction (nc_math.h)
                    }
                    *reinterpret_cast<f32x2*>(yt + (size_t)R * cstride + lane_off) = v;
                }
            }
    }
}

typedef void (*conv_kernel_fn)(const ConvArgs);

#ifdef NC_EXPERIMENTS   // measured-and-overtaken variant: `make EXPERIMENTS=1` only
// Streaming pointwise convolution for the narrow long rows (Cin <= 192: the SNAC residual / noise / attention projections at
// 5 s x 44.1 kHz, the DAC C = 192 units): these layers move 12*C bytes per 2*C*C flops -- 16-32 flop/B, the HBM side of the machine
// balance -- and the tile-per-workgroup kernel above reaches 3.0-3.5 TB/s on them: its B ring holds 4 steps (2 KB per wave) and
// every 16 channels cost a workgroup barrier and a weight-tile hand-over.  Here the WHOLE weight tile of the row tile (BM x Cin
// floats = the packed image of the row tile, <= 72 KB) is loaded into LDS once per workgroup and the workgroup walks many column
// tiles: the loop has no barrier at all (waves run free), the B ring is 16 steps deep (8 KB per wave in flight) and runs ACROSS
// column tiles -- the first steps of tile k+1 are read before the stores of tile k, so no read waits behind a store acknowledgement.
// Arithmetic unchanged: per output one chain over ci ascending from +0, + bias, + residual (or residual + noise * .), Snake.
// Schedule note (round 4): the compiler issues the 16 refills of a group as one batch behind its last step and opens the next group
// with `s_waitcnt vmcnt(0)`.  The "proper" ring -- refill in place behind each step, `vmcnt(15)` per step, units and groups in ONE loop
// so that the wait-count pass keeps the in-order counter across the epilogue's stores -- was built and measures 2-4 % SLOWER on the
// C = 192 layers (777 -> 814 us; a 32-step ring the same): with two waves per SIMD the batched form lets one wave run its 16 steps
// uninterrupted while the other waits for its batch, and that coarse alternation feeds the matrix pipe better than two streams
// interleaved step by step.  conv1x1_kernel above (3-5 waves per SIMD, tile per workgroup) gains 3-5 % from the in-place ring, and with
// buffer-load addressing on top it overtook this kernel on these very layers (1-6 % per layer in steady state, DAC conv_k1 class
// 3.92 -> 3.79 ms): since then this kernel runs only under NC_PW_STREAM=1 (tests/test_children_gpu.py keeps it held to the oracle).
template <int TM, int MODE>
__global__ __launch_bounds__(256, 2) void conv1x1_stream_kernel(const ConvArgs p) {
    constexpr int TN = 2, BM = 32 * TM, BNW = 32 * TN, BN = 4 * BNW;
    constexpr int PF = (TM == 3 && MODE == 3) ? 8 : 16;   // (residual + Snake at 96 rows: a 16-step ring spills)
    constexpr bool RES = (MODE & 5) != 0, SNAKE = (MODE & 2) != 0, NOISE = (MODE & 4) != 0;
    extern __shared__ __attribute__((aligned(16))) float sa[];   // [Cin][BM] packed weights | [3][BM] bias, alpha, 1/alpha
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Cin = p.Cin, KS = Cin >> 1, T = p.Tout;
    const int n_co = p.n_co_tiles, n_t = p.n_t_tiles;
    const int co_tile = blockIdx.x % n_co, slot = blockIdx.x / n_co, stride = gridDim.x / n_co;   // host: gridDim.x % n_co == 0
    const int units = p.B * n_t;
    float* const Ep = sa + Cin * BM;
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(p.w + (int64_t)co_tile * Cin * BM);
        f32x4* dst = reinterpret_cast<f32x4*>(sa);
        const int nv = Cin * BM / 4;
        for (int i0 = 0; i0 < nv; i0 += 8 * 256) {
            f32x4 r[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) r[u] = src[min(i0 + tid + 256 * u, nv - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + tid + 256 * u < nv) dst[i0 + tid + 256 * u] = r[u];
        }
        for (int i = tid; i < BM; i += 256) {
            const int co = co_tile * BM + i;
            const float ao = p.alpha_out ? p.alpha_out[co] : 0.0f;
            Ep[i] = p.bias ? p.bias[co] : 0.0f;
            Ep[BM + i] = ao;
            Ep[2 * BM + i] = nc_snake_inv(ao);
        }
    }
    __syncthreads();
    if (slot >= units) return;
    const unsigned x_cstride = (unsigned)p.x_cstride, cstride = (unsigned)p.y_cstride;
    const float* const Ab = sa + hi * BM + nc_a_lane_off<TM>(l31);
    // per-unit addressing: unit u -> clip b = u / n_t, column tile t = u % n_t
    auto unit_x = [&](int u, const float*& xb, unsigned& xoff, int& col, int& bclip) __attribute__((always_inline)) {
        const int b = u / n_t, t = u - b * n_t;
        col = t * BN + wave * BNW + TN * l31;
        const int colc = min(col, T - TN);
        xb = p.x + (int64_t)b * p.x_bstride;
        xoff = (unsigned)hi * x_cstride + (unsigned)colc;
        bclip = b;
    };
    const float* xb; unsigned xoff; int col, bclip;
    unit_x(slot, xb, xoff, col, bclip);
    f32x2 bq[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) bq[u] = *reinterpret_cast<const f32x2*>(xb + (size_t)(2 * u) * x_cstride + xoff);
    for (int unit = slot; unit < units; unit += stride) {
        // the unit after this one (its first PF steps are read during this unit's last PF steps)
        const int nunit = unit + stride;
        const float* xbn = xb; unsigned xoffn = xoff; int coln = col, bn = bclip;
        if (nunit < units) unit_x(nunit, xbn, xoffn, coln, bn);
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        for (int g0 = 0; g0 < KS; g0 += PF) {
            const bool tail = g0 + PF >= KS;                       // the refills of this group belong to the next unit
            const float* rb = tail ? xbn : xb;
            const unsigned ro = tail ? xoffn : xoff;
            const int gb = tail ? 0 : g0 + PF;
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                float a[TM];
                nc_load_a_frag<TM>(Ab + 2 * (g0 + u) * BM, l31, a);
                const f32x2 bv = bq[u];
                bq[u] = *reinterpret_cast<const f32x2*>(rb + (size_t)(2 * (gb + u)) * x_cstride + ro);   // (last unit: re-reads its own rows, unused)
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bv[0], acc[i][0], 0, 0, 0);
                    acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bv[1], acc[i][1], 0, 0, 0);
                }
            }
        }
        // ---- epilogue of this unit (full row tiles only: host).  Phase A: every global read (residual quads, noise), folded into the
        //      accumulators; phase B: stores.
        if (col < T) {
            const int64_t tile_base = (int64_t)bclip * p.y_bstride + (int64_t)co_tile * BM * p.y_cstride;
            const unsigned lane_off = (unsigned)(4 * hi) * cstride + (unsigned)col;
            f32x2 nz = {0.0f, 0.0f};
            if constexpr (NOISE) nz = *reinterpret_cast<const f32x2*>(p.noise + (int64_t)bclip * p.noise_bstride + col);
            const float* const rt = p.res + tile_base;
            f32x2 rs[2][4];
            unsigned loff = lane_off;
            auto load_quad = [&](int q, f32x2 (&dst)[4]) __attribute__((always_inline)) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int R = (q >> 2) * 32 + rr + 8 * (q & 3);
                    dst[rr] = *reinterpret_cast<const f32x2*>(rt + (size_t)R * cstride + loff);
                }
            };
            if constexpr (RES) load_quad(0, rs[0]);
            nc_static_for_1x1<4 * TM>([&](auto qt) __attribute__((always_inline)) {
                constexpr int q = decltype(qt)::value, i = q >> 2, rq = q & 3;
                if constexpr (RES && q + 1 < 4 * TM) {
                    if constexpr (q >= 1) asm volatile("" : "+v"(loff) : "v"(acc[(q - 1) >> 2][0][4 * ((q - 1) & 3) + 3]));
                    load_quad(q + 1, rs[(q + 1) & 1]);
                }
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int r = 4 * rq + rr;
                    const float bias = Ep[i * 32 + rr + 8 * rq + 4 * hi];
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        float v = acc[i][j][r] + bias;
                        if constexpr (NOISE) v = rs[q & 1][rr][j] + nz[j] * v;
                        else if constexpr (RES) v = v + rs[q & 1][rr][j];
                        acc[i][j][r] = v;
                    }
                }
            });
            float* const yt = p.y + tile_base;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int R = i * 32 + (r & 3) + 8 * (r >> 2);
                    f32x2 v = {acc[i][0][r], acc[i][1][r]};
                    if constexpr (SNAKE) {
                        const float ao = Ep[BM + R + 4 * hi], ao_inv = Ep[2 * BM + R + 4 * hi];
                        v = nc_snakef2(v, nc_f2{ao, ao}, nc_f2{ao_inv, ao_inv});   // two values per packed instruction (nc_math.h)
                    }
                    *reinterpret_cast<f32x2*>(yt + (size_t)R * cstride + lane_off) = v;
                }
        }
        xb = xbn; xoff = xoffn; col = coln; bclip = bn;
    }
}

template <int TM>
static conv_kernel_fn conv1x1_stream_by_mode(int mode) {
    switch (mode) {
        case 0: return &conv1x1_stream_kernel<TM, 0>;
        case 1: return &conv1x1_stream_kernel<TM, 1>;
        case 2: return &conv1x1_stream_kernel<TM, 2>;
        case 3: return &conv1x1_stream_kernel<TM, 3>;
        case 4: return &conv1x1_stream_kernel<TM, 4>;
    }
    return nullptr;
}
// streaming variant (weights resident in LDS): TM = 2 / 3, modes 0..4
conv_kernel_fn conv1x1_stream_kernel_table(int TM, int mode) {
    switch (TM) {
        case 2: return conv1x1_stream_by_mode<2>(mode);
        case 3: return conv1x1_stream_by_mode<3>(mode);
    }
    return nullptr;
}
#endif  // NC_EXPERIMENTS

// Skinny projection: 1x1 conv with Cout <= 16 (the RVQ in_proj 1024 -> 8, VectorQuantizer.cs:47,76) over N = B*T frames.  The
// generic tile would run one workgroup per clip with a barrier every 16 channels; here one wavefront owns 16 frames and walks the
// whole reduction with v_mfma_f32_16x16x4_f32 (exact k-ordered chain), weights pre-packed [Cin/4][64 lanes] (rows >= Cout are 0),
// loads issued a 16-step register chunk ahead.
typedef float f32x4s __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void skinny_proj_kernel(const float* __restrict__ x, int64_t x_bstride, int64_t x_cstride,
                                                         const float* __restrict__ wp, const float* __restrict__ bias, float* __restrict__ y,
                                                         int64_t y_bstride, int64_t y_cstride, int B, int Cin, int Cout, int64_t T) {
    constexpr int CH = 16;
    const int lane = threadIdx.x;
    const int k4 = lane >> 4, cl = lane & 15;
    const int64_t total = (int64_t)B * T;
    const int64_t f = min((int64_t)blockIdx.x * 16 + cl, total - 1);
    const int64_t b = f / T, t = f - b * T;
    const float* xp = x + b * x_bstride + (int64_t)k4 * x_cstride + t;
    const int64_t xs = 4 * x_cstride;
    const int KS = Cin / 4;
    f32x4s acc = {0.0f, 0.0f, 0.0f, 0.0f};
    float a0[CH], b0[CH], a1[CH], b1[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) { a0[i] = wp[(int64_t)i * 64 + lane]; b0[i] = xp[(int64_t)i * xs]; }
    for (int kc = 0; kc < KS; kc += CH) {
        const bool more = kc + CH < KS;
        if (more) {
#pragma unroll
            for (int i = 0; i < CH; ++i) { a1[i] = wp[(int64_t)(kc + CH + i) * 64 + lane]; b1[i] = xp[(int64_t)(kc + CH + i) * xs]; }
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i], b0[i], acc, 0, 0, 0);
        if (more) {
#pragma unroll
            for (int i = 0; i < CH; ++i) { a0[i] = a1[i]; b0[i] = b1[i]; }
        }
    }
    if ((int64_t)blockIdx.x * 16 + cl >= total) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int d = 4 * k4 + r;                     // D row of this lane's register r
        if (d < Cout) y[b * y_bstride + (int64_t)d * y_cstride + t] = acc[r] + (bias ? bias[d] : 0.0f);
    }
}

void launch_skinny_proj(const float* x, int64_t x_bstride, int64_t x_cstride, const float* wp, const float* bias, float* y, int64_t y_bstride,
                        int64_t y_cstride, int B, int Cin, int Cout, int64_t T, hipStream_t s) {
    const int64_t total = (int64_t)B * T;
    hipLaunchKernelGGL(skinny_proj_kernel, dim3((unsigned)((total + 15) / 16)), dim3(64), 0, s, x, x_bstride, x_cstride, wp, bias, y, y_bstride,
                       y_cstride, B, Cin, Cout, T);
    NC_HIP(hipGetLastError());
}

template <int TM>
static conv_kernel_fn conv1x1_by_mode(int mode) {
    switch (mode) {
        case 0: return &conv1x1_kernel<TM, 16, 0>;
        case 1: return &conv1x1_kernel<TM, 16, 1>;
        case 2: return &conv1x1_kernel<TM, 16, 2>;
        case 3: return &conv1x1_kernel<TM, 16, 3>;
        case 4: return &conv1x1_kernel<TM, 16, 4>;
        case 8: return &conv1x1_kernel<TM, 16, 0, true>;   // plain epilogue + Encodec input mode
    }
    return nullptr;
}

// mode: bit 0 residual, bit 1 Snake-out, 4 = noise injection (with residual, no Snake), 8 = plain epilogue + Encodec input mode
conv_kernel_fn conv1x1_kernel_table(int TM, int mode) {
    switch (TM) {
        case 1: return conv1x1_by_mode<1>(mode);
        case 2: return conv1x1_by_mode<2>(mode);
        case 3: return conv1x1_by_mode<3>(mode);
        case 4: return conv1x1_by_mode<4>(mode);
    }
    return nullptr;
}

}  // namespace nc
