// Fused two-layer persistent LSTM (Encodec SLSTM, Modules/Encodec/SLSTM.cs:31,40-57; gate order i,f,g,o): ALL T steps of BOTH layers
// for up to two 16-row column tiles in ONE launch.  Arithmetic = the canonical order of oracle/c/nc_ref_encodec.c slstm():
//   pre = (chain_ih + b_ih) + (((q0 + q1) + (q2 + q3)) + b_hh),  chain_ih = one fma chain over k ascending, q_i = the four quarter chains
//   of the recurrent contraction; sigma, tanh = nc_math.h; c = (f*c) + (i*g); h = o*tanh(c).  v_mfma_f32_16x16x4_f32 is bitwise a
//   k-ordered fmaf chain, so every chain below is that chain.
//
// STATUS: both kernels of this file are bit-exact against the oracle and MEASURED SLOWER than lstm_seq_kernel (nc_encodec.hip), which stays
// the default: NC_LSTM_FUSED=1 selects lstm2_kernel (C3 9.5-10.0 against 9.05-9.15 ms), NC_LSTM_SPLIT=1 lstm1_kernel (10.1-10.2 ms).  The
// in-kernel trace (NC_LSTM2_TRACE, tools/probe/lstm2_trace.py) and the reasons are in DESIGN.md 8, round 4.  Kept as working,
// instrumented starting points and as the record of what the protocol costs.
//
// Why this shape (round 4; the per-layer kernel it would replace, lstm_seq_kernel in nc_encodec.hip, runs 5.7-6.2 us per step):
//   * the step is bound by the exchange of h between the workgroups that share W (4 MB per matrix: no CU can hold it), and the old
//     protocol serialised four memory round trips per step: payload store -> DRAIN (wait for the write-through acknowledgement) -> flag
//     -> poll -> payload read.  Here nothing drains: every exchange word is validated by VALUE.  The exchange regions are never reused
//     (one region per step, pre-filled with a NaN sentinel no gate output can take), so a consumer that sees the sentinel simply
//     reads again; the per-workgroup flags are only a HINT that says when reading is worth it (one wave-wide poll, no sweeps).
//   * on gfx950 a wave's loads queue behind the acknowledgement of its own earlier stores (one in-order counter).  So the roles are
//     split: CHAIN wavefronts only ever load (flags, h), GATE wavefronts take the partial tiles through LDS, apply the gates and are
//     the only ones that store -- and they issue the next step's prefetch loads BEFORE the step's stores.
//   * no workgroup barrier inside the loop: the eight wavefronts of a tile synchronise through LDS words (post / wait), so the two
//     tiles of a launch (wavefronts 0-7 and 8-15 of a workgroup, sharing the weights in LDS) run as independent instruction
//     streams and fill each other's exchange latency on the shared matrix pipes.
//   * layer 1 runs inside the same launch: its input projection W_ih1 h0_t is one 512-long dependent chain (the canonical order),
//     walked by two wavefronts in sequence with the accumulator handed over through LDS, off the critical path (it only has to
//     finish before layer 1 reaches step t); the tensors between the layers, the chunked projection GEMMs and their cross-stream
//     events are gone.
// Workgroup u (of C/4) owns hidden units {16m + 4i + k : i = 0..3}, m = u / 4, k = u % 4, of both layers: its 4 x 16 outputs of a
// step are then ONE contiguous 256-byte run of the B-fragment-major exchange layout [C/16 groups][64 lanes][4], which every consumer
// lane reads with 16-byte loads (4 k-steps of its matrix-core B operand per load).
// Roles of the 8 wavefronts of a tile:   0,1  layer-0 recurrent chains (quarters 0-1 / 2-3)      2,3  layer-1 recurrent chains
//                                        4,5  layer-1 input-projection chain (halves)            6 / 7  gate + publish, layer 0 / 1
// (the second tile rotates the roles by two so that the matrix work of both tiles is even over the four SIMDs).
#include "nc_lstm.h"

#include "nc_common.h"
#include "nc_math.h"

namespace nc {

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* l2_gptr;
typedef __attribute__((address_space(3))) void* l2_lptr;
typedef __attribute__((address_space(1))) unsigned l2_gu32;

namespace {

constexpr int L2_PART = 2 * 2 * 4 * 256;   // [layer][parity][quarter][64 lanes x 4]
constexpr int L2_IH = 2 * 2 * 256;         // [parity][mid | final][64 x 4]
constexpr int L2_HTR = 2 * 64;             // [layer][16 clips x 4 units]: transposes a gate wave's h into 16-byte rows
constexpr int L2_SYNC = 64;                // sync words (unsigned)
constexpr int L2_GROUP = L2_PART + L2_IH + L2_HTR + L2_SYNC;
enum { SY_P0A = 0, SY_P0B, SY_P1A, SY_P1B, SY_IHMID, SY_IHFIN, SY_ACKMID, SY_ACKFIN, SY_DEAD, SY_READY0, SY_READY1 };
constexpr unsigned L2_SPINS = 1u << 22;

struct Ctx {
    __amdgpu_buffer_rsrc_t rs;    // the exchange regions
    const Lstm2Args* a;
    volatile unsigned* sync;      // this tile's sync words
    volatile unsigned* dead;      // one word per workgroup
    int lane, u, tiles, gl;       // gl: tile index within the launch
    int region_bytes;             // C * 16 * 4
    unsigned long long* tr;       // this wave's trace rows (nullable)
    unsigned* tmo;                // timeout word
    int64_t T;
};

__device__ __forceinline__ int region_off(const Ctx& c, int layer, int64_t t) {
    return (int)(((int64_t)layer * c.T + t) * c.tiles + c.gl) * c.region_bytes;
}

__device__ __forceinline__ void stamp(const Ctx& c, int64_t t, int slot) {
    if (c.tr && t >= LSTM2_TRACE_T0 && t < LSTM2_TRACE_T0 + LSTM2_TRACE_STEPS && c.lane == 0)
        c.tr[(t - LSTM2_TRACE_T0) * 4 + slot] = __builtin_amdgcn_s_memrealtime();
}

__device__ __forceinline__ void give_up(const Ctx& c) {
    if (c.lane == 0) {
        __hip_atomic_store((l2_gu32*)c.tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *c.dead = 1u;
    }
}

// LDS hand-offs inside a tile: data writes, then the word (one wave's LDS operations are performed in order); a reader polls the
// word and only then reads the data.
__device__ __forceinline__ void lds_post(volatile unsigned* f, unsigned v) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    *f = v;
}
__device__ __forceinline__ bool lds_wait_ge(const Ctx& c, volatile unsigned* f, unsigned v) {
    for (unsigned spins = 0; spins < L2_SPINS; ++spins) {
        if (*f >= v) {
            asm volatile("" ::: "memory");
            return true;
        }
        if ((spins & 255u) == 255u && *c.dead) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    give_up(c);
    return false;
}

// hint: have the n (<= 128) producers of a layer's h published step `want - 1`?  ONE wave per tile and layer polls (the gate wave, after
// its own publish) and hands the answer to the chain waves through LDS: with every chain wave polling for itself, 1536 polling waves
// kept the few lines that hold the flags so busy that a poll took 2-3 us and the workgroups drifted 4 us apart (trace, round 4).
__device__ __forceinline__ bool poll_flags(const Ctx& c, const unsigned* f, int n, unsigned want) {
    for (unsigned spins = 0; spins < L2_SPINS; ++spins) {
        unsigned v = c.lane < n ? __hip_atomic_load((l2_gu32*)(f + c.lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : want;
        if (c.lane + 64 < n) v = min(v, __hip_atomic_load((l2_gu32*)(f + c.lane + 64), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (__all(v >= want)) return true;
        if ((spins & 255u) == 255u &&
            (*c.dead || __hip_atomic_load((l2_gu32*)c.tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) return false;
        __builtin_amdgcn_s_sleep(4);
    }
    give_up(c);
    return false;
}

// NGH 16-byte operand rows per lane, write-through loads; valid once no word is the sentinel any more (the producers' payload stores
// are not drained before their flag, so the flag can be early: read again)
template <int NGH>
__device__ __forceinline__ bool load_valid(const Ctx& c, int off, u32x4v (&hb)[NGH]) {
    for (unsigned spins = 0; spins < L2_SPINS; ++spins) {
#pragma unroll
        for (int g = 0; g < NGH; ++g) hb[g] = __builtin_amdgcn_raw_buffer_load_b128(c.rs, off + g * 1024, 0, 16 /* sc1 */);
        bool ok = true;
#pragma unroll
        for (int g = 0; g < NGH; ++g)
            ok = ok && hb[g].x != LSTM2_SENTINEL && hb[g].y != LSTM2_SENTINEL && hb[g].z != LSTM2_SENTINEL && hb[g].w != LSTM2_SENTINEL;
        if (__all(ok)) return true;
        if ((spins & 63u) == 63u &&
            (*c.dead || __hip_atomic_load((l2_gu32*)c.tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) return false;
        __builtin_amdgcn_s_sleep(2);
    }
    give_up(c);
    return false;
}

__device__ __forceinline__ f32x4v mfma4(f32x4v a4, u32x4v b4, f32x4v acc) {
    const f32x4v b = __builtin_bit_cast(f32x4v, b4);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[e], b[e], acc, 0, 0, 0);
    return acc;
}

// Recurrent chains of one layer, half w (quarters 2w and 2w + 1, two independent accumulators side by side)
template <int NG>
__device__ __forceinline__ void chain_role(const Ctx& c, const f32x4v* W, float* part, int layer, int w, volatile unsigned* posted) {
    constexpr int NGH = NG / 2, NGQ = NG / 4;
    volatile unsigned* ready = c.sync + (layer == 0 ? SY_READY0 : SY_READY1);
    const f32x4v* Wl = W + (w * NGH) * 64 + c.lane;
    const int64_t T = c.a->T;
    __builtin_amdgcn_s_setprio(2);
    for (int64_t t = 0; t < T; ++t) {
        f32x4v acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
        if (t > 0) {                                                  // h_{-1} = 0: every quarter chain of step 0 is +0
            stamp(c, t, 0);
            if (w == 0) {                                             // the polling wave of a tile and layer is a chain wave (load-only)
                if (!poll_flags(c, c.a->flags + ((int64_t)c.gl * 2 + layer) * (NG * 4), NG * 4, (unsigned)t)) return;
                lds_post(ready, (unsigned)t);
            } else if (!lds_wait_ge(c, ready, (unsigned)t)) return;
            stamp(c, t, 1);
            u32x4v hb[NGH];
            if (!load_valid<NGH>(c, region_off(c, layer, t - 1) + (w * NGH * 64 + c.lane) * 16, hb)) return;
            stamp(c, t, 2);
#pragma unroll
            for (int g = 0; g < NGQ; ++g) {                            // the two chains alternate instruction by instruction
                const f32x4v a0 = Wl[g * 64], a1 = Wl[(NGQ + g) * 64];
                const f32x4v b0 = __builtin_bit_cast(f32x4v, hb[g]), b1 = __builtin_bit_cast(f32x4v, hb[NGQ + g]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], b0[e], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], b1[e], acc1, 0, 0, 0);
                }
            }
        }
        f32x4v* P = reinterpret_cast<f32x4v*>(part + ((layer * 2 + (int)(t & 1)) * 4 + 2 * w) * 256) + c.lane;
        P[0] = acc0;
        P[64] = acc1;
        lds_post(posted, (unsigned)(t + 1));
        stamp(c, t, 3);
    }
    if (layer == 0 && w == 0) {                                        // the projection chain of layer 1 still waits for the last h0
        if (!poll_flags(c, c.a->flags + ((int64_t)c.gl * 2 + layer) * (NG * 4), NG * 4, (unsigned)T)) return;
        lds_post(ready, (unsigned)T);
    }
}

// Input projection of layer 1, chain_ih = sum_k W_ih1[., k] h0_s[k] as ONE chain: wave A walks k < C/2 and hands the accumulator to
// wave B, which finishes it for the gate wave.  Double-buffered by step parity with explicit acknowledgements: only layer 0 feeds
// these waves, so nothing else stops them from running ahead of their consumer.
template <int NG>
__device__ __forceinline__ void ih_role(const Ctx& c, const f32x4v* W, float* ih, int w) {
    constexpr int NGH = NG / 2;
    const f32x4v* Wl = W + (w * NGH) * 64 + c.lane;
    const int64_t T = c.a->T;
    for (int64_t s = 0; s < T; ++s) {
        stamp(c, s, 0);
        if (!lds_wait_ge(c, c.sync + SY_READY0, (unsigned)(s + 1))) return;
        stamp(c, s, 1);
        u32x4v hb[NGH];
        if (!load_valid<NGH>(c, region_off(c, 0, s) + (w * NGH * 64 + c.lane) * 16, hb)) return;
        stamp(c, s, 2);
        f32x4v* mid = reinterpret_cast<f32x4v*>(ih + ((int)(s & 1) * 2 + 0) * 256) + c.lane;
        f32x4v* fin = reinterpret_cast<f32x4v*>(ih + ((int)(s & 1) * 2 + 1) * 256) + c.lane;
        f32x4v acc = {0.0f, 0.0f, 0.0f, 0.0f};
        if (w == 1) {
            if (!lds_wait_ge(c, c.sync + SY_IHMID, (unsigned)(s + 1))) return;
            acc = *mid;
            lds_post(c.sync + SY_ACKMID, (unsigned)(s + 1));
        }
#pragma unroll
        for (int g = 0; g < NGH; ++g) acc = mfma4(Wl[g * 64], hb[g], acc);
        if (w == 0) {
            if (s >= 2 && !lds_wait_ge(c, c.sync + SY_ACKMID, (unsigned)(s - 1))) return;   // step s - 2 has been taken over
            *mid = acc;
            lds_post(c.sync + SY_IHMID, (unsigned)(s + 1));
        } else {
            if (s >= 2 && !lds_wait_ge(c, c.sync + SY_ACKFIN, (unsigned)(s - 1))) return;
            *fin = acc;
            lds_post(c.sync + SY_IHFIN, (unsigned)(s + 1));
        }
        stamp(c, s, 3);
    }
}

// Gates + publish of one layer: lane (i = lane >> 4, n = lane & 15) holds the four gate pre-activations of unit 16m + 4i + k, clip n
// (the accumulator layout of v_mfma_f32_16x16x4_f32 with tile row = 4 * unit + gate).
template <int NG>
__device__ __forceinline__ void gate_role(const Ctx& c, float* part, float* ih, float* htr, int layer) {
    const Lstm2Args& a = *c.a;
    const int C = a.C, N = a.N;
    const int64_t T = a.T;
    const int m = c.u >> 2, kap = c.u & 3, io = c.lane >> 4, n = c.lane & 15;
    const int j = 16 * m + 4 * io + kap;                                // this lane's hidden unit
    const int b = (a.tile0 + c.gl) * 16 + n, bb = min(b, N - 1);
    const float* bh = layer == 0 ? a.bhh0 : a.bhh1;
    const float bh0 = bh[j], bh1 = bh[C + j], bh2 = bh[2 * C + j], bh3 = bh[3 * C + j];
    float bi0 = 0.0f, bi1 = 0.0f, bi2 = 0.0f, bi3 = 0.0f;
    if (layer == 1) { bi0 = a.bih1[j]; bi1 = a.bih1[C + j]; bi2 = a.bih1[2 * C + j]; bi3 = a.bih1[3 * C + j]; }
    // layer 0: the input projections [4C][T][N] incl. b_ih; layer 1: the skip operand x [N,C,T]; both one step ahead of their use
    const float* g = a.gi0 + bb;
    const int64_t gr = T * (int64_t)N;
    const float* sk = a.skip + ((int64_t)bb * C + j) * T;
    float* orow = a.out + ((int64_t)bb * C + j) * T;
    float g0 = 0.0f, g1 = 0.0f, g2 = 0.0f, g3 = 0.0f, skv = 0.0f;
    if (layer == 0) { g0 = g[(int64_t)j * gr]; g1 = g[(int64_t)(C + j) * gr]; g2 = g[(int64_t)(2 * C + j) * gr]; g3 = g[(int64_t)(3 * C + j) * gr]; }
    else skv = sk[0];
    unsigned* flag = a.flags + ((int64_t)c.gl * 2 + layer) * (NG * 4) + c.u;
    volatile unsigned* pa = c.sync + (layer == 0 ? SY_P0A : SY_P1A);
    volatile unsigned* pb = c.sync + (layer == 0 ? SY_P0B : SY_P1B);
    float cst = 0.0f;
    __builtin_amdgcn_s_setprio(3);
    for (int64_t t = 0; t < T; ++t) {
        stamp(c, t, 0);
        if (a.trace && layer == 0 && c.u == 0 && c.gl == 0 && c.lane == 0 && (t == LSTM2_TRACE_T0 || t == LSTM2_TRACE_T0 + LSTM2_TRACE_STEPS - 1)) {
            unsigned long long* ck = a.trace + (int64_t)(C / 4) * a.tiles * 8 * LSTM2_TRACE_STEPS * 4 + (t == LSTM2_TRACE_T0 ? 0 : 2);
            ck[0] = __builtin_readcyclecounter();                       // shader clock
            ck[1] = __builtin_amdgcn_s_memrealtime();                   // 100 MHz
        }
        if (!lds_wait_ge(c, pa, (unsigned)(t + 1)) || !lds_wait_ge(c, pb, (unsigned)(t + 1))) return;
        stamp(c, t, 1);
        const f32x4v* P = reinterpret_cast<const f32x4v*>(part + ((layer * 2 + (int)(t & 1)) * 4) * 256) + c.lane;
        const f32x4v q0 = P[0], q1 = P[64], q2 = P[128], q3 = P[192];
        f32x4v r;
#pragma unroll
        for (int v = 0; v < 4; ++v) r[v] = (q0[v] + q1[v]) + (q2[v] + q3[v]);
        float pi, pf, pg, po;
        if (layer == 0) {
            pi = g0 + (r[0] + bh0); pf = g1 + (r[1] + bh1); pg = g2 + (r[2] + bh2); po = g3 + (r[3] + bh3);
        } else {
            if (!lds_wait_ge(c, c.sync + SY_IHFIN, (unsigned)(t + 1))) return;
            const f32x4v x = *(reinterpret_cast<const f32x4v*>(ih + ((int)(t & 1) * 2 + 1) * 256) + c.lane);
            lds_post(c.sync + SY_ACKFIN, (unsigned)(t + 1));
            pi = (x[0] + bi0) + (r[0] + bh0); pf = (x[1] + bi1) + (r[1] + bh1); pg = (x[2] + bi2) + (r[2] + bh2); po = (x[3] + bi3) + (r[3] + bh3);
        }
        const float ig = nc_sigmoidf(pi), fg = nc_sigmoidf(pf), gg = nc_tanhf(pg), og = nc_sigmoidf(po);
        cst = (fg * cst) + (ig * gg);
        const float h = og * nc_tanhf(cst);
        stamp(c, t, 2);
        // 4 units x 16 clips -> one 16-byte row per clip (units i = 0..3 of clip n adjacent): the B-fragment order of the exchange
        float* tr = htr + layer * 64;
        tr[n * 4 + io] = h;
        const float yo = layer == 1 ? h + skv : 0.0f;
        const int64_t tn = t + 1 < T ? t + 1 : t;
        // the next step's operands: loads issued BEFORE this step's stores, so they never wait for a store acknowledgement
        if (layer == 0) { g0 = g[(int64_t)j * gr + tn * N]; g1 = g[(int64_t)(C + j) * gr + tn * N]; g2 = g[(int64_t)(2 * C + j) * gr + tn * N]; g3 = g[(int64_t)(3 * C + j) * gr + tn * N]; }
        else skv = sk[tn];
        asm volatile("" ::: "memory");
        if (layer == 0 || t + 1 < T) {                                  // (layer 1's last h has no consumer; layer 0's feeds the projection chain)
            if (c.lane < 16) {
                const f32x4v h4 = *reinterpret_cast<const f32x4v*>(tr + c.lane * 4);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, h4), c.rs, region_off(c, layer, t) + ((m * 64 + kap * 16 + c.lane) * 16), 0, 16 /* sc1 */);
            }
            asm volatile("" ::: "memory");                              // payload, then the hint (no drain: consumers validate by value)
            if (c.lane == 0) __hip_atomic_store((l2_gu32*)flag, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (layer == 1 && b < N) orow[t] = a.elu_out ? nc_eluf(yo) : yo;
        stamp(c, t, 3);
    }
}

template <int NG>
__global__ __launch_bounds__(1024, 1) void lstm2_kernel(const Lstm2Args a) {
    extern __shared__ __attribute__((aligned(16))) float l2_lds[];   // [3 images][NG][64][4] | per tile: part, ih, htr, sync
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = 8 * a.tiles;
    float* const Wf = l2_lds;
    constexpr int IMG = NG * 64 * 4;                                   // floats per image of one workgroup
    // the workgroup's three weight images -> LDS, once (LDS DMA, 1 KB per wave-instruction)
    for (int ch = wave; ch < 3 * NG; ch += nwaves) {
        const int img = ch / NG, gq = ch - img * NG;
        const float* src = (img == 0 ? a.whh0 : img == 1 ? a.wih1 : a.whh1) + ((int64_t)blockIdx.x * NG + gq) * 256 + lane * 4;
        __builtin_amdgcn_global_load_lds((l2_gptr)src, (l2_lptr)(Wf + img * IMG + gq * 256), 16, 0, 0);
    }
    const int gl = wave >> 3, gw = wave & 7;
    float* const scratch = l2_lds + 3 * IMG + gl * L2_GROUP;
    volatile unsigned* const sync = reinterpret_cast<volatile unsigned*>(scratch + L2_PART + L2_IH + L2_HTR);
    if (gw == 0 && lane < L2_SYNC) sync[lane] = 0u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                     // the only workgroup barrier of the launch
    Ctx c;
    c.a = &a;
    c.tmo = a.tmo;
    c.T = a.T;
    c.tiles = a.tiles;
    c.region_bytes = a.C * 16 * 4;
    c.rs = __builtin_amdgcn_make_buffer_rsrc(a.S, 0, (int)((int64_t)2 * a.T * a.tiles * a.C * 16 * 4), 0x00020000);
    c.sync = sync;
    c.dead = reinterpret_cast<volatile unsigned*>(l2_lds + 3 * IMG + L2_PART + L2_IH + L2_HTR) + SY_DEAD;   // tile 0's word serves the workgroup
    c.lane = lane;
    c.u = blockIdx.x;
    c.gl = gl;
    c.tr = a.trace ? a.trace + (((int64_t)blockIdx.x * a.tiles + gl) * 8 + (gl == 0 ? gw : ((gw + 2) & 7))) * (LSTM2_TRACE_STEPS * 4) : nullptr;
    float* const part = scratch;
    float* const ih = scratch + L2_PART;
    float* const htr = scratch + L2_PART + L2_IH;
    const f32x4v* const W0 = reinterpret_cast<const f32x4v*>(Wf);
    const f32x4v* const W1 = reinterpret_cast<const f32x4v*>(Wf + IMG);
    const f32x4v* const W2 = reinterpret_cast<const f32x4v*>(Wf + 2 * IMG);
    const int role = gl == 0 ? gw : ((gw + 2) & 7);                     // second tile: heavy roles on the other SIMD pair
    switch (role) {
        case 0: chain_role<NG>(c, W0, part, 0, 0, sync + SY_P0A); break;
        case 1: chain_role<NG>(c, W0, part, 0, 1, sync + SY_P0B); break;
        case 2: chain_role<NG>(c, W2, part, 1, 0, sync + SY_P1A); break;
        case 3: chain_role<NG>(c, W2, part, 1, 1, sync + SY_P1B); break;
        case 4: ih_role<NG>(c, W1, ih, 0); break;
        case 5: ih_role<NG>(c, W1, ih, 1); break;
        case 6: gate_role<NG>(c, part, ih, htr, 0); break;
        default: gate_role<NG>(c, part, ih, htr, 1); break;
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Per-layer form (round 4, second attempt): ONE layer, ONE 16-row column tile per workgroup, the exchange among the C / 16 = 32
// workgroups of a tile as in lstm_seq_kernel (nc_encodec.hip) -- the fused kernel above showed that an exchange among 128 costs more than
// the drain it removes saves -- but with the fused kernel's protocol: load-only chain wavefronts, store-only gate wavefronts, never-reused
// exchange regions validated by value (no drain), one polling wave per workgroup, no workgroup barrier in the loop.  Workgroup m owns the
// 16 hidden units of exchange group m: 8 chain waves (4 unit blocks x 2 halves of the reduction, two quarter chains side by side each)
// and 4 gate waves; the chain waves hand the gate wave q0 + q1 and q2 + q3 (the canonical (q0 + q1) + (q2 + q3) finishes there).
// Same LstmSeq-style arguments as the kernel it replaces: chunked launches resume from carried state, operand layouts by strides.
struct Lstm1Args {
    const float* gi; const float* w; const float* bhh; const float* skip; float* out; int elu_out;
    float* S;              // exchange regions [T][tiles of the CALL][C * 16], sentinel-filled once per call
    unsigned* flags;       // [tiles of the call][C / 4]
    unsigned* tmo; float* cstate;
    int64_t gi_b, gi_c, gi_t, out_b, out_c, out_t;
    int N, C; int64_t T, t0, t1; int tile0, tiles_total;
};
constexpr int L1_PART = 2 * 4 * 2 * 256;   // [parity][unit block][half][64 x 4]
constexpr int L1_HTR = 4 * 64;
enum { S1_P = 0 /* 8 words: [unit block][half] */, S1_READY = 8, S1_DEAD = 9 };

template <int NG>
__global__ __launch_bounds__(768, 1) void lstm1_kernel(const Lstm1Args a) {
    extern __shared__ __attribute__((aligned(16))) float l1_lds[];   // W [4 unit blocks][NG][64][4] | part | htr | sync
    constexpr int IMG = NG * 64 * 4, NGH = NG / 2, NGQ = NG / 4;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = blockIdx.x, tile = a.tile0 + blockIdx.y;
    float* const Wf = l1_lds;
    for (int ch = wave; ch < 4 * NG; ch += 12) {
        const float* src = a.w + ((int64_t)(4 * m) * NG + ch) * 256 + lane * 4;      // images of unit blocks 4m .. 4m + 3 are contiguous
        __builtin_amdgcn_global_load_lds((l2_gptr)src, (l2_lptr)(Wf + ch * 256), 16, 0, 0);
    }
    float* const part = l1_lds + 4 * IMG;
    float* const htr = part + L1_PART;
    volatile unsigned* const sync = reinterpret_cast<volatile unsigned*>(htr + L1_HTR);
    if (wave == 0 && lane < 16) sync[lane] = 0u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    Ctx c;
    c.a = nullptr;
    c.tiles = a.tiles_total;
    c.region_bytes = a.C * 16 * 4;
    c.rs = __builtin_amdgcn_make_buffer_rsrc(a.S, 0, (int)((int64_t)a.T * a.tiles_total * a.C * 16 * 4), 0x00020000);
    c.sync = sync;
    c.dead = sync + S1_DEAD;
    c.lane = lane;
    c.u = m;
    c.gl = tile;
    c.tr = nullptr;
    c.tmo = a.tmo;
    c.T = a.T;
    const int64_t T = a.T;
    unsigned* const flags = a.flags + (int64_t)tile * (NG * 4);
    if (wave < 8) {
        // ---- chain wave: unit block kap, half w of the reduction
        const int kap = wave >> 1, w = wave & 1;
        const f32x4v* Wl = reinterpret_cast<const f32x4v*>(Wf + kap * IMG) + (w * NGH) * 64 + lane;
        __builtin_amdgcn_s_setprio(2);
        for (int64_t t = a.t0; t < a.t1; ++t) {
            f32x4v acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
            if (t > 0) {
                // the polling wave of the workgroup is a CHAIN wave: it never stores, so its polls do not queue behind store
                // acknowledgements (a gate wave polling after its own publish waits ~1 us for them: the drain again, measured)
                if (wave == 0) {
                    if (!poll_flags(c, flags, NG * 4, (unsigned)t)) return;
                    lds_post(sync + S1_READY, (unsigned)t);
                } else if (!lds_wait_ge(c, sync + S1_READY, (unsigned)t)) return;
                u32x4v hb[NGH];
                if (!load_valid<NGH>(c, region_off(c, 0, t - 1) + (w * NGH * 64 + lane) * 16, hb)) return;
#pragma unroll
                for (int g = 0; g < NGQ; ++g) {
                    const f32x4v a0 = Wl[g * 64], a1 = Wl[(NGQ + g) * 64];
                    const f32x4v b0 = __builtin_bit_cast(f32x4v, hb[g]), b1 = __builtin_bit_cast(f32x4v, hb[NGQ + g]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], b0[e], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], b1[e], acc1, 0, 0, 0);
                    }
                }
            }
            f32x4v s2;
#pragma unroll
            for (int v = 0; v < 4; ++v) s2[v] = acc0[v] + acc1[v];          // q_{2w} + q_{2w+1}
            *(reinterpret_cast<f32x4v*>(part + (((int)(t & 1) * 4 + kap) * 2 + w) * 256) + lane) = s2;
            lds_post(sync + S1_P + kap * 2 + w, (unsigned)(t + 1));
        }
        return;
    }
    // ---- gate wave of unit block kap: lane (i = lane >> 4, n = lane & 15) -> unit 16 m + 4 i + kap, clip n of the tile
    const int kap = wave - 8, io = lane >> 4, n = lane & 15;
    const int C = a.C, N = a.N;
    const int j = 16 * m + 4 * io + kap;
    const int b = tile * 16 + n, bb = min(b, N - 1);
    const float bh0 = a.bhh[j], bh1 = a.bhh[C + j], bh2 = a.bhh[2 * C + j], bh3 = a.bhh[3 * C + j];
    const float* g = a.gi + (int64_t)bb * a.gi_b;
    const int64_t gc = a.gi_c, gt = a.gi_t;
    const float* sk = a.skip ? a.skip + ((int64_t)bb * C + j) * T : nullptr;
    float* const orow = a.out + (int64_t)bb * a.out_b + (int64_t)j * a.out_c;
    float* const cs_slot = a.cstate + (((int64_t)tile * NG + m) * 4 + kap) * 64 + lane;
    float cst = a.t0 > 0 ? *cs_slot : 0.0f;
    float g0, g1, g2, g3, skv = 0.0f;
    {
        const int64_t ts = a.t0;
        g0 = g[(int64_t)j * gc + ts * gt]; g1 = g[(int64_t)(C + j) * gc + ts * gt]; g2 = g[(int64_t)(2 * C + j) * gc + ts * gt]; g3 = g[(int64_t)(3 * C + j) * gc + ts * gt];
        if (sk) skv = sk[ts];
    }
    __builtin_amdgcn_s_setprio(3);
    for (int64_t t = a.t0; t < a.t1; ++t) {
        if (!lds_wait_ge(c, sync + S1_P + kap * 2, (unsigned)(t + 1)) || !lds_wait_ge(c, sync + S1_P + kap * 2 + 1, (unsigned)(t + 1))) return;
        const f32x4v* P = reinterpret_cast<const f32x4v*>(part + (((int)(t & 1) * 4 + kap) * 2) * 256) + lane;
        const f32x4v s01 = P[0], s23 = P[64];
        const float pi = g0 + ((s01[0] + s23[0]) + bh0), pf = g1 + ((s01[1] + s23[1]) + bh1);
        const float pg = g2 + ((s01[2] + s23[2]) + bh2), po = g3 + ((s01[3] + s23[3]) + bh3);
        const float ig = nc_sigmoidf(pi), fg = nc_sigmoidf(pf), gg = nc_tanhf(pg), og = nc_sigmoidf(po);
        cst = (fg * cst) + (ig * gg);
        const float h = og * nc_tanhf(cst);
        float* tr = htr + kap * 64;
        tr[n * 4 + io] = h;
        const float yo = sk ? h + skv : h;
        const int64_t tn = t + 1 < T ? t + 1 : t;
        g0 = g[(int64_t)j * gc + tn * gt]; g1 = g[(int64_t)(C + j) * gc + tn * gt]; g2 = g[(int64_t)(2 * C + j) * gc + tn * gt]; g3 = g[(int64_t)(3 * C + j) * gc + tn * gt];
        if (sk) skv = sk[tn];
        asm volatile("" ::: "memory");
        if (t + 1 < T) {
            if (lane < 16) {
                const f32x4v h4 = *reinterpret_cast<const f32x4v*>(tr + lane * 4);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, h4), c.rs, region_off(c, 0, t) + ((m * 64 + kap * 16 + lane) * 16), 0, 16 /* sc1 */);
            }
            asm volatile("" ::: "memory");
            if (lane == 0) __hip_atomic_store((l2_gu32*)(flags + 4 * m + kap), (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (b < N) orow[t * a.out_t] = a.elu_out ? nc_eluf(yo) : yo;
    }
    if (a.t1 < T) *cs_slot = cst;                                            // the next chunk launch of this layer resumes from here
}

}  // namespace

bool lstm2_supported(int C) { return C == 64 || C == 128 || C == 256 || C == 512; }

size_t lstm2_lds_bytes(int C, int tiles) { return ((size_t)3 * (C / 16) * 256 + (size_t)tiles * L2_GROUP) * sizeof(float); }

size_t lstm2_exchange_floats(int C, int64_t T, int tiles) { return (size_t)2 * T * tiles * C * 16; }

void lstm2_pack_image(const float* W, int C, float* image) {
    const int NG = C / 16;
    for (int u = 0; u < C / 4; ++u) {
        const int m = u >> 2, kap = u & 3;
        for (int g = 0; g < NG; ++g)
            for (int l = 0; l < 64; ++l) {
                const int r = l & 15, io = r >> 2, gate = r & 3, k4 = l >> 4;
                const int j = 16 * m + 4 * io + kap;
                for (int e = 0; e < 4; ++e)
                    image[(((size_t)u * NG + g) * 64 + l) * 4 + e] = W[(size_t)(gate * C + j) * C + 4 * (4 * g + e) + k4];
            }
    }
}

size_t lstm1_lds_bytes(int C) { return ((size_t)4 * (C / 16) * 256 + L1_PART + L1_HTR + 16) * sizeof(float); }

void lstm1_launch(const LstmSplitArgs& h, int tiles, hipStream_t stream) {
    if (!lstm2_supported(h.C)) fail(NC_ESTATE, "lstm1_launch: unsupported width");
    Lstm1Args a{};
    a.gi = h.gi; a.w = h.w; a.bhh = h.bhh; a.skip = h.skip; a.out = h.out; a.elu_out = h.elu_out; a.S = h.S; a.flags = h.flags; a.tmo = h.tmo;
    a.cstate = h.cstate; a.gi_b = h.gi_b; a.gi_c = h.gi_c; a.gi_t = h.gi_t; a.out_b = h.out_b; a.out_c = h.out_c; a.out_t = h.out_t;
    a.N = h.N; a.C = h.C; a.T = h.T; a.t0 = h.t0; a.t1 = h.t1; a.tile0 = h.tile0; a.tiles_total = h.tiles_total;
    const size_t lds = lstm1_lds_bytes(h.C);
    auto go = [&](auto kern) {
        ensure_dynamic_lds((const void*)kern, lds);
        hipLaunchKernelGGL(kern, dim3((unsigned)(h.C / 16), (unsigned)tiles), dim3(768), lds, stream, a);
    };
    switch (h.C) {
        case 512: go(lstm1_kernel<32>); break;
        case 256: go(lstm1_kernel<16>); break;
        case 128: go(lstm1_kernel<8>); break;
        default: go(lstm1_kernel<4>); break;
    }
    NC_HIP(hipGetLastError());
}

void lstm2_launch(const Lstm2Args& a, hipStream_t stream) {
    if (!lstm2_supported(a.C) || a.tiles < 1 || a.tiles > 2) fail(NC_ESTATE, "lstm2_launch: unsupported shape");
    const size_t lds = lstm2_lds_bytes(a.C, a.tiles);
    auto go = [&](auto kern) {
        ensure_dynamic_lds((const void*)kern, lds);
        hipLaunchKernelGGL(kern, dim3((unsigned)(a.C / 4)), dim3((unsigned)(512 * a.tiles)), lds, stream, a);
    };
    switch (a.C) {
        case 512: go(lstm2_kernel<32>); break;
        case 256: go(lstm2_kernel<16>); break;
        case 128: go(lstm2_kernel<8>); break;
        default: go(lstm2_kernel<4>); break;
    }
    NC_HIP(hipGetLastError());
}

}  // namespace nc
