// SNAC residual unit in ONE launch (Modules/SNAC/ResidualUnit.cs:25-60, depthwise flavour):
//     y = x + W1 . snake_a2( dw7_d( snake_a1(x) ) + b7 ) + b1        (then the Snake of the only consumer, where there is one)
// for the narrow long rows (C = 64 / 96 at 221 184 steps x 8 clips in C5): as two launches the unit is five passes over the tensor
// (depthwise: read x, write h; pointwise: read h, read x, write y) at 16 flop per byte -- HBM-bound, 0.39 / 0.9 ms per unit.  Here the
// depthwise output never exists in memory: two passes.
// Shape: a persistent workgroup of 4 wavefronts keeps W1 (packed [ci][C], the image of the streaming pointwise kernel) in LDS and walks
// 256-column tiles.  Per tile the input channels go through LDS in blocks of 16: the block's window (256 + 6 d columns) is loaded with
// 16-byte loads one block ahead, Snake-activated two values per packed instruction and written to LDS; every lane then builds its OWN
// B fragments of the 8 matrix-core steps of the block from that window -- 7 packed fmas (the lane's two columns per instruction), + b7,
// packed Snake(a2) -- one step ahead of the step whose 2 TM v_mfma_f32_32x32x2_f32 are being issued; the other co-resident workgroup
// fills the gaps.  The epilogue goes row block by row block (32 skip reads, fold, Snake, 32 stores) with uniform base + 32-bit lane
// offsets.  Measured (NC_SNAC_UNIT_TRACE, C = 96): a block 4.5-5 us (1.3 us of matrix-core time: building operands on the vector ALU
// does not hide under a saturated matrix stream of the same SIMD), staging 1.0 us, epilogue 10 us of a 45 us tile.
// Arithmetic = the two launches it replaces, operation for operation (bit-exact against the oracle: the SNAC suites run it at the C5
// size by default and on the reduced-width fixtures under NC_SNAC_FUSE_MIN_COLS=0, tests/test_children_gpu.py): Snake and the depthwise
// chain as dwconv_vec_kernel (k ascending from +0, + bias, Snake), the pointwise chain over ci ascending from +0 on the matrix cores,
// + bias, + skip, Snake.
#include "nc_conv.h"
#include "nc_elem.h"
#include "nc_frag.h"
#include "nc_math.h"

#include <utility>

namespace nc {

namespace {

template <int N, class F, int... I>
__device__ __forceinline__ void nc_static_for_su_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void nc_static_for_su(F&& f) { nc_static_for_su_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

typedef float su_f32x4 __attribute__((ext_vector_type(4)));
typedef float su_f32x16 __attribute__((ext_vector_type(16)));

struct UnitArgs {
    const float* x;        // [B][C][T]
    float* y;              // [B][C][T]
    const float* w1;       // [C ci][C rows] packed with a_tile_pos (one row tile = all channels)
    const float* tab;      // [C][12]: w7[0..6], b7, a2, 1/a2, a1, 1/a1
    const float* b1;       // [C] nullable
    const float* alpha_next;   // [C] nullable
    int B, T, n_t;
    unsigned long long* trace;   // nullable diagnostic (NC_SNAC_UNIT_TRACE): [4 workgroups][4 tiles][6 blocks][4] s_memrealtime stamps
};

template <int TM, int DIL, bool SNK>
__global__ __launch_bounds__(256, 2) void snac_unit_kernel(const UnitArgs p) {
    constexpr int C = 32 * TM, CB = 16, NBLK = C / CB, BN = 256, K = 7, HALO = 3 * DIL;
    constexpr int SH = (4 - (HALO & 3)) & 3;                    // window slot 0 sits on a 16-byte boundary of the row
    constexpr int XWORDS = (BN + 2 * HALO + SH + 3) / 4;        // 16-byte words per channel row of the window
    constexpr int XP = XWORDS * 4;
    constexpr int NITEM = CB * XWORDS, NL = (NITEM + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float su_lds[];   // W1 [C][C] | window [CB][XP] | tab [C][12] | epilogue [3][C]
    float* const Ws = su_lds;
    float* const Xs = Ws + C * C;
    float* const Tb = Xs + CB * XP;
    float* const Ep = Tb + C * 12;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = p.T, xw4 = T >> 2;
    {   // weights and tables, once per workgroup
        const su_f32x4* src = reinterpret_cast<const su_f32x4*>(p.w1);
        su_f32x4* dst = reinterpret_cast<su_f32x4*>(Ws);
        for (int i = tid; i < C * C / 4; i += 256) dst[i] = src[i];
        const su_f32x4* ts = reinterpret_cast<const su_f32x4*>(p.tab);
        su_f32x4* td = reinterpret_cast<su_f32x4*>(Tb);
        for (int i = tid; i < C * 3; i += 256) td[i] = ts[i];
        for (int i = tid; i < C; i += 256) {
            const float ao = p.alpha_next ? p.alpha_next[i] : 0.0f;
            Ep[i] = p.b1 ? p.b1[i] : 0.0f;
            Ep[C + i] = ao;
            Ep[2 * C + i] = nc_snake_inv(ao);
        }
    }
    __syncthreads();
    const int units = p.B * p.n_t, stride = gridDim.x;
    // staging items of this thread: item = tid + 256 u -> (channel of the block, 16-byte word of its window row)
    auto item_ch = [&](int u) __attribute__((always_inline)) { return min(tid + 256 * u, NITEM - 1) / XWORDS; };   // (a multiply-shift: XWORDS is a constant)
    auto item_wq = [&](int u) __attribute__((always_inline)) { const int it = min(tid + 256 * u, NITEM - 1); return it - (it / XWORDS) * XWORDS; };
    su_f32x4 st[NL];
    auto issue = [&](int unit, int cb) __attribute__((always_inline)) {   // global reads of block cb of a unit (clamped: always in bounds)
        const int b = unit / p.n_t, t = unit - b * p.n_t;
        const int g4 = (t * BN - HALO - SH) >> 2;                            // (arithmetic shift: the window starts left of the row)
        const su_f32x4* xb = reinterpret_cast<const su_f32x4*>(p.x + ((int64_t)b * C + cb * CB) * T);
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int wq = g4 + item_wq(u);
            st[u] = xb[(unsigned)item_ch(u) * (unsigned)xw4 + (unsigned)min(max(wq, 0), xw4 - 1)];
        }
    };
    auto stage = [&](int unit, int cb) __attribute__((always_inline)) {   // zero padding, Snake(a1), LDS
        const int t = unit % p.n_t;
        const int g4 = (t * BN - HALO - SH) >> 2;
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int wq = g4 + item_wq(u);
            su_f32x4 r = st[u];
            if (wq < 0 || wq >= xw4) r = su_f32x4{0.0f, 0.0f, 0.0f, 0.0f};   // (T % 4 == 0: a word is all in or all out)
            const float ai = Tb[(cb * CB + item_ch(u)) * 12 + 10], ai_inv = Tb[(cb * CB + item_ch(u)) * 12 + 11];
            const nc_f2 lo = nc_snakef2_m(nc_f2{r[0], r[1]}, nc_f2{ai, ai}, nc_f2{ai_inv, ai_inv});
            const nc_f2 hi2 = nc_snakef2_m(nc_f2{r[2], r[3]}, nc_f2{ai, ai}, nc_f2{ai_inv, ai_inv});
            if (tid + 256 * u < NITEM) reinterpret_cast<su_f32x4*>(Xs)[item_ch(u) * XWORDS + item_wq(u)] = su_f32x4{lo[0], lo[1], hi2[0], hi2[1]};
            __builtin_amdgcn_sched_barrier(0);   // one item at a time: interleaved, the Snake temporaries of all items are live at once
        }
    };
    const float* const Ab = Ws + hi * C + nc_a_lane_off<TM>(l31);
    const int c0 = wave * 64 + l31 + SH;                                     // window slot of tap 0 for column tile 0 (tile 1: + 32)
    int unit = blockIdx.x;
    if (unit < units) issue(unit, 0);
    for (; unit < units; unit += stride) {
        su_f32x16 acc[TM][2];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        const int nunit = unit + stride;
#pragma unroll 1
        for (int cb = 0; cb < NBLK; ++cb) {
            unsigned long long* tr = nullptr;
            if (p.trace && blockIdx.x < 4 && tid == 0 && (unit - (int)blockIdx.x) / stride < 4 && cb < 6)
                tr = p.trace + ((blockIdx.x * 4 + (unit - blockIdx.x) / stride) * 6 + cb) * 4;
            __syncthreads();                                                 // everybody is done with the previous window
            if (tr) tr[0] = __builtin_amdgcn_s_memrealtime();
            stage(unit, cb);
            if (tr) tr[1] = __builtin_amdgcn_s_memrealtime();
            __syncthreads();
            if (tr) tr[2] = __builtin_amdgcn_s_memrealtime();
            if (cb + 1 < NBLK) issue(unit, cb + 1);                          // the next block's reads fly under this block's work
            // B fragments of step u + 1 are built while the matrix cores work on step u: a two-stage software pipeline inside the wave.
            // (One stage -- window reads, 7 packed fmas, Snake, then the step's 2 TM matrix-core instructions -- left the pipe idle
            // while a wave computed and the vector ALU idle while it issued: 0.56 us per step against 0.16 us of matrix-core time.)
            auto build_h = [&](int u) __attribute__((always_inline)) -> nc_f2 {
                const int chl = 2 * u + hi, ch = cb * CB + chl;
                const su_f32x4 t0 = *reinterpret_cast<const su_f32x4*>(Tb + ch * 12), t1 = *reinterpret_cast<const su_f32x4*>(Tb + ch * 12 + 4);
                const float bv = t1[3], ao = Tb[ch * 12 + 8], ao_inv = Tb[ch * 12 + 9];
                const float wk[K] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2]};
                const float* row = Xs + chl * XP + c0;
                nc_f2 a2 = {0.0f, 0.0f};                                     // the two columns of a lane: one packed fma per tap
#pragma unroll
                for (int k = 0; k < K; ++k) a2 = nc_fma2(nc_f2{wk[k], wk[k]}, nc_f2{row[k * DIL], row[32 + k * DIL]}, a2);
                return nc_snakef2_m(a2 + nc_f2{bv, bv}, nc_f2{ao, ao}, nc_f2{ao_inv, ao_inv});
            };
            nc_f2 h = build_h(0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < CB / 2; ++u) {
                nc_f2 hn = h;
                if (u + 1 < CB / 2) hn = build_h(u + 1);
                float a[TM];
                nc_load_a_frag<TM>(Ab + 2 * (cb * (CB / 2) + u) * C, l31, a);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], h[0], acc[i][0], 0, 0, 0);
                    acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], h[1], acc[i][1], 0, 0, 0);
                }
                if (u + 1 < CB / 2) {                                        // deal the next step's reads and arithmetic into the matrix-core slots
#pragma unroll
                    for (int g = 0; g < 2 * TM; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one matrix-core instruction
                        __builtin_amdgcn_sched_group_barrier(0x100, (18 + 2 * TM - 1) / (2 * TM), 0);   // LDS reads
                        __builtin_amdgcn_sched_group_barrier(0x002, (48 + 2 * TM - 1) / (2 * TM), 0);   // vector ALU
                    }
                }
                __builtin_amdgcn_sched_barrier(0);                           // (one step of look-ahead: hoisted further, the reads cost 100+ registers)
                h = hn;
            }
            if (tr) tr[3] = __builtin_amdgcn_s_memrealtime();
        }
        // ---- epilogue: row R = 32 i + (r & 3) + 8 (r >> 2) + 4 hi, column t0 + 64 wave + 32 j + l31.  Phase A folds bias and the skip
        //      operand into the accumulators (every global read of the tile before its first store), phase B stores.
        const int b = unit / p.n_t, t = unit - b * p.n_t;
        const int col = t * BN + wave * 64 + l31;
        // uniform 64-bit base of the clip + 32-bit lane offsets (C * T < 2^30: SnacFusedUnit::usable checks): one offset register per access instead of
        // a 64-bit pointer pair -- with pointer arithmetic per row the tile's ~200 addresses cost the kernel its register budget
        const float* const xr = p.x + (int64_t)b * C * T;
        float* const yr = p.y + (int64_t)b * C * T;
        const bool ok0 = col < T, ok1 = col + 32 < T;
        const unsigned uT = (unsigned)T;
        const unsigned cc0 = (unsigned)(4 * hi) * uT + (unsigned)(ok0 ? col : 0), cc1 = (unsigned)(4 * hi) * uT + (unsigned)(ok1 ? col + 32 : 0);
        const unsigned so0 = (unsigned)(4 * hi) * uT + (unsigned)col;
        // Row block by row block: 32 skip reads in flight, fold, Snake, 32 stores.  (All of a tile's reads ahead of its first store would
        // be the rule -- a read behind a store waits for the store's acknowledgement -- but 64 TM values in flight beside the
        // accumulators spill, and spill reloads between stores wait the same way: measured 28 of the 68 us of a tile.  This form pays
        // TM - 1 such waits per tile.)  The next tile's first block is fetched first, ahead of every store.
        if (nunit < units) issue(nunit, 0);
        __builtin_amdgcn_sched_barrier(0);
        constexpr bool ALL_FIRST = TM <= 2;                                  // 64 rows: every skip read of the tile fits beside the accumulators
        float rsa[ALL_FIRST ? TM : 1][32];
        if constexpr (ALL_FIRST) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned R = (unsigned)(i * 32 + (r & 3) + 8 * (r >> 2));
                    rsa[i][2 * r] = xr[R * uT + cc0];
                    rsa[i][2 * r + 1] = xr[R * uT + cc1];
                }
        }
        nc_static_for_su<TM>([&](auto it) __attribute__((always_inline)) {
            constexpr int i = decltype(it)::value;
            float rs[32];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned R = (unsigned)(i * 32 + (r & 3) + 8 * (r >> 2));
                if constexpr (ALL_FIRST) { rs[2 * r] = rsa[i][2 * r]; rs[2 * r + 1] = rsa[i][2 * r + 1]; }
                else { rs[2 * r] = xr[R * uT + cc0]; rs[2 * r + 1] = xr[R * uT + cc1]; }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int R = i * 32 + (r & 3) + 8 * (r >> 2);
                const float bias = Ep[R + 4 * hi];
                nc_f2 v = {(acc[i][0][r] + bias) + rs[2 * r], (acc[i][1][r] + bias) + rs[2 * r + 1]};
                if constexpr (SNK) {
                    const float ao = Ep[C + R + 4 * hi], ao_inv = Ep[2 * C + R + 4 * hi];
                    v = nc_snakef2_m(v, nc_f2{ao, ao}, nc_f2{ao_inv, ao_inv});
                }
                acc[i][0][r] = v[0];
                acc[i][1][r] = v[1];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned R = (unsigned)(i * 32 + (r & 3) + 8 * (r >> 2));
                if (ok0) yr[R * uT + so0] = acc[i][0][r];
                if (ok1) yr[R * uT + so0 + 32u] = acc[i][1][r];
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    }
}

template <int TM>
size_t unit_lds(int dil) {
    const int C = 32 * TM, halo = 3 * dil, sh = (4 - (halo & 3)) & 3, xw = (256 + 2 * halo + sh + 3) / 4;
    return sizeof(float) * ((size_t)C * C + (size_t)16 * xw * 4 + (size_t)C * 12 + (size_t)3 * C);
}

}  // namespace

bool SnacFusedUnit::supported(int C, int K, int dil) { return (C == 64 || C == 96) && K == 7 && (dil == 1 || dil == 3 || dil == 9); }

void SnacFusedUnit::build(int C_, int dil_, const float* w7, const float* b7, const float* a1, const float* a2, const float* w1_dense, const float* b1_h) {
    C = C_; dil = dil_;
    const int TM = C / 32;
    std::vector<float> pk((size_t)C * C), tb((size_t)C * 12);
    for (int ci = 0; ci < C; ++ci)
        for (int co = 0; co < C; ++co) pk[(size_t)ci * C + a_tile_pos(TM, co / 32, co % 32)] = w1_dense[(size_t)co * C + ci];
    for (int c = 0; c < C; ++c) {
        for (int k = 0; k < 7; ++k) tb[(size_t)c * 12 + k] = w7[(size_t)c * 7 + k];
        tb[(size_t)c * 12 + 7] = b7 ? b7[c] : 0.0f;
        tb[(size_t)c * 12 + 8] = a2[c];
        tb[(size_t)c * 12 + 9] = nc_snake_inv(a2[c]);
        tb[(size_t)c * 12 + 10] = a1[c];
        tb[(size_t)c * 12 + 11] = nc_snake_inv(a1[c]);
    }
    w1.reserve(pk.size() * 4); tab.reserve(tb.size() * 4);
    NC_HIP(hipMemcpy(w1.p, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
    NC_HIP(hipMemcpy(tab.p, tb.data(), tb.size() * 4, hipMemcpyHostToDevice));
    has_b1 = b1_h != nullptr;
    if (has_b1) {
        b1.reserve((size_t)C * 4);
        NC_HIP(hipMemcpy(b1.p, b1_h, (size_t)C * 4, hipMemcpyHostToDevice));
    }
    ready = true;
}

bool SnacFusedUnit::usable(const float* x, const float* y, int64_t T, int B) const {
    static const bool off = env_flag("NC_SNAC_NO_FUSE");
    static const int64_t min_cols = env_int("NC_SNAC_FUSE_MIN_COLS", 65536);
    // the epilogue addresses a clip's [C][T] block with 32-bit BYTE offsets off a uniform 64-bit base: C * T floats must stay below 2^30
    // (the bound had been on T alone -- a 254 s clip at C = 96 wrapped); longer clips take the two-launch path
    static const int64_t max_elems = std::min<int64_t>(env_int("NC_SNAC_FUSE_MAX_ELEMS", 1 << 30), (int64_t)1 << 30);
    return ready && !off && (T & 3) == 0 && T >= 256 && (int64_t)B * T >= min_cols && (int64_t)C * T < max_elems &&
           ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
}

void SnacFusedUnit::launch(const float* x, const float* alpha_next, float* y, int B, int64_t T, int cu_count, hipStream_t s, Profiler* prof) const {
    UnitArgs a{};
    a.x = x; a.y = y; a.w1 = w1.as<float>(); a.tab = tab.as<float>(); a.b1 = has_b1 ? b1.as<float>() : nullptr; a.alpha_next = alpha_next;
    a.B = B; a.T = (int)T; a.n_t = (int)((T + 255) / 256);
    const int64_t units = (int64_t)B * a.n_t;
    const unsigned grid = (unsigned)std::min<int64_t>(units, (int64_t)2 * cu_count);
    ProfScope ps(prof, s, NC_KC_CONV_K1, (2.0 * C + 2.0 * 7) * C * (double)T * B, 8.0 * C * (double)T * B);
    auto go = [&](auto kern, size_t lds) {
        ensure_dynamic_lds((const void*)kern, lds);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
    };
    static const char* trace_path = env_str("NC_SNAC_UNIT_TRACE");
    static bool traced = false;
    DevBuf tbuf;
    if (trace_path && !traced && C == 96) {
        tbuf.reserve(4 * 4 * 6 * 4 * 8);
        NC_HIP(hipMemsetAsync(tbuf.p, 0, 4 * 4 * 6 * 4 * 8, s));
        a.trace = tbuf.as<unsigned long long>();
    }
    const bool snk = alpha_next != nullptr;
    auto pick = [&](auto tm, auto dl) {
        constexpr int TMv = decltype(tm)::value, DLv = decltype(dl)::value;
        if (snk) go(snac_unit_kernel<TMv, DLv, true>, unit_lds<TMv>(DLv)); else go(snac_unit_kernel<TMv, DLv, false>, unit_lds<TMv>(DLv));
    };
    auto by_dil = [&](auto tm) {
        if (dil == 1) pick(tm, std::integral_constant<int, 1>{}); else if (dil == 3) pick(tm, std::integral_constant<int, 3>{}); else pick(tm, std::integral_constant<int, 9>{});
    };
    if (C == 64) by_dil(std::integral_constant<int, 2>{}); else by_dil(std::integral_constant<int, 3>{});
    NC_HIP(hipGetLastError());
    if (a.trace) {
        traced = true;
        std::vector<unsigned long long> hb(4 * 4 * 6 * 4);
        NC_HIP(hipStreamSynchronize(s));
        NC_HIP(hipMemcpy(hb.data(), a.trace, hb.size() * 8, hipMemcpyDeviceToHost));
        if (FILE* f = std::fopen(trace_path, "w")) {
            for (int w = 0; w < 4; ++w)
                for (int t = 0; t < 4; ++t)
                    for (int cb = 0; cb < 6; ++cb) {
                        const unsigned long long* r = &hb[((w * 4 + t) * 6 + cb) * 4];
                        std::fprintf(f, "wg %d tile %d block %d: stage-start %.2f staged +%.2f barrier +%.2f computed +%.2f us\n", w, t, cb,
                                     (double)(r[0] - hb[0]) * 0.01, (double)(r[1] - r[0]) * 0.01, (double)(r[2] - r[1]) * 0.01, (double)(r[3] - r[2]) * 0.01);
                    }
            std::fclose(f);
        }
        tbuf.release();
    }
}

}  // namespace nc
