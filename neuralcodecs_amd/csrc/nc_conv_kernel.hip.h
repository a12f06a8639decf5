// Device code of the implicit-GEMM convolution (included by the per-K instantiation units).
#pragma once
#include <hip/hip_runtime.h>

#include "nc_conv.h"
#include "nc_math.h"
#include "nc_frag.h"
#include "nc_gn.h"

#include <type_traits>
#include <utility>

#ifdef NC_CONV_TRACE
#define NC_TRACE_WGS 16
#ifndef NC_TRACE_WG0
#define NC_TRACE_WG0 1024   // first traced workgroup: well behind the launch front, where the workgroups of the chip have drifted apart
#endif
#define NC_TRACE_CB0 8
#define NC_TRACE_NCB 8
#endif
// ---- compile-time knobs of the template.  The SHIPPED build fixes every one of them to its shipped value: the A/B experiments of DESIGN 8
// (rounds 4-5; results under profiles/r05_ab_*.txt) can only be compiled with -DNC_EXPERIMENTS (`make EXPERIMENTS=1` + tools/probe/mk_abl.sh),
// and a -D override without it is an error -- one of them, NC_XV_SNAKE_ON=0, is a timing probe that computes WRONG results on purpose.
#if !defined(NC_EXPERIMENTS) && (defined(NC_STAGE_PRIO) || defined(NC_LEGACY_ROWS) || defined(NC_XV_SNAKE_ILV) || defined(NC_XV_SNAKE_ON) || \
                                 defined(NC_XV_FD) || defined(NC_XV_STORE_SEG))
#error "NC_STAGE_PRIO / NC_LEGACY_ROWS / NC_XV_SNAKE_ILV / NC_XV_SNAKE_ON / NC_XV_FD / NC_XV_STORE_SEG are experiment knobs: build with -DNC_EXPERIMENTS"
#endif
#ifndef NC_STAGE_PRIO
#define NC_STAGE_PRIO 0     // s_setprio of the staging runs of the segmented pipeline (0 = leave the priority alone; measured: no gain)
#endif
#ifndef NC_LEGACY_ROWS      // staging Snake of the legacy instances: 2 = sign-free chains side by side (shipped), 1 = side by side with the parity select, 0 = pair by pair
#define NC_LEGACY_ROWS 2
#endif
#ifndef NC_XV_SNAKE_ILV     // 1 = the Snake chains of a staging run step by step side by side (shipped); 0: word by word, the round-5 first form
#define NC_XV_SNAKE_ILV 1
#endif
#ifndef NC_XV_SNAKE_ON      // 1 (shipped); 0 = timing probe only (WRONG results): what the staging Snake costs in total
#define NC_XV_SNAKE_ON 1
#endif
#ifndef NC_XV_FD
#define NC_XV_FD 1          // XV-only instances: fragment prefetch depth (2 measured: no gain)
#endif
#ifndef NC_XV_STORE_SEG
#define NC_XV_STORE_SEG 3   // XV staging: the block's one staging run sits at the head of this matrix-core segment (0 .. 3; 4 = behind the last):
#endif                      // segment 3 measured best (profiles/r05_ab_xv_store_segment.txt)

namespace nc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int N, class F, int... I>
__device__ __forceinline__ void nc_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
// f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N-1>{}), in order
template <int N, class F>
__device__ __forceinline__ void nc_static_for(F&& f) {
    nc_static_for_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}
// a flag that is either a compile-time constant (std::true_type / std::false_type) or this run-time value
struct nc_rt_flag {
    bool v;
    __device__ constexpr operator bool() const { return v; }
};
typedef __attribute__((address_space(1))) const void* nc_gptr;
typedef __attribute__((address_space(3))) void* nc_lptr;

// Block = NW waves (4, or 8 for the wide variants).  Wave w owns all BM = 32*TM output channels of the tile and the
// 32*TN columns [w*32*TN, (w+1)*32*TN).  The reduction runs over blocks of CB input channels
// (KB = CB*K flattened kk = ci*K + k, ascending: the canonical chain order).  Per block the
// workgroup holds in LDS, double-buffered,
//   As[kk][BM]   packed weights (a linear copy of the pre-packed global image)
//   Xs[ci][...]  the input window of the tile (Snake applied on the way in), de-interleaved by
//                stride phase so the 32 lanes of an MFMA B-fragment always read consecutive words
// Software pipeline, one barrier per reduction block: the KB/2 matrix-core steps of block cb are cut
// into NSEG segments; the global loads of block cb+1 are issued in NSEG-1 groups, group g at the head
// of segment g, and group g is transformed (Snake) and written to the other LDS buffer at the head of
// segment g+1 -- so a load has a whole segment of MFMA time to land, only 1/(NSEG-1) of the staging
// registers are live at once, and the VALU work sits between matrix-core segments.
//   MFMA step kp: lane l supplies A[row = l&31][kk = 2*kp + (l>>5)] and B[kk][col = l&31].
// NP > 0: wave-specialised variant.  NW consumer waves do nothing but fragment reads and matrix-core steps; NP producer waves
// stage the next reduction block (global reads, Snake, LDS writes).  In the classic variant every wave alternates between the two
// roles, and its vector-ALU staging run issues slowly while the co-resident wave saturates the matrix pipe -- time during which
// its own matrix-core steps cannot issue.  With the roles split, the consumers always have matrix-core work ready.
// DIST: distributed staging.  A wave issues in order, and a matrix-core instruction holds its wave until the pipe accepts it; the
// few instructions that sit BETWEEN two matrix-core instructions issue in the shadow of the first (64 cycles), but a cluster of
// staging instructions between two matrix-core streaks keeps the wave away from the pipe for its whole length.  So the reads of the
// next reduction block are issued at the top of the step and its transform + LDS writes are dealt out over the step's matrix-core
// steps, a unit (one weight vector or one window item) per step, in ONE basic block (no `if (more)`: the last step re-stages the
// last block into the idle buffer).
// SUB: sub-pixel form of a transposed convolution (WNConvTranspose1d.cs:142-163 with stride s = 2^sub_shift).  The s polyphase
// sub-convolutions share their input window (taps x[q], x[q-1], ...), so they run as ONE convolution with s*Cout output rows, row
// R = co*s + r, whose element (R, q) is output sample t = q*s + r - pad of channel co.  The 4 consecutive D rows a lane holds are
// then consecutive samples of one channel, and the 32 columns x 2 lane halves of a store instruction cover a contiguous run of the
// output row -- where the per-phase launches wrote every s-th sample from different workgroups (3x write amplification in HBM).
// SUB == 2: the same for ANY stride with two taps per phase (k = 2s; SNAC's stride-3 and Encodec's stride-5 up-convolutions): channel and
// phase of a packed row come from a multiply-shift division (ConvArgs::sub_stride / sub_magic) instead of a shift and a mask, and a
// lane half's 4 extra rows no longer fold into one lane offset -- each row's offset is picked per lane half from two scalar values.
// IN2: two-input form of the Encodec input mode (the sum of a residual block's shortcut and branch, each a raw conv output with its own
// pending GroupNorm: SEANetResnetBlock.cs:72-85 feeding the next SConv1d / SConvTranspose1d): both operands are read with the same
// window addresses, normalised separately, added, then ELU + pad -- the summed / activated / padded copy is never written.
// XVK (round 5): the XV-ONLY instances.  One staging form (vectorised words, one run per block, rotating register set: see the XV note
// below), launched by the host only where that form applies (plain or Snake input with aligned rows, one-clip tiles, constant window pitch),
// with everything else compiled OUT: no Encodec input mode, no flattened column axis, no item-form staging state, no generic-pitch loop,
// no GroupNorm epilogue.  This is the "scalar-register diet" of VERDICT r4: the legacy instances keep every run-time mode and pay for it
// in spilled scalar registers whose reloads are vector instructions.
template <int TM, int TN, int K, int CB, int NX, bool FUSE = false, int OCC = 2, int NW = 4, int NP = 0, bool DIST = false, int SUB = 0,
          bool IN2 = false, bool XVK = false, bool DUO = false>
__global__ __launch_bounds__((DUO ? 2 : 1) * 64 * (NW + NP), DUO ? 1 : OCC) void conv_mfma_kernel(const ConvArgs p) {
    // DUO (round 5): TWO tiles per workgroup.  A workgroup of 8 wavefronts is two "sub-workgroups" of 4 (one wavefront of each per SIMD),
    // each with its own tile, its own half of the LDS and the XV-only loop -- but the second runs HALF A REDUCTION BLOCK behind the first,
    // held there by the workgroup barrier, which both now pass twice per block (at the block's end and in its middle: one sub-workgroup's
    // end is the other's middle).  Per half block one of them runs its staging run + two matrix-core segments, the other two segments:
    // a staging run always has a partner inside a matrix-core segment on its SIMD.  With two independent workgroups per CU that
    // anti-phase is left to chance (the phase trace: the pipe idles where both waves of a SIMD are outside their segments at once).
    static_assert(!DUO || (XVK && !FUSE), "DUO belongs to the XV-only plain instances");
    constexpr bool SPEC = NP > 0;
    constexpr int NT = 64 * (NW + NP);             // threads per workgroup
    constexpr int SW = SPEC ? NP : NW;             // waves that stage
    constexpr int SNT = 64 * SW;
    constexpr int BM = 32 * TM;
    constexpr int BNW = 32 * TN;
    constexpr int BN = NW * BNW;
    constexpr int KB = CB * K;
    constexpr int KP = KB / 2;
    constexpr int A_FLOATS = KB * BM;
    constexpr int A_VEC = A_FLOATS / 4;            // float4 words in the weight tile
    constexpr int NA = (A_VEC + SNT - 1) / SNT;    // float4 copies per staging thread
    constexpr bool A_DMA = false;
    constexpr int NSEG = SPEC ? 3 : DIST ? 2 : (KP >= 16 ? 4 : 2);   // (specialised: the producers stage a block in two groups)
    constexpr int NG = NSEG - 1;
    constexpr int GA = (NA + NG - 1) / NG;         // per-group register footprint
    constexpr int GX = (NX + NG - 1) / NG;
    constexpr int GXX = IN2 ? 2 * GX : GX;         // staging registers per group: the second operand's samples sit behind the first's
    static_assert(!IN2 || (!FUSE && !SPEC && !DIST), "the two-input staging mode belongs to the plain template");
    static_assert(KB % 2 == 0, "reduction block must hold an even number of kk");
    static_assert(A_FLOATS % 4 == 0, "A tile must be 16-byte granular");

    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    const int sub = DUO ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) : 0;      // sub-workgroup (DUO), wave-uniform
    float* const smem = DUO ? smem_all + sub * ((p.ep_off + 6 * BM + 3) & ~3) : smem_all;    // its half of the LDS

    const int tid = DUO ? (int)(threadIdx.x & 255) : (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hi = lane >> 5;
#ifdef NC_CONV_TRACE
    // In-kernel phase trace (diagnostic builds only: -DNC_CONV_TRACE; VERDICT r4 item 3 iv).  Workgroups [NC_TRACE_WG0, + NC_TRACE_WGS) stamp the
    // shader clock (s_memtime) at 8 points of reduction blocks [NC_TRACE_CB0, +NC_TRACE_NCB): 0 = block top (behind the barrier), 1 / 2 =
    // before / behind the staging work at the head of segment 1, 3 / 4 segment 2, 5 / 6 segment 3, 7 = in front of the closing barrier.
    // One lane per wavefront stores; the buffer rides in ConvArgs::x2 (unused outside the two-input instances), set by launch_conv for the
    // launch NC_CONV_TRACE_FILE selects.  tools/probe/conv_trace.py prints segment, staging and barrier-wait times per wavefront.
    unsigned long long* const trace_p = (!IN2 && p.x2 != nullptr && blockIdx.x >= NC_TRACE_WG0 && blockIdx.x < NC_TRACE_WG0 + NC_TRACE_WGS)
        ? reinterpret_cast<unsigned long long*>(const_cast<float*>(p.x2)) + ((size_t)(blockIdx.x - NC_TRACE_WG0) * 8 + (tid >> 6)) * NC_TRACE_NCB * 8 : nullptr;
#define NC_STAMP(cb, slot)                                                                                         \
    do {                                                                                                            \
        if (trace_p && (cb) >= NC_TRACE_CB0 && (cb) < NC_TRACE_CB0 + NC_TRACE_NCB && (tid & 63) == 0)               \
            trace_p[((cb) - NC_TRACE_CB0) * 8 + (slot)] = __builtin_amdgcn_s_memtime();                             \
    } while (0)
#else
#define NC_STAMP(cb, slot) do {} while (0)
#endif
    const bool producer = SPEC && wave >= NW;
    const int swave = SPEC ? wave - NW : wave;     // index among the staging waves (consumers of the specialised variant: unused)
    const int stid = SPEC ? tid - 64 * NW : tid;

    // ---- XCD-aware block -> tile map: block id b runs on XCD b%8 (observed; speed only).  Give each
    // XCD a contiguous range of the (phase, co_tile, clip, t_tile) order so the blocks resident on
    // one XCD share a weight panel in that XCD's L2.
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    int lin;
    {
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    bool duo_dead = false;   // DUO: the second sub-workgroup of the last workgroup of an odd tile count repeats the last tile without storing
    if constexpr (DUO) {
        const int n_tiles = p.n_phase * p.n_co_tiles * p.B * p.n_t_tiles;
        lin = 2 * lin + sub;
        duo_dead = lin >= n_tiles;
        lin = min(lin, n_tiles - 1);
    }
    // (integer division runs on the vector ALU: hand the wave-uniform results back to scalar registers)
    // Polyphase (transposed) launches: the stride phases of one output tile are adjacent in the order, so they run together on
    // one XCD -- their interleaved 4-byte stores (every stride-th sample of the same lines) merge in that L2 instead of each
    // going to HBM as a partial line, and they share the input window.
    // Row tiles in groups of p.co_group (host: as many weight panels as fit ~2 MB of an XCD's 4 MB L2): the co_group row tiles of
    // one column tile are adjacent in the order, so they run together on one XCD and the input window is fetched from the fabric
    // once per GROUP instead of once per row tile (C = 384: 4.4x -> 2x input traffic), while the group's panels stay L2-resident.
    const int phase = __builtin_amdgcn_readfirstlane(lin % p.n_phase);
    lin /= p.n_phase;
    const int co_in = __builtin_amdgcn_readfirstlane(lin % p.co_group);
    lin /= p.co_group;
    const int t_tile = __builtin_amdgcn_readfirstlane(lin % p.n_t_tiles);
    lin /= p.n_t_tiles;
    const int co_tile = __builtin_amdgcn_readfirstlane((lin / p.B) * p.co_group + co_in);
    // Flattened column axis (p.flat, host: launch_conv): the columns of all clips form ONE axis n = clip*n_cols + column, cut into
    // BN-wide tiles regardless of clip boundaries -- rows of 87 / 150 / 696 columns otherwise leave 6-40 % of every 96/128/256-wide tile
    // on padding and give the deep layers one short tile per (clip, weight panel).  A tile then touches up to 4 clips ("segments"):
    // segment m of the tile is clip b + m.  In LDS every segment keeps its own halo: column n of segment m sits at window column
    // n + m*flat_hc, i.e. clips are laid out at a pitch of n_cols + flat_hc columns (flat_px x-slots), and the segment of a window
    // slot / tile column is found by comparing against multiples of the pitch.  Without p.flat both pitches are huge (segment 0 always)
    // and every formula below reduces to the one-clip tile: b = clip of the tile, col0 = its first column.
    // (With GroupNorm sums in the epilogue the clip pitch p.flat_pc is n_cols rounded up to whole 32-column blocks, so every 32x32
    // accumulator tile is ONE canonical block of ONE sample; the columns between n_cols and the pitch are padding.)
    const bool flatm = XVK ? false : (p.flat != 0);
    const int b = __builtin_amdgcn_readfirstlane(flatm ? (t_tile * BN) / p.flat_pc : lin % p.B);   // first clip of the tile
    const int col0 = flatm ? t_tile * BN - b * p.flat_pc : t_tile * BN;                              // first column, within clip b
    const int flat_px = XVK ? 0x1fffffff : p.flat_px, flat_pc = XVK ? 0x1fffffff : p.flat_pc;
    const int flat_hc = XVK ? 0 : p.flat_hc;
    auto seg_of = [&](int r, int pitch) __attribute__((always_inline)) -> int { return (int)(r >= pitch) + (int)(r >= 2 * pitch) + (int)(r >= 3 * pitch); };

    const int s = p.stride;
    const int xs0 = col0 * s - p.pad - p.xneg;  // global x position of window slot 0

    const f32x4* const wbase =
        reinterpret_cast<const f32x4*>(p.w + (int64_t)phase * p.w_phase_stride + (int64_t)co_tile * p.n_cb * A_FLOATS);
    const float* const xb = p.x + (int64_t)b * p.x_bstride;
    const float* const xb2 = IN2 ? p.x2 + (int64_t)b * p.x_bstride : nullptr;

    float* const As0 = smem;
    float* const Xs0 = smem + 2 * A_FLOATS;

    // kernel arguments used inside the pipeline, pinned in registers
    const int nchunk = p.nchunk, chunk_magic = p.chunk_magic, stride_magic = p.stride_magic;
    const int xwp = p.xwp, xrow = p.xrow, xbuf = p.xbuf, Cin = p.Cin, x_len = p.x_len, n_cb = p.n_cb;
    const unsigned x_cstride = (unsigned)p.x_cstride;
    const float* const alpha_in = p.alpha_in;
    // Encodec input mode (ConvArgs::in_mode): pending GroupNorm + ELU + reflect pad applied while staging
    const int in_mode = XVK ? 0 : p.in_mode;
    const float in_mu = (in_mode & 1) ? p.in_stats[2 * b] : 0.0f, in_rs = (in_mode & 1) ? p.in_stats[2 * b + 1] : 1.0f;
    const float in_mu2 = (IN2 && (in_mode & 1)) ? p.in_stats2[2 * b] : 0.0f, in_rs2 = (IN2 && (in_mode & 1)) ? p.in_stats2[2 * b + 1] : 1.0f;
    int tap[K];  // window slot of tap k (wave-uniform scalars)
#pragma unroll
    for (int k = 0; k < K; ++k) tap[k] = p.tapoff[k];
    // Snake alphas of all input channels, staged once per block behind the tile buffers
    float2* const Al = reinterpret_cast<float2*>(Xs0 + 2 * xbuf);   // (alpha, 1/alpha) per input channel
    float2* const Al2 = Al + p.n_cb * CB;                           // IN2: (gamma, beta) of the second operand
    // per-row epilogue operands of this tile: bias, Snake alpha and 1/alpha (and the fused unit's second set).  The epilogue
    // reads them from LDS: a global read issued after the first store would wait for every earlier store to be acknowledged.
    // (Both tables are filled in the prologue below, behind the first tile's reads.)
    float* const Ep = smem + p.ep_off;

    // ---- staging (branch-free; every address is clamped into the tensor) ------------------------
    // weights: thread t copies float4 words t + 256*n; input window: wave-level items of 64 consecutive
    // window slots of one channel, item -> wave item%4; items past n_items land in pad rows.
    f32x4 ra[GA];
    float rx[GXX];
    // loop-invariant part of the window reads: item i of this wave covers channel xc[i] (wave-uniform) of the reduction block and
    // the 64 window slots starting at xj[i]; its lanes read x[clamp(xs0 + slot)] = xg[i] of that channel row
    unsigned xg[NX];
    int xc[NX];
    unsigned okm = 0;   // bit i = window item i of this lane reads a real sample (else zero padding / the zero extension / tile overrun)
    if constexpr (!XVK)
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int item = swave + SW * i;
        xc[i] = (item * chunk_magic) >> 20;
        const int jw = (item - xc[i] * nchunk) * 64 + lane;          // window slot of this lane
        const int sg = seg_of(col0 * s + jw, flat_px);               // its segment (0 without p.flat)
        const int gp = xs0 + jw - sg * flat_px;                      // x position within clip b + sg
        xg[i] = (unsigned)min(sg, p.Bc - 1 - b) * (unsigned)p.x_bstride + (unsigned)min(max(gp, 0), x_len - 1);   // (address stays inside the tensor)
        if ((b + sg < p.Bc) & (gp >= 0) & (gp < x_len)) okm |= 1u << i;
        if (in_mode & 4) {   // SConv1d.Pad1d (SConv1d.cs:258-274): padded position gp -> sample |gp - left| mirrored at Lz - 1
            int q = gp - p.in_left;
            q = q < 0 ? -q : q;
            if (q >= p.in_Lz) q = 2 * (p.in_Lz - 1) - q;
            if (!((q >= 0) & (q < p.in_L))) okm &= ~(1u << i);
            xg[i] = (unsigned)min(sg, p.Bc - 1 - b) * (unsigned)p.x_bstride + (unsigned)min(max(q, 0), p.in_L - 1);
        }
    }
    // Interior tiles (no padding, no zero extension, no tile overrun anywhere in the window: all but the first / last tile of a clip)
    // skip the per-item zero select while staging: wave-uniform, decided once.  The compiler keeps the per-item predicates as 64-bit
    // scalar masks -- 2 NX scalar registers, which the two-tap instances (NX = 20) spill and reload with a vector instruction each,
    // inside the loop (found in the ISA, round 4).
    const bool tile_allok = __builtin_amdgcn_ballot_w64(okm != (NX >= 32 ? ~0u : (1u << NX) - 1u)) == 0;
    // ---- XV (round 5): vectorised window staging of the two-tap (sub-pixel up-convolution) instances.  With the constant row pitch the
    // window image of a reduction block is one linear array [CB][XROWC] = CB * 80 float4 words: thread t copies words t + NT * n -- FIVE
    // 16-byte loads and five `ds_write_b128` per thread and block where the item form spends twenty dword loads, twenty `ds_write_b32`
    // and an address instruction each -- and holds 5 offsets instead of 20.  With so few registers per block the reads can stay in
    // flight for a WHOLE block (rotating schedule in the main loop) instead of one matrix-core segment: a segment of these 16-step
    // blocks is 1.3 us of matrix-pipe time, less than a loaded L2 / HBM round trip -- measured on DAC's up-convolutions (same box,
    // 60 launches per layer): item form = XV written one or two segments after its reads (no gain: 2.33 ms), written three segments
    // after them -2.7 ... -4.4 % per layer; the latency cover is what these short-reduction classes were missing.  The host
    // (launch_conv: EPI_XVEC) grants it for plain inputs with 16-byte aligned rows, a window start that is a multiple of 4 samples (xneg
    // is raised for that) and row lengths that are multiples of 4, so that a float4 is inside the row or outside it as a whole.
    // K = 7 (every launch whose rows start on 64-byte boundaries -- NC_XV_K7_MIN_COLS=<n> raises a column floor; NC_NO_XV_K7=1 keeps the legacy instances): the same for the dilated residual-unit convolutions with 8-byte words -- [8][320] floats = 5
    // float2 per thread, so the Snake work per thread stays exactly the item form's 10 elements, as packed pairs of one channel.
    constexpr bool XVCAND = XVK;
    static_assert(!XVK || (TN == 2 && NW == 4 && !IN2 && !SPEC && !DIST && ((K == 2 && SUB != 0 && !FUSE) || (K == 7 && SUB == 0))),
                  "XV-only instances: two-tap sub-pixel or k = 7, 256-column tiles");
    constexpr int XVW = K == 2 ? 4 : 2;                          // floats per staged word
    constexpr int XVROW = 320;                                   // (== XROWC of the 256-column tiles, defined with the fragment reads below)
    constexpr int XVN = CB * XVROW / XVW;                        // words of the window image
    constexpr int NV = XVCAND ? (XVN + NT - 1) / NT : 1;
    typedef float xv_t __attribute__((ext_vector_type(XVW)));
    constexpr bool use_xv = XVCAND;                              // (the host launches these instances only where the form applies)
    unsigned xvo[NV];                                            // float offset of word n from the first channel row of a reduction block
    unsigned xvok = 0;                                           // bit n: word n lies inside its row (else zero padding)
    int xvc[(XVCAND && K == 7) ? NV : 1];                        // K = 7: channel (within the block) of word n, for its Snake operands
    xv_t rxv[NV];
    f32x4 rav[XVCAND ? NA : 1];
    if constexpr (XVCAND) {
        {
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                const int f = min(tid + NT * n, XVN - 1);
                const int ch = f / (XVROW / XVW), q = f - ch * (XVROW / XVW);
                const int gp = xs0 + XVW * q;
                const bool ok = (gp >= 0) & (gp + XVW - 1 < x_len);
                xvo[n] = (unsigned)ch * x_cstride + (unsigned)(ok ? gp : 0);
                if constexpr (K == 7) xvc[n] = ch;
                if (ok) xvok |= 1u << n;
            }
        }
    }
    const bool xv_allok = __builtin_amdgcn_ballot_w64(xvok != (1u << NV) - 1u) == 0;
    auto xv_issue = [&](int cbn) __attribute__((always_inline)) {
        if constexpr (XVCAND) {
            const float* base = xb + (size_t)((unsigned)(cbn * CB) * x_cstride);   // uniform
#pragma unroll
            for (int n = 0; n < NV; ++n) rxv[n] = *reinterpret_cast<const xv_t*>(base + xvo[n]);
            const f32x4* src = wbase + (size_t)cbn * A_VEC;
#pragma unroll
            for (int n = 0; n < NA; ++n) rav[n] = (src + SNT * n)[(A_VEC % SNT == 0) ? (unsigned)stid : min((unsigned)stid, (unsigned)(A_VEC - 1 - SNT * n))];
        }
    };
    auto xv_store = [&](int cbn, float* Ad, float* Xd, auto allok_tag, auto snake_tag) __attribute__((always_inline)) {
        if constexpr (XVCAND) {
#pragma unroll
            for (int n = 0; n < NA; ++n)
                if ((A_VEC % SNT == 0) || stid + SNT * n < A_VEC) reinterpret_cast<f32x4*>(Ad)[stid + SNT * n] = rav[n];
            if constexpr (K == 7 && NC_XV_SNAKE_ON && NC_XV_SNAKE_ILV && decltype(snake_tag)::value) {
                // Snake of the consumed tensor (snake(0) == 0: the zero padding survives it), the sine up to its sign ((-s)^2 == s^2 exactly:
                // bit-identical, 8 of the ~25 instructions of a pair fewer; nc_math.h), and the NV words' chains STEP BY STEP side by side:
                // read -> chain -> store word by word made every packed instruction wait on the one before it (an `s_nop` between each
                // pair: 75 in a staging run) because the LDS store of one word orders the alpha read of the next behind it
                nc_f2 xv[NV], a[NV], iv[NV];
#pragma unroll
                for (int n = 0; n < NV; ++n) {
                    const float2 al = Al[cbn * CB + xvc[n]];
                    a[n] = nc_f2{al.x, al.x}; iv[n] = nc_f2{al.y, al.y};
                    xv[n] = nc_f2{rxv[n][0], rxv[n][1]};
                    if constexpr (!decltype(allok_tag)::value)
                        if (!((xvok >> n) & 1u)) xv[n] = nc_f2{0.0f, 0.0f};
                }
                nc_snakef2_m_rows<NV>(xv, a, iv);
#pragma unroll
                for (int n = 0; n < NV; ++n)
                    if ((XVN % NT == 0) || tid + NT * n < XVN) reinterpret_cast<xv_t*>(Xd)[tid + NT * n] = xv_t{xv[n][0], xv[n][1]};
            } else {
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                xv_t v = rxv[n];
                if constexpr (!decltype(allok_tag)::value)
                    if (!((xvok >> n) & 1u)) v = xv_t(0.0f);
                if constexpr (K == 7 && NC_XV_SNAKE_ON && decltype(snake_tag)::value) {
                    const float2 al = Al[cbn * CB + xvc[n]];
                    const nc_f2 sm = nc_snakef2_m(nc_f2{v[0], v[1]}, nc_f2{al.x, al.x}, nc_f2{al.y, al.y});
                    v[0] = sm[0]; v[1] = sm[1];
                }
                if ((XVN % NT == 0) || tid + NT * n < XVN) reinterpret_cast<xv_t*>(Xd)[tid + NT * n] = v;
            }
            }
        }
    };
    // one rotation of the XV schedule inside block cb: the words of block cb + 1 to the idle buffers, the reads of block cb + 2 behind them
    auto xv_stage = [&](int cb, float* An, float* Xn) __attribute__((always_inline)) {
        if constexpr (XVCAND) {
            if (K == 7 && alpha_in != nullptr) {
                if (xv_allok) xv_store(cb + 1, An, Xn, std::true_type{}, std::true_type{});
                else xv_store(cb + 1, An, Xn, std::false_type{}, std::true_type{});
            } else {
                if (xv_allok) xv_store(cb + 1, An, Xn, std::true_type{}, std::false_type{});
                else xv_store(cb + 1, An, Xn, std::false_type{}, std::false_type{});
            }
            if (cb + 2 < n_cb) xv_issue(cb + 2);
        }
    };
    auto issue_group_to = [&](int cbn, auto gtag, f32x4 (&ra)[GA], float (&rx)[GXX]) __attribute__((always_inline)) {
        constexpr int g = decltype(gtag)::value;
        const f32x4* src = wbase + (size_t)cbn * A_VEC;
        nc_static_for<GX>([&](auto ut) __attribute__((always_inline)) {
            constexpr int u = decltype(ut)::value, i = g * GX + u;
            if constexpr (i < NX) {
                // (the item's channel is recomputed -- two scalar instructions -- instead of being held: NX pinned scalars pushed the
                //  two-tap instances, NX = 20, into scalar-register spills whose reloads are vector instructions inside the loop)
                const int ci = min(cbn * CB + (((swave + SW * i) * chunk_magic) >> 20), Cin - 1);           // wave-uniform
                const float* row = xb + (size_t)((unsigned)ci * x_cstride);   // uniform base + 32-bit lane offset
                rx[u] = row[xg[i]];
                if constexpr (IN2) rx[GX + u] = (xb2 + (size_t)((unsigned)ci * x_cstride))[xg[i]];
            }
        });
        nc_static_for<GA>([&](auto ut) __attribute__((always_inline)) {
            constexpr int u = decltype(ut)::value, n = g * GA + u;
            if constexpr (n < NA) {
                // uniform pointer (scalar arithmetic) + the thread index: no per-read vector address arithmetic
                const f32x4* srcn = src + SNT * n;
                ra[u] = srcn[(A_VEC % SNT == 0) ? (unsigned)stid : min((unsigned)stid, (unsigned)(A_VEC - 1 - SNT * n))];
            }
        });
    };
    auto issue_group = [&](int cbn, auto gtag) __attribute__((always_inline)) { issue_group_to(cbn, gtag, ra, rx); };
    auto store_group_from_x = [&](int cbn, float* Ad, float* Xd, auto gtag, auto snake_tag, const f32x4 (&ra)[GA],
                                  const float (&rx)[GXX], auto allok_tag) __attribute__((always_inline)) {
        constexpr int g = decltype(gtag)::value;
        constexpr bool ALLOK = decltype(allok_tag)::value;       // every window item of this block is a real sample: no zero select
        constexpr int IMODE = (int)decltype(snake_tag)::value;   // 0 plain, 1 (true_type) Snake, 2 Encodec input mode
        constexpr bool SNAKE = IMODE == 1;
        nc_static_for<GA>([&](auto ut) __attribute__((always_inline)) {
            constexpr int u = decltype(ut)::value, n = g * GA + u;
            if constexpr (n < NA) {
                const int idx = stid + SNT * n;
                if constexpr (!A_DMA)
                    if ((A_VEC % SNT == 0) || idx < A_VEC) reinterpret_cast<f32x4*>(Ad)[idx] = ra[u];
            }
        });
        float2 al[GX] = {};
        float2 al2[IN2 ? GX : 1];
        if (SNAKE || (IMODE == 2 && (in_mode & 1))) {
            nc_static_for<GX>([&](auto ut) __attribute__((always_inline)) {
                constexpr int u = decltype(ut)::value, i = g * GX + u;
                if constexpr (i < NX) {
                    const int item = swave + SW * i;
                    al[u] = Al[cbn * CB + ((item * chunk_magic) >> 20)];
                    if constexpr (IN2) al2[u] = Al2[cbn * CB + ((item * chunk_magic) >> 20)];
                }
            });
        }
        float v[GX] = {};
        int off[GX];
        nc_static_for<GX>([&](auto ut) __attribute__((always_inline)) {
            constexpr int u = decltype(ut)::value, i = g * GX + u;
            if constexpr (i < NX) {
                const int item = swave + SW * i;
                const int c = (item * chunk_magic) >> 20;
                const int ci = cbn * CB + c;
                const int j = (item - c * nchunk) * 64 + lane;
                const bool ok = ALLOK || ((ci < Cin) & (((okm >> i) & 1u) != 0));  // slots past xw / items past n_items are never read
                if constexpr (IMODE == 2) {
                    float t = rx[u];
                    if (in_mode & 1) t = ((t - in_mu) * in_rs) * al[u].x + al[u].y;   // GroupNorm(1,C) apply (NormConv1d.cs:155)
                    if constexpr (IN2) {   // shortcut + branch (SEANetResnetBlock.cs:84): each normalised with its own statistics, then added
                        float t2 = rx[GX + u];
                        if (in_mode & 1) t2 = ((t2 - in_mu2) * in_rs2) * al2[u].x + al2[u].y;
                        t = t + t2;
                    }
                    if (in_mode & 2) t = nc_eluf(t);
                    v[u] = ok ? t : 0.0f;                                                // the pad zero-extends the ACTIVATED row
                } else {
                    v[u] = ok ? rx[u] : 0.0f;
                }
                off[u] = item * 64 + lane;
                if (s != 1) {
                    const int q = (j * stride_magic) >> 20;
                    off[u] = c * xrow + (j - q * s) * xwp + q;
                }
            }
        });
        if constexpr (SNAKE) {   // Snake of the consumed tensor, two window values per packed instruction (nc_math.h; items past NX: unused)
            // groups of >= 2 pairs: the sine up to its sign and the pairs' chains step by step side by side (nc_math.h; round 5: in the k = 7
            // instances 57 -> 37 `s_nop` and 40 vector instructions fewer per loop, conv_k7 -0.17 ms, bit-identical; NC_LEGACY_ROWS=0
            // at compile time keeps the pair-by-pair form with the parity select)
            if constexpr (GX >= 4 && NC_LEGACY_ROWS != 0) {
                constexpr int NPR = GX / 2;
                nc_f2 xs[NPR], as[NPR], is[NPR];
#pragma unroll
                for (int q = 0; q < NPR; ++q) { xs[q] = nc_f2{v[2 * q], v[2 * q + 1]}; as[q] = nc_f2{al[2 * q].x, al[2 * q + 1].x}; is[q] = nc_f2{al[2 * q].y, al[2 * q + 1].y}; }
                nc_snakef2_m_rows<NPR, NC_LEGACY_ROWS == 1>(xs, as, is);
#pragma unroll
                for (int q = 0; q < NPR; ++q) { v[2 * q] = xs[q][0]; v[2 * q + 1] = xs[q][1]; }
            } else
#pragma unroll
            for (int u = 0; u + 1 < GX; u += 2) nc_snake_pair(v[u], v[u + 1], al[u].x, al[u].y, al[u + 1].x, al[u + 1].y);
            if constexpr (GX & 1) v[GX - 1] = nc_snakef(v[GX - 1], al[GX - 1].x, al[GX - 1].y);
        }
        nc_static_for<GX>([&](auto ut) __attribute__((always_inline)) {
            constexpr int u = decltype(ut)::value, i = g * GX + u;
            if constexpr (i < NX) Xd[off[u]] = v[u];
        });
    };
    auto store_group_from = [&](int cbn, float* Ad, float* Xd, auto gtag, auto snake_tag, const f32x4 (&ra)[GA],
                                const float (&rx)[GXX]) __attribute__((always_inline)) {
        store_group_from_x(cbn, Ad, Xd, gtag, snake_tag, ra, rx, std::false_type{});
    };
    auto store_group = [&](int cbn, float* Ad, float* Xd, auto gtag, auto snake_tag) __attribute__((always_inline)) {
        store_group_from(cbn, Ad, Xd, gtag, snake_tag, ra, rx);
    };
    auto store_group_any = [&](int cbn, float* Ad, float* Xd, auto gtag) __attribute__((always_inline)) {
        if constexpr (!FUSE && !SPEC && !DIST) {
            if (in_mode) { store_group(cbn, Ad, Xd, gtag, std::integral_constant<int, 2>{}); return; }
        }
        // (instances with >= 16 window items per lane only: on the k = 7 instances, NX = 10, whose masks fit the scalar file, the extra
        //  code path cost 1.5 % -- conv_k7 42.2 -> 42.8 ms on one box -- where the two-tap instances gained 2-3 %)
        if (NX >= 16 && tile_allok && (cbn + 1) * CB <= Cin) {   // wave-uniform
            if (alpha_in != nullptr) store_group_from_x(cbn, Ad, Xd, gtag, std::true_type{}, ra, rx, std::true_type{});
            else store_group_from_x(cbn, Ad, Xd, gtag, std::false_type{}, ra, rx, std::true_type{});
            return;
        }
        if (alpha_in != nullptr)
            store_group(cbn, Ad, Xd, gtag, std::true_type{});
        else store_group(cbn, Ad, Xd, gtag, std::false_type{});
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // ---- prologue: stage reduction block 0.  Every global read of the prologue -- the first tile's groups and the operands of
    // the two LDS tables -- is issued before anything waits: one memory round trip in all.
    {
        f32x4 ra0[XVK ? 1 : NG][GA];
        float rx0[XVK ? 1 : NG][GXX];
        if constexpr (XVK) xv_issue(0);
        else
        if (!SPEC || producer)
            nc_static_for<NG>([&](auto g) __attribute__((always_inline)) { issue_group_to(0, g, ra0[decltype(g)::value], rx0[decltype(g)::value]); });
        // Snake alphas of all input channels -> (alpha, 1/alpha), 4 reads per thread in flight per pass
        if (alpha_in != nullptr) {
            const int n_al = n_cb * CB;
            for (int i0 = 0; i0 < n_al; i0 += 4 * NT) {
                float av[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) av[u] = alpha_in[min(i0 + tid + u * NT, Cin - 1)];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i0 + tid + u * NT < n_al) Al[i0 + tid + u * NT] = make_float2(av[u], nc_snake_inv(av[u]));
            }
        }
        if (in_mode & 1) {   // (gamma, beta) of the pending GroupNorm, per input channel, in the Snake table's place
            const int n_al = n_cb * CB;
            for (int i = tid; i < n_al; i += NT) Al[i] = make_float2(p.in_gamma[min(i, Cin - 1)], p.in_beta[min(i, Cin - 1)]);
            if constexpr (IN2)
                for (int i = tid; i < n_al; i += NT) Al2[i] = make_float2(p.in_gamma2[min(i, Cin - 1)], p.in_beta2[min(i, Cin - 1)]);
        }
        for (int i = tid; i < BM; i += NT) {
            const int co = SUB == 2 ? min(((co_tile * BM + i) * p.sub_magic) >> 20, p.sub_cout - 1)
                           : SUB ? min((co_tile * BM + i) >> p.sub_shift, (p.Cout >> p.sub_shift) - 1) : min(co_tile * BM + i, p.Cout - 1);
            const float ao = p.alpha_out ? p.alpha_out[co] : 0.0f;
            Ep[i] = p.bias ? p.bias[co] : 0.0f;
            Ep[BM + i] = ao;
            Ep[2 * BM + i] = nc_snake_inv(ao);
            if constexpr (FUSE) {
                const float ao2 = p.alpha_out2 ? p.alpha_out2[co] : 0.0f;
                Ep[3 * BM + i] = p.bias2[co];
                Ep[4 * BM + i] = ao2;
                Ep[5 * BM + i] = nc_snake_inv(ao2);
            }
        }
        __syncthreads();   // the alpha table is complete before the Snake of the first tile reads it
        if constexpr (XVK) {
            if (K == 7 && alpha_in != nullptr) xv_store(0, As0, Xs0, std::false_type{}, std::true_type{});
            else xv_store(0, As0, Xs0, std::false_type{}, std::false_type{});
        } else
        if (!SPEC || producer)
            nc_static_for<NG>([&](auto g) __attribute__((always_inline)) {
                constexpr int gi = decltype(g)::value;
                if constexpr (!FUSE && !SPEC && !DIST) {
                    if (in_mode) { store_group_from(0, As0, Xs0, g, std::integral_constant<int, 2>{}, ra0[gi], rx0[gi]); return; }
                }
                if (alpha_in != nullptr)
                    store_group_from(0, As0, Xs0, g, std::true_type{}, ra0[gi], rx0[gi]);
                else store_group_from(0, As0, Xs0, g, std::false_type{}, ra0[gi], rx0[gi]);
            });
    }
    __syncthreads();

    // window column of this lane's tile columns (one per 32-column block j): column + the halos of the segments before it
    int sgc[TN], ej[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) sgc[j] = seg_of(col0 + wave * BNW + j * 32 + l31, flat_pc);
#pragma unroll
    for (int j = 0; j < TN; ++j) ej[j] = (sgc[j] - sgc[0]) * flat_hc;
    const int x_lane = wave * BNW + l31 + sgc[0] * flat_hc;
    // B-fragment addressing: MFMA step kp pairs kk = 2kp (lanes 0-31) with kk+1 (lanes 32-63), i.e. tap k0 of channel c0 with the
    // next tap (or tap 0 of the next channel).  The lane-half difference depends only on k0: one base per k0, so a step's address
    // is base[k0] + (wave-uniform offset of (c0, k0)) -- one vector add per step.
    int xt[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int d = (k + 1 < K) ? tap[k + 1] - tap[k] : xrow + tap[0] - tap[k];
        xt[k] = x_lane + (hi ? d : 0);
    }

    constexpr int FD = XVK ? NC_XV_FD : 1;   // fragment reads run FD matrix-core steps ahead (FD + 1 register sets)
    float fa[FD + 1][TM], fb[FD + 1][TN];
    // Ac = current weight buffer + hi*BM (the kk row of this lane half at step 0); xsc = scalar float index of the current window
    // Constant row pitch (XR, round 4): the k = 7 instances with 256-column tiles always stage 5 chunks of 64 slots per channel (a
    // window of 256 + 6 d (+ segment halos) slots, capped at 5 chunks by the staging registers), so xrow == 320 whatever the dilation.
    // With the pitch a compile-time constant the B-fragment address of step kp is a per-tap lane pointer (tap offset, segment halo and
    // buffer parity folded in: K * TN pointers, flipped between the two window buffers once per block) plus an IMMEDIATE offset:
    // no vector ALU instruction per step, where the generic form spends three (found in the ISA: `v_add_lshl_u32`, a copy and the
    // halo add in front of every pair of `ds_read_b32`; vector instructions are issued in the matrix pipe's time on this SIMD).
    // The launcher's xrow is checked at run time; any other pitch (a strided k = 7 layer) takes the generic form below.
    // The two-tap instances (sub-pixel up-convolutions: a window of 256 + 1 (+ halos) slots) have the same pitch for the same reason.
    // 128-column tiles (TN = 1: the wide fused units): 128 + 6 d <= 182 slots, 3 chunks, pitch 192.
    constexpr bool XRCAND = ((K == 7 && SUB == 0) || K == 2) && (TN == 2 || TN == 1) && NW == 4 && !IN2 && !SPEC && !DIST;
    constexpr int XROWC = TN == 2 ? 320 : 192;
    // (The same for the strided down-convolutions -- pitch, phase-row pitch and tap offsets all follow from (K, tile width), two lane
    //  pointers per column block -- and for the K = 1 instances was built and measured on one box: strided SLOWER, conv_down 2.82 ->
    //  2.97 ms on DAC, 4.22 -> 4.78 on SNAC 44 kHz; K = 1 no change.  Not kept.)
    const float* xq[XRCAND ? TN : 1][XRCAND ? K : 1];
    if constexpr (XRCAND) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int k = 0; k < K; ++k) xq[j][k] = smem + 2 * A_FLOATS + xt[k] + tap[k] + ej[j];
    }
    auto load_frag_x = [&](const float* Ac, int xsc, auto kp_tag, auto xr_tag) __attribute__((always_inline)) {
        constexpr int kp = decltype(kp_tag)::value;
        constexpr int c0 = (2 * kp) / K, k0 = (2 * kp) % K;
        nc_load_a_frag<TM>(Ac + 2 * kp * BM, l31, fa[kp % (FD + 1)]);   // Ac carries the lane part: immediate offsets only
        if constexpr (decltype(xr_tag)::value == 1) {
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[kp % (FD + 1)][j] = xq[j][k0][c0 * XROWC + j * 32];
        } else {
            // (readfirstlane pins the wave-uniform part in a scalar register: vector + scalar is then one add per step)
            const int o = xt[k0] + __builtin_amdgcn_readfirstlane(xsc + c0 * xrow + tap[k0]);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[kp % (FD + 1)][j] = smem[(j == 0 ? o : o + ej[j]) + j * 32];
        }
    };
    auto load_frag = [&](const float* Ac, int xsc, auto kp_tag) __attribute__((always_inline)) { load_frag_x(Ac, xsc, kp_tag, std::integral_constant<int, 0>{}); };

    if constexpr (SPEC) {
        static_assert(!SPEC || !FUSE, "the fused tail has workgroup barriers: not combined with producer waves");
        if (producer) {
            // Producer waves: their own loop (no accumulators live), one barrier per reduction block like the consumers.  Each
            // block is staged in NG groups, a group's reads all in flight before its Snake + LDS writes.
            // The reads of group q+1 are issued before group q is transformed and written (two register sets), and the producer
            // waves run at raised issue priority: their short vector instructions are not starved by the consumers' matrix stream.
            __builtin_amdgcn_s_setprio(3);
            f32x4 pa[2][GA];
            float px[2][GXX];
            if (n_cb > 1) issue_group_to(1, std::integral_constant<int, 0>{}, pa[0], px[0]);
            for (int cb = 0; cb < n_cb; ++cb) {
                const int cur = cb & 1;
                float* const An = As0 + (cur ^ 1) * A_FLOATS;
                float* const Xn = Xs0 + (cur ^ 1) * xbuf;
                if (cb + 1 < n_cb)
                    nc_static_for<NG>([&](auto g) __attribute__((always_inline)) {
                        constexpr int gi = decltype(g)::value;
                        // next group in flight: group gi+1 of this block, or group 0 of the block after
                        if constexpr (gi + 1 < NG) issue_group_to(cb + 1, std::integral_constant<int, gi + 1>{}, pa[(gi + 1) & 1], px[(gi + 1) & 1]);
                        else if (cb + 2 < n_cb) issue_group_to(cb + 2, std::integral_constant<int, 0>{}, pa[(gi + 1) & 1], px[(gi + 1) & 1]);
                        if (alpha_in != nullptr) store_group_from(cb + 1, An, Xn, g, std::true_type{}, pa[gi & 1], px[gi & 1]);
                        else store_group_from(cb + 1, An, Xn, g, std::false_type{}, pa[gi & 1], px[gi & 1]);
                    });
                __syncthreads();
            }
            return;
        }
    }
    if constexpr (DIST) {
        static_assert(!DIST || NG == 1, "distributed staging holds the whole next block in one register group");
        // one staging unit: weight vector `un` (un < NA) or window item un - NA of block cbn
        auto stage_unit = [&](int cbn, float* Ad, float* Xd, auto unit_tag, auto snake_tag) __attribute__((always_inline)) {
            constexpr int un = decltype(unit_tag)::value;
            constexpr bool SNAKE = decltype(snake_tag)::value;
            if constexpr (un < NA) {
                // (threads past the tile's end hold a copy of its last vector and rewrite it: branch-free)
                const int idx = (A_VEC % SNT == 0) ? stid + SNT * un : min(stid + SNT * un, A_VEC - 1);
                reinterpret_cast<f32x4*>(Ad)[idx] = ra[un];
            } else {
                constexpr int i = un - NA;
                const int item = swave + SW * i;
                const int c = xc[i];
                const int j = (item - c * nchunk) * 64 + lane;
                const bool ok = (cbn * CB + c < Cin) & (((okm >> i) & 1u) != 0);
                float v = ok ? rx[i] : 0.0f;
                if constexpr (SNAKE) {
                    const float2 al = Al[cbn * CB + c];
                    v = nc_snakef(v, al.x, al.y);
                }
                int off = item * 64 + lane;
                if (s != 1) {
                    const int q = (j * stride_magic) >> 20;
                    off = c * xrow + (j - q * s) * xwp + q;
                }
                Xd[off] = v;
            }
        };
        auto run_loop = [&](auto snake_tag) __attribute__((always_inline)) {
            constexpr int UN = NA + NX;                       // staging units per block
            constexpr int KS0 = KP >= 12 ? KP / 3 : 0;        // the reads get the first third of the step to land
            for (int cb = 0; cb < n_cb; ++cb) {
                const int cur = cb & 1;
                const float* Ac = As0 + cur * A_FLOATS + hi * BM + nc_a_lane_off<TM>(l31);
                const int Xc = 2 * A_FLOATS + cur * xbuf;
                float* const An = As0 + (cur ^ 1) * A_FLOATS;
                float* const Xn = Xs0 + (cur ^ 1) * xbuf;
                const int cbn = min(cb + 1, n_cb - 1);
                issue_group(cbn, std::integral_constant<int, 0>{});
                nc_static_for<FD>([&](auto d) __attribute__((always_inline)) {
                    if constexpr (decltype(d)::value < KP) load_frag(Ac, Xc, d);
                });
                nc_static_for<KP>([&](auto d) __attribute__((always_inline)) {
                    constexpr int kp = decltype(d)::value;
                    if constexpr (kp + FD < KP) load_frag(Ac, Xc, std::integral_constant<int, kp + FD>{});
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kp % (FD + 1)][i], fb[kp % (FD + 1)][j], acc[i][j], 0, 0, 0);
                    constexpr int lo = kp < KS0 ? 0 : (kp - KS0) * UN / (KP - KS0), hi_u = kp < KS0 ? 0 : (kp - KS0 + 1) * UN / (KP - KS0);
                    nc_static_for<hi_u - lo>([&](auto e) __attribute__((always_inline)) {
                        stage_unit(cbn, An, Xn, std::integral_constant<int, lo + decltype(e)::value>{}, snake_tag);
                    });
                    // deal this step's staging instructions into the shadows of its matrix-core instructions
#pragma unroll
                    for (int m = 0; m < TM * TN; ++m) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // one MFMA
                        __builtin_amdgcn_sched_group_barrier(0x006, 8, 0);    // up to 8 VALU / SALU
                        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);    // an LDS write
                    }
                });
                __syncthreads();
            }
        };
        if (alpha_in != nullptr) run_loop(std::true_type{});
        else run_loop(std::false_type{});
    } else {
    auto main_loop = [&](auto xr_tag) __attribute__((always_inline)) {
    constexpr int XR = decltype(xr_tag)::value;   // 0 generic, 1 constant pitch
    if constexpr (XVCAND && XR == 1) {
        if (use_xv && n_cb > 1) xv_issue(1);      // (block 0 came through the prologue; see the rotating schedule below)
    }
    if constexpr (DUO) {
        if (sub == 1) __syncthreads();            // the second sub-workgroup starts half a block late ...
    }
    for (int cb = 0; cb < n_cb; ++cb) {
        const int cur = cb & 1;
        const float* Ac = As0 + cur * A_FLOATS + hi * BM + nc_a_lane_off<TM>(l31);
        const int Xc = 2 * A_FLOATS + cur * xbuf;   // scalar index of the window buffer in smem; the lane part lives in xt[]
        float* const An = As0 + (cur ^ 1) * A_FLOATS;
        float* const Xn = Xs0 + (cur ^ 1) * xbuf;
        const bool more = cb + 1 < n_cb;
        if constexpr (XR == 1) {
            if (cb > 0) {   // the tap pointers follow the window buffer of this block
                const int dx = cur ? xbuf : -xbuf;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int k = 0; k < K; ++k) xq[j][k] += dx;
            }
        }

        if constexpr (SPEC) {
            nc_static_for<FD>([&](auto d) __attribute__((always_inline)) {
                if constexpr (decltype(d)::value < KP) load_frag(Ac, Xc, d);
            });
            nc_static_for<KP>([&](auto d) __attribute__((always_inline)) {
                constexpr int kp = decltype(d)::value;
                if constexpr (kp + FD < KP) load_frag(Ac, Xc, std::integral_constant<int, kp + FD>{});
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kp % (FD + 1)][i], fb[kp % (FD + 1)][j], acc[i][j], 0, 0, 0);
            });
        } else
        nc_static_for<NSEG>([&](auto seg_tag) __attribute__((always_inline)) {
            constexpr int seg = decltype(seg_tag)::value;
            if constexpr (seg == 0) NC_STAMP(cb, 0);
            else if constexpr (seg <= 3) NC_STAMP(cb, 2 * seg - 1);
#if NC_STAGE_PRIO
            if (more) __builtin_amdgcn_s_setprio(NC_STAGE_PRIO);   // (experiment: the staging run at raised issue priority)
#endif
            if constexpr (DUO && seg == NSEG / 2) __syncthreads();   // the partner's block boundary
            if (more) {
                if constexpr (XVCAND && XR == 1) {   // (the XV-only instances)
                    // rotating schedule: ONE register set.  At the head of segment NC_XV_STORE_SEG the words of block cb + 1 (in flight
                    // since the same point of the block before: a whole block of latency cover) go to the idle LDS buffers, and the
                    // reads of block cb + 2 are issued straight behind them (that block's buffers are the ones being read now, but its
                    // words stay in registers until this block's closing barrier has passed).
                    if constexpr (seg == (DUO ? 0 : NC_XV_STORE_SEG)) xv_stage(cb, An, Xn);   // (DUO: the run opens the block, beside the partner's segments 2-3)
                } else {
                    if constexpr (seg >= 1) store_group_any(cb + 1, An, Xn, std::integral_constant<int, (seg >= 1 ? seg - 1 : 0)>{});
                    if constexpr (seg < NG) issue_group(cb + 1, seg_tag);
                }
            }
#if NC_STAGE_PRIO
            if (more) __builtin_amdgcn_s_setprio(0);
#endif
            if constexpr (seg >= 1 && seg <= 3) NC_STAMP(cb, 2 * seg);
            // ---- matrix-core steps of this segment, ascending kk; fragments of step kp+1 are read before
            //      the MFMAs of step kp are issued (register double buffer fa/fb)
            constexpr int kp_lo = seg * KP / NSEG, kp_hi = (seg + 1) * KP / NSEG;
            if constexpr (seg == 0)
                nc_static_for<FD>([&](auto d) __attribute__((always_inline)) {
                    if constexpr (decltype(d)::value < KP) load_frag_x(Ac, Xc, d, xr_tag);
                });
            nc_static_for<kp_hi - kp_lo>([&](auto d) __attribute__((always_inline)) {
                constexpr int kp = kp_lo + decltype(d)::value;
                if constexpr (kp + FD < KP) load_frag_x(Ac, Xc, std::integral_constant<int, kp + FD>{}, xr_tag);
#if defined(NC_XV_NOSB)
                if constexpr (!XVK)
#endif
                __builtin_amdgcn_sched_barrier(0);   // keep the fragment reads of step kp+1 ahead of the MFMAs of step kp
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kp % (FD + 1)][i], fb[kp % (FD + 1)][j], acc[i][j], 0, 0, 0);
            });
        });
        if constexpr (XVCAND && XR == 1 && NC_XV_STORE_SEG >= NSEG && !DUO) {   // (experiment: the words written behind the block's last matrix-core step)
            if (more && use_xv) xv_stage(cb, An, Xn);
        }
        NC_STAMP(cb, 7);
        __syncthreads();
    }
    if constexpr (DUO) {
        if (sub == 0) __syncthreads();            // ... and the first passes the barrier once more at the end: equal counts
    }
    };
    if constexpr (XVK) {
        main_loop(std::integral_constant<int, 1>{});   // (the host guarantees the constant pitch)
    } else if constexpr (XRCAND) {
        if (xrow == XROWC && s == 1 && !(p.epi & EPI_NO_XR)) main_loop(std::integral_constant<int, 1>{});
        else main_loop(std::integral_constant<int, 0>{});
    } else {
        main_loop(std::integral_constant<int, 0>{});
    }
    }
    // ---- epilogue: D[row = (r&3) + 8*(r>>2) + 4*hi][col = l31]
    // Element (row block ib, register r, column j) of this lane lives at tile_base + lane_off[j] + R(ib,r)*cstride with
    // R = 32*ib + (r&3) + 8*(r>>2) a compile-time row: one uniform 64-bit base, 32-bit lane offsets (the host bounds them).
    constexpr bool SUBG = SUB == 2;
    const int sh = SUB == 1 ? p.sub_shift : 0, smask = SUBG ? p.sub_stride - 1 : (1 << sh) - 1;   // smask: last phase of a packed channel
    const int64_t tile_base = (int64_t)b * p.y_bstride + (SUBG ? (int64_t)0 : (int64_t)((co_tile * BM) >> sh) * p.y_cstride);
    // SUBG: (channel, phase) of packed row x -- scalar arithmetic on compile-time tile rows
    const int sub_s = SUBG ? p.sub_stride : 1, sub_m = SUBG ? p.sub_magic : 0, row0 = co_tile * BM;
    auto sub_ch = [&](int x) __attribute__((always_inline)) -> int { return (x * sub_m) >> 20; };
    auto sub_ph = [&](int x) __attribute__((always_inline)) -> int { return x - sub_ch(x) * sub_s; };
    const unsigned cstride = (unsigned)p.y_cstride;
    const int rows_left = p.Cout - co_tile * BM - 4 * hi;   // row R of this lane half is inside the tensor iff R < rows_left
    // offset of tile row R (lane half 0) from the tile base; the lane half's 4 extra rows are folded into lane_off
    auto roff = [&](int R) __attribute__((always_inline)) -> unsigned {
        if constexpr (SUBG) {   // row R of lane half 0, row R + 4 of lane half 1
            const unsigned o0 = (unsigned)sub_ch(row0 + R) * cstride + (unsigned)sub_ph(row0 + R);
            const unsigned o1 = (unsigned)sub_ch(row0 + R + 4) * cstride + (unsigned)sub_ph(row0 + R + 4);
            return hi ? o1 : o0;
        } else if constexpr (SUB != 0) return (unsigned)(R >> sh) * cstride + (unsigned)(R & smask);
        else return (unsigned)R * cstride;
    };
    unsigned lane_off[TN];
    bool okc[TN];
    int tcol[TN];
    bool cols_ok = true;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = col0 + wave * BNW + j * 32 + l31 - sgc[j] * flat_pc;   // column within its clip b + sgc[j]
        const int t = col * p.y_tstride + p.y_toff + phase;
        const bool clip_ok = (col < p.n_cols) & (b + sgc[j] < p.Bc) & !duo_dead;
        const unsigned clip_off = (unsigned)min(sgc[j], p.Bc - 1 - b) * (unsigned)p.y_bstride;   // (clamped: masked reads stay inside the tensor)
        if constexpr (SUB != 0) {   // t = sample of sub-row r = 0; row R adds (R + 4*hi) & smask.  Time bounds are checked per row.
            okc[j] = clip_ok;
            tcol[j] = t;
            if constexpr (SUBG) lane_off[j] = clip_off + (unsigned)t;
            else lane_off[j] = clip_off + (unsigned)((4 * hi) >> sh) * cstride + (unsigned)((4 * hi) & smask) + (unsigned)t;
            cols_ok = cols_ok & okc[j] & (t >= 0) & (t + smask < p.Tout);
        } else {
            okc[j] = clip_ok & (t >= 0) & (t < p.Tout);
            tcol[j] = min(max(t, 0), p.Tout - 1);
            lane_off[j] = clip_off + (unsigned)(4 * hi) * cstride + (unsigned)tcol[j];
            cols_ok = cols_ok & okc[j];
        }
    }
    // element (row R of lane half hi, column j) is inside the tensor
    auto okrow = [&](int j, int R) __attribute__((always_inline)) -> bool {
        if constexpr (SUB != 0) {
            int tq;
            if constexpr (SUBG) tq = tcol[j] + (hi ? sub_ph(row0 + R + 4) : sub_ph(row0 + R));
            else tq = tcol[j] + ((R + 4 * hi) & smask);
            return okc[j] & (R < rows_left) & (tq >= 0) & (tq < p.Tout);
        } else {
            return okc[j] & (R < rows_left);
        }
    };
    const bool tile_full = __builtin_amdgcn_ballot_w64(!cols_ok) == 0 && p.Cout - co_tile * BM >= BM;   // wave-uniform
    // Residual operands of the 32-row block `ib`, all 16*TN reads issued back to back (one memory round trip per row block).
    // full_tag: every element of this wave's part of the tile is inside the tensor (no per-element predicates: straight-line code)
    auto load_res = [&](int ib, float (&rv)[16][TN], const float* res_p, auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
        const float* rt = res_p + tile_base;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int R = ib * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (FULL) {
                    rv[r][j] = rt[lane_off[j] + roff(R)];
                } else if constexpr (SUB != 0) {
                    rv[r][j] = okrow(j, R) ? rt[lane_off[j] + roff(R)] : 0.0f;
                } else {   // branch-free: the address is clamped into the tile's valid rows / columns, the value masked
                    const float val = rt[lane_off[j] + (unsigned)max(min(R, rows_left - 1), -4 * hi) * cstride];
                    rv[r][j] = (okc[j] & (R < rows_left)) ? val : 0.0f;
                }
            }
        }
    };
    // v[j][r] = (v[j][r] + bias) (+ residual | residual + noise * .): the value of the canonical chain before the activation
    auto add_rows = [&](int ib, f32x16 (&v)[TN], const float (&rv)[16][TN], const float (&nz)[TN], const float* bias_t, auto has_res,
                        auto gen_tag) __attribute__((always_inline)) {
        const bool noise = decltype(gen_tag)::value && (p.epi & EPI_NOISE) != 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float bias = bias_t[ib * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float val = v[j][r] + bias;
                if (noise) val = rv[r][j] + nz[j] * val;
                else if (has_res) val = val + rv[r][j];
                v[j][r] = val;
            }
        }
    };
    // store one 32-row block `ib`: (Snake) (tanh) -> y, or RVQ accumulate.  No global read happens here except the RVQ operands,
    // which are read per row quad ahead of the quad's stores.
    auto store_rows = [&](int ib, const f32x16 (&v)[TN], const float* ao_t, auto snake, auto gen_tag, auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
        // opaque copy of the lane offsets: keeps the compiler from carrying the read phase's 16*TM*TN addresses over to the stores
        unsigned lane_off_s[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            lane_off_s[j] = lane_off[j];
            asm volatile("" : "+v"(lane_off_s[j]));
        }
        const bool rvq = decltype(gen_tag)::value && (p.epi & EPI_RVQ) != 0;
        float* const yt = (rvq ? p.rvq_zq : p.y) + tile_base;
        float* const st = (rvq && p.rvq_res) ? p.rvq_res + tile_base : nullptr;
        nc_static_for<4>([&](auto qt) __attribute__((always_inline)) {
            constexpr int rq = decltype(qt)::value;
            float zv[4][TN], sv[4][TN];
            if (rvq) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int R = ib * 32 + rr + 8 * rq;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const bool ok = FULL || okrow(j, R);
                        const unsigned o = lane_off_s[j] + roff(R);
                        zv[rr][j] = ok ? yt[o] : 0.0f;
                        sv[rr][j] = (ok && st) ? st[o] : 0.0f;
                    }
                }
            }
            float vq[4][TN];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int j = 0; j < TN; ++j) vq[rr][j] = v[j][4 * rq + rr];
            if constexpr (XVK && NC_XV_SNAKE_ILV) {   // (XV-only instances: sign-free sine, the 2 x TN chains step by step side by side)
                if (snake) {
                    nc_f2 xs[2 * TN], as[2 * TN], is[2 * TN];
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        const int R = ib * 32 + 2 * h2 + 8 * rq + 4 * hi;
                        const nc_f2 a = {ao_t[R], ao_t[R + 1]}, iv = {ao_t[BM + R], ao_t[BM + R + 1]};
#pragma unroll
                        for (int j = 0; j < TN; ++j) { xs[h2 * TN + j] = nc_f2{vq[2 * h2][j], vq[2 * h2 + 1][j]}; as[h2 * TN + j] = a; is[h2 * TN + j] = iv; }
                    }
                    nc_snakef2_m_rows<2 * TN>(xs, as, is);
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                        for (int j = 0; j < TN; ++j) { vq[2 * h2][j] = xs[h2 * TN + j][0]; vq[2 * h2 + 1][j] = xs[h2 * TN + j][1]; }
                }
            } else
            if (snake) {   // Snake of the consuming layer, two rows per packed instruction (nc_math.h)
#pragma unroll
                for (int rr = 0; rr < 4; rr += 2) {
                    const int R = ib * 32 + rr + 8 * rq + 4 * hi;
                    const float a0 = ao_t[R], i0 = ao_t[BM + R], a1 = ao_t[R + 1], i1 = ao_t[BM + R + 1];
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if constexpr (XVK) nc_snake_pair_m(vq[rr][j], vq[rr + 1][j], a0, i0, a1, i1);   // (sign-free sine: nc_math.h)
                        else nc_snake_pair(vq[rr][j], vq[rr + 1][j], a0, i0, a1, i1);
                    }
                }
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int R = ib * 32 + rr + 8 * rq;     // D row of register r = 4*rq + rr (plus 4*hi)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (!(FULL || okrow(j, R))) continue;
                    const unsigned o = lane_off_s[j] + roff(R);
                    float val = vq[rr][j];
                    if (decltype(gen_tag)::value && (p.epi & EPI_TANH)) val = nc_tanhf(val);
                    if (rvq) {
                        yt[o] = zv[rr][j] + val;
                        if (st) st[o] = sv[rr][j] - val;
                    } else {
                        yt[o] = val;
                    }
                }
            }
        });
    };

    // Generic / edge-tile emitter (row-partial or column-partial tiles, RVQ accumulate, noise injection, tanh): rows go out in
    // quads with per-element predicates; the global reads of a quad are issued before its first store.
    auto emit_rows_quad = [&](int ib, const f32x16 (&v)[TN], const float* bias_t, const float* ao_t, bool snake, const float* res_p)
        __attribute__((always_inline)) {
        const bool rvq = (p.epi & EPI_RVQ) != 0, noise = (p.epi & EPI_NOISE) != 0;
        float* const yt = (rvq ? p.rvq_zq : p.y) + tile_base;
        float* const st = (rvq && p.rvq_res) ? p.rvq_res + tile_base : nullptr;
        const float* const rt = res_p ? res_p + tile_base : nullptr;
        nc_static_for<4>([&](auto qt) __attribute__((always_inline)) {
            constexpr int rq = decltype(qt)::value;
            float rv[4][TN], zv[4][TN], sv[4][TN];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int R = ib * 32 + rr + 8 * rq;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const bool ok = okrow(j, R);
                    const unsigned o = lane_off[j] + roff(R);
                    rv[rr][j] = (ok && rt) ? rt[o] : 0.0f;
                    zv[rr][j] = (ok && rvq) ? yt[o] : 0.0f;
                    sv[rr][j] = (ok && st) ? st[o] : 0.0f;
                }
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int R = ib * 32 + rr + 8 * rq;
                const float bias = bias_t[R + 4 * hi], ao = ao_t[R + 4 * hi], ao_inv = ao_t[BM + R + 4 * hi];
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (!okrow(j, R)) continue;
                    const unsigned o = lane_off[j] + roff(R);
                    float val = v[j][4 * rq + rr] + bias;
                    if (noise) val = rv[rr][j] + p.noise[(int64_t)b * p.noise_bstride + tcol[j] + (SUB == 1 ? ((R + 4 * hi) & smask) : 0)] * val;   // (the host keeps noise rows off the SUB == 2 form)
                    else if (rt) val = val + rv[rr][j];
                    if (snake) val = nc_snakef(val, ao, ao_inv);
                    if (p.epi & EPI_TANH) val = nc_tanhf(val);
                    if (rvq) {
                        yt[o] = zv[rr][j] + val;
                        if (st) st[o] = sv[rr][j] - val;
                    } else {
                        yt[o] = val;
                    }
                }
            }
        });
    };

    if constexpr (!FUSE && !SPEC && !XVK) {
        // GroupNorm(1,C) block sums of the output (Encodec's NormConv1d, NormConv1d.cs:155): every 32x32 accumulator tile is reduced in
        // registers in the canonical order of nc_gn.h and leaves ONE (S1, S2) pair -- the tensor is not read back for its statistics.
        if (p.gn_part != nullptr) {
            const int gn_n = p.gn_nrb * p.gn_ncb;
            double* const gp = p.gn_part + (int64_t)b * gn_n * 2;
            auto gn_blocks = [&](auto full_tag) __attribute__((always_inline)) {
                constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        float vv[16];
                        unsigned okm16 = 0;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int R = i * 32 + (r & 3) + 8 * (r >> 2);
                            vv[r] = acc[i][j][r] + Ep[R + 4 * hi];
                            if constexpr (!FULL)
                                if (okrow(j, R)) okm16 |= 1u << r;
                        }
                        double s1, s2;
                        nc_gn_slot_sums<FULL>(vv, okm16, s1, s2);
                        nc_gn_butterfly(s1, s2);
                        // (flattened axis: the tile's 32 columns belong to sample b + sg, wave-uniform since the pitch is a multiple of 32)
                        const int sg = __builtin_amdgcn_readfirstlane(sgc[j]);
                        const int rbk = co_tile * TM + i, cbk = (col0 + wave * BNW + j * 32 - sg * flat_pc) >> 5;
                        if (lane == 0 && rbk < p.gn_nrb && cbk < p.gn_ncb && b + sg < p.Bc)
                            nc_gn_store_partial(gp + ((int64_t)sg * gn_n + (int64_t)rbk * p.gn_ncb + cbk) * 2, s1, s2);
                        __builtin_amdgcn_sched_barrier(0);   // one block at a time: the sums of several blocks in flight spill
                    }
            };
            if (tile_full) gn_blocks(std::true_type{});
            else gn_blocks(std::false_type{});
            // the last workgroup of the sample to arrive turns the block sums into (mean, rstd): no follow-up launch.  (Reads -- the
            // arrival's returned count, the last arriver's loads -- all precede this workgroup's first output store.)
            if (p.gn_count != nullptr) {
                if (flatm) {   // arrivals counted in block sums: this tile holds inc[m] of them for sample b + m
                    const int rb = min(TM, p.gn_nrb - co_tile * TM), f0 = t_tile * BN;
                    unsigned inc[4];
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const int lo = max(f0, (b + m) * flat_pc), hi_c = min(f0 + BN, (b + m + 1) * flat_pc);
                        inc[m] = (b + m < p.Bc && hi_c > lo && rb > 0) ? (unsigned)(((hi_c - lo) >> 5) * rb) : 0u;
                    }
                    nc_gn_arrive_blocks(gp, p.gn_count + b, p.gn_stats + 2 * b, gn_n, p.gn_n, inc[0], inc[1], inc[2], inc[3]);
                } else {
                    nc_gn_arrive_and_finish(gp, p.gn_count + b, p.gn_stats + 2 * b, gn_n, (unsigned)(p.n_co_tiles * p.n_t_tiles), p.gn_n);
                }
            }
        }
    }
    if constexpr (!FUSE) {
        // phase A: every global read of the epilogue (residual tile, noise row), folded into the accumulators; phase B: stores only.
        // The rare RVQ-accumulate / noise-injection epilogues take the generic instance, everything else the lean one.
        auto run_epilogue = [&](auto has_res, auto snake) __attribute__((always_inline)) {
            constexpr std::false_type gen_tag{};
            constexpr std::true_type full_tag{};
            const float nz[TN] = {};
            constexpr int NRV = TM * TN >= 6 ? 1 : 2;   // read one row block ahead where the registers allow it
            float rv[NRV][16][TN];
            if (has_res) load_res(0, rv[0], p.res, full_tag);
            nc_static_for<TM>([&](auto it) __attribute__((always_inline)) {
                constexpr int i = decltype(it)::value;
                if constexpr (NRV == 2 && i + 1 < TM)
                    if (has_res) load_res(i + 1, rv[(i + 1) & 1], p.res, full_tag);
                add_rows(i, acc[i], rv[i % NRV], nz, Ep, has_res, gen_tag);
                if constexpr (NRV == 1 && i + 1 < TM)
                    if (has_res) load_res(i + 1, rv[0], p.res, full_tag);
            });
            nc_static_for<TM>([&](auto it) __attribute__((always_inline)) {
                store_rows(decltype(it)::value, acc[decltype(it)::value], Ep + BM, snake, gen_tag, full_tag);
            });
        };
        // Full tiles of the plain epilogues run straight-line code (flags folded at compile time); edge tiles and the RVQ / noise /
        // tanh epilogues (small layers) take the quad emitter.
        const bool has_res = p.res != nullptr, snake = p.alpha_out != nullptr;
        if ((p.epi & (EPI_RVQ | EPI_NOISE | EPI_TANH)) || !tile_full) {
#pragma unroll
            for (int i = 0; i < TM; ++i) emit_rows_quad(i, acc[i], Ep, Ep + BM, snake, p.res);
        } else if (has_res) {
            if (snake) run_epilogue(std::true_type{}, std::true_type{});
            else run_epilogue(std::true_type{}, std::false_type{});
        } else {
            if (snake) run_epilogue(std::false_type{}, std::true_type{});
            else run_epilogue(std::false_type{}, std::false_type{});
        }
    } else {
        // ---- fused ResidualUnit tail (ResidualUnit.cs:29-34,50-59): this block holds ALL channels of
        //      h_pre = conv7(snake(x)) for its columns (n_co_tiles == 1, Cin == Cout == BM), so
        //          y = x + W1 . snake_a2(h_pre + b7) + b1
        //      is finished here: the activation never leaves the registers.  The 1x1 contraction runs on the
        //      matrix cores with B fragments taken from the accumulators: v_permlane32_swap turns the pair of
        //      D registers holding rows (2m, 2m+1) of one lane half into the B operands of k-steps m and m+2.
        static_assert(!FUSE || K == 7, "the fused tail belongs to the k=7 residual-unit convolution");
        // 1) h = snake(h_pre + b7, a2), in place
        if constexpr (XVK && NC_XV_SNAKE_ILV) {   // (XV-only instances: sign-free sine, eight chains step by step side by side -- pair by
                                                  //  pair the compiler serialises them with an `s_nop` between every two instructions)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int rg = 0; rg < 16; rg += 8) {
                    nc_f2 xs[4 * TN], as[4 * TN], is[4 * TN];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int r = rg + 2 * q;
                        const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        const nc_f2 b = {Ep[row], Ep[row + 1]}, a = {Ep[BM + row], Ep[BM + row + 1]}, iv = {Ep[2 * BM + row], Ep[2 * BM + row + 1]};
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            xs[q * TN + j] = nc_f2{acc[i][j][r], acc[i][j][r + 1]} + b;
                            as[q * TN + j] = a; is[q * TN + j] = iv;
                        }
                    }
                    nc_snakef2_m_rows<4 * TN>(xs, as, is);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            acc[i][j][rg + 2 * q] = xs[q * TN + j][0];
                            acc[i][j][rg + 2 * q + 1] = xs[q * TN + j][1];
                        }
                }
        } else
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {   // rows r, r + 1 of a register quad: two values per packed instruction (nc_math.h)
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const float b0 = Ep[row], a0 = Ep[BM + row], i0 = Ep[2 * BM + row];
                const float b1 = Ep[row + 1], a1 = Ep[BM + row + 1], i1 = Ep[2 * BM + row + 1];
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    float x0 = acc[i][j][r] + b0, x1 = acc[i][j][r + 1] + b1;
                    if constexpr (XVK) nc_snake_pair_m(x0, x1, a0, i0, a1, i1);
                    else nc_snake_pair(x0, x1, a0, i0, a1, i1);
                    acc[i][j][r] = x0;
                    acc[i][j][r + 1] = x1;
                }
            }
        // 2) accumulator layout -> B-operand layout, in place
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int qe = 0; qe < 8; ++qe) {
                    const int r0 = 4 * (qe >> 1) + 2 * (qe & 1);
                    auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i][j][r0]), __float_as_uint(acc[i][j][r0 + 1]),
                                                               false, false);
                    acc[i][j][r0] = __uint_as_float(sw[0]);      // rows (2m, 2m+1) of the low half : k-step m = 4q+e
                    acc[i][j][r0 + 1] = __uint_as_float(sw[1]);  // rows of the high half           : k-step m+2
                }
        // Wide units (TM > 4: C = 192 / 256, one 128-column tile per workgroup): W1 does not fit the LDS at two workgroups per CU, so
        // its 32-row blocks are streamed through two LDS buffers, one block ahead: the global reads of block i2+1 (weights and skip
        // operand) are issued BEFORE the stores of block i2 and land under the matrix-core chain of block i2 + 1's predecessor.
        if constexpr (TM > 4) {
            constexpr int W1V = BM * 32 / 4;                    // float4 words per row block
            constexpr int NW1 = (W1V + NT - 1) / NT;
            const f32x4* w2 = reinterpret_cast<const f32x4*>(p.w2);
            f32x4 wr[NW1];
            auto w1_issue = [&](int i2) __attribute__((always_inline)) {
#pragma unroll
                for (int n = 0; n < NW1; ++n) wr[n] = w2[(size_t)i2 * W1V + min(tid + NT * n, W1V - 1)];
            };
            auto w1_store = [&](float* dstf) __attribute__((always_inline)) {
                f32x4* dst = reinterpret_cast<f32x4*>(dstf);
#pragma unroll
                for (int n = 0; n < NW1; ++n)
                    if (W1V % NT == 0 || tid + NT * n < W1V) dst[tid + NT * n] = wr[n];
            };
            float rv2[2][16][TN];
            w1_issue(0);
            if (tile_full) load_res(0, rv2[0], p.res, std::true_type{});
            w1_store(smem);
            __syncthreads();
            nc_static_for<TM>([&](auto i2t) __attribute__((always_inline)) {
                constexpr int i2 = decltype(i2t)::value;
                if constexpr (i2 + 1 < TM) {
                    w1_issue(i2 + 1);
                    if (tile_full) load_res(i2 + 1, rv2[(i2 + 1) & 1], p.res, std::true_type{});
                }
                f32x16 acc2[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[j][r] = 0.0f;
                const float* W1s = smem + (i2 & 1) * (BM * 32) + hi * 32 + l31;
                nc_static_for<BM / 2>([&](auto kpt) __attribute__((always_inline)) {
                    constexpr int kp = decltype(kpt)::value;
                    constexpr int i = kp / 16, m = kp % 16;
                    constexpr int reg = 4 * (m >> 2) + 2 * (m & 1) + ((m >> 1) & 1);
                    const float a = W1s[2 * kp * 32];
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc2[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, acc[i][j][reg], acc2[j], 0, 0, 0);
                });
                if constexpr (i2 + 1 < TM) w1_store(smem + ((i2 + 1) & 1) * (BM * 32));   // the other buffer: its readers passed the last barrier
                const float nz0[TN] = {};
                if (!tile_full) {
                    emit_rows_quad(i2, acc2, Ep + 3 * BM, Ep + 4 * BM, p.alpha_out2 != nullptr, p.res);
                } else {
                    add_rows(i2, acc2, rv2[i2 & 1], nz0, Ep + 3 * BM, std::true_type{}, std::false_type{});
                    if (p.alpha_out2 != nullptr) store_rows(i2, acc2, Ep + 4 * BM, std::true_type{}, std::false_type{}, std::true_type{});
                    else store_rows(i2, acc2, Ep + 4 * BM, std::false_type{}, std::false_type{}, std::true_type{});
                }
                if constexpr (i2 + 1 < TM) __syncthreads();
            });
        } else {
        // 3) W1 (packed [row block][ci][32 rows]) -> LDS; the main loop's last barrier freed the tile buffers
        {
            const f32x4* w2 = reinterpret_cast<const f32x4*>(p.w2);
            f32x4* dst = reinterpret_cast<f32x4*>(smem);
#pragma unroll 4
            for (int idx = tid; idx < BM * BM / 4; idx += NT) dst[idx] = w2[idx];
        }
        __syncthreads();
        // 4) y = W1 . h + b1 + x, one 32-row block at a time (ascending ci chain, like the stand-alone 1x1 kernel).  The skip
        //    operand is read before the first store where the registers allow it (all row blocks up front), else per row block.
        constexpr bool RES_UPFRONT = TM * TN <= 6;
        float rvall[RES_UPFRONT ? TM : 1][16][TN];
        if constexpr (RES_UPFRONT) {
            if (tile_full) {
#pragma unroll
                for (int i = 0; i < TM; ++i) load_res(i, rvall[i], p.res, std::true_type{});
            }
        }
        nc_static_for<TM>([&](auto i2t) __attribute__((always_inline)) {
            constexpr int i2 = decltype(i2t)::value;
            f32x16 acc2[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[j][r] = 0.0f;
            const float* W1s = smem + i2 * BM * 32 + hi * 32 + l31;
            nc_static_for<BM / 2>([&](auto kpt) __attribute__((always_inline)) {
                constexpr int kp = decltype(kpt)::value;
                constexpr int i = kp / 16, m = kp % 16;
                constexpr int reg = 4 * (m >> 2) + 2 * (m & 1) + ((m >> 1) & 1);
                const float a = W1s[2 * kp * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) acc2[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, acc[i][j][reg], acc2[j], 0, 0, 0);
            });
            const float nz0[TN] = {};
            if (!tile_full) {
                emit_rows_quad(i2, acc2, Ep + 3 * BM, Ep + 4 * BM, p.alpha_out2 != nullptr, p.res);
            } else {
                if constexpr (RES_UPFRONT) {
                    add_rows(i2, acc2, rvall[i2], nz0, Ep + 3 * BM, std::true_type{}, std::false_type{});
                } else {
                    float rv[16][TN];
                    load_res(i2, rv, p.res, std::true_type{});
                    add_rows(i2, acc2, rv, nz0, Ep + 3 * BM, std::true_type{}, std::false_type{});
                }
                if (p.alpha_out2 != nullptr) store_rows(i2, acc2, Ep + 4 * BM, std::true_type{}, std::false_type{}, std::true_type{});
                else store_rows(i2, acc2, Ep + 4 * BM, std::false_type{}, std::false_type{}, std::true_type{});
            }
        });
        }
    }
}

typedef void (*conv_kernel_fn)(const ConvArgs);

template <int TM, int TN, int K, int CB, int NX, bool FUSE = false, int OCC = 2, int NW = 4, int NP = 0, bool DIST = false, int SUB = 0,
          bool IN2 = false, bool XVK = false, bool DUO = false>
inline conv_kernel_fn get_conv_kernel() {
    return &conv_mfma_kernel<TM, TN, K, CB, NX, FUSE, OCC, NW, NP, DIST, SUB, IN2, XVK, DUO>;
}

}  // namespace nc

// Instantiation helper: one translation unit per K registers its 8 (TM,TN) variants.
#define NC_INSTANTIATE_CONV_K(KVAL, CBVAL, NXVAL)                                                          \
    namespace nc {                                                                                         \
    conv_kernel_fn conv_kernel_table_k##KVAL(int TM, int TN) {                                             \
        switch (TM * 10 + TN) {                                                                            \
            case 11: return get_conv_kernel<1, 1, KVAL, CBVAL, NXVAL>();                                   \
            case 12: return get_conv_kernel<1, 2, KVAL, CBVAL, NXVAL>();                                   \
            case 21: return get_conv_kernel<2, 1, KVAL, CBVAL, NXVAL>();                                   \
            case 22: return get_conv_kernel<2, 2, KVAL, CBVAL, NXVAL>();                                   \
            case 31: return get_conv_kernel<3, 1, KVAL, CBVAL, NXVAL>();                                   \
            case 32: return get_conv_kernel<3, 2, KVAL, CBVAL, NXVAL>();                                   \
            case 41: return get_conv_kernel<4, 1, KVAL, CBVAL, NXVAL>();                                   \
            case 42: return get_conv_kernel<4, 2, KVAL, CBVAL, NXVAL>();                                   \
        }                                                                                                  \
        return nullptr;                                                                                    \
    }                                                                                                      \
    int conv_kernel_cb_k##KVAL() { return CBVAL; }                                                         \
    int conv_kernel_nx_k##KVAL() { return NXVAL; }                                                         \
    }

// Two-input variants of the Encodec input mode (IN2): the strided down-convolutions and the up-convolutions that consume the sum of a
// residual block's shortcut and branch.  SUBV selects the sub-pixel form.
#define NC_INSTANTIATE_CONV_IN2(NAME, KVAL, CBVAL, NXVAL, SUBV)                                            \
    namespace nc {                                                                                         \
    conv_kernel_fn conv_kernel_table_##NAME(int TM, int TN) {                                              \
        switch (TM * 10 + TN) {                                                                            \
            case 11: return get_conv_kernel<1, 1, KVAL, CBVAL, NXVAL, false, 2, 4, 0, false, SUBV, true>(); \
            case 12: return get_conv_kernel<1, 2, KVAL, CBVAL, NXVAL, false, 2, 4, 0, false, SUBV, true>(); \
            case 21: return get_conv_kernel<2, 1, KVAL, CBVAL, NXVAL, false, 2, 4, 0, false, SUBV, true>(); \
            case 22: return get_conv_kernel<2, 2, KVAL, CBVAL, NXVAL, false, 2, 4, 0, false, SUBV, true>(); \
        }                                                                                                  \
        return nullptr;                                                                                    \
    }                                                                                                      \
    }

// Sub-pixel transposed-convolution variants (stride 2 / 4 / 8 up-convolutions: rows = (channel, phase) pairs).
#define NC_INSTANTIATE_CONV_SUB(KVAL, CBVAL, NXVAL)                                                        \
    namespace nc {                                                                                         \
    conv_kernel_fn conv_kernel_table_sub_k##KVAL(int TM, int TN) {                                         \
        switch (TM * 10 + TN) {                                                                            \
            case 11: return get_conv_kernel<1, 1, KVAL, CBVAL, NXVAL, false, 2, 4, 0, false, true>();      \
            case 12: return get_conv_kernel<1, 2, KVAL, CBVAL, NXVAL, false, 2, 4, 0, false, true>();      \
            case 21: return get_conv_kernel<2, 1, KVAL, CBVAL, NXVAL, false, 2, 4, 0, false, true>();      \
            case 22: return get_conv_kernel<2, 2, KVAL, CBVAL, NXVAL, false, 2, 4, 0, false, true>();      \
            case 31: return get_conv_kernel<3, 1, KVAL, CBVAL, NXVAL, false, 2, 4, 0, false, true>();      \
            case 32: return get_conv_kernel<3, 2, KVAL, CBVAL, NXVAL, false, 2, 4, 0, false, true>();      \
            case 41: return get_conv_kernel<4, 1, KVAL, CBVAL, NXVAL, false, 2, 4, 0, false, true>();      \
            case 42: return get_conv_kernel<4, 2, KVAL, CBVAL, NXVAL, false, 2, 4, 0, false, true>();      \
        }                                                                                                  \
        return nullptr;                                                                                    \
    }                                                                                                      \
    }

#define NC_INSTANTIATE_CONV_SUB_NARROW(KVAL, CBVAL, NXVAL)                                                 \
    namespace nc {                                                                                         \
    conv_kernel_fn conv_kernel_table_sub_narrow_k##KVAL(int TM) {                                          \
        switch (TM) {                                                                                      \
            case 1: return get_conv_kernel<1, 1, KVAL, CBVAL, NXVAL, false, 2, 3, 0, false, true>();       \
            case 2: return get_conv_kernel<2, 1, KVAL, CBVAL, NXVAL, false, 2, 3, 0, false, true>();       \
            case 3: return get_conv_kernel<3, 1, KVAL, CBVAL, NXVAL, false, 2, 3, 0, false, true>();       \
            case 4: return get_conv_kernel<4, 1, KVAL, CBVAL, NXVAL, false, 2, 3, 0, false, true>();       \
        }                                                                                                  \
        return nullptr;                                                                                    \
    }                                                                                                      \
    }

// Fused residual-unit variants (k=7 conv + Snake + 1x1 conv + skip in one launch): Cin == Cout == 32*TM.
#define NC_INSTANTIATE_CONV_FUSED(KVAL, CBVAL, NXVAL)                                                      \
    namespace nc {                                                                                         \
    conv_kernel_fn conv_kernel_table_fused_k##KVAL(int TM, int TN) {                                       \
        switch (TM * 10 + TN) {                                                                            \
            case 21: return get_conv_kernel<2, 1, KVAL, CBVAL, NXVAL, true>();                             \
            case 22: return get_conv_kernel<2, 2, KVAL, CBVAL, NXVAL, true>();                             \
            case 31: return get_conv_kernel<3, 1, KVAL, CBVAL, NXVAL, true>();                             \
            case 32: return get_conv_kernel<3, 2, KVAL, CBVAL, NXVAL, true>();                             \
            case 41: return get_conv_kernel<4, 1, KVAL, CBVAL, NXVAL, true>();                             \
            case 42: return get_conv_kernel<4, 2, KVAL, CBVAL, NXVAL, true>();                             \
        }                                                                                                  \
        return nullptr;                                                                                    \
    }                                                                                                      \
    }

// Wide fused residual units (C = 192 / 256): the tile spans all channels at 128 columns (TM = 6 / 8, TN = 1), reduction block of 4
// channels so two workgroups share a CU; W1 streamed through LDS by row block.
#define NC_INSTANTIATE_CONV_FUSED_WIDE(KVAL, CBVAL, NXVAL)                                                 \
    namespace nc {                                                                                         \
    conv_kernel_fn conv_kernel_table_fusedw_k##KVAL(int TM, int TN) {                                      \
        switch (TM * 10 + TN) {                                                                            \
            case 61: return get_conv_kernel<6, 1, KVAL, CBVAL, NXVAL, true>();                             \
            case 81: return get_conv_kernel<8, 1, KVAL, CBVAL, NXVAL, true>();                             \
        }                                                                                                  \
        return nullptr;                                                                                    \
    }                                                                                                      \
    }

// Light variants: a shorter reduction block (smaller LDS tiles, fewer staging registers) compiled for 3 waves per SIMD, for
// launches whose block count fits the chip in one round only at 3 workgroups per CU.  Same packed weight image as the
// standard variant when its CB is a multiple of this one's (kk = ci*K + k is contiguous across reduction blocks).
#define NC_INSTANTIATE_CONV_LIGHT(KVAL, CBVAL, NXVAL)                                                      \
    namespace nc {                                                                                         \
    conv_kernel_fn conv_kernel_table_light_k##KVAL(int TM, int TN) {                                       \
        switch (TM * 10 + TN) {                                                                            \
            case 22: return get_conv_kernel<2, 2, KVAL, CBVAL, NXVAL, false, 3>();                         \
            case 32: return get_conv_kernel<3, 2, KVAL, CBVAL, NXVAL, false, 3>();                         \
        }                                                                                                  \
        return nullptr;                                                                                    \
    }                                                                                                      \
    }

// Slim variants: half-size reduction block (half the LDS per workgroup) compiled for 4 waves per SIMD, for the narrow long-T layers
// (Cout <= 64) whose workgroups are bound by their memory round trips: more workgroups per CU overlap them.  Same packed weight image.
#define NC_INSTANTIATE_CONV_SLIM(KVAL, CBVAL, NXVAL) NC_INSTANTIATE_CONV_SLIM_N(slim, KVAL, CBVAL, NXVAL)
#define NC_INSTANTIATE_CONV_SLIM_N(NAME, KVAL, CBVAL, NXVAL)                                               \
    namespace nc {                                                                                         \
    conv_kernel_fn conv_kernel_table_##NAME##_k##KVAL(int TM, int TN) {                                        \
        switch (TM * 10 + TN) {                                                                            \
            case 11: return get_conv_kernel<1, 1, KVAL, CBVAL, NXVAL, false, 4>();                         \
            case 12: return get_conv_kernel<1, 2, KVAL, CBVAL, NXVAL, false, 4>();                         \
            case 21: return get_conv_kernel<2, 1, KVAL, CBVAL, NXVAL, false, 4>();                         \
            case 22: return get_conv_kernel<2, 2, KVAL, CBVAL, NXVAL, false, 4>();                         \
        }                                                                                                  \
        return nullptr;                                                                                    \
    }                                                                                                      \
    }

// Wide variants: 8 waves share one weight tile (BN = 512 columns), halving the weight traffic and staging per MFMA for the
// long-clip layers; one workgroup per CU.
#define NC_INSTANTIATE_CONV_WIDE(KVAL, CBVAL, NXVAL)                                                       \
    namespace nc {                                                                                         \
    conv_kernel_fn conv_kernel_table_wide_k##KVAL(int TM, int TN) {                                        \
        switch (TM * 10 + TN) {                                                                            \
            case 22: return get_conv_kernel<2, 2, KVAL, CBVAL, NXVAL, false, 2, 8>();                      \
            case 32: return get_conv_kernel<3, 2, KVAL, CBVAL, NXVAL, false, 2, 8>();                      \
            case 42: return get_conv_kernel<4, 2, KVAL, CBVAL, NXVAL, false, 2, 8>();                      \
        }                                                                                                  \
        return nullptr;                                                                                    \
    }                                                                                                      \
    }

// Narrow variants: 3 waves / 96 columns per workgroup (TN = 1), for the deep layers whose rows hold ~90 frames: a 128-column
// tile would idle a quarter of its matrix-core work on padding.
#define NC_INSTANTIATE_CONV_NARROW(KVAL, CBVAL, NXVAL)                                                     \
    namespace nc {                                                                                         \
    conv_kernel_fn conv_kernel_table_narrow_k##KVAL(int TM) {                                              \
        switch (TM) {                                                                                      \
            case 1: return get_conv_kernel<1, 1, KVAL, CBVAL, NXVAL, false, 2, 3>();                       \
            case 2: return get_conv_kernel<2, 1, KVAL, CBVAL, NXVAL, false, 2, 3>();                       \
            case 3: return get_conv_kernel<3, 1, KVAL, CBVAL, NXVAL, false, 2, 3>();                       \
            case 4: return get_conv_kernel<4, 1, KVAL, CBVAL, NXVAL, false, 2, 3>();                       \
        }                                                                                                  \
        return nullptr;                                                                                    \
    }                                                                                                      \
    }

// Wave-specialised variants: 4 consumer + 2 producer waves (384 threads), 3 waves per SIMD with two workgroups per CU.
#define NC_INSTANTIATE_CONV_SPEC(KVAL, CBVAL, NXVAL)                                                       \
    namespace nc {                                                                                         \
    conv_kernel_fn conv_kernel_table_spec_k##KVAL(int TM, int TN) {                                        \
        switch (TM * 10 + TN) {                                                                            \
            case 22: return get_conv_kernel<2, 2, KVAL, CBVAL, NXVAL, false, 3, 4, 2>();                   \
            case 32: return get_conv_kernel<3, 2, KVAL, CBVAL, NXVAL, false, 3, 4, 2>();                   \
        }                                                                                                  \
        return nullptr;                                                                                    \
    }                                                                                                      \
    }

// Distributed-staging variants (see DIST above).
#define NC_INSTANTIATE_CONV_DIST(KVAL, CBVAL, NXVAL)                                                       \
    namespace nc {                                                                                         \
    conv_kernel_fn conv_kernel_table_dist_k##KVAL(int TM, int TN) {                                        \
        switch (TM * 10 + TN) {                                                                            \
            case 22: return get_conv_kernel<2, 2, KVAL, CBVAL, NXVAL, false, 2, 4, 0, true>();             \
            case 32: return get_conv_kernel<3, 2, KVAL, CBVAL, NXVAL, false, 2, 4, 0, true>();             \
            case 42: return get_conv_kernel<4, 2, KVAL, CBVAL, NXVAL, false, 2, 4, 0, true>();             \
        }                                                                                                  \
        return nullptr;                                                                                    \
    }                                                                                                      \
    }

// Distributed staging for the one-workgroup-per-CU grids of the deep strided / sub-pixel layers (Encodec at 150 frames): with no
// co-resident partner to feed the matrix pipe during a workgroup's staging runs, the runs are dealt into the matrix-core shadows.
#define NC_INSTANTIATE_CONV_DIST_SMALL(NAME, KVAL, CBVAL, NXVAL, SUBV)                                     \
    namespace nc {                                                                                         \
    conv_kernel_fn conv_kernel_table_##NAME(int TM, int TN) {                                              \
        switch (TM * 10 + TN) {                                                                            \
            case 11: return get_conv_kernel<1, 1, KVAL, CBVAL, NXVAL, false, 2, 4, 0, true, SUBV>();       \
            case 12: return get_conv_kernel<1, 2, KVAL, CBVAL, NXVAL, false, 2, 4, 0, true, SUBV>();       \
            case 21: return get_conv_kernel<2, 1, KVAL, CBVAL, NXVAL, false, 2, 4, 0, true, SUBV>();       \
            case 22: return get_conv_kernel<2, 2, KVAL, CBVAL, NXVAL, false, 2, 4, 0, true, SUBV>();       \
            case 41: return get_conv_kernel<4, 1, KVAL, CBVAL, NXVAL, false, 2, 4, 0, true, SUBV>();       \
        }                                                                                                  \
        return nullptr;                                                                                    \
    }                                                                                                      \
    }

// Sub-pixel transposed convolution for ANY stride with two taps per phase (SUB == 2: multiply-shift row -> (channel, phase) map).
#define NC_INSTANTIATE_CONV_SUBG(KVAL, CBVAL, NXVAL)                                                       \
    namespace nc {                                                                                         \
    conv_kernel_fn conv_kernel_table_subg_k##KVAL(int TM, int TN) {                                        \
        switch (TM * 10 + TN) {                                                                            \
            case 11: return get_conv_kernel<1, 1, KVAL, CBVAL, NXVAL, false, 2, 4, 0, false, 2>();         \
            case 12: return get_conv_kernel<1, 2, KVAL, CBVAL, NXVAL, false, 2, 4, 0, false, 2>();         \
            case 21: return get_conv_kernel<2, 1, KVAL, CBVAL, NXVAL, false, 2, 4, 0, false, 2>();         \
            case 22: return get_conv_kernel<2, 2, KVAL, CBVAL, NXVAL, false, 2, 4, 0, false, 2>();         \
            case 31: return get_conv_kernel<3, 1, KVAL, CBVAL, NXVAL, false, 2, 4, 0, false, 2>();         \
            case 32: return get_conv_kernel<3, 2, KVAL, CBVAL, NXVAL, false, 2, 4, 0, false, 2>();         \
            case 41: return get_conv_kernel<4, 1, KVAL, CBVAL, NXVAL, false, 2, 4, 0, false, 2>();         \
            case 42: return get_conv_kernel<4, 2, KVAL, CBVAL, NXVAL, false, 2, 4, 0, false, 2>();         \
        }                                                                                                  \
        return nullptr;                                                                                    \
    }                                                                                                      \
    }

// XV-only instances (see XVK above): NAME_k<K>(TM) for the 256-column tiles; FUSEV selects the fused residual unit, SUBV the sub-pixel form.
#define NC_INSTANTIATE_CONV_XV(NAME, KVAL, CBVAL, NXVAL, FUSEV, SUBV)                                       \
    namespace nc {                                                                                         \
    conv_kernel_fn conv_kernel_table_##NAME(int TM) {                                                      \
        switch (TM) {                                                                                      \
            case 2: return get_conv_kernel<2, 2, KVAL, CBVAL, NXVAL, FUSEV, 2, 4, 0, false, SUBV, false, true>(); \
            case 3: return get_conv_kernel<3, 2, KVAL, CBVAL, NXVAL, FUSEV, 2, 4, 0, false, SUBV, false, true>(); \
            case 4: return get_conv_kernel<4, 2, KVAL, CBVAL, NXVAL, FUSEV, 2, 4, 0, false, SUBV, false, true>(); \
        }                                                                                                  \
        return nullptr;                                                                                    \
    }                                                                                                      \
    }

// DUO instances (see DUO above): two tiles per workgroup of 8 wavefronts, the second half a block behind the first.
#define NC_INSTANTIATE_CONV_DUO(NAME, KVAL, CBVAL, NXVAL, SUBV)                                            \
    namespace nc {                                                                                         \
    conv_kernel_fn conv_kernel_table_##NAME(int TM) {                                                      \
        switch (TM) {                                                                                      \
            case 2: return get_conv_kernel<2, 2, KVAL, CBVAL, NXVAL, false, 2, 4, 0, false, SUBV, false, true, true>(); \
            case 3: return get_conv_kernel<3, 2, KVAL, CBVAL, NXVAL, false, 2, 4, 0, false, SUBV, false, true, true>(); \
        }                                                                                                  \
        return nullptr;                                                                                    \
    }                                                                                                      \
    }
