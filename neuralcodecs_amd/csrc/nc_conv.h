// Implicit-GEMM 1-D convolution on the fp32 matrix cores of gfx950 -- host-side descriptors.
//
// One kernel template covers every dense convolution of the DAC / SNAC / Encodec hot path:
//   conv1d (any K, stride, dilation, zero "same" padding)   WNConv1d.cs:152, EncoderBlock.cs:27-33
//   conv_transpose1d as `stride` polyphase sub-convolutions  WNConvTranspose1d.cs:152
//   1x1 projections of the quantizer                         VectorQuantizer.cs:47-48
// with the element-wise neighbours fused: Snake on the input tile (Snake1d.cs:52), bias,
// residual add (ResidualUnit.cs:58), Snake for the next layer, tanh (Decoder.cs:46) and the
// RVQ accumulate/subtract (ResidualVectorQuantizer.cs:68-69).
//
// GEMM view:  M = Cout (A = weights, pre-packed per tile),  N = output time steps of one clip,
//             Kdim = Cin*K flattened as kk = ci*K + k  -- the canonical accumulation order.
// Each output element is ONE chain of v_mfma_f32_32x32x2_f32 updates in ascending kk, which is
// bit-identical to a scalar fmaf loop (MI355X_MICROARCH.md, "FP32-input MFMA").
#pragma once
#include <memory>

#include "nc_common.h"

namespace nc {

enum : int { EPI_TANH = 1, EPI_RVQ = 2, EPI_NOISE = 4,
             EPI_NO_XR = 1 << 20,   // NC_NO_XR=1 (a test switch riding in the epilogue flags: a field of its own shifted the argument block and
                                    // cost the k = 7 loop 51 scalar-register reloads)
             EPI_XVEC = 1 << 21 };  // vectorised window staging granted by the host (two-tap sub-pixel instances: the kernel's XV note)

struct ConvArgs {
    // input activations [B][Cin][x_len] (row stride x_cstride); positions outside [0,x_len) read as 0
    const float* x;
    int64_t x_bstride, x_cstride;
    int32_t Cin, x_len;
    const float* alpha_in;  // nullable: Snake applied while staging the input tile
    // packed weights: [phase][co_tile][ci_block][CB*K][BM]
    const float* w;
    int64_t w_phase_stride;
    const float* bias;       // nullable [Cout]
    const float* alpha_out;  // nullable [Cout]: Snake of the NEXT layer fused into the store
    const float* res;        // nullable residual, same geometry as y
    float* y;
    int64_t y_bstride, y_cstride;  // elements
    int32_t y_tstride, y_toff;     // t_out = col*y_tstride + y_toff + phase
    int32_t Tout;                  // stores outside [0,Tout) are dropped
    const float* noise;            // EPI_NOISE: y = res + noise[b*noise_bstride + t] * conv   (NoiseBlock.cs:38-45)
    int64_t noise_bstride;
    const float* w2;               // fused residual unit: packed 1x1 weights [row block][ci][32], bias2 [Cout]
    const float* bias2;
    const float* alpha_out2;       // fused residual unit: Snake of the layer consuming y (nullable)
    float* rvq_zq;                 // EPI_RVQ: zq += out ; rvq_res -= out  (same geometry as y)
    float* rvq_res;
    int32_t Cout, n_cols;          // columns (output steps per phase) handled by this launch
    int32_t stride, dil, pad;      // x position of (col,k) = col*stride + k*dil - pad
    int32_t B, n_co_tiles, n_t_tiles, n_cb, n_phase;
    int32_t xw, xwp, xrow, xneg;   // staged input window geometry (see kernel)
    int32_t xbuf;                  // floats per window buffer (CB*xrow rounded up to 4)
    int32_t nchunk, n_items;       // 64-slot chunks per channel row; CB*nchunk wave-level staging items
    int32_t chunk_magic;           // item / nchunk  == (item * chunk_magic) >> 20   (host-verified)
    int32_t stride_magic;          // j / stride     == (j * stride_magic) >> 20     (host-verified)
    int32_t tapoff[16];            // window slot of tap k at column 0 (phase de-interleave folded in)
    int32_t epi;
    int32_t ep_off;                // float offset in LDS of the per-row epilogue table (bias, Snake alpha, 1/alpha [, fused set])
    // Encodec input mode (SConv1d.cs:144-173 fused into the tile load): the input is a RAW conv output whose GroupNorm(1,C) is still
    // pending; the staged value is  pad(elu?(gn?(x)))  with the asymmetric reflect pad of SConv1d.Pad1d evaluated as an index map.
    int32_t in_mode;               // 0 off; bit 0: GroupNorm affine ((x-mu)*r)*gamma+beta; bit 1: ELU; bit 2: reflect addressing
    const float* in_stats;         // [B][2] (mean, rstd) per sample (bit 0)
    const float* in_gamma;         // [Cin]
    const float* in_beta;          // [Cin]
    int32_t sub_shift;             // SUB kernels: log2(stride) of the sub-pixel transposed convolution (rows = co*stride + phase; Cout = rows)
    int32_t sub_stride, sub_magic, sub_cout;   // SUB == 2 kernels (any stride): row x is channel (x * sub_magic) >> 20 = x / sub_stride of sub_cout, phase x % sub_stride
    // flattened (clip, column) axis (see the kernel's tile map).  flat = 0: one clip per tile (B clips in the tile map, both pitches
    // 0x1fffffff).  flat = 1: B = 1 in the tile map, n_t_tiles covers Bc*n_cols columns, flat_pc = n_cols, flat_hc = halo columns per
    // segment, flat_px = (n_cols + flat_hc) * stride window slots per clip.  Bc = number of clips (both modes).
    int32_t flat, flat_px, flat_pc, flat_hc, Bc;
    int32_t co_group;              // row tiles per group of the block -> tile order (see the kernel's tile map); >= 1, divides n_co_tiles
    int32_t in_left, in_Lz, in_L;  // bit 2: padded position j reads q = reflect(j - left) over [0,Lz); samples q >= L are the zero extension (D9)
    // two-input form (in_mode bit 3, IN2 kernels): second operand with the geometry of x and its own pending GroupNorm
    const float* x2;
    const float* in_stats2;
    const float* in_gamma2;
    const float* in_beta2;
    // GroupNorm(1,C) statistics of the OUTPUT emitted from the epilogue (nc_gn.h): (S1, S2) of every 32x32 block of the tile, binary64,
    // at gn_part[((clip*gn_nrb + row block)*gn_ncb + column block)*2]; null = off.  Plain epilogues only (bias, no residual / activation).
    double* gn_part;
    int32_t gn_nrb, gn_ncb;
    // finishing inside the launch (nc_gn.h nc_gn_arrive_and_finish): per-sample arrival counters (self-resetting), the (mean, rstd)
    // output [B][2] and the element count C*T; gn_count null = the caller runs gn_final_kernel afterwards
    unsigned* gn_count;
    float* gn_stats;
    double gn_n;
};

// ConvIO::gn_n as the kernels take it: the element count of a sample, NEGATED when NC_SYNC_ACQUIRE=1 asks the in-launch GroupNorm finish
// (nc_gn.h) for an acquire fence in front of its reads.
inline double gn_count_arg(double n) {
    static const bool acquire = env_flag("NC_SYNC_ACQUIRE");
    return acquire ? -n : n;
}

// Position of row (32*i + r) of a weight tile inside one kk row of BM = 32*TM floats.  The TM values of one matrix-core lane
// (rows r, 32+r, ...) sit next to each other, so a lane fetches its A fragments with one wide LDS read (b64 / b128; TM = 3:
// b64 + b32) whose address is lane base + immediate.
__host__ __device__ constexpr int a_tile_pos(int TM, int i, int r) {
    // (TM = 6 / 8, the whole-channel tiles of the wide fused residual units: two TM = 3 / 4 images side by side)
    return TM == 6 ? (i / 3) * 96 + (i % 3 < 2 ? 2 * r + i % 3 : 64 + r) : TM == 8 ? (i / 4) * 128 + 4 * r + i % 4 :
           TM == 1 ? r : TM == 2 ? 2 * r + i : TM == 4 ? 4 * r + i : (i < 2 ? 2 * r + i : 64 + r);
}

struct TileCfg {
    int TM, TN, K, CB;
    int NW = 4;                       // waves per workgroup (8: wide variants)
    int BM() const { return 32 * TM; }
    int BN() const { return 32 * NW * TN; }
    int KB() const { return CB * K; }
};

// A convolution layer resident on the device: folded + packed weights and launch geometry.
struct ConvLayer {
    int Cin = 0, Cout = 0, K = 0, stride = 1, pad = 0, dil = 1, out_pad = 0;
    bool transposed = false;
    TileCfg cfg{};
    int n_phase = 1, Ktaps = 0;  // taps per phase (== K for conv, ceil(K/stride) for conv-transpose)
    int sub_stride = 0;          // > 0: transposed conv packed in sub-pixel form (rows = (channel, phase) pairs, one launch, n_phase == 1)
    int sub_shift = 0;           // log2(sub_stride) when that is a power of two (the shift / mask kernels), else 0 (the multiply-shift kernels)
    int rows() const { return sub_stride ? Cout * sub_stride : Cout; }   // GEMM rows of the packed image
    DevBuf w, bias;
    // additional row-tile heights (32*TM dividing Cout) packed at load; the launch picks the one that fills the chip best
    struct Alt {
        TileCfg cfg{};
        DevBuf w;
        int64_t w_phase_stride = 0;
    };
    std::vector<std::unique_ptr<Alt>> alts;
    DevBuf w_thin;   // Cout <= 2, stride 1: dense [Cout][Cin][K] image for the streaming thin-output kernel (nc_conv_thin.hip)
    DevBuf w_stem;   // Cin == 1, K == 7, stride 1: dense [Cout][K] image for the streaming stem kernel (nc_conv_thin.hip)
    DevBuf w_small;  // strided, K >= 4, Cin*K % 16 == 0: A-fragment image of conv_small_kernel (nc_conv_small.hip: short rows of few-clip batches)
    DevBuf w_skinny; // K==1, Cout<=16, Cin%64==0: [Cin/4][64 lanes] A-fragment image of skinny_proj_kernel (rows >= Cout zero)
    DevBuf w_fused;  // K==1, Cin==Cout<=128: [row block][ci][32 rows] image consumed by the fused residual-unit kernel
    bool has_bias = false;
    int64_t w_phase_stride = 0;
    int kclass = NC_KC_CONV_MISC;
    // host: dense folded weight in the reference's layout ([Cout,Cin,K] or [Cin,Cout,K])
    void build(const float* dense_w, const float* bias_h, int Cin, int Cout, int K, int stride, int pad, int dil, int out_pad,
               bool transposed);
    void release_all() {   // op-level hooks build throw-away layers
        w.release(); bias.release(); w_skinny.release(); w_fused.release(); w_thin.release(); w_stem.release(); w_small.release();
        for (auto& a : alts) a->w.release();
        alts.clear();
    }
    int64_t out_len(int64_t Tin) const;
    double flops(int B, int64_t Tin) const;
};

struct ConvIO {
    const float* x = nullptr;
    int64_t x_bstride = 0, x_cstride = 0;
    int32_t x_len = 0;     // valid samples
    int64_t Tin = 0;       // logical input length (>= x_len; the tail is zero: DAC.Preprocess right-pad)
    const float* alpha_in = nullptr;
    const float* alpha_out = nullptr;
    const float* res = nullptr;
    float* y = nullptr;
    int64_t y_bstride = 0, y_cstride = 0;
    float* rvq_zq = nullptr;
    float* rvq_res = nullptr;
    const float* noise = nullptr;  // EPI_NOISE multiplier [B,1,Tout]
    int epi = 0;
    // Encodec input mode (see ConvArgs::in_mode): with in_L > 0 the tensor handed in is the UN-padded row of in_L samples and
    // Tin / x_len describe the padded row the convolution runs over
    const float* in_stats = nullptr;
    const float* in_gamma = nullptr;
    const float* in_beta = nullptr;
    bool in_elu = false;
    int64_t in_left = 0, in_Lz = 0, in_L = 0;
    const float* x2 = nullptr;                  // Encodec two-input mode: second operand (same strides / lengths as x), added after its own GroupNorm
    const float* in_stats2 = nullptr;
    const float* in_gamma2 = nullptr;
    const float* in_beta2 = nullptr;
    double* gn_part = nullptr;                  // GroupNorm block sums of the output ([B][gn_nrb][gn_ncb][2], see ConvArgs::gn_part); needs conv_gn_fusable()
    int gn_nrb = 0, gn_ncb = 0;
    unsigned* gn_count = nullptr;               // with gn_part: finish the statistics inside the launch (per-sample counters, zero between launches)
    float* gn_stats = nullptr;                  // [B][2] (mean, rstd)
    double gn_n = 0.0;                          // elements per sample (C * T of the normalised tensor)
    const float* alpha_out2 = nullptr;          // with fuse_k1: Snake applied to the unit's output y (consumer's activation)
    const struct ConvLayer* fuse_k1 = nullptr;  // fused residual unit: the 1x1 layer applied to snake(alpha_out)(this conv) + res
};

// arguments of the thin-output convolution in the Encodec input mode (nc_conv_thin.hip conv_thin_inm_kernel)
struct ThinInmArgs {
    const float* xa;
    const float* xb2;            // nullable second operand (same geometry)
    int64_t x_bstride, x_cstride;
    int Cin, L, left, Lz, Lp;    // rows of L samples; padded position j in [0, Lp) reads q = reflect(j - left) over [0, Lz); q >= L is zero
    const float* stats_a; const float* gamma_a; const float* beta_a;
    const float* stats_b; const float* gamma_b; const float* beta_b;
    int elu;
    const float* w;              // dense [COUT][Cin][7]
    const float* bias;
    float* y;
    int64_t y_bstride, y_cstride;
    int Tout, n_t_tiles;
    double* gn_part; int gn_ncb; unsigned* gn_count; float* gn_stats; double gn_n;
};
bool launch_conv_thin_inm(const ThinInmArgs& a, int B, int Cout, hipStream_t s);

// arguments of the fused first pass of a SEANetResnetBlock (nc_resa.hip res_a_kernel): shortcut 1x1 + k = 3 branch from one read of x
struct ResAArgs {
    const float* x;              // block input [B][Cin][T] view (row stride x_cstride) with its pending GroupNorm
    int64_t x_bstride, x_cstride;
    int Cin, T;
    const float* in_stats;       // nullable [B][2] (mean, rstd): GroupNorm(1,C) of x still pending
    const float* in_gamma; const float* in_beta;
    const float* w_s; const float* bias_s; float* ys; int64_t ys_bstride, ys_cstride; int Cs;   // shortcut: packed 1x1 image (one row tile), Cs = Cin rows
    const float* w_b; const float* bias_b; float* yb; int64_t yb_bstride, yb_cstride; int Cb;   // branch: packed k = 3 image (one row tile), Cb = Cin/2 rows
    int B, n_t_tiles, n_cb;
    // GroupNorm block sums of the two outputs (all null = off) with the in-launch finish on each output's own counters
    double* gn_part_s; double* gn_part_b; int gn_nrb_s, gn_nrb_b, gn_ncb;
    unsigned* gn_count_s; unsigned* gn_count_b; float* gn_stats_s; float* gn_stats_b; double gn_n_s, gn_n_b;
};
bool launch_res_a(const ResAArgs& a, int TMS, bool aligned, hipStream_t s);

// arguments of the streaming two-input k = 4, stride-2 down-convolution of the Encodec encoder (nc_down2.hip down2_kernel)
struct Down2Args {
    const float* xa; const float* xb;   // the two pending operands [B][Cin][T] (same strides): shortcut and branch of the residual block
    int64_t x_bstride, x_cstride;
    int Cin, T, Tout;
    const float* stats_a; const float* gamma_a; const float* beta_a;   // nullable together with the b set: GroupNorm(1,C) still pending
    const float* stats_b; const float* gamma_b; const float* beta_b;
    const float* w; const float* bias;  // packed image of the layer (K = 4, CB = 8, one row tile)
    float* y; int64_t y_bstride, y_cstride; int Cout;
    int B, n_t_tiles, n_cb;
    int n_co_tiles; int64_t w_co_stride;   // down5_kernel: row tiles of 128 and the float offset between their weight images (1 / 0 elsewhere)
    double* gn_part; int gn_nrb, gn_ncb; unsigned* gn_count; float* gn_stats; double gn_n;
};
bool launch_down2(const Down2Args& a, int TM, bool aligned, hipStream_t s);
bool launch_down4(const Down2Args& a, int TM, hipStream_t s);   // nc_down4.hip: k = 8, stride 4, 128 output rows, 16-byte aligned rows
bool launch_down5(const Down2Args& a, int TM, hipStream_t s);   // nc_down5.hip: k = 10, stride 5, 256 output rows as two row tiles

// arguments of the streaming two-input k = 4, stride-2 up-convolution of the Encodec decoder (nc_up2.hip up2_kernel)
struct Up2Args {
    const float* xa; const float* xb;   // the two pending operands [B][Cin][L] (same strides)
    int64_t x_bstride, x_cstride;
    int Cin, L, elu;
    const float* stats_a; const float* gamma_a; const float* beta_a;   // nullable together with the b set
    const float* stats_b; const float* gamma_b; const float* beta_b;
    const float* w; const float* bias;  // sub-pixel image of the layer (rows co*2 + phase, two taps, CB = 16, one row tile); bias [Cout]
    float* y; int64_t y_bstride, y_cstride; int Cout;   // UNTRIMMED output [B][Cout][S L + S]
    int B, n_t_tiles, n_cb, n_co_tiles;
    double* gn_part; int gn_nrb, gn_ncb; unsigned* gn_count; float* gn_stats; double gn_n;
};
bool launch_up2(const Up2Args& a, int TM, int S, bool aligned, hipStream_t s);   // S = stride: 2 (k = 4, TM = 2) or 4 (k = 8, TM = 4, two row tiles)

// short-row strided convolution on v_mfma_f32_16x16x4_f32 (nc_conv_small.hip): plain input, bias-only epilogue
bool conv_small_eligible(int Cin, int Cout, int K, int stride, int dil, bool transposed);
void pack_conv_small(const float* dense_w, int Cin, int Cout, int K, std::vector<float>& out);
int conv_small_max_tn(int Cin, int K, int stride, int dil);
struct ConvSmallGn { double* part; int nrb, ncb; unsigned* count; float* stats; double n; };   // GroupNorm sums from the short-row kernel's epilogue
bool conv_small_inm_available(int Cin, int K, int stride, int dil);
bool conv_small_gn_available(int Cin, int K, int stride, int dil, bool inm);
bool launch_conv_small(const float* x, int64_t x_bstride, int64_t x_cstride, int x_len, int in_left, int in_Lz, int in_L, const float* in_stats,
                       const float* in_gamma, const float* in_beta, int in_elu, const ConvSmallGn* gn, const float* wp, const float* bias, const float* alpha_out, float* y,
                       int64_t y_bstride, int64_t y_cstride, int B, int Cin, int Cout, int K, int stride, int pad, int dil, int Tout, int want_tn, hipStream_t s);

TileCfg pick_tile(int Cout, int Ktaps);
// true when `k7` followed by `k1` can run as one fused residual-unit launch
bool can_fuse_res_unit(const ConvLayer& k7, const ConvLayer& k1);  // TN is chosen per launch from the column count
void launch_conv(const ConvLayer& L, const ConvIO& io, int B, hipStream_t stream, Profiler* prof);
// true when launch_conv can emit the GroupNorm block sums of this launch's output from the kernel epilogue (ConvIO::gn_part); the
// streaming thin / stem / skinny kernels and the per-phase transposed launches cannot (the caller then runs the stand-alone pass)
bool conv_gn_fusable(const ConvLayer& L, const ConvIO& io, int B = -1);   // B: clips of the launch (the short-row kernel, which has no such epilogue, is chosen per launch)
// true when launch_conv has a two-input (ConvIO::x2) kernel for this layer
bool conv_in2_available(const ConvLayer& L);
// block view of a [C][T] GroupNorm input behind layer L (sub-pixel transposed convolutions: rows = (channel, phase) pairs)
inline int conv_gn_sub(int K, int stride, int Cout, bool transposed) {
    return (transposed && (stride == 2 || stride == 4 || stride == 8) && K == 2 * stride && (Cout * stride) % 32 == 0) ? stride : 1;
}

}  // namespace nc
