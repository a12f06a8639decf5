// Host side of the implicit-GEMM convolution: tile selection, weight packing, launch.
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <set>

#include "nc_conv.h"

namespace nc {

typedef void (*conv_kernel_fn)(const ConvArgs);
conv_kernel_fn conv_kernel_table_k1(int, int);
int conv_kernel_cb_k1();
int conv_kernel_nx_k1();
conv_kernel_fn conv_kernel_table_k2(int, int);
int conv_kernel_cb_k2();
int conv_kernel_nx_k2();
conv_kernel_fn conv_kernel_table_k3(int, int);
int conv_kernel_cb_k3();
int conv_kernel_nx_k3();
conv_kernel_fn conv_kernel_table_k4(int, int);
int conv_kernel_cb_k4();
int conv_kernel_nx_k4();
conv_kernel_fn conv_kernel_table_k6(int, int);
int conv_kernel_cb_k6();
int conv_kernel_nx_k6();
conv_kernel_fn conv_kernel_table_k7(int, int);
int conv_kernel_cb_k7();
int conv_kernel_nx_k7();
conv_kernel_fn conv_kernel_table_k8(int, int);
int conv_kernel_cb_k8();
int conv_kernel_nx_k8();
conv_kernel_fn conv_kernel_table_k10(int, int);
int conv_kernel_cb_k10();
int conv_kernel_nx_k10();
conv_kernel_fn conv_kernel_table_k16(int, int);
int conv_kernel_cb_k16();
int conv_kernel_nx_k16();

#define NC_K_CASES(X) X(1) X(2) X(3) X(4) X(6) X(7) X(8) X(10) X(16)

conv_kernel_fn conv_kernel_table_fused_k7(int, int);
// Measured-and-rejected k=7 variants (light / wide / wave-specialised / distributed staging): built only with
// `make EXPERIMENTS=1` (-DNC_EXPERIMENTS); the default library carries neither their code objects nor their switches.
#ifdef NC_EXPERIMENTS
conv_kernel_fn conv_kernel_table_light_k7(int, int);
conv_kernel_fn conv_kernel_table_wide_k7(int, int);
conv_kernel_fn conv_kernel_table_spec_k7(int, int);
conv_kernel_fn conv_kernel_table_dist_k7(int, int);
static int experiment_mode(const char* name) { return (int)env_int(name, 0); }
#else
static conv_kernel_fn conv_kernel_table_light_k7(int, int) { return nullptr; }
static conv_kernel_fn conv_kernel_table_wide_k7(int, int) { return nullptr; }
static conv_kernel_fn conv_kernel_table_spec_k7(int, int) { return nullptr; }
static conv_kernel_fn conv_kernel_table_dist_k7(int, int) { return nullptr; }
static int experiment_mode(const char*) { return 0; }
#endif
conv_kernel_fn conv_kernel_table_xv_k7(int);
conv_kernel_fn conv_kernel_table_xv_fused_k7(int);
#ifdef NC_EXPERIMENTS
conv_kernel_fn conv_kernel_table_duo_k7(int);   // (the DUO form of the XV-only k = 7 instances, NC_DUO=1: bit-exact, 6 % slower on the k = 7 class)
#else
static conv_kernel_fn conv_kernel_table_duo_k7(int) { return nullptr; }
#endif
conv_kernel_fn conv_kernel_table_xv_sub_k2(int);
conv_kernel_fn conv_kernel_table_xv_subg_k2(int);
conv_kernel_fn conv_kernel_table_sub_k2(int, int);
conv_kernel_fn conv_kernel_table_subg_k2(int, int);
conv_kernel_fn conv_kernel_table_dist_k16(int, int);
conv_kernel_fn conv_kernel_table_dist_sub_k2(int, int);
conv_kernel_fn conv_kernel_table_sub_narrow_k2(int);
conv_kernel_fn conv_kernel_table_fusedw_k7(int, int);
conv_kernel_fn conv_kernel_table_slim_k3(int, int);
conv_kernel_fn conv_kernel_table_slim_k7(int, int);
conv_kernel_fn conv_kernel_table_narrow_k2(int);
conv_kernel_fn conv_kernel_table_narrow_k3(int);
conv_kernel_fn conv_kernel_table_narrow_k7(int);
conv_kernel_fn conv_kernel_table_narrow_k16(int);
static conv_kernel_fn narrow_kernel(int K, int TM) {
    switch (K) {
        case 2: return conv_kernel_table_narrow_k2(TM);
        case 3: return conv_kernel_table_narrow_k3(TM);
        case 7: return conv_kernel_table_narrow_k7(TM);
        case 16: return conv_kernel_table_narrow_k16(TM);
    }
    return nullptr;
}
conv_kernel_fn conv_kernel_table_in2_k4(int, int);
conv_kernel_fn conv_kernel_table_in2_k8(int, int);
conv_kernel_fn conv_kernel_table_in2_k10(int, int);
conv_kernel_fn conv_kernel_table_in2_k16(int, int);
conv_kernel_fn conv_kernel_table_in2_k2(int, int);
conv_kernel_fn conv_kernel_table_in2_sub_k2(int, int);
// two-input (ConvIO::x2) kernel of a layer's tile shape; null: none instantiated
static conv_kernel_fn in2_kernel(int Ktaps, bool sub, int TM, int TN) {
    if (sub) return Ktaps == 2 ? conv_kernel_table_in2_sub_k2(TM, TN) : nullptr;
    switch (Ktaps) {
        case 2: return conv_kernel_table_in2_k2(TM, TN);
        case 4: return conv_kernel_table_in2_k4(TM, TN);
        case 8: return conv_kernel_table_in2_k8(TM, TN);
        case 10: return conv_kernel_table_in2_k10(TM, TN);
        case 16: return conv_kernel_table_in2_k16(TM, TN);
    }
    return nullptr;
}
conv_kernel_fn conv1x1_kernel_table(int, int);
#ifdef NC_EXPERIMENTS
conv_kernel_fn conv1x1_stream_kernel_table(int, int);   // (round 2's streaming pointwise variant: overtaken in round 4, EXPERIMENTS=1 builds only)
#else
static conv_kernel_fn conv1x1_stream_kernel_table(int, int) { return nullptr; }
#endif
bool launch_conv_thin(const float* x, int64_t x_bstride, int64_t x_cstride, int Cin, int x_len, const float* w_dense, const float* bias, float* y,
                      int64_t y_bstride, int64_t y_cstride, int B, int Cout, int K, int pad, int dil, int64_t Tout, bool tanh_out, hipStream_t s);
bool launch_conv_stem(const float* x, int64_t x_bstride, int x_len, const float* w_dense, const float* bias, const float* alpha_out, float* y,
                      int64_t y_bstride, int64_t y_cstride, int B, int Cout, int K, int pad, int64_t Tout, hipStream_t s);
void launch_skinny_proj(const float* x, int64_t x_bstride, int64_t x_cstride, const float* wp, const float* bias, float* y, int64_t y_bstride,
                        int64_t y_cstride, int B, int Cin, int Cout, int64_t T, hipStream_t s);

static int cb_for_k(int K) {
    switch (K) {
#define X(k) case k: return conv_kernel_cb_k##k();
        NC_K_CASES(X)
#undef X
    }
    fail(NC_EUNSUPPORTED, "convolution with %d taps per phase has no kernel instantiation", K);
}

static int nx_for_k(int K) {
    switch (K) {
#define X(k) case k: return conv_kernel_nx_k##k();
        NC_K_CASES(X)
#undef X
    }
    fail(NC_EUNSUPPORTED, "convolution with %d taps per phase has no kernel instantiation", K);
}

// exact small-range division by multiplication: n / d == (n * magic) >> 20 for all 0 <= n < limit
static int32_t magic_div(int d, int limit) {
    const int32_t m = (int32_t)(((1u << 20) + d - 1) / d);
    for (int n = 0; n < limit; ++n)
        if ((int)(((int64_t)n * m) >> 20) != n / d || (int64_t)n * m > 0x7fffffffLL)
            fail(NC_EUNSUPPORTED, "internal: no exact reciprocal for /%d below %d", d, limit);
    return m;
}

static conv_kernel_fn lookup_kernel(const TileCfg& c) {
    conv_kernel_fn f = nullptr;
    switch (c.K) {
#define X(k) case k: f = conv_kernel_table_k##k(c.TM, c.TN); break;
        NC_K_CASES(X)
#undef X
    }
    if (!f) fail(NC_EUNSUPPORTED, "no conv kernel for TM=%d TN=%d K=%d", c.TM, c.TN, c.K);
    return f;
}

TileCfg pick_tile(int Cout, int Ktaps) {
    TileCfg c{};
    int best = 1 << 30;
    for (int tm = 4; tm >= 1; --tm) {
        int bm = 32 * tm;
        int padded = (Cout + bm - 1) / bm * bm;
        if (padded < best) {
            best = padded;
            c.TM = tm;
        }
    }
    if (const int tm = (int)env_int("NC_TM_FORCE", 0)) {   // experiment: force the row-tile height where it divides Cout
        if (tm >= 1 && tm <= 4 && Cout % (32 * tm) == 0) c.TM = tm;
    }
    c.TN = 2;
    c.K = Ktaps;
    c.CB = cb_for_k(Ktaps);
    return c;
}

int64_t ConvLayer::out_len(int64_t Tin) const {
    if (transposed) return (Tin - 1) * stride - 2 * (int64_t)pad + K + out_pad;
    return (Tin + 2 * (int64_t)pad - (int64_t)dil * (K - 1) - 1) / stride + 1;
}

double ConvLayer::flops(int B, int64_t Tin) const {
    if (transposed) return 2.0 * Cin * Cout * K * (double)Tin * B;  // counted on L_in
    return 2.0 * Cin * Cout * K * (double)out_len(Tin) * B;
}

void ConvLayer::build(const float* dense_w, const float* bias_h, int Cin_, int Cout_, int K_, int stride_, int pad_, int dil_,
                      int out_pad_, bool transposed_) {
    Cin = Cin_; Cout = Cout_; K = K_; stride = stride_; pad = pad_; dil = dil_; out_pad = out_pad_; transposed = transposed_;
    sub_shift = 0;
    sub_stride = 0;
    if (transposed) {
        if (dil != 1) fail(NC_EUNSUPPORTED, "dilated conv_transpose1d is not on the hot path");
        n_phase = stride;
        Ktaps = (K + stride - 1) / stride;
        // sub-pixel form (one launch, rows = (channel, phase)): power-of-two strides with two taps per phase, i.e. the k = 2s
        // up-convolutions of DAC / SNAC (DecoderBlock.cs:27-33); other strides keep the per-phase launches
        static const bool no_sub = env_flag("NC_NO_SUBPIXEL");
        if (!no_sub && (stride == 2 || stride == 4 || stride == 8) && K == 2 * stride && (Cout * stride) % 32 == 0 && out_pad == 0) {
            sub_shift = stride == 2 ? 1 : stride == 4 ? 2 : 3;
            sub_stride = stride;
            n_phase = 1;
        }
        // ... and the same form for the other strides (SNAC's stride-3 and Encodec's stride-5 up-convolutions, k = 2s): one launch with
        // full row tiles instead of s launches that each write every s-th sample (NC_NO_SUBPIXEL_ANY=1: per-phase launches)
        static const bool no_sub_any = env_flag("NC_NO_SUBPIXEL_ANY");
        // (output_padding -- stride % 2 in SNAC's DecoderBlock -- only moves the right crop: the extra samples lie inside the (Tin + 1) * s
        // samples the rows cover as long as out_pad <= pad, and the store bounds come from out_len())
        if (!no_sub && !no_sub_any && !sub_stride && stride >= 3 && stride <= 16 && K == 2 * stride && (Cout * stride) % 32 == 0 && out_pad <= pad) {
            sub_stride = stride;
            n_phase = 1;
        }
    } else {
        if (stride > 1 && dil != 1) fail(NC_EUNSUPPORTED, "strided+dilated conv1d is not on the hot path");
        n_phase = 1;
        Ktaps = K;
    }
    cfg = pick_tile(rows(), Ktaps);
    auto pack = [&](const TileCfg& tc, DevBuf& dstbuf, int64_t& phase_stride) {
        const int BM = tc.BM(), CB = tc.CB, KB = tc.KB();
        const int n_co = (rows() + BM - 1) / BM;
        const int n_cb = (Cin + CB - 1) / CB;
        phase_stride = (int64_t)n_co * n_cb * KB * BM;
        std::vector<float> packed((size_t)phase_stride * n_phase, 0.0f);
        for (int ph = 0; ph < n_phase; ++ph)
            for (int ct = 0; ct < n_co; ++ct)
                for (int cb = 0; cb < n_cb; ++cb) {
                    float* dst = packed.data() + (size_t)ph * phase_stride + ((size_t)ct * n_cb + cb) * KB * BM;
                    for (int kk = 0; kk < KB; ++kk) {
                        const int ci = cb * CB + kk / Ktaps, k = kk % Ktaps;
                        if (ci >= Cin) continue;
                        for (int r = 0; r < BM; ++r) {
                            int co = ct * BM + r, php = ph;
                            if (co >= rows()) continue;
                            if (sub_stride) { php = co % stride; co /= stride; }   // row = co*stride + phase
                            float v;
                            if (transposed) {
                                const int kt = php + k * stride;  // tap of this phase, ascending (canonical order)
                                if (kt >= K) continue;
                                v = dense_w[((size_t)ci * Cout + co) * K + kt];
                            } else {
                                v = dense_w[((size_t)co * Cin + ci) * K + k];
                            }
                            dst[(size_t)kk * BM + a_tile_pos(tc.TM, r / 32, r % 32)] = v;
                        }
                    }
                }
        dstbuf.reserve(packed.size() * sizeof(float));
        NC_HIP(hipMemcpy(dstbuf.p, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
    };
    pack(cfg, w, w_phase_stride);
    alts.clear();
    static const bool no_alts = env_flag("NC_NO_TILE_ALTS");
    if (!no_alts && rows() >= 128)
        for (int tm = 3; tm >= 1; --tm) {   // (single-row-block tiles: slower on every filled grid, chosen only for the tiny grids of choose_tile)
            if (tm == cfg.TM || rows() % (32 * tm) != 0) continue;
            alts.emplace_back(new Alt());
            alts.back()->cfg = cfg;
            alts.back()->cfg.TM = tm;
            pack(alts.back()->cfg, alts.back()->w, alts.back()->w_phase_stride);
        }
    // whole-channel tile of the wide fused residual units (C = 192 / 256 -> TM = 6 / 8; launched only through ConvIO::fuse_k1)
    static const bool no_wide_fuse = env_flag("NC_NO_WIDE_FUSE");
    const bool wide_c = Cin == Cout && (Cout == 256 || Cout == 192);
    if (!no_alts && !no_wide_fuse && !transposed && K == 7 && stride == 1 && wide_c) {
        alts.emplace_back(new Alt());
        alts.back()->cfg = cfg;
        alts.back()->cfg.TM = Cout / 32;
        pack(alts.back()->cfg, alts.back()->w, alts.back()->w_phase_stride);
    }
    if (!transposed && K == 1 && Cin == Cout && Cin % 32 == 0 && (Cin <= 128 || (wide_c && !no_wide_fuse))) {
        // image for the fused residual-unit tail: [row block][ci][32 rows]
        std::vector<float> f((size_t)Cin * Cout);
        for (int i2 = 0; i2 < Cout / 32; ++i2)
            for (int ci = 0; ci < Cin; ++ci)
                for (int rr = 0; rr < 32; ++rr) f[((size_t)i2 * Cin + ci) * 32 + rr] = dense_w[(size_t)(i2 * 32 + rr) * Cin + ci];
        w_fused.reserve(f.size() * sizeof(float));
        NC_HIP(hipMemcpy(w_fused.p, f.data(), f.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    if (!transposed && K == 1 && Cout <= 16 && Cin % 64 == 0) {
        // image for skinny_proj_kernel: lane = (k4 << 4) | row, value W[row][4*kp + k4]
        std::vector<float> f((size_t)Cin * 16, 0.0f);
        for (int kp = 0; kp < Cin / 4; ++kp)
            for (int l = 0; l < 64; ++l) {
                const int k4 = l >> 4, r = l & 15;
                if (r < Cout) f[(size_t)kp * 64 + l] = dense_w[(size_t)r * Cin + 4 * kp + k4];
            }
        w_skinny.reserve(f.size() * sizeof(float));
        NC_HIP(hipMemcpy(w_skinny.p, f.data(), f.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    if (!transposed && stride == 1 && dil == 1 && Cin == 1 && K == 7) {   // dense [Cout][1][K] image for the streaming stem kernel
        w_stem.reserve(sizeof(float) * (size_t)Cout * K);
        NC_HIP(hipMemcpy(w_stem.p, dense_w, sizeof(float) * (size_t)Cout * K, hipMemcpyHostToDevice));
    }
    {   // image for the short-row kernel (nc_conv_small.hip): the strided down-convolutions, taken when a launch has few columns
        static const bool no_small = env_flag("NC_NO_CONV_SMALL");
        if (!no_small && conv_small_eligible(Cin, Cout, K, stride, dil, transposed)) {
            std::vector<float> img;
            pack_conv_small(dense_w, Cin, Cout, K, img);
            w_small.reserve(img.size() * sizeof(float));
            NC_HIP(hipMemcpy(w_small.p, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    if (!transposed && stride == 1 && Cout <= 2 && (K == 7 || K == 3 || K == 1)) {
        w_thin.reserve(sizeof(float) * (size_t)Cout * Cin * K);
        NC_HIP(hipMemcpy(w_thin.p, dense_w, sizeof(float) * (size_t)Cout * Cin * K, hipMemcpyHostToDevice));
    }
    has_bias = bias_h != nullptr;
    if (has_bias) {
        bias.reserve(sizeof(float) * Cout);
        NC_HIP(hipMemcpy(bias.p, bias_h, sizeof(float) * Cout, hipMemcpyHostToDevice));
    }
}

bool can_fuse_res_unit(const ConvLayer& k7, const ConvLayer& k1) {
    bool tile = k7.Cout <= 128;
    for (const auto& a : k7.alts) tile = tile || a->cfg.BM() == k7.Cout;   // wide units: the whole-channel tile was packed at load
    return !k7.transposed && k7.K == 7 && k7.stride == 1 && k7.Cin == k7.Cout && k7.Cout % 32 == 0 && k7.Cout >= 64 && tile &&
           k1.K == 1 && k1.Cin == k7.Cout && k1.Cout == k7.Cout && k1.w_fused.p != nullptr &&
           k7.has_bias && k1.has_bias;
}


// Which packed row-tile height to launch: estimated time = rounds * (blocks per CU) * TM * penalty(TM), with
// rounds = ceil(blocks / (256 CUs * blocks per CU)).  Smaller tiles waste a little more LDS/issue bandwidth per MFMA (penalty) but
// can turn a 1.1-round grid into a full one (measured: C=768 at T=696 76 -> 99 TFLOP/s with 96-row tiles).
struct TileChoice {
    TileCfg cfg;
    const float* w;
    int64_t w_phase_stride;
};
static TileChoice choose_tile(const ConvLayer& L, int64_t blocks_per_rowtile, bool pointwise_fast, bool xv_cand = false) {
    auto cost = [&](const TileCfg& c) {
        const int n_co = (L.rows() + c.BM() - 1) / c.BM();
        const double blocks = (double)blocks_per_rowtile * n_co;
        static const int bpc_gen[5] = {0, 4, 3, 2, 2}, bpc_pw[5] = {0, 6, 5, 3, 3};
        static const double pen_gen[5] = {0, 1.30, 1.10, 1.05, 1.00};
        // pointwise kernel (re-fitted in round 4 after its ring / addressing changes: 96-row tiles are now its most efficient -- C = 384
        // over 32 x 5568 columns 460 us with 128-row tiles, 427 with 96)
        static const double pen_pw[5] = {0, 1.30, 1.10, 1.02, 1.06};
        const double* pen = pointwise_fast ? pen_pw : pen_gen;
        const int bpc = pointwise_fast ? bpc_pw[c.TM] : bpc_gen[c.TM];
        const double rounds = std::ceil(blocks / (256.0 * bpc));
        // (k = 7 at 128 rows x 256 columns runs out of registers -- 212 B of scratch per lane; C = 256: 1.80 ms against 1.65 ms with 64-row tiles)
        double inst = (c.TM == 4 && c.K == 7 && !pointwise_fast) ? 1.12 : 1.0;
        // Two-tap sub-pixel up-convolutions with SHORT reductions (Cin <= 384: 12-24 reduction blocks per tile): a tile's prologue and
        // epilogue -- an output-heavy epilogue: stride x more samples out than in -- are 10-15 % of its life, and the 64-row instance is the
        // only one of the three that does not spill (241 registers; 96 rows: 256 + 26 spilled, 128 rows: 256 + 136).  Measured with the XV
        // staging (tools/probe/tm_pick_up.sh, one box): 192 -> 96 at 32 x 22 272 columns 1079 us with 96-row tiles, 980 with 64-row tiles
        // (SNAC's at 8 x 110 592: 1318 / 1200); 384 -> 192: 1943 / 1834; from 768 input channels on the 96-row tiles win (2119 / 2215).
        static const bool xv_off = env_flag("NC_NO_XV") || env_flag("NC_NO_XR");   // (the legacy 64-row instance does not win: conv_up 5.94 -> 6.15 ms)
        if (xv_cand && !xv_off && L.sub_stride && c.K == 2 && c.TM == 2 && L.Cin <= 384) inst = 0.88;   // (xv_cand: this launch takes the XV-only instance)
        return rounds * bpc * c.TM * pen[c.TM] * inst;
    };
    TileChoice best{L.cfg, L.w.as<float>(), L.w_phase_stride};
    double bc = cost(L.cfg);
    {   // Tiny grids (one-clip / short-row launches: even with 32-row tiles every workgroup gets a CU of its own): a workgroup's life is
        // its serial reduction, TM matrix-core chains long per step, so the SMALLEST row tile finishes first -- SNAC 24 kHz at one clip:
        // the 384 -> 768 k=16 down-convolution ran 8 workgroups of 96 rows for 450 us (C1: 2.85 ms in all).
        static const bool no_tiny = env_flag("NC_NO_TINY_TILES");
        static const int tiny_blocks = (int)env_int("NC_TINY_BLOCKS", 256);
        if (!no_tiny) {
            // the smallest packed row tile whose grid still stays under `tiny_blocks` workgroups
            int tm = 0;
            for (int cand = 1; cand < L.cfg.TM && !tm; ++cand) {
                if (blocks_per_rowtile * ((L.rows() + 32 * cand - 1) / (32 * cand)) > tiny_blocks) continue;
                for (const auto& a : L.alts)
                    if (a->cfg.TM == cand) { tm = cand; best = TileChoice{a->cfg, a->w.as<float>(), a->w_phase_stride}; }
            }
            if (tm) return best;
        }
    }
    static const int tm_pick = (int)env_int("NC_TM_PICK", 0);   // experiment: force a packed variant
    if (tm_pick) {
        for (const auto& a : L.alts)
            if (a->cfg.TM == tm_pick && tm_pick <= 4) return TileChoice{a->cfg, a->w.as<float>(), a->w_phase_stride};
        return best;
    }
    for (const auto& a : L.alts) {
        if (a->cfg.TM > 4 || a->cfg.TM == 1) continue;   // whole-channel tiles of the wide fused units; 32-row tiles: tiny grids only (above)
        const double c = cost(a->cfg);
        if (c < bc) {
            bc = c;
            best = TileChoice{a->cfg, a->w.as<float>(), a->w_phase_stride};
        }
    }
    return best;
}

// Row tiles per group of the block -> tile order (ConvArgs::co_group).  The ~64 workgroups resident on an XCD advance through their
// reduction blocks roughly in step, so operands they have in common are fetched from the fabric once and then hit in that XCD's L2
// (temporal sharing: the instantaneous working set is a few tiles, far below the 4 MB).  With groups of G row tiles the resident set
// is G row tiles x 64/G column tiles: an input window is fetched once per GROUP (x_bytes * n_co / G in all) and a weight panel once per
// resident set that holds it (w_bytes * n_col_tiles * G / 64 in all).  Pick the divisor of n_co_tiles (<= 8) with the least traffic.
// Measured (PMC FETCH_SIZE, DAC C2): G = 1 -> 2 on the C = 384 k = 7 layers 1156 -> 709 MiB per launch, as this model predicts.
// NC_CO_GROUP=<n> caps the group size (1 = one panel per XCD, the round-1 order).
static int pick_co_group(int n_co_tiles, double x_bytes, double w_bytes, double n_col_tiles) {
    static const int cap = (int)env_int("NC_CO_GROUP", 0);
    int best = 1;
    double bt = 0.0;
    for (int g = 1; g <= std::min(n_co_tiles, cap > 0 ? cap : 8); ++g) {
        if (n_co_tiles % g) continue;
        const double t = x_bytes * n_co_tiles / g + w_bytes * n_col_tiles * g / 64.0;
        if (g == 1 || t < bt) { bt = t; best = g; }
    }
    return best;
}

// Streaming k = 3 path of the Encodec residual branches (nc_conv3s.hip): reflect pad 1 + 1 of SConv1d folded into the lane exchange,
// pending GroupNorm + ELU applied once per element in registers, no LDS for the activations.
conv_kernel_fn conv3_stream_kernel_table(int, bool);
static bool launch_conv3_stream(const ConvLayer& L, const ConvIO& io, int B, hipStream_t stream, Profiler* prof) {
    static const bool off = env_flag("NC_NO_CONV3S");
    if (off || L.transposed || L.K != 3 || L.stride != 1 || L.dil != 1 || L.pad != 0 || L.cfg.CB != 16 || (L.Cin & 1) || L.Cin > 512) return false;
    if (io.alpha_in || io.alpha_out || io.res || io.epi || io.fuse_k1 || io.x2 || io.noise) return false;
    // the Encodec input mode with the non-causal pad of a k = 3, stride 1 SConv1d: one reflected sample on either side, no zero extension
    const int64_t T = io.in_L;
    if (T < 4 || (T & 1) || io.in_left != 1 || io.in_Lz != T || io.Tin != T + 2 || io.x_len != T + 2) return false;
    if ((io.y_cstride & 1) || (io.y_bstride & 1)) return false;
    auto al8 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 7) == 0; };
    if (!al8(io.y)) return false;
    const bool x_aligned = al8(io.x) && !(io.x_cstride & 1) && !(io.x_bstride & 1);
    if ((int64_t)(L.Cin + 1) * io.x_cstride + T >= ((int64_t)1 << 32)) return false;
    const TileChoice tc = choose_tile(L, (int64_t)B * ((T + 255) / 256), true);
    conv_kernel_fn fn = conv3_stream_kernel_table(tc.cfg.TM, x_aligned);
    if (!fn) return false;
    ConvArgs a{};
    a.x = io.x; a.x_bstride = io.x_bstride; a.x_cstride = io.x_cstride; a.Cin = L.Cin; a.x_len = (int32_t)T;
    a.w = tc.w;
    a.bias = L.has_bias ? L.bias.as<float>() : nullptr;
    a.y = io.y; a.y_bstride = io.y_bstride; a.y_cstride = io.y_cstride;
    a.in_mode = (io.in_stats ? 1 : 0) | (io.in_elu ? 2 : 0);
    a.in_stats = io.in_stats; a.in_gamma = io.in_gamma; a.in_beta = io.in_beta;
    a.gn_part = io.gn_part; a.gn_nrb = io.gn_nrb; a.gn_ncb = io.gn_ncb; a.gn_count = io.gn_count; a.gn_stats = io.gn_stats; a.gn_n = io.gn_n;
    a.Cout = L.Cout; a.B = B; a.Tout = (int32_t)T;
    const int BM = tc.cfg.BM();
    a.n_co_tiles = (L.Cout + BM - 1) / BM;
    a.n_t_tiles = (int32_t)((T + 255) / 256);
    a.n_cb = (L.Cin + 15) / 16;
    a.co_group = pick_co_group(a.n_co_tiles, 4.0 * B * L.Cin * (double)T, 4.0 * 3 * L.Cin * (double)L.Cout, (double)B * a.n_t_tiles);
    const int64_t grid = (int64_t)a.n_co_tiles * B * a.n_t_tiles;
    if (prof && prof->on)
        prof->begin(stream, L.kclass, 2.0 * L.Cin * L.Cout * 3 * (double)T * B, 4.0 * ((double)B * L.Cin * T + (double)B * L.Cout * T + 3.0 * L.Cin * L.Cout));
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(256), 0, stream, a);
    NC_HIP(hipGetLastError());
    if (prof && prof->on) prof->end(stream);
    return true;
}

// Pointwise fast path (nc_conv1x1.hip): B fragments straight from global memory, 2-wide vector loads/stores.
static bool launch_conv1x1(const ConvLayer& L, const ConvIO& io, int B, hipStream_t stream, Profiler* prof) {
    static const bool off = env_flag("NC_NO_CONV1X1");
    const int64_t T = io.Tin;
    if (off || L.transposed || L.K != 1 || L.stride != 1 || L.pad != 0 || L.cfg.CB != 16 || io.fuse_k1 || (L.Cin & 1)) return false;
    if (io.alpha_in || (io.epi & ~EPI_NOISE) || T < 2 || (T & 1) || io.x_len != T) return false;
    if (io.alpha_out && (io.epi & EPI_NOISE)) return false;
    if ((io.x_cstride & 1) || (io.x_bstride & 1) || (io.y_cstride & 1) || (io.y_bstride & 1)) return false;
    if ((int64_t)L.Cin * io.x_cstride * 4 >= ((int64_t)1 << 32)) return false;   // (the B reads are buffer loads over one clip's rows)
    auto al8 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 7) == 0; };
    if (!al8(io.x) || !al8(io.y) || (io.res && !al8(io.res)) || (io.noise && !al8(io.noise))) return false;
    if ((io.epi & EPI_NOISE) && (!io.noise || !io.res)) return false;
    const int in_mode = (io.in_stats ? 1 : 0) | (io.in_elu ? 2 : 0);
    if (io.in_L > 0 || io.x2) return false;   // (reflect addressing / two operands: the windowed template)
    if (in_mode && (io.res || io.alpha_out || io.epi || L.Cin > 512)) return false;
    const TileChoice tc = choose_tile(L, (int64_t)B * ((T + 255) / 256), true);
    const int mode = in_mode ? 8 : (io.epi & EPI_NOISE) ? 4 : ((io.res ? 1 : 0) | (io.alpha_out ? 2 : 0));
    if (io.gn_part && mode != 8) return false;   // block sums are emitted by the input-mode instance only (the windowed template has them everywhere)
    conv_kernel_fn fn = conv1x1_kernel_table(tc.cfg.TM, mode);
    if (!fn) return false;
    ConvArgs a{};
    a.x = io.x; a.x_bstride = io.x_bstride; a.x_cstride = io.x_cstride; a.Cin = L.Cin; a.x_len = io.x_len;
    a.w = tc.w;
    a.bias = L.has_bias ? L.bias.as<float>() : nullptr;
    a.res = io.res; a.noise = io.noise; a.noise_bstride = T; a.epi = io.epi; a.alpha_out = io.alpha_out;
    a.y = io.y; a.y_bstride = io.y_bstride; a.y_cstride = io.y_cstride;
    a.in_mode = in_mode; a.in_stats = io.in_stats; a.in_gamma = io.in_gamma; a.in_beta = io.in_beta;
    a.gn_part = io.gn_part; a.gn_nrb = io.gn_nrb; a.gn_ncb = io.gn_ncb; a.gn_count = io.gn_count; a.gn_stats = io.gn_stats; a.gn_n = io.gn_n;
    a.Cout = L.Cout; a.B = B; a.Tout = (int32_t)T;
    const int BM = tc.cfg.BM();
    a.n_co_tiles = (L.Cout + BM - 1) / BM;
    a.n_t_tiles = (int32_t)((T + 255) / 256);
    a.n_cb = (L.Cin + 15) / 16;
    a.co_group = pick_co_group(a.n_co_tiles, 4.0 * B * L.Cin * (double)T, 4.0 * L.Cin * (double)L.Cout, (double)B * a.n_t_tiles);
    int64_t grid = (int64_t)a.n_co_tiles * B * a.n_t_tiles;
    size_t lds = 0;
    {   // streaming variant: narrow long rows, whole weight tile of a row tile resident in LDS (see conv1x1_stream_kernel)
        // (not the default since round 4: with the in-place B ring and buffer-load addressing the tile-per-workgroup kernel is 1-6 % faster on
        // these layers in steady state, DAC conv_k1 class 3.92 -> 3.79 ms; round 5: the streaming variant is compiled by `make EXPERIMENTS=1`
        // only, where NC_PW_STREAM=1 selects it -- tools/probe/envmatrix.sh builds that library and runs the parity suites under it)
#ifdef NC_EXPERIMENTS
        static const bool no_stream = !env_flag("NC_PW_STREAM");
#else
        constexpr bool no_stream = true;
#endif
        const size_t need = sizeof(float) * ((size_t)L.Cin * BM + 3 * (size_t)BM);
        conv_kernel_fn sfn = (!no_stream && !in_mode && mode <= 4 && L.Cin % 32 == 0 && L.Cin <= 192 && L.Cout % BM == 0 && need <= 76 * 1024 &&
                              grid >= 2048)
                                 ? conv1x1_stream_kernel_table(tc.cfg.TM, mode)
                                 : nullptr;
        if (sfn) {
            fn = sfn;
            lds = need;
            ensure_dynamic_lds((const void*)fn, 80 * 1024);
            const int64_t per = std::max<int64_t>(1, 512 / a.n_co_tiles);            // two workgroups per CU, a whole number per row tile
            grid = (int64_t)a.n_co_tiles * std::min<int64_t>(per, (int64_t)B * a.n_t_tiles);
        }
    }
    if (prof && prof->on) {
        const double bytes = 4.0 * ((double)B * L.Cin * T + (double)B * L.Cout * T * (io.res ? 2 : 1) + (double)L.Cin * L.Cout);
        prof->begin(stream, L.kclass, L.flops(B, T), bytes);
    }
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(256), lds, stream, a);
    NC_HIP(hipGetLastError());
    if (prof && prof->on) prof->end(stream);
    return true;
}

// thin-output layers the input-mode streaming kernel serves: Conv1d(C -> 1|2, k = 7), stride 1, no dilation, explicit padding
static bool thin_inm_layer(const ConvLayer& L) {
    static const bool off = env_flag("NC_NO_THIN_INM");
    return !off && L.w_thin.p && !L.transposed && L.K == 7 && L.stride == 1 && L.dil == 1 && L.pad == 0 && L.Cout <= 2;
}

bool conv_in2_available(const ConvLayer& L) {
    static const bool off = env_flag("NC_NO_IN2");
    if (off || L.w_stem.p || L.w_skinny.p) return false;
    if (L.w_thin.p) return thin_inm_layer(L);   // the PCM head in the input mode takes both operands (conv_thin_inm_kernel)
    // Only where ONE row tile covers all output rows: every row tile re-stages (normalises twice, adds, activates) the window it
    // shares with the others, and on the matrix-core-bound deep layers that vector work sits on the critical path -- measured on C3:
    // 64 -> 128 k8 419 -> 432 us, 128 -> 256 k10 493 -> 1224 us, 256 -> 512 k16 +370 us against the summed copy + one-input launch,
    // while the single-tile layers gain (32 -> 64 k4: 314 -> 275 us, 64 -> 32 up-conv: 368 -> 311 us).
    if (L.rows() > 64 || L.cfg.TM == 3) return false;
    if (L.sub_stride && !L.sub_shift) return false;   // (the multiply-shift sub-pixel form has no two-input instance)
    return in2_kernel(L.Ktaps, L.sub_shift != 0, L.cfg.TM, 1) != nullptr;
}

// Short-row kernel (nc_conv_small.hip) for this launch?  0 = no, 1 = 16-column tiles (latency-bound launches: few workgroups of any
// shape), 2 = 32-column tiles (k = 16 layers whose template grid would not fill the chip twice: 256 -> 512 at 150 frames x 32 rows
// 365 -> 272 us, 512 -> 1024 at 87 frames x 32 clips 713 -> 572 us; with >= 512 template workgroups the template wins).
static int conv_small_choice(const ConvLayer& L, const ConvIO& io, int B) {
    if (!L.w_small.p || B <= 0) return 0;
    const int in_mode = (io.in_stats ? 1 : 0) | (io.in_elu ? 2 : 0) | (io.in_L > 0 ? 4 : 0) | (io.x2 ? 8 : 0);
    if ((in_mode & 8) || io.alpha_in || io.res || io.fuse_k1 || io.epi != 0) return 0;   // (the reflect-padded view alone is fine: an index map)
    if ((in_mode & 3) && !conv_small_inm_available(L.Cin, L.K, L.stride, L.dil)) return 0;
    static const int64_t max_grid = env_int("NC_SMALL_MAX_GRID", 2048);
    static const int64_t wide_below = env_int("NC_SMALL_WIDE_BELOW", 512);
    const int64_t Tout = L.out_len(io.Tin);
    if (L.K == 1) {   // wide pointwise GEMMs over few columns (the chunked LSTM input projections): 32-column form only
        // (round 6: up to 1024 columns, was 4096 -- every 64-row x 32-column workgroup streams its 128 KB weight panel out of L2, so the 44-step
        //  chunks of C3 (1408 columns: 1408 workgroups, 180 MB of panels, 54 us) run faster on the pointwise kernel's 256-column tiles:
        //  C3 8.47 -> 8.38 ms, C2 53.80 -> 53.67 ms, C5 / C1 unchanged; tools/probe/r6_k1cols2.sh)
        static const int64_t k1_cols = env_int("NC_SMALL_K1_COLS", 1024);
        return (!io.gn_part && (int64_t)B * Tout <= k1_cols) ? 2 : 0;
    }
    const int64_t grid16 = (int64_t)B * ((Tout + 15) / 16) * ((L.Cout + 63) / 64);
    if (grid16 <= max_grid) return 1;
    const int64_t template_grid = (int64_t)((L.rows() + L.cfg.BM() - 1) / L.cfg.BM()) * (((int64_t)B * Tout + 255) / 256);
    if (conv_small_max_tn(L.Cin, L.K, L.stride, L.dil) >= 2 && template_grid < wide_below) return 2;
    return 0;
}

bool conv_gn_fusable(const ConvLayer& L, const ConvIO& io, int B) {
    static const bool off = env_flag("NC_NO_GN_FUSE");
    // plain epilogues only; one launch covering the whole output (no per-phase transposed launches); the streaming thin-output /
    // stem / skinny kernels keep the stand-alone statistics pass (launch_conv skips them when gn_part is set, so the answer here only
    // has to say which layers are WORTH routing through the matrix-core template: all but those three)
    if (off || io.res || io.alpha_out || io.alpha_in || io.epi || io.fuse_k1 || L.n_phase != 1 || (L.sub_stride && !L.sub_shift)) return false;
    if (L.w_thin.p) return thin_inm_layer(L) && io.in_L > 0;   // (the input-mode head kernel emits its sums; the plain head does not)
    if (L.w_stem.p || L.w_skinny.p) return false;
    if (conv_small_choice(L, io, B))   // (its 32-column instances reduce the blocks from an LDS copy of the tile; the others: stand-alone pass)
        return conv_small_gn_available(L.Cin, L.K, L.stride, L.dil, io.in_stats != nullptr || io.in_elu);
    return true;
}

void launch_conv(const ConvLayer& L, const ConvIO& io, int B, hipStream_t stream, Profiler* prof) {
    static const bool no_skinny = env_flag("NC_NO_SKINNY");
    const int in_mode = (io.in_stats ? 1 : 0) | (io.in_elu ? 2 : 0) | (io.in_L > 0 ? 4 : 0) | (io.x2 ? 8 : 0);
    if (in_mode && (io.alpha_in || io.fuse_k1)) fail(NC_ESTATE, "internal: the Encodec input mode does not combine with Snake / fused units");
    if (io.x2 && (!conv_in2_available(L) || (io.in_stats != nullptr) != (io.in_stats2 != nullptr)))
        fail(NC_ESTATE, "internal: no two-input kernel for this layer");
    if (io.gn_part && !conv_gn_fusable(L, io, B)) fail(NC_ESTATE, "internal: this launch cannot emit GroupNorm block sums");
    if (L.w_skinny.p && !no_skinny && !in_mode && !io.alpha_in && !io.alpha_out && !io.res && io.epi == 0 && !io.fuse_k1 && io.x_len == io.Tin) {
        if (prof && prof->on)
            prof->begin(stream, L.kclass, L.flops(B, io.Tin), 4.0 * ((double)B * L.Cin * io.Tin + (double)B * L.Cout * io.Tin + (double)L.Cin * L.Cout));
        launch_skinny_proj(io.x, io.x_bstride, io.x_cstride, L.w_skinny.as<float>(), L.has_bias ? L.bias.as<float>() : nullptr, io.y,
                           io.y_bstride, io.y_cstride, B, L.Cin, L.Cout, io.Tin, stream);
        if (prof && prof->on) prof->end(stream);
        return;
    }
    if (thin_inm_layer(L) && io.in_L > 0 && !io.alpha_in && !io.alpha_out && !io.res && !io.fuse_k1 && io.epi == 0 &&
        (!io.x2 || (io.in_stats != nullptr) == (io.in_stats2 != nullptr))) {
        // the PCM head in the Encodec input mode: both operands, normalise + add + ELU + reflect pad while staging, GroupNorm sums of the output
        ThinInmArgs t{};
        t.xa = io.x; t.xb2 = io.x2; t.x_bstride = io.x_bstride; t.x_cstride = io.x_cstride;
        t.Cin = L.Cin; t.L = (int)io.in_L; t.left = (int)io.in_left; t.Lz = (int)io.in_Lz; t.Lp = (int)io.Tin;
        t.stats_a = io.in_stats; t.gamma_a = io.in_gamma; t.beta_a = io.in_beta;
        t.stats_b = io.in_stats2; t.gamma_b = io.in_gamma2; t.beta_b = io.in_beta2;
        t.elu = io.in_elu ? 1 : 0;
        t.w = L.w_thin.as<float>(); t.bias = L.has_bias ? L.bias.as<float>() : nullptr;
        t.y = io.y; t.y_bstride = io.y_bstride; t.y_cstride = io.y_cstride;
        t.Tout = (int)L.out_len(io.Tin);
        t.gn_part = io.gn_part; t.gn_ncb = io.gn_ncb; t.gn_count = io.gn_count; t.gn_stats = io.gn_stats; t.gn_n = io.gn_n;
        if (io.gn_part && io.gn_nrb != 1) fail(NC_ESTATE, "internal: thin head with more than one GroupNorm row block");
        ProfScope ps(prof, stream, L.kclass, L.flops(B, io.Tin),
                     4.0 * ((double)B * L.Cin * io.in_L * (io.x2 ? 2 : 1) + (double)B * L.Cout * t.Tout + (double)L.Cin * L.Cout * L.K));
        if (launch_conv_thin_inm(t, B, L.Cout, stream)) return;
    }
    {   // thin-output layers (PCM heads): streaming kernel instead of a 32-row matrix tile with 1-2 live rows
        static const bool no_thin = env_flag("NC_NO_THIN");
        if (L.w_thin.p && !no_thin && !in_mode && !io.alpha_in && !io.alpha_out && !io.res && !io.fuse_k1 && (io.epi & ~EPI_TANH) == 0) {
            const int64_t Tout = L.out_len(io.Tin);
            if (prof && prof->on)
                prof->begin(stream, L.kclass, L.flops(B, io.Tin), 4.0 * ((double)B * L.Cin * io.Tin + (double)B * L.Cout * Tout + (double)L.Cin * L.Cout * L.K));
            const bool done = launch_conv_thin(io.x, io.x_bstride, io.x_cstride, L.Cin, io.x_len, L.w_thin.as<float>(), L.has_bias ? L.bias.as<float>() : nullptr,
                                               io.y, io.y_bstride, io.y_cstride, B, L.Cout, L.K, L.pad, L.dil, Tout, (io.epi & EPI_TANH) != 0, stream);
            if (prof && prof->on) prof->end(stream);
            if (done) return;
        }
    }
    {   // thin-input layers (stems, Cin == 1): streaming store of Cout rows
        static const bool no_stem = env_flag("NC_NO_STEM");
        if (L.w_stem.p && !no_stem && !in_mode && !io.alpha_in && !io.res && !io.fuse_k1 && io.epi == 0 && io.x_cstride >= 0) {
            const int64_t Tout = L.out_len(io.Tin);
            ProfScope ps(prof, stream, L.kclass, L.flops(B, io.Tin), 4.0 * ((double)B * io.Tin + (double)B * L.Cout * Tout));
            if (launch_conv_stem(io.x, io.x_bstride, io.x_len, L.w_stem.as<float>(), L.has_bias ? L.bias.as<float>() : nullptr, io.alpha_out, io.y,
                                 io.y_bstride, io.y_cstride, B, L.Cout, L.K, L.pad, Tout, stream))
                return;
        }
    }
    const int64_t Tout = L.out_len(io.Tin);
    const bool small_gn_ok = !io.gn_part || conv_small_gn_available(L.Cin, L.K, L.stride, L.dil, io.in_stats != nullptr || io.in_elu);
    if (const int small_tn = small_gn_ok ? conv_small_choice(L, io, B) : 0) {
        // Short rows of a few-clip batch (one-clip SNAC / DAC: the deep down-convolutions over 47 .. 375 frames) and the k = 16 layers
        // whose template grid leaves most of the chip to lone workgroups: the 16x16x4 kernel of nc_conv_small.hip
        if ((int64_t)(L.Cin) * io.x_cstride + io.x_len < ((int64_t)1 << 40)) {
            ProfScope ps(prof, stream, L.kclass, L.flops(B, io.Tin),
                         4.0 * ((double)B * L.Cin * io.Tin + (double)B * L.Cout * Tout + (double)L.Cin * L.Cout * L.K));
            const ConvSmallGn sgn{io.gn_part, io.gn_nrb, io.gn_ncb, io.gn_count, io.gn_stats, io.gn_n};
            if (launch_conv_small(io.x, io.x_bstride, io.x_cstride, io.x_len, (int)io.in_left, (int)io.in_Lz, (int)io.in_L, io.in_stats, io.in_gamma, io.in_beta,
                                  io.in_elu ? 1 : 0, &sgn, L.w_small.as<float>(), L.has_bias ? L.bias.as<float>() : nullptr, io.alpha_out,
                                  io.y, io.y_bstride, io.y_cstride, B, L.Cin, L.Cout, L.K, L.stride, L.pad, L.dil, (int)Tout, small_tn, stream))
                return;
        }
    }
    if (launch_conv1x1(L, io, B, stream, prof)) return;
    if (launch_conv3_stream(L, io, B, stream, prof)) return;
    const int64_t n_cols_all = L.transposed ? io.Tin + L.Ktaps - 1 : Tout;
    TileChoice tsel{L.cfg, L.w.as<float>(), L.w_phase_stride};
    // (launches that will take the XV-only two-tap instance: plain input, rows on 64-byte boundaries -- the same alignment test the grant
    //  below applies; its remaining conditions depend on the tile and are checked there)
    const bool xv_cand = L.sub_stride && L.n_phase == 1 && !io.in_stats && !io.in_elu && io.in_L == 0 && !io.x2 && !io.alpha_in && !io.gn_part &&
                         !io.fuse_k1 && io.x_len % 4 == 0 && io.x_cstride % 16 == 0 && io.x_bstride % 16 == 0 && (reinterpret_cast<uintptr_t>(io.x) & 63) == 0;
    if (!io.fuse_k1) tsel = choose_tile(L, (int64_t)L.n_phase * B * ((n_cols_all + 255) / 256), false, xv_cand);
    else
        for (const auto& alt : L.alts)   // the fused residual unit needs the tile that spans all channels
            if (alt->cfg.BM() == L.Cout) tsel = TileChoice{alt->cfg, alt->w.as<float>(), alt->w_phase_stride};
    TileCfg c = tsel.cfg;
    static const int tn_thresh = (int)env_int("NC_TN_THRESH", 192);
    c.TN = n_cols_all >= tn_thresh ? 2 : 1;  // 256-column tiles for long clips, 128 for the deep (short) layers
    {   // the per-lane staging registers bound the window: fall back to 128-column tiles when it does not fit
        const int sx0 = L.transposed ? 1 : L.stride, ad0 = L.transposed ? 1 : L.dil;
        const int xw2 = (c.BN() - 1) * sx0 + (L.Ktaps - 1) * ad0 + 1;
        if (c.TN == 2 && c.CB * ((xw2 + 63) / 64) > 4 * nx_for_k(c.K)) c.TN = 1;
    }
    // narrow variant (3 waves, 96 columns): rows of 65..96 columns would leave a quarter of a 128-column tile on padding
    bool narrow = false;
    {
        static const bool no_narrow = env_flag("NC_NO_NARROW");
        const int rem = (int)(n_cols_all % 128);
        const int sx0 = L.transposed ? 1 : L.stride, ad0 = L.transposed ? 1 : L.dil;
        const int xw96 = 95 * sx0 + (L.Ktaps - 1) * ad0 + 1;
        const bool fits = c.CB * ((xw96 + 63) / 64) <= 3 * nx_for_k(c.K);
        if (!no_narrow && fits && !io.fuse_k1 && !io.x2 && !(L.sub_stride && !L.sub_shift) && c.TN == 1 && n_cols_all <= 96 && rem > 64 && narrow_kernel(c.K, c.TM)) {
            narrow = true;
            c.NW = 3;
        }
    }
    // Flattened (clip, column) axis (kernel: "Flattened column axis"): when the rows are short or leave a good part of their last
    // tile on padding, the columns of all clips are cut into tiles as one axis.  Needs the one-launch forms (no per-phase launches),
    // no per-clip scalars in the kernel (a pending GroupNorm of the Encodec input mode -- measured: per-segment statistics through an LDS table made every
    // instance of the template ~5 % slower for 0.07 ms on C3 -- and noise rows), a window (tile + one halo per touched clip) that still fits
    // the staging registers, and 32-bit offsets that reach 3 clips ahead.
    bool flat = false;
    int flat_S = 0, flat_hc = 0;
    int64_t flat_pitch = 0;   // columns per clip on the flattened axis (>= n_cols_all)
    {
        static const bool no_flat = env_flag("NC_NO_FLAT");
        const int sx0 = L.transposed ? 1 : L.stride, ad0 = L.transposed ? 1 : L.dil;
        const int hc = ((L.Ktaps - 1) * ad0) / sx0;
        // clip pitch on the flattened axis: the row length, or -- when the epilogue emits GroupNorm block sums -- the row length rounded up
        // to whole 32-column blocks (= 32 * gn_ncb), so that every 32x32 accumulator tile is one canonical block of one sample
        static const bool no_flat_gn = env_flag("NC_NO_FLAT_GN");
        const int64_t Tq = io.gn_part ? (int64_t)32 * io.gn_ncb : n_cols_all;
        auto segs = [&](int BN) { return (int)((BN - 2) / Tq) + 2; };
        auto fits = [&](int TN) {
            const int BN = 128 * TN, S = segs(BN);
            if (S > 4) return false;
            const int xw = (BN - 1 + (S - 1) * hc) * sx0 + (L.Ktaps - 1) * ad0 + 1;
            return c.CB * ((xw + 63) / 64) <= 4 * nx_for_k(c.K);
        };
        const bool cand = !no_flat && B > 1 && L.n_phase == 1 && !io.fuse_k1 && !(in_mode & 1) && !(io.gn_part && (no_flat_gn || Tq < n_cols_all)) && !io.x2 &&
                          !(io.epi & EPI_NOISE) && Tq >= 32 && L.Cin * L.Ktaps >= 64 &&
                          3 * io.x_bstride + io.x_len < ((int64_t)1 << 32) &&
                          (int64_t)(c.BM() + 4) * io.y_cstride + Tout + 3 * io.y_bstride < ((int64_t)1 << 31) &&
                          (Tq + hc) * sx0 < (1 << 28);
        int ftn = 0;
        if (cand) ftn = ((int64_t)B * Tq >= 192 && fits(2)) ? 2 : fits(1) ? 1 : 0;
        if (ftn < c.TN && n_cols_all >= 192) ftn = 0;   // (dilation-9 windows: the extra halo would halve the tile width -- keep the one-clip tiles)
        if (ftn) {
            const int64_t bn_nf = c.BN(), cols_nf = (int64_t)B * ((n_cols_all + bn_nf - 1) / bn_nf) * bn_nf;
            const int64_t bn_f = 128 * ftn, cols_f = (((int64_t)B * Tq + bn_f - 1) / bn_f) * bn_f;
            if ((double)cols_f <= 0.97 * (double)cols_nf) {
                flat = true;
                tsel = choose_tile(L, ((int64_t)B * Tq + 255) / 256, false);
                c = tsel.cfg;
                c.TN = ftn;
                narrow = false;
                flat_S = segs(c.BN());
                flat_pitch = Tq;
                flat_hc = hc;
            }
        }
    }
    {   // Small grids of small layers: two 128-column tiles instead of one 256-column tile when that fills the chip better (the
        // latency-bound layers of the 1-clip / 150-frame configurations: C1 2.93 -> 2.66 ms, C3 11.66 -> 11.43 ms, Encodec 24 kHz
        // 7.05 -> 6.83 ms).  Only where the whole weight set is a few MB: the deep DAC layers stream 16-75 MB of weights per launch and
        // a 128-column tile re-reads them twice as often -- there the same switch LOSES 8-23 % (C = 768 k=7 1.74 -> 1.88 ms, up-conv
        // 1536->768 1.31 -> 1.62 ms) although the round count says otherwise.
        static const bool no_tn_rounds = env_flag("NC_NO_TN_ROUNDS");
        if (!no_tn_rounds && c.TN == 2 && !io.fuse_k1 && !narrow && c.TM <= 4) {
            static const int bpc_gen[5] = {0, 4, 3, 2, 2};
            const double slots = 256.0 * bpc_gen[c.TM];
            const int64_t n_co = (L.rows() + c.BM() - 1) / c.BM();
            auto ntiles = [&](int64_t bn) { return flat ? ((int64_t)B * flat_pitch + bn - 1) / bn : (int64_t)B * ((n_cols_all + bn - 1) / bn); };
            const double r2 = std::ceil((double)(L.n_phase * n_co * ntiles(256)) / slots), r1 = std::ceil((double)(L.n_phase * n_co * ntiles(128)) / slots);
            const double w_bytes = 4.0 * L.rows() * (double)L.Cin * L.Ktaps;
            if (r2 <= 3 && r1 * 1.08 < 2.0 * r2 && w_bytes <= 4.0 * 1024 * 1024) {
                c.TN = 1;
                if (flat) flat_S = (int)((128 - 2) / flat_pitch) + 2;
            }
        }
    }
    // light variant (reduction block of 4 channels, 3 workgroups per CU): same packed weights when Cin is a multiple of 8
    int nx = nx_for_k(c.K);
    bool light = false;
    {
        static const int light_mode = experiment_mode("NC_LIGHT");
        if (light_mode == 1 && !narrow && !flat && !io.fuse_k1 && c.K == 7 && c.CB == 8 && L.Cin % 8 == 0 && c.TN == 2 && (c.TM == 2 || c.TM == 3)) {
            light = true;
            c.CB = 4;
            nx = 5;
        }
    }
    const bool fused_wide = io.fuse_k1 && c.TM > 4;   // C = 192 / 256 residual unit: whole-channel tile, 128 columns, 4-channel blocks
    if (fused_wide) { c.TN = 1; c.CB = 4; nx = 5; }
    bool slim = false;
    conv_kernel_fn slim_fn = nullptr;
    {   // Slim variant: half-size reduction block (half the LDS per workgroup), 4+ workgroups per CU.  The narrow long-T layers
        // (Cout <= 64: k=3 residual-branch convolutions of the SEANet blocks, the 2-channel Encodec stem) spend their time in memory
        // round trips -- two reduction blocks per tile never fill the software pipeline -- so more resident workgroups overlap them:
        // 215 -> 145 us (32->16 k3, 48000 steps x 32 clips), 152 -> 118 us (64->32), 80 -> 51 us (2->32 k7).  Measured neutral or
        // slower for the strided k=4 / k=8 layers and the sub-pixel up-convolutions, which keep the standard blocks.
        static const bool no_slim = env_flag("NC_NO_SLIM");
        if (!no_slim && !flat && !light && !narrow && !io.fuse_k1 && !io.x2 && !L.sub_stride && !L.transposed && c.TM <= 2 && n_cols_all >= 1024) {
            int cb2 = 0, nx2 = 0;
            if (c.K == 3 && c.CB == 16) { slim_fn = conv_kernel_table_slim_k3(c.TM, c.TN); cb2 = 8; nx2 = 10; }
            else if (c.K == 7 && c.CB == 8 && L.Cin <= 4 && L.stride == 1 && L.dil == 1) { slim_fn = conv_kernel_table_slim_k7(c.TM, c.TN); cb2 = 4; nx2 = 5; }
            if (slim_fn) { slim = true; c.CB = cb2; nx = nx2; }
        }
    }
    bool wide = false;
    {
        static const int wide_mode = experiment_mode("NC_WIDE");
        if (wide_mode == 1 && !light && !narrow && !flat && !io.fuse_k1 && c.K == 7 && c.TN == 2 && c.TM >= 2 && n_cols_all >= 2048) {
            wide = true;
            c.NW = 8;
            nx = 9;
        }
    }
    // wave-specialised variant (4 consumer + 2 producer waves): k=7, 64/96-row tiles, long rows
    int n_prod = 0;
    {
        static const int spec_mode = experiment_mode("NC_SPEC");
        if (spec_mode == 1 && !light && !narrow && !flat && !wide && !io.fuse_k1 && c.K == 7 && c.TN == 2 && (c.TM == 2 || c.TM == 3)) {
            n_prod = 2;
            nx = 20;
        }
    }
    bool dist = false;
    {
        static const int dist_mode = experiment_mode("NC_DIST");
        if (dist_mode == 1 && !n_prod && !light && !narrow && !flat && !wide && !io.fuse_k1 && c.K == 7 && c.TN == 2 && c.TM >= 2) dist = true;
    }
    // Distributed staging for the grids that leave a workgroup alone on its CU (the deep strided / sub-pixel layers of Encodec at
    // 150 frames: 20 GFLOP per launch): nobody feeds the matrix pipe during the staging runs of the segmented pipeline -- measured
    // 58 % pipe duty for a lone workgroup against 71 % for a co-resident pair -- so the runs are dealt into the matrix-core shadows.
    conv_kernel_fn dist_small_fn = nullptr;
    {
        static const bool off = env_flag("NC_NO_DIST_SMALL");
        static const int64_t max_grid = env_int("NC_DIST_MAX_GRID", 768);
        const int64_t n_co = (L.rows() + c.BM() - 1) / c.BM();
        const int64_t n_tt = flat ? ((int64_t)B * flat_pitch + c.BN() - 1) / c.BN() : (int64_t)B * ((n_cols_all + c.BN() - 1) / c.BN());
        if (!off && !dist && !n_prod && !light && !narrow && !wide && !slim && !in_mode && !io.x2 && !io.fuse_k1 && io.epi == 0 && L.n_phase == 1 &&
            n_co * n_tt <= max_grid) {
            if (L.sub_shift && c.K == 2) dist_small_fn = conv_kernel_table_dist_sub_k2(c.TM, c.TN);
            else if (!L.sub_stride && !L.transposed && c.K == 16) dist_small_fn = conv_kernel_table_dist_k16(c.TM, c.TN);
        }
    }
    const int NW = n_prod ? n_prod : c.NW;   // waves that stage the input window
    const int BM = c.BM(), BN = c.BN(), CB = c.CB, KB = c.KB();
    ConvArgs a{};
    a.x = io.x; a.x_bstride = io.x_bstride; a.x_cstride = io.x_cstride; a.Cin = L.Cin; a.x_len = io.x_len;
    a.alpha_in = io.alpha_in;
    a.in_mode = in_mode; a.in_stats = io.in_stats; a.in_gamma = io.in_gamma; a.in_beta = io.in_beta;
    a.x2 = io.x2; a.in_stats2 = io.in_stats2; a.in_gamma2 = io.in_gamma2; a.in_beta2 = io.in_beta2;
    a.in_left = (int32_t)io.in_left; a.in_Lz = (int32_t)io.in_Lz; a.in_L = (int32_t)io.in_L;
    a.w = tsel.w; a.w_phase_stride = tsel.w_phase_stride;
    a.bias = L.has_bias ? L.bias.as<float>() : nullptr;
    a.alpha_out = io.alpha_out; a.res = io.res;
    a.y = io.y; a.y_bstride = io.y_bstride; a.y_cstride = io.y_cstride;
    a.rvq_zq = io.rvq_zq; a.rvq_res = io.rvq_res;
    a.gn_part = io.gn_part; a.gn_nrb = io.gn_nrb; a.gn_ncb = io.gn_ncb; a.gn_count = io.gn_count; a.gn_stats = io.gn_stats; a.gn_n = io.gn_n;
    a.noise = io.noise; a.noise_bstride = L.out_len(io.Tin);
    if ((io.epi & EPI_NOISE) && (!io.noise || !io.res)) fail(NC_ESTATE, "internal: noise epilogue needs noise and residual");
    a.Cout = L.rows(); a.sub_shift = L.sub_shift; a.B = B; a.epi = io.epi;
    if (L.sub_stride && !L.sub_shift) {
        if (io.epi & EPI_NOISE) fail(NC_ESTATE, "internal: noise epilogue on the multiply-shift sub-pixel form");
        a.sub_stride = L.sub_stride; a.sub_cout = L.Cout; a.sub_magic = magic_div(L.sub_stride, L.rows() + 256);
        if ((int64_t)(L.Cout + 4) * io.y_cstride + Tout + 3 * io.y_bstride >= (int64_t)1 << 31)
            fail(NC_EUNSUPPORTED, "conv output of %lld samples per row exceeds the 32-bit offsets of the sub-pixel form", (long long)io.y_cstride);
    }
    a.Tout = (int32_t)Tout;
    if ((int64_t)(c.BM() + 4) * io.y_cstride + Tout >= (int64_t)1 << 31)
        fail(NC_EUNSUPPORTED, "conv output rows of %lld samples exceed the 32-bit tile offsets", (long long)io.y_cstride);
    int sx;  // x step per output column
    if (L.transposed) {
        sx = 1; a.stride = 1; a.dil = -1; a.pad = 0;
        a.n_cols = (int32_t)(io.Tin + L.Ktaps - 1);
        a.y_tstride = L.stride; a.y_toff = -L.pad;
        a.n_phase = L.n_phase;   // 1 in sub-pixel form
    } else {
        sx = L.stride; a.stride = L.stride; a.dil = L.dil; a.pad = L.pad;
        a.n_cols = (int32_t)Tout;
        a.y_tstride = 1; a.y_toff = 0;
        a.n_phase = 1;
    }
    const int ad = a.dil < 0 ? -a.dil : a.dil;
    a.xneg = a.dil < 0 ? (L.Ktaps - 1) * ad : 0;
    int xv_extra = 0;
    bool duo_launch = false;          // ... in its two-tiles-per-workgroup form (nc_conv_kernel.hip.h "DUO")
    conv_kernel_fn xv_fn = nullptr;   // XV-only instance of this launch (nc_conv_kernel.hip.h "XVK"), when its staging form applies
    {   // XV (round 5): vectorised window staging of the two-tap sub-pixel instances (the kernel's XV note) -- plain input, rows and window
        // start on 16-byte boundaries (xneg is raised by up to 3 slots for that: the window still fits its 320-slot pitch), whole float4s
        static const bool no_xv = env_flag("NC_NO_XV");
        static const bool no_xr = env_flag("NC_NO_XR");
        // k = 7: the XV-only instances are worth 1.4-2.9 % per layer wherever the rows start on 64-byte boundaries (row pitch a multiple of 16
        // samples) and LOSE 25 % where they do not: C = 768 at 696 steps (2784-byte rows: every other channel row starts 32 bytes into a
        // 64-byte sector) 1632 -> 2038 us, at 704 steps 1601 -> 1577, at 1024 2104 -> 2041, at 5568 11 254 -> 10 943
        // (tools/probe/xvk7_rows.py, profiles/r05_xvk7_rows.txt): the 8-byte vector loads are that sensitive, the legacy dword loads are not.
        static const bool no_xv_k7 = env_flag("NC_NO_XV_K7");
        static const int64_t xv_k7_min = env_int("NC_XV_K7_MIN_COLS", 0);
        const bool xv_k7 = !no_xv_k7 && n_cols_all >= xv_k7_min;
        const bool two_tap = L.sub_stride && L.n_phase == 1 && c.K == 2 && c.CB == 16 && !io.alpha_in && !io.fuse_k1;
#ifdef NC_EXPERIMENTS
        static const bool duo_env = env_flag("NC_DUO");
#else
        constexpr bool duo_env = false;
#endif
        const bool duo7 = duo_env && !io.fuse_k1 && (c.TM == 2 || c.TM == 3);
        const bool k7 = (xv_k7 || duo7) && !L.transposed && !L.sub_stride && c.K == 7 && c.CB == 8 && L.stride == 1 && !fused_wide;   // (the fused units included)
        const int vw = two_tap ? 4 : 2;   // floats per staged word
        if (!no_xv && !no_xr && (two_tap || k7) && c.TN == 2 && c.NW == 4 && c.TM >= 2 && c.TM <= 4 && !narrow && !flat && !dist_small_fn && !n_prod &&
            !light && !wide && !dist && !slim && !in_mode && !io.x2 && !io.gn_part && sx == 1 && L.Cin % c.CB == 0 && io.x_len % vw == 0 &&
            io.x_cstride % 16 == 0 && io.x_bstride % 16 == 0 && (reinterpret_cast<uintptr_t>(io.x) & 63) == 0) {   // (rows on 64-byte boundaries: see above)
            xv_extra = (vw - (a.pad + a.xneg) % vw) % vw;
            if ((BN - 1) * sx + (L.Ktaps - 1) * ad + 1 + xv_extra <= 320) {
                xv_fn = two_tap ? (L.sub_shift ? conv_kernel_table_xv_sub_k2(c.TM) : conv_kernel_table_xv_subg_k2(c.TM))
                                : duo7 ? conv_kernel_table_duo_k7(c.TM)
                                       : (io.fuse_k1 ? conv_kernel_table_xv_fused_k7(c.TM) : conv_kernel_table_xv_k7(c.TM));
                duo_launch = xv_fn && !two_tap && duo7;
            }
            if (xv_fn) { a.xneg += xv_extra; a.epi |= EPI_XVEC; }
            else xv_extra = 0;
        }
    }
    a.xw = (BN - 1 + (flat ? (flat_S - 1) * flat_hc : 0)) * sx + (L.Ktaps - 1) * ad + 1 + xv_extra;
    a.nchunk = (a.xw + 63) / 64;
    a.xwp = (a.nchunk * 64 + sx - 1) / sx;   // rows are padded to whole 64-slot chunks: every staging store is in-bounds
    a.xrow = sx == 1 ? a.nchunk * 64 : sx * a.xwp;
    a.n_co_tiles = (L.rows() + BM - 1) / BM;
    a.n_t_tiles = (a.n_cols + BN - 1) / BN;
    {
        static const bool no_xr = env_flag("NC_NO_XR");
        if (no_xr) a.epi |= EPI_NO_XR;
    }
    a.Bc = B; a.flat = 0; a.flat_px = a.flat_pc = 0x1fffffff; a.flat_hc = 0;
    if (flat) {   // one column axis over all clips: B = 1 in the tile map
        a.flat = 1; a.flat_pc = (int32_t)flat_pitch; a.flat_hc = flat_hc; a.flat_px = (int32_t)(flat_pitch + flat_hc) * sx;
        a.n_t_tiles = (int32_t)(((int64_t)B * flat_pitch + BN - 1) / BN);
        a.B = 1;
    }
    a.n_cb = (L.Cin + CB - 1) / CB;
    a.co_group = 1;   // (set below, once the column tiling is known)
    a.n_items = CB * a.nchunk;
    if (a.n_items > NW * nx)
        fail(NC_EUNSUPPORTED, "conv K=%d stride=%d dil=%d: input window of %d words per channel exceeds the staging registers",
             L.K, L.stride, L.dil, a.xw);
    a.xbuf = (((NW * nx - 1) / a.nchunk + 1) * a.xrow + 3) & ~3;   // items past n_items land in pad rows
    a.chunk_magic = magic_div(a.nchunk, NW * nx + NW);
    a.stride_magic = magic_div(sx, a.nchunk * 64 + 64);
    for (int k = 0; k < 16; ++k) {
        const int q = k * a.dil + a.xneg;
        a.tapoff[k] = (k < L.Ktaps) ? (sx == 1 ? q : (q % sx) * a.xwp + q / sx) : 0;
    }
    size_t lds_f = 2 * (size_t)KB * BM + 2 * (size_t)a.xbuf + ((io.alpha_in || (in_mode & 1)) ? 2 * (size_t)a.n_cb * CB * (io.x2 ? 2 : 1) : 0);
    if (io.fuse_k1) lds_f = std::max(lds_f, fused_wide ? (size_t)2 * BM * 32 : (size_t)BM * BM);   // the 1x1 weights reuse the tile buffers
    a.ep_off = (int32_t)lds_f;
    size_t lds = sizeof(float) * (lds_f + 6 * (size_t)BM);
    conv_kernel_fn fn = nullptr;
    if (io.fuse_k1) {
        if (!can_fuse_res_unit(L, *io.fuse_k1) || !io.alpha_out || !io.res || io.epi != 0)
            fail(NC_ESTATE, "internal: residual unit is not fusable");
        a.w2 = io.fuse_k1->w_fused.as<float>();
        a.bias2 = io.fuse_k1->bias.as<float>();
        a.alpha_out2 = io.alpha_out2;
        fn = fused_wide ? conv_kernel_table_fusedw_k7(c.TM, c.TN) : conv_kernel_table_fused_k7(c.TM, c.TN);
        if (!fn) fail(NC_EUNSUPPORTED, "no fused residual-unit kernel for TM=%d TN=%d", c.TM, c.TN);
        if (xv_fn) fn = xv_fn;
    } else if (xv_fn) {
        fn = xv_fn;
    } else if (dist_small_fn) {
        fn = dist_small_fn;
    } else if (dist) {
        fn = conv_kernel_table_dist_k7(c.TM, c.TN);
        if (!fn) fail(NC_EUNSUPPORTED, "no distributed-staging conv kernel for TM=%d TN=%d", c.TM, c.TN);
    } else if (n_prod) {
        fn = conv_kernel_table_spec_k7(c.TM, c.TN);
        if (!fn) fail(NC_EUNSUPPORTED, "no specialised conv kernel for TM=%d TN=%d", c.TM, c.TN);
    } else if (io.x2) {
        fn = in2_kernel(c.K, L.sub_shift != 0, c.TM, c.TN);
        if (!fn) fail(NC_EUNSUPPORTED, "no two-input conv kernel for K=%d TM=%d TN=%d", c.K, c.TM, c.TN);
    } else if (L.sub_stride && !L.sub_shift) {
        fn = c.K != 2 ? nullptr : conv_kernel_table_subg_k2(c.TM, c.TN);
        if (!fn) fail(NC_EUNSUPPORTED, "no sub-pixel conv kernel for K=%d TM=%d TN=%d (stride %d)", c.K, c.TM, c.TN, L.sub_stride);
    } else if (L.sub_shift) {
        fn = c.K != 2 ? nullptr : narrow ? conv_kernel_table_sub_narrow_k2(c.TM) : conv_kernel_table_sub_k2(c.TM, c.TN);
        if (!fn) fail(NC_EUNSUPPORTED, "no sub-pixel conv kernel for K=%d TM=%d TN=%d", c.K, c.TM, c.TN);
    } else if (narrow) {
        fn = narrow_kernel(c.K, c.TM);
    } else if (wide) {
        fn = conv_kernel_table_wide_k7(c.TM, c.TN);
        if (!fn) fail(NC_EUNSUPPORTED, "no wide conv kernel for TM=%d TN=%d", c.TM, c.TN);
    } else if (slim) {
        fn = slim_fn;
    } else if (light) {
        fn = conv_kernel_table_light_k7(c.TM, c.TN);
        if (!fn) fail(NC_EUNSUPPORTED, "no light conv kernel for TM=%d TN=%d", c.TM, c.TN);
    } else {
        fn = lookup_kernel(c);
    }
    {   // experiment: NC_LDS_MIN=<bytes> raises the LDS request (fewer co-resident workgroups per CU)
        static const size_t lds_min = (size_t)env_int("NC_LDS_MIN", 0);
        lds = std::max(lds, std::min<size_t>(lds_min, 160 * 1024));
    }
    if (lds > 160 * 1024) fail(NC_EUNSUPPORTED, "conv tile needs %zu B of LDS", lds);
    ensure_dynamic_lds((const void*)fn, 160 * 1024);
    a.co_group = pick_co_group(a.n_co_tiles, 4.0 * B * L.Cin * (double)io.Tin, 4.0 * L.Cin * (double)L.rows() * L.Ktaps,
                               (double)a.B * a.n_t_tiles);
    const int64_t grid = (int64_t)a.n_phase * a.n_co_tiles * a.B * a.n_t_tiles;
    if (grid <= 0) return;
    if (prof && prof->on) {
        const double bytes = 4.0 * ((double)B * L.Cin * io.Tin + (double)B * L.Cout * Tout + (double)L.Cin * L.Cout * L.K);
        double fl = L.flops(B, io.Tin);
        if (io.fuse_k1) fl += io.fuse_k1->flops(B, io.Tin);
        prof->begin(stream, L.kclass, fl, bytes);
    }
#ifdef NC_CONV_TRACE
    // diagnostic builds: NC_CONV_TRACE_FILE=<path> + NC_CONV_TRACE_SEL="K,Cin,dil" picks the first matching launch (see the kernel's NC_STAMP)
    static const char* trace_file = env_str("NC_CONV_TRACE_FILE");
    static bool traced = false;
    DevBuf trace_buf;
    bool trace_now = false;
    if (trace_file && !traced && !io.x2) {
        int tk = 7, tc = 384, td = 1;
        if (const char* sel = env_str("NC_CONV_TRACE_SEL")) std::sscanf(sel, "%d,%d,%d", &tk, &tc, &td);
        if (c.K == tk && L.Cin == tc && (L.dil == td || L.transposed) && a.n_cb >= 16) {
            trace_buf.reserve((size_t)16 * 8 * 8 * 8 * 8);
            NC_HIP(hipMemsetAsync(trace_buf.p, 0, (size_t)16 * 8 * 8 * 8 * 8, stream));
            a.x2 = trace_buf.as<float>();
            trace_now = traced = true;
        }
    }
#endif
    if (duo_launch) {   // two tiles per workgroup: 8 wavefronts, each sub-workgroup its own LDS half
        const size_t lds_sub = sizeof(float) * (((size_t)a.ep_off + 6 * (size_t)BM + 3) & ~(size_t)3);
        if (2 * lds_sub > 160 * 1024) fail(NC_EUNSUPPORTED, "DUO conv tile needs %zu B of LDS", 2 * lds_sub);
        hipLaunchKernelGGL(fn, dim3((unsigned)((grid + 1) / 2)), dim3(512), 2 * lds_sub, stream, a);
    } else
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(64 * (c.NW + n_prod)), lds, stream, a);
    NC_HIP(hipGetLastError());
    if (prof && prof->on) prof->end(stream);
#ifdef NC_CONV_TRACE
    if (trace_now) {
        std::vector<unsigned long long> hb((size_t)16 * 8 * 8 * 8);
        NC_HIP(hipStreamSynchronize(stream));
        NC_HIP(hipMemcpy(hb.data(), trace_buf.p, hb.size() * 8, hipMemcpyDeviceToHost));
        if (FILE* f = std::fopen(trace_file, "wb")) {
            const int hdr[8] = {16, c.NW + n_prod, 8, 8, c.TM, c.TN, c.K, (a.epi & EPI_XVEC) ? 1 : 0};
            std::fwrite(hdr, sizeof(int), 8, f);
            std::fwrite(hb.data(), 8, hb.size(), f);
            std::fclose(f);
        }
        trace_buf.release();
    }
#endif
    {   // NC_LAUNCH_LOG=<path>: one line per conv-template launch (class, threads, shape) in launch order.  The template serves several
        // kernel classes under one kernel name; tools/pmc_classes.py zips this log with the rocprofv3 counter rows of the same
        // kernel name (dispatch order) to attribute HBM traffic / matrix-core busy cycles to exactly the launches a class counts.
        static FILE* lf = [] { const char* p = env_str("NC_LAUNCH_LOG"); return p ? std::fopen(p, "w") : (FILE*)nullptr; }();
        if (lf) {
            std::fprintf(lf, "conv_mfma %d %lld %d %d %d %lld %d\n", L.kclass, (long long)grid * 64 * (c.NW + n_prod), L.Cin, L.Cout, L.K,
                         (long long)io.Tin, io.fuse_k1 ? 1 : 0);
            std::fflush(lf);
        }
    }
}

}  // namespace nc
