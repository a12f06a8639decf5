// SNAC on the engine: weight import (fold D3 + pack) and the Encode / FromCodes / Decode launch sequences.
//
// Reference call stacks restated as kernel launches:
//   SNAC.Encode   Models/SNAC.cs:113-150 -> Modules/SNAC/Encoder.cs:26-69 -> ResidualVectorQuantizer.cs:69-90
//   SNAC.Decode   Models/SNAC.cs:157-192 -> ResidualVectorQuantizer.cs:100-135 -> Decoder.cs:31-86 / DecoderBlock.cs:29-70
// Per ResidualUnit (ResidualUnit.cs:33-59, depthwise flavour):
//   launch 1: depthwise k7 dil d  [Snake(a1) on the input window | bias | Snake(a2) on the store]     (HBM-bound kernel)
//   launch 2: dense 1x1 on the matrix cores [bias | + x residual]
// NoiseBlock (NoiseBlock.cs:36-46) is one 1x1 launch with the epilogue  y = x + noise * conv(x).
#include <cmath>

#include "nc_elem.h"
#include "nc_model.h"

namespace nc {

// w = (v / ||v||) * (g - 1e-7) per dim-0 slice (Modules/SNAC/WNConv1d.cs:132-135, deviation D3; canonical fold of DESIGN.md)
void fold_weight_norm_snac(const float* v, const float* g, int64_t d0, int64_t inner, float* w) {
    for (int64_t i = 0; i < d0; ++i) {
        double ss = 0.0;
        for (int64_t j = 0; j < inner; ++j) {
            const float q = v[i * inner + j] * v[i * inner + j];
            ss += (double)q;
        }
        const float norm = std::sqrt((float)ss);
        const float gg = g[i] - 1e-7f;
        for (int64_t j = 0; j < inner; ++j) w[i * inner + j] = (v[i * inner + j] / norm) * gg;
    }
}

static const char* G0 = ".parametrizations.weight.original0";
static const char* G1 = ".parametrizations.weight.original1";

static void upload(DevBuf& d, const float* h, size_t n) {
    d.reserve(n * sizeof(float));
    NC_HIP(hipMemcpy(d.p, h, n * sizeof(float), hipMemcpyHostToDevice));
}

static void load_vec(const Blob& b, const std::string& name, int64_t n, DevBuf& dst) {
    const BlobTensor& t = b.get(name);
    if (t.numel() != n) fail(NC_EINVAL, "%s: expected %lld elements, found %lld", name.c_str(), (long long)n, (long long)t.numel());
    upload(dst, static_cast<const float*>(t.data), (size_t)n);
}

// folded dense weight of one weight-normalised conv; returns bias pointer (nullable)
static const float* fold_conv(const Blob& b, const std::string& prefix, int64_t d0, int64_t d1, int64_t K, int64_t n_bias,
                              std::vector<float>& w) {
    const BlobTensor& v = b.get(prefix + G1);
    const BlobTensor& g = b.get(prefix + G0);
    const BlobTensor* bias = b.find(prefix + ".bias");
    if (v.dims.size() != 3 || v.dims[0] != d0 || v.dims[1] != d1 || v.dims[2] != K)
        fail(NC_EINVAL, "%s: weight has the wrong shape", prefix.c_str());
    if (g.numel() != d0) fail(NC_EINVAL, "%s: weight_g must hold one value per dim-0 slice", prefix.c_str());
    if (bias && bias->numel() != n_bias) fail(NC_EINVAL, "%s.bias has the wrong length", prefix.c_str());
    w.resize((size_t)v.numel());
    fold_weight_norm_snac(static_cast<const float*>(v.data), static_cast<const float*>(g.data), d0, d1 * K, w.data());
    return bias ? static_cast<const float*>(bias->data) : nullptr;
}

static void load_dense(const Blob& b, const std::string& prefix, ConvLayer& L, int Cin, int Cout, int K, int stride, int pad, int dil,
                       int out_pad, bool transposed, int kclass) {
    std::vector<float> w;
    const float* bias = fold_conv(b, prefix, transposed ? Cin : Cout, transposed ? Cout : Cin, K, Cout, w);
    L.kclass = kclass;
    L.build(w.data(), bias, Cin, Cout, K, stride, pad, dil, out_pad, transposed);
}

static void load_dw(const Blob& b, const std::string& prefix, DwConvLayer& L, int C, int K, int pad, int dil) {
    std::vector<float> w;
    const float* bias = fold_conv(b, prefix, C, 1, K, C, w);
    L.build(w.data(), bias, C, K, pad, dil);
}

SnacModel::SnacModel(const nc_snac_config& c) : cfg(c) {
    if (c.n_encoder_rates <= 0 || c.n_encoder_rates > 8 || c.n_decoder_rates <= 0 || c.n_decoder_rates > 8 || c.n_vq_strides <= 0 ||
        c.n_vq_strides > 8)
        fail(NC_EINVAL, "rate / stride lists must hold 1..8 entries");
    if (c.encoder_dim <= 0 || c.decoder_dim <= 0 || c.codebook_size <= 0 || c.codebook_dim <= 0 || c.sample_rate <= 0)
        fail(NC_EINVAL, "SNAC config fields must be positive");                       // SNACValidator.cs:21-61
    hop = 1;
    for (int i = 0; i < c.n_encoder_rates; ++i) {
        if (c.encoder_rates[i] <= 0) fail(NC_EINVAL, "encoder rate must be positive");
        hop *= c.encoder_rates[i];
    }
    for (int i = 0; i < c.n_decoder_rates; ++i)
        if (c.decoder_rates[i] <= 0) fail(NC_EINVAL, "decoder rate must be positive");
    for (int i = 0; i < c.n_vq_strides; ++i)
        if (c.vq_strides[i] <= 0) fail(NC_EINVAL, "vq stride must be positive");
    if (c.attn_window_size < 0 || c.attn_window_size > 32) fail(NC_EINVAL, "attention window must be in 0..32");
    latent = c.latent_dim > 0 ? c.latent_dim : c.encoder_dim * (1 << c.n_encoder_rates);
    cfg.latent_dim = latent;
    int64_t a = c.vq_strides[0], b = c.attn_window_size > 0 ? c.attn_window_size : 1, x = a, y = b;
    while (y) { const int64_t t = x % y; x = y; y = t; }
    pad_to = (int64_t)hop * (a / x * b);                                              // SNAC.cs:74-76, MathUtils.cs:11-62
}

void SnacModel::load_res_unit(const Blob& b, const std::string& q, ResUnit& ru, int C, int dil) {
    load_vec(b, q + ".block.0.alpha", C, ru.a1);
    if (cfg.depthwise) load_dw(b, q + ".block.1", ru.dw, C, 7, 3 * dil, dil);
    else load_dense(b, q + ".block.1", ru.c7, C, C, 7, 1, 3 * dil, dil, 0, false, NC_KC_CONV_K7);
    load_vec(b, q + ".block.2.alpha", C, ru.a2);
    load_dense(b, q + ".block.3", ru.c1, C, C, 1, 1, 0, 1, 0, false, NC_KC_CONV_K1);
    if (cfg.depthwise && SnacFusedUnit::supported(C, 7, dil)) {   // the unit as one launch (nc_snac_unit.hip): same folded weights
        std::vector<float> w7, w1;
        const float* b7 = fold_conv(b, q + ".block.1", C, 1, 7, C, w7);
        const float* b1 = fold_conv(b, q + ".block.3", C, C, 1, C, w1);
        ru.fu.build(C, dil, w7.data(), b7, static_cast<const float*>(b.get(q + ".block.0.alpha").data),
                    static_cast<const float*>(b.get(q + ".block.2.alpha").data), w1.data(), b1);
    }
}

void SnacModel::load_mha(const Blob& b, const std::string& p, Mha& m, int C) {
    m.C = C;
    load_vec(b, p + ".norm.weight", C, m.gamma);
    load_vec(b, p + ".norm.bias", C, m.beta);
    const BlobTensor& wq = b.get(p + ".to_qkv.weight");
    const BlobTensor& wo = b.get(p + ".to_out.weight");
    const BlobTensor& fr = b.get(p + ".rel_pos.inv_freq");
    if (wq.numel() != (int64_t)3 * C * C || wo.numel() != (int64_t)C * C || fr.numel() != 32)
        fail(NC_EINVAL, "%s: LocalMHA tensors have the wrong shape", p.c_str());
    m.qkv.kclass = NC_KC_CONV_K1;
    m.qkv.build(static_cast<const float*>(wq.data), nullptr, C, 3 * C, 1, 1, 0, 1, 0, false);   // Linear(C -> 3C, no bias)
    m.out.kclass = NC_KC_CONV_K1;
    m.out.build(static_cast<const float*>(wo.data), nullptr, C, C, 1, 1, 0, 1, 0, false);       // Linear(C -> C, no bias)
    // rotary tables (SinusoidalEmbedding.cs:67-80, RotaryEmbedding.cs:46-68): f = fl32(pos*inv_freq), rounded binary64 cos/sin
    const int W = cfg.attn_window_size;
    const float* invf = static_cast<const float*>(fr.data);
    std::vector<float> cs((size_t)W * 64), sn((size_t)W * 64);
    for (int i = 0; i < W; ++i)
        for (int j = 0; j < 64; ++j) {
            const float f = (float)i * invf[j & 31];
            cs[(size_t)i * 64 + j] = (float)std::cos((double)f);
            sn[(size_t)i * 64 + j] = (float)std::sin((double)f);
        }
    upload(m.cs, cs.data(), cs.size());
    upload(m.sn, sn.data(), sn.size());
}

void SnacModel::load(const Blob& b) {
    use_device();
    char nm[256];
    int d = cfg.encoder_dim;
    load_dense(b, "encoder.block.0", enc_stem, 1, d, 7, 1, 3, 1, 0, false, NC_KC_STEM);
    static const int kDil[3] = {1, 3, 9};
    for (int bi = 0; bi < cfg.n_encoder_rates; ++bi) {
        const int s = cfg.encoder_rates[bi];
        for (int u = 0; u < 3; ++u) {
            snprintf(nm, sizeof nm, "encoder.block.%d.block.%d", bi + 1, u);
            load_res_unit(b, nm, enc[bi].ru[u], d, kDil[u]);
        }
        snprintf(nm, sizeof nm, "encoder.block.%d", bi + 1);
        const std::string p = nm;
        load_vec(b, p + ".block.3.alpha", d, enc[bi].a_down);
        load_dense(b, p + ".block.4", enc[bi].down, d, 2 * d, 2 * s, s, (s + 1) / 2, 1, 0, false, NC_KC_CONV_DOWN);
        d *= 2;
    }
    if (d != latent) fail(NC_EINVAL, "latent_dim %d does not match the encoder width %d", latent, d);
    int n = cfg.n_encoder_rates + 1;
    if (cfg.attn_window_size > 0) {
        snprintf(nm, sizeof nm, "encoder.block.%d", n++);
        load_mha(b, nm, enc_mha, d);
    }
    snprintf(nm, sizeof nm, "encoder.block.%d", n);
    if (cfg.depthwise) load_dw(b, nm, enc_out_dw, d, 7, 3, 1);
    else load_dense(b, nm, enc_out, d, d, 7, 1, 3, 1, 0, false, NC_KC_CONV_MISC);

    in_proj.clear(); out_proj.clear(); codebooks.clear();
    for (int i = 0; i < cfg.n_vq_strides; ++i) {
        snprintf(nm, sizeof nm, "quantizer.quantizers.%d", i);
        const std::string p = nm;
        in_proj.emplace_back(new ConvLayer());
        out_proj.emplace_back(new ConvLayer());
        codebooks.emplace_back(new Codebook());
        load_dense(b, p + ".in_proj", *in_proj.back(), latent, cfg.codebook_dim, 1, 1, 0, 1, 0, false, NC_KC_CONV_K1);
        load_dense(b, p + ".out_proj", *out_proj.back(), cfg.codebook_dim, latent, 1, 1, 0, 1, 0, false, NC_KC_CONV_K1);
        const BlobTensor& cb = b.get(p + ".codebook.weight");
        if (cb.dims.size() != 2 || cb.dims[0] != cfg.codebook_size || cb.dims[1] != cfg.codebook_dim)
            fail(NC_EINVAL, "%s.codebook.weight has the wrong shape", p.c_str());
        codebooks.back()->build(static_cast<const float*>(cb.data), cfg.codebook_size, cfg.codebook_dim);
    }

    const int ch = cfg.decoder_dim;
    if (cfg.depthwise) {
        load_dw(b, "decoder.model.0", dec_in_dw, latent, 7, 3, 1);
        load_dense(b, "decoder.model.1", dec_in, latent, ch, 1, 1, 0, 1, 0, false, NC_KC_CONV_K1);
        n = 2;
    } else {
        load_dense(b, "decoder.model.0", dec_in, latent, ch, 7, 1, 3, 1, 0, false, NC_KC_CONV_MISC);
        n = 1;
    }
    if (cfg.attn_window_size > 0) {
        snprintf(nm, sizeof nm, "decoder.model.%d", n++);
        load_mha(b, nm, dec_mha, ch);
    }
    int out_dim = ch;
    for (int bi = 0; bi < cfg.n_decoder_rates; ++bi) {
        const int s = cfg.decoder_rates[bi];
        const int in_dim = ch >> bi;
        out_dim = ch >> (bi + 1);
        if (out_dim <= 0) fail(NC_EINVAL, "decoder_dim too small for the number of decoder blocks");
        snprintf(nm, sizeof nm, "decoder.model.%d", n++);
        const std::string p = nm;
        load_vec(b, p + ".block.0.alpha", in_dim, dec[bi].a_up);
        load_dense(b, p + ".block.1", dec[bi].up, in_dim, out_dim, 2 * s, s, (s + 1) / 2, 1, s % 2, true, NC_KC_CONV_UP);
        int k = 2;
        if (cfg.noise) {
            load_dense(b, p + ".block.2.linear", dec[bi].noise, out_dim, out_dim, 1, 1, 0, 1, 0, false, NC_KC_CONV_K1);
            k = 3;
        }
        for (int u = 0; u < 3; ++u) {
            snprintf(nm, sizeof nm, "%s.block.%d", p.c_str(), k + u);
            load_res_unit(b, nm, dec[bi].ru[u], out_dim, kDil[u]);
        }
    }
    snprintf(nm, sizeof nm, "decoder.model.%d.alpha", n);
    load_vec(b, nm, out_dim, dec_alpha_out);
    snprintf(nm, sizeof nm, "decoder.model.%d", n + 1);
    load_dense(b, nm, dec_out, out_dim, 1, 7, 1, 3, 1, 0, false, NC_KC_HEAD);
    NC_HIP(hipDeviceSynchronize());
    loaded = true;
}

int64_t SnacModel::decoded_len(int64_t fr) const {
    int64_t L = fr;
    for (int i = 0; i < cfg.n_decoder_rates; ++i) L = up_len(L, cfg.decoder_rates[i]);
    return L;
}

int64_t SnacModel::noise_len(int B, int64_t fr) const {
    if (!cfg.noise) return 0;
    int64_t L = fr, n = 0;
    for (int i = 0; i < cfg.n_decoder_rates; ++i) {
        L = up_len(L, cfg.decoder_rates[i]);
        n += (int64_t)B * L;
    }
    return n;
}

static ConvIO io_for(const float* x, int C, int64_t L, float* y, int Cy, int64_t Ly) {
    ConvIO io{};
    io.x = x; io.x_bstride = (int64_t)C * L; io.x_cstride = L; io.x_len = (int32_t)L; io.Tin = L;
    io.y = y; io.y_bstride = (int64_t)Cy * Ly; io.y_cstride = Ly;
    return io;
}

// x + conv1(snake(conv7(snake(x))))
float* SnacModel::run_res_unit(ResUnit& ru, float* cur, int C, int64_t L, int B, int& cur_idx, const float* alpha_next) {
    const int h_idx = (cur_idx + 1) % 3, o_idx = (cur_idx + 2) % 3;
    float* h = act[h_idx].as<float>();
    float* o = act[o_idx].as<float>();
    if (cfg.depthwise && ru.fu.usable(cur, o, L, B)) {
        ru.fu.launch(cur, alpha_next, o, B, L, cu_count, stream, &prof);
        cur_idx = o_idx;
        return o;
    }
    if (cfg.depthwise) {
        launch_dwconv(ru.dw, cur, ru.a1.as<float>(), ru.a2.as<float>(), h, B, L, stream, &prof);
    } else {
        ConvIO io = io_for(cur, C, L, h, C, L);
        io.alpha_in = ru.a1.as<float>(); io.alpha_out = ru.a2.as<float>();
        launch_conv(ru.c7, io, B, stream, &prof);
    }
    ConvIO i2 = io_for(h, C, L, o, C, L);
    i2.res = cur; i2.alpha_out = alpha_next;   // Snake of the only consumer, fused into the store
    launch_conv(ru.c1, i2, B, stream, &prof);
    cur_idx = o_idx;
    return o;
}

// LocalMHA.cs:78-115: LayerNorm -> qkv -> rotary windowed attention -> out projection + residual
float* SnacModel::run_mha(Mha& m, float* cur, int C, int64_t L, int B, int& cur_idx, const float* alpha_next) {
    if (L % cfg.attn_window_size != 0) fail(NC_EINVAL, "sequence of %lld frames is not a multiple of the attention window", (long long)L);
    const int n_idx = (cur_idx + 1) % 3, o_idx = (cur_idx + 2) % 3;
    float* xn = act[n_idx].as<float>();
    float* o = act[o_idx].as<float>();
    qkv_ws.reserve((size_t)B * 3 * C * L * 4);
    launch_layernorm_ct(cur, m.gamma.as<float>(), m.beta.as<float>(), xn, B, C, L, stream, &prof);
    ConvIO iq = io_for(xn, C, L, qkv_ws.as<float>(), 3 * C, L);
    launch_conv(m.qkv, iq, B, stream, &prof);
    launch_local_attn(qkv_ws.as<float>(), m.cs.as<float>(), m.sn.as<float>(), xn, B, C, L, cfg.attn_window_size, stream, &prof);
    ConvIO io = io_for(xn, C, L, o, C, L);
    io.res = cur; io.alpha_out = alpha_next;
    launch_conv(m.out, io, B, stream, &prof);
    cur_idx = o_idx;
    return o;
}

void SnacModel::reserve_act(int B, int64_t Tp) {
    // widest tensor of either direction: encoder_dim x Tp (stem output) or decoder last block (decoder_dim>>n x Tp)
    int64_t maxel = (int64_t)cfg.encoder_dim * Tp;
    {
        int c = cfg.encoder_dim; int64_t L = Tp;
        for (int i = 0; i < cfg.n_encoder_rates; ++i) { c *= 2; L /= cfg.encoder_rates[i]; maxel = std::max(maxel, (int64_t)c * L); }
        int64_t Ld = Tp / hop;
        maxel = std::max(maxel, (int64_t)cfg.decoder_dim * Ld);
        for (int i = 0; i < cfg.n_decoder_rates; ++i) {
            Ld = up_len(Ld, cfg.decoder_rates[i]);
            maxel = std::max(maxel, (int64_t)(cfg.decoder_dim >> (i + 1)) * Ld);
        }
    }
    for (auto& a : act) a.reserve((size_t)B * maxel * sizeof(float));
}

// Encode(Tensor) as written (SNAC.cs:113-122): `preprocessed` is computed and dropped, the encoder sees the raw tensor.  Every
// strided conv floors ((L + 2*ceil(s/2) - 2s)/s + 1, EncoderBlock.cs:46-53); the quantizer then needs T' % vq_stride == 0 for its
// repeat_interleave + add (VectorQuantizer.cs:99-101, ResidualVectorQuantizer.cs:82-83: a libtorch shape exception otherwise) and
// LocalMHA reshapes into T'/window windows (LocalMHA.cs:84-91).  Lengths on which the reference throws return NC_EINVAL; the
// attention models additionally need whole windows here (the reference's accidental wider-window cases are not reproduced).
int64_t SnacModel::unpadded_frames(int64_t T) const {
    int64_t L = T;
    for (int bi = 0; bi < cfg.n_encoder_rates; ++bi) {
        L = enc[bi].down.out_len(L);
        if (L <= 0) fail(NC_EINVAL, "input of %lld samples is too short for the encoder", (long long)T);
    }
    for (int i = 0; i < cfg.n_vq_strides; ++i)
        if (L % cfg.vq_strides[i] != 0)
            fail(NC_EINVAL, "un-padded input of %lld samples gives %lld frames, not a multiple of vq stride %d (the reference's quantizer throws)",
                 (long long)T, (long long)L, cfg.vq_strides[i]);
    if (cfg.attn_window_size > 0 && L % cfg.attn_window_size != 0)
        fail(NC_EINVAL, "un-padded input of %lld samples gives %lld frames, not whole attention windows of %d", (long long)T, (long long)L,
             cfg.attn_window_size);
    return L;
}

void SnacModel::encode_dev(const float* pcm, int B, int64_t T, int64_t* codes, float* z_out, float* zq_out, bool pad) {
    if (!loaded) fail(NC_ESTATE, "weights not loaded (call nc_codec_load_weights first)");
    if (!pcm || !codes) fail(NC_EINVAL, "pcm and codes must not be null");
    if (B <= 0 || T <= 0 || T > ((int64_t)1 << 30)) fail(NC_EINVAL, "B and T must be positive");
    use_device();
    const int64_t Tp = pad ? padded_len(T) : T, Tz = pad ? Tp / hop : unpadded_frames(T);
    reserve_act(B, Tp);
    const int D = cfg.codebook_dim;
    resid.reserve((size_t)B * latent * Tz * 4);
    zq.reserve((size_t)B * latent * Tz * 4);
    pooled.reserve((size_t)B * latent * Tz * 4);
    qbuf.reserve((size_t)B * latent * Tz * 4);
    lat.reserve((size_t)B * D * Tz * 4);
    st.reserve((size_t)B * D * Tz * 4);

    int C = cfg.encoder_dim;
    int64_t L = Tp;
    int cur_idx = 0;
    float* cur = act[0].as<float>();
    {   // stem; Preprocess's right zero-pad is the x_len < Tin bound
        ConvIO io{};
        io.x = pcm; io.x_bstride = T; io.x_cstride = T; io.x_len = (int32_t)T; io.Tin = Tp;
        io.y = cur; io.y_bstride = (int64_t)C * L; io.y_cstride = L;
        launch_conv(enc_stem, io, B, stream, &prof);
    }
    for (int bi = 0; bi < cfg.n_encoder_rates; ++bi) {
        for (int u = 0; u < 3; ++u) cur = run_res_unit(enc[bi].ru[u], cur, C, L, B, cur_idx, u == 2 ? enc[bi].a_down.as<float>() : nullptr);
        const int64_t Lo = enc[bi].down.out_len(L);
        const int o_idx = (cur_idx + 1) % 3;
        float* o = act[o_idx].as<float>();
        ConvIO io = io_for(cur, C, L, o, 2 * C, Lo);
        launch_conv(enc[bi].down, io, B, stream, &prof);   // input already carries Snake(a_down) (EncoderBlock.cs:42)
        cur = o; cur_idx = o_idx; C *= 2; L = Lo;
    }
    if (L != Tz) fail(NC_ESTATE, "internal: encoder produced %lld frames, expected %lld", (long long)L, (long long)Tz);
    if (cfg.attn_window_size > 0) cur = run_mha(enc_mha, cur, C, L, B, cur_idx, nullptr);
    float* residual = resid.as<float>();
    if (cfg.depthwise) {
        launch_dwconv(enc_out_dw, cur, nullptr, nullptr, residual, B, L, stream, &prof);
    } else {
        ConvIO io = io_for(cur, C, L, residual, latent, Tz);
        launch_conv(enc_out, io, B, stream, &prof);
    }
    if (z_out) NC_HIP(hipMemcpyAsync(z_out, residual, (size_t)B * latent * Tz * 4, hipMemcpyDeviceToDevice, stream));
    // ---- multi-rate residual vector quantizer (ResidualVectorQuantizer.cs:69-90, VectorQuantizer.cs:82-103)
    float* zq_d = zq_out ? zq_out : zq.as<float>();
    NC_HIP(hipMemsetAsync(zq_d, 0, (size_t)B * latent * Tz * 4, stream));
    int64_t total = 0;
    for (int i = 0; i < cfg.n_vq_strides; ++i) total += Tz / cfg.vq_strides[i];
    int64_t off = 0;
    for (int i = 0; i < cfg.n_vq_strides; ++i) {
        const int s = cfg.vq_strides[i];
        if (Tz % s != 0) fail(NC_EINVAL, "frame count %lld is not a multiple of vq stride %d", (long long)Tz, s);
        const int64_t Ts = Tz / s;
        const float* src = residual;
        if (s > 1) {
            launch_avg_pool(residual, pooled.as<float>(), (int64_t)B * latent, Tz, s, stream, &prof);
            src = pooled.as<float>();
        }
        ConvIO pi = io_for(src, latent, Ts, lat.as<float>(), D, Ts);
        launch_conv(*in_proj[i], pi, B, stream, &prof);
        launch_vq_argmin(*codebooks[i], lat.as<float>(), (int64_t)D * Ts, B, Ts, codes + off, total, st.as<float>(), stream, &prof);
        ConvIO po = io_for(st.as<float>(), D, Ts, qbuf.as<float>(), latent, Ts);
        launch_conv(*out_proj[i], po, B, stream, &prof);
        launch_rvq_update(qbuf.as<float>(), zq_d, residual, (int64_t)B * latent, Tz, s, false, stream, &prof);
        off += Ts;
    }
}

void SnacModel::from_codes_dev(const int64_t* codes, int B, int64_t Tz, float* zq_out) {
    if (!loaded) fail(NC_ESTATE, "weights not loaded (call nc_codec_load_weights first)");
    if (!codes || !zq_out) fail(NC_EINVAL, "codes and zq must not be null");
    if (B <= 0 || Tz <= 0) fail(NC_EINVAL, "B and frames must be positive");
    use_device();
    const int D = cfg.codebook_dim;
    st.reserve((size_t)B * D * Tz * 4);
    qbuf.reserve((size_t)B * latent * Tz * 4);
    int64_t total = 0;
    for (int i = 0; i < cfg.n_vq_strides; ++i) {
        if (Tz % cfg.vq_strides[i] != 0) fail(NC_EINVAL, "frame count %lld is not a multiple of vq stride %d", (long long)Tz, cfg.vq_strides[i]);
        total += Tz / cfg.vq_strides[i];
    }
    int64_t off = 0;
    for (int i = 0; i < cfg.n_vq_strides; ++i) {
        const int s = cfg.vq_strides[i];
        const int64_t Ts = Tz / s;
        launch_vq_gather(*codebooks[i], codes + off, total, B, Ts, st.as<float>(), stream, &prof);
        ConvIO po = io_for(st.as<float>(), D, Ts, qbuf.as<float>(), latent, Ts);
        launch_conv(*out_proj[i], po, B, stream, &prof);
        launch_rvq_update(qbuf.as<float>(), zq_out, nullptr, (int64_t)B * latent, Tz, s, i == 0, stream, &prof);
        off += Ts;
    }
}

void SnacModel::decode_dev(const int64_t* codes, int B, int64_t Tz, const float* noise, uint64_t seed, float* pcm) {
    if (!pcm) fail(NC_EINVAL, "pcm must not be null");
    zq.reserve((size_t)B * latent * Tz * 4);
    from_codes_dev(codes, B, Tz, zq.as<float>());
    reserve_act(B, Tz * hop);
    const float* nz = noise;
    if (cfg.noise && !noise) {
        const int64_t n = noise_len(B, Tz);
        noise_ws.reserve((size_t)n * 4);
        launch_randn(noise_ws.as<float>(), n, seed, stream);
        nz = noise_ws.as<float>();
    }
    int C = latent;
    int64_t L = Tz;
    int cur_idx = 0;
    float* cur = act[0].as<float>();
    if (cfg.depthwise) {
        launch_dwconv(dec_in_dw, zq.as<float>(), nullptr, nullptr, act[1].as<float>(), B, L, stream, &prof);
        ConvIO io = io_for(act[1].as<float>(), C, L, cur, cfg.decoder_dim, L);
        if (cfg.attn_window_size <= 0) io.alpha_out = dec[0].a_up.as<float>();   // consumed only through DecoderBlock's Snake
        launch_conv(dec_in, io, B, stream, &prof);
    } else {
        ConvIO io = io_for(zq.as<float>(), C, L, cur, cfg.decoder_dim, L);
        if (cfg.attn_window_size <= 0) io.alpha_out = dec[0].a_up.as<float>();
        launch_conv(dec_in, io, B, stream, &prof);
    }
    C = cfg.decoder_dim;
    if (cfg.attn_window_size > 0) cur = run_mha(dec_mha, cur, C, L, B, cur_idx, dec[0].a_up.as<float>());
    int64_t noff = 0;
    for (int bi = 0; bi < cfg.n_decoder_rates; ++bi) {
        const int Co = C / 2;
        const int64_t Lo = dec[bi].up.out_len(L);
        int o_idx = (cur_idx + 1) % 3;
        float* o = act[o_idx].as<float>();
        ConvIO io = io_for(cur, C, L, o, Co, Lo);
        launch_conv(dec[bi].up, io, B, stream, &prof);   // input already carries Snake(a_up) (DecoderBlock.cs:37)
        cur = o; cur_idx = o_idx; C = Co; L = Lo;
        if (cfg.noise) {   // x + noise * conv1x1_nobias(x)
            o_idx = (cur_idx + 1) % 3;
            o = act[o_idx].as<float>();
            ConvIO in = io_for(cur, C, L, o, C, L);
            in.res = cur; in.noise = nz + noff; in.epi = EPI_NOISE;
            launch_conv(dec[bi].noise, in, B, stream, &prof);
            cur = o; cur_idx = o_idx;
            noff += (int64_t)B * L;
        }
        const float* a_next = bi + 1 < cfg.n_decoder_rates ? dec[bi + 1].a_up.as<float>() : dec_alpha_out.as<float>();
        for (int u = 0; u < 3; ++u) cur = run_res_unit(dec[bi].ru[u], cur, C, L, B, cur_idx, u == 2 ? a_next : nullptr);
    }
    ConvIO io = io_for(cur, C, L, pcm, 1, L);
    io.epi = EPI_TANH;
    launch_conv(dec_out, io, B, stream, &prof);
}

}  // namespace nc
