// DAC on the engine: weight import (fold + pack) and the Encode / Decode / FromCodes launch sequences.
//
// Reference call stacks restated as kernel launches:
//   DAC.Encode     Models/DAC.cs:163-181  -> Encoder.cs:21-58 -> ResidualVectorQuantizer.cs:54-103
//   DAC.Decode     Models/DAC.cs:231-234  -> Decoder.cs:22-58 / DecoderBlock.cs:20-44
//   DAC.FromCodes  Models/DAC.cs:101-106  -> ResidualVectorQuantizer.cs:211-238
// Fusion plan per ResidualUnit (ResidualUnit.cs:29-34,50-59):
//   launch 1: conv k7 dil d   [Snake(a1) on the input tile | bias | Snake(a2) on the store]
//   launch 2: conv k1         [bias | + x residual]
// so no activation makes an HBM round trip of its own.
#include <cstdlib>
#include <cstring>

#include "nc_model.h"

namespace nc {

Codec::~Codec() {
    if (own_stream) (void)hipStreamDestroy(own_stream);
}

void Codec::init_device(int device_index) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) fail(NC_EDEVICE, "no HIP device available (the engine has no CPU fallback)");
    if (device_index < 0 || device_index >= n) fail(NC_EINVAL, "device index %d out of range (0..%d)", device_index, n - 1);
    device = device_index;
    NC_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    NC_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        fail(NC_EDEVICE, "device %d is %s; this engine is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    cu_count = prop.multiProcessorCount;
    lds_per_cu = prop.maxSharedMemoryPerMultiProcessor;
    NC_HIP(hipStreamCreateWithFlags(&own_stream, hipStreamNonBlocking));
    stream = own_stream;
}

void Codec::use_device() const { NC_HIP(hipSetDevice(device)); }

void Codec::switch_stream(hipStream_t s) {
    if (s == stream) return;
    use_device();
    hipEvent_t e;
    NC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipError_t r = hipEventRecord(e, stream);
    if (r == hipSuccess) r = hipStreamWaitEvent(s, e, 0);
    (void)hipEventDestroy(e);   // destruction is deferred by the runtime until the recorded work completes
    if (r != hipSuccess) fail(NC_EDEVICE, "stream hand-over failed: %s", hipGetErrorString(r));
    stream = s;
}

static void upload(DevBuf& d, const float* h, size_t n) {
    d.reserve(n * sizeof(float));
    NC_HIP(hipMemcpy(d.p, h, n * sizeof(float), hipMemcpyHostToDevice));
}

static void load_alpha(const Blob& b, const std::string& name, int C, DevBuf& dst) {
    const BlobTensor& t = b.get(name);
    if (t.numel() != C) fail(NC_EINVAL, "%s: expected %d elements, found %lld", name.c_str(), C, (long long)t.numel());
    upload(dst, static_cast<const float*>(t.data), C);
}

// weight_v / weight_g / bias -> folded dense weight -> packed device image
static void load_wn_conv(const Blob& b, const std::string& prefix, ConvLayer& L, int Cin, int Cout, int K, int stride, int pad,
                         int dil, bool transposed, int kclass) {
    const BlobTensor& v = b.get(prefix + ".weight_v");
    const BlobTensor& g = b.get(prefix + ".weight_g");
    const BlobTensor* bias = b.find(prefix + ".bias");
    const int64_t d0 = transposed ? Cin : Cout, d1 = transposed ? Cout : Cin;
    if (v.dims.size() != 3 || v.dims[0] != d0 || v.dims[1] != d1 || v.dims[2] != K)
        fail(NC_EINVAL, "%s.weight_v has the wrong shape", prefix.c_str());
    if (g.numel() != d0) fail(NC_EINVAL, "%s.weight_g: expected %lld elements (one per dim-0 slice)", prefix.c_str(), (long long)d0);
    if (bias && bias->numel() != Cout) fail(NC_EINVAL, "%s.bias has the wrong length", prefix.c_str());
    std::vector<float> w((size_t)v.numel());
    fold_weight_norm_dac(static_cast<const float*>(v.data), static_cast<const float*>(g.data), d0, d1 * K, w.data());
    L.kclass = kclass;
    L.build(w.data(), bias ? static_cast<const float*>(bias->data) : nullptr, Cin, Cout, K, stride, pad, dil, 0, transposed);
}

DacModel::DacModel(const nc_dac_config& c) : cfg(c) {
    if (c.n_encoder_rates <= 0 || c.n_encoder_rates > 8 || c.n_decoder_rates <= 0 || c.n_decoder_rates > 8)
        fail(NC_EINVAL, "encoder/decoder rate lists must hold 1..8 entries");
    if (c.encoder_dim <= 0 || c.decoder_dim <= 0 || c.n_codebooks <= 0 || c.codebook_size <= 0 || c.codebook_dim <= 0 ||
        c.sample_rate <= 0)
        fail(NC_EINVAL, "DAC config fields must be positive");
    hop = 1;
    for (int i = 0; i < c.n_encoder_rates; ++i) {
        if (c.encoder_rates[i] <= 0) fail(NC_EINVAL, "encoder rate must be positive");
        hop *= c.encoder_rates[i];
    }
    for (int i = 0; i < c.n_decoder_rates; ++i)
        if (c.decoder_rates[i] <= 0) fail(NC_EINVAL, "decoder rate must be positive");
    if ((c.decoder_dim >> c.n_decoder_rates) <= 0) fail(NC_EINVAL, "decoder_dim too small for the number of decoder blocks");
    latent = c.latent_dim > 0 ? c.latent_dim : c.encoder_dim * (1 << c.n_encoder_rates);  // DAC.cs:64
    if (env_flag("NC_NO_FUSE")) fuse_res_units = false;
    cfg.latent_dim = latent;
}

static const int kDil[3] = {1, 3, 9};  // EncoderBlock.cs:23-25 / DecoderBlock.cs:32-34

void DacModel::load(const Blob& b) {
    use_device();
    char nm[256];
    int d = cfg.encoder_dim;
    load_wn_conv(b, "encoder.block.0", enc_stem, 1, d, 7, 1, 3, 1, false, NC_KC_STEM);
    for (int bi = 0; bi < cfg.n_encoder_rates; ++bi) {
        const int s = cfg.encoder_rates[bi];
        for (int u = 0; u < 3; ++u) {
            snprintf(nm, sizeof nm, "encoder.block.%d.block.%d", bi + 1, u);
            const std::string q = nm;
            load_alpha(b, q + ".block.0.alpha", d, enc[bi].ru[u].a1);
            load_wn_conv(b, q + ".block.1", enc[bi].ru[u].c7, d, d, 7, 1, 3 * kDil[u], kDil[u], false, NC_KC_CONV_K7);
            load_alpha(b, q + ".block.2.alpha", d, enc[bi].ru[u].a2);
            load_wn_conv(b, q + ".block.3", enc[bi].ru[u].c1, d, d, 1, 1, 0, 1, false, NC_KC_CONV_K1);
        }
        snprintf(nm, sizeof nm, "encoder.block.%d", bi + 1);
        const std::string p = nm;
        load_alpha(b, p + ".block.3.alpha", d, enc[bi].a_down);
        load_wn_conv(b, p + ".block.4", enc[bi].down, d, 2 * d, 2 * s, s, (s + 1) / 2, 1, false, NC_KC_CONV_DOWN);
        d *= 2;
    }
    snprintf(nm, sizeof nm, "encoder.block.%d.alpha", cfg.n_encoder_rates + 1);
    load_alpha(b, nm, d, enc_alpha_out);
    snprintf(nm, sizeof nm, "encoder.block.%d", cfg.n_encoder_rates + 2);
    load_wn_conv(b, nm, enc_out, d, latent, 3, 1, 1, 1, false, NC_KC_CONV_MISC);

    in_proj.clear(); out_proj.clear(); codebooks.clear();
    for (int i = 0; i < cfg.n_codebooks; ++i) {
        snprintf(nm, sizeof nm, "quantizer.quantizers.%d", i);
        const std::string p = nm;
        in_proj.emplace_back(new ConvLayer());
        out_proj.emplace_back(new ConvLayer());
        codebooks.emplace_back(new Codebook());
        load_wn_conv(b, p + ".in_proj", *in_proj.back(), latent, cfg.codebook_dim, 1, 1, 0, 1, false, NC_KC_CONV_K1);
        load_wn_conv(b, p + ".out_proj", *out_proj.back(), cfg.codebook_dim, latent, 1, 1, 0, 1, false, NC_KC_CONV_K1);
        const BlobTensor& cb = b.get(p + ".codebook.weight");
        if (cb.dims.size() != 2 || cb.dims[0] != cfg.codebook_size || cb.dims[1] != cfg.codebook_dim)
            fail(NC_EINVAL, "%s.codebook.weight has the wrong shape", p.c_str());
        codebooks.back()->build(static_cast<const float*>(cb.data), cfg.codebook_size, cfg.codebook_dim);
    }
    {   // dense projection weights of every stage for the stage-fused quantizer (same folded values as the packed images)
        const int D = cfg.codebook_dim;
        rvq_dense.clear();
        std::vector<RvqStage> tab((size_t)cfg.n_codebooks);
        auto dense = [&](const std::string& prefix, int64_t d0, int64_t inner, bool transpose, const float** w_dev, const float** b_dev, int64_t nb) {
            const BlobTensor& v = b.get(prefix + ".weight_v");
            const BlobTensor& g = b.get(prefix + ".weight_g");
            const BlobTensor* bias = b.find(prefix + ".bias");
            std::vector<float> w((size_t)(d0 * inner)), wt;
            fold_weight_norm_dac(static_cast<const float*>(v.data), static_cast<const float*>(g.data), d0, inner, w.data());
            if (transpose) {
                wt.resize(w.size());
                for (int64_t r = 0; r < d0; ++r)
                    for (int64_t c = 0; c < inner; ++c) wt[(size_t)(c * d0 + r)] = w[(size_t)(r * inner + c)];
            }
            rvq_dense.emplace_back(new DevBuf());
            upload(*rvq_dense.back(), transpose ? wt.data() : w.data(), w.size());
            *w_dev = rvq_dense.back()->as<float>();
            std::vector<float> bz((size_t)nb, 0.0f);
            if (bias) std::memcpy(bz.data(), bias->data, (size_t)nb * sizeof(float));
            rvq_dense.emplace_back(new DevBuf());
            upload(*rvq_dense.back(), bz.data(), bz.size());
            *b_dev = rvq_dense.back()->as<float>();
        };
        for (int i = 0; i < cfg.n_codebooks; ++i) {
            snprintf(nm, sizeof nm, "quantizer.quantizers.%d", i);
            const std::string p = nm;
            RvqStage& t = tab[(size_t)i];
            dense(p + ".in_proj", D, latent, true, &t.w_inT, &t.b_in, D);          // [D][latent] -> [latent][D]
            dense(p + ".out_proj", latent, D, false, &t.w_out, &t.b_out, latent);  // [latent][D]
            t.cbT = codebooks[(size_t)i]->cbT.as<float>();
            t.c2 = codebooks[(size_t)i]->c2.as<float>();
            t.cb = codebooks[(size_t)i]->cb.as<float>();
        }
        rvq_stages.reserve(tab.size() * sizeof(RvqStage));
        NC_HIP(hipMemcpy(rvq_stages.p, tab.data(), tab.size() * sizeof(RvqStage), hipMemcpyHostToDevice));
    }

    int ch = cfg.decoder_dim;
    load_wn_conv(b, "decoder.model.0", dec_in, latent, ch, 7, 1, 3, 1, false, NC_KC_CONV_MISC);
    int out_dim = ch;
    for (int bi = 0; bi < cfg.n_decoder_rates; ++bi) {
        const int s = cfg.decoder_rates[bi];
        const int in_dim = ch >> bi;
        out_dim = ch >> (bi + 1);
        snprintf(nm, sizeof nm, "decoder.model.%d", bi + 1);
        const std::string p = nm;
        load_alpha(b, p + ".block.0.alpha", in_dim, dec[bi].a_up);
        load_wn_conv(b, p + ".block.1", dec[bi].up, in_dim, out_dim, 2 * s, s, (s + 1) / 2, 1, true, NC_KC_CONV_UP);
        for (int u = 0; u < 3; ++u) {
            snprintf(nm, sizeof nm, "decoder.model.%d.block.%d", bi + 1, u + 2);
            const std::string q = nm;
            load_alpha(b, q + ".block.0.alpha", out_dim, dec[bi].ru[u].a1);
            load_wn_conv(b, q + ".block.1", dec[bi].ru[u].c7, out_dim, out_dim, 7, 1, 3 * kDil[u], kDil[u], false, NC_KC_CONV_K7);
            load_alpha(b, q + ".block.2.alpha", out_dim, dec[bi].ru[u].a2);
            load_wn_conv(b, q + ".block.3", dec[bi].ru[u].c1, out_dim, out_dim, 1, 1, 0, 1, false, NC_KC_CONV_K1);
        }
    }
    snprintf(nm, sizeof nm, "decoder.model.%d.alpha", cfg.n_decoder_rates + 1);
    load_alpha(b, nm, out_dim, dec_alpha_out);
    snprintf(nm, sizeof nm, "decoder.model.%d", cfg.n_decoder_rates + 2);
    load_wn_conv(b, nm, dec_out, out_dim, 1, 7, 1, 3, 1, false, NC_KC_HEAD);
    NC_HIP(hipDeviceSynchronize());
    loaded = true;
}

int64_t DacModel::decoded_len(int64_t fr) const {
    int64_t L = fr;
    for (int i = 0; i < cfg.n_decoder_rates; ++i) {
        const int s = cfg.decoder_rates[i];
        L = (L - 1) * s - 2 * ((s + 1) / 2) + 2 * s;
    }
    return L;
}

// x + conv1(snake(conv7(snake(x))));  buffers rotate through act[0..2]
// Row pitch of an activation [B][C][L] in the arena.  NC_DAC_PITCH=1: rows of 256 samples or more start on 64-byte boundaries (pitch = L
// rounded up to 16 samples), which puts DAC 44.1 kHz's 696-step layers (C = 512 / 768) on the XV-only instances of the convolution template
// (they need 64-byte aligned rows; every other shipped preset has L % 16 == 0 there already).  Built and measured in round 6 on VERDICT r5's
// request and NOT the default: on the C2 step the k = 7 class went 40.45 -> 41.20 ms with it (one box, alternating runs,
// profiles/r06_ab_dac_pitch.txt) -- the legacy instances at pitch 696 beat the XV-only ones at pitch 704 on these 3-column-tile layers.
// The pad columns are never read as data: every kernel bounds its reads by x_len.
static int64_t row_pitch(int64_t L) {
    static const bool on = env_flag("NC_DAC_PITCH");
    return (on && L >= 256) ? ((L + 15) & ~(int64_t)15) : L;
}

float* DacModel::run_res_unit(ResUnit& ru, int dil, float* cur, int C, int64_t L, int B, int& cur_idx, const float* alpha_next) {
    (void)dil;
    const int64_t P = row_pitch(L);
    const int h_idx = (cur_idx + 1) % 3, o_idx = (cur_idx + 2) % 3;
    float* h = act[h_idx].as<float>();
    float* o = act[o_idx].as<float>();
    ConvIO io{};
    io.x = cur; io.x_bstride = (int64_t)C * P; io.x_cstride = P; io.x_len = (int32_t)L; io.Tin = L;
    io.alpha_in = ru.a1.as<float>(); io.alpha_out = ru.a2.as<float>();
    // Which units run as one launch: C <= 128 (the 1x1 weights stay in LDS) and C = 256 (whole-channel 128-column tile, W1 streamed:
    // 1.79 ms against 1.88 ms in two launches).  At C = 192 the fused form only breaks even (2.00 against 2.02 ms per unit: the
    // 96-row x 256-column tiles of the two-launch form are the most efficient instances of the template, 124 TFLOP/s), so it keeps
    // the two launches unless NC_WIDE_FUSE_192=1.
    static const bool wide_192 = env_flag("NC_WIDE_FUSE_192");
    if (fuse_res_units && can_fuse_res_unit(ru.c7, ru.c1) && (C != 192 || wide_192)) {
        // one launch: y = x + W1.snake(conv7(snake(x)) + b7) + b1 ; h never reaches HBM
        io.res = cur; io.fuse_k1 = &ru.c1; io.alpha_out2 = alpha_next;
        io.y = o; io.y_bstride = (int64_t)C * P; io.y_cstride = P;
        launch_conv(ru.c7, io, B, stream, &prof);
        cur_idx = o_idx;
        return o;
    }
    io.y = h; io.y_bstride = (int64_t)C * P; io.y_cstride = P;
    launch_conv(ru.c7, io, B, stream, &prof);
    ConvIO i2{};
    i2.x = h; i2.x_bstride = (int64_t)C * P; i2.x_cstride = P; i2.x_len = (int32_t)L; i2.Tin = L;
    i2.res = cur; i2.alpha_out = alpha_next;
    i2.y = o; i2.y_bstride = (int64_t)C * P; i2.y_cstride = P;
    launch_conv(ru.c1, i2, B, stream, &prof);
    cur_idx = o_idx;
    return o;
}

void DacModel::encode_dev(const float* pcm, int B, int64_t T, int sample_rate, int n_q, int64_t* codes, float* z, float* latents) {
    if (!loaded) fail(NC_ESTATE, "weights not loaded (call nc_codec_load_weights first)");
    if (!pcm || !codes) fail(NC_EINVAL, "pcm and codes must not be null");
    if (B <= 0 || T <= 0) fail(NC_EINVAL, "B and T must be positive");
    if (sample_rate != 0 && sample_rate != cfg.sample_rate)
        fail(NC_EINVAL, "Input audio sample rate %dHz does not match model sample rate %dHz", sample_rate, cfg.sample_rate);
    if (T > (int64_t)1 << 30) fail(NC_EINVAL, "clip too long");
    use_device();
    const int nq = (n_q <= 0 || n_q > cfg.n_codebooks) ? cfg.n_codebooks : n_q;
    const int64_t Tp = padded_len(T), Tz = frames(T);
    // activation arena: the widest tensor is encoder_dim x Tp (stem output)
    int64_t maxel = 0;
    {
        int c = cfg.encoder_dim;
        int64_t L = Tp;
        maxel = (int64_t)c * row_pitch(L);
        for (int i = 0; i < cfg.n_encoder_rates; ++i) {
            c *= 2;
            L /= cfg.encoder_rates[i];
            if ((int64_t)c * row_pitch(L) > maxel) maxel = (int64_t)c * row_pitch(L);
        }
    }
    for (auto& a : act) a.reserve((size_t)B * maxel * sizeof(float));
    const int D = cfg.codebook_dim;
    resid.reserve((size_t)B * latent * Tz * 4);
    zq.reserve((size_t)B * latent * Tz * 4);
    lat.reserve((size_t)B * nq * D * Tz * 4);
    st.reserve((size_t)B * D * Tz * 4);

    int C = cfg.encoder_dim;
    int64_t L = Tp;
    int cur_idx = 0;
    float* cur = act[0].as<float>();
    {  // stem; DAC.Preprocess right zero-pad is the x_len < Tin bound
        ConvIO io{};
        io.x = pcm; io.x_bstride = T; io.x_cstride = T; io.x_len = (int32_t)T; io.Tin = Tp;
        io.y = cur; io.y_bstride = (int64_t)C * row_pitch(L); io.y_cstride = row_pitch(L);
        launch_conv(enc_stem, io, B, stream, &prof);
    }
    for (int bi = 0; bi < cfg.n_encoder_rates; ++bi) {
        // the block output is consumed only through Snake(a_down) (EncoderBlock.cs:26): fuse it into the last unit's store
        for (int u = 0; u < 3; ++u)
            cur = run_res_unit(enc[bi].ru[u], kDil[u], cur, C, L, B, cur_idx, u == 2 ? enc[bi].a_down.as<float>() : nullptr);
        const int s = cfg.encoder_rates[bi];
        const int64_t Lo = enc[bi].down.out_len(L);
        const int o_idx = (cur_idx + 1) % 3;
        float* o = act[o_idx].as<float>();
        ConvIO io{};
        io.x = cur; io.x_bstride = (int64_t)C * row_pitch(L); io.x_cstride = row_pitch(L); io.x_len = (int32_t)L; io.Tin = L;
        if (bi + 1 == cfg.n_encoder_rates) io.alpha_out = enc_alpha_out.as<float>();   // Encoder.cs:44 Snake, consumed by the k3 conv only
        io.y = o; io.y_bstride = (int64_t)2 * C * row_pitch(Lo); io.y_cstride = row_pitch(Lo);
        launch_conv(enc[bi].down, io, B, stream, &prof);
        (void)s;
        cur = o; cur_idx = o_idx; C *= 2; L = Lo;
    }
    if (L != Tz) fail(NC_ESTATE, "internal: encoder produced %lld frames, expected %lld", (long long)L, (long long)Tz);
    float* residual = resid.as<float>();
    {  // Snake -> conv k3 -> z, written straight into the RVQ residual buffer (residual = z.clone())
        ConvIO io{};
        io.x = cur; io.x_bstride = (int64_t)C * row_pitch(L); io.x_cstride = row_pitch(L); io.x_len = (int32_t)L; io.Tin = L;
        io.y = residual; io.y_bstride = (int64_t)latent * Tz; io.y_cstride = Tz;
        launch_conv(enc_out, io, B, stream, &prof);
    }
    // ---- residual vector quantizer (ResidualVectorQuantizer.cs:54-103)
    float* zq_d = z ? z : zq.as<float>();
    float* lat_d = latents ? latents : lat.as<float>();
    if (launch_dac_rvq_fused(rvq_stages.as<RvqStage>(), nq, latent, D, cfg.codebook_size, residual, B, Tz, codes, zq_d, lat_d, stream, &prof)) return;
    NC_HIP(hipMemsetAsync(zq_d, 0, (size_t)B * latent * Tz * 4, stream));
    for (int i = 0; i < nq; ++i) {
        ConvIO pi{};  // in_proj 1x1: residual -> z_e, stored as channel slice i of `latents`
        pi.x = residual; pi.x_bstride = (int64_t)latent * Tz; pi.x_cstride = Tz; pi.x_len = (int32_t)Tz; pi.Tin = Tz;
        pi.y = lat_d + (int64_t)i * D * Tz; pi.y_bstride = (int64_t)nq * D * Tz; pi.y_cstride = Tz;
        launch_conv(*in_proj[i], pi, B, stream, &prof);
        launch_vq_argmin(*codebooks[i], lat_d + (int64_t)i * D * Tz, (int64_t)nq * D * Tz, B, Tz, codes + (int64_t)i * Tz,
                         (int64_t)nq * Tz, st.as<float>(), stream, &prof);
        ConvIO po{};  // out_proj 1x1 with fused zq += q ; residual -= q
        po.x = st.as<float>(); po.x_bstride = (int64_t)D * Tz; po.x_cstride = Tz; po.x_len = (int32_t)Tz; po.Tin = Tz;
        po.y = zq_d; po.y_bstride = (int64_t)latent * Tz; po.y_cstride = Tz;
        po.rvq_zq = zq_d; po.rvq_res = residual; po.epi = EPI_RVQ;
        launch_conv(*out_proj[i], po, B, stream, &prof);
    }
}

void DacModel::from_codes_dev(const int64_t* codes, int B, int n_q, int64_t Tz, float* z) {
    if (!loaded) fail(NC_ESTATE, "weights not loaded (call nc_codec_load_weights first)");
    if (!codes || !z) fail(NC_EINVAL, "codes and z must not be null");
    if (B <= 0 || Tz <= 0 || n_q <= 0 || n_q > cfg.n_codebooks) fail(NC_EINVAL, "bad codes shape [%d,%d,%lld]", B, n_q, (long long)Tz);
    use_device();
    const int D = cfg.codebook_dim;
    st.reserve((size_t)B * D * Tz * 4);
    NC_HIP(hipMemsetAsync(z, 0, (size_t)B * latent * Tz * 4, stream));
    for (int i = 0; i < n_q; ++i) {
        launch_vq_gather(*codebooks[i], codes + (int64_t)i * Tz, (int64_t)n_q * Tz, B, Tz, st.as<float>(), stream, &prof);
        ConvIO po{};
        po.x = st.as<float>(); po.x_bstride = (int64_t)D * Tz; po.x_cstride = Tz; po.x_len = (int32_t)Tz; po.Tin = Tz;
        po.y = z; po.y_bstride = (int64_t)latent * Tz; po.y_cstride = Tz;
        po.rvq_zq = z; po.rvq_res = nullptr; po.epi = EPI_RVQ;
        launch_conv(*out_proj[i], po, B, stream, &prof);
    }
}

// Dia <-> DAC glue on the device (Models/Dia.cs:973-1002): Dia keeps codes as [T, n_q] matrices, DAC as [B, n_q, T]
__global__ void code_matrix_transpose_kernel(const int64_t* __restrict__ src, int64_t* __restrict__ dst, int B, int R, int C) {
    // src [B][R][C] -> dst [B][C][R]
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * R * C) return;
    const int64_t c = i % C, r = (i / C) % R, b = i / ((int64_t)R * C);
    dst[(b * C + c) * R + r] = src[i];
}

// Dia.Decode(audioCodes[T, n_q]) batched: FromCodes(codes.transpose(1, 2)) -> Decode -> [B, T*hop]   (Dia.cs:973-981)
void DacModel::decode_code_matrix_dev(const int64_t* codes_tq, int B, int64_t Tz, int n_q, float* pcm) {
    if (!codes_tq || !pcm) fail(NC_EINVAL, "codes and pcm must not be null");
    if (B <= 0 || Tz <= 0 || n_q <= 0 || n_q > cfg.n_codebooks) fail(NC_EINVAL, "bad code matrix shape [%d,%lld,%d]", B, (long long)Tz, n_q);
    use_device();
    codes_ws.reserve((size_t)B * n_q * Tz * 8);
    zq.reserve((size_t)B * latent * Tz * 4);
    const int64_t n = (int64_t)B * Tz * n_q;
    hipLaunchKernelGGL(code_matrix_transpose_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, codes_tq, codes_ws.as<int64_t>(), B, (int)Tz, n_q);
    NC_HIP(hipGetLastError());
    from_codes_dev(codes_ws.as<int64_t>(), B, n_q, Tz, zq.as<float>());
    decode_dev(zq.as<float>(), B, Tz, pcm);
}

// Dia.Encode(audio[1, T]) batched: Encode -> codes.squeeze(0).transpose(0, 1) -> [B, T', n_q]   (Dia.cs:989-1002)
void DacModel::encode_code_matrix_dev(const float* pcm, int B, int64_t T, int sample_rate, int64_t* codes_tq) {
    if (!pcm || !codes_tq) fail(NC_EINVAL, "pcm and codes must not be null");
    if (B <= 0 || T <= 0) fail(NC_EINVAL, "B and T must be positive");
    use_device();
    const int nq = cfg.n_codebooks;
    const int64_t Tz = frames(T);
    codes_ws.reserve((size_t)B * nq * Tz * 8);
    encode_dev(pcm, B, T, sample_rate, 0, codes_ws.as<int64_t>(), nullptr, nullptr);
    const int64_t n = (int64_t)B * Tz * nq;
    hipLaunchKernelGGL(code_matrix_transpose_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, codes_ws.as<int64_t>(), codes_tq, B, nq, (int)Tz);
    NC_HIP(hipGetLastError());
}

void DacModel::decode_dev(const float* z, int B, int64_t Tz, float* pcm) {
    if (!loaded) fail(NC_ESTATE, "weights not loaded (call nc_codec_load_weights first)");
    if (!z || !pcm) fail(NC_EINVAL, "z and pcm must not be null");
    if (B <= 0 || Tz <= 0) fail(NC_EINVAL, "B and frames must be positive");
    use_device();
    int64_t maxel = (int64_t)cfg.decoder_dim * row_pitch(Tz);
    {
        int64_t L = Tz;
        for (int i = 0; i < cfg.n_decoder_rates; ++i) {
            L = dec[i].up.out_len(L);
            const int64_t c = cfg.decoder_dim >> (i + 1);
            if (c * row_pitch(L) > maxel) maxel = c * row_pitch(L);
        }
    }
    for (auto& a : act) a.reserve((size_t)B * maxel * sizeof(float));
    int C = cfg.decoder_dim;
    int64_t L = Tz;
    int cur_idx = 0;
    float* cur = act[0].as<float>();
    {
        ConvIO io{};
        io.x = z; io.x_bstride = (int64_t)latent * Tz; io.x_cstride = Tz; io.x_len = (int32_t)Tz; io.Tin = Tz;
        io.alpha_out = dec[0].a_up.as<float>();   // consumed only through the first DecoderBlock's Snake (DecoderBlock.cs:24)
        io.y = cur; io.y_bstride = (int64_t)C * row_pitch(L); io.y_cstride = row_pitch(L);
        launch_conv(dec_in, io, B, stream, &prof);
    }
    for (int bi = 0; bi < cfg.n_decoder_rates; ++bi) {
        const int Co = C / 2;
        const int64_t Lo = dec[bi].up.out_len(L);
        const int o_idx = (cur_idx + 1) % 3;
        float* o = act[o_idx].as<float>();
        ConvIO io{};
        io.x = cur; io.x_bstride = (int64_t)C * row_pitch(L); io.x_cstride = row_pitch(L); io.x_len = (int32_t)L; io.Tin = L;
        io.y = o; io.y_bstride = (int64_t)Co * row_pitch(Lo); io.y_cstride = row_pitch(Lo);
        launch_conv(dec[bi].up, io, B, stream, &prof);
        cur = o; cur_idx = o_idx; C = Co; L = Lo;
        const float* a_next = bi + 1 < cfg.n_decoder_rates ? dec[bi + 1].a_up.as<float>() : dec_alpha_out.as<float>();
        for (int u = 0; u < 3; ++u) cur = run_res_unit(dec[bi].ru[u], kDil[u], cur, C, L, B, cur_idx, u == 2 ? a_next : nullptr);
    }
    {
        ConvIO io{};
        io.x = cur; io.x_bstride = (int64_t)C * row_pitch(L); io.x_cstride = row_pitch(L); io.x_len = (int32_t)L; io.Tin = L;
        io.y = pcm; io.y_bstride = L; io.y_cstride = L;
        io.epi = EPI_TANH;
        launch_conv(dec_out, io, B, stream, &prof);
    }
}

}  // namespace nc
