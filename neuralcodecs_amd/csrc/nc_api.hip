// C ABI (include/nc_mi355x.h): argument validation, exception -> status translation, host-buffer variants.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <fstream>

#include "nc_model.h"

using namespace nc;


namespace {

template <class F>
nc_status guard(F&& f) {
    try {
        f();
        return NC_OK;
    } catch (const Error& e) {
        set_last_error(e.what());
        return e.code;
    } catch (const std::bad_alloc&) {
        set_last_error("host allocation failed");
        return NC_ENOMEM;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return NC_ESTATE;
    }
}

DacModel& as_dac(nc_codec* h) {
    if (!h || !h->impl) fail(NC_EINVAL, "null codec handle");
    if (h->kind != 0) fail(NC_EINVAL, "handle is not a DAC codec");
    return static_cast<DacModel&>(*h->impl);
}

SnacModel& as_snac(nc_codec* h) {
    if (!h || !h->impl) fail(NC_EINVAL, "null codec handle");
    if (h->kind != 1) fail(NC_EINVAL, "handle is not a SNAC codec");
    return static_cast<SnacModel&>(*h->impl);
}

EncodecModel& as_encodec(nc_codec* h) {
    if (!h || !h->impl) fail(NC_EINVAL, "null codec handle");
    if (h->kind != 2) fail(NC_EINVAL, "handle is not an Encodec codec");
    return static_cast<EncodecModel&>(*h->impl);
}

void h2d(void* d, const void* h, size_t n, hipStream_t s) { NC_HIP(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s)); }
void d2h(void* h, const void* d, size_t n, hipStream_t s) { NC_HIP(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s)); }

}  // namespace

extern "C" {

const char* nc_last_error(void) { return get_last_error(); }
#ifdef NC_EXPERIMENTS
const char* nc_version(void) { return "nc_mi355x 0.1 (gfx950) +experiments"; }   // make EXPERIMENTS=1: measured-and-rejected kernels and their switches
#else
const char* nc_version(void) { return "nc_mi355x 0.1 (gfx950)"; }
#endif

const char* nc_debug_switches(void) { return env_switch_table(); }

int nc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

nc_status nc_dac_create(const nc_dac_config* cfg, int device_index, nc_codec** out) {
    return guard([&] {
        if (!cfg || !out) fail(NC_EINVAL, "cfg and out must not be null");
        *out = nullptr;
        std::unique_ptr<DacModel> m(new DacModel(*cfg));
        m->init_device(device_index);
        nc_codec* h = new nc_codec();
        h->impl = std::move(m);
        h->kind = 0;
        *out = h;
    });
}

nc_status nc_codec_destroy(nc_codec* h) {
    return guard([&] {
        if (!h) return;
        if (h->impl) {
            h->impl->use_device();
            (void)hipStreamSynchronize(h->impl->stream);
        }
        delete h;
    });
}

nc_status nc_codec_load_weights_mem(nc_codec* h, const void* blob, size_t nbytes) {
    return guard([&] {
        if (!h || !h->impl) fail(NC_EINVAL, "null codec handle");
        if (!blob) fail(NC_EINVAL, "blob must not be null");
        Blob b;
        b.parse(blob, nbytes);
        h->impl->load(b);
    });
}

nc_status nc_blob_check(const void* blob, size_t nbytes, int32_t* n_tensors) {
    return guard([&] {
        if (!blob) fail(NC_EINVAL, "blob must not be null");
        Blob b;
        b.parse(blob, nbytes);
        if (n_tensors) *n_tensors = (int32_t)b.tensors.size();
    });
}

nc_status nc_codec_load_weights(nc_codec* h, const char* path) {
    return guard([&] {
        if (!h || !h->impl) fail(NC_EINVAL, "null codec handle");
        if (!path) fail(NC_EINVAL, "path must not be null");
        std::ifstream f(path, std::ios::binary | std::ios::ate);
        if (!f) fail(NC_ENOTFOUND, "Weights not found at %s", path);  // DAC.cs:347-350 FileNotFoundException
        const std::streamsize n = f.tellg();
        f.seekg(0);
        std::vector<char> buf((size_t)n);
        if (!f.read(buf.data(), n)) fail(NC_ESTATE, "Failed to read weights from %s", path);
        Blob b;
        b.parse(buf.data(), buf.size());
        h->impl->load(b);
    });
}

nc_status nc_codec_set_stream(nc_codec* h, void* hip_stream) {
    return guard([&] {
        if (!h || !h->impl) fail(NC_EINVAL, "null codec handle");
        h->impl->switch_stream(static_cast<hipStream_t>(hip_stream));
    });
}

nc_status nc_codec_reset_stream(nc_codec* h) {
    return guard([&] {
        if (!h || !h->impl) fail(NC_EINVAL, "null codec handle");
        h->impl->switch_stream(h->impl->own_stream);
    });
}

nc_status nc_codec_synchronize(nc_codec* h) {
    return guard([&] {
        if (!h || !h->impl) fail(NC_EINVAL, "null codec handle");
        h->impl->use_device();
        NC_HIP(hipStreamSynchronize(h->impl->stream));
        h->impl->check_async_errors();
    });
}

nc_status nc_codec_check_errors(nc_codec* h) {
    return guard([&] {
        if (!h || !h->impl) fail(NC_EINVAL, "null codec handle");
        h->impl->check_async_errors();
    });
}

nc_status nc_encodec_lstm_stats(const nc_codec* h, int32_t* stepwise, int64_t* timeouts) {
    return guard([&] {
        EncodecModel& m = as_encodec(const_cast<nc_codec*>(h));
        if (stepwise) *stepwise = m.lstm_force_stepwise ? 1 : 0;
        if (timeouts) *timeouts = m.lstm_timeouts;
    });
}

nc_status nc_dac_query(const nc_codec* h, int64_t T, int64_t* T_padded, int64_t* frames) {
    return guard([&] {
        DacModel& m = as_dac(const_cast<nc_codec*>(h));
        if (T <= 0) fail(NC_EINVAL, "T must be positive");
        if (T_padded) *T_padded = m.padded_len(T);
        if (frames) *frames = m.frames(T);
    });
}

nc_status nc_dac_encode_dev(nc_codec* h, const float* pcm, int32_t B, int64_t T, int32_t sample_rate, int32_t n_q, int64_t* codes,
                            float* z, float* latents) {
    return guard([&] { as_dac(h).encode_dev(pcm, B, T, sample_rate, n_q, codes, z, latents); });
}

nc_status nc_dac_decode_dev(nc_codec* h, const float* z, int32_t B, int64_t frames, float* pcm) {
    return guard([&] { as_dac(h).decode_dev(z, B, frames, pcm); });
}

nc_status nc_dac_from_codes_dev(nc_codec* h, const int64_t* codes, int32_t B, int32_t n_q, int64_t frames, float* z) {
    return guard([&] { as_dac(h).from_codes_dev(codes, B, n_q, frames, z); });
}

nc_status nc_dac_encode(nc_codec* h, const float* pcm, int32_t B, int64_t T, int32_t sample_rate, int32_t n_q, int64_t* codes,
                        float* z, float* latents) {
    return guard([&] {
        DacModel& m = as_dac(h);
        if (!pcm || !codes) fail(NC_EINVAL, "pcm and codes must not be null");
        if (B <= 0 || T <= 0) fail(NC_EINVAL, "B and T must be positive");
        m.use_device();
        OwnStreamScope own(m);
        const int nq = (n_q <= 0 || n_q > m.cfg.n_codebooks) ? m.cfg.n_codebooks : n_q;
        const int64_t Tz = m.frames(T);
        const size_t n_in = (size_t)B * T * 4, n_codes = (size_t)B * nq * Tz * 8, n_z = (size_t)B * m.latent * Tz * 4,
                     n_lat = (size_t)B * nq * m.cfg.codebook_dim * Tz * 4;
        m.h_in.reserve(n_in); m.h_codes.reserve(n_codes); m.h_aux0.reserve(n_z); m.h_aux1.reserve(n_lat);
        h2d(m.h_in.p, pcm, n_in, m.stream);
        m.encode_dev(m.h_in.as<float>(), B, T, sample_rate, n_q, m.h_codes.as<int64_t>(), m.h_aux0.as<float>(), m.h_aux1.as<float>());
        d2h(codes, m.h_codes.p, n_codes, m.stream);
        if (z) d2h(z, m.h_aux0.p, n_z, m.stream);
        if (latents) d2h(latents, m.h_aux1.p, n_lat, m.stream);
        NC_HIP(hipStreamSynchronize(m.stream));
    });
}

nc_status nc_dac_decode(nc_codec* h, const float* z, int32_t B, int64_t frames, float* pcm) {
    return guard([&] {
        DacModel& m = as_dac(h);
        if (!z || !pcm) fail(NC_EINVAL, "z and pcm must not be null");
        if (B <= 0 || frames <= 0) fail(NC_EINVAL, "B and frames must be positive");
        m.use_device();
        OwnStreamScope own(m);
        const size_t n_z = (size_t)B * m.latent * frames * 4, n_out = (size_t)B * m.decoded_len(frames) * 4;
        m.h_aux0.reserve(n_z); m.h_out.reserve(n_out);
        h2d(m.h_aux0.p, z, n_z, m.stream);
        m.decode_dev(m.h_aux0.as<float>(), B, frames, m.h_out.as<float>());
        d2h(pcm, m.h_out.p, n_out, m.stream);
        NC_HIP(hipStreamSynchronize(m.stream));
    });
}

nc_status nc_dac_from_codes(nc_codec* h, const int64_t* codes, int32_t B, int32_t n_q, int64_t frames, float* z) {
    return guard([&] {
        DacModel& m = as_dac(h);
        if (!codes || !z) fail(NC_EINVAL, "codes and z must not be null");
        if (B <= 0 || frames <= 0 || n_q <= 0) fail(NC_EINVAL, "bad codes shape");
        m.use_device();
        OwnStreamScope own(m);
        const size_t n_codes = (size_t)B * n_q * frames * 8, n_z = (size_t)B * m.latent * frames * 4;
        m.h_codes.reserve(n_codes); m.h_aux0.reserve(n_z);
        h2d(m.h_codes.p, codes, n_codes, m.stream);
        m.from_codes_dev(m.h_codes.as<int64_t>(), B, n_q, frames, m.h_aux0.as<float>());
        d2h(z, m.h_aux0.p, n_z, m.stream);
        NC_HIP(hipStreamSynchronize(m.stream));
    });
}

nc_status nc_dac_decode_code_matrix_dev(nc_codec* h, const int64_t* codes_tq, int32_t B, int64_t frames, int32_t n_q, float* pcm) {
    return guard([&] { as_dac(h).decode_code_matrix_dev(codes_tq, B, frames, n_q, pcm); });
}
nc_status nc_dac_encode_code_matrix_dev(nc_codec* h, const float* pcm, int32_t B, int64_t T, int32_t sample_rate, int64_t* codes_tq) {
    return guard([&] { as_dac(h).encode_code_matrix_dev(pcm, B, T, sample_rate, codes_tq); });
}
nc_status nc_dac_decode_code_matrix(nc_codec* h, const int64_t* codes_tq, int32_t B, int64_t frames, int32_t n_q, float* pcm) {
    return guard([&] {
        DacModel& m = as_dac(h);
        if (!codes_tq || !pcm) fail(NC_EINVAL, "codes and pcm must not be null");
        if (B <= 0 || frames <= 0 || n_q <= 0) fail(NC_EINVAL, "bad code matrix shape");
        m.use_device();
        OwnStreamScope own(m);
        const size_t n_codes = (size_t)B * n_q * frames * 8, n_out = (size_t)B * m.decoded_len(frames) * 4;
        m.h_codes.reserve(n_codes); m.h_out.reserve(n_out);
        h2d(m.h_codes.p, codes_tq, n_codes, m.stream);
        m.decode_code_matrix_dev(m.h_codes.as<int64_t>(), B, frames, n_q, m.h_out.as<float>());
        d2h(pcm, m.h_out.p, n_out, m.stream);
        NC_HIP(hipStreamSynchronize(m.stream));
    });
}
nc_status nc_dac_encode_code_matrix(nc_codec* h, const float* pcm, int32_t B, int64_t T, int32_t sample_rate, int64_t* codes_tq) {
    return guard([&] {
        DacModel& m = as_dac(h);
        if (!pcm || !codes_tq) fail(NC_EINVAL, "pcm and codes must not be null");
        if (B <= 0 || T <= 0) fail(NC_EINVAL, "B and T must be positive");
        m.use_device();
        OwnStreamScope own(m);
        const size_t n_in = (size_t)B * T * 4, n_codes = (size_t)B * m.cfg.n_codebooks * m.frames(T) * 8;
        m.h_in.reserve(n_in); m.h_codes.reserve(n_codes);
        h2d(m.h_in.p, pcm, n_in, m.stream);
        m.encode_code_matrix_dev(m.h_in.as<float>(), B, T, sample_rate, m.h_codes.as<int64_t>());
        d2h(codes_tq, m.h_codes.p, n_codes, m.stream);
        NC_HIP(hipStreamSynchronize(m.stream));
    });
}

// ---- SNAC ------------------------------------------------------------------------------------------
nc_status nc_snac_create(const nc_snac_config* cfg, int device_index, nc_codec** out) {
    return guard([&] {
        if (!cfg || !out) fail(NC_EINVAL, "cfg and out must not be null");
        *out = nullptr;
        std::unique_ptr<SnacModel> m(new SnacModel(*cfg));
        m->init_device(device_index);
        nc_codec* h = new nc_codec();
        h->impl = std::move(m);
        h->kind = 1;
        *out = h;
    });
}

nc_status nc_snac_query(const nc_codec* h, int64_t T, int64_t* T_padded, int64_t* frames, int32_t* n_levels, int64_t* level_widths,
                        int64_t* decoded_len) {
    return guard([&] {
        SnacModel& m = as_snac(const_cast<nc_codec*>(h));
        if (T <= 0) fail(NC_EINVAL, "T must be positive");
        const int64_t Tp = m.padded_len(T), Tz = Tp / m.hop;
        if (T_padded) *T_padded = Tp;
        if (frames) *frames = Tz;
        if (n_levels) *n_levels = m.cfg.n_vq_strides;
        if (level_widths)
            for (int i = 0; i < m.cfg.n_vq_strides; ++i) level_widths[i] = Tz / m.cfg.vq_strides[i];
        if (decoded_len) *decoded_len = m.decoded_len(Tz);
    });
}

nc_status nc_snac_noise_len(const nc_codec* h, int32_t B, int64_t frames, int64_t* n) {
    return guard([&] {
        SnacModel& m = as_snac(const_cast<nc_codec*>(h));
        if (!n || B <= 0 || frames <= 0) fail(NC_EINVAL, "bad arguments");
        *n = m.noise_len(B, frames);
    });
}

nc_status nc_snac_encode_dev(nc_codec* h, const float* pcm, int32_t B, int64_t T, int64_t* codes, float* z, float* zq) {
    return guard([&] { as_snac(h).encode_dev(pcm, B, T, codes, z, zq); });
}
nc_status nc_snac_encode_tensor_dev(nc_codec* h, const float* pcm, int32_t B, int64_t T, int64_t* codes, float* z, float* zq) {
    return guard([&] { as_snac(h).encode_dev(pcm, B, T, codes, z, zq, false); });
}
nc_status nc_snac_query_tensor(const nc_codec* h, int64_t T, int64_t* frames, int32_t* n_levels, int64_t* level_widths) {
    return guard([&] {
        SnacModel& m = as_snac(const_cast<nc_codec*>(h));
        if (T <= 0) fail(NC_EINVAL, "T must be positive");
        const int64_t Tz = m.unpadded_frames(T);
        if (frames) *frames = Tz;
        if (n_levels) *n_levels = m.cfg.n_vq_strides;
        if (level_widths)
            for (int i = 0; i < m.cfg.n_vq_strides; ++i) level_widths[i] = Tz / m.cfg.vq_strides[i];
    });
}
nc_status nc_snac_from_codes_dev(nc_codec* h, const int64_t* codes, int32_t B, int64_t frames, float* zq) {
    return guard([&] { as_snac(h).from_codes_dev(codes, B, frames, zq); });
}
nc_status nc_snac_decode_dev(nc_codec* h, const int64_t* codes, int32_t B, int64_t frames, const float* noise, uint64_t seed,
                             float* pcm) {
    return guard([&] { as_snac(h).decode_dev(codes, B, frames, noise, seed, pcm); });
}

static void snac_encode_host(nc_codec* h, const float* pcm, int32_t B, int64_t T, int64_t* codes, float* z, float* zq, bool pad) {
    SnacModel& m = as_snac(h);
    if (!pcm || !codes) fail(NC_EINVAL, "pcm and codes must not be null");
    if (B <= 0 || T <= 0) fail(NC_EINVAL, "B and T must be positive");
    m.use_device();
    OwnStreamScope own(m);
    const int64_t Tz = pad ? m.padded_len(T) / m.hop : m.unpadded_frames(T);
    const size_t n_in = (size_t)B * T * 4, n_codes = (size_t)B * m.codes_per_clip(Tz) * 8, n_z = (size_t)B * m.latent * Tz * 4;
    m.h_in.reserve(n_in); m.h_codes.reserve(n_codes); m.h_aux0.reserve(n_z); m.h_aux1.reserve(n_z);
    h2d(m.h_in.p, pcm, n_in, m.stream);
    m.encode_dev(m.h_in.as<float>(), B, T, m.h_codes.as<int64_t>(), m.h_aux0.as<float>(), m.h_aux1.as<float>(), pad);
    d2h(codes, m.h_codes.p, n_codes, m.stream);
    if (z) d2h(z, m.h_aux0.p, n_z, m.stream);
    if (zq) d2h(zq, m.h_aux1.p, n_z, m.stream);
    NC_HIP(hipStreamSynchronize(m.stream));
}

nc_status nc_snac_encode(nc_codec* h, const float* pcm, int32_t B, int64_t T, int64_t* codes, float* z, float* zq) {
    return guard([&] { snac_encode_host(h, pcm, B, T, codes, z, zq, true); });
}
nc_status nc_snac_encode_tensor(nc_codec* h, const float* pcm, int32_t B, int64_t T, int64_t* codes, float* z, float* zq) {
    return guard([&] { snac_encode_host(h, pcm, B, T, codes, z, zq, false); });
}

nc_status nc_snac_from_codes(nc_codec* h, const int64_t* codes, int32_t B, int64_t frames, float* zq) {
    return guard([&] {
        SnacModel& m = as_snac(h);
        if (!codes || !zq) fail(NC_EINVAL, "codes and zq must not be null");
        if (B <= 0 || frames <= 0) fail(NC_EINVAL, "B and frames must be positive");
        m.use_device();
        OwnStreamScope own(m);
        const size_t n_codes = (size_t)B * m.codes_per_clip(frames) * 8, n_z = (size_t)B * m.latent * frames * 4;
        m.h_codes.reserve(n_codes); m.h_aux0.reserve(n_z);
        h2d(m.h_codes.p, codes, n_codes, m.stream);
        m.from_codes_dev(m.h_codes.as<int64_t>(), B, frames, m.h_aux0.as<float>());
        d2h(zq, m.h_aux0.p, n_z, m.stream);
        NC_HIP(hipStreamSynchronize(m.stream));
    });
}

nc_status nc_snac_decode(nc_codec* h, const int64_t* codes, int32_t B, int64_t frames, const float* noise, uint64_t seed,
                         float* pcm) {
    return guard([&] {
        SnacModel& m = as_snac(h);
        if (!codes || !pcm) fail(NC_EINVAL, "codes and pcm must not be null");   // ArgumentNullException, SNAC.cs:175
        if (B <= 0 || frames <= 0) fail(NC_EINVAL, "Codes list cannot be empty");   // ArgumentException, SNAC.cs:177-180
        m.use_device();
        OwnStreamScope own(m);
        const size_t n_codes = (size_t)B * m.codes_per_clip(frames) * 8, n_out = (size_t)B * m.decoded_len(frames) * 4;
        const size_t n_noise = (size_t)m.noise_len(B, frames) * 4;
        m.h_codes.reserve(n_codes); m.h_out.reserve(n_out);
        h2d(m.h_codes.p, codes, n_codes, m.stream);
        const float* nz = nullptr;
        if (noise && n_noise) {
            m.h_noise.reserve(n_noise);
            h2d(m.h_noise.p, noise, n_noise, m.stream);
            nz = m.h_noise.as<float>();
        }
        m.decode_dev(m.h_codes.as<int64_t>(), B, frames, nz, seed, m.h_out.as<float>());
        d2h(pcm, m.h_out.p, n_out, m.stream);
        NC_HIP(hipStreamSynchronize(m.stream));
    });
}

// SNAC.ProcessAudio (Models/SNAC.cs:255-282): resample (SNAC.cs:284-308) -> forward (:91-106) with the clip resident in HBM throughout.
nc_status nc_snac_process_audio_len(const nc_codec* h, int64_t n, int32_t sample_rate, int64_t* n_out) {
    return guard([&] {
        SnacModel& m = as_snac(const_cast<nc_codec*>(h));
        if (!n_out) fail(NC_EINVAL, "n_out must not be null");
        if (n <= 0) fail(NC_EINVAL, "Audio data cannot be empty");
        if (sample_rate <= 0) fail(NC_EINVAL, "sample rate must be positive");
        *n_out = sample_rate == m.cfg.sample_rate ? n : nc_audio_resample_len(n, sample_rate, m.cfg.sample_rate);
    });
}

nc_status nc_snac_process_audio(nc_codec* h, const float* audio, int64_t n, int32_t sample_rate, const float* noise, uint64_t seed,
                                float* out) {
    return guard([&] {
        SnacModel& m = as_snac(h);
        if (!audio || n <= 0) fail(NC_EINVAL, "Audio data cannot be empty");   // ArgumentException, SNAC.cs:257-258
        if (!out) fail(NC_EINVAL, "out must not be null");
        if (sample_rate <= 0) fail(NC_EINVAL, "sample rate must be positive");
        m.use_device();
        OwnStreamScope own(m);
        const bool resample = sample_rate != m.cfg.sample_rate;
        const int64_t T = resample ? nc_audio_resample_len(n, sample_rate, m.cfg.sample_rate) : n;
        if (T <= 0) fail(NC_EINVAL, "resampled clip would be empty");
        const int64_t Tz = m.padded_len(T) / m.hop;
        const size_t n_codes = (size_t)m.codes_per_clip(Tz) * 8, n_dec = (size_t)m.decoded_len(Tz) * 4, n_z = (size_t)m.latent * Tz * 4;
        const size_t n_noise = (size_t)m.noise_len(1, Tz) * 4;
        m.h_in.reserve((size_t)T * 4); m.h_codes.reserve(n_codes); m.h_out.reserve(n_dec); m.h_aux0.reserve(std::max((size_t)n * 4, n_z));
        m.h_aux1.reserve(n_z);
        if (resample) {
            h2d(m.h_aux0.p, audio, (size_t)n * 4, m.stream);
            const nc_status st = nc_audio_resample_linear_dev(m.device, m.h_aux0.as<float>(), 1, n, sample_rate, m.cfg.sample_rate,
                                                              m.h_in.as<float>(), m.stream);
            if (st != NC_OK) fail(st, "%s", get_last_error());
        } else {
            h2d(m.h_in.p, audio, (size_t)n * 4, m.stream);
        }
        const float* nz = nullptr;
        if (noise && n_noise) {
            m.h_noise.reserve(n_noise);
            h2d(m.h_noise.p, noise, n_noise, m.stream);
            nz = m.h_noise.as<float>();
        }
        m.encode_dev(m.h_in.as<float>(), 1, T, m.h_codes.as<int64_t>(), m.h_aux0.as<float>(), m.h_aux1.as<float>(), true);
        m.decode_dev(m.h_codes.as<int64_t>(), 1, Tz, nz, seed, m.h_out.as<float>());
        d2h(out, m.h_out.p, (size_t)T * 4, m.stream);                          // SNAC.cs:103 narrow(-1, 0, length)
        NC_HIP(hipStreamSynchronize(m.stream));
    });
}

// ---- Encodec ---------------------------------------------------------------------------------------
nc_status nc_encodec_create(const nc_encodec_config* cfg, int device_index, nc_codec** out) {
    return guard([&] {
        if (!cfg || !out) fail(NC_EINVAL, "cfg and out must not be null");
        *out = nullptr;
        std::unique_ptr<EncodecModel> m(new EncodecModel(*cfg));
        m->init_device(device_index);
        nc_codec* h = new nc_codec();
        h->impl = std::move(m);
        h->kind = 2;
        *out = h;
    });
}

nc_status nc_encodec_set_bandwidth(nc_codec* h, float bw) {
    return guard([&] { as_encodec(h).set_bandwidth(bw); });
}

nc_status nc_encodec_query(const nc_codec* h, int64_t T, int32_t* n_frames, int32_t* n_q, int64_t* frame_lens, int32_t cap,
                           int64_t* decoded_len) {
    return guard([&] {
        EncodecModel& m = as_encodec(const_cast<nc_codec*>(h));
        if (T <= 0) fail(NC_EINVAL, "T must be positive");
        const auto segs = m.segments(T);
        if (n_frames) *n_frames = (int32_t)segs.size();
        if (n_q) *n_q = m.n_q;
        if (frame_lens)
            for (size_t i = 0; i < segs.size() && (int32_t)i < cap; ++i) frame_lens[i] = segs[i].frames;
        if (decoded_len) {
            if (m.cfg.segment_length <= 0) *decoded_len = m.decoded_for(segs[0].frames);
            else *decoded_len = (int64_t)m.cfg.segment_stride * ((int64_t)segs.size() - 1) + m.decoded_for(segs.back().frames);
        }
    });
}

nc_status nc_encodec_clip_length(const nc_codec* h, int32_t n_frames, const int64_t* frame_lens, int64_t* T) {
    return guard([&] {
        EncodecModel& m = as_encodec(const_cast<nc_codec*>(h));
        if (!T || !frame_lens) fail(NC_EINVAL, "frame_lens and T must not be null");
        if (n_frames <= 0) fail(NC_EINVAL, "No frames provided to decode");                               // Encodec.cs:215-218
        const int64_t tail_frames = frame_lens[n_frames - 1];
        if (tail_frames <= 0) fail(NC_EINVAL, "a frame without codes");
        if (m.cfg.segment_length <= 0) {
            if (n_frames != 1) fail(NC_EINVAL, "Expected single frame when no segmentation is used");   // Encodec.cs:222-225
            for (int64_t L = std::max<int64_t>(1, (tail_frames - 2) * m.hop); L <= (tail_frames + 1) * m.hop; ++L)
                if (m.frames_for(L) == tail_frames) { *T = L; return; }
            fail(NC_EINVAL, "no clip length yields %lld frames", (long long)tail_frames);
        }
        const int64_t base = (int64_t)(n_frames - 1) * m.cfg.segment_stride;
        for (int64_t tail = 1; tail <= m.cfg.segment_stride; ++tail) {   // a longer tail would start another segment (Encodec.cs:278-282)
            if (m.frames_for(std::min<int64_t>(tail, m.cfg.segment_length)) != tail_frames) continue;
            const auto segs = m.segments(base + tail);                    // the segments before the last can be cut short by the clip end too
            bool same = (int32_t)segs.size() == n_frames;
            for (int32_t i = 0; same && i < n_frames; ++i) same = segs[(size_t)i].frames == frame_lens[i];
            if (same) { *T = base + tail; return; }
        }
        fail(NC_EINVAL, "no clip length yields this layout of %d segments (%lld frames in the last)", n_frames, (long long)tail_frames);
    });
}

nc_status nc_encodec_encode_dev(nc_codec* h, const float* pcm, int32_t B, int64_t T, int64_t* codes, float* scales, float* emb) {
    return guard([&] { as_encodec(h).encode_dev(pcm, B, T, codes, scales, emb); });
}
nc_status nc_encodec_decode_dev(nc_codec* h, const int64_t* codes, const float* scales, int32_t B, int64_t T, int32_t n_q, float* pcm) {
    return guard([&] { as_encodec(h).decode_dev(codes, scales, B, T, n_q, pcm); });
}

nc_status nc_encodec_encode(nc_codec* h, const float* pcm, int32_t B, int64_t T, int64_t* codes, float* scales, float* emb) {
    return guard([&] {
        EncodecModel& m = as_encodec(h);
        if (!pcm || !codes) fail(NC_EINVAL, "pcm and codes must not be null");                  // ArgumentNullException, Encodec.cs:245
        if (B <= 0 || T <= 0) fail(NC_EINVAL, "B and T must be positive");
        m.use_device();
        OwnStreamScope own(m);
        const auto segs = m.segments(T);
        int64_t fr = 0;
        for (auto& s : segs) fr += s.frames;
        const size_t n_in = (size_t)B * m.cfg.channels * T * 4, n_codes = (size_t)B * m.n_q * fr * 8, n_sc = segs.size() * (size_t)B * 4,
                     n_emb = (size_t)B * m.cfg.dimension * fr * 4;
        m.h_in.reserve(n_in); m.h_codes.reserve(n_codes); m.h_scales.reserve(n_sc); m.h_emb.reserve(n_emb);
        m.absorb_stale_timeout();
        h2d(m.h_in.p, pcm, n_in, m.stream);
        for (int attempt = 0;; ++attempt) {
            m.encode_dev(m.h_in.as<float>(), B, T, m.h_codes.as<int64_t>(), m.h_scales.as<float>(), emb ? m.h_emb.as<float>() : nullptr);
            NC_HIP(hipStreamSynchronize(m.stream));
            if (!m.lstm_timed_out() || attempt) break;
            try { m.check_async_errors(); } catch (const Error&) {}   // clears the word, switches the handle to the step-wise LSTM: run again
        }
        m.check_async_errors();
        d2h(codes, m.h_codes.p, n_codes, m.stream);
        if (scales && m.cfg.normalize) d2h(scales, m.h_scales.p, n_sc, m.stream);
        if (emb) d2h(emb, m.h_emb.p, n_emb, m.stream);
        NC_HIP(hipStreamSynchronize(m.stream));
    });
}

nc_status nc_encodec_decode(nc_codec* h, const int64_t* codes, const float* scales, int32_t B, int64_t T, int32_t n_q, float* pcm) {
    return guard([&] {
        EncodecModel& m = as_encodec(h);
        if (!codes || !pcm) fail(NC_EINVAL, "Invalid frame codes in Encodec Decode");              // Encodec.cs:438-442
        if (B <= 0 || T <= 0 || n_q <= 0) fail(NC_EINVAL, "No frames provided to decode");
        if (m.cfg.normalize && !scales) fail(NC_EINVAL, "this model normalises frames: scales must be given");
        m.use_device();
        OwnStreamScope own(m);
        const auto segs = m.segments(T);
        int64_t fr = 0;
        for (auto& s : segs) fr += s.frames;
        const int64_t Lout = m.cfg.segment_length <= 0 ? m.decoded_for(segs[0].frames)
                                                       : (int64_t)m.cfg.segment_stride * ((int64_t)segs.size() - 1) + m.decoded_for(segs.back().frames);
        const size_t n_codes = (size_t)B * n_q * fr * 8, n_sc = segs.size() * (size_t)B * 4, n_out = (size_t)B * m.cfg.channels * Lout * 4;
        m.h_codes.reserve(n_codes); m.h_scales.reserve(n_sc); m.h_out.reserve(n_out);
        m.absorb_stale_timeout();
        h2d(m.h_codes.p, codes, n_codes, m.stream);
        if (scales) h2d(m.h_scales.p, scales, n_sc, m.stream);
        for (int attempt = 0;; ++attempt) {
            m.decode_dev(m.h_codes.as<int64_t>(), scales ? m.h_scales.as<float>() : nullptr, B, T, n_q, m.h_out.as<float>());
            NC_HIP(hipStreamSynchronize(m.stream));
            if (!m.lstm_timed_out() || attempt) break;
            try { m.check_async_errors(); } catch (const Error&) {}   // step-wise LSTM from here on: run again
        }
        m.check_async_errors();
        d2h(pcm, m.h_out.p, n_out, m.stream);
        NC_HIP(hipStreamSynchronize(m.stream));
    });
}

nc_status nc_codec_profile_enable(nc_codec* h, int32_t on) {
    return guard([&] {
        if (!h || !h->impl) fail(NC_EINVAL, "null codec handle");
        h->impl->prof.on = on != 0;
    });
}

nc_status nc_codec_profile_reset(nc_codec* h) {
    return guard([&] {
        if (!h || !h->impl) fail(NC_EINVAL, "null codec handle");
        h->impl->use_device();
        h->impl->prof.reset();
    });
}

nc_status nc_codec_profile_read(nc_codec* h, nc_profile_entry* out) {
    return guard([&] {
        if (!h || !h->impl || !out) fail(NC_EINVAL, "null argument");
        h->impl->use_device();
        NC_HIP(hipStreamSynchronize(h->impl->stream));
        h->impl->prof.resolve();
        for (int i = 0; i < NC_KC_COUNT; ++i) out[i] = h->impl->prof.acc[i];
    });
}

// ---- op-level test hooks -----------------------------------------------------------------------
nc_status nc_op_fold_weight_norm(const float* v, const float* g, int64_t d0, int64_t inner, float* w) {
    return guard([&] {
        if (!v || !g || !w || d0 <= 0 || inner <= 0) fail(NC_EINVAL, "bad arguments");
        fold_weight_norm_dac(v, g, d0, inner, w);
    });
}

static void op_set_device(int device_index) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) fail(NC_EDEVICE, "no HIP device available (the engine has no CPU fallback)");
    if (device_index < 0 || device_index >= n) fail(NC_EINVAL, "device index out of range");
    NC_HIP(hipSetDevice(device_index));
}

nc_status nc_op_conv1d(int device_index, const nc_conv_desc* d, const float* x, const float* weight, const float* bias,
                       const float* alpha_in, const float* alpha_out, const float* residual, float* y, int64_t* Tout_p) {
    return guard([&] {
        if (!d || !x || !weight || !y) fail(NC_EINVAL, "null argument");
        if (d->B <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->K <= 0 || d->stride <= 0 || d->Tin <= 0) fail(NC_EINVAL, "bad conv shape");
        op_set_device(device_index);
        ConvLayer L;
        L.build(weight, bias, d->Cin, d->Cout, d->K, d->stride, d->pad, d->dil, d->out_pad, d->transposed != 0);
        const int64_t Tout = L.out_len(d->Tin);
        if (Tout <= 0) fail(NC_EINVAL, "empty output");
        if (Tout_p) *Tout_p = Tout;
        DevBuf dx, dy, dai, dao, dr;
        const size_t nx = (size_t)d->B * d->Cin * d->Tin * 4, ny = (size_t)d->B * d->Cout * Tout * 4;
        dx.reserve(nx); dy.reserve(ny);
        NC_HIP(hipMemcpy(dx.p, x, nx, hipMemcpyHostToDevice));
        NC_HIP(hipMemset(dy.p, 0, ny));
        if (alpha_in) { dai.reserve(d->Cin * 4); NC_HIP(hipMemcpy(dai.p, alpha_in, d->Cin * 4, hipMemcpyHostToDevice)); }
        if (alpha_out) { dao.reserve(d->Cout * 4); NC_HIP(hipMemcpy(dao.p, alpha_out, d->Cout * 4, hipMemcpyHostToDevice)); }
        if (residual) { dr.reserve(ny); NC_HIP(hipMemcpy(dr.p, residual, ny, hipMemcpyHostToDevice)); }
        ConvIO io{};
        io.x = dx.as<float>(); io.x_bstride = (int64_t)d->Cin * d->Tin; io.x_cstride = d->Tin; io.x_len = (int32_t)d->Tin; io.Tin = d->Tin;
        io.alpha_in = alpha_in ? dai.as<float>() : nullptr;
        io.alpha_out = alpha_out ? dao.as<float>() : nullptr;
        io.res = residual ? dr.as<float>() : nullptr;
        io.y = dy.as<float>(); io.y_bstride = (int64_t)d->Cout * Tout; io.y_cstride = Tout;
        io.epi = d->tanh_out ? EPI_TANH : 0;
        launch_conv(L, io, d->B, nullptr, nullptr);
        NC_HIP(hipDeviceSynchronize());
        NC_HIP(hipMemcpy(y, dy.p, ny, hipMemcpyDeviceToHost));
        dx.release(); dy.release(); dai.release(); dao.release(); dr.release(); L.release_all();
    });
}

nc_status nc_op_conv1d_bench(int device_index, const nc_conv_desc* d, int32_t fuse, int32_t iters, double* avg_ms) {
    return guard([&] {
        if (!d || !avg_ms || iters <= 0) fail(NC_EINVAL, "null argument");
        if (d->B <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->K <= 0 || d->stride <= 0 || d->Tin <= 0) fail(NC_EINVAL, "bad conv shape");
        op_set_device(device_index);
        uint64_t st = 0x9E3779B97F4A7C15ull;
        auto rnd = [&]() {  // uniform [-1,1)
            st = st * 6364136223846793005ull + 1442695040888963407ull;
            return (float)((int64_t)(st >> 40) - (1 << 23)) * (1.0f / (1 << 23));
        };
        const size_t nw = (size_t)d->Cin * d->Cout * d->K;
        std::vector<float> w(nw), bias(d->Cout), al(std::max(d->Cin, d->Cout));
        const float sc = 1.0f / std::sqrt((float)d->Cin * (d->transposed ? 2 : d->K));
        for (auto& v : w) v = rnd() * sc;
        for (auto& v : bias) v = rnd() * 0.1f;
        for (auto& v : al) v = 1.25f + 0.75f * rnd();
        ConvLayer L;
        L.build(w.data(), bias.data(), d->Cin, d->Cout, d->K, d->stride, d->pad, d->dil, d->out_pad, d->transposed != 0);
        const int64_t Tout = L.out_len(d->Tin);
        if (Tout <= 0) fail(NC_EINVAL, "empty output");
        const size_t nx = (size_t)d->B * d->Cin * d->Tin, ny = (size_t)d->B * d->Cout * Tout;
        std::vector<float> hx(nx);
        for (auto& v : hx) v = rnd();
        DevBuf dx, dy, dai, dao, dr;
        dx.reserve(nx * 4); dy.reserve(ny * 4); dai.reserve(al.size() * 4); dao.reserve(al.size() * 4); dr.reserve(ny * 4);
        NC_HIP(hipMemcpy(dx.p, hx.data(), nx * 4, hipMemcpyHostToDevice));
        NC_HIP(hipMemcpy(dai.p, al.data(), al.size() * 4, hipMemcpyHostToDevice));
        NC_HIP(hipMemcpy(dao.p, al.data(), al.size() * 4, hipMemcpyHostToDevice));
        NC_HIP(hipMemset(dr.p, 0, ny * 4));
        ConvIO io{};
        io.x = dx.as<float>(); io.x_bstride = (int64_t)d->Cin * d->Tin; io.x_cstride = d->Tin; io.x_len = (int32_t)d->Tin; io.Tin = d->Tin;
        io.alpha_in = (fuse & 1) ? dai.as<float>() : nullptr;
        io.alpha_out = (fuse & 2) ? dao.as<float>() : nullptr;
        io.res = (fuse & 4) ? dr.as<float>() : nullptr;
        io.y = dy.as<float>(); io.y_bstride = (int64_t)d->Cout * Tout; io.y_cstride = Tout;
        io.epi = d->tanh_out ? EPI_TANH : 0;
        // fuse & 8: Encodec GroupNorm block sums from the epilogue, finished in the launch; fuse & 16: Encodec input mode (pending
        // GroupNorm + ELU applied while staging)
        DevBuf gpart, gcnt, gstats, istats, igam;
        if (fuse & 8) {
            const int sub = conv_gn_sub(d->K, d->stride, d->Cout, d->transposed != 0);
            const int nrb = (int)(((int64_t)d->Cout * sub + 31) / 32), ncb = (int)(((Tout + sub - 1) / sub + 31) / 32);
            gpart.reserve((size_t)d->B * nrb * ncb * 16); gcnt.reserve((size_t)d->B * 4); gstats.reserve((size_t)d->B * 8);
            NC_HIP(hipMemset(gcnt.p, 0, (size_t)d->B * 4));
            if (!conv_gn_fusable(L, io)) fail(NC_EINVAL, "this layer cannot emit GroupNorm sums");
            io.gn_part = gpart.as<double>(); io.gn_nrb = nrb; io.gn_ncb = ncb;
            io.gn_count = gcnt.as<unsigned>(); io.gn_stats = gstats.as<float>(); io.gn_n = gn_count_arg((double)d->Cout * (double)Tout);
        }
        if (fuse & 16) {
            std::vector<float> st((size_t)d->B * 2), g((size_t)d->Cin * 2);
            for (int b = 0; b < d->B; ++b) { st[(size_t)2 * b] = 0.01f * rnd(); st[(size_t)2 * b + 1] = 1.0f + 0.1f * rnd(); }
            for (int c = 0; c < d->Cin; ++c) { g[(size_t)c] = 1.0f + 0.1f * rnd(); g[(size_t)d->Cin + c] = 0.1f * rnd(); }
            istats.reserve(st.size() * 4); igam.reserve(g.size() * 4);
            NC_HIP(hipMemcpy(istats.p, st.data(), st.size() * 4, hipMemcpyHostToDevice));
            NC_HIP(hipMemcpy(igam.p, g.data(), g.size() * 4, hipMemcpyHostToDevice));
            io.in_stats = istats.as<float>(); io.in_gamma = igam.as<float>(); io.in_beta = igam.as<float>() + d->Cin; io.in_elu = true;
        }
        hipEvent_t e0, e1;
        NC_HIP(hipEventCreate(&e0)); NC_HIP(hipEventCreate(&e1));
        for (int i = 0; i < 2; ++i) launch_conv(L, io, d->B, nullptr, nullptr);
        NC_HIP(hipEventRecord(e0, nullptr));
        for (int i = 0; i < iters; ++i) launch_conv(L, io, d->B, nullptr, nullptr);
        NC_HIP(hipEventRecord(e1, nullptr));
        NC_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        NC_HIP(hipEventElapsedTime(&ms, e0, e1));
        *avg_ms = (double)ms / iters;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        dx.release(); dy.release(); dai.release(); dao.release(); dr.release(); L.release_all();
        gpart.release(); gcnt.release(); gstats.release(); istats.release(); igam.release();
    });
}

nc_status nc_op_res_unit(int device_index, int32_t B, int32_t C, int64_t T, int32_t dil, const float* x, const float* w7,
                         const float* b7, const float* a1, const float* a2, const float* w1, const float* b1, int32_t fused,
                         float* y, int32_t iters, double* avg_ms) {
    return guard([&] {
        if (!x || !w7 || !b7 || !a1 || !a2 || !w1 || !b1 || !y) fail(NC_EINVAL, "null argument");
        if (B <= 0 || C <= 0 || T <= 0 || dil <= 0 || iters < 0 || (iters > 0 && !avg_ms)) fail(NC_EINVAL, "bad arguments");
        op_set_device(device_index);
        ConvLayer c7, c1;
        c7.build(w7, b7, C, C, 7, 1, 3 * dil, dil, 0, false);
        c1.build(w1, b1, C, C, 1, 1, 0, 1, 0, false);
        if (fused && !can_fuse_res_unit(c7, c1)) fail(NC_EUNSUPPORTED, "no fused residual-unit kernel for %d channels", C);
        const size_t n = (size_t)B * C * T * 4;
        DevBuf dx, dh, dy, d1, d2;
        dx.reserve(n); dh.reserve(n); dy.reserve(n); d1.reserve(C * 4); d2.reserve(C * 4);
        NC_HIP(hipMemcpy(dx.p, x, n, hipMemcpyHostToDevice));
        NC_HIP(hipMemcpy(d1.p, a1, C * 4, hipMemcpyHostToDevice));
        NC_HIP(hipMemcpy(d2.p, a2, C * 4, hipMemcpyHostToDevice));
        auto run = [&]() {
            ConvIO io{};
            io.x = dx.as<float>(); io.x_bstride = (int64_t)C * T; io.x_cstride = T; io.x_len = (int32_t)T; io.Tin = T;
            io.alpha_in = d1.as<float>(); io.alpha_out = d2.as<float>();
            io.y_bstride = (int64_t)C * T; io.y_cstride = T;
            if (fused) {
                io.res = dx.as<float>(); io.fuse_k1 = &c1; io.y = dy.as<float>();
                launch_conv(c7, io, B, nullptr, nullptr);
            } else {
                io.y = dh.as<float>();
                launch_conv(c7, io, B, nullptr, nullptr);
                ConvIO i2{};
                i2.x = dh.as<float>(); i2.x_bstride = (int64_t)C * T; i2.x_cstride = T; i2.x_len = (int32_t)T; i2.Tin = T;
                i2.res = dx.as<float>(); i2.y = dy.as<float>(); i2.y_bstride = (int64_t)C * T; i2.y_cstride = T;
                launch_conv(c1, i2, B, nullptr, nullptr);
            }
        };
        run();
        NC_HIP(hipDeviceSynchronize());
        NC_HIP(hipMemcpy(y, dy.p, n, hipMemcpyDeviceToHost));
        if (iters > 0) {
            hipEvent_t e0, e1;
            NC_HIP(hipEventCreate(&e0)); NC_HIP(hipEventCreate(&e1));
            NC_HIP(hipEventRecord(e0, nullptr));
            for (int i = 0; i < iters; ++i) run();
            NC_HIP(hipEventRecord(e1, nullptr));
            NC_HIP(hipEventSynchronize(e1));
            float ms = 0.f;
            NC_HIP(hipEventElapsedTime(&ms, e0, e1));
            *avg_ms = (double)ms / iters;
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        }
        dx.release(); dh.release(); dy.release(); d1.release(); d2.release();
        c7.release_all(); c1.release_all();
    });
}

nc_status nc_op_vq_argmin(int device_index, const float* z_e, int32_t B, int32_t D, int64_t T, const float* codebook, int32_t N,
                          int64_t* idx, float* st) {
    return guard([&] {
        if (!z_e || !codebook || !idx || !st || B <= 0 || D <= 0 || T <= 0 || N <= 0) fail(NC_EINVAL, "bad arguments");
        op_set_device(device_index);
        Codebook cb;
        cb.build(codebook, N, D);
        DevBuf dz, di, ds;
        const size_t nz = (size_t)B * D * T * 4;
        dz.reserve(nz); ds.reserve(nz); di.reserve((size_t)B * T * 8);
        NC_HIP(hipMemcpy(dz.p, z_e, nz, hipMemcpyHostToDevice));
        launch_vq_argmin(cb, dz.as<float>(), (int64_t)D * T, B, T, di.as<int64_t>(), T, ds.as<float>(), nullptr, nullptr);
        NC_HIP(hipDeviceSynchronize());
        NC_HIP(hipMemcpy(idx, di.p, (size_t)B * T * 8, hipMemcpyDeviceToHost));
        NC_HIP(hipMemcpy(st, ds.p, nz, hipMemcpyDeviceToHost));
        dz.release(); di.release(); ds.release(); cb.cbT.release(); cb.cb.release(); cb.c2.release();
    });
}

nc_status nc_op_euclid_rvq(int device_index, const float* residual, int32_t B, int32_t D, int64_t T, const float* codebooks, int32_t n_q, int32_t N,
                           int32_t form, int64_t* codes, float* residual_out) {
    return guard([&] {
        if (!residual || !codebooks || !codes || B <= 0 || D <= 0 || T <= 0 || N <= 0 || n_q <= 0 || form < 0 || form > 1) fail(NC_EINVAL, "bad arguments");
        op_set_device(device_index);
        op_euclid_rvq(residual, B, D, T, codebooks, n_q, N, form, codes, residual_out);
    });
}

}  // extern "C"
