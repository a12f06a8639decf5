/* A C99 consumer of include/nc_mi355x.h that runs the codec on the GPU -- what the managed [DllImport] binding will do,
 * with no Python and no ctypes between the caller and the library (VERDICT r4 "missing" item 4).
 *
 *   abi_consumer <config.bin> <weights.blob> <pcm.f32> <B> <T> <want_codes.i64> <want_pcm.f32>
 *
 * config.bin = the bytes of an nc_dac_config; pcm.f32 = B*T floats; the fixtures hold what the C oracle emitted for the same input
 * (tests/test_dac_gpu.py::test_c99_consumer_runs_the_codec_through_the_abi writes them).  The program walks the calls of
 * Models/DAC.cs in the order Examples/Program.cs:252-291 makes them: create -> LoadWeights -> Encode -> Decode -> Dispose, checks the
 * error convention on the way (sample-rate mismatch -> NC_EINVAL + message, DAC.cs:146), and prints "CONSUMER_OK codes=<n> pcm_max_abs=<d>". */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nc_mi355x.h"

static void* slurp(const char* path, size_t want_bytes) {
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    void* p = malloc(want_bytes ? want_bytes : 1);
    if (!p || fread(p, 1, want_bytes, f) != want_bytes) { fprintf(stderr, "%s: short read (%zu bytes wanted)\n", path, want_bytes); exit(2); }
    fclose(f);
    return p;
}

#define CHECK(call)                                                                         \
    do {                                                                                    \
        nc_status st__ = (call);                                                            \
        if (st__ != NC_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, (int)st__, nc_last_error()); return 1; } \
    } while (0)

int main(int argc, char** argv) {
    if (argc != 8) { fprintf(stderr, "usage: abi_consumer config.bin weights.blob pcm.f32 B T want_codes.i64 want_pcm.f32\n"); return 2; }
    const int32_t B = (int32_t)atoi(argv[4]);
    const int64_t T = (int64_t)atoll(argv[5]);
    nc_dac_config* cfg = (nc_dac_config*)slurp(argv[1], sizeof(nc_dac_config));
    float* pcm = (float*)slurp(argv[3], (size_t)B * (size_t)T * sizeof(float));

    if (nc_device_count() < 1) { fprintf(stderr, "no gfx950 device: %s\n", nc_last_error()); return 3; }
    nc_codec* h = NULL;
    CHECK(nc_dac_create(cfg, 0, &h));
    if (nc_codec_load_weights(h, "/nonexistent/weights.blob") != NC_ENOTFOUND) { fprintf(stderr, "missing file was not NC_ENOTFOUND\n"); return 1; }
    CHECK(nc_codec_load_weights(h, argv[2]));                                 /* INeuralCodec.LoadWeights */

    int64_t T_pad = 0, frames = 0;
    CHECK(nc_dac_query(h, T, &T_pad, &frames));                                /* DAC.Preprocess */
    const int32_t nq = cfg->n_codebooks;
    const size_t n_codes = (size_t)B * (size_t)nq * (size_t)frames, n_z = (size_t)B * (size_t)cfg->latent_dim * (size_t)frames;
    /* decoded length: every DecoderBlock's transposed convolution maps L -> (L - 1) s - 2 ceil(s / 2) + 2 s (DecoderBlock.cs:20-44,
     * WNConvTranspose1d.cs:142-163): = T_pad for even strides, shorter when a stride is odd (the 16 / 24 kHz presets' stride 5) */
    int64_t L_out = frames;
    for (int i = 0; i < cfg->n_decoder_rates; ++i) {
        const int64_t st = cfg->decoder_rates[i];
        L_out = (L_out - 1) * st - 2 * ((st + 1) / 2) + 2 * st;
    }
    if (L_out > T_pad) { fprintf(stderr, "decoded length %lld exceeds the padded input %lld\n", (long long)L_out, (long long)T_pad); return 1; }
    const size_t n_out = (size_t)B * (size_t)L_out;
    int64_t* codes = (int64_t*)malloc(n_codes * sizeof(int64_t));
    float* z = (float*)malloc(n_z * sizeof(float));
    float* out = (float*)malloc(n_out * sizeof(float));
    if (!codes || !z || !out) return 2;

    /* the reference throws ArgumentException on a sample-rate mismatch (DAC.cs:146): status + thread-local message here */
    if (nc_dac_encode(h, pcm, B, T, cfg->sample_rate + 1, 0, codes, z, NULL) != NC_EINVAL || strlen(nc_last_error()) == 0) {
        fprintf(stderr, "sample-rate mismatch was not rejected with NC_EINVAL + message\n");
        return 1;
    }
    CHECK(nc_dac_encode(h, pcm, B, T, cfg->sample_rate, 0, codes, z, NULL));   /* DAC.Encode(Tensor, nQ, sr) */
    CHECK(nc_dac_decode(h, z, B, frames, out));                                /* DAC.Decode(z) */
    CHECK(nc_codec_destroy(h));                                                /* Dispose */

    int64_t* want_codes = (int64_t*)slurp(argv[6], n_codes * sizeof(int64_t));
    float* want_pcm = (float*)slurp(argv[7], n_out * sizeof(float));
    size_t bad = 0;
    for (size_t i = 0; i < n_codes; ++i) bad += codes[i] != want_codes[i];
    double worst = 0.0;
    for (size_t i = 0; i < n_out; ++i) {
        const double d = fabs((double)out[i] - (double)want_pcm[i]);
        if (d > worst) worst = d;
    }
    if (bad) { fprintf(stderr, "%zu of %zu codes differ from the fixture\n", bad, n_codes); return 1; }
    if (!(worst <= 1e-4)) { fprintf(stderr, "decoded PCM differs from the fixture by %g (> 1e-4)\n", worst); return 1; }
    printf("CONSUMER_OK codes=%zu pcm_max_abs=%g\n", n_codes, worst);
    free(cfg); free(pcm); free(codes); free(z); free(out); free(want_codes); free(want_pcm);
    return 0;
}
