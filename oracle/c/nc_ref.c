/* ORACLE (test infrastructure; never linked into or called by the product library).
 *
 * Plain-C CPU restatement of the reference's codec hot path, written against the parity
 * spec's canonical arithmetic:
 *   - every convolution output is ONE binary32 fma chain over (ci ascending, k ascending),
 *     started from +0, bias added last:  y = chain + bias
 *   - sin/tanh/exp are the fixed polynomial definitions of ref_math.h
 *   - weight-norm is folded once: w = (v / (fl(sqrt(fl(sum_f64 fl(v*v)))) + 1e-7f)) * g
 * so that a conforming GPU implementation is bit-identical to this file, and this file is
 * pinned against the PyTorch-CPU restatement's golden vectors (tests/golden) within fp32
 * round-off (the reference itself has no tests: "parity unpinned").
 *
 * Reference functions restated (under /root/reference/NeuralCodecs.Torch/):
 *   Modules/DAC/WNConv1d.cs:140-156, WNConvTranspose1d.cs:142-163, Snake1d.cs:49-58,
 *   ResidualUnit.cs:24-59, EncoderBlock.cs:20-43, Encoder.cs:21-58, DecoderBlock.cs:20-44,
 *   Decoder.cs:22-58, VectorQuantizer.cs:64-142, ResidualVectorQuantizer.cs:54-103,211-238,
 *   Models/DAC.cs:141-154,163-181,231-234,101-106.
 *
 * Build: make -C oracle   (gcc -O3 -mfma -mavx2 -ffp-contract=off -fopenmp)
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "ref_math.h"


#include "nc_ref_internal.h"

int ref_blob_parse(const uint8_t* buf, int64_t len, ref_blob* out) {
    if (len < 24 || memcmp(buf, "NCWB0001", 8) != 0) return -1;
    uint64_t n, idx_len;
    memcpy(&n, buf + 8, 8);
    memcpy(&idx_len, buf + 16, 8);
    int64_t data0 = (24 + (int64_t)idx_len + 63) & ~63LL;
    out->n = (int)n;
    out->t = (ref_tensor*)calloc(n, sizeof(ref_tensor));
    int64_t p = 24;
    for (uint64_t i = 0; i < n; i++) {
        uint16_t ln;
        memcpy(&ln, buf + p, 2); p += 2;
        if (ln >= sizeof(out->t[i].name)) return -2;
        memcpy(out->t[i].name, buf + p, ln); p += ln;
        out->t[i].dtype = buf[p]; out->t[i].ndim = buf[p + 1]; p += 2;
        for (int d = 0; d < out->t[i].ndim; d++) { uint64_t v; memcpy(&v, buf + p, 8); p += 8; out->t[i].dims[d] = (int64_t)v; }
        uint64_t off, nb;
        memcpy(&off, buf + p, 8); memcpy(&nb, buf + p + 8, 8); p += 16;
        out->t[i].data = buf + data0 + off;
        out->t[i].nbytes = (int64_t)nb;
        if (data0 + (int64_t)off + (int64_t)nb > len) return -3;
    }
    return 0;
}

const ref_tensor* ref_blob_find(const ref_blob* b, const char* name) {
    for (int i = 0; i < b->n; i++)
        if (strcmp(b->t[i].name, name) == 0) return &b->t[i];
    return NULL;
}

/* ------------------------------------------------------------------ ops */

/* Weight-norm fold, DAC flavour (D2): w = v/(||v||+1e-7) * g, norm over all dims but 0. */
REF_API void ref_fold_wn_dac(const float* v, const float* g, int64_t d0, int64_t inner, float* w) {
    for (int64_t i = 0; i < d0; i++) {
        double ss = 0.0;
        for (int64_t j = 0; j < inner; j++) { float q = v[i * inner + j] * v[i * inner + j]; ss += (double)q; }
        float denom = sqrtf((float)ss) + 1e-7f;
        for (int64_t j = 0; j < inner; j++) w[i * inner + j] = (v[i * inner + j] / denom) * g[i];
    }
}

REF_API void ref_snake(const float* x, const float* alpha, int64_t B, int64_t C, int64_t T, float* y) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t b = 0; b < B; b++)
        for (int64_t c = 0; c < C; c++) {
            const float a = alpha[c];
            const float* xr = x + (b * C + c) * T;
            float* yr = y + (b * C + c) * T;
            for (int64_t t = 0; t < T; t++) yr[t] = ref_snakef(xr[t], a);
        }
}

REF_API void ref_tanh(const float* x, int64_t n, float* y) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) y[i] = ref_tanhf(x[i]);
}

static inline int64_t conv_out_len(int64_t Tin, int K, int stride, int pad, int dil) {
    return (Tin + 2 * (int64_t)pad - (int64_t)dil * (K - 1) - 1) / stride + 1;
}

#define TB 1024 /* time block kept in L1 */
#define CO_B 4  /* output channels sharing one pass over x */

/* y[b,co,t] = bias[co] + chain_{ci asc, k asc} w[co,ci,k] * x[b, g*Cin_g+ci, t*stride + k*dil - pad]
 * x is logically zero outside [0, Tin_valid) (conv zero padding; also DAC.Preprocess right-pad when
 * Tin_valid < Tin_logical).  residual (nullable) is added after the bias: y = (chain + bias) + res. */
REF_API void ref_conv1d(const float* x, int64_t B, int Cin, int64_t Tin, const float* w, const float* bias, int Cout, int K,
                        int stride, int pad, int dil, int groups, const float* residual, float* y, int64_t Tout) {
    const int cin_g = Cin / groups, cout_g = Cout / groups;
    const int64_t nblk = (Tout + TB - 1) / TB;
    const int ncb = (cout_g + CO_B - 1) / CO_B;
#pragma omp parallel for collapse(3) schedule(dynamic, 1)
    for (int64_t b = 0; b < B; b++)
        for (int gc = 0; gc < groups * ncb; gc++)
            for (int64_t tb = 0; tb < nblk; tb++) {
                const int g = gc / ncb, cb = gc % ncb;
                const int co0 = g * cout_g + cb * CO_B;
                const int nco = (cb * CO_B + CO_B <= cout_g) ? CO_B : (cout_g - cb * CO_B);
                const int64_t t0 = tb * TB, t1 = (t0 + TB < Tout) ? t0 + TB : Tout;
                float acc[CO_B][TB];
                for (int j = 0; j < nco; j++)
                    for (int64_t t = 0; t < t1 - t0; t++) acc[j][t] = 0.0f;
                for (int ci = 0; ci < cin_g; ci++) {
                    const float* xr = x + ((int64_t)b * Cin + g * cin_g + ci) * Tin;
                    for (int k = 0; k < K; k++) {
                        const int64_t off = (int64_t)k * dil - pad;
                        /* valid t: 0 <= t*stride+off < Tin */
                        int64_t lo = t0, hi = t1;
                        if (off < 0) { int64_t m = (-off + stride - 1) / stride; if (m > lo) lo = m; }
                        { int64_t m = (Tin - 1 - off) >= 0 ? (Tin - 1 - off) / stride + 1 : 0; if (m < hi) hi = m; }
                        if (lo >= hi) continue;
                        if (stride == 1) {
                            const float* xs = xr + off;
                            for (int j = 0; j < nco; j++) {
                                const float wv = w[((int64_t)(co0 + j) * cin_g + ci) * K + k];
                                float* a = acc[j] - t0;
                                for (int64_t t = lo; t < hi; t++) a[t] = fmaf(wv, xs[t], a[t]);
                            }
                        } else {
                            for (int j = 0; j < nco; j++) {
                                const float wv = w[((int64_t)(co0 + j) * cin_g + ci) * K + k];
                                float* a = acc[j] - t0;
                                for (int64_t t = lo; t < hi; t++) a[t] = fmaf(wv, xr[t * stride + off], a[t]);
                            }
                        }
                    }
                }
                for (int j = 0; j < nco; j++) {
                    float* yr = y + ((int64_t)b * Cout + co0 + j) * Tout;
                    const float bv = bias ? bias[co0 + j] : 0.0f;
                    const float* rr = residual ? residual + ((int64_t)b * Cout + co0 + j) * Tout : NULL;
                    for (int64_t t = t0; t < t1; t++) {
                        float v = acc[j][t - t0] + bv;
                        if (rr) v = v + rr[t];
                        yr[t] = v;
                    }
                }
            }
}

/* conv_transpose1d, weight [Cin, Cout, K] (groups=1, dilation=1):
 * y[b,co,t] = bias[co] + chain_{ci asc, k asc, (t+pad-k)%stride==0, 0<=q<Tin} w[ci,co,k] * x[b,ci,q],  q=(t+pad-k)/stride
 * Tout = (Tin-1)*stride - 2*pad + K + output_padding. */
REF_API void ref_conv_transpose1d(const float* x, int64_t B, int Cin, int64_t Tin, const float* w, const float* bias, int Cout,
                                  int K, int stride, int pad, int out_pad, float* y, int64_t Tout) {
    (void)out_pad;
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int64_t b = 0; b < B; b++)
        for (int co = 0; co < Cout; co++) {
            float* yr = y + ((int64_t)b * Cout + co) * Tout;
            for (int64_t t = 0; t < Tout; t++) yr[t] = 0.0f;
            for (int ci = 0; ci < Cin; ci++) {
                const float* xr = x + ((int64_t)b * Cin + ci) * Tin;
                for (int k = 0; k < K; k++) {
                    const float wv = w[((int64_t)ci * Cout + co) * K + k];
                    /* t = q*stride - pad + k */
                    int64_t q_lo = 0, q_hi = Tin;
                    const int64_t base = (int64_t)k - pad;
                    if (base < 0) q_lo = (-base + stride - 1) / stride;
                    { int64_t m = (Tout - 1 - base) >= 0 ? (Tout - 1 - base) / stride + 1 : 0; if (m < q_hi) q_hi = m; }
                    float* yo = yr + base;
                    for (int64_t q = q_lo; q < q_hi; q++) yo[q * stride] = fmaf(wv, xr[q], yo[q * stride]);
                }
            }
            const float bv = bias ? bias[co] : 0.0f;
            for (int64_t t = 0; t < Tout; t++) yr[t] = yr[t] + bv;
        }
}

/* One VQ stage on projected latents z_e [B,D,T]: squared-Euclidean argmin over the codebook [N,D]
 * (VectorQuantizer.cs:99-125, D1) -> idx [B,T] (ATen argmin: first index on ties, the first NaN of a row that holds one), and the
 * straight-through value st = z_e + (cb[idx] - z_e) [B,D,T] (VectorQuantizer.cs:81). */
REF_API void ref_vq_argmin(const float* z_e, int64_t B, int D, int64_t T, const float* cb, int N, int64_t* idx, float* st,
                           float* best_dist /*nullable [B,T]*/) {
    float* c2 = (float*)malloc(sizeof(float) * N);
    for (int n = 0; n < N; n++) {
        float a = 0.0f;
        for (int d = 0; d < D; d++) a = fmaf(cb[n * D + d], cb[n * D + d], a);
        c2[n] = a;
    }
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t b = 0; b < B; b++)
        for (int64_t t = 0; t < T; t++) {
            float e[256];
            float e2 = 0.0f;
            for (int d = 0; d < D; d++) { e[d] = z_e[((int64_t)b * D + d) * T + t]; e2 = fmaf(e[d], e[d], e2); }
            float best = INFINITY;
            int bi = 0;
            for (int n = 0; n < N; n++) {
                float cr = 0.0f;
                for (int d = 0; d < D; d++) cr = fmaf(e[d], cb[n * D + d], cr);
                float dist = (e2 + c2[n]) - 2.0f * cr;
                /* ATen argmin order (VectorQuantizer.cs:121 dist.argmin(1)): a NaN beats every number, the first one wins; equal
                 * distances keep the lower index (ascending scan) */
                if (!(dist >= best) && best == best) { best = dist; bi = n; }
            }
            idx[b * T + t] = bi;
            if (best_dist) best_dist[b * T + t] = best;
            for (int d = 0; d < D; d++) {
                float q = cb[bi * D + d];
                st[((int64_t)b * D + d) * T + t] = e[d] + (q - e[d]);
            }
        }
    free(c2);
}

/* Embedding gather + transpose: codes [B,T] -> [B,D,T]  (VectorQuantizer.cs:135-142) */
REF_API void ref_vq_gather(const int64_t* idx, int64_t B, int D, int64_t T, const float* cb, float* out) {
    for (int64_t b = 0; b < B; b++)
        for (int64_t t = 0; t < T; t++)
            for (int d = 0; d < D; d++) out[((int64_t)b * D + d) * T + t] = cb[idx[b * T + t] * D + d];
}

/* ------------------------------------------------------------------ DAC model */
typedef struct {
    int sample_rate, encoder_dim, n_enc_rates, enc_rates[8], decoder_dim, n_dec_rates, dec_rates[8];
    int latent_dim, n_codebooks, codebook_size, codebook_dim;
} ref_dac_config;

typedef struct {
    float *w, *b; /* folded weight, bias */
    int cin, cout, k;
} ref_conv_p;

typedef struct {
    ref_dac_config cfg;
    int hop;
    /* encoder */
    ref_conv_p enc_stem;
    struct { const float *a1[3], *a2[3]; ref_conv_p c7[3], c1[3]; const float* a_down; ref_conv_p down; } enc_blk[8];
    const float* enc_alpha_out;
    ref_conv_p enc_out;
    /* quantizer */
    ref_conv_p in_proj[64], out_proj[64];
    const float* codebook[64];
    /* decoder */
    ref_conv_p dec_in;
    struct { const float* a_up; ref_conv_p up; const float *a1[3], *a2[3]; ref_conv_p c7[3], c1[3]; } dec_blk[8];
    const float* dec_alpha_out;
    ref_conv_p dec_out;
    uint8_t* blob_copy;
    ref_blob blob;
} ref_dac;

static int load_wn(const ref_blob* bl, const char* prefix, ref_conv_p* p, int transpose) {
    char nm[256];
    snprintf(nm, sizeof nm, "%s.weight_v", prefix);
    const ref_tensor* v = ref_blob_find(bl, nm);
    snprintf(nm, sizeof nm, "%s.weight_g", prefix);
    const ref_tensor* g = ref_blob_find(bl, nm);
    snprintf(nm, sizeof nm, "%s.bias", prefix);
    const ref_tensor* b = ref_blob_find(bl, nm);
    if (!v || !g) { fprintf(stderr, "nc_ref: missing %s\n", prefix); return -1; }
    int64_t d0 = v->dims[0], inner = v->dims[1] * v->dims[2];
    p->w = (float*)malloc(sizeof(float) * d0 * inner);
    ref_fold_wn_dac((const float*)v->data, (const float*)g->data, d0, inner, p->w);
    p->b = b ? (float*)b->data : NULL;
    p->k = (int)v->dims[2];
    if (transpose) { p->cin = (int)v->dims[0]; p->cout = (int)v->dims[1]; }
    else { p->cout = (int)v->dims[0]; p->cin = (int)v->dims[1]; }
    return 0;
}

static const float* load_alpha(const ref_blob* bl, const char* fmt_name) {
    const ref_tensor* t = ref_blob_find(bl, fmt_name);
    if (!t) { fprintf(stderr, "nc_ref: missing %s\n", fmt_name); return NULL; }
    return (const float*)t->data;
}

REF_API void ref_dac_destroy(ref_dac* m);

REF_API ref_dac* ref_dac_create(const ref_dac_config* cfg, const uint8_t* blob, int64_t blob_len) {
    ref_dac* m = (ref_dac*)calloc(1, sizeof(ref_dac));
    m->cfg = *cfg;
    m->blob_copy = (uint8_t*)malloc(blob_len);
    memcpy(m->blob_copy, blob, blob_len);
    if (ref_blob_parse(m->blob_copy, blob_len, &m->blob) != 0) { ref_dac_destroy(m); return NULL; }
    const ref_blob* bl = &m->blob;
    char nm[256];
    int bad = 0;
    m->hop = 1;
    for (int i = 0; i < cfg->n_enc_rates; i++) m->hop *= cfg->enc_rates[i];
    bad |= load_wn(bl, "encoder.block.0", &m->enc_stem, 0);
    for (int bi = 0; bi < cfg->n_enc_rates; bi++) {
        for (int u = 0; u < 3; u++) {
            snprintf(nm, sizeof nm, "encoder.block.%d.block.%d.block.0.alpha", bi + 1, u); m->enc_blk[bi].a1[u] = load_alpha(bl, nm);
            snprintf(nm, sizeof nm, "encoder.block.%d.block.%d.block.1", bi + 1, u); bad |= load_wn(bl, nm, &m->enc_blk[bi].c7[u], 0);
            snprintf(nm, sizeof nm, "encoder.block.%d.block.%d.block.2.alpha", bi + 1, u); m->enc_blk[bi].a2[u] = load_alpha(bl, nm);
            snprintf(nm, sizeof nm, "encoder.block.%d.block.%d.block.3", bi + 1, u); bad |= load_wn(bl, nm, &m->enc_blk[bi].c1[u], 0);
            bad |= !m->enc_blk[bi].a1[u] || !m->enc_blk[bi].a2[u];
        }
        snprintf(nm, sizeof nm, "encoder.block.%d.block.3.alpha", bi + 1); m->enc_blk[bi].a_down = load_alpha(bl, nm);
        snprintf(nm, sizeof nm, "encoder.block.%d.block.4", bi + 1); bad |= load_wn(bl, nm, &m->enc_blk[bi].down, 0);
        bad |= !m->enc_blk[bi].a_down;
    }
    snprintf(nm, sizeof nm, "encoder.block.%d.alpha", cfg->n_enc_rates + 1); m->enc_alpha_out = load_alpha(bl, nm);
    snprintf(nm, sizeof nm, "encoder.block.%d", cfg->n_enc_rates + 2); bad |= load_wn(bl, nm, &m->enc_out, 0);
    for (int i = 0; i < cfg->n_codebooks; i++) {
        snprintf(nm, sizeof nm, "quantizer.quantizers.%d.in_proj", i); bad |= load_wn(bl, nm, &m->in_proj[i], 0);
        snprintf(nm, sizeof nm, "quantizer.quantizers.%d.out_proj", i); bad |= load_wn(bl, nm, &m->out_proj[i], 0);
        snprintf(nm, sizeof nm, "quantizer.quantizers.%d.codebook.weight", i);
        const ref_tensor* t = ref_blob_find(bl, nm);
        if (!t) bad = 1; else m->codebook[i] = (const float*)t->data;
    }
    bad |= load_wn(bl, "decoder.model.0", &m->dec_in, 0);
    for (int bi = 0; bi < cfg->n_dec_rates; bi++) {
        snprintf(nm, sizeof nm, "decoder.model.%d.block.0.alpha", bi + 1); m->dec_blk[bi].a_up = load_alpha(bl, nm);
        snprintf(nm, sizeof nm, "decoder.model.%d.block.1", bi + 1); bad |= load_wn(bl, nm, &m->dec_blk[bi].up, 1);
        bad |= !m->dec_blk[bi].a_up;
        for (int u = 0; u < 3; u++) {
            snprintf(nm, sizeof nm, "decoder.model.%d.block.%d.block.0.alpha", bi + 1, u + 2); m->dec_blk[bi].a1[u] = load_alpha(bl, nm);
            snprintf(nm, sizeof nm, "decoder.model.%d.block.%d.block.1", bi + 1, u + 2); bad |= load_wn(bl, nm, &m->dec_blk[bi].c7[u], 0);
            snprintf(nm, sizeof nm, "decoder.model.%d.block.%d.block.2.alpha", bi + 1, u + 2); m->dec_blk[bi].a2[u] = load_alpha(bl, nm);
            snprintf(nm, sizeof nm, "decoder.model.%d.block.%d.block.3", bi + 1, u + 2); bad |= load_wn(bl, nm, &m->dec_blk[bi].c1[u], 0);
            bad |= !m->dec_blk[bi].a1[u] || !m->dec_blk[bi].a2[u];
        }
    }
    snprintf(nm, sizeof nm, "decoder.model.%d.alpha", cfg->n_dec_rates + 1); m->dec_alpha_out = load_alpha(bl, nm);
    snprintf(nm, sizeof nm, "decoder.model.%d", cfg->n_dec_rates + 2); bad |= load_wn(bl, nm, &m->dec_out, 0);
    bad |= !m->enc_alpha_out || !m->dec_alpha_out;
    if (bad) { ref_dac_destroy(m); return NULL; }
    return m;
}

static void free_conv(ref_conv_p* p) { free(p->w); p->w = NULL; }

REF_API void ref_dac_destroy(ref_dac* m) {
    if (!m) return;
    free_conv(&m->enc_stem); free_conv(&m->enc_out); free_conv(&m->dec_in); free_conv(&m->dec_out);
    for (int bi = 0; bi < 8; bi++) {
        for (int u = 0; u < 3; u++) { free_conv(&m->enc_blk[bi].c7[u]); free_conv(&m->enc_blk[bi].c1[u]); free_conv(&m->dec_blk[bi].c7[u]); free_conv(&m->dec_blk[bi].c1[u]); }
        free_conv(&m->enc_blk[bi].down); free_conv(&m->dec_blk[bi].up);
    }
    for (int i = 0; i < 64; i++) { free_conv(&m->in_proj[i]); free_conv(&m->out_proj[i]); }
    free(m->blob.t);
    free(m->blob_copy);
    free(m);
}

REF_API int64_t ref_dac_padded_length(const ref_dac* m, int64_t T) { return (T + m->hop - 1) / m->hop * m->hop; }
REF_API int64_t ref_dac_frames(const ref_dac* m, int64_t T) { return (T + m->hop - 1) / m->hop; }

/* x + conv1(snake(conv7_dil(snake(x))))  (ResidualUnit.cs:29-34,50-59) */
static float* res_unit(float* x, int64_t B, int C, int64_t T, const float* a1, const ref_conv_p* c7, const float* a2,
                       const ref_conv_p* c1, int dil) {
    float* s = (float*)malloc(sizeof(float) * B * C * T);
    float* h = (float*)malloc(sizeof(float) * B * C * T);
    ref_snake(x, a1, B, C, T, s);
    ref_conv1d(s, B, C, T, c7->w, c7->b, C, 7, 1, 3 * dil, dil, 1, NULL, h, T);
    ref_snake(h, a2, B, C, T, s);
    ref_conv1d(s, B, C, T, c1->w, c1->b, C, 1, 1, 0, 1, 1, x, h, T);
    free(s);
    free(x);
    return h;
}

static const int DIL[3] = {1, 9 / 3, 9};

/* Encoder.forward on zero-right-padded audio (DAC.cs:141-154 + Encoder.cs:58). Returns z [B,latent,T'] (malloc). */
static float* dac_encoder(const ref_dac* m, const float* pcm, int64_t B, int64_t T, int64_t* Tz) {
    const ref_dac_config* c = &m->cfg;
    int64_t Tp = ref_dac_padded_length(m, T);
    float* xin = (float*)calloc(B * Tp, sizeof(float));
    for (int64_t b = 0; b < B; b++) memcpy(xin + b * Tp, pcm + b * T, sizeof(float) * T);
    int C = c->encoder_dim;
    int64_t L = Tp;
    float* x = (float*)malloc(sizeof(float) * B * C * L);
    ref_conv1d(xin, B, 1, L, m->enc_stem.w, m->enc_stem.b, C, 7, 1, 3, 1, 1, NULL, x, L);
    free(xin);
    for (int bi = 0; bi < c->n_enc_rates; bi++) {
        for (int u = 0; u < 3; u++) x = res_unit(x, B, C, L, m->enc_blk[bi].a1[u], &m->enc_blk[bi].c7[u], m->enc_blk[bi].a2[u], &m->enc_blk[bi].c1[u], DIL[u]);
        int s = c->enc_rates[bi];
        int pad = (s + 1) / 2;
        int64_t Lo = conv_out_len(L, 2 * s, s, pad, 1);
        float* sn = (float*)malloc(sizeof(float) * B * C * L);
        ref_snake(x, m->enc_blk[bi].a_down, B, C, L, sn);
        float* y = (float*)malloc(sizeof(float) * B * 2 * C * Lo);
        ref_conv1d(sn, B, C, L, m->enc_blk[bi].down.w, m->enc_blk[bi].down.b, 2 * C, 2 * s, s, pad, 1, 1, NULL, y, Lo);
        free(sn); free(x);
        x = y; C *= 2; L = Lo;
    }
    float* sn = (float*)malloc(sizeof(float) * B * C * L);
    ref_snake(x, m->enc_alpha_out, B, C, L, sn);
    float* z = (float*)malloc(sizeof(float) * B * c->latent_dim * L);
    ref_conv1d(sn, B, C, L, m->enc_out.w, m->enc_out.b, c->latent_dim, 3, 1, 1, 1, 1, NULL, z, L);
    free(sn); free(x);
    *Tz = L;
    return z;
}

/* DAC.Encode(Tensor, nQuantizers) -> (z_q, codes, latents).  Outputs nullable except codes.
 * pcm [B,1,T]; codes [B,nq,T'] int64; zq [B,latent,T']; latents [B,nq*D,T']; z_enc (pre-quantizer encoder output) nullable. */
REF_API int ref_dac_encode(const ref_dac* m, const float* pcm, int64_t B, int64_t T, int n_q, int64_t* codes, float* zq_out,
                           float* latents_out, float* z_enc_out) {
    const ref_dac_config* c = &m->cfg;
    int nq = (n_q <= 0 || n_q > c->n_codebooks) ? c->n_codebooks : n_q;
    int64_t Tz;
    float* z = dac_encoder(m, pcm, B, T, &Tz);
    const int D = c->codebook_dim, LD = c->latent_dim;
    const int64_t nz = B * LD * Tz;
    if (z_enc_out) memcpy(z_enc_out, z, sizeof(float) * nz);
    float* residual = z; /* residual = z.clone() */
    float* zq = (float*)calloc(nz, sizeof(float));
    float* ze = (float*)malloc(sizeof(float) * B * D * Tz);
    float* st = (float*)malloc(sizeof(float) * B * D * Tz);
    float* zqi = (float*)malloc(sizeof(float) * nz);
    int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * B * Tz);
    for (int i = 0; i < nq; i++) {
        ref_conv1d(residual, B, LD, Tz, m->in_proj[i].w, m->in_proj[i].b, D, 1, 1, 0, 1, 1, NULL, ze, Tz);
        ref_vq_argmin(ze, B, D, Tz, m->codebook[i], c->codebook_size, idx, st, NULL);
        ref_conv1d(st, B, D, Tz, m->out_proj[i].w, m->out_proj[i].b, LD, 1, 1, 0, 1, 1, NULL, zqi, Tz);
        for (int64_t j = 0; j < nz; j++) { zq[j] = zq[j] + zqi[j]; residual[j] = residual[j] - zqi[j]; }
        for (int64_t b = 0; b < B; b++) {
            memcpy(codes + (b * nq + i) * Tz, idx + b * Tz, sizeof(int64_t) * Tz);
            if (latents_out) memcpy(latents_out + ((b * nq + i) * D) * Tz, ze + b * D * Tz, sizeof(float) * D * Tz);
        }
    }
    if (zq_out) memcpy(zq_out, zq, sizeof(float) * nz);
    free(residual); free(zq); free(ze); free(st); free(zqi); free(idx);
    return 0;
}

/* DAC.FromCodes: sum_i out_proj_i(codebook_i[codes[:,i,:]])  (ResidualVectorQuantizer.cs:211-238) */
REF_API int ref_dac_from_codes(const ref_dac* m, const int64_t* codes, int64_t B, int nq, int64_t Tz, float* zq_out) {
    const ref_dac_config* c = &m->cfg;
    const int D = c->codebook_dim, LD = c->latent_dim;
    const int64_t nz = B * LD * Tz;
    float* zp = (float*)malloc(sizeof(float) * B * D * Tz);
    float* zqi = (float*)malloc(sizeof(float) * nz);
    int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * B * Tz);
    for (int64_t j = 0; j < nz; j++) zq_out[j] = 0.0f;
    for (int i = 0; i < nq; i++) {
        for (int64_t b = 0; b < B; b++) memcpy(idx + b * Tz, codes + (b * nq + i) * Tz, sizeof(int64_t) * Tz);
        ref_vq_gather(idx, B, D, Tz, m->codebook[i], zp);
        ref_conv1d(zp, B, D, Tz, m->out_proj[i].w, m->out_proj[i].b, LD, 1, 1, 0, 1, 1, NULL, zqi, Tz);
        for (int64_t j = 0; j < nz; j++) zq_out[j] = zq_out[j] + zqi[j];
    }
    free(zp); free(zqi); free(idx);
    return 0;
}

/* DAC.Decode(z) -> pcm [B,1,Tz*hop]  (Decoder.cs:22-58) */
REF_API int ref_dac_decode(const ref_dac* m, const float* z, int64_t B, int64_t Tz, float* pcm) {
    const ref_dac_config* c = &m->cfg;
    int C = c->decoder_dim;
    int64_t L = Tz;
    float* x = (float*)malloc(sizeof(float) * B * C * L);
    ref_conv1d(z, B, c->latent_dim, L, m->dec_in.w, m->dec_in.b, C, 7, 1, 3, 1, 1, NULL, x, L);
    for (int bi = 0; bi < c->n_dec_rates; bi++) {
        int s = c->dec_rates[bi], pad = (s + 1) / 2, Co = C / 2;
        int64_t Lo = (L - 1) * s - 2 * pad + 2 * s;
        float* sn = (float*)malloc(sizeof(float) * B * C * L);
        ref_snake(x, m->dec_blk[bi].a_up, B, C, L, sn);
        float* y = (float*)malloc(sizeof(float) * B * Co * Lo);
        ref_conv_transpose1d(sn, B, C, L, m->dec_blk[bi].up.w, m->dec_blk[bi].up.b, Co, 2 * s, s, pad, 0, y, Lo);
        free(sn); free(x);
        x = y; C = Co; L = Lo;
        for (int u = 0; u < 3; u++) x = res_unit(x, B, C, L, m->dec_blk[bi].a1[u], &m->dec_blk[bi].c7[u], m->dec_blk[bi].a2[u], &m->dec_blk[bi].c1[u], DIL[u]);
    }
    float* sn = (float*)malloc(sizeof(float) * B * C * L);
    ref_snake(x, m->dec_alpha_out, B, C, L, sn);
    float* y = (float*)malloc(sizeof(float) * B * L);
    ref_conv1d(sn, B, C, L, m->dec_out.w, m->dec_out.b, 1, 7, 1, 3, 1, 1, NULL, y, L);
    ref_tanh(y, B * L, pcm);
    free(sn); free(x); free(y);
    return 0;
}

REF_API int ref_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
