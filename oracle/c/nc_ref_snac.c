/* ORACLE (test infrastructure; never linked into or called by the product library).
 *
 * Plain-C CPU restatement of the reference's SNAC Encode / Decode path in the canonical arithmetic of DESIGN.md
 * (binary32 fma chains in ascending reduction order, ref_math.h elementary functions).  Pinned against the golden
 * vectors of oracle/torch_ref/snac.py (tests/golden/snac_*.npz); the reference itself has no tests ("parity unpinned").
 *
 * Reference functions restated (under /root/reference/NeuralCodecs.Torch/):
 *   Models/SNAC.cs:70-80 (Preprocess), :129-150 (Encode), :157-192 (Decode)
 *   Modules/SNAC/Encoder.cs:26-69, EncoderBlock.cs:27-55, ResidualUnit.cs:25-60, Decoder.cs:31-86, DecoderBlock.cs:29-70
 *   Modules/SNAC/WNConv1d.cs:120-143 and WNConvTranspose1d.cs:126-148: w = (v / ||v||) * (g - 1e-7)   (deviation D3)
 *   Modules/SNAC/NoiseBlock.cs:36-46 (noise is an input: deviation D8), Snake1d.cs:52-63
 *   Modules/SNAC/VectorQuantizer.cs:82-141, ResidualVectorQuantizer.cs:69-135
 *   Modules/SNAC/LocalMHA.cs:78-135, SinusoidalEmbedding.cs:67-80, RotaryEmbedding.cs:16-68
 *
 * Canonical definitions added here:
 *   avg_pool1d(s):   ((x0 + x1) + ... + x_{s-1}) / s                       (VectorQuantizer.cs:88)
 *   LayerNorm(C):    m = sum_f64(x)/C ; v = sum_f64((x-m)^2)/C ; r = (float)(1/sqrt(v + 1e-5))
 *                    y = ((x - (float)m) * r) * gamma + beta               (LocalMHA.cs:85)
 *   rotary tables:   f = fl32(pos * inv_freq[j%32]) ; cos = (float)cos((double)f), sin likewise
 *                    q' = (q * cos) + (rot_half(q) * sin)                  (RotaryEmbedding.cs:61-65, scale == 1)
 *   attention:       s_ij = chain_d(q_i[d], k_j[d]) * 0.125f ; e_j = exp(s_ij - max_j) ; p_j = e_j / (sum_j e_j, j ascending)
 *                    o[d] = chain_j(p_j, v_j[d])                           (LocalMHA.cs:105, non-causal, per 32-step window)
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nc_ref_internal.h"
#include "ref_math.h"

typedef struct {
    int sample_rate, encoder_dim, n_enc_rates, enc_rates[8], decoder_dim, n_dec_rates, dec_rates[8];
    int latent_dim, attn_window, codebook_size, codebook_dim, n_vq, vq_strides[8], noise, depthwise;
} ref_snac_config;

typedef struct {
    ref_snac_config cfg;
    int hop;
    uint8_t* blob_copy;
    ref_blob blob;
    int bad;
    int no_pad;   /* Encode(Tensor) as written: the encoder sees the un-padded clip (SNAC.cs:113-122) */
} ref_snac;

/* D3 fold; norm over all dims but 0 */
REF_API void ref_fold_wn_snac(const float* v, const float* g, int64_t d0, int64_t inner, float* w) {
    for (int64_t i = 0; i < d0; i++) {
        double ss = 0.0;
        for (int64_t j = 0; j < inner; j++) { float q = v[i * inner + j] * v[i * inner + j]; ss += (double)q; }
        const float norm = sqrtf((float)ss);
        const float gg = g[i] - 1e-7f;
        for (int64_t j = 0; j < inner; j++) w[i * inner + j] = (v[i * inner + j] / norm) * gg;
    }
}

typedef struct { float* w; const float* b; int d0, d1, k; } snac_conv;

static int get_conv(ref_snac* m, const char* prefix, snac_conv* c) {
    char nm[320];
    snprintf(nm, sizeof nm, "%s.parametrizations.weight.original1", prefix);
    const ref_tensor* v = ref_blob_find(&m->blob, nm);
    snprintf(nm, sizeof nm, "%s.parametrizations.weight.original0", prefix);
    const ref_tensor* g = ref_blob_find(&m->blob, nm);
    snprintf(nm, sizeof nm, "%s.bias", prefix);
    const ref_tensor* b = ref_blob_find(&m->blob, nm);
    if (!v || !g) { fprintf(stderr, "nc_ref_snac: missing %s\n", prefix); m->bad = 1; c->w = NULL; return -1; }
    c->d0 = (int)v->dims[0]; c->d1 = (int)v->dims[1]; c->k = (int)v->dims[2];
    c->w = (float*)malloc(sizeof(float) * c->d0 * c->d1 * c->k);
    ref_fold_wn_snac((const float*)v->data, (const float*)g->data, c->d0, (int64_t)c->d1 * c->k, c->w);
    c->b = b ? (const float*)b->data : NULL;
    return 0;
}

static const float* get_f(ref_snac* m, const char* name, int64_t want) {
    const ref_tensor* t = ref_blob_find(&m->blob, name);
    if (!t || (want > 0 && t->nbytes != want * 4)) { fprintf(stderr, "nc_ref_snac: missing/ill-sized %s\n", name); m->bad = 1; return NULL; }
    return (const float*)t->data;
}

REF_API ref_snac* ref_snac_create(const ref_snac_config* cfg, const uint8_t* blob, int64_t len) {
    ref_snac* m = (ref_snac*)calloc(1, sizeof(ref_snac));
    m->cfg = *cfg;
    m->blob_copy = (uint8_t*)malloc(len);
    memcpy(m->blob_copy, blob, len);
    if (ref_blob_parse(m->blob_copy, len, &m->blob) != 0) { free(m->blob_copy); free(m); return NULL; }
    m->hop = 1;
    for (int i = 0; i < cfg->n_enc_rates; i++) m->hop *= cfg->enc_rates[i];
    return m;
}
REF_API void ref_snac_destroy(ref_snac* m) {
    if (!m) return;
    free(m->blob.t); free(m->blob_copy); free(m);
}

static int64_t gcd64(int64_t a, int64_t b) { while (b) { int64_t t = a % b; a = b; b = t; } return a; }
REF_API int64_t ref_snac_padded_length(const ref_snac* m, int64_t T) {
    int64_t a = m->cfg.vq_strides[0], b = m->cfg.attn_window > 0 ? m->cfg.attn_window : 1;
    int64_t pad_to = m->hop * (a / gcd64(a, b) * b);
    return (T + pad_to - 1) / pad_to * pad_to;
}
/* pad = 1 (default): Encode(float[]) / forward semantics; pad = 0: Encode(Tensor) as written (Models/SNAC.cs:113-122, D7) */
REF_API void ref_snac_set_pad(ref_snac* m, int pad) { m->no_pad = !pad; }
/* frames of the encoder; -1 where the un-padded path makes the reference's quantizer / LocalMHA throw */
REF_API int64_t ref_snac_frames(const ref_snac* m, int64_t T) {
    if (!m->no_pad) return ref_snac_padded_length(m, T) / m->hop;
    int64_t L = T;
    for (int i = 0; i < m->cfg.n_enc_rates; i++) {
        const int s = m->cfg.enc_rates[i];
        L = (L + 2 * (int64_t)((s + 1) / 2) - 2 * (int64_t)s) / s + 1;          /* EncoderBlock.cs:46-53: k = 2s, stride s, pad ceil(s/2) */
        if (L <= 0) return -1;
    }
    for (int i = 0; i < m->cfg.n_vq; i++) if (L % m->cfg.vq_strides[i]) return -1;
    if (m->cfg.attn_window > 0 && L % m->cfg.attn_window) return -1;
    return L;
}

/* ---- layers ------------------------------------------------------------------------------------ */
static float* conv_apply(ref_snac* m, const char* prefix, float* x, int64_t B, int Cin, int64_t L, int stride, int pad, int dil,
                         int groups, const float* residual, int64_t* Lo, int* Cout, int free_x) {
    snac_conv c;
    if (get_conv(m, prefix, &c)) return x;
    const int64_t Lout = (L + 2 * (int64_t)pad - (int64_t)dil * (c.k - 1) - 1) / stride + 1;
    float* y = (float*)malloc(sizeof(float) * B * c.d0 * Lout);
    ref_conv1d(x, B, Cin, L, c.w, c.b, c.d0, c.k, stride, pad, dil, groups, residual, y, Lout);
    free(c.w);
    if (free_x) free(x);
    *Lo = Lout; *Cout = c.d0;
    return y;
}

static float* snake_apply(ref_snac* m, const char* name, float* x, int64_t B, int C, int64_t L, int free_x) {
    const float* a = get_f(m, name, C);
    float* y = (float*)malloc(sizeof(float) * B * C * L);
    if (a) ref_snake(x, a, B, C, L, y);
    if (free_x) free(x);
    return y;
}

/* x + conv1(snake(conv7_dil(snake(x))))   (ResidualUnit.cs:33-59) */
static float* res_unit(ref_snac* m, const char* p, float* x, int64_t B, int C, int64_t L, int dil, int groups) {
    char nm[320];
    int64_t Lo; int Co;
    snprintf(nm, sizeof nm, "%s.block.0.alpha", p);
    float* h = snake_apply(m, nm, x, B, C, L, 0);
    snprintf(nm, sizeof nm, "%s.block.1", p);
    h = conv_apply(m, nm, h, B, C, L, 1, 3 * dil, dil, groups, NULL, &Lo, &Co, 1);
    snprintf(nm, sizeof nm, "%s.block.2.alpha", p);
    h = snake_apply(m, nm, h, B, C, L, 1);
    snprintf(nm, sizeof nm, "%s.block.3", p);
    float* y = conv_apply(m, nm, h, B, C, L, 1, 0, 1, 1, x, &Lo, &Co, 1);
    free(x);
    return y;
}

/* LocalMHA.cs:78-115 on x [B,C,T]; returns a new buffer and frees x */
static float* local_mha(ref_snac* m, const char* p, float* x, int64_t B, int C, int64_t T) {
    char nm[320];
    const int W = m->cfg.attn_window, H = C / 64, NW = (int)(T / W);
    snprintf(nm, sizeof nm, "%s.norm.weight", p); const float* gam = get_f(m, nm, C);
    snprintf(nm, sizeof nm, "%s.norm.bias", p); const float* bet = get_f(m, nm, C);
    snprintf(nm, sizeof nm, "%s.to_qkv.weight", p); const float* wqkv = get_f(m, nm, (int64_t)3 * C * C);
    snprintf(nm, sizeof nm, "%s.to_out.weight", p); const float* wout = get_f(m, nm, (int64_t)C * C);
    snprintf(nm, sizeof nm, "%s.rel_pos.inv_freq", p); const float* invf = get_f(m, nm, 32);
    if (!gam || !bet || !wqkv || !wout || !invf) return x;
    float* xn = (float*)malloc(sizeof(float) * B * C * T);
    /* LayerNorm over channels, kept in [B,C,T] layout */
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t b = 0; b < B; b++)
        for (int64_t t = 0; t < T; t++) {
            const float* xp = x + b * C * T + t;
            double s1 = 0.0;
            for (int c = 0; c < C; c++) s1 += (double)xp[(int64_t)c * T];
            const double mu = s1 / C;
            double s2 = 0.0;
            for (int c = 0; c < C; c++) { double d = (double)xp[(int64_t)c * T] - mu; s2 += d * d; }
            const float r = (float)(1.0 / sqrt(s2 / C + 1e-5));
            const float muf = (float)mu;
            for (int c = 0; c < C; c++) xn[(b * C + c) * T + t] = ((xp[(int64_t)c * T] - muf) * r) * gam[c] + bet[c];
        }
    /* qkv = Linear(C -> 3C, no bias) == 1x1 conv with weight [3C, C, 1] */
    float* qkv = (float*)malloc(sizeof(float) * B * 3 * C * T);
    ref_conv1d(xn, B, C, T, wqkv, NULL, 3 * C, 1, 1, 0, 1, 1, NULL, qkv, T);
    /* rotary tables [W][64] */
    float* ct = (float*)malloc(sizeof(float) * W * 64);
    float* st = (float*)malloc(sizeof(float) * W * 64);
    for (int i = 0; i < W; i++)
        for (int j = 0; j < 64; j++) {
            const float f = (float)i * invf[j & 31];
            ct[i * 64 + j] = (float)cos((double)f);
            st[i * 64 + j] = (float)sin((double)f);
        }
    float* att = xn; /* reuse as the attention output [B,C,T] (channel = head*64 + d) */
#pragma omp parallel for collapse(3) schedule(static)
    for (int64_t b = 0; b < B; b++)
        for (int h = 0; h < H; h++)
            for (int wdx = 0; wdx < NW; wdx++) {
                float q[32][64], k[32][64], v[32][64];
                if (W > 32) continue; /* window sizes above 32 are not on the path */
                for (int i = 0; i < W; i++)
                    for (int d = 0; d < 64; d++) {
                        const int64_t t = (int64_t)wdx * W + i;
                        const float* base = qkv + b * 3 * C * T;
                        const float qv = base[((int64_t)(h * 64 + d)) * T + t], qr = base[((int64_t)(h * 64 + (d < 32 ? d + 32 : d - 32))) * T + t];
                        const float kv = base[((int64_t)(C + h * 64 + d)) * T + t], kr = base[((int64_t)(C + h * 64 + (d < 32 ? d + 32 : d - 32))) * T + t];
                        const float rq = d < 32 ? -qr : qr, rk = d < 32 ? -kr : kr;
                        q[i][d] = (qv * ct[i * 64 + d]) + (rq * st[i * 64 + d]);
                        k[i][d] = (kv * ct[i * 64 + d]) + (rk * st[i * 64 + d]);
                        v[i][d] = base[((int64_t)(2 * C + h * 64 + d)) * T + t];
                    }
                for (int i = 0; i < W; i++) {
                    float s[32], mx = -INFINITY;
                    for (int j = 0; j < W; j++) {
                        float a = 0.0f;
                        for (int d = 0; d < 64; d++) a = fmaf(q[i][d], k[j][d], a);
                        s[j] = a * 0.125f;
                        if (s[j] > mx) mx = s[j];
                    }
                    float sum = 0.0f;
                    for (int j = 0; j < W; j++) { s[j] = ref_expf(s[j] - mx); sum = sum + s[j]; }
                    for (int j = 0; j < W; j++) s[j] = s[j] / sum;
                    for (int d = 0; d < 64; d++) {
                        float a = 0.0f;
                        for (int j = 0; j < W; j++) a = fmaf(s[j], v[j][d], a);
                        att[(b * C + h * 64 + d) * T + (int64_t)wdx * W + i] = a;
                    }
                }
            }
    free(qkv); free(ct); free(st);
    /* out = Linear(C -> C, no bias)(att) + residual */
    float* y = (float*)malloc(sizeof(float) * B * C * T);
    ref_conv1d(att, B, C, T, wout, NULL, C, 1, 1, 0, 1, 1, x, y, T);
    free(att); free(x);
    return y;
}

/* Encoder.forward on the right-zero-padded clip -> z [B, latent, T'] */
static float* snac_encoder(ref_snac* m, const float* pcm, int64_t B, int64_t T, int64_t* Tz) {
    const ref_snac_config* c = &m->cfg;
    char nm[320];
    const int64_t Tp = m->no_pad ? T : ref_snac_padded_length(m, T);
    float* x = (float*)calloc(B * Tp, sizeof(float));
    for (int64_t b = 0; b < B; b++) memcpy(x + b * Tp, pcm + b * T, sizeof(float) * T);
    int C = 1; int64_t L = Tp;
    x = conv_apply(m, "encoder.block.0", x, B, 1, L, 1, 3, 1, 1, NULL, &L, &C, 1);
    static const int DIL[3] = {1, 3, 9};
    for (int bi = 0; bi < c->n_enc_rates; bi++) {
        const int s = c->enc_rates[bi];
        for (int u = 0; u < 3; u++) {
            snprintf(nm, sizeof nm, "encoder.block.%d.block.%d", bi + 1, u);
            x = res_unit(m, nm, x, B, C, L, DIL[u], c->depthwise ? C : 1);
        }
        snprintf(nm, sizeof nm, "encoder.block.%d.block.3.alpha", bi + 1);
        x = snake_apply(m, nm, x, B, C, L, 1);
        snprintf(nm, sizeof nm, "encoder.block.%d.block.4", bi + 1);
        x = conv_apply(m, nm, x, B, C, L, s, (s + 1) / 2, 1, 1, NULL, &L, &C, 1);
    }
    int n = c->n_enc_rates + 1;
    if (c->attn_window > 0) { snprintf(nm, sizeof nm, "encoder.block.%d", n); x = local_mha(m, nm, x, B, C, L); n++; }
    snprintf(nm, sizeof nm, "encoder.block.%d", n);
    x = conv_apply(m, nm, x, B, C, L, 1, 3, 1, c->depthwise ? C : 1, NULL, &L, &C, 1);
    *Tz = L;
    return x;
}

/* SNAC.Encode: codes_concat [B, sum_i T'/s_i] (levels side by side), zq/z nullable [B, latent, T'] */
REF_API int ref_snac_encode(ref_snac* m, const float* pcm, int64_t B, int64_t T, int64_t* codes_concat, float* zq_out, float* z_out) {
    const ref_snac_config* c = &m->cfg;
    char nm[320];
    m->bad = 0;
    if (ref_snac_frames(m, T) < 0) return 2;   /* the reference throws on this length */
    int64_t Tz;
    float* z = snac_encoder(m, pcm, B, T, &Tz);
    const int LD = c->latent_dim, D = c->codebook_dim;
    const int64_t nz = B * LD * Tz;
    if (z_out) memcpy(z_out, z, sizeof(float) * nz);
    float* residual = z;
    float* zq = (float*)calloc(nz, sizeof(float));
    int64_t total = 0;
    for (int i = 0; i < c->n_vq; i++) total += Tz / c->vq_strides[i];
    int64_t off = 0;
    for (int i = 0; i < c->n_vq; i++) {
        const int s = c->vq_strides[i];
        const int64_t Ts = Tz / s;
        float* pooled = residual;
        if (s > 1) {
            pooled = (float*)malloc(sizeof(float) * B * LD * Ts);
            for (int64_t r = 0; r < B * LD; r++)
                for (int64_t t = 0; t < Ts; t++) {
                    float a = residual[r * Tz + t * s];
                    for (int j = 1; j < s; j++) a = a + residual[r * Tz + t * s + j];
                    pooled[r * Ts + t] = a / (float)s;
                }
        }
        int64_t Lo; int Co;
        snprintf(nm, sizeof nm, "quantizer.quantizers.%d.in_proj", i);
        float* ze = conv_apply(m, nm, pooled, B, LD, Ts, 1, 0, 1, 1, NULL, &Lo, &Co, 0);
        if (s > 1) free(pooled);
        snprintf(nm, sizeof nm, "quantizer.quantizers.%d.codebook.weight", i);
        const float* cb = get_f(m, nm, (int64_t)c->codebook_size * D);
        if (m->bad) { free(ze); break; }
        int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * B * Ts);
        float* st = (float*)malloc(sizeof(float) * B * D * Ts);
        ref_vq_argmin(ze, B, D, Ts, cb, c->codebook_size, idx, st, NULL);
        snprintf(nm, sizeof nm, "quantizer.quantizers.%d.out_proj", i);
        float* q = conv_apply(m, nm, st, B, D, Ts, 1, 0, 1, 1, NULL, &Lo, &Co, 1);
        for (int64_t r = 0; r < B * LD; r++)
            for (int64_t t = 0; t < Tz; t++) {
                const float qv = q[r * Ts + t / s];                          /* repeat_interleave(s) */
                zq[r * Tz + t] = zq[r * Tz + t] + qv;
                residual[r * Tz + t] = residual[r * Tz + t] - qv;
            }
        for (int64_t b = 0; b < B; b++) memcpy(codes_concat + b * total + off, idx + b * Ts, sizeof(int64_t) * Ts);
        off += Ts;
        free(idx); free(q); free(ze);
    }
    if (zq_out) memcpy(zq_out, zq, sizeof(float) * nz);
    free(residual); free(zq);
    return m->bad ? -1 : 0;
}

/* ResidualVectorQuantizer.FromCodes (:100-135) */
REF_API int ref_snac_from_codes(ref_snac* m, const int64_t* codes_concat, int64_t B, int64_t Tz, float* zq_out) {
    const ref_snac_config* c = &m->cfg;
    char nm[320];
    m->bad = 0;
    const int LD = c->latent_dim, D = c->codebook_dim;
    int64_t total = 0;
    for (int i = 0; i < c->n_vq; i++) total += Tz / c->vq_strides[i];
    int64_t off = 0;
    for (int i = 0; i < c->n_vq; i++) {
        const int s = c->vq_strides[i];
        const int64_t Ts = Tz / s;
        snprintf(nm, sizeof nm, "quantizer.quantizers.%d.codebook.weight", i);
        const float* cb = get_f(m, nm, (int64_t)c->codebook_size * D);
        if (m->bad) return -1;
        int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * B * Ts);
        for (int64_t b = 0; b < B; b++) memcpy(idx + b * Ts, codes_concat + b * total + off, sizeof(int64_t) * Ts);
        float* zp = (float*)malloc(sizeof(float) * B * D * Ts);
        ref_vq_gather(idx, B, D, Ts, cb, zp);
        int64_t Lo; int Co;
        snprintf(nm, sizeof nm, "quantizer.quantizers.%d.out_proj", i);
        float* q = conv_apply(m, nm, zp, B, D, Ts, 1, 0, 1, 1, NULL, &Lo, &Co, 1);
        for (int64_t r = 0; r < B * LD; r++)
            for (int64_t t = 0; t < Tz; t++) {
                const float qv = q[r * Ts + t / s];
                zq_out[r * Tz + t] = (i == 0) ? qv : zq_out[r * Tz + t] + qv;
            }
        off += Ts;
        free(idx); free(q);
    }
    return m->bad ? -1 : 0;
}

REF_API int64_t ref_snac_decoded_length(const ref_snac* m, int64_t Tz) {
    int64_t L = Tz;
    for (int i = 0; i < m->cfg.n_dec_rates; i++) {
        const int s = m->cfg.dec_rates[i];
        L = (L - 1) * s - 2 * ((s + 1) / 2) + 2 * s + (s % 2);
    }
    return L;
}

/* Decoder.forward(zq) with the NoiseBlock inputs supplied: noise = concatenation over decoder blocks of [B,1,T_i] */
REF_API int ref_snac_decode(ref_snac* m, const float* zq, int64_t B, int64_t Tz, const float* noise, float* pcm) {
    const ref_snac_config* c = &m->cfg;
    char nm[320];
    m->bad = 0;
    int C = c->latent_dim; int64_t L = Tz;
    float* x;
    int n;
    if (c->depthwise) {
        x = conv_apply(m, "decoder.model.0", (float*)zq, B, C, L, 1, 3, 1, C, NULL, &L, &C, 0);
        x = conv_apply(m, "decoder.model.1", x, B, C, L, 1, 0, 1, 1, NULL, &L, &C, 1);
        n = 2;
    } else {
        x = conv_apply(m, "decoder.model.0", (float*)zq, B, C, L, 1, 3, 1, 1, NULL, &L, &C, 0);
        n = 1;
    }
    if (c->attn_window > 0) { snprintf(nm, sizeof nm, "decoder.model.%d", n); x = local_mha(m, nm, x, B, C, L); n++; }
    static const int DIL[3] = {1, 3, 9};
    int64_t noff = 0;
    for (int bi = 0; bi < c->n_dec_rates && !m->bad; bi++) {
        const int s = c->dec_rates[bi], pad = (s + 1) / 2, Co = C / 2;
        snprintf(nm, sizeof nm, "decoder.model.%d.block.0.alpha", n);
        x = snake_apply(m, nm, x, B, C, L, 1);
        snprintf(nm, sizeof nm, "decoder.model.%d.block.1", n);
        snac_conv ct;
        if (get_conv(m, nm, &ct)) break;
        const int64_t Lo = (L - 1) * s - 2 * pad + 2 * s + (s % 2);
        float* y = (float*)malloc(sizeof(float) * B * Co * Lo);
        ref_conv_transpose1d(x, B, C, L, ct.w, ct.b, Co, 2 * s, s, pad, s % 2, y, Lo);
        free(ct.w); free(x);
        x = y; C = Co; L = Lo;
        int k = 2;
        if (c->noise) {
            int64_t L2; int C2;
            snprintf(nm, sizeof nm, "decoder.model.%d.block.2.linear", n);
            float* h = conv_apply(m, nm, x, B, C, L, 1, 0, 1, 1, NULL, &L2, &C2, 0);
            for (int64_t b = 0; b < B; b++)
                for (int ch = 0; ch < C; ch++)
                    for (int64_t t = 0; t < L; t++) {
                        const int64_t o = (b * C + ch) * L + t;
                        const float nv = noise[noff + b * L + t] * h[o];
                        x[o] = x[o] + nv;
                    }
            free(h);
            noff += B * L;
            k = 3;
        }
        for (int u = 0; u < 3; u++) {
            snprintf(nm, sizeof nm, "decoder.model.%d.block.%d", n, k + u);
            x = res_unit(m, nm, x, B, C, L, DIL[u], c->depthwise ? C : 1);
        }
        n++;
    }
    snprintf(nm, sizeof nm, "decoder.model.%d.alpha", n);
    x = snake_apply(m, nm, x, B, C, L, 1);
    snprintf(nm, sizeof nm, "decoder.model.%d", n + 1);
    int64_t Lo = 0; int Co = 0;   // (set by conv_apply)
    float* y = conv_apply(m, nm, x, B, C, L, 1, 3, 1, 1, NULL, &Lo, &Co, 1);
    ref_tanh(y, B * Lo, pcm);
    free(y);
    return m->bad ? -1 : 0;
}
