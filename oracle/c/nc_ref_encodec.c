/* ORACLE (test infrastructure; never linked into or called by the product library).
 *
 * Plain-C CPU restatement of the reference's Encodec Encode / Decode path in the canonical arithmetic of DESIGN.md.
 * Pinned against the golden vectors of oracle/torch_ref/encodec.py (tests/golden/encodec_*.npz); the reference itself has no
 * tests ("parity unpinned").
 *
 * Reference functions restated (under /root/reference/NeuralCodecs.Torch/):
 *   Models/Encodec.cs:436-455 (DecodeFrame), :457-489 (EncodeFrame, RMS normalise)
 *   Modules/Encodec/SEANetEncoder.cs:37-148, SEANetDecoder.cs:40-153, SEANetResnetBlock.cs:29-85
 *   Modules/Encodec/SConv1d.cs:144-173,245-274 (asymmetric reflect pad, extra pad, small-input path D9)
 *   Modules/Encodec/SConvTranspose1d.cs:116-171, NormConv1d.cs:122-164 (GroupNorm(1,C) after the conv, before the trim)
 *   Modules/Encodec/WNConv1d.cs:110-126, WNConvTranspose1d.cs:120-150 (D3 fold), SLSTM.cs:40-57
 *   Modules/Encodec/ResidualVectorQuantizer.cs:107-157, EuclideanCodebook.cs:82-85,155-182
 *   AudioTools/AudioTensorDSP.cs:161-261 (LinearOverlapAdd), Utils/TorchUtils.cs:26-30 (ELU, alpha = 1)
 *
 * Canonical definitions added here:
 *   GroupNorm(1,C):  S1 = sum x, S2 = sum x*x in binary64, summed hierarchically over 32x32 blocks in the accumulator layout of the
 *                    fp32 matrix-core instruction (gn_sums / block_sums below: 16-row slot sums, xor butterfly, then the block sums of a
 *                    sample by 64 strided slots + butterfly).  mu = S1/N, var = max(S2/N - mu*mu, 0), r = (float)(1/sqrt(var + 1e-5)),
 *                    y = ((x - (float)mu) * r) * gamma[c] + beta[c]
 *   ELU:             x > 0 ? x : exp(x) - 1
 *   LSTM cell:       pre = (chain_ih + b_ih) + (chain_hh + b_hh); chain_ih one fma chain (k ascending), chain_hh four quarter chains
 *                    combined as (q0 + q1) + (q2 + q3); sigmoid(x) = 1/(1 + exp(-x));
 *                    c = (f*c) + (i*g); h = o * tanh(c); gate order i, f, g, o
 *   RMS scale:       mono = (sum_c x)/C; vol = sqrtf((float)(S/T)), S = hierarchical binary64 sum of fl32(mono*mono);
 *                    scale = vol + 1e-8f; x / scale
 *   overlap-add:     t_i = (float)((double)(i+1)/(L0+1)), w_i = 0.5f - |t_i - 0.5f|; out = (sum_frames f*w) / (sum_frames w)
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nc_ref_internal.h"
#include "ref_math.h"

typedef struct {
    int sample_rate, channels, dimension, n_filters, n_ratios, ratios[8], lstm_layers, compress;
    int kernel_size, last_kernel_size, residual_kernel_size, group_norm, causal, normalize, codebook_size, n_q_total;
} ref_encodec_config;

typedef struct {
    ref_encodec_config cfg;
    uint8_t* blob_copy;
    ref_blob blob;
    int bad;
} ref_encodec;

REF_API ref_encodec* ref_encodec_create(const ref_encodec_config* cfg, const uint8_t* blob, int64_t len) {
    ref_encodec* m = (ref_encodec*)calloc(1, sizeof(ref_encodec));
    m->cfg = *cfg;
    m->blob_copy = (uint8_t*)malloc(len);
    memcpy(m->blob_copy, blob, len);
    if (ref_blob_parse(m->blob_copy, len, &m->blob) != 0) { free(m->blob_copy); free(m); return NULL; }
    return m;
}
REF_API void ref_encodec_destroy(ref_encodec* m) {
    if (!m) return;
    free(m->blob.t); free(m->blob_copy); free(m);
}

static const float* getf(ref_encodec* m, const char* name, int64_t want) {
    const ref_tensor* t = ref_blob_find(&m->blob, name);
    if (!t || (want > 0 && t->nbytes != want * 4)) { fprintf(stderr, "nc_ref_encodec: missing/ill-sized %s\n", name); m->bad = 1; return NULL; }
    return (const float*)t->data;
}

void ref_fold_wn_snac(const float* v, const float* g, int64_t d0, int64_t inner, float* w);

/* dense weight of layer `key` ([d0,d1,k]); *owned is set when the caller must free it (weight-norm fold) */
static const float* get_weight(ref_encodec* m, const char* key, int* d0, int* d1, int* k, int* owned) {
    char nm[320];
    snprintf(nm, sizeof nm, "%s.conv.weight", key);
    const ref_tensor* w = ref_blob_find(&m->blob, nm);
    *owned = 0;
    if (w) { *d0 = (int)w->dims[0]; *d1 = (int)w->dims[1]; *k = (int)w->dims[2]; return (const float*)w->data; }
    snprintf(nm, sizeof nm, "%s.conv.weight_v", key);
    const ref_tensor* v = ref_blob_find(&m->blob, nm);
    snprintf(nm, sizeof nm, "%s.conv.weight_g", key);
    const ref_tensor* g = ref_blob_find(&m->blob, nm);
    if (!v || !g) { fprintf(stderr, "nc_ref_encodec: missing %s\n", key); m->bad = 1; return NULL; }
    *d0 = (int)v->dims[0]; *d1 = (int)v->dims[1]; *k = (int)v->dims[2];
    float* f = (float*)malloc(sizeof(float) * v->dims[0] * v->dims[1] * v->dims[2]);
    ref_fold_wn_snac((const float*)v->data, (const float*)g->data, v->dims[0], v->dims[1] * v->dims[2], f);
    *owned = 1;
    return f;
}

/* ---- canonical reductions --------------------------------------------------------------------------- */
#define GN_CHUNK 256     /* RMS-scale chunks */
/* xor butterfly 1,2,4,8,16,32 over 64 slot sums (p_i <- p_i + p_{i^off}; addition commutes, so every slot ends with the same value):
 * a fixed tree of the width of a CDNA wavefront */
static void butterfly64(double* p1, double* p2) {
    double q1[64], q2[64];
    for (int off = 1; off <= 32; off <<= 1) {
        for (int i = 0; i < 64; i++) { q1[i] = p1[i] + p1[i ^ off]; q2[i] = p2[i] + p2[i ^ off]; }
        memcpy(p1, q1, sizeof q1); memcpy(p2, q2, sizeof q2);
    }
}

/* GroupNorm(1,C) sums of one sample y [C][T], canonical order.  The tensor is viewed as the matrix the matrix-core kernels emit it from:
 * rows R = c*sub + (t % sub), columns q = t / sub, with sub = 1 for every convolution and sub = stride behind a stride-2/4/8 transposed
 * convolution with k = 2*stride (its sub-pixel form: rows are (channel, phase) pairs).  The matrix is cut into 32x32 blocks (row block rb,
 * column block cb, aligned at row 0 / column 0; elements outside the tensor count as +0).  Inside a block, slot i = 32*h + c (h = 0,1;
 * c = 0..31) adds the 16 elements of column 32*cb + c in rows 32*rb + 8*j + 4*h + k (j = 0..3 outer, k = 0..3 inner: ascending rows),
 * binary64 from +0 (squares are exact in binary64), and the 64 slot sums meet in the xor butterfly 1,2,4,8,16,32 -- the accumulator layout
 * of v_mfma_f32_32x32x2_f32, so a wavefront reduces the block it has just computed without moving a value.  The block sums of a sample,
 * listed as idx = rb*ncb + cb, are then added by 64 slots (slot i: idx = i, i+64, ... ascending) and one more butterfly. */
static int gn_sub_for(int k, int stride, int cout) {
    return ((stride == 2 || stride == 4 || stride == 8) && k == 2 * stride && (cout * stride) % 32 == 0) ? stride : 1;
}
static void block_sums(const float* y, int C, int64_t T, int sub, int64_t rb, int64_t cb, double* s1_out, double* s2_out) {
    double p1[64], p2[64];
    for (int i = 0; i < 64; i++) {
        const int h = i >> 5, c = i & 31;
        const int64_t q = cb * 32 + c;
        double a = 0.0, b = 0.0;
        for (int r = 0; r < 16; r++) {
            const int64_t R = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int64_t co = R / sub, t = q * sub + R % sub;
            if (co < C && t < T) { const double v = (double)y[co * T + t]; a += v; b += v * v; }
        }
        p1[i] = a; p2[i] = b;
    }
    butterfly64(p1, p2);
    *s1_out = p1[0]; *s2_out = p2[0];
}
static void gn_sums(const float* y, int C, int64_t T, int sub, double* s1_out, double* s2_out) {
    const int64_t nrb = ((int64_t)C * sub + 31) / 32, ncb = ((T + sub - 1) / sub + 31) / 32, n = nrb * ncb;
    double* part = (double*)malloc(sizeof(double) * 2 * (size_t)n);
    for (int64_t rb = 0; rb < nrb; rb++)
        for (int64_t cb = 0; cb < ncb; cb++) block_sums(y, C, T, sub, rb, cb, &part[2 * (rb * ncb + cb)], &part[2 * (rb * ncb + cb) + 1]);
    double p1[64], p2[64];
    for (int i = 0; i < 64; i++) {
        double a = 0.0, b = 0.0;
        for (int64_t k = i; k < n; k += 64) { a += part[2 * k]; b += part[2 * k + 1]; }
        p1[i] = a; p2[i] = b;
    }
    butterfly64(p1, p2);
    free(part);
    *s1_out = p1[0]; *s2_out = p2[0];
}

/* GroupNorm(1,C) over x [B,C,T] (NormConv1d.cs:155); `sub` selects the canonical block view (gn_sums) */
REF_API void ref_group_norm1(const float* x, int64_t B, int C, int64_t T, int sub, const float* gamma, const float* beta, float* y) {
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; b++) {
        double s1, s2;
        gn_sums(x + b * C * T, C, T, sub < 1 ? 1 : sub, &s1, &s2);
        const double N = (double)C * (double)T;
        const double mu = s1 / N;
        double var = s2 / N - mu * mu;
        if (var < 0.0) var = 0.0;
        const float r = (float)(1.0 / sqrt(var + 1e-5));
        const float muf = (float)mu;
        for (int c = 0; c < C; c++)
            for (int64_t t = 0; t < T; t++) {
                const int64_t o = (b * C + c) * T + t;
                y[o] = ((x[o] - muf) * r) * gamma[c] + beta[c];
            }
    }
}

static inline float eluf(float x) { return x > 0.0f ? x : ref_expf(x) - 1.0f; }
static inline float sigmoidf(float x) { return 1.0f / (1.0f + ref_expf(-x)); }

/* ---- SConv1d / SConvTranspose1d ----------------------------------------------------------------------- */
typedef struct { int64_t left, right, Lz, Lout; } pad_plan;   /* Lz = length after the small-input zero pad (D9) */

static pad_plan plan_sconv(int64_t L, int k, int stride, int dil, int causal) {
    pad_plan p;
    const int64_t eff = (int64_t)(k - 1) * dil + 1, pt = eff - stride;
    const float nf = (float)(L - eff + pt) / (float)stride + 1.0f;                    /* SConv1d.cs:245-250, float division */
    const int64_t ideal = ((int64_t)ceilf(nf) - 1) * stride + (eff - pt);
    const int64_t extra = ideal - L;
    if (causal) { p.left = pt; p.right = extra; }
    else { const int64_t r = pt / 2; p.left = pt - r; p.right = r + extra; }
    const int64_t mx = p.left > p.right ? p.left : p.right;
    p.Lz = L <= mx ? L + (mx - L + 1) : L;                                             /* SConv1d.cs:258-274 */
    p.Lout = (p.Lz + p.left + p.right - eff) / stride + 1;
    return p;
}

/* y = [GroupNorm](conv(reflect_pad(x)));  x [B,Cin,L] (already activated by the caller); returns malloc'd [B,Cout,Lout] */
static float* sconv(ref_encodec* m, const char* key, const float* x, int64_t B, int Cin, int64_t L, int k_expect, int stride, int dil,
                    int* Cout, int64_t* Lout) {
    int d0, d1, k, owned;
    const float* w = get_weight(m, key, &d0, &d1, &k, &owned);
    char nm[320];
    snprintf(nm, sizeof nm, "%s.conv.bias", key);
    const ref_tensor* bt = ref_blob_find(&m->blob, nm);
    if (!w || d1 != Cin || k != k_expect) { if (w && owned) free((void*)w); m->bad = 1; *Cout = Cin; *Lout = L; return (float*)calloc(B * Cin * L, 4); }
    const pad_plan p = plan_sconv(L, k, stride, dil, m->cfg.causal);
    const int64_t Lp = p.Lz + p.left + p.right;
    float* xp = (float*)malloc(sizeof(float) * B * Cin * Lp);
    for (int64_t r = 0; r < B * Cin; r++)
        for (int64_t j = 0; j < Lp; j++) {
            int64_t q = j - p.left;                                                       /* position in the (zero-extended) row */
            if (q < 0) q = -q;
            if (q >= p.Lz) q = 2 * (p.Lz - 1) - q;
            xp[r * Lp + j] = q < L ? x[r * L + q] : 0.0f;
        }
    float* y = (float*)malloc(sizeof(float) * B * d0 * p.Lout);
    ref_conv1d(xp, B, Cin, Lp, w, bt ? (const float*)bt->data : NULL, d0, k, stride, 0, dil, 1, NULL, y, p.Lout);
    free(xp);
    if (owned) free((void*)w);
    if (m->cfg.group_norm) {
        snprintf(nm, sizeof nm, "%s.norm.weight", key); const float* g = getf(m, nm, d0);
        snprintf(nm, sizeof nm, "%s.norm.bias", key); const float* b = getf(m, nm, d0);
        if (g && b) ref_group_norm1(y, B, d0, p.Lout, 1, g, b, y);
    }
    *Cout = d0; *Lout = p.Lout;
    return y;
}

static float* sconvT(ref_encodec* m, const char* key, const float* x, int64_t B, int Cin, int64_t L, int stride, int* Cout, int64_t* Lout) {
    int d0, d1, k, owned;
    const float* w = get_weight(m, key, &d0, &d1, &k, &owned);
    char nm[320];
    snprintf(nm, sizeof nm, "%s.conv.bias", key);
    const ref_tensor* bt = ref_blob_find(&m->blob, nm);
    if (!w || d0 != Cin) { if (w && owned) free((void*)w); m->bad = 1; *Cout = Cin; *Lout = L; return (float*)calloc(B * Cin * L, 4); }
    const int64_t Lfull = (L - 1) * stride + k;
    float* y = (float*)malloc(sizeof(float) * B * d1 * Lfull);
    ref_conv_transpose1d(x, B, Cin, L, w, bt ? (const float*)bt->data : NULL, d1, k, stride, 0, 0, y, Lfull);
    if (owned) free((void*)w);
    if (m->cfg.group_norm) {
        snprintf(nm, sizeof nm, "%s.norm.weight", key); const float* g = getf(m, nm, d1);
        snprintf(nm, sizeof nm, "%s.norm.bias", key); const float* b = getf(m, nm, d1);
        if (g && b) ref_group_norm1(y, B, d1, Lfull, gn_sub_for(k, stride, d1), g, b, y);
    }
    const int64_t pt = k - stride;
    int64_t right, left;
    if (m->cfg.causal) { right = pt; left = 0; }                                       /* trimRightRatio = 1 */
    else { right = pt / 2; left = pt - right; }
    const int64_t Lt = Lfull - left - right;
    float* o = (float*)malloc(sizeof(float) * B * d1 * Lt);
    for (int64_t r = 0; r < B * d1; r++) memcpy(o + r * Lt, y + r * Lfull + left, sizeof(float) * Lt);
    free(y);
    *Cout = d1; *Lout = Lt;
    return o;
}

static float* elu_new(const float* x, int64_t n) {
    float* y = (float*)malloc(sizeof(float) * n);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) y[i] = eluf(x[i]);
    return y;
}

/* SEANetResnetBlock.cs:53-85: shortcut(x) + conv1(elu(conv3(elu(x)))); frees x */
static float* resblock(ref_encodec* m, const char* key, float* x, int64_t B, int C, int64_t L) {
    char nm[320];
    int Co; int64_t Lo;
    snprintf(nm, sizeof nm, "%s.shortcut", key);
    float* s = sconv(m, nm, x, B, C, L, 1, 1, 1, &Co, &Lo);
    float* a = elu_new(x, B * C * L);
    snprintf(nm, sizeof nm, "%s.block.1", key);
    int Ch; int64_t Lh;
    float* h = sconv(m, nm, a, B, C, L, m->cfg.residual_kernel_size, 1, 1, &Ch, &Lh);
    free(a);
    a = elu_new(h, B * Ch * Lh);
    free(h);
    snprintf(nm, sizeof nm, "%s.block.3", key);
    float* y = sconv(m, nm, a, B, Ch, Lh, 1, 1, 1, &Co, &Lo);
    free(a);
    if (Lo != L || Co != C) { m->bad = 1; free(s); free(x); return y; }   /* degenerate: block branch longer than the shortcut (D9) */
    for (int64_t i = 0; i < B * C * L; i++) y[i] = s[i] + y[i];
    free(s); free(x);
    return y;
}

/* SLSTM.cs:40-57 on x [B,C,T]; frees x */
static float* slstm(ref_encodec* m, const char* key, float* x, int64_t B, int C, int64_t T) {
    char nm[320];
    float* in = (float*)malloc(sizeof(float) * T * B * C);   /* [T][B][C] */
    for (int64_t b = 0; b < B; b++) for (int c = 0; c < C; c++) for (int64_t t = 0; t < T; t++) in[(t * B + b) * C + c] = x[(b * C + c) * T + t];
    float* cur = (float*)malloc(sizeof(float) * T * B * C);
    memcpy(cur, in, sizeof(float) * T * B * C);
    float* gi = (float*)malloc(sizeof(float) * B * 4 * C);
    for (int l = 0; l < m->cfg.lstm_layers; l++) {
        snprintf(nm, sizeof nm, "%s.lstm.weight_ih_l%d", key, l); const float* wih = getf(m, nm, (int64_t)4 * C * C);
        snprintf(nm, sizeof nm, "%s.lstm.weight_hh_l%d", key, l); const float* whh = getf(m, nm, (int64_t)4 * C * C);
        snprintf(nm, sizeof nm, "%s.lstm.bias_ih_l%d", key, l); const float* bih = getf(m, nm, 4 * C);
        snprintf(nm, sizeof nm, "%s.lstm.bias_hh_l%d", key, l); const float* bhh = getf(m, nm, 4 * C);
        if (!wih || !whh || !bih || !bhh) break;
        float* h = (float*)calloc(B * C, sizeof(float));
        float* cs = (float*)calloc(B * C, sizeof(float));
        float* out = (float*)malloc(sizeof(float) * T * B * C);
        for (int64_t t = 0; t < T; t++) {
#pragma omp parallel for collapse(2) schedule(static)
            for (int64_t b = 0; b < B; b++)
                for (int j = 0; j < 4 * C; j++) {
                    const float* xv = cur + (t * B + b) * C;
                    const float* hv = h + b * C;
                    float a = 0.0f;
                    for (int k = 0; k < C; k++) a = fmaf(wih[(int64_t)j * C + k], xv[k], a);
                    /* recurrent contraction: FOUR quarter chains (k ascending inside a quarter, each from +0) combined as
                     * (q0 + q1) + (q2 + q3) -- the dependent chain is the critical path of every time step on the device, and four
                     * wavefronts walk the quarters side by side (C % 4 == 0; otherwise one chain) */
                    float r;
                    if (C % 4 == 0) {
                        float q[4];
                        for (int s4 = 0; s4 < 4; s4++) {
                            float c = 0.0f;
                            for (int k = s4 * (C / 4); k < (s4 + 1) * (C / 4); k++) c = fmaf(whh[(int64_t)j * C + k], hv[k], c);
                            q[s4] = c;
                        }
                        r = (q[0] + q[1]) + (q[2] + q[3]);
                    } else {
                        r = 0.0f;
                        for (int k = 0; k < C; k++) r = fmaf(whh[(int64_t)j * C + k], hv[k], r);
                    }
                    gi[b * 4 * C + j] = (a + bih[j]) + (r + bhh[j]);
                }
            for (int64_t b = 0; b < B; b++)
                for (int j = 0; j < C; j++) {
                    const float* g = gi + b * 4 * C;
                    const float ig = sigmoidf(g[j]), fg = sigmoidf(g[C + j]), gg = ref_tanhf(g[2 * C + j]), og = sigmoidf(g[3 * C + j]);
                    const float cn = (fg * cs[b * C + j]) + (ig * gg);
                    cs[b * C + j] = cn;
                    out[(t * B + b) * C + j] = og * ref_tanhf(cn);
                }
            memcpy(h, out + t * B * C, sizeof(float) * B * C);
        }
        free(h); free(cs); free(cur);
        cur = out;
    }
    free(gi);
    for (int64_t b = 0; b < B; b++) for (int c = 0; c < C; c++) for (int64_t t = 0; t < T; t++)
        x[(b * C + c) * T + t] = cur[(t * B + b) * C + c] + in[(t * B + b) * C + c];
    free(cur); free(in);
    return x;
}

/* frames produced by the encoder for a segment of L samples (follows the pad plans, incl. D9) */
REF_API int64_t ref_encodec_frames(const ref_encodec* m, int64_t L) {
    const ref_encodec_config* c = &m->cfg;
    L = plan_sconv(L, c->kernel_size, 1, 1, c->causal).Lout;
    for (int i = c->n_ratios - 1; i >= 0; i--) {
        L = plan_sconv(L, c->residual_kernel_size, 1, 1, c->causal).Lout;
        L = plan_sconv(L, 2 * c->ratios[i], c->ratios[i], 1, c->causal).Lout;
    }
    return plan_sconv(L, c->last_kernel_size, 1, 1, c->causal).Lout;
}
REF_API int64_t ref_encodec_decoded_length(const ref_encodec* m, int64_t Tz) {
    const ref_encodec_config* c = &m->cfg;
    int64_t L = plan_sconv(Tz, c->kernel_size, 1, 1, c->causal).Lout;
    for (int i = 0; i < c->n_ratios; i++) {
        L = L * c->ratios[i];
        L = plan_sconv(L, c->residual_kernel_size, 1, 1, c->causal).Lout;
    }
    return plan_sconv(L, c->last_kernel_size, 1, 1, c->causal).Lout;
}

/* Encodec.EncodeFrame: x [B,C,L] -> codes [B,n_q,T'] int64, scale [B] (normalize) , emb nullable [B,dim,T'] */
REF_API int ref_encodec_encode_frame(ref_encodec* m, const float* x_in, int64_t B, int64_t L, int n_q, int64_t* codes, float* scale_out,
                                     float* emb_out) {
    const ref_encodec_config* c = &m->cfg;
    char nm[320];
    m->bad = 0;
    int C = c->channels;
    float* x = (float*)malloc(sizeof(float) * B * C * L);
    memcpy(x, x_in, sizeof(float) * B * C * L);
    if (c->normalize) {
        for (int64_t b = 0; b < B; b++) {
            float* mono = (float*)malloc(sizeof(float) * L);
            for (int64_t t = 0; t < L; t++) {
                float a = x[(b * C) * L + t];
                for (int ch = 1; ch < C; ch++) a = a + x[(b * C + ch) * L + t];
                const float mv = a / (float)C;
                mono[t] = mv * mv;
            }
            double s1 = 0.0;
            for (int64_t t0 = 0; t0 < L; t0 += GN_CHUNK) {
                const int64_t t1 = t0 + GN_CHUNK < L ? t0 + GN_CHUNK : L;
                double c1 = 0.0;
                for (int64_t t = t0; t < t1; t++) c1 += (double)mono[t];
                s1 += c1;
            }
            free(mono);
            const float scale = sqrtf((float)(s1 / (double)L)) + 1e-8f;
            if (scale_out) scale_out[b] = scale;
            for (int64_t i = 0; i < C * L; i++) x[b * C * L + i] = x[b * C * L + i] / scale;
        }
    }
    int64_t Lc = L;
    float* y = sconv(m, "encoder.layers.0", x, B, C, Lc, c->kernel_size, 1, 1, &C, &Lc);
    free(x); x = y;
    int n = 1;
    for (int i = c->n_ratios - 1; i >= 0 && !m->bad; i--) {
        const int r = c->ratios[i];
        snprintf(nm, sizeof nm, "encoder.layers.%d", n);
        x = resblock(m, nm, x, B, C, Lc);
        float* a = elu_new(x, B * C * Lc);
        free(x);
        snprintf(nm, sizeof nm, "encoder.layers.%d", n + 2);
        int Co; int64_t Lo;
        x = sconv(m, nm, a, B, C, Lc, 2 * r, r, 1, &Co, &Lo);
        free(a);
        C = Co; Lc = Lo;
        n += 3;
    }
    snprintf(nm, sizeof nm, "encoder.layers.%d", n);
    x = slstm(m, nm, x, B, C, Lc);
    {
        float* a = elu_new(x, B * C * Lc);
        free(x);
        snprintf(nm, sizeof nm, "encoder.layers.%d", n + 2);
        int Co; int64_t Lo;
        x = sconv(m, nm, a, B, C, Lc, c->last_kernel_size, 1, 1, &Co, &Lo);
        free(a);
        C = Co; Lc = Lo;
    }
    const int64_t Tz = Lc;
    const int D = C;
    if (emb_out) memcpy(emb_out, x, sizeof(float) * B * D * Tz);
    /* ResidualVectorQuantizer.Encode (:133-157): residual -= embed[argmin] */
    int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * B * Tz);
    float* st = (float*)malloc(sizeof(float) * B * D * Tz);
    for (int i = 0; i < n_q && !m->bad; i++) {
        snprintf(nm, sizeof nm, "quantizer.layers.%d.codebook.embed", i);
        const float* cb = getf(m, nm, (int64_t)c->codebook_size * D);
        if (!cb) break;
        ref_vq_argmin(x, B, D, Tz, cb, c->codebook_size, idx, st, NULL);
        for (int64_t b = 0; b < B; b++) {
            memcpy(codes + (b * n_q + i) * Tz, idx + b * Tz, sizeof(int64_t) * Tz);
            for (int d = 0; d < D; d++)
                for (int64_t t = 0; t < Tz; t++) {
                    const int64_t o = (b * D + d) * Tz + t;
                    x[o] = x[o] - cb[idx[b * Tz + t] * D + d];
                }
        }
    }
    free(idx); free(st); free(x);
    return m->bad ? -1 : (int)Tz;
}

/* Encodec.DecodeFrame: codes [B,n_q,T'] -> out [B,channels,Lout] (x scale[b] when scale != NULL) */
REF_API int ref_encodec_decode_frame(ref_encodec* m, const int64_t* codes, int64_t B, int n_q, int64_t Tz, const float* scale, float* out,
                                     float* emb_out) {
    const ref_encodec_config* c = &m->cfg;
    char nm[320];
    m->bad = 0;
    const int D = c->dimension;
    float* x = (float*)calloc(B * D * Tz, sizeof(float));
    for (int i = 0; i < n_q; i++) {
        snprintf(nm, sizeof nm, "quantizer.layers.%d.codebook.embed", i);
        const float* cb = getf(m, nm, (int64_t)c->codebook_size * D);
        if (!cb) { free(x); return -1; }
        for (int64_t b = 0; b < B; b++)
            for (int d = 0; d < D; d++)
                for (int64_t t = 0; t < Tz; t++) {
                    const int64_t o = (b * D + d) * Tz + t;
                    x[o] = x[o] + cb[codes[(b * n_q + i) * Tz + t] * D + d];
                }
    }
    if (emb_out) memcpy(emb_out, x, sizeof(float) * B * D * Tz);
    int C = D; int64_t L = Tz;
    float* y = sconv(m, "decoder.layers.0", x, B, C, L, c->kernel_size, 1, 1, &C, &L);
    free(x); x = y;
    x = slstm(m, "decoder.layers.1", x, B, C, L);
    int n = 2;
    for (int i = 0; i < c->n_ratios && !m->bad; i++) {
        const int r = c->ratios[i];
        float* a = elu_new(x, B * C * L);
        free(x);
        snprintf(nm, sizeof nm, "decoder.layers.%d", n + 1);
        int Co; int64_t Lo;
        x = sconvT(m, nm, a, B, C, L, r, &Co, &Lo);
        free(a);
        C = Co; L = Lo;
        snprintf(nm, sizeof nm, "decoder.layers.%d", n + 2);
        x = resblock(m, nm, x, B, C, L);
        n += 3;
    }
    float* a = elu_new(x, B * C * L);
    free(x);
    snprintf(nm, sizeof nm, "decoder.layers.%d", n + 1);
    int Co; int64_t Lo;
    x = sconv(m, nm, a, B, C, L, c->last_kernel_size, 1, 1, &Co, &Lo);
    free(a);
    for (int64_t b = 0; b < B; b++)
        for (int64_t i = 0; i < Co * Lo; i++) out[b * Co * Lo + i] = scale ? x[b * Co * Lo + i] * scale[b] : x[b * Co * Lo + i];
    free(x);
    return m->bad ? -1 : (int)Lo;
}

/* DSP.LinearOverlapAdd: `rows` independent rows; frame f holds rows x lens[f] samples at frames[offs[f]..] */
REF_API void ref_linear_overlap_add(const float* frames, const int64_t* offs, const int64_t* lens, int n_frames, int64_t rows, int64_t stride,
                                    float* out, int64_t total) {
    const int64_t L0 = lens[0];
    float* w = (float*)malloc(sizeof(float) * L0);
    for (int64_t i = 0; i < L0; i++) {
        const float t = (float)((double)(i + 1) / (double)(L0 + 1));
        w[i] = 0.5f - fabsf(t - 0.5f);
    }
    float* sw = (float*)calloc(total, sizeof(float));
    for (int64_t i = 0; i < rows * total; i++) out[i] = 0.0f;
    int64_t off = 0;
    for (int f = 0; f < n_frames; f++) {
        for (int64_t r = 0; r < rows; r++)
            for (int64_t i = 0; i < lens[f]; i++) out[r * total + off + i] = out[r * total + off + i] + frames[offs[f] + r * lens[f] + i] * w[i];
        for (int64_t i = 0; i < lens[f]; i++) sw[off + i] = sw[off + i] + w[i];
        off += stride;
    }
    float mn = INFINITY;
    for (int64_t i = 0; i < total; i++) if (sw[i] < mn) mn = sw[i];
    if (mn <= 1e-10f) for (int64_t i = 0; i < total; i++) sw[i] = sw[i] + 1e-10f;
    for (int64_t r = 0; r < rows; r++)
        for (int64_t i = 0; i < total; i++) out[r * total + i] = out[r * total + i] / sw[i];
    free(w); free(sw);
}
