/* ORACLE (test infrastructure): declarations shared by the oracle's translation units. */
#ifndef NC_REF_INTERNAL_H
#define NC_REF_INTERNAL_H
#include <stdint.h>

#define REF_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ weight blob (NCWB0001) */
typedef struct {
    char name[176];
    int dtype, ndim;
    int64_t dims[6];
    const void* data;
    int64_t nbytes;
} ref_tensor;

typedef struct {
    int n;
    ref_tensor* t;
} ref_blob;

int ref_blob_parse(const uint8_t* buf, int64_t len, ref_blob* out);
const ref_tensor* ref_blob_find(const ref_blob* b, const char* name);

/* ops of nc_ref.c (canonical arithmetic) */
REF_API void ref_snake(const float* x, const float* alpha, int64_t B, int64_t C, int64_t T, float* y);
REF_API void ref_tanh(const float* x, int64_t n, float* y);
REF_API void ref_conv1d(const float* x, int64_t B, int Cin, int64_t Tin, const float* w, const float* bias, int Cout, int K,
                        int stride, int pad, int dil, int groups, const float* residual, float* y, int64_t Tout);
REF_API void ref_conv_transpose1d(const float* x, int64_t B, int Cin, int64_t Tin, const float* w, const float* bias, int Cout,
                                  int K, int stride, int pad, int out_pad, float* y, int64_t Tout);
REF_API void ref_vq_argmin(const float* z_e, int64_t B, int D, int64_t T, const float* cb, int N, int64_t* idx, float* st,
                           float* best_dist);
REF_API void ref_vq_gather(const int64_t* idx, int64_t B, int D, int64_t T, const float* cb, float* out);
#endif
