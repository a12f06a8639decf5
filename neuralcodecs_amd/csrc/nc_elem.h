// Host-side descriptors of the HBM-bound element-wise / small-reduction kernels (nc_elem.hip).
#pragma once
#include "nc_common.h"

namespace nc {

// depthwise (groups == channels) convolution, weight [C,1,K] folded to dense [C][K]
struct DwConvLayer {
    int C = 0, K = 0, pad = 0, dil = 1;
    DevBuf w, bias;
    bool has_bias = false;
    void build(const float* dense_w, const float* bias_h, int C, int K, int pad, int dil);
};
// SNAC residual unit (depthwise k = 7 + Snake + pointwise + skip) in one launch: nc_snac_unit.hip
struct SnacFusedUnit {
    int C = 0, dil = 1;
    bool ready = false, has_b1 = false;
    DevBuf w1, tab, b1;
    static bool supported(int C, int K, int dil);
    void build(int C, int dil, const float* w7, const float* b7, const float* a1, const float* a2, const float* w1_dense, const float* b1);
    bool usable(const float* x, const float* y, int64_t T, int B) const;
    void launch(const float* x, const float* alpha_next, float* y, int B, int64_t T, int cu_count, hipStream_t s, Profiler* prof) const;
};
void launch_dwconv(const DwConvLayer& L, const float* x, const float* alpha_in, const float* alpha_out, float* y, int B, int64_t T,
                   hipStream_t s, Profiler* prof);
void launch_avg_pool(const float* x, float* y, int64_t rows, int64_t T, int s, hipStream_t st, Profiler* prof = nullptr);
void launch_rvq_update(const float* q, float* zq, float* residual, int64_t rows, int64_t T, int s, bool first, hipStream_t st,
                       Profiler* prof = nullptr);
void launch_layernorm_ct(const float* x, const float* gamma, const float* beta, float* y, int B, int C, int64_t T, hipStream_t st,
                         Profiler* prof = nullptr);
void launch_local_attn(const float* qkv, const float* cs, const float* sn, float* out, int B, int C, int64_t T, int W, hipStream_t st,
                       Profiler* prof = nullptr);
void launch_randn(float* out, int64_t n, uint64_t seed, hipStream_t st);

}  // namespace nc
