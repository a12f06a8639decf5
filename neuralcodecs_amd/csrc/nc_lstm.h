// Fused two-layer persistent LSTM of the Encodec SLSTM (Modules/Encodec/SLSTM.cs:31,40-57): launcher interface of nc_lstm.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace nc {

// One launch runs ALL T steps of BOTH layers for up to two 16-row column tiles (see nc_lstm.hip for the protocol).
struct Lstm2Args {
    const float* gi0;      // [4C][T][N] input projections of layer 0 incl. b_ih (W_ih0 x_t + b_ih0: one pointwise GEMM beforehand)
    const float* whh0;     // fragment images [C/4 workgroups][C/16 k-groups][64 lanes][4]: W_hh of layer 0,
    const float* wih1;     //   W_ih of layer 1,
    const float* whh1;     //   W_hh of layer 1 (lstm2_pack_image)
    const float* bhh0;     // [4C]
    const float* bih1;     // [4C]
    const float* bhh1;     // [4C]
    const float* skip;     // [N,C,T] x: added to the last layer's output (SLSTM.cs:50-53)
    float* out;            // [N,C,T]
    int elu_out;           // ELU applied to the stored value (the activation in front of the consuming convolution)
    float* S;              // exchange regions [2 layers][T][tiles of this launch][C*16], pre-filled with LSTM2_SENTINEL words
    unsigned* flags;       // [tiles of this launch][2 layers][C/4]: steps published per workgroup (zeroed before the launch)
    unsigned* tmo;         // timeout word (zeroed at model creation; host-visible)
    int N, C;
    int64_t T;
    int tile0;             // first column tile of this launch
    int tiles;             // 1 or 2 (workgroup = 8 wavefronts per tile)
    unsigned long long* trace;   // nullable diagnostic: [C/4][tiles][8 roles][LSTM2_TRACE_STEPS][4] s_memrealtime stamps (100 MHz) of steps
                                 // [LSTM2_TRACE_T0, +LSTM2_TRACE_STEPS): NC_LSTM2_TRACE=<file> (tools/probe/lstm2_trace.py reads it)
};

constexpr int LSTM2_TRACE_T0 = 64, LSTM2_TRACE_STEPS = 8;
constexpr unsigned LSTM2_SENTINEL = 0xFFFFFFFFu;   // a NaN pattern no gate output can take (|h| < 1); the byte memset value 0xFF

// Per-layer launch with the same protocol (one layer, `tiles` column tiles starting at tile0, steps [t0, t1) of T): see nc_lstm.hip
struct LstmSplitArgs {
    const float* gi;       // input projections incl. b_ih, element strides (gi_b, gi_c, gi_t) over (clip, channel row, step)
    const float* w;        // W_hh fragment images (lstm2_pack_image)
    const float* bhh;      // [4C]
    const float* skip;     // nullable [N,C,T]
    float* out;            // strides (out_b, out_c, out_t)
    int elu_out;
    float* S;              // exchange regions [T][tiles_total][C*16], filled with LSTM2_SENTINEL words once per call
    unsigned* flags;       // [tiles_total][C/4], zeroed once per call
    unsigned* tmo;
    float* cstate;         // [tiles_total][C][16] cell state carried between chunk launches
    int64_t gi_b, gi_c, gi_t, out_b, out_c, out_t;
    int N, C;
    int64_t T, t0, t1;
    int tile0, tiles_total;
};
size_t lstm1_lds_bytes(int C);
void lstm1_launch(const LstmSplitArgs& a, int tiles, hipStream_t stream);

size_t lstm2_lds_bytes(int C, int tiles);
size_t lstm2_exchange_floats(int C, int64_t T, int tiles);
bool lstm2_supported(int C);
// W [4C][C] (PyTorch gate-major rows i,f,g,o) -> the A-fragment image of the fused kernel (host side, at load)
void lstm2_pack_image(const float* W, int C, float* image);
void lstm2_launch(const Lstm2Args& a, hipStream_t stream);

}  // namespace nc
