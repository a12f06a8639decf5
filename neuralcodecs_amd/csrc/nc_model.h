// Engine objects behind the opaque nc_codec handle.
#pragma once
#include <memory>

#include "nc_common.h"
#include "nc_conv.h"
#include "nc_elem.h"

namespace nc {

const char* get_last_error();
void fold_weight_norm_dac(const float* v, const float* g, int64_t d0, int64_t inner, float* w);
void fold_weight_norm_snac(const float* v, const float* g, int64_t d0, int64_t inner, float* w);

// ---- RVQ kernels (nc_rvq.hip) ----------------------------------------------------------------
// Codebook resident on the device in the two layouts the kernels want.
struct Codebook {
    int N = 0, D = 0;
    DevBuf cbT;  // [D][N] transposed copy: conflict-free LDS image for the argmin
    DevBuf cb;   // [N][D] row-major: gather
    DevBuf c2;   // [N] squared norms (canonical fma chain, computed once on the host)
    void build(const float* host_cb, int N, int D);
};
// One stage of the stage-fused DAC quantizer (nc_rvq.hip): dense projection weights beside the codebook images, all device pointers.
struct RvqStage {
    const float* w_inT;   // [latent][D]  in_proj weight, transposed (folded weight norm)
    const float* b_in;    // [D]
    const float* w_out;   // [latent][D]  out_proj weight
    const float* b_out;   // [latent]
    const float* cbT;     // [D][N]
    const float* c2;      // [N]
    const float* cb;      // [N][D]
};
// ResidualVectorQuantizer.forward for n_q stages in ONE launch (ResidualVectorQuantizer.cs:54-103); returns false when the shape has
// no instantiation (the caller then runs the stage-by-stage launches).  residual [B,L,T] (read only), zq [B,L,T], latents
// [B,n_q*D,T], codes [B,n_q,T].
bool launch_dac_rvq_fused(const RvqStage* stages_dev, int n_q, int L, int D, int N, const float* residual, int B, int64_t T, int64_t* codes,
                          float* zq, float* latents, hipStream_t s, Profiler* prof);
// z_e [B,D,T] (batch stride ze_bstride) -> codes[b*codes_bstride + t] (int64) and st [B,D,T] = z_e + (cb[idx] - z_e)
void launch_vq_argmin(const Codebook& cb, const float* z_e, int64_t ze_bstride, int B, int64_t T, int64_t* codes,
                      int64_t codes_bstride, float* st, hipStream_t s, Profiler* prof);
// codes -> out [B,D,T] = cb[codes]  (Embedding + transpose, VectorQuantizer.cs:135-142)
// Test hook (nc_op_euclid_rvq): the Encodec Euclidean RVQ (ResidualVectorQuantizer.cs:133-157) on residual [B,D,T] (updated in place) with n_q
// codebooks [n_q][N][D] (host): form 0 = the per-stage kernel, 1 = the all-stages matrix-core kernel (D == 128, N % 512 == 0)
void op_euclid_rvq(const float* residual_in, int B, int D, int64_t T, const float* books_host, int n_q, int N, int form, int64_t* codes_host,
                   float* residual_out);
void launch_vq_gather(const Codebook& cb, const int64_t* codes, int64_t codes_bstride, int B, int64_t T, float* out, hipStream_t s,
                      Profiler* prof);

// ---- codec objects ---------------------------------------------------------------------------
struct Codec {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    bool loaded = false;
    int cu_count = 0;              // compute units of the device (multiProcessorCount)
    size_t lds_per_cu = 0;         // LDS a workgroup may opt into (maxSharedMemoryPerMultiProcessor)
    Profiler prof;
    virtual ~Codec();
    virtual void load(const Blob& blob) = 0;
    void init_device(int device_index);
    void use_device() const;
    // errors raised by device code after the launch returned (bounded spins of the persistent kernels): called once the stream is idle
    virtual void check_async_errors() {}
    // Rebind the handle to another stream.  The workspaces are shared by all calls on the handle, so work already queued on the old
    // stream is ordered before anything the new stream will launch (event record + wait; no host synchronisation).
    void switch_stream(hipStream_t s);
};

// Host-pointer entry points run on the handle's own stream whatever stream the device-pointer API was last bound to.
struct OwnStreamScope {
    Codec& c;
    hipStream_t prev;
    explicit OwnStreamScope(Codec& c_) : c(c_), prev(c_.stream) { c.use_device(); c.switch_stream(c.own_stream); }
    ~OwnStreamScope() {
        try { c.switch_stream(prev); } catch (...) {}
    }
};

struct DacModel : Codec {
    nc_dac_config cfg{};
    int latent = 0, hop = 1;
    bool fuse_res_units = true;  // NC_NO_FUSE=1 in the environment selects the two-launch residual units (A/B, tests)

    struct ResUnit {
        DevBuf a1, a2;
        ConvLayer c7, c1;
    };
    ConvLayer enc_stem;
    struct EncBlk {
        ResUnit ru[3];
        DevBuf a_down;
        ConvLayer down;
    } enc[8];
    DevBuf enc_alpha_out;
    ConvLayer enc_out;

    std::vector<std::unique_ptr<ConvLayer>> in_proj, out_proj;
    std::vector<std::unique_ptr<Codebook>> codebooks;
    std::vector<std::unique_ptr<DevBuf>> rvq_dense;   // dense projection weights / biases of the stage-fused quantizer
    DevBuf rvq_stages;                                // RvqStage[n_codebooks]

    ConvLayer dec_in;
    struct DecBlk {
        DevBuf a_up;
        ConvLayer up;
        ResUnit ru[3];
    } dec[8];
    DevBuf dec_alpha_out;
    ConvLayer dec_out;

    // workspace (grow-only)
    DevBuf act[3], resid, zq, lat, st, codes_ws, h_in, h_out, h_codes, h_aux0, h_aux1;

    explicit DacModel(const nc_dac_config& c);
    void load(const Blob& blob) override;
    int64_t padded_len(int64_t T) const { return (T + hop - 1) / hop * hop; }
    int64_t frames(int64_t T) const { return (T + hop - 1) / hop; }
    int64_t decoded_len(int64_t frames) const;
    // device-pointer entry points (async on `stream`)
    void encode_dev(const float* pcm, int B, int64_t T, int sample_rate, int n_q, int64_t* codes, float* z, float* latents);
    void decode_dev(const float* z, int B, int64_t frames, float* pcm);
    void from_codes_dev(const int64_t* codes, int B, int n_q, int64_t frames, float* z);
    void decode_code_matrix_dev(const int64_t* codes_tq, int B, int64_t frames, int n_q, float* pcm);
    void encode_code_matrix_dev(const float* pcm, int B, int64_t T, int sample_rate, int64_t* codes_tq);

  private:
    float* run_res_unit(ResUnit& ru, int dil, float* cur, int C, int64_t L, int B, int& cur_idx, const float* alpha_next);
};

struct SnacModel : Codec {
    nc_snac_config cfg{};
    int latent = 0, hop = 1;
    int64_t pad_to = 1;

    struct ResUnit {
        DevBuf a1, a2;
        DwConvLayer dw;   // depthwise flavour
        ConvLayer c7;     // dense flavour (depthwise == 0)
        ConvLayer c1;
        SnacFusedUnit fu; // depthwise flavour, narrow long rows: the whole unit in one launch
    };
    struct Mha {
        int C = 0;
        DevBuf gamma, beta, cs, sn;
        ConvLayer qkv, out;
    };
    ConvLayer enc_stem;
    struct EncBlk {
        ResUnit ru[3];
        DevBuf a_down;
        ConvLayer down;
    } enc[8];
    Mha enc_mha, dec_mha;
    DwConvLayer enc_out_dw, dec_in_dw;
    ConvLayer enc_out, dec_in;

    std::vector<std::unique_ptr<ConvLayer>> in_proj, out_proj;
    std::vector<std::unique_ptr<Codebook>> codebooks;

    struct DecBlk {
        DevBuf a_up;
        ConvLayer up, noise;
        ResUnit ru[3];
    } dec[8];
    DevBuf dec_alpha_out;
    ConvLayer dec_out;

    DevBuf act[3], resid, zq, pooled, qbuf, lat, st, qkv_ws, noise_ws, h_in, h_out, h_codes, h_aux0, h_aux1, h_noise;

    explicit SnacModel(const nc_snac_config& c);
    void load(const Blob& blob) override;
    int64_t padded_len(int64_t T) const { return (T + pad_to - 1) / pad_to * pad_to; }
    static int64_t up_len(int64_t L, int s) { return (L - 1) * s - 2 * ((s + 1) / 2) + 2 * s + (s % 2); }
    int64_t decoded_len(int64_t frames) const;
    int64_t noise_len(int B, int64_t frames) const;
    int64_t codes_per_clip(int64_t frames) const {
        int64_t n = 0;
        for (int i = 0; i < cfg.n_vq_strides; ++i) n += frames / cfg.vq_strides[i];
        return n;
    }
    // pad = true: SNAC.Encode(float[]) / forward (Preprocess, then the encoder on the padded tensor; SNAC.cs:129-150, 91-106)
    // pad = false: SNAC.Encode(Tensor) AS WRITTEN (SNAC.cs:113-122, deviation D7): the encoder runs on the un-padded tensor
    void encode_dev(const float* pcm, int B, int64_t T, int64_t* codes, float* z, float* zq, bool pad = true);
    // frames the encoder emits for an un-padded row of T samples; NC_EINVAL where the reference's quantizer / LocalMHA would throw
    int64_t unpadded_frames(int64_t T) const;
    void from_codes_dev(const int64_t* codes, int B, int64_t frames, float* zq);
    void decode_dev(const int64_t* codes, int B, int64_t frames, const float* noise, uint64_t seed, float* pcm);

  private:
    void load_res_unit(const Blob& b, const std::string& q, ResUnit& ru, int C, int dil);
    void load_mha(const Blob& b, const std::string& p, Mha& m, int C);
    float* run_res_unit(ResUnit& ru, float* cur, int C, int64_t L, int B, int& cur_idx, const float* alpha_next);
    float* run_mha(Mha& m, float* cur, int C, int64_t L, int B, int& cur_idx, const float* alpha_next);
    void reserve_act(int B, int64_t Tp);
};

struct EncodecModel : Codec {
    nc_encodec_config cfg{};
    int hop = 1, n_q = 1;

    struct SConv {                 // SConv1d / SConvTranspose1d: conv (+ GroupNorm affine when time_group_norm)
        int K = 0, stride = 1, Cin = 0, Cout = 0;
        bool transposed = false;
        ConvLayer conv;
        DevBuf gamma, beta;
    };
    struct ResBlock { SConv c1, c2, sc; };
    struct LstmLayer { ConvLayer ih; DevBuf whh, whhp, bhh, bih, w2hh, w2ih; };   // w2*: fragment images of the fused two-layer kernel (nc_lstm.h)
    struct Lstm { int C = 0; std::vector<std::unique_ptr<LstmLayer>> layers; };
    struct Plan { int64_t left = 0, right = 0, Lz = 0, Lp = 0, Lout = 0; };
    struct Seg { int64_t off = 0, len = 0, frames = 0; };
    struct Act {                   // [N,C,L] view of a raw conv output + its pending GroupNorm
        const float* p = nullptr;
        int C = 0;
        int64_t L = 0, rs = 0, off = 0;
        const float* stats = nullptr;
        const float* gamma = nullptr;
        const float* beta = nullptr;
    };

    SConv enc_in, enc_down[8], enc_out, dec_in, dec_up[8], dec_out;
    ResBlock enc_res[8], dec_res[8];
    Lstm enc_lstm, dec_lstm;
    std::vector<std::unique_ptr<Codebook>> books;
    DevBuf book_ptrs, book_ptrsT, book_ptrs2;   // per stage: codebook [N][D], its transpose [D][N], squared norms [N] (device pointer arrays)

    std::vector<std::unique_ptr<DevBuf>> pool;   // per-call intermediates, same allocation order every call (grow-only)
    size_t pool_i = 0;
    DevBuf h_in, h_out, h_codes, h_scales, h_emb;
    // Timeout word of the persistent LSTM kernels: ONE word of pinned, device-mapped host memory -- a kernel that gives up its spin
    // writes it over PCIe, the host reads it without touching the stream.  After a timeout the handle runs the step-wise kernels
    // (lstm_force_stepwise): the persistent form needs its workgroups co-resident, which a busy / partitioned device may not grant.
    unsigned* lstm_tmo_host = nullptr;
    unsigned* lstm_tmo_dev = nullptr;
    bool lstm_force_stepwise = false;
    int64_t lstm_timeouts = 0;                   // timeouts this handle has seen (nc_encodec_lstm_stats)
    struct LstmTicket* lstm_ticket = nullptr;    // per-device serialisation of persistent LSTM sections across handles (nc_encodec.hip)
    bool lstm_timed_out() const { return lstm_tmo_host && *reinterpret_cast<volatile unsigned*>(lstm_tmo_host) != 0; }
    // segment groups of one call are independent until the overlap-add: the first runs on the handle's stream, the others on side
    // streams (forked / joined with events), so the short tail segment of a clip hides behind the full-length batch
    hipStream_t side_stream[2] = {nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
    // layer-pipelined LSTM (run_lstm): second stream for layer 1 and the chunk events; only the primary segment group pipelines
    hipStream_t lstm_stream = nullptr;
    std::vector<hipEvent_t> lstm_events;
    // overlap-add operands (decode_dev): the window and the weight sum depend on the frame geometry only and stay on the device;
    // the per-call frame pointers travel through a small ring of pinned host slots, so decode_dev never synchronises the stream
    std::vector<int64_t> ola_key;
    DevBuf ola_w, ola_sw;
    void* ola_pin = nullptr;
    size_t ola_slot_bytes = 0;
    int ola_next = 0;
    hipEvent_t ola_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool ola_ev_used[4] = {false, false, false, false};
    bool on_side_group = false;
    ~EncodecModel() override;

    explicit EncodecModel(const nc_encodec_config& c);
    void check_async_errors() override;
    void absorb_stale_timeout();
    void load(const Blob& blob) override;
    void set_bandwidth(float bw);
    Plan plan_sconv(int64_t L, int k, int stride, int dil) const;
    int64_t frames_for(int64_t L) const;
    int64_t decoded_for(int64_t Tz) const;
    std::vector<Seg> segments(int64_t T) const;
    void encode_dev(const float* pcm, int B, int64_t T, int64_t* codes, float* scales, float* emb);
    void decode_dev(const int64_t* codes, const float* scales, int B, int64_t T, int nq, float* pcm);

  private:
    void load_sconv(const Blob& b, const std::string& key, SConv& L, int Cin, int Cout, int K, int stride, bool transposed);
    void load_resblock(const Blob& b, const std::string& key, ResBlock& r, int dim);
    void load_lstm(const Blob& b, const std::string& key, Lstm& l, int C);
    float* alloc(size_t n_floats);
    float* pad_act(const Act& a, const Act* b2, bool elu, int N, const Plan& pl);
    struct GnJob { bool on = false, fused = false, finished = false; int sub = 1, nrb = 0, ncb = 0; double* part = nullptr; float* stats = nullptr; };
    static constexpr int GN_MAX_SAMPLES = 4096;   // rows of a segment group (encode_dev caps a group at 4096)
    DevBuf gn_counters;                           // [3 groups][2][GN_MAX_SAMPLES] arrival counters of the in-launch GroupNorm finish (zero between launches; the second set: the
                                                  // branch output of the fused first pass of a residual block, two outputs finishing in one launch)
    int cur_group = 0;
    GnJob gn_begin(const ConvLayer& conv, ConvIO& io, int N, int C, int64_t L, int sub);
    const float* gn_end(const GnJob& j, const float* raw, int N, int C, int64_t L, int64_t rs = 0);   // rs: row pitch of `raw` (0 = dense rows of L)
    Act sconv(SConv& L, const Act& a, const Act* b2, bool elu, int N);
    Act sconvT(SConv& L, const Act& a, const Act* b2, bool elu, int N);
    void resblock(ResBlock& r, const Act& x, int N, Act& s, Act& y);
    bool resblock_first_pass(ResBlock& r, const Act& x, int N, Act& s, Act& h);   // shortcut + k = 3 branch in one launch (nc_resa.hip); false: not this shape
    float* materialize(const Act& a, int N, const float* scale, int mode);
    float* run_lstm(Lstm& l, const float* x, int N, int64_t T, bool elu_out);
    void encode_batch(const float* x, int N, int64_t L, int64_t Tz, int64_t* codes, float* scale_out, float* emb_out);
    float* decode_batch(const int64_t* codes, int N, int nq, int64_t Tz, const float* scale, int64_t* Lout);
};

}  // namespace nc

// the opaque handle of the C ABI
struct nc_codec {
    std::unique_ptr<nc::Codec> impl;
    int kind = 0;  // 0 = DAC, 1 = SNAC, 2 = Encodec
};
