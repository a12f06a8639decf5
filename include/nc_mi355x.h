/* nc_mi355x.h -- C ABI of the MI355X-native neural-audio-codec engine (libnc_mi355x.so).
 *
 * This is the drop-in boundary for the reference's Encode/Decode hot path.  The reference
 * (DillionLowry/NeuralCodecs, C#/.NET 8 on TorchSharp) has no FFI of its own for this path:
 * every op is a TorchSharp P/Invoke into libtorch.  The functions below are what a C# shim
 * binds with [DllImport("nc_mi355x")] to keep the managed API surface
 * (INeuralCodec, DAC.Encode/Decode/FromCodes, SNAC.Encode/Decode, Encodec.Encode/Decode)
 * while every FLOP runs in hand-written HIP kernels for gfx950.  See INTEGRATION.md.
 *
 * Conventions
 *   - plain C types only; tensors are dense row-major float32 [B,C,T] / int64 codes [B,Nq,T'].
 *   - every function returns nc_status; nc_last_error() gives the thread-local message.
 *     Mapping to the reference's exceptions: NC_EINVAL -> ArgumentException /
 *     ArgumentNullException, NC_ENOTFOUND -> FileNotFoundException, NC_ESTATE ->
 *     InvalidOperationException, NC_EDEVICE / NC_ENOMEM -> NeuralCodecException.
 *   - one handle = one device + one HIP stream; calls on one handle must be serialised by the
 *     caller (the reference's modules are not thread-safe either: WNConv1d.cs:150 mutates state
 *     in forward); distinct handles may run concurrently.
 *   - "*_dev" variants take DEVICE pointers, enqueue on the handle's stream and return without
 *     synchronising (zero-copy path used by benchmarks / multi-GPU sharding); the plain variants
 *     take HOST pointers and are synchronous (the float[] / Tensor overloads of the reference).
 *   - there is NO CPU fallback: every entry point fails with NC_EDEVICE when no gfx950 device
 *     is usable.
 */
#ifndef NC_MI355X_H
#define NC_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NC_API __attribute__((visibility("default")))

typedef enum {
    NC_OK = 0,
    NC_EINVAL = 1,     /* bad argument (null pointer, wrong sample rate, bad shape) */
    NC_ENOTFOUND = 2,  /* weight file / tensor not found */
    NC_ESTATE = 3,     /* call not valid in this state (weights not loaded, ...) */
    NC_EDEVICE = 4,    /* HIP error / no usable device */
    NC_ENOMEM = 5,
    NC_EUNSUPPORTED = 6
} nc_status;

typedef struct nc_codec nc_codec; /* opaque: replaces the managed DAC / SNAC / Encodec object (INeuralCodec + IDisposable,
                                     NeuralCodecs.Core/INeuralCodec.cs:8-20) */

NC_API const char* nc_last_error(void);
NC_API const char* nc_version(void);
/* number of HIP devices visible (0 when none); never fails */
NC_API int nc_device_count(void);
/* Every diagnostic environment switch the engine reads, one "NAME\tkind\twhat it does\n" line each; kind b = set to 1,
 * p = set at all, i = integer, s = string.  None of them changes a result (every path is bit-exact against the oracle); they exist so
 * that an A/B on one box is one environment variable.  The engine refuses to read a switch that is not in this table. */
NC_API const char* nc_debug_switches(void);

/* ------------------------------------------------------------------------------------------ DAC
 * replaces: new DAC(DACConfig)                    NeuralCodecs.Torch/Models/DAC.cs:51-93
 *           fields consumed                       NeuralCodecs.Torch/Config/DAC/DACConfig.cs:22-76 */
typedef struct {
    int32_t sample_rate;     /* 44100 */
    int32_t encoder_dim;     /* 64 */
    int32_t n_encoder_rates; /* 4 */
    int32_t encoder_rates[8];/* 2,4,8,8 */
    int32_t decoder_dim;     /* 1536 */
    int32_t n_decoder_rates; /* 4 */
    int32_t decoder_rates[8];/* 8,8,4,2 */
    int32_t latent_dim;      /* 0 => encoder_dim * 2^n_encoder_rates (DAC.cs:64) */
    int32_t n_codebooks;     /* 9 */
    int32_t codebook_size;   /* 1024 */
    int32_t codebook_dim;    /* 8 */
} nc_dac_config;

NC_API nc_status nc_dac_create(const nc_dac_config* cfg, int device_index, nc_codec** out);

/* replaces: IDisposable.Dispose                   Models/DAC.cs:329-338 */
NC_API nc_status nc_codec_destroy(nc_codec* h);

/* replaces: INeuralCodec.LoadWeights(path)        Models/DAC.cs:345-389 (FileNotFound / InvalidOperation)
 * The file is an NCWB0001 weight blob whose tensor names are the reference's TorchSharp
 * state-dict keys (weight_v / weight_g / bias / alpha / codebook.weight); the
 * weight-norm fold w = v/(||v||+1e-7)*g (WNConv1d.cs:145-150) happens once here. */
NC_API nc_status nc_codec_load_weights(nc_codec* h, const char* path);
NC_API nc_status nc_codec_load_weights_mem(nc_codec* h, const void* blob, size_t nbytes);
/* Host-only validation of an NCWB0001 image (what the two loaders run first): every index field bounds-checked, (offset, size) pairs
 * checked without overflow, dtype / byte-count consistency.  NC_EINVAL on a truncated or crafted image; needs no device. */
NC_API nc_status nc_blob_check(const void* blob, size_t nbytes, int32_t* n_tensors);

/* Run the handle's work on a caller-owned hipStream_t.  NULL is HIP's legacy default (null) stream -- the stream PyTorch-ROCm
 * uses by default -- so device buffers produced by the caller's framework are ordered with the engine's kernels.
 * nc_codec_reset_stream returns to the handle's own (non-blocking) stream. */
NC_API nc_status nc_codec_set_stream(nc_codec* h, void* hip_stream);
NC_API nc_status nc_codec_reset_stream(nc_codec* h);
NC_API nc_status nc_codec_synchronize(nc_codec* h);
/* Errors raised by device code after a device-pointer call returned (today: the bounded spins of Encodec's persistent LSTM kernel, which
 * needs its workgroups co-resident).  Non-blocking: call it once the caller knows the stream is idle (after its own stream / device
 * synchronisation); nc_codec_synchronize and the host-pointer entry points check by themselves, and every device-pointer call checks
 * for a failure of the previous one first.  NC_EDEVICE = the results of that call are invalid; the handle has switched to the
 * step-wise kernels, so repeating the call succeeds (the host-pointer entry points repeat it themselves). */
NC_API nc_status nc_codec_check_errors(nc_codec* h);
/* Encodec handles: which LSTM kernels the handle runs (SLSTM.cs:40-57) -- *stepwise = 1 once a persistent launch has timed out (or the
 * device cannot hold one) -- and how many such timeouts this handle has seen.  Persistent LSTM sections of DIFFERENT handles on one
 * device are ordered one after the other on the GPU (a per-device ticket inside the engine: their workgroups must be co-resident and two
 * handles' worth do not fit), so concurrent handles keep the persistent kernels; only other PROCESSES sharing the device can starve them. */
NC_API nc_status nc_encodec_lstm_stats(const nc_codec* h, int32_t* stepwise, int64_t* timeouts);

/* Shape helper for DAC.Preprocess (Models/DAC.cs:141-154): padded length and frame count T'. */
NC_API nc_status nc_dac_query(const nc_codec* h, int64_t T, int64_t* T_padded, int64_t* frames);

/* replaces: DAC.Encode(Tensor audio[B,1,T], int? nQuantizers, int? sampleRate)   Models/DAC.cs:163-181
 *   pcm        [B,1,T] float32
 *   sample_rate 0 = model rate; a mismatch returns NC_EINVAL (ArgumentException, DAC.cs:146)
 *   n_q        0 = all codebooks (the null overload, ResidualVectorQuantizer.cs:54-103)
 *   codes      [B,n_q,T'] int64            (required)
 *   z          [B,latent,T'] float32       (nullable; the quantized latents zQ)
 *   latents    [B,n_q*codebook_dim,T']     (nullable; the projected latents zE)
 * The two loss outputs of the reference are constant 0 in inference (D5) and are not returned. */
NC_API nc_status nc_dac_encode(nc_codec* h, const float* pcm, int32_t B, int64_t T, int32_t sample_rate, int32_t n_q,
                               int64_t* codes, float* z, float* latents);
NC_API nc_status nc_dac_encode_dev(nc_codec* h, const float* pcm, int32_t B, int64_t T, int32_t sample_rate, int32_t n_q,
                                   int64_t* codes, float* z, float* latents);

/* replaces: DAC.Decode(Tensor z[B,latent,T'])  -> [B,1,L]   Models/DAC.cs:231-234 (no trim: D6).  L = T'*hop when every decoder stride is
 * even (the 44.1 kHz presets); each DecoderBlock maps L -> (L - 1) s - 2 ceil(s / 2) + 2 s (DecoderBlock.cs:20-44), so a stride-5 block
 * (the 16 / 24 kHz presets, DACConfig.cs:115-135) yields 5 L - 1: a caller sizes `pcm` for T'*hop (= nc_dac_query's T_padded), never less. */
NC_API nc_status nc_dac_decode(nc_codec* h, const float* z, int32_t B, int64_t frames, float* pcm);
NC_API nc_status nc_dac_decode_dev(nc_codec* h, const float* z, int32_t B, int64_t frames, float* pcm);

/* replaces: DAC.FromCodes(Tensor codes[B,n_q,T']) -> z[B,latent,T']   Models/DAC.cs:101-106 */
NC_API nc_status nc_dac_from_codes(nc_codec* h, const int64_t* codes, int32_t B, int32_t n_q, int64_t frames, float* z);
NC_API nc_status nc_dac_from_codes_dev(nc_codec* h, const int64_t* codes, int32_t B, int32_t n_q, int64_t frames, float* z);

/* replaces: Dia.Decode(Tensor audioCodes[T, n_q])  Models/Dia.cs:973-981  (FromCodes(codes.unsqueeze(0).transpose(1, 2)) -> Decode) and
 *           AudioUtils.Decode(DAC, codes)           Modules/Dia/AudioUtils.cs:189-199, batched: codes_tq int64 [B, T', n_q] (Dia's layout)
 *           -> pcm [B, 1, L] (L as for nc_dac_decode).  The transpose to DAC's [B, n_q, T'] is a device kernel.
 * replaces: Dia.Encode(Tensor audio[1, T])          Models/Dia.cs:989-1002 (Encode -> squeeze(0).transpose(0, 1)), batched:
 *           pcm [B, 1, T] -> codes_tq int64 [B, T', n_q] (all codebooks) */
NC_API nc_status nc_dac_decode_code_matrix(nc_codec* h, const int64_t* codes_tq, int32_t B, int64_t frames, int32_t n_q, float* pcm);
NC_API nc_status nc_dac_decode_code_matrix_dev(nc_codec* h, const int64_t* codes_tq, int32_t B, int64_t frames, int32_t n_q, float* pcm);
NC_API nc_status nc_dac_encode_code_matrix(nc_codec* h, const float* pcm, int32_t B, int64_t T, int32_t sample_rate, int64_t* codes_tq);
NC_API nc_status nc_dac_encode_code_matrix_dev(nc_codec* h, const float* pcm, int32_t B, int64_t T, int32_t sample_rate, int64_t* codes_tq);

/* ----------------------------------------------------------------------------------------- SNAC
 * replaces: new SNAC(SNACConfig)                   NeuralCodecs.Torch/Models/SNAC.cs:34-63
 *           fields consumed                        NeuralCodecs.Torch/Config/SNAC/SNACConfig.cs:40-100 */
typedef struct {
    int32_t sample_rate;      /* 24000 */
    int32_t encoder_dim;      /* 48 */
    int32_t n_encoder_rates;  /* 4 */
    int32_t encoder_rates[8]; /* 2,4,8,8 */
    int32_t decoder_dim;      /* 1024 */
    int32_t n_decoder_rates;
    int32_t decoder_rates[8]; /* 8,8,4,2 */
    int32_t latent_dim;       /* 0 => encoder_dim * 2^n_encoder_rates (SNAC.cs:41) */
    int32_t attn_window_size; /* 0 = no LocalMHA (24 kHz); 32 for the 32/44 kHz presets */
    int32_t codebook_size;    /* 4096 */
    int32_t codebook_dim;     /* 8 */
    int32_t n_vq_strides;
    int32_t vq_strides[8];    /* 4,2,1 */
    int32_t noise;            /* NoiseBlock in every decoder block */
    int32_t depthwise;        /* depthwise k7 convolutions */
} nc_snac_config;

NC_API nc_status nc_snac_create(const nc_snac_config* cfg, int device_index, nc_codec** out);

/* Shape helper for SNAC.Preprocess (Models/SNAC.cs:70-80): padded length = multiple of hop*lcm(vq_strides[0], window),
 * frames T' = padded/hop, per-level code widths T'/stride_i (level_widths has room for 8), decoded length. */
NC_API nc_status nc_snac_query(const nc_codec* h, int64_t T, int64_t* T_padded, int64_t* frames, int32_t* n_levels,
                               int64_t* level_widths, int64_t* decoded_len);

/* replaces: SNAC.Encode(float[])   Models/SNAC.cs:129-150 and the encode half of forward :91-106  (Preprocess pads; for the
 * Tensor overload exactly as written see nc_snac_encode_tensor below; the two agree on lengths that are already multiples)
 *   pcm   [B,1,T] float32
 *   codes [B, sum_i T'/stride_i] int64: the levels of one clip side by side, coarse first (the reference returns a List of
 *         [B, T'/stride_i] tensors; nc_snac_query gives the widths)
 *   z / zq nullable [B, latent, T']: encoder output / quantized latents */
NC_API nc_status nc_snac_encode(nc_codec* h, const float* pcm, int32_t B, int64_t T, int64_t* codes, float* z, float* zq);
NC_API nc_status nc_snac_encode_dev(nc_codec* h, const float* pcm, int32_t B, int64_t T, int64_t* codes, float* z, float* zq);

/* replaces: SNAC.Encode(Tensor) AS WRITTEN   Models/SNAC.cs:113-122 (deviation D7: Preprocess's result is dropped and the encoder
 * runs on the UN-padded tensor).  Frames T' follow the strided convs' floor lengths (nc_snac_query_tensor); where the reference
 * throws (T' not a multiple of every vq stride: repeat_interleave + add shape mismatch, VectorQuantizer.cs:99-101; LocalMHA
 * window reshape, LocalMHA.cs:84-91) the call returns NC_EINVAL.  Same buffer layout as nc_snac_encode with T' from the query. */
NC_API nc_status nc_snac_query_tensor(const nc_codec* h, int64_t T, int64_t* frames, int32_t* n_levels, int64_t* level_widths);
NC_API nc_status nc_snac_encode_tensor(nc_codec* h, const float* pcm, int32_t B, int64_t T, int64_t* codes, float* z, float* zq);
NC_API nc_status nc_snac_encode_tensor_dev(nc_codec* h, const float* pcm, int32_t B, int64_t T, int64_t* codes, float* z, float* zq);

/* replaces: ResidualVectorQuantizer.FromCodes      Modules/SNAC/ResidualVectorQuantizer.cs:100-135 */
NC_API nc_status nc_snac_from_codes(nc_codec* h, const int64_t* codes, int32_t B, int64_t frames, float* zq);
NC_API nc_status nc_snac_from_codes_dev(nc_codec* h, const int64_t* codes, int32_t B, int64_t frames, float* zq);

/* replaces: SNAC.Decode(List<Tensor> codes)        Models/SNAC.cs:157-192 -> [B,1,decoded_len]
 *   noise: the NoiseBlock inputs, one [B,1,T_i] block per decoder stage laid end to end (reference: torch.randn at inference,
 *          NoiseBlock.cs:41, hence not reproducible); NULL = draw N(0,1) on the device from `seed` (counter-based). */
NC_API nc_status nc_snac_decode(nc_codec* h, const int64_t* codes, int32_t B, int64_t frames, const float* noise, uint64_t seed,
                                float* pcm);
NC_API nc_status nc_snac_decode_dev(nc_codec* h, const int64_t* codes, int32_t B, int64_t frames, const float* noise,
                                    uint64_t seed, float* pcm);
/* total number of noise floats nc_snac_decode consumes for (B, frames) */
NC_API nc_status nc_snac_noise_len(const nc_codec* h, int32_t B, int64_t frames, int64_t* n);

/* replaces: SNAC.ProcessAudio(float[] audioData, int sampleRate)   Models/SNAC.cs:255-282 with ResampleAudio :284-308
 * One upload, then resample (when sample_rate differs from the model's; binary64 position arithmetic as nc_audio_resample_linear_dev)
 * -> forward (Preprocess pad, encode, quantize, decode, narrow to the resampled length: SNAC.cs:91-106) on the device, one download.
 *   audio [n] float32 mono; NULL or n <= 0 -> NC_EINVAL (ArgumentException "Audio data cannot be empty", SNAC.cs:257-258)
 *   noise / seed as nc_snac_decode (frames from nc_snac_query on the resampled length, B = 1)
 *   out   [n_out], n_out from nc_snac_process_audio_len (= n when the rates agree, else nc_audio_resample_len) */
NC_API nc_status nc_snac_process_audio_len(const nc_codec* h, int64_t n, int32_t sample_rate, int64_t* n_out);
NC_API nc_status nc_snac_process_audio(nc_codec* h, const float* audio, int64_t n, int32_t sample_rate, const float* noise, uint64_t seed,
                                       float* out);

/* -------------------------------------------------------------------------------------- Encodec
 * replaces: new Encodec(EncodecConfig)             NeuralCodecs.Torch/Models/Encodec.cs:46-90
 * Only channels / dimension / norm / causal reach SEANet in the reference (Encodec.cs:57-68, deviation D11); the other SEANet
 * fields are its hard defaults (SEANetEncoder.cs:37-56) and are carried here so reduced-width test models can be built. */
typedef struct {
    int32_t sample_rate;        /* 24000 | 48000 */
    int32_t channels;           /* 1 | 2 */
    int32_t dimension;          /* 128 (HiddenSize == CodebookDim) */
    int32_t n_filters;          /* 32 */
    int32_t n_ratios;           /* 4 */
    int32_t ratios[8];          /* 8,5,4,2 (decoder order; the encoder runs them reversed) */
    int32_t lstm_layers;        /* 2 */
    int32_t compress;           /* 2 */
    int32_t kernel_size;        /* 7 */
    int32_t last_kernel_size;   /* 7 */
    int32_t residual_kernel_size; /* 3 */
    int32_t time_group_norm;    /* 0: weight_norm (24 kHz), 1: GroupNorm(1,C) after every conv (48 kHz) */
    int32_t causal;             /* 24 kHz: 1 */
    int32_t normalize;          /* 48 kHz: 1 (per-frame RMS scale) */
    int32_t segment_length;     /* samples; 0 = no segmentation (24 kHz); 48 kHz: 48000 */
    int32_t segment_stride;     /* samples; 48 kHz: 47520 (1 % overlap) */
    int32_t codebook_size;      /* 1024 */
    int32_t n_codebooks;        /* quantizer layers built: 1000*max(bandwidths)/(frame_rate*10) (Encodec.cs:70-71): 32 | 16 */
    int32_t frame_rate;         /* ceil(sample_rate / hop) */
    float bandwidth;            /* target kbps -> n_q = max(1, floor(bandwidth*1000 / (log2(codebook_size)*frame_rate))) */
} nc_encodec_config;

NC_API nc_status nc_encodec_create(const nc_encodec_config* cfg, int device_index, nc_codec** out);
/* replaces: Encodec.SetTargetBandwidth(float)      Models/Encodec.cs:409-419 (the caller validates membership in TargetBandwidths) */
NC_API nc_status nc_encodec_set_bandwidth(nc_codec* h, float bandwidth_kbps);

/* Frame layout of Encodec.Encode for clips of T samples: number of EncodedFrames (segments), codebooks in use n_q, frames T'_f of
 * every segment (frame_lens has room for `cap` entries), decoded length of Decode (before forward()'s trim to T). */
NC_API nc_status nc_encodec_query(const nc_codec* h, int64_t T, int32_t* n_frames, int32_t* n_q, int64_t* frame_lens, int32_t cap,
                                  int64_t* decoded_len);
/* The reference's Decode(List<EncodedFrame>) takes no clip length (Models/Encodec.cs:213-235: the frames alone fix the output,
 * (n-1)*stride + decoded(last frame)).  This gives the smallest clip length T whose segment layout is exactly `n_frames` segments
 * with frame_lens[i] code frames in segment i (with overlapping segments the last TWO can be short) -- the T to hand to
 * nc_encodec_decode for such a list.  NC_EINVAL when no clip length produces that layout. */
NC_API nc_status nc_encodec_clip_length(const nc_codec* h, int32_t n_frames, const int64_t* frame_lens, int64_t* T);

/* replaces: Encodec.Encode(Tensor x[B,C,T]) -> List<EncodedFrame>     Models/Encodec.cs:259-285, EncodeFrame :457-489
 *   codes  int64: the EncodedFrame.Codes tensors [B,n_q,T'_f] laid end to end in segment order
 *   scales float32 [n_frames,B] (EncodedFrame.Scale; written only when normalize) -- nullable otherwise
 *   emb    nullable: encoder outputs [B,dimension,T'_f] end to end (test hook) */
NC_API nc_status nc_encodec_encode(nc_codec* h, const float* pcm, int32_t B, int64_t T, int64_t* codes, float* scales, float* emb);
NC_API nc_status nc_encodec_encode_dev(nc_codec* h, const float* pcm, int32_t B, int64_t T, int64_t* codes, float* scales, float* emb);

/* replaces: Encodec.Decode(List<EncodedFrame>) -> [B,C,decoded_len]    Models/Encodec.cs:213-235, DecodeFrame :436-455,
 * DSP.LinearOverlapAdd AudioTools/AudioTensorDSP.cs:161-261.  T is the clip length the frames were encoded from (it fixes the
 * segment layout); n_q the number of codebooks present in `codes`. */
NC_API nc_status nc_encodec_decode(nc_codec* h, const int64_t* codes, const float* scales, int32_t B, int64_t T, int32_t n_q, float* pcm);
NC_API nc_status nc_encodec_decode_dev(nc_codec* h, const int64_t* codes, const float* scales, int32_t B, int64_t T, int32_t n_q,
                                       float* pcm);

/* ------------------------------------------------------------------------------- code containers
 * Device-side bit packing of code tensors: the wire layout of the reference's BitPacker / BitUnpacker
 * (Modules/Encodec/BitPacker.cs: values LSB-first into a little-endian bit stream, `bits` per value, flushed to a whole byte) in the
 * order EncodecCompressor writes a frame (EncodecCompressor.cs:170-181: t outer, codebook k inner).  One packed row per clip.
 *   codes  [B,K,T] int64 (device)         packed [B, nc_packed_bytes(K*T, bits)] uint8 (device)
 * Used for `.ecdc` payloads without the language model and to shrink the multi-GPU code all-gather 64/bits times. */
NC_API int64_t nc_packed_bytes(int64_t n_values, int32_t bits);
NC_API nc_status nc_pack_codes_dev(int device_index, const int64_t* codes, int32_t B, int32_t K, int64_t T, int32_t bits, uint8_t* packed,
                                   void* hip_stream);
NC_API nc_status nc_unpack_codes_dev(int device_index, const uint8_t* packed, int32_t B, int32_t K, int64_t T, int32_t bits,
                                     int64_t* codes, void* hip_stream);
/* host-pointer, synchronous variants */
NC_API nc_status nc_pack_codes(int device_index, const int64_t* codes, int32_t B, int32_t K, int64_t T, int32_t bits, uint8_t* packed);
NC_API nc_status nc_unpack_codes(int device_index, const uint8_t* packed, int32_t B, int32_t K, int64_t T, int32_t bits, int64_t* codes);

/* ------------------------------------------------------------------------------- audio pre / post
 * The host-side steps either side of the tensor path as device kernels, so a float[]-level caller can keep its
 * audio in HBM from the decoded WAV bytes to the codec and back.  Device pointers, asynchronous on `hip_stream` (NULL = null
 * stream).  Arithmetic follows the reference statement by statement (bit-identical results):
 *   pcm16_to_float   replaces AudioUtils.AudioBytesToFloatArray (Core/Utils/AudioUtils.cs:13-36): out = int16 * (1/32768f), same
 *                    order; planar != 0 writes channel-major [channels][n_frames] like NAudioUtils.cs:94-104 / Examples/Program.cs:391-401
 *   float_to_pcm16   replaces AudioUtils.FloatArrayToAudioBytes (AudioUtils.cs:172-186) with the clamp of Dia.SaveAudio
 *                    (Models/Dia.cs:918-923): (short)(clamp(x,-1,1) * 32767), truncating
 *   mix_to_mono      replaces AudioUtils.ConvertToMono (AudioUtils.cs:45-61): interleaved [n_frames][channels] -> [n_frames]
 *   interleave       replaces AudioUtils.DeinterleaveToInterleave (AudioUtils.cs:90-101), any channel count
 *   deinterleave     replaces AudioUtils.InterleaveToDeinterleave (AudioUtils.cs:204-219)
 *   resample_linear  replaces AudioUtils.ResampleLinear (AudioUtils.cs:329-354) == SNAC.ResampleAudio (Models/SNAC.cs:284-308):
 *                    [B][n_in] -> [B][nc_audio_resample_len(n_in, src, dst)], binary64 position arithmetic */
NC_API int64_t nc_audio_resample_len(int64_t n_in, int32_t src_rate, int32_t dst_rate);
NC_API nc_status nc_audio_pcm16_to_float_dev(int device_index, const int16_t* pcm, int64_t n_frames, int32_t channels, int32_t planar,
                                             float* out, void* hip_stream);
NC_API nc_status nc_audio_float_to_pcm16_dev(int device_index, const float* in, int64_t n, int16_t* out, void* hip_stream);
NC_API nc_status nc_audio_mix_to_mono_dev(int device_index, const float* in, int64_t n_frames, int32_t channels, float* out, void* hip_stream);
NC_API nc_status nc_audio_interleave_dev(int device_index, const float* planar, int64_t n_frames, int32_t channels, float* out,
                                         void* hip_stream);
NC_API nc_status nc_audio_deinterleave_dev(int device_index, const float* interleaved, int64_t n_frames, int32_t channels, float* out,
                                           void* hip_stream);
NC_API nc_status nc_audio_resample_linear_dev(int device_index, const float* in, int32_t B, int64_t n_in, int32_t src_rate, int32_t dst_rate,
                                              float* out, void* hip_stream);

/* ------------------------------------------------------------------------------------ multi-GPU groups
 * The reference has no multi-device code; its callers batch clips (Examples/Program.cs:228-322, Models/Dia.cs:973-1002).  The path
 * shards over clips with no data-path exchange (every operator is per sample), so a group = one codec replica per GPU + ONE
 * collective: the RCCL all-gather of the emitted int64 codes (DAC [B,n_q,T'], SNAC.Encode's List<Tensor> as the levels of a clip side
 * by side: Models/SNAC.cs:129-150), issued on a side stream per device behind the encode, so the local decode overlaps it.
 * librccl is opened at run time by these entry points only.
 *   rank  mode: one process per GPU; rank 0 draws a unique id (nc_group_unique_id) and hands it to the others out of band.
 *   local mode: one process drives ndev GPUs (handles[i] created on device i): the natural layout of a single C# host process.
 * Lifetime: a group BORROWS its codec handles -- destroy the group (nc_group_destroy) before the codecs it was created over. */
typedef struct nc_group nc_group;
#define NC_GROUP_UID_BYTES 128
NC_API nc_status nc_group_unique_id(void* uid /* [NC_GROUP_UID_BYTES] */);
NC_API nc_status nc_group_create_rank(int32_t world, int32_t rank, const void* uid, nc_codec* local, nc_group** out);
NC_API nc_status nc_group_create_local(int32_t ndev, nc_codec* const* handles, nc_group** out);
/* local mode with options.  NC_GROUP_PEER_COPY: the all-gather is moved by peer copies (hipMemcpyPeerAsync on the members' side streams:
 * member e pulls slot d from member d's buffer behind d's "slot final" event) instead of RCCL -- same slots, same results, no librccl; the
 * members may then SHARE a device (an RCCL communicator cannot hold two ranks of one GPU), e.g. eight handles on one GPU split a batch
 * eight ways exactly as eight GPUs would.  flags = 0 is nc_group_create_local. */
#define NC_GROUP_PEER_COPY 1u
NC_API nc_status nc_group_create_local_ex(int32_t ndev, nc_codec* const* handles, uint32_t flags, nc_group** out);
NC_API nc_status nc_group_destroy(nc_group* g);
NC_API nc_status nc_group_info(const nc_group* g, int32_t* world, int32_t* rank /* -1 in local mode */);
/* Payload of the code all-gather: bits = 0 (default) moves the int64 codes as they are; 1..24 moves them bit-packed in the wire layout of
 * the reference's BitPacker (Modules/Encodec/BitPacker.cs: `bits` per value, LSB first, one packed row per clip; bits = 10 for 1024-entry
 * codebooks, 12 for SNAC's 4096) -- 64 / bits times fewer bytes on xGMI; pack and unpack run on the device around the collective and
 * the caller's tensors stay int64.  A code that does not fit `bits` is truncated to its low bits: pick bits >= log2(codebook size).
 * Rank mode: EVERY rank must set the same value before the next gather (the byte counts of the collective follow from it; a mismatch is
 * not detectable without another collective and hangs the gather). */
NC_API nc_status nc_group_set_code_bits(nc_group* g, int32_t bits);
/* rank mode, device pointers, asynchronous: encode this rank's B_local clips (nc_dac_encode_dev / nc_snac_encode_dev semantics) with the
 * codes written straight into slot `rank` of codes_all [world*B_local, ...], then the in-place all-gather on the group's side stream.
 * B_local must be the same on every rank (a ragged batch: pad the short shards to the longest one and drop the padding rows).
 * nc_group_wait makes the codec's stream wait for the gather (no host synchronisation). */
NC_API nc_status nc_group_dac_encode_allgather_dev(nc_group* g, const float* pcm, int32_t B_local, int64_t T, int32_t sample_rate, int32_t n_q,
                                                   int64_t* codes_all, float* z_local, float* latents_local);
NC_API nc_status nc_group_snac_encode_allgather_dev(nc_group* g, const float* pcm, int32_t B_local, int64_t T, int64_t* codes_all);
NC_API nc_status nc_group_wait(nc_group* g);
/* local mode, host pointers, synchronous: B_total clips split into contiguous blocks over the devices (the first B_total % ndev devices
 * hold one clip more; slots of the gathered buffer are sized for the largest block, so ragged batches gather in one collective);
 * codes [B_total, ...] come back gathered in clip order; z nullable [B_total, latent, T'] (each device returns its block).  One host
 * thread per device stages its block through pinned memory, so the uploads and encodes of all devices overlap. */
NC_API nc_status nc_group_dac_encode_allgather(nc_group* g, const float* pcm, int32_t B_total, int64_t T, int32_t sample_rate, int32_t n_q,
                                               int64_t* codes, float* z);
NC_API nc_status nc_group_snac_encode_allgather(nc_group* g, const float* pcm, int32_t B_total, int64_t T, int64_t* codes);
/* local mode, device pointers, asynchronous (the single-host layout of Examples/Program.cs:228-322 with the batch already split and
 * resident): pcm[d] [B_local[d],1,T] lives on member d's GPU and is encoded on member d's stream (nc_dac_encode_dev / nc_snac_encode_dev
 * semantics; z_local / latents_local nullable arrays of per-device outputs) with the codes written into slot d of codes_all[d], member
 * d's OWN copy of the gathered tensor [ndev * B_max, ...] (B_max = the largest block; a shorter block's slot is zero-padded); the grouped
 * in-place all-gather runs on the members' side streams, so every device ends up holding every block.  Nothing waits on the host:
 * nc_group_wait orders each codec's stream behind its gather; the decode of the local block can be queued before it. */
NC_API nc_status nc_group_dac_encode_allgather_local_dev(nc_group* g, const float* const* pcm, const int32_t* B_local, int64_t T,
                                                         int32_t sample_rate, int32_t n_q, int64_t* const* codes_all, float* const* z_local,
                                                         float* const* latents_local);
NC_API nc_status nc_group_snac_encode_allgather_local_dev(nc_group* g, const float* const* pcm, const int32_t* B_local, int64_t T,
                                                          int64_t* const* codes_all);

/* Encodec (Models/Encodec.cs:259-285: Encode emits one EncodedFrame per segment).  A rank's block is what nc_encodec_encode_dev writes: the
 * frames' code tensors [B_local, n_q, T'_f] end to end in segment order (n_q * sum T'_f values per clip) and scales [n_frames, B_local].
 * codes_all = [world][that block], scales_all = [world][n_frames][B_local] (nullable unless the model normalises): two collectives on the
 * side stream; rank r's frames are the views of block r.  Equal B_local everywhere.  nc_group_wait as above. */
NC_API nc_status nc_group_encodec_encode_allgather_dev(nc_group* g, const float* pcm, int32_t B_local, int64_t T, int64_t* codes_all,
                                                       float* scales_all);
NC_API nc_status nc_group_encodec_encode_allgather_local_dev(nc_group* g, const float* const* pcm, const int32_t* B_local, int64_t T,
                                                             int64_t* const* codes_all, float* const* scales_all);

/* ------------------------------------------------------------------------------------ profiling
 * Per-kernel-class timing with HIP events recorded on the handle's stream around every launch
 * (used by bench.py for the roofline object).  Classes are stable small integers. */
typedef enum {
    NC_KC_CONV_K7 = 0,   /* dilated k=7 residual-unit convolutions (MFMA implicit GEMM) */
    NC_KC_CONV_K1 = 1,   /* 1x1 channel-mix convolutions */
    NC_KC_CONV_DOWN = 2, /* strided down-sampling convolutions */
    NC_KC_CONV_UP = 3,   /* transposed (polyphase) up-sampling convolutions */
    NC_KC_CONV_MISC = 4, /* stem / head / k3 / decoder-input convolutions */
    NC_KC_RVQ = 5,       /* codebook argmin + gather kernels */
    NC_KC_ELEM = 6,      /* HBM-bound element-wise kernels: pooling, RVQ update, pad/activate, scale, embedding sum, overlap-add */
    NC_KC_DWCONV = 7,    /* depthwise k=7 convolutions (SNAC ResidualUnit.cs:25-60): 8 B per output */
    NC_KC_NORM = 8,      /* LayerNorm over channels (LocalMHA.cs:56) / GroupNorm(1,C) statistics (NormConv1d.cs:155) */
    NC_KC_ATTN = 9,      /* windowed rotary attention (LocalMHA.cs:84-113) */
    NC_KC_LSTM = 10,     /* LSTM recurrence (SLSTM.cs:40-57); the input projections are counted under conv_k1 */
    NC_KC_STEM = 11,     /* Cin <= 2 stem convolutions (Encoder.cs:31): a streaming store of Cout rows */
    NC_KC_HEAD = 12,     /* Cout <= 2 PCM-head convolutions (+tanh, Decoder.cs:44-46): a streaming read of Cin rows */
    NC_KC_COUNT = 13
} nc_kernel_class;

typedef struct {
    int64_t launches;
    double ms;          /* sum of event-to-event durations */
    double flops;       /* algorithmic flops of those launches (2*Cout*Cin/g*K*Tout*B) */
    double bytes;       /* algorithmic HBM bytes (input read once + output written once + weights once) */
} nc_profile_entry;

NC_API nc_status nc_codec_profile_enable(nc_codec* h, int32_t on);
NC_API nc_status nc_codec_profile_reset(nc_codec* h);
/* synchronises the stream, resolves pending events, fills out[NC_KC_COUNT] */
NC_API nc_status nc_codec_profile_read(nc_codec* h, nc_profile_entry* out);

/* ------------------------------------------------------------------------ op-level test hooks
 * Host-pointer, synchronous single-operator entry points over the same kernels the codecs use.
 * They exist so the parity tests can compare each HIP kernel with the oracle in isolation. */
typedef struct {
    int32_t B, Cin, Cout, K, stride, pad, dil, out_pad;
    int64_t Tin;
    int32_t transposed;      /* 0: conv1d weight [Cout,Cin,K]; 1: conv_transpose1d weight [Cin,Cout,K] */
    int32_t tanh_out;        /* apply tanh after bias/residual */
} nc_conv_desc;

/* y = epilogue(conv(snake_in?(x))), weight is the already-folded dense weight.
 * alpha_in [Cin] / alpha_out [Cout] / bias [Cout] / residual [B,Cout,Tout] are nullable. */
NC_API nc_status nc_op_conv1d(int device_index, const nc_conv_desc* d, const float* x, const float* weight, const float* bias,
                              const float* alpha_in, const float* alpha_out, const float* residual, float* y, int64_t* Tout);
/* Timing hook over the same launch path: runs the convolution `iters` times on device-resident seeded random data
 * (fuse bit0: Snake on input, bit1: Snake on output, bit2: residual add) and returns the HIP-event average per launch. */
NC_API nc_status nc_op_conv1d_bench(int device_index, const nc_conv_desc* d, int32_t fuse, int32_t iters, double* avg_ms);
/* One DAC ResidualUnit (ResidualUnit.cs:24-59): y = x + conv1(snake_a2(conv7_dil(snake_a1(x)))) on x [B,C,T].
 * Dense weights w7 [C,C,7], w1 [C,C,1].  fused != 0 requests the single-launch kernel (NC_EUNSUPPORTED when the shape has
 * none); iters > 0 additionally times `iters` repetitions with HIP events into *avg_ms (nullable when iters == 0). */
NC_API nc_status nc_op_res_unit(int device_index, int32_t B, int32_t C, int64_t T, int32_t dil, const float* x, const float* w7,
                                const float* b7, const float* a1, const float* a2, const float* w1, const float* b1,
                                int32_t fused, float* y, int32_t iters, double* avg_ms);
/* one VQ stage on projected latents z_e [B,D,T] against codebook [N,D] -> idx [B,T], st [B,D,T] */
NC_API nc_status nc_op_vq_argmin(int device_index, const float* z_e, int32_t B, int32_t D, int64_t T, const float* codebook,
                                 int32_t N, int64_t* idx, float* st);
/* the Encodec Euclidean RVQ (Modules/Encodec/ResidualVectorQuantizer.cs:133-157 over EuclideanCodebook.cs:155-182) on residual [B,D,T]
 * with codebooks [n_q,N,D] -> codes [B,n_q,T], residual_out [B,D,T] (nullable; the residual after the last stage in form 0, the input in form 1, whose residual never
 * leaves LDS).  form 0: one launch per stage; form 1: the all-stages matrix-core kernel (D == 128 and N = 512 or 1024, else NC_EUNSUPPORTED). */
NC_API nc_status nc_op_euclid_rvq(int device_index, const float* residual, int32_t B, int32_t D, int64_t T, const float* codebooks,
                                  int32_t n_q, int32_t N, int32_t form, int64_t* codes, float* residual_out);
/* weight-norm fold w = v/(||v||+1e-7)*g over dim-0 slices (host-side, what load_weights does) */
NC_API nc_status nc_op_fold_weight_norm(const float* v, const float* g, int64_t d0, int64_t inner, float* w);

#ifdef __cplusplus
}
#endif
#endif /* NC_MI355X_H */
