// Host-side pre/post steps either side of the tensor path, as small device kernels: 16-bit PCM <-> float,
// mono mix, stereo (de)interleave and the linear resampler the demo flows use, so a float[]-level caller stays on the device.
// All HBM-bound element-wise work: one thread per output sample, coalesced.  Arithmetic follows the reference statement by
// statement (same operand types and order), so results are bit-identical to the oracle (oracle/audio_ref.py):
//   AudioBytesToFloatArray   Core/Utils/AudioUtils.cs:13-36      (int16 * (1/32768f); planar variant Utils/NAudioUtils.cs:94-104)
//   FloatArrayToAudioBytes   Core/Utils/AudioUtils.cs:172-186    ((short)(x * 32767), with the clamp of Models/Dia.cs:918-923)
//   ConvertToMono            Core/Utils/AudioUtils.cs:45-61      (float sum in channel order, / channels)
//   DeinterleaveToInterleave / InterleaveToDeinterleave  Core/Utils/AudioUtils.cs:90-101,204-219
//   ResampleLinear           Core/Utils/AudioUtils.cs:329-354 == Models/SNAC.cs:284-308  (binary64 position / fraction)
#include "nc_common.h"

namespace nc {

__global__ void pcm16_to_float_kernel(const int16_t* __restrict__ in, int64_t n_frames, int channels, int planar, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // index into the interleaved input
    if (i >= n_frames * channels) return;
    const float v = (float)in[i] * (1.0f / 32768.0f);
    if (!planar) {
        out[i] = v;
    } else {
        const int64_t f = i / channels;
        const int c = (int)(i - f * channels);
        out[(int64_t)c * n_frames + f] = v;
    }
}

__global__ void float_to_pcm16_kernel(const float* __restrict__ in, int64_t n, int16_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float c = fmaxf(-1.0f, fminf(1.0f, in[i]));
    out[i] = (int16_t)(int)(c * 32767.0f);   // conversion truncates toward zero, like the C# cast
}

__global__ void mix_to_mono_kernel(const float* __restrict__ in, int64_t n_frames, int channels, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_frames) return;
    float sum = 0.0f;
    for (int c = 0; c < channels; ++c) sum += in[i * channels + c];
    out[i] = sum / (float)channels;
}

// planar [channels][n_frames] -> interleaved [n_frames][channels]
__global__ void interleave_kernel(const float* __restrict__ in, int64_t n_frames, int channels, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_frames * channels) return;
    const int64_t f = i / channels;
    const int c = (int)(i - f * channels);
    out[i] = in[(int64_t)c * n_frames + f];
}
__global__ void deinterleave_kernel(const float* __restrict__ in, int64_t n_frames, int channels, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // index into the planar output
    if (i >= n_frames * channels) return;
    const int c = (int)(i / n_frames);
    const int64_t f = i - (int64_t)c * n_frames;
    out[i] = in[f * channels + c];
}

__global__ void resample_linear_kernel(const float* __restrict__ in, int B, int64_t n_in, double ratio, int64_t n_out, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * n_out) return;
    const int64_t b = i / n_out, o = i - b * n_out;
    const float* x = in + b * n_in;
    const double position = (double)o / ratio;
    const int64_t index = (int64_t)position;
    const double fraction = position - (double)index;
    float y;
    if (index >= n_in - 1) y = x[n_in - 1];
    else y = (float)(((1.0 - fraction) * (double)x[index]) + (fraction * (double)x[index + 1]));
    out[i] = y;
}

static void audio_device(int device_index) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) fail(NC_EDEVICE, "no HIP device available (the engine has no CPU fallback)");
    if (device_index < 0 || device_index >= n) fail(NC_EINVAL, "device index out of range");
    NC_HIP(hipSetDevice(device_index));
}
static unsigned grid_for(int64_t n) {
    if (n > (int64_t)0x7fffffff * 256) fail(NC_EINVAL, "too many samples for one launch");
    return (unsigned)((n + 255) / 256);
}

}  // namespace nc

using namespace nc;

#define NC_AUDIO_GUARD(body)            \
    try {                               \
        body;                           \
        NC_HIP(hipGetLastError());      \
        return NC_OK;                   \
    } catch (const Error& e) {          \
        set_last_error(e.what());       \
        return e.code;                  \
    }

extern "C" {

int64_t nc_audio_resample_len(int64_t n_in, int32_t src_rate, int32_t dst_rate) {
    if (n_in < 0 || src_rate <= 0 || dst_rate <= 0) return -1;
    const double ratio = (double)dst_rate / (double)src_rate;
    return (int64_t)(int32_t)((double)n_in * ratio);   // `(int)(input.Length * ratio)`
}

nc_status nc_audio_pcm16_to_float_dev(int device_index, const int16_t* pcm, int64_t n_frames, int32_t channels, int32_t planar, float* out,
                                      void* hip_stream) {
    NC_AUDIO_GUARD({
        if (!pcm || !out) fail(NC_EINVAL, "null pointer");
        if (n_frames <= 0 || channels <= 0) fail(NC_EINVAL, "n_frames and channels must be positive");
        audio_device(device_index);
        hipLaunchKernelGGL(pcm16_to_float_kernel, dim3(grid_for(n_frames * channels)), dim3(256), 0, static_cast<hipStream_t>(hip_stream), pcm,
                           n_frames, channels, planar, out);
    })
}

nc_status nc_audio_float_to_pcm16_dev(int device_index, const float* in, int64_t n, int16_t* out, void* hip_stream) {
    NC_AUDIO_GUARD({
        if (!in || !out) fail(NC_EINVAL, "null pointer");
        if (n <= 0) fail(NC_EINVAL, "n must be positive");
        audio_device(device_index);
        hipLaunchKernelGGL(float_to_pcm16_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(hip_stream), in, n, out);
    })
}

nc_status nc_audio_mix_to_mono_dev(int device_index, const float* in, int64_t n_frames, int32_t channels, float* out, void* hip_stream) {
    NC_AUDIO_GUARD({
        if (!in || !out) fail(NC_EINVAL, "null pointer");
        if (n_frames <= 0 || channels <= 0) fail(NC_EINVAL, "n_frames and channels must be positive");
        audio_device(device_index);
        hipLaunchKernelGGL(mix_to_mono_kernel, dim3(grid_for(n_frames)), dim3(256), 0, static_cast<hipStream_t>(hip_stream), in, n_frames, channels, out);
    })
}

nc_status nc_audio_interleave_dev(int device_index, const float* planar, int64_t n_frames, int32_t channels, float* out, void* hip_stream) {
    NC_AUDIO_GUARD({
        if (!planar || !out) fail(NC_EINVAL, "null pointer");
        if (n_frames <= 0 || channels <= 0) fail(NC_EINVAL, "n_frames and channels must be positive");
        audio_device(device_index);
        hipLaunchKernelGGL(interleave_kernel, dim3(grid_for(n_frames * channels)), dim3(256), 0, static_cast<hipStream_t>(hip_stream), planar, n_frames,
                           channels, out);
    })
}

nc_status nc_audio_deinterleave_dev(int device_index, const float* interleaved, int64_t n_frames, int32_t channels, float* out, void* hip_stream) {
    NC_AUDIO_GUARD({
        if (!interleaved || !out) fail(NC_EINVAL, "null pointer");
        if (n_frames <= 0 || channels <= 0) fail(NC_EINVAL, "n_frames and channels must be positive");
        audio_device(device_index);
        hipLaunchKernelGGL(deinterleave_kernel, dim3(grid_for(n_frames * channels)), dim3(256), 0, static_cast<hipStream_t>(hip_stream), interleaved,
                           n_frames, channels, out);
    })
}

nc_status nc_audio_resample_linear_dev(int device_index, const float* in, int32_t B, int64_t n_in, int32_t src_rate, int32_t dst_rate, float* out,
                                       void* hip_stream) {
    NC_AUDIO_GUARD({
        if (!in || !out) fail(NC_EINVAL, "null pointer");
        if (B <= 0 || n_in <= 0) fail(NC_EINVAL, "B and n_in must be positive");
        if (src_rate <= 0 || dst_rate <= 0) fail(NC_EINVAL, "sample rates must be positive");
        const int64_t n_out = nc_audio_resample_len(n_in, src_rate, dst_rate);
        if (n_out <= 0) fail(NC_EINVAL, "resampled clip would be empty");
        audio_device(device_index);
        const double ratio = (double)dst_rate / (double)src_rate;
        hipLaunchKernelGGL(resample_linear_kernel, dim3(grid_for((int64_t)B * n_out)), dim3(256), 0, static_cast<hipStream_t>(hip_stream), in, B, n_in,
                           ratio, n_out, out);
    })
}

}  // extern "C"
