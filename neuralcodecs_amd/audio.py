"""Host-side audio pre/post steps of the reference's demo flows, run on the device (csrc/nc_audio.hip).

Mirror of the reference's helpers (same names in snake case, same argument meaning):
    pcm16_to_float(pcm, channels, planar)      AudioUtils.AudioBytesToFloatArray   Core/Utils/AudioUtils.cs:13-36
    float_to_pcm16(x)                          AudioUtils.FloatArrayToAudioBytes   Core/Utils/AudioUtils.cs:172-186 (+ Dia.cs:918-923 clamp)
    convert_to_mono(x, channels)               AudioUtils.ConvertToMono            Core/Utils/AudioUtils.cs:45-61
    deinterleave_to_interleave(x, channels)    AudioUtils.DeinterleaveToInterleave Core/Utils/AudioUtils.cs:90-101
    interleave_to_deinterleave(x, channels)    AudioUtils.InterleaveToDeinterleave Core/Utils/AudioUtils.cs:204-219
    resample_linear(x, src, dst)               AudioUtils.ResampleLinear           Core/Utils/AudioUtils.cs:329-354 / SNAC.cs:284-308

torch device tensors stay on the device (zero-copy, torch's current stream); numpy arrays are copied in and out.  There is no CPU
implementation here: without the HIP library and a GPU every call raises.
"""
from __future__ import annotations

import numpy as np

from . import _lib


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise _lib.NcDeviceError("no HIP device available (the engine has no CPU fallback)")
    return torch


def _run(x, np_dtype, out_shape, out_dtype_np, launch):
    """x -> contiguous device tensor, allocate the output, launch(dev_index, x_ptr, out_ptr, stream), hand back in x's kind."""
    torch = _torch()
    was_torch = _is_torch(x)
    if was_torch:
        xt = x.contiguous()
        if not xt.is_cuda:
            xt = xt.cuda()
    else:
        xt = torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np_dtype))).cuda()
    tdt = {np.float32: torch.float32, np.int16: torch.int16}[out_dtype_np]
    out = torch.empty(out_shape, dtype=tdt, device=xt.device)
    stream = torch.cuda.current_stream(xt.device).cuda_stream
    _lib.check(launch(xt.device.index or 0, xt.data_ptr(), out.data_ptr(), stream))
    if was_torch:
        return out
    torch.cuda.synchronize(xt.device)
    return out.cpu().numpy()


def pcm16_to_float(pcm, channels: int = 1, planar: bool = False):
    n = int(np.prod(pcm.shape)) if hasattr(pcm, "shape") else len(pcm)
    if channels <= 0 or n == 0 or n % channels:
        raise ValueError("sample count must be a positive multiple of the channel count")
    L = _lib.lib()
    return _run(pcm, np.int16, (n,), np.float32,
                lambda d, xi, xo, s: L.nc_audio_pcm16_to_float_dev(d, xi, n // channels, channels, 1 if planar else 0, xo, s))


def float_to_pcm16(x):
    n = int(np.prod(x.shape)) if hasattr(x, "shape") else len(x)
    L = _lib.lib()
    return _run(x, np.float32, (n,), np.int16, lambda d, xi, xo, s: L.nc_audio_float_to_pcm16_dev(d, xi, n, xo, s))


def convert_to_mono(x, channels: int):
    n = int(np.prod(x.shape)) if hasattr(x, "shape") else len(x)
    if channels <= 0 or n // channels == 0:
        raise ValueError("not enough samples for one frame")
    L = _lib.lib()
    return _run(x, np.float32, (n // channels,), np.float32,
                lambda d, xi, xo, s: L.nc_audio_mix_to_mono_dev(d, xi, n // channels, channels, xo, s))


def deinterleave_to_interleave(x, channels: int = 2):
    n = int(np.prod(x.shape)) if hasattr(x, "shape") else len(x)
    L = _lib.lib()
    return _run(x, np.float32, (n,), np.float32, lambda d, xi, xo, s: L.nc_audio_interleave_dev(d, xi, n // channels, channels, xo, s))


def interleave_to_deinterleave(x, channels: int = 2):
    n = int(np.prod(x.shape)) if hasattr(x, "shape") else len(x)
    L = _lib.lib()
    return _run(x, np.float32, (n,), np.float32, lambda d, xi, xo, s: L.nc_audio_deinterleave_dev(d, xi, n // channels, channels, xo, s))


def resample_linear(x, src: int, dst: int):
    """[n] or [B, n] -> [n_out] or [B, n_out] with n_out = (int)(n * dst / src)."""
    shape = tuple(x.shape) if hasattr(x, "shape") else (len(x),)
    n_in = shape[-1]
    B = int(np.prod(shape[:-1])) if len(shape) > 1 else 1
    L = _lib.lib()
    n_out = L.nc_audio_resample_len(n_in, src, dst)
    if n_out <= 0:
        raise ValueError("resampled clip would be empty")
    return _run(x, np.float32, shape[:-1] + (n_out,), np.float32,
                lambda d, xi, xo, s: L.nc_audio_resample_linear_dev(d, xi, B, n_in, src, dst, xo, s))
