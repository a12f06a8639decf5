"""TEST INFRASTRUCTURE -- CPU restatement of the reference's host-side audio pre/post helpers.

Only tests/ may import this module; the product path (neuralcodecs_amd/audio.py -> csrc/nc_audio.hip) never does.
Each function follows the reference statement by statement (same operand types, same order of operations):

    pcm16_to_float      Core/Utils/AudioUtils.cs:13-36 (16-bit branch); planar=True: Core/Utils/NAudioUtils.cs:94-104
    float_to_pcm16      Core/Utils/AudioUtils.cs:172-186 with the clamp of Models/Dia.cs:918-923
    mix_to_mono         Core/Utils/AudioUtils.cs:45-61
    interleave          Core/Utils/AudioUtils.cs:90-101
    deinterleave        Core/Utils/AudioUtils.cs:204-219
    resample_linear     Core/Utils/AudioUtils.cs:329-354 == Models/SNAC.cs:284-308

Parity unpinned against the reference itself (no .NET in the image, the reference has no tests); pinned by hand-worked known answers
in tests/test_audio_cpu.py.
"""
import numpy as np


def pcm16_to_float(pcm, channels=1, planar=False):
    pcm = np.asarray(pcm, dtype=np.int16)
    out = pcm.astype(np.float32) * np.float32(1.0 / 32768.0)
    if planar:
        out = out.reshape(-1, channels).T.copy().reshape(-1)
    return out


def float_to_pcm16(x):
    x = np.asarray(x, dtype=np.float32)
    c = np.maximum(np.float32(-1.0), np.minimum(np.float32(1.0), x))
    c = np.where(np.isnan(x), np.float32(0.0), c)
    return np.trunc(c * np.float32(32767.0)).astype(np.int16)


def mix_to_mono(x, channels):
    x = np.asarray(x, dtype=np.float32)
    n = len(x) // channels
    x = x[: n * channels].reshape(n, channels)
    s = np.zeros(n, np.float32)
    for c in range(channels):          # float accumulation in channel order
        s = (s + x[:, c]).astype(np.float32)
    return (s / np.float32(channels)).astype(np.float32)


def interleave(planar, channels=2):
    p = np.asarray(planar, dtype=np.float32).reshape(channels, -1)
    return p.T.copy().reshape(-1)


def deinterleave(inter, channels=2):
    p = np.asarray(inter, dtype=np.float32).reshape(-1, channels)
    return p.T.copy().reshape(-1)


def resample_len(n_in, src, dst):
    return int(n_in * (float(dst) / float(src)))


def resample_linear(x, src, dst):
    x = np.asarray(x, dtype=np.float32)
    ratio = float(dst) / float(src)
    n = int(len(x) * ratio)
    pos = np.arange(n, dtype=np.float64) / ratio
    idx = pos.astype(np.int64)
    frac = pos - idx
    last = idx >= len(x) - 1
    i0 = np.minimum(idx, len(x) - 1)
    i1 = np.minimum(idx + 1, len(x) - 1)
    out = (((1.0 - frac) * x[i0].astype(np.float64)) + (frac * x[i1].astype(np.float64))).astype(np.float32)
    out[last] = x[-1]
    return out
