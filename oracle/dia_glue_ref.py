"""TEST INFRASTRUCTURE (oracle): restatement of the Dia <-> DAC glue of the reference, statement by statement, over any DAC object
that offers encode / from_codes / decode on numpy arrays (the C oracle's RefDAC in the tests).

  Models/Dia.cs:973-981    Decode(audioCodes[T, C]):  FromCodes(audioCodes.unsqueeze(0).transpose(1, 2)) -> Decode -> squeeze
  Models/Dia.cs:989-1002   Encode(audio[C, T]):       audio.unsqueeze(0) -> Encode(sampleRate) -> codes.squeeze(0).transpose(0, 1)
  Modules/Dia/AudioUtils.cs:189-199  Decode(model, audioCodes[1, C, T]): exactly one frame, else ArgumentException
"""
import numpy as np


def dia_decode(dac, audio_codes):
    """[T, C] int codes -> waveform [T*hop] (Dia.cs:973-981)."""
    codes = np.asarray(audio_codes)
    if codes.ndim != 2:
        raise ValueError("audio codes must be [T, C]")
    batched = np.transpose(codes[None, :, :], (0, 2, 1))                  # unsqueeze(0).transpose(1, 2) -> [1, C, T]
    audio_values = dac.from_codes(np.ascontiguousarray(batched, np.int64))
    return np.squeeze(dac.decode(audio_values))                           # squeeze_()


def dia_encode(dac, audio, sample_rate=None):
    """[C=1, T] audio -> [T', n_q] codes (Dia.cs:989-1002)."""
    a = np.asarray(audio, np.float32)
    if a.ndim != 2:
        raise ValueError("audio must be [C, T]")
    out = dac.encode(a[None, :, :]) if sample_rate is None else dac.encode(a[None, :, :], sample_rate=sample_rate)
    encoded_frame = out[1]                                                # (z, codes, latents, ...)
    return np.transpose(encoded_frame[0], (1, 0))                         # squeeze(0).transpose(0, 1)


def audio_utils_decode(dac, audio_codes):
    """AudioUtils.Decode (AudioUtils.cs:189-199): [1, C, T] -> FromCodes -> Decode; one frame only."""
    codes = np.asarray(audio_codes)
    if codes.shape[0] != 1:
        raise ValueError(f"Expected one frame, got {codes.shape[0]}")
    return dac.decode(dac.from_codes(np.ascontiguousarray(codes, np.int64)))
