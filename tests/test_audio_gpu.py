"""GPU suite: the audio pre/post kernels (csrc/nc_audio.hip through the C ABI) against the oracle, bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from neuralcodecs_amd import audio  # noqa: E402
from oracle import audio_ref as R  # noqa: E402


def test_pcm16_round_trip_and_layouts():
    rng = np.random.default_rng(0)
    pcm = rng.integers(-32768, 32768, size=2 * 50001, dtype=np.int16)
    np.testing.assert_array_equal(audio.pcm16_to_float(pcm), R.pcm16_to_float(pcm))
    np.testing.assert_array_equal(audio.pcm16_to_float(pcm, 2, planar=True), R.pcm16_to_float(pcm, 2, planar=True))
    x = (rng.standard_normal(100003) * 0.7).astype(np.float32)
    x[:4] = [1.0, -1.0, 7.0, -9.0]
    np.testing.assert_array_equal(audio.float_to_pcm16(x), R.float_to_pcm16(x))
    # every int16 except -32768 survives float -> pcm16 of (v/32768 * 32768/32767)... use the exact inverse scale instead
    allv = np.arange(-32767, 32768, dtype=np.int16)
    f = (allv.astype(np.float32) / np.float32(32767.0))
    back = audio.float_to_pcm16(f)
    assert np.abs(back.astype(np.int32) - allv.astype(np.int32)).max() <= 1


@pytest.mark.parametrize("channels", [1, 2, 3, 6])
def test_mix_to_mono(channels):
    rng = np.random.default_rng(channels)
    x = rng.standard_normal(channels * 44101).astype(np.float32)
    np.testing.assert_array_equal(audio.convert_to_mono(x, channels), R.mix_to_mono(x, channels))


@pytest.mark.parametrize("channels", [2, 5])
def test_interleave_deinterleave(channels):
    rng = np.random.default_rng(7)
    x = rng.standard_normal(channels * 12345).astype(np.float32)
    inter = audio.deinterleave_to_interleave(x, channels)
    np.testing.assert_array_equal(inter, R.interleave(x, channels))
    np.testing.assert_array_equal(audio.interleave_to_deinterleave(inter, channels), x)


@pytest.mark.parametrize("src,dst", [(44100, 24000), (24000, 44100), (48000, 44100), (16000, 16000), (8000, 44100)])
def test_resample_linear(src, dst):
    rng = np.random.default_rng(src + dst)
    x = rng.standard_normal((3, src // 4 + 17)).astype(np.float32)
    y = audio.resample_linear(x, src, dst)
    assert y.shape == (3, R.resample_len(x.shape[1], src, dst))
    for b in range(3):
        np.testing.assert_array_equal(y[b], R.resample_linear(x[b], src, dst))
    np.testing.assert_array_equal(audio.resample_linear(x[0], src, dst), R.resample_linear(x[0], src, dst))


def test_device_tensors_stay_on_device():
    import torch
    x = torch.randn(2 * 4096, device="cuda")
    m = audio.convert_to_mono(x, 2)
    assert m.is_cuda and m.shape == (4096,)
    np.testing.assert_array_equal(m.cpu().numpy(), R.mix_to_mono(x.cpu().numpy(), 2))
    p = audio.float_to_pcm16(m)
    assert p.is_cuda and p.dtype == torch.int16


def test_argument_errors():
    with pytest.raises(ValueError):
        audio.pcm16_to_float(np.zeros(3, np.int16), channels=2)
    with pytest.raises(ValueError):
        audio.resample_linear(np.zeros(1, np.float32), 44100, 8000)
