"""CPU suite: the audio pre/post oracle (oracle/audio_ref.py) against hand-worked known answers of the reference's helpers
(Core/Utils/AudioUtils.cs:13-36,45-61,90-101,172-186,204-219,329-354; Models/Dia.cs:918-923)."""
import numpy as np

from oracle import audio_ref as R


def test_pcm16_to_float_known_answers():
    pcm = np.array([0, 1, -1, 16384, -32768, 32767], np.int16)
    out = R.pcm16_to_float(pcm)
    assert out.dtype == np.float32
    np.testing.assert_array_equal(out, np.array([0.0, 2.0 ** -15, -(2.0 ** -15), 0.5, -1.0, 32767.0 / 32768.0], np.float32))
    # planar: L0 R0 L1 R1 -> L0 L1 R0 R1   (NAudioUtils.cs:94-104)
    st = np.array([100, -100, 200, -200], np.int16)
    np.testing.assert_array_equal(R.pcm16_to_float(st, 2, planar=True), np.array([100, 200, -100, -200], np.float32) / 32768.0)


def test_float_to_pcm16_truncates_and_clamps():
    x = np.array([0.0, 0.5, -0.5, 1.0, -1.0, 2.0, -3.0, 0.99999, 1e-5], np.float32)
    out = R.float_to_pcm16(x)
    # 0.5*32767 = 16383.5 -> 16383 (toward zero); -16383.5 -> -16383
    np.testing.assert_array_equal(out, np.array([0, 16383, -16383, 32767, -32767, 32767, -32767, 32766, 0], np.int16))


def test_mix_to_mono_order_and_division():
    x = np.array([1.0, 2.0, 4.0,  1e8, 1.0, -1e8], np.float32)
    out = R.mix_to_mono(x, 3)
    # frame 1: (1e8 + 1) rounds to 1e8 in float32, then - 1e8 = 0 -> 0/3
    np.testing.assert_array_equal(out, np.array([np.float32(7.0) / np.float32(3.0), 0.0], np.float32))


def test_interleave_round_trip():
    planar = np.arange(10, dtype=np.float32)          # L = 0..4, R = 5..9
    inter = R.interleave(planar, 2)
    np.testing.assert_array_equal(inter, np.array([0, 5, 1, 6, 2, 7, 3, 8, 4, 9], np.float32))
    np.testing.assert_array_equal(R.deinterleave(inter, 2), planar)


def test_resample_linear_known_answers():
    x = np.array([0.0, 1.0, 2.0, 3.0], np.float32)
    up = R.resample_linear(x, 1, 2)                   # positions 0, .5, 1, 1.5, 2, 2.5, 3(last), 3.5(last)
    np.testing.assert_array_equal(up, np.array([0, 0.5, 1, 1.5, 2, 2.5, 3, 3], np.float32))
    down = R.resample_linear(x, 2, 1)                 # positions 0, 2
    np.testing.assert_array_equal(down, np.array([0, 2], np.float32))
    assert R.resample_len(44100, 44100, 24000) == 24000
    assert R.resample_len(1000, 44100, 24000) == int(1000 * (24000 / 44100))
    # a non-dyadic ratio: every output is the binary64 blend rounded once to float32
    y = R.resample_linear(np.array([0.1, 0.7, -0.3], np.float32), 3, 4)
    ratio = 4 / 3
    exp = []
    xin = np.array([0.1, 0.7, -0.3], np.float32)
    for i in range(int(3 * ratio)):
        pos = i / ratio
        idx = int(pos)
        fr = pos - idx
        exp.append(xin[-1] if idx >= 2 else np.float32((1 - fr) * float(xin[idx]) + fr * float(xin[idx + 1])))
    np.testing.assert_array_equal(y, np.array(exp, np.float32))
