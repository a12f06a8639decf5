"""GPU suite: non-finite inputs through the C ABI.  The quantizers' argmin is ATen's (Modules/DAC/VectorQuantizer.cs:121,
Modules/SNAC/VectorQuantizer.cs:137, Modules/Encodec/EuclideanCodebook.cs:181): a NaN distance wins and the FIRST NaN index is returned.  The
goldens are ATen's own results (tools/make_golden.py --round6); the engine must also stay bit-equal to the C oracle (NaNs compared as equal)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from conftest import dac_cfg_from_meta, encodec_cfg_from_meta, load_golden, snac_cfg_from_meta  # noqa: E402
from neuralcodecs_amd import DAC, SNAC, Encodec, ops  # noqa: E402
from neuralcodecs_amd.weights import (dac_synthetic_state_dict, encodec_synthetic_state_dict, save_blob, snac_noise,  # noqa: E402
                                      snac_synthetic_state_dict)
from oracle import c_oracle  # noqa: E402
from test_nonfinite_cpu import close_nonfinite  # noqa: E402

PCM_TOL, LATENT_TOL = 1e-4, 5e-5


def same(a, b):
    return np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("kind", ["dac", "snac"])
def test_vq_argmin_kernel_partly_nan_rows(kind):
    g = load_golden("vq_nonfinite")
    idx, st = ops.vq_argmin(g[f"{kind}_ze"], g[f"{kind}_cb"])
    assert np.array_equal(idx, g[f"{kind}_idx"].astype(np.int64))
    ridx, rst = c_oracle.vq_argmin(g[f"{kind}_ze"], g[f"{kind}_cb"])[:2]
    assert np.array_equal(idx, ridx) and same(st, rst)


@pytest.mark.parametrize("form", [0, 1])
def test_euclid_rvq_kernels_partly_nan_rows(form):
    """Both Encodec RVQ kernels (per-stage scan, all-stages matrix-core form) on latents with infinite components: stage 0 must return ATen's
    first-NaN index; the later stages see a NaN residual (all-NaN rows -> 0) exactly where stage 0 subtracted from an infinity."""
    g = load_golden("vq_nonfinite")
    ze, cb = g["encodec_ze"], g["encodec_cb"]
    books = np.stack([cb, cb[::-1].copy()], 0)
    codes, res = ops.euclid_rvq(ze, books, form=form)
    assert np.array_equal(codes[:, 0, :], g["encodec_idx"].astype(np.int64))
    ridx, rst = c_oracle.vq_argmin(ze, cb)[:2]
    r1 = ze - cb[ridx].transpose(0, 2, 1)                                   # residual - embed[idx]
    ridx2 = c_oracle.vq_argmin(r1, books[1])[0]
    assert np.array_equal(codes[:, 1, :], ridx2)
    if form == 0:                                                            # (the all-stages kernel keeps the residual in LDS: codes only)
        assert same(res, r1 - books[1][ridx2].transpose(0, 2, 1))


def test_dac_nonfinite_clips():
    g = load_golden("dac_small_nonfinite")
    cfg = dac_cfg_from_meta(g["meta"])
    blob = save_blob(dac_synthetic_state_dict(cfg, seed=g["meta"]["weight_seed"]))
    ref = c_oracle.RefDAC(cfg, blob)
    with DAC(cfg) as m:
        m.load_blob(blob)
        z, codes, lat, _, _ = m.encode(g["pcm"])
        audio = m.decode(z)
        audio_g = m.decode(g["zq"])
    assert np.array_equal(codes, g["codes"])                                  # ATen
    assert close_nonfinite(z, g["zq"], LATENT_TOL) and close_nonfinite(audio_g, g["audio"], PCM_TOL)
    rz, rcodes, rlat, _ = ref.encode(g["pcm"])                                # C oracle: bit for bit, NaN == NaN
    assert np.array_equal(codes, rcodes) and same(z, rz) and same(lat, rlat) and same(audio, ref.decode(rz))


def test_snac_nonfinite_clips():
    g = load_golden("snac_small_nonfinite")
    cfg = snac_cfg_from_meta(g["meta"])
    blob = save_blob(snac_synthetic_state_dict(cfg, seed=g["meta"]["weight_seed"]))
    ref = c_oracle.RefSNAC(cfg, blob)
    with SNAC(cfg) as m:
        m.load_blob(blob)
        codes = m.encode(g["pcm"])
        nz = snac_noise(cfg, g["meta"]["B"], codes[-1].shape[-1], seed=g["meta"]["noise_seed"])
        audio = m.decode(codes, nz)
    _, _, rcodes = ref.encode(g["pcm"])
    for i, (c, rc) in enumerate(zip(codes, rcodes)):
        assert np.array_equal(c, g[f"codes{i}"]) and np.array_equal(c, rc)
    assert close_nonfinite(audio, g["audio"], PCM_TOL) and same(audio, ref.decode(rcodes, nz))


@pytest.mark.parametrize("name", ["encodec_small48_nonfinite", "encodec_small24_nonfinite"])
def test_encodec_nonfinite_clips(name):
    g = load_golden(name)
    cfg = encodec_cfg_from_meta(g["meta"])
    blob = save_blob(encodec_synthetic_state_dict(cfg, seed=g["meta"]["weight_seed"]))
    ref = c_oracle.RefEncodec(cfg, blob)
    T = g["meta"]["T"]
    with Encodec(cfg) as m:
        m.load_blob(blob)
        frames = m.encode(g["pcm"])
        audio = m.decode(frames, T)
    rfr = ref.encode(g["pcm"])
    assert len(frames) == g["meta"]["n_frames"]
    for i, (f, r) in enumerate(zip(frames, rfr)):
        assert np.array_equal(f.codes, g[f"codes{i}"]) and np.array_equal(f.codes, r[0])
        if cfg.normalize:
            assert close_nonfinite(f.scale, g[f"scale{i}"], 1e-6) and same(f.scale, r[1])
    assert close_nonfinite(audio[..., :g["audio"].shape[-1]], g["audio"], PCM_TOL)
    assert same(audio, ref.decode(rfr))


def test_encodec48k_full_width_nonfinite_segments():
    """The full-width 48 kHz model (the round-6 streaming kernels: fused first pass, strided two-input down / up convolutions, one-launch RMS
    scale) on clips whose non-finite sample poisons ONE segment: that segment's GroupNorm statistics, scale and every value behind them turn
    NaN / inf exactly as in the C oracle (bit for bit, NaN == NaN), its codes are ATen's all-NaN-row answer 0, and the other segments of the
    same clip stay finite and exact."""
    from neuralcodecs_amd.weights import synthetic_pcm
    g = load_golden("encodec48k_b1")
    cfg = encodec_cfg_from_meta(g["meta"])
    blob = save_blob(encodec_synthetic_state_dict(cfg, seed=42))
    ref = c_oracle.RefEncodec(cfg, blob)
    T = 96000
    pcm = synthetic_pcm(2, cfg.channels, T, cfg.sampling_rate, seed=31)
    pcm[0, 0, 1000] = np.inf                      # segment 0 of clip 0 only (segments start at 0, 47520, 95040)
    pcm[1, 1, 60000] = np.nan                     # segment 1 of clip 1 only
    with Encodec(cfg) as m:
        m.load_blob(blob)
        frames = m.encode(pcm)
        audio = m.decode(frames, T)
    rfr = ref.encode(pcm)
    assert len(frames) == len(rfr) == 3
    for f, r in zip(frames, rfr):
        assert np.array_equal(f.codes, r[0]) and same(f.scale, r[1])
    assert same(audio, ref.decode(rfr))
    assert np.all(frames[0].codes[0] == 0) and np.all(frames[1].codes[1] == 0)            # poisoned segments: every row NaN -> index 0
    assert frames[0].codes[1].max() > 0 and frames[1].codes[0].max() > 0 and frames[2].codes.max() > 0   # the others are ordinary
    assert np.isinf(frames[0].scale[0]).all() and np.isnan(frames[1].scale[1]).all() and np.isfinite(frames[2].scale).all()
