"""CPU suite: non-finite inputs.  The reference hands its distance matrix to ATen's argmin
(Modules/DAC/VectorQuantizer.cs:121, Modules/SNAC/VectorQuantizer.cs:137, Modules/Encodec/EuclideanCodebook.cs:181), which treats NaN as
smaller than every number and returns the FIRST NaN of a row; the goldens (tools/make_golden.py --round6, ATen on clips holding an inf / a
NaN sample, and on latents with infinite components that make rows PARTLY NaN) pin that, and the C oracle must reproduce them."""
import numpy as np
import pytest

from conftest import dac_cfg_from_meta, encodec_cfg_from_meta, load_golden, snac_cfg_from_meta
from neuralcodecs_amd.weights import (dac_synthetic_state_dict, encodec_synthetic_state_dict, save_blob, snac_noise,
                                      snac_synthetic_state_dict)
from oracle import c_oracle

PCM_TOL, LATENT_TOL = 1e-4, 5e-5


def close_nonfinite(a, b, tol):
    """Same NaN mask, same infinities, finite values within tol."""
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape or not np.array_equal(np.isnan(a), np.isnan(b)):
        return False
    inf = np.isinf(a) | np.isinf(b)
    if not np.array_equal(a[inf], b[inf]):
        return False
    fin = np.isfinite(a) & np.isfinite(b)
    return bool(np.all(np.abs(a[fin] - b[fin]) <= tol))


@pytest.mark.parametrize("kind", ["dac", "snac", "encodec"])
def test_vq_stage_partly_nan_rows_first_nan_wins(kind):
    g = load_golden("vq_nonfinite")
    assert int(g[f"{kind}_partly_nan_rows"]) >= 20
    idx = c_oracle.vq_argmin(g[f"{kind}_ze"], g[f"{kind}_cb"])[0]
    assert np.array_equal(idx, g[f"{kind}_idx"].astype(np.int64))


def test_dac_nonfinite_clips_match_aten():
    g = load_golden("dac_small_nonfinite")
    cfg = dac_cfg_from_meta(g["meta"])
    ref = c_oracle.RefDAC(cfg, save_blob(dac_synthetic_state_dict(cfg, seed=g["meta"]["weight_seed"])))
    assert np.isinf(g["pcm"]).sum() == 2 and np.isnan(g["pcm"]).sum() == 2
    zq, codes, _, _ = ref.encode(g["pcm"])
    assert np.array_equal(codes, g["codes"])
    assert close_nonfinite(zq, g["zq"], LATENT_TOL)
    nan_frames = np.isnan(g["zq"]).any(1)
    assert 0 < nan_frames[0].sum() < nan_frames.shape[1] and not nan_frames[3].any()      # poisoned receptive field only; clip 3 is clean
    assert np.all(g["codes"].transpose(0, 2, 1)[nan_frames] == 0)                          # an all-NaN row: index 0
    assert close_nonfinite(ref.decode(g["zq"]), g["audio"], PCM_TOL)


def test_snac_nonfinite_clips_match_aten():
    g = load_golden("snac_small_nonfinite")
    cfg = snac_cfg_from_meta(g["meta"])
    ref = c_oracle.RefSNAC(cfg, save_blob(snac_synthetic_state_dict(cfg, seed=g["meta"]["weight_seed"])))
    _, _, codes = ref.encode(g["pcm"])
    for i, c in enumerate(codes):
        assert np.array_equal(c, g[f"codes{i}"])
    nz = snac_noise(cfg, g["meta"]["B"], codes[-1].shape[-1], seed=g["meta"]["noise_seed"])
    assert close_nonfinite(ref.decode([c.astype(np.int64) for c in codes], nz), g["audio"], PCM_TOL)


@pytest.mark.parametrize("name", ["encodec_small48_nonfinite", "encodec_small24_nonfinite"])
def test_encodec_nonfinite_clips_match_aten(name):
    g = load_golden(name)
    cfg = encodec_cfg_from_meta(g["meta"])
    ref = c_oracle.RefEncodec(cfg, save_blob(encodec_synthetic_state_dict(cfg, seed=g["meta"]["weight_seed"])))
    frames = ref.encode(g["pcm"])
    assert len(frames) == g["meta"]["n_frames"]
    for i, fr in enumerate(frames):
        assert np.array_equal(fr[0], g[f"codes{i}"])
        if cfg.normalize:
            assert close_nonfinite(fr[1], g[f"scale{i}"], 1e-6)
    gold = [(g[f"codes{i}"].astype(np.int64), g.get(f"scale{i}")) for i in range(len(frames))]
    assert close_nonfinite(ref.decode(gold), g["audio"], PCM_TOL)
