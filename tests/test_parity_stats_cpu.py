"""CPU suite (build container: needs oracle/torch_ref): statistics behind "bit-exact codes".  tests/golden/parity_stats.json records, for
the FULL-SIZE bench configurations, C oracle vs the ATen restatement over many clips and two weight sets (tools/parity_stats.py):
>= 25 k codes per codec.  This test (a) holds the record to the claims DESIGN.md 2 makes -- every code flip is a near-tie of the ATen
argmin (top-2 distance gap < 1e-5), flips are rare (< 1 frame in 1000), latents / PCM of the unflipped clips within tolerance -- and
(b) recomputes one clip per codec and compares counts and the SHA-256 of the oracle's codes, so the record cannot drift from the oracle."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
REC = os.path.join(ROOT, "tests", "golden", "parity_stats.json")
GAP_TOL, PCM_TOL, LATENT_TOL = 1e-5, 1e-4, 5e-5


def _rec():
    return json.load(open(REC))


@pytest.mark.parametrize("codec,min_clips", [("dac44k", 32), ("encodec48k", 16), ("snac44k", 8)])
def test_recorded_statistics_hold_the_claims(codec, min_clips):
    s = _rec()[codec]["summary"]
    assert s["clips"] >= min_clips and len(s["weight_seeds"]) >= 2 and s["n_codes"] >= 25000
    assert s["flipped_frames"] <= 1e-3 * s["n_frames"], "code flips are not rare"
    assert s["max_flip_gap"] < GAP_TOL, "a flipped code is not a near-tie of the ATen argmin"
    assert s["pcm_max_abs"] <= PCM_TOL and s["latents_max_abs"] <= LATENT_TOL
    for c in _rec()[codec]["clips"]:
        assert all(g < GAP_TOL for g in c["flip_gaps"])


def test_dac_has_no_flip_at_all():
    assert _rec()["dac44k"]["summary"]["flipped_frames"] == 0


@pytest.mark.parametrize("codec", ["dac44k", "encodec48k", "snac44k"])
def test_one_clip_recomputed_matches_the_record(codec):
    pytest.importorskip("torch")
    import parity_stats as ps
    want = _rec()[codec]["clips"][0]
    fn = {"dac44k": ps.dac_clip, "encodec48k": ps.encodec_clip, "snac44k": ps.snac_clip}[codec]
    got = fn(want["weight_seed"], want["pcm_seed"], {})
    assert got["codes_sha256"] == want["codes_sha256"], "the C oracle's codes changed: regenerate tests/golden/parity_stats.json (tools/parity_stats.py) and re-audit"
    assert got["n_codes"] == want["n_codes"] and got["flipped_frames"] == want["flipped_frames"]
    if "pcm_max_abs" in want:
        assert got["pcm_max_abs"] <= PCM_TOL


# round 5: the reference's OTHER presets at full width -- DAC 44 kHz-16 kbps / 24 kHz / 16 kHz (Config/DAC/DACConfig.cs:103-135),
# SNAC 32 / 24 kHz, Encodec 24 kHz -- four clips each, two weight sets (tools/parity_stats.py --codec presets)
OTHER_PRESETS = ["dac44k_16kbps", "dac24k", "dac16k", "snac32k", "snac24k", "encodec24k"]


@pytest.mark.parametrize("codec", OTHER_PRESETS)
def test_other_presets_recorded_statistics(codec):
    s = _rec()[codec]["summary"]
    assert s["clips"] >= 4 and len(s["weight_seeds"]) >= 2 and s["n_codes"] >= 600
    assert s["flipped_frames"] <= max(1, 1e-3 * s["n_frames"]) and s["max_flip_gap"] < GAP_TOL
    assert s["pcm_max_abs"] <= PCM_TOL and s["latents_max_abs"] <= LATENT_TOL


@pytest.mark.parametrize("codec", ["dac24k", "dac44k_16kbps"])
def test_one_clip_of_another_preset_recomputed_matches_the_record(codec):
    pytest.importorskip("torch")
    import parity_stats as ps
    want = _rec()[codec]["clips"][0]
    got = ps.dac_clip(want["weight_seed"], want["pcm_seed"], {}, preset={"dac24k": "dac_24khz", "dac44k_16kbps": "dac_44khz_16kbps"}[codec])
    assert got["codes_sha256"] == want["codes_sha256"] and got["n_codes"] == want["n_codes"] and got["flipped_frames"] == want["flipped_frames"]
