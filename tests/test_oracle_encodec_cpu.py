"""CPU suite: the C oracle's Encodec path against the golden vectors of the PyTorch-CPU restatement (oracle/torch_ref/encodec.py)."""
import numpy as np
import pytest

from conftest import audit_code_mismatches, encodec_cfg_from_meta, load_golden
from neuralcodecs_amd.weights import encodec_synthetic_state_dict, save_blob, synthetic_pcm
from oracle import c_oracle

PCM_TOL, LATENT_TOL, GAP_TOL = 1e-4, 5e-5, 1e-4


def _ref(name):
    g = load_golden(name)
    cfg = encodec_cfg_from_meta(g["meta"])
    return g, cfg, c_oracle.RefEncodec(cfg, save_blob(encodec_synthetic_state_dict(cfg, seed=g["meta"]["weight_seed"])))


def _gold_frames(g):
    n = g["meta"]["n_frames"]
    return [(g[f"codes{i}"].astype(np.int64), g.get(f"scale{i}")) for i in range(n)]


@pytest.mark.parametrize("name", ["encodec_small48", "encodec_small24"])
def test_c_oracle_encodec_small_matches_golden(name):
    g, cfg, ref = _ref(name)
    meta = g["meta"]
    assert np.array_equal(synthetic_pcm(meta["B"], cfg.channels, meta["T"], cfg.sampling_rate, seed=meta["pcm_seed"]), g["pcm"])
    frames = ref.encode(g["pcm"], want_emb=True)
    assert len(frames) == meta["n_frames"] and ref.n_q() == meta["n_q"]
    for i, (codes, scale, emb) in enumerate(frames):
        assert codes.shape == g[f"codes{i}"].shape and codes.dtype == np.int64
        assert np.abs(emb - g[f"emb{i}"]).max() < LATENT_TOL
        assert audit_code_mismatches(codes, g[f"codes{i}"], g[f"gap{i}"], GAP_TOL) == 0   # zero flips against the torch restatement (a flip would also have to be a near-tie)
        if cfg.normalize:
            assert np.abs(scale - g[f"scale{i}"]).max() < 1e-6
    audio = ref.decode(_gold_frames(g))
    assert audio.shape == g["audio"].shape
    assert np.abs(audio - g["audio"]).max() < PCM_TOL


def test_segments_and_small_input_reflect_path():
    """48 kHz-style segmentation: stride = 99 % of the segment; the short tail hits the small-input reflect path (D9):
    SConv1d zero-pads and never trims, so the tail emits one frame more than ceil(len/hop)."""
    g, cfg, ref = _ref("encodec_small48")
    assert (ref.segment_length, ref.segment_stride) == (4000, 3960)
    assert [g[f"codes{i}"].shape[-1] for i in range(3)] == [84, 84, 4]           # tail: 180 samples / hop 48 -> 4 frames (not 3.75 -> 4)
    lib = c_oracle.lib()
    assert lib.ref_encodec_frames(ref._h, 4000) == 84 and lib.ref_encodec_frames(ref._h, 180) == 4
    assert lib.ref_encodec_frames(ref._h, 100) == 4                                # 100 samples -> 3 frames before the last conv -> 4 (D9)
    assert g["audio"].shape[-1] == 2 * 3960 + lib.ref_encodec_decoded_length(ref._h, 4)


@pytest.mark.parametrize("name,slice_step", [("encodec48k_b1", 23), ("encodec24k_b1", 23)])
def test_c_oracle_encodec_full_size(name, slice_step):
    """BASELINE config C3 model (48 kHz stereo 12 kbps, 2 s -> segments 48000/48000/960 -> 150/150/4 frames, 8 codebooks)
    and the 24 kHz causal weight-norm model."""
    g, cfg, ref = _ref(name)
    meta = g["meta"]
    pcm = synthetic_pcm(1, cfg.channels, meta["T"], cfg.sampling_rate, seed=meta["pcm_seed"])
    frames = ref.encode(pcm, want_emb=True)
    for i, (codes, scale, emb) in enumerate(frames):
        assert codes.shape == g[f"codes{i}"].shape
        assert np.abs(emb[:, ::8, :] - g[f"emb{i}"]).max() < LATENT_TOL
        assert audit_code_mismatches(codes, g[f"codes{i}"], g[f"gap{i}"], GAP_TOL) == 0   # zero flips against the torch restatement (a flip would also have to be a near-tie)
    audio = ref.decode(_gold_frames(g))
    assert np.abs(audio[:, :, ::slice_step] - g["audio_slice"]).max() < PCM_TOL
