"""CPU suite: the C oracle's SNAC path against the golden vectors of the PyTorch-CPU restatement (oracle/torch_ref/snac.py)."""
import numpy as np
import pytest

from conftest import audit_snac_levels, load_golden, snac_cfg_from_meta
from neuralcodecs_amd.weights import save_blob, snac_noise, snac_synthetic_state_dict, synthetic_pcm
from oracle import c_oracle

PCM_TOL, LATENT_TOL, GAP_TOL = 1e-4, 3e-5, 1e-4


@pytest.mark.parametrize("name", ["snac_small", "snac_small_attn"])
def test_c_oracle_snac_small_matches_golden(name):
    g = load_golden(name)
    meta = g["meta"]
    cfg = snac_cfg_from_meta(meta)
    ref = c_oracle.RefSNAC(cfg, save_blob(snac_synthetic_state_dict(cfg, seed=meta["weight_seed"])))
    assert np.array_equal(synthetic_pcm(meta["B"], 1, meta["T"], cfg.sampling_rate, seed=meta["pcm_seed"]), g["pcm"])
    z, zq, codes = ref.encode(g["pcm"])
    Tz = g["z"].shape[-1]
    assert [c.shape for c in codes] == [(meta["B"], Tz // s) for s in cfg.vq_strides] and codes[0].dtype == np.int64
    assert np.abs(z - g["z"]).max() < LATENT_TOL
    assert audit_snac_levels(codes, g, GAP_TOL) == 0, "the golden fixture no longer matches: codes flipped (regenerate only with an audited reason)"
    assert np.abs(zq - g["zq"]).max() < LATENT_TOL
    gold_codes = [g[f"codes{i}"].astype(np.int64) for i in range(len(cfg.vq_strides))]
    noises = snac_noise(cfg, meta["B"], Tz, seed=meta["noise_seed"])
    audio = ref.decode(gold_codes, noises)
    assert audio.shape == g["audio"].shape
    assert np.abs(audio - g["audio"]).max() < PCM_TOL
    assert np.abs(ref.from_codes(gold_codes) - g["zq"]).max() < LATENT_TOL


def test_c_oracle_snac24k_full_size():
    """BASELINE config C1: SNAC 24 kHz mono, one 1 s clip -> padded 24576, T'=48, codes 12/24/48."""
    g = load_golden("snac24k_b1")
    meta = g["meta"]
    cfg = snac_cfg_from_meta(meta)
    ref = c_oracle.RefSNAC(cfg, save_blob(snac_synthetic_state_dict(cfg, seed=meta["weight_seed"])))
    pcm = synthetic_pcm(1, 1, meta["T"], cfg.sampling_rate, seed=meta["pcm_seed"])
    z, zq, codes = ref.encode(pcm)
    assert [c.shape for c in codes] == [(1, 12), (1, 24), (1, 48)]
    assert audit_snac_levels(codes, g, GAP_TOL) == 0, "the golden fixture no longer matches: codes flipped (regenerate only with an audited reason)"
    assert np.abs(zq[:, ::16, :] - g["zq_slice"]).max() < LATENT_TOL
    gold_codes = [g[f"codes{i}"].astype(np.int64) for i in range(3)]
    audio = ref.decode(gold_codes, snac_noise(cfg, 1, 48, seed=meta["noise_seed"]))
    assert audio.shape == (1, 1, 24576)
    assert np.abs(audio[:, :, ::17] - g["audio_slice"]).max() < PCM_TOL


def test_snac_pad_rule_and_noise_requirement():
    cfg = snac_cfg_from_meta(load_golden("snac_small_attn")["meta"])
    assert cfg.pad_multiple == cfg.hop_length * 8          # lcm(vq_strides[0]=4, window=8)
    ref = c_oracle.RefSNAC(cfg, save_blob(snac_synthetic_state_dict(cfg, seed=6)))
    assert c_oracle.lib().ref_snac_padded_length(ref._h, 2500) == 3072
    with pytest.raises(ValueError):
        ref.decode_latents(np.zeros((1, cfg.resolved_latent_dim, 8), np.float32))      # D8: noise must be injected


def test_c_oracle_snac44k_with_local_attention():
    """Full-width SNAC 44.1 kHz (54.5 M parameters, LocalMHA dim 1024 / 1536, stride-3 stage with output_padding) on a short
    clip: 20000 samples -> padded 24576 (= 2 * 12288), T' = 64 = two attention windows, codes 8/16/32/64."""
    g = load_golden("snac44k_short")
    meta = g["meta"]
    cfg = snac_cfg_from_meta(meta)
    ref = c_oracle.RefSNAC(cfg, save_blob(snac_synthetic_state_dict(cfg, seed=meta["weight_seed"])))
    pcm = synthetic_pcm(1, 1, meta["T"], cfg.sampling_rate, seed=meta["pcm_seed"])
    z, zq, codes = ref.encode(pcm)
    assert [c.shape for c in codes] == [(1, 8), (1, 16), (1, 32), (1, 64)]
    assert audit_snac_levels(codes, g, GAP_TOL) == 0, "the golden fixture no longer matches: codes flipped (regenerate only with an audited reason)"
    assert np.abs(zq[:, ::16, :] - g["zq_slice"]).max() < LATENT_TOL
    gold_codes = [g[f"codes{i}"].astype(np.int64) for i in range(4)]
    audio = ref.decode(gold_codes, snac_noise(cfg, 1, 64, seed=meta["noise_seed"]))
    assert np.abs(audio[:, :, ::17] - g["audio_slice"]).max() < PCM_TOL


def test_c_oracle_snac_encode_tensor_overload_as_written():
    """SNAC.Encode(Tensor) as written (Models/SNAC.cs:113-122, D7): the encoder runs on the UN-padded tensor.  Goldens come from
    oracle/torch_ref TorchSNAC.encode_tensor; the padded Encode(float[]) path gives a different frame count on the same input."""
    g = load_golden("snac_small_tensor")
    meta = g["meta"]
    cfg = snac_cfg_from_meta(meta)
    ref = c_oracle.RefSNAC(cfg, save_blob(snac_synthetic_state_dict(cfg, seed=meta["weight_seed"])))
    z, zq, codes = ref.encode_tensor(g["pcm"])
    assert z.shape == g["z"].shape == (2, 128, 32)
    assert np.abs(z - g["z"]).max() < LATENT_TOL
    assert audit_snac_levels(codes, g, GAP_TOL) == 0
    assert np.abs(zq - g["zq"]).max() < LATENT_TOL
    zp, _, cp = ref.encode(g["pcm"])                                     # Encode(float[]) pads: 36 frames, other codes
    assert zp.shape[-1] == 36 and cp[-1].shape[-1] == 36
    with pytest.raises(ValueError):                                      # 3001 samples -> 31 frames: the reference's quantizer throws
        ref.encode_tensor(synthetic_pcm(1, 1, 3001, cfg.sampling_rate, seed=1))
    # full-size 24 kHz model, 22628 samples -> 44 frames (the padded path gives 48)
    g = load_golden("snac24k_tensor_b1")
    meta = g["meta"]
    cfg = snac_cfg_from_meta(meta)
    ref = c_oracle.RefSNAC(cfg, save_blob(snac_synthetic_state_dict(cfg, seed=meta["weight_seed"])))
    pcm = synthetic_pcm(1, 1, meta["T"], cfg.sampling_rate, seed=meta["pcm_seed"])
    z, zq, codes = ref.encode_tensor(pcm)
    assert [c.shape for c in codes] == [(1, 11), (1, 22), (1, 44)]
    assert audit_snac_levels(codes, g, GAP_TOL) == 0
    assert np.abs(zq[:, ::16, :] - g["zq_slice"]).max() < LATENT_TOL
