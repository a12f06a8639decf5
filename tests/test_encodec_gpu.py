"""GPU suite: Encodec Encode / Decode through the C ABI against the C oracle (bit for bit) and the golden vectors."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from conftest import audit_code_mismatches, encodec_cfg_from_meta, load_golden  # noqa: E402
from neuralcodecs_amd import EncodedFrame, Encodec  # noqa: E402
from neuralcodecs_amd.config import EncodecConfig  # noqa: E402
from neuralcodecs_amd.weights import encodec_synthetic_state_dict, save_blob, synthetic_pcm  # noqa: E402
from oracle import c_oracle  # noqa: E402

PCM_TOL, LATENT_TOL, GAP_TOL = 1e-4, 5e-5, 1e-4


def _setup(name):
    g = load_golden(name)
    cfg = encodec_cfg_from_meta(g["meta"])
    blob = save_blob(encodec_synthetic_state_dict(cfg, seed=g["meta"]["weight_seed"]))
    m = Encodec(cfg)
    m.load_blob(blob)
    return g, cfg, m, c_oracle.RefEncodec(cfg, blob)


def _check_vs_oracle(m, ref, pcm):
    frames, embs = m.encode(pcm, return_emb=True)
    rframes = ref.encode(pcm, want_emb=True)
    assert len(frames) == len(rframes)
    for f, e, (rc, rs, re) in zip(frames, embs, rframes):
        assert f.codes.dtype == np.int64 and np.array_equal(e, re)
        assert np.array_equal(f.codes, rc)                                       # bit-exact codes vs the C oracle
        if rs is not None:
            assert np.array_equal(f.scale, rs)
    audio = m.decode(frames, pcm.shape[-1])
    assert np.array_equal(audio, ref.decode([(rc, rs) for rc, rs, _ in rframes]))
    return frames, embs, audio


@pytest.mark.parametrize("name", ["encodec_small48", "encodec_small24"])
def test_encodec_small_vs_golden_and_oracle(name):
    g, cfg, m, ref = _setup(name)
    frames, embs, audio = _check_vs_oracle(m, ref, g["pcm"])
    for i, (f, e) in enumerate(zip(frames, embs)):
        assert np.abs(e - g[f"emb{i}"]).max() < LATENT_TOL
        assert audit_code_mismatches(f.codes, g[f"codes{i}"], g[f"gap{i}"], GAP_TOL) == 0   # zero flips against the torch restatement (a flip would also have to be a near-tie)
    gold = [EncodedFrame(g[f"codes{i}"].astype(np.int64), g.get(f"scale{i}")) for i in range(g["meta"]["n_frames"])]
    ga = m.decode(gold, g["pcm"].shape[-1])
    assert ga.shape == g["audio"].shape and np.abs(ga - g["audio"]).max() < PCM_TOL
    out = m.forward(g["pcm"])
    assert out.shape == g["pcm"].shape and np.array_equal(out, audio[..., : g["pcm"].shape[-1]])
    m.dispose()


def test_encodec48k_config_c3_shape():
    """BASELINE config C3 model: 48 kHz stereo 12 kbps, 2 s clips -> segments 48000/48000/960 -> 150/150/4 frames x 8 codebooks."""
    g, cfg, m, ref = _setup("encodec48k_b1")
    meta = g["meta"]
    assert (m.frame_rate, m.bits_per_codebook, m.num_codebooks, m.segment_length, m.segment_stride) == (150, 10, 16, 48000, 47520)
    pcm = synthetic_pcm(2, 2, meta["T"], cfg.sampling_rate, seed=meta["pcm_seed"])
    frames, embs, audio = _check_vs_oracle(m, ref, pcm)
    assert [f.codes.shape for f in frames] == [(2, 8, 150), (2, 8, 150), (2, 8, 4)] and audio.shape == (2, 2, 96320)
    for i, (f, e) in enumerate(zip(frames, embs)):
        assert np.abs(e[:1, ::8, :] - g[f"emb{i}"]).max() < LATENT_TOL
        assert audit_code_mismatches(f.codes[:1], g[f"codes{i}"], g[f"gap{i}"], GAP_TOL) == 0   # zero flips against the torch restatement (a flip would also have to be a near-tie)
    gold = [EncodedFrame(g[f"codes{i}"].astype(np.int64), g[f"scale{i}"]) for i in range(3)]
    assert np.abs(m.decode(gold, meta["T"])[:, :, ::23] - g["audio_slice"]).max() < PCM_TOL
    # bandwidth switch: 6 kbps -> 4 codebooks (SetTargetBandwidth, Encodec.cs:409-419)
    m.set_target_bandwidth(6.0)
    ref.bandwidth = 6.0
    f6 = m.encode(pcm)
    assert f6[0].codes.shape == (2, 4, 150) and np.array_equal(f6[0].codes, frames[0].codes[:, :4])
    assert np.array_equal(m.decode(f6, meta["T"]), ref.decode([(f.codes, f.scale) for f in f6]))
    with pytest.raises(ValueError):
        m.set_target_bandwidth(7.0)
    m.dispose()


def test_encodec24k_causal_weight_norm():
    g, cfg, m, ref = _setup("encodec24k_b1")
    meta = g["meta"]
    pcm = synthetic_pcm(2, 1, meta["T"], cfg.sampling_rate, seed=meta["pcm_seed"])
    frames, embs, audio = _check_vs_oracle(m, ref, pcm)
    assert len(frames) == 1 and frames[0].codes.shape == (2, 8, 75) and frames[0].scale is None and audio.shape == (2, 1, 24000)
    assert np.abs(embs[0][:1, ::8, :] - g["emb0"]).max() < LATENT_TOL
    assert audit_code_mismatches(frames[0].codes[:1], g["codes0"], g["gap0"], GAP_TOL) == 0   # zero flips against the torch restatement (a flip would also have to be a near-tie)
    gold = [EncodedFrame(g["codes0"].astype(np.int64), None)]
    assert np.abs(m.decode(gold)[:, :, ::23] - g["audio_slice"]).max() < PCM_TOL
    m.dispose()


def test_encodec_errors_ragged_and_device_api():
    import torch
    g, cfg, m, ref = _setup("encodec_small48")
    with pytest.raises(ValueError, match="3D"):
        m.encode(np.zeros((2, 100), np.float32))                                  # ArgumentException, Encodec.cs:493-497
    with pytest.raises(ValueError, match="channels"):
        m.encode(np.zeros((1, 1, 100), np.float32))                               # Encodec.cs:499-503
    with pytest.raises(ValueError):
        m.decode([])                                                              # "No frames provided to decode"
    with pytest.raises(ValueError):
        Encodec(EncodecConfig(bandwidth=5.0))                                     # invalid bandwidth, Encodec.cs:49-54
    fresh = Encodec(cfg)
    with pytest.raises(RuntimeError):
        fresh.encode(g["pcm"])
    fresh.dispose()
    with pytest.raises(ValueError, match="too short"):
        m.encode(np.zeros((1, 2, 1), np.float32))                                 # a 1-sample clip degenerates inside a residual block
    for T in (47, 48, 100, 3960, 3990, 4000, 4047, 9000):                         # ragged: short clips, exact stride / segment multiples
        pcm = synthetic_pcm(1, 2, T, cfg.sampling_rate, seed=T)
        _check_vs_oracle(m, ref, pcm)
    # Decode(List<EncodedFrame>) takes no clip length in the reference (Encodec.cs:213-235): the frames alone fix the output.
    # nc_encodec_clip_length gives the T of that layout; decoding with it equals decoding with the encoded clip's own length.
    for T in (48, 100, 3960, 3990, 4047, 9000):
        pcm = synthetic_pcm(1, 2, T, cfg.sampling_rate, seed=T + 1)
        fr = m.encode(pcm)
        Ti = m._infer_length(fr)
        assert Ti <= T and m.query(Ti)[0] == len(fr) and m.query(Ti)[2] == [f.codes.shape[-1] for f in fr]
        assert np.array_equal(m.decode(fr), m.decode(fr, T))
    with pytest.raises(ValueError):
        m._infer_length([EncodedFrame(np.zeros((1, 2, 10 ** 6), np.int64), None)])   # no clip yields that layout
    pcm = synthetic_pcm(3, 2, 5000, cfg.sampling_rate, seed=1)
    frames = m.encode(pcm)
    one = m.encode(pcm[2:3])
    assert all(np.array_equal(a.codes[2:3], b.codes) for a, b in zip(frames, one))      # batch invariance
    dframes = m.encode(torch.from_numpy(pcm).cuda())
    da = m.decode(dframes, 5000)
    torch.cuda.synchronize()
    assert all(np.array_equal(a.codes.cpu().numpy(), b.codes) for a, b in zip(dframes, frames))
    assert np.array_equal(da.cpu().numpy(), m.decode(frames, 5000))
    m.dispose()


@pytest.mark.parametrize("preset,bw", [("encodec_48khz", 12.0), ("encodec_24khz", 24.0)])
def test_encodec_tied_codebooks_first_index_full_width(preset, bw):
    """Both presets (Config/Encodec/EncodecConfig.cs) at full width with adversarial codebooks: every embedding row twice (an exact tie in every
    frame of every stage; the matrix-core distance kernel scans 1024 codes per frame) and dead codes.  EuclideanCodebook.Quantize's argmin
    (EuclideanCodebook.cs:181, ATen) returns the first index: codes in the lower half only, engine == C oracle bit for bit.  24 kHz at
    24 kbps runs all 32 stages."""
    from neuralcodecs_amd.config import EncodecConfig
    from neuralcodecs_amd.weights import tie_codebooks
    import dataclasses
    cfg = dataclasses.replace(getattr(EncodecConfig, preset)(), bandwidth=bw)
    blob = save_blob(tie_codebooks(encodec_synthetic_state_dict(cfg, seed=42)))
    pcm = synthetic_pcm(2, cfg.channels, 2 * cfg.sampling_rate, cfg.sampling_rate, seed=19)
    with Encodec(cfg) as m:
        m.load_blob(blob)
        frames, _, _ = _check_vs_oracle(m, c_oracle.RefEncodec(cfg, blob), pcm)
    for f in frames:
        assert f.codes.max() < cfg.codebook_size // 2 and not np.any(f.codes % 7 == 0), "a tie was not resolved to the first index"


@pytest.mark.parametrize("tail", [482, 500, 966, 1000, 4444, 12346, 20002])
def test_encodec48k_streaming_kernels_on_odd_segment_lengths(tail):
    """Round 6: the streaming kernels of the 48 kHz model's outer stages (fused first pass of the residual blocks, stride-2 / 4 / 5 down- and
    stride-2 / 4 up-convolutions: nc_resa / nc_down2 / nc_down4 / nc_down5 / nc_up2) take a layer only when its row length suits their lane
    layout (even, a multiple of 4 / 5, 16-byte aligned rows ...); everything else keeps the windowed launches.  One clip whose SECOND
    segment is `tail` samples long walks the stack at lengths where some layers qualify and others do not (482 -> 241 -> ...; 1000 -> 500 ->
    125 -> 25; 12346 -> 6173 ...), with the reflect fixes and halo clamps of short rows: engine == C oracle bit for bit at full width."""
    g, cfg, m, ref = _setup("encodec48k_b1")
    T = 47520 + tail                                             # segment stride 47520 (Encodec.cs:278-282): segments of 48000 and `tail` samples
    pcm = synthetic_pcm(2, cfg.channels, T, cfg.sampling_rate, seed=100 + tail)
    frames, _, _ = _check_vs_oracle(m, ref, pcm)
    assert len(frames) == 2
    m.dispose()
