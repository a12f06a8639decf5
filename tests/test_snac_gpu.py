"""GPU suite: SNAC Encode / FromCodes / Decode through the C ABI against the C oracle (bit for bit) and the golden vectors."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from conftest import audit_snac_levels, load_golden, snac_cfg_from_meta  # noqa: E402
from neuralcodecs_amd import SNAC  # noqa: E402
from neuralcodecs_amd.weights import save_blob, snac_noise, snac_synthetic_state_dict, synthetic_pcm  # noqa: E402
from oracle import c_oracle  # noqa: E402

PCM_TOL, LATENT_TOL, GAP_TOL = 1e-4, 3e-5, 1e-4


def _setup(name):
    g = load_golden(name)
    cfg = snac_cfg_from_meta(g["meta"])
    blob = save_blob(snac_synthetic_state_dict(cfg, seed=g["meta"]["weight_seed"]))
    m = SNAC(cfg)
    m.load_blob(blob)
    return g, cfg, m, c_oracle.RefSNAC(cfg, blob)


@pytest.mark.parametrize("name", ["snac_small", "snac_small_attn"])
def test_snac_small_vs_golden_and_oracle(name):
    g, cfg, m, ref = _setup(name)
    meta = g["meta"]
    codes, z, zq = m.encode(g["pcm"], return_latents=True)
    rz, rzq, rcodes = ref.encode(g["pcm"])
    assert all(c.dtype == np.int64 for c in codes)
    for a, b in zip(codes, rcodes):
        assert np.array_equal(a, b)                                   # bit-exact codes vs the C oracle
    assert np.array_equal(z, rz) and np.array_equal(zq, rzq)
    assert np.abs(z - g["z"]).max() < LATENT_TOL
    assert audit_snac_levels(codes, g, GAP_TOL) == 0, "the golden fixture no longer matches: codes flipped (regenerate only with an audited reason)"
    assert np.abs(zq - g["zq"]).max() < LATENT_TOL
    Tz = z.shape[-1]
    noises = snac_noise(cfg, meta["B"], Tz, seed=meta["noise_seed"])
    gold_codes = [g[f"codes{i}"].astype(np.int64) for i in range(len(cfg.vq_strides))]
    audio = m.decode(gold_codes, noises)
    assert np.array_equal(audio, ref.decode(gold_codes, noises))
    assert np.abs(audio - g["audio"]).max() < PCM_TOL
    assert np.array_equal(m.from_codes(gold_codes), ref.from_codes(gold_codes))
    m.dispose()


def test_snac24k_full_size_config_c1():
    """BASELINE config C1 (SNAC 24 kHz mono, 1 s clip) plus a second clip: codes 12/24/48 per clip."""
    g, cfg, m, ref = _setup("snac24k_b1")
    meta = g["meta"]
    pcm = synthetic_pcm(2, 1, meta["T"], cfg.sampling_rate, seed=meta["pcm_seed"])
    codes, z, zq = m.encode(pcm, return_latents=True)
    assert [c.shape for c in codes] == [(2, 12), (2, 24), (2, 48)]
    rz, rzq, rcodes = ref.encode(pcm)
    for a, b in zip(codes, rcodes):
        assert np.array_equal(a, b)
    assert np.array_equal(zq, rzq)
    assert audit_snac_levels([c[:1] for c in codes], g, GAP_TOL) == 0, "the golden fixture no longer matches: codes flipped (regenerate only with an audited reason)"
    assert np.abs(zq[:1, ::16, :] - g["zq_slice"]).max() < LATENT_TOL
    noises = snac_noise(cfg, 2, 48, seed=meta["noise_seed"])
    audio = m.decode(codes, noises)
    assert audio.shape == (2, 1, 24576)
    assert np.array_equal(audio, ref.decode(rcodes, noises))
    gold_codes = [g[f"codes{i}"].astype(np.int64) for i in range(3)]
    a1 = m.decode(gold_codes, snac_noise(cfg, 1, 48, seed=meta["noise_seed"]))
    assert np.abs(a1[:, :, ::17] - g["audio_slice"]).max() < PCM_TOL
    m.dispose()


def test_snac_api_shapes_errors_and_noise_source():
    g, cfg, m, ref = _setup("snac_small")
    pcm = g["pcm"]
    # SNACValidator.ValidateModel contract (Config/SNAC/SNACValidator.cs:95-111): #codebooks, 3-D output, ~input length
    codes = m.encode(pcm)
    assert len(codes) == len(cfg.vq_strides)
    audio, codes2 = m.forward(pcm, seed=5)
    assert audio.shape == pcm.shape and all(np.array_equal(a, b) for a, b in zip(codes, codes2))
    # device noise source: deterministic in the seed, different across seeds, and actually used
    a5 = m.decode(codes, seed=5); a5b = m.decode(codes, seed=5); a6 = m.decode(codes, seed=6)
    assert np.array_equal(a5, a5b) and not np.array_equal(a5, a6)
    # float[] overloads
    fl = m.encode_array(pcm[0, 0])
    assert all(c.dtype == np.float32 for c in fl) and np.array_equal(fl[2].astype(np.int64), codes[2][0])
    nz = snac_noise(cfg, 1, codes[-1].shape[-1] * cfg.vq_strides[-1], seed=1)
    assert np.array_equal(m.decode_array(fl, nz), m.decode([c[:1] for c in codes], nz).reshape(-1))
    with pytest.raises(ValueError):
        m.decode(codes[:2])                                            # ArgumentException: wrong number of codebooks
    with pytest.raises(ValueError):
        m.decode_array([])
    with pytest.raises(ValueError):
        m.encode(None)
    fresh = SNAC(cfg)
    with pytest.raises(RuntimeError):
        fresh.encode(pcm)
    fresh.dispose()
    out = m.process_audio(pcm[0, 0], cfg.sampling_rate // 2, seed=3)      # resample x2 then forward
    assert out.shape[0] == 2 * pcm.shape[-1]
    m.dispose()


def test_snac_process_audio_one_call_vs_oracle_composition():
    """SNAC.ProcessAudio (Models/SNAC.cs:255-308) behind ONE ABI call (nc_snac_process_audio: upload, resample, forward, download)
    against the oracle composition ResampleAudio -> Preprocess -> encode -> decode -> narrow, bit for bit; same-rate input skips the
    resampler; empty input is the reference's ArgumentException."""
    from oracle import audio_ref
    g, cfg, m, ref = _setup("snac_small")
    x = synthetic_pcm(1, 1, 1777, cfg.sampling_rate, seed=31)[0, 0]
    for src in (cfg.sampling_rate, cfg.sampling_rate * 2 // 3, 44100):
        xr = x if src == cfg.sampling_rate else audio_ref.resample_linear(x, src, cfg.sampling_rate)
        _, frames, _, _ = m.query(xr.size)
        nz = snac_noise(cfg, 1, frames, seed=9)
        out = m.process_audio(x, src, noise=nz)
        _, _, rcodes = ref.encode(xr.reshape(1, 1, -1))
        want = ref.decode(rcodes, nz).reshape(-1)[: xr.size]
        assert out.shape == (xr.size,) and np.array_equal(out, want)
        assert np.array_equal(out, m.forward(xr.reshape(1, 1, -1), nz)[0].reshape(-1))   # == the two-step path through the other entry points
    with pytest.raises(ValueError):
        m.process_audio(np.zeros(0, np.float32), cfg.sampling_rate)
    with pytest.raises(ValueError):
        m.process_audio(None, cfg.sampling_rate)
    m.dispose()


def test_snac_device_tensor_api_and_batch_invariance():
    import torch
    g, cfg, m, ref = _setup("snac_small_attn")
    pcm = synthetic_pcm(3, 1, 2100, cfg.sampling_rate, seed=21)
    codes = m.encode(pcm)
    one = m.encode(pcm[1:2])
    assert all(np.array_equal(a[1:2], b) for a, b in zip(codes, one))
    dc = m.encode(torch.from_numpy(pcm).cuda())
    nz = snac_noise(cfg, 3, codes[-1].shape[-1], seed=2)
    da = m.decode(dc, [torch.from_numpy(n).cuda() for n in nz])
    torch.cuda.synchronize()
    assert all(np.array_equal(a.cpu().numpy(), b) for a, b in zip(dc, codes))
    assert np.array_equal(da.cpu().numpy(), m.decode(codes, nz))
    # marshalling fast paths: encode()'s level views and flat_noise()'s views are handed to the ABI as the flat buffers underneath
    # (no copy); lists assembled from other tensors take the concatenating path -- same result either way
    assert m._as_one_buffer(dc, 3) is not None and m._as_one_buffer([c.clone() for c in dc], 3) is None
    fn = m.flat_noise(nz, "cuda")
    assert m._noise_end_to_end(fn) is not None and all(np.array_equal(a.cpu().numpy(), b) for a, b in zip(fn, nz))
    d1 = m.decode(dc, fn)
    d2 = m.decode([c.clone() for c in dc], [n.clone() for n in fn])
    fn[0].zero_()                                                      # in-place change of the noise must reach the next decode (nothing is cached)
    d3 = m.decode(dc, fn)
    torch.cuda.synchronize()
    assert torch.equal(d1, da) and torch.equal(d2, da) and not torch.equal(d3, da)
    m.dispose()


def test_snac44k_attention_full_width_config_c5_shape():
    """BASELINE config C5 model (SNAC 44.1 kHz, LocalMHA window 32): golden short clip + one full 5 s clip vs the C oracle."""
    g, cfg, m, ref = _setup("snac44k_short")
    meta = g["meta"]
    pcm = synthetic_pcm(2, 1, meta["T"], cfg.sampling_rate, seed=meta["pcm_seed"])
    codes, z, zq = m.encode(pcm, return_latents=True)
    rz, rzq, rcodes = ref.encode(pcm)
    assert [c.shape for c in codes] == [(2, 8), (2, 16), (2, 32), (2, 64)]
    for a, b in zip(codes, rcodes):
        assert np.array_equal(a, b)
    assert np.array_equal(z, rz) and np.array_equal(zq, rzq)
    assert audit_snac_levels([c[:1] for c in codes], g, GAP_TOL) == 0, "the golden fixture no longer matches: codes flipped (regenerate only with an audited reason)"
    assert np.abs(zq[:1, ::16, :] - g["zq_slice"]).max() < LATENT_TOL
    nz = snac_noise(cfg, 2, 64, seed=meta["noise_seed"])
    audio = m.decode(codes, nz)
    assert np.array_equal(audio, ref.decode(rcodes, nz))
    gold_codes = [g[f"codes{i}"].astype(np.int64) for i in range(4)]
    a1 = m.decode(gold_codes, snac_noise(cfg, 1, 64, seed=meta["noise_seed"]))
    assert np.abs(a1[:, :, ::17] - g["audio_slice"]).max() < PCM_TOL
    # one 5 s clip: 220500 -> padded 221184 = 18 * 12288, T' = 576 (18 windows), codes 72/144/288/576
    pcm5 = synthetic_pcm(1, 1, 220500, cfg.sampling_rate, seed=8)
    c5 = m.encode(pcm5)
    assert [c.shape for c in c5] == [(1, 72), (1, 144), (1, 288), (1, 576)]
    _, _, r5 = ref.encode(pcm5)
    for a, b in zip(c5, r5):
        assert np.array_equal(a, b)
    nz5 = snac_noise(cfg, 1, 576, seed=9)
    assert np.array_equal(m.decode(c5, nz5), ref.decode(r5, nz5))
    m.dispose()


def test_snac_encode_tensor_overload_as_written():
    """SNAC.Encode(Tensor) as written (Models/SNAC.cs:113-122, D7): no pad.  Engine == C oracle bit for bit, == torch goldens within
    tolerance; the padding path gives other frame counts on the same input; lengths the reference throws on give ValueError."""
    g, cfg, m, ref = _setup("snac_small_tensor")
    codes, z, zq = m.encode_tensor(g["pcm"], return_latents=True)
    rz, rzq, rcodes = ref.encode_tensor(g["pcm"])
    for a, b in zip(codes, rcodes):
        assert np.array_equal(a, b)
    assert np.array_equal(z, rz) and np.array_equal(zq, rzq)
    assert np.abs(z - g["z"]).max() < LATENT_TOL
    assert audit_snac_levels(codes, g, GAP_TOL) == 0
    assert np.abs(zq - g["zq"]).max() < LATENT_TOL
    assert m.encode(g["pcm"])[-1].shape[-1] == 36 and codes[-1].shape[-1] == 32
    with pytest.raises(ValueError):
        m.encode_tensor(synthetic_pcm(1, 1, 3001, cfg.sampling_rate, seed=1))
    import torch
    cd = m.encode_tensor(torch.from_numpy(g["pcm"]).cuda())
    for a, b in zip(cd, codes):
        assert np.array_equal(a.cpu().numpy(), b)
    m.dispose()
    g, cfg, m, ref = _setup("snac24k_tensor_b1")
    meta = g["meta"]
    pcm = synthetic_pcm(2, 1, meta["T"], cfg.sampling_rate, seed=meta["pcm_seed"])
    codes, z, zq = m.encode_tensor(pcm, return_latents=True)
    rz, rzq, rcodes = ref.encode_tensor(pcm)
    assert [c.shape for c in codes] == [(2, 11), (2, 22), (2, 44)]
    for a, b in zip(codes, rcodes):
        assert np.array_equal(a, b)
    assert np.array_equal(zq, rzq)
    assert audit_snac_levels([c[:1] for c in codes], g, GAP_TOL) == 0
    assert np.abs(zq[:1, ::16, :] - g["zq_slice"]).max() < LATENT_TOL
    m.dispose()


@pytest.mark.parametrize("preset,seconds", [("snac_44khz", 1), ("snac_32khz", 1), ("snac_24khz", 1)])
def test_snac_tied_codebooks_first_index_and_other_presets_full_width(preset, seconds):
    """Every preset the reference ships (Config/SNAC/SNACConfig.cs: 24 / 32 / 44.1 kHz) at full width with ADVERSARIAL codebooks: every row
    of the 4096-entry codebooks twice (an exact tie in every frame of every level) and dead codes (neuralcodecs_amd.weights.tie_codebooks).
    ATen's argmin (SNAC/VectorQuantizer.cs:137) returns the first index of a tie: all codes in the lower half, none on a dead row, and the
    engine equal to the C oracle bit for bit (whose tie rule tests/test_oracle_snac_cpu.py holds to the ATen restatement)."""
    from neuralcodecs_amd.config import SNACConfig
    from neuralcodecs_amd.weights import tie_codebooks
    cfg = getattr(SNACConfig, preset)()
    blob = save_blob(tie_codebooks(snac_synthetic_state_dict(cfg, seed=42)))
    pcm = synthetic_pcm(2, 1, seconds * cfg.sampling_rate, cfg.sampling_rate, seed=31)
    ref = c_oracle.RefSNAC(cfg, blob)
    with SNAC(cfg) as m:
        m.load_blob(blob)
        codes = m.encode(pcm)
        nz = snac_noise(cfg, 2, codes[-1].shape[1], seed=8)
        audio = m.decode(codes, nz)
    _, _, rcodes = ref.encode(pcm)
    for a, b in zip(codes, rcodes):
        assert np.array_equal(a, b)
        assert a.max() < cfg.codebook_size // 2 and not np.any(a % 7 == 0), "a tie was not resolved to the first index"
    assert np.array_equal(audio, ref.decode(rcodes, nz))
