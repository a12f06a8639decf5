"""GPU suite: DAC Encode / Decode / FromCodes through the C ABI against the oracle and the golden vectors."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from conftest import audit_code_mismatches, dac_cfg_from_meta  # noqa: E402
from neuralcodecs_amd import DAC, DACConfig  # noqa: E402
from neuralcodecs_amd.weights import dac_synthetic_state_dict, save_blob, synthetic_pcm  # noqa: E402
from oracle import c_oracle  # noqa: E402

PCM_TOL = 1e-4
LATENT_TOL = 2e-5
GAP_TOL = 1e-4


@pytest.fixture(scope="module")
def small(golden_small):
    cfg = dac_cfg_from_meta(golden_small["meta"])
    blob = save_blob(dac_synthetic_state_dict(cfg, seed=golden_small["meta"]["weight_seed"]))
    m = DAC(cfg)
    m.load_blob(blob)
    yield cfg, m, c_oracle.RefDAC(cfg, blob)
    m.dispose()


@pytest.fixture(scope="module")
def full(golden_full):
    cfg = dac_cfg_from_meta(golden_full["meta"])
    blob = save_blob(dac_synthetic_state_dict(cfg, seed=golden_full["meta"]["weight_seed"]))
    m = DAC(cfg)
    m.load_blob(blob)
    yield cfg, m, c_oracle.RefDAC(cfg, blob)
    m.dispose()


def test_small_encode_decode_vs_golden(small, golden_small):
    cfg, m, ref = small
    z, codes, lat, cl, cbl = m.encode(golden_small["pcm"])
    assert codes.dtype == np.int64 and codes.shape == (2, 4, 7)
    assert float(cl) == 0.0 and float(cbl) == 0.0                      # D5: dummy losses
    audit_code_mismatches(codes, golden_small["codes"], golden_small["gap"], GAP_TOL)
    assert np.array_equal(codes, golden_small["codes"]), "the golden fixture no longer matches: codes flipped (regenerate only with an audited reason)"
    assert np.abs(z - golden_small["zq"]).max() < LATENT_TOL
    assert np.abs(lat - golden_small["latents"]).max() < LATENT_TOL
    audio = m.decode(golden_small["zq"])
    assert audio.shape == golden_small["audio"].shape
    assert np.abs(audio - golden_small["audio"]).max() < PCM_TOL


def test_small_bit_exact_vs_c_oracle(small, golden_small):
    cfg, m, ref = small
    z, codes, lat, _, _ = m.encode(golden_small["pcm"])
    rz, rcodes, rlat, _ = ref.encode(golden_small["pcm"])
    assert np.array_equal(codes, rcodes)
    assert np.array_equal(lat, rlat)
    assert np.array_equal(z, rz)
    assert np.array_equal(m.decode(z), ref.decode(rz))
    assert np.array_equal(m.from_codes(codes), ref.from_codes(rcodes))


def test_small_n_quantizers_and_from_codes(small, golden_small):
    cfg, m, ref = small
    z2, codes2, lat2, _, _ = m.encode(golden_small["pcm"], n_quantizers=2)
    assert codes2.shape == (2, 2, 7) and lat2.shape == (2, 16, 7)
    assert np.array_equal(codes2, golden_small["codes_nq2"])
    assert np.abs(z2 - golden_small["zq_nq2"]).max() < LATENT_TOL
    zf = m.from_codes(golden_small["codes"].astype(np.int64))
    assert np.abs(zf - golden_small["from_codes"]).max() < LATENT_TOL


def test_float_array_overloads(small, golden_small):
    """DAC.Encode(float[]) returns the flattened zQ latents, Decode(float[]) consumes them (D12)."""
    cfg, m, ref = small
    pcm = golden_small["pcm"][0, 0]
    zflat = m.encode_array(pcm)
    assert zflat.shape == (128 * 7,)
    rz = ref.encode(pcm.reshape(1, 1, -1))[0]
    assert np.array_equal(zflat, rz.reshape(-1))
    out = m.decode_array(zflat)
    assert np.array_equal(out, ref.decode(rz).reshape(-1))
    assert np.array_equal(m.forward_array(pcm), out)


def test_error_conventions(small, golden_small):
    cfg, m, ref = small
    with pytest.raises(ValueError, match="sample rate"):
        m.encode(golden_small["pcm"], sample_rate=cfg.sample_rate + 1)          # ArgumentException, DAC.cs:146
    with pytest.raises(ValueError):
        m.encode(None)
    with pytest.raises(ValueError):
        m.decode(np.zeros((1, 3, 7), np.float32))
    with pytest.raises(FileNotFoundError):
        m.load_weights("/nonexistent/weights.ncwb")                               # FileNotFoundException, DAC.cs:347
    fresh = DAC(cfg)
    with pytest.raises(RuntimeError):
        fresh.encode(golden_small["pcm"])                                         # weights not loaded
    fresh.dispose()


def test_ragged_and_tiny_inputs(small):
    cfg, m, ref = small
    for T in (1, 319, 320, 321, 5000):
        pcm = synthetic_pcm(1, 1, T, cfg.sample_rate, seed=T)
        z, codes, lat, _, _ = m.encode(pcm)
        rz, rcodes, rlat, _ = ref.encode(pcm)
        assert codes.shape == (1, cfg.n_codebooks, -(-T // cfg.hop_length))
        assert np.array_equal(codes, rcodes) and np.array_equal(z, rz)
        assert np.array_equal(m.decode(z), ref.decode(rz))


def test_full_size_dac44k_vs_golden_and_oracle(full, golden_full):
    """BASELINE config C2 shape at B=2: clip 0 is the golden clip; both clips must equal the C oracle bit for bit."""
    cfg, m, ref = full
    meta = golden_full["meta"]
    pcm = synthetic_pcm(2, 1, meta["T"], cfg.sample_rate, seed=meta["pcm_seed"])
    z, codes, lat, _, _ = m.encode(pcm)
    assert codes.shape == (2, 9, 87)
    diverged = audit_code_mismatches(codes[:1], golden_full["codes"], golden_full["gap"], GAP_TOL)
    audio = m.decode(z)
    assert audio.shape == (2, 1, 44544)
    assert diverged == 0, "the golden fixture no longer matches: codes flipped (regenerate only with an audited reason)"
    assert np.abs(z[:1, ::16, :] - golden_full["zq_slice"]).max() < LATENT_TOL
    assert np.abs(audio[:1, :, ::29] - golden_full["audio_slice"]).max() < PCM_TOL
    rz, rcodes, rlat, _ = ref.encode(pcm)
    assert np.array_equal(codes, rcodes)
    assert np.array_equal(z, rz)
    assert np.array_equal(audio, ref.decode(rz))


def test_device_tensor_api_matches_host_api(full, golden_full):
    """torch CUDA tensors in/out (zero-copy *_dev entry points on torch's stream) == host-buffer API."""
    import torch
    cfg, m, ref = full
    meta = golden_full["meta"]
    pcm = synthetic_pcm(2, 1, meta["T"], cfg.sample_rate, seed=meta["pcm_seed"])
    z, codes, lat, _, _ = m.encode(pcm)
    audio = m.decode(z)
    xd = torch.from_numpy(pcm).cuda()
    zd, cd, ld, _, _ = m.encode(xd)
    ad = m.decode(zd)
    fd = m.from_codes(cd)
    torch.cuda.synchronize()
    assert np.array_equal(cd.cpu().numpy(), codes)
    assert np.array_equal(zd.cpu().numpy(), z)
    assert np.array_equal(ad.cpu().numpy(), audio)
    assert np.array_equal(fd.cpu().numpy(), m.from_codes(codes))


def test_batch_invariance_and_determinism(full, golden_full):
    """Clips are independent: clip i of a batch equals the same clip encoded alone; repeated calls are identical."""
    cfg, m, ref = full
    meta = golden_full["meta"]
    pcm = synthetic_pcm(3, 1, meta["T"], cfg.sample_rate, seed=meta["pcm_seed"])
    z, codes, _, _, _ = m.encode(pcm)
    z1, codes1, _, _, _ = m.encode(pcm[2:3])
    assert np.array_equal(codes[2:3], codes1) and np.array_equal(z[2:3], z1)
    z_again, codes_again, _, _, _ = m.encode(pcm)
    assert np.array_equal(codes, codes_again) and np.array_equal(z, z_again)


def test_dia_glue_code_matrix(small, golden_small):
    """Dia's [T, n_q] code-matrix round trip over DAC (Models/Dia.cs:973-1002, Modules/Dia/AudioUtils.cs:189-199)."""
    cfg, m, ref = small
    pcm = golden_small["pcm"]
    mat = m.encode_to_code_matrix(pcm[0], sample_rate=cfg.sample_rate)              # [T', n_q]
    _, codes, _, _, _ = m.encode(pcm[:1])
    assert mat.shape == (codes.shape[2], cfg.n_codebooks) and np.array_equal(mat, codes[0].T)
    wav = m.decode_code_matrix(mat)
    assert np.array_equal(wav, m.decode(m.from_codes(codes)).reshape(-1))
    batch = m.encode_to_code_matrix(pcm)                                           # batched prompts
    assert batch.shape[0] == 2 and np.array_equal(batch[0], mat)
    assert np.array_equal(m.decode_code_matrix(batch)[0], wav)
    assert np.array_equal(type(m).decode_one_frame(m, codes), m.decode(m.from_codes(codes)))
    with pytest.raises(ValueError, match="one frame"):
        type(m).decode_one_frame(m, np.concatenate([codes, codes]))


def test_c99_consumer_runs_the_codec_through_the_abi(tmp_path, golden_small):
    """A compiled C99 program (tests/abi_consumer.c) -- no Python, no ctypes between caller and library -- creates a DAC handle, loads the
    weight blob from a file, encodes and decodes through the host-pointer entry points and compares with fixtures the C oracle produced:
    the nearest thing in this image to what the managed [DllImport] binding does (bindings/csharp/DAC.Native.cs)."""
    import ctypes as C
    import os
    import subprocess
    from neuralcodecs_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = dac_cfg_from_meta(golden_small["meta"])
    blob = save_blob(dac_synthetic_state_dict(cfg, seed=golden_small["meta"]["weight_seed"]))
    pcm = synthetic_pcm(3, 1, 2999, cfg.sample_rate, seed=21)              # (a ragged length: Preprocess pads it)
    ref = c_oracle.RefDAC(cfg, blob)
    rz, rcodes, _, _ = ref.encode(pcm)
    raudio = ref.decode(rz)
    c = _lib.NcDacConfig()
    c.sample_rate, c.encoder_dim, c.decoder_dim = cfg.sample_rate, cfg.encoder_dim, cfg.decoder_dim
    c.n_encoder_rates, c.n_decoder_rates = len(cfg.encoder_rates), len(cfg.decoder_rates)
    for i, r in enumerate(cfg.encoder_rates):
        c.encoder_rates[i] = r
    for i, r in enumerate(cfg.decoder_rates):
        c.decoder_rates[i] = r
    c.latent_dim = cfg.resolved_latent_dim
    c.n_codebooks, c.codebook_size, c.codebook_dim = cfg.n_codebooks, cfg.codebook_size, cfg.codebook_dim
    (tmp_path / "config.bin").write_bytes(bytes(C.string_at(C.addressof(c), C.sizeof(c))))
    (tmp_path / "weights.blob").write_bytes(blob)
    pcm.tofile(tmp_path / "pcm.f32")
    rcodes.astype(np.int64).tofile(tmp_path / "codes.i64")
    raudio.astype(np.float32).tofile(tmp_path / "audio.f32")
    exe = tmp_path / "abi_consumer"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-O1", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "abi_consumer.c"),
                           "-o", str(exe), "-L", libdir, "-l:libnc_mi355x.so", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-lm"])
    r = subprocess.run([str(exe), str(tmp_path / "config.bin"), str(tmp_path / "weights.blob"), str(tmp_path / "pcm.f32"), "3", "2999",
                        str(tmp_path / "codes.i64"), str(tmp_path / "audio.f32")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "CONSUMER_OK" in r.stdout, (r.returncode, r.stdout[-1000:], r.stderr[-2000:])


# ---- round 5: the reference's other presets at full width + an adversarial quantizer ---------------------------

def _full_case(name, B):
    from conftest import load_golden
    from neuralcodecs_amd.weights import tie_codebooks
    g = load_golden(name)
    meta = g["meta"]
    cfg = dac_cfg_from_meta(meta)
    sd = dac_synthetic_state_dict(cfg, seed=meta["weight_seed"])
    if meta.get("ties"):
        tie_codebooks(sd)
    blob = save_blob(sd)
    m = DAC(cfg)
    m.load_blob(blob)
    return g, cfg, m, c_oracle.RefDAC(cfg, blob), synthetic_pcm(B, 1, meta["T"], cfg.sample_rate, seed=meta["pcm_seed"])


def test_dac24k_full_width_stride5_32_codebooks_vs_golden_and_oracle():
    """DAC 24 kHz (Config/DAC/DACConfig.cs:115-124): 32 codebooks, rates 2-4-5-8 -- the stride-5 layers at FULL width (640 -> 1280 channels
    k = 10 down, 768 -> 384 up) and a 32-stage quantizer; clip 0 against the ATen golden, both clips bit-exact against the C oracle."""
    g, cfg, m, ref, pcm = _full_case("dac24k_b1", 2)
    try:
        z, codes, lat, _, _ = m.encode(pcm)
        assert codes.shape == (2, 32, 75)
        assert audit_code_mismatches(codes[:1], g["codes"], g["gap"], GAP_TOL) == 0
        audio = m.decode(z)
        assert audio.shape == (2, 1, 23992)      # (L_out = 5 L - 1 through the stride-5 DecoderBlock: DecoderBlock.cs:20-44)
        assert np.abs(z[:1, ::16, :] - g["zq_slice"]).max() < LATENT_TOL
        assert np.abs(audio[:1, :, ::29] - g["audio_slice"]).max() < PCM_TOL
        rz, rcodes, rlat, _ = ref.encode(pcm)
        assert np.array_equal(codes, rcodes) and np.array_equal(z, rz) and np.array_equal(lat, rlat)
        assert np.array_equal(audio, ref.decode(rz))
        assert np.array_equal(m.from_codes(codes), ref.from_codes(rcodes))
    finally:
        m.dispose()


@pytest.mark.parametrize("preset,shape", [("dac_44khz_16kbps", (2, 18, 87)), ("dac_16khz", (2, 12, 50))])
def test_other_dac_presets_full_width_bit_exact_vs_oracle(preset, shape):
    """DAC 44 kHz-16 kbps (18 codebooks, latent 128: the quantizer's stage-wise path) and 16 kHz (12 codebooks, stride 5), full width."""
    cfg = getattr(DACConfig, preset)()
    blob = save_blob(dac_synthetic_state_dict(cfg, seed=42))
    pcm = synthetic_pcm(2, 1, cfg.sample_rate, cfg.sample_rate, seed=77)
    ref = c_oracle.RefDAC(cfg, blob)
    with DAC(cfg) as m:
        m.load_blob(blob)
        z, codes, lat, _, _ = m.encode(pcm)
        audio = m.decode(z)
    assert codes.shape == shape
    rz, rcodes, rlat, _ = ref.encode(pcm)
    assert np.array_equal(codes, rcodes) and np.array_equal(z, rz) and np.array_equal(lat, rlat)
    assert np.array_equal(audio, ref.decode(rz))


def test_tied_codebooks_return_atens_first_index_full_size():
    """Every codebook row twice (an exact tie in EVERY frame of EVERY stage) + dead codes, full-size DAC 44.1 kHz: the fused quantizer kernel
    (wavefront min-reduction with the lowest index on ties) must emit exactly the codes ATen's argmin emits (tests/golden/dac44k_ties_b1.npz),
    all in the lower half, none on a dead row -- through the stage-fused kernel and the stage-wise one (n_quantizers < 9 takes the same
    kernel; the C oracle is the third witness)."""
    g, cfg, m, ref, pcm = _full_case("dac44k_ties_b1", 2)
    try:
        want = g["codes"].astype(np.int64)
        assert float(g["gap"].max()) == 0.0 and want.max() < cfg.codebook_size // 2 and not np.any(want % 7 == 0)
        z, codes, lat, _, _ = m.encode(pcm)
        assert np.array_equal(codes[:1], want), f"{int((codes[:1] != want).sum())} codes differ from ATen's first-index choice"
        assert codes.max() < cfg.codebook_size // 2 and not np.any(codes % 7 == 0)
        rz, rcodes, _, _ = ref.encode(pcm)
        assert np.array_equal(codes, rcodes) and np.array_equal(z, rz)
        assert np.abs(m.decode(z)[:1, :, ::29] - g["audio_slice"]).max() < PCM_TOL
    finally:
        m.dispose()
