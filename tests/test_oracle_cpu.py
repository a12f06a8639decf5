"""CPU suite: the C oracle against the golden vectors of the PyTorch-CPU restatement, and oracle unit cases."""
import numpy as np
import pytest

from conftest import audit_code_mismatches, dac_cfg_from_meta
from neuralcodecs_amd.weights import dac_synthetic_state_dict, save_blob, synthetic_pcm
from oracle import c_oracle

PCM_TOL = 1e-4      # north_star: decoded float PCM within 1e-4 max-abs
LATENT_TOL = 2e-5   # fp32 round-off through ~25 layers on O(1..5) activations
GAP_TOL = 1e-4      # a flipped argmin must have a top-2 squared-distance gap below this


@pytest.fixture(scope="module")
def small(golden_small):
    cfg = dac_cfg_from_meta(golden_small["meta"])
    blob = save_blob(dac_synthetic_state_dict(cfg, seed=golden_small["meta"]["weight_seed"]))
    return cfg, c_oracle.RefDAC(cfg, blob)


def test_synthetic_pcm_is_reproducible(golden_small):
    m = golden_small["meta"]
    pcm = synthetic_pcm(m["B"], 1, m["T"], m["cfg"]["sample_rate"], seed=m["pcm_seed"])
    assert np.array_equal(pcm, golden_small["pcm"])


def test_c_oracle_encode_matches_golden(small, golden_small):
    cfg, ref = small
    zq, codes, lat, _ = ref.encode(golden_small["pcm"])
    assert codes.shape == golden_small["codes"].shape and codes.dtype == np.int64
    audit_code_mismatches(codes, golden_small["codes"], golden_small["gap"], GAP_TOL)
    assert np.array_equal(codes, golden_small["codes"]), "the golden fixture no longer matches: codes flipped (regenerate only with an audited reason)"
    assert np.abs(zq - golden_small["zq"]).max() < LATENT_TOL
    assert np.abs(lat - golden_small["latents"]).max() < LATENT_TOL


def test_c_oracle_encode_nq2(small, golden_small):
    cfg, ref = small
    zq, codes, lat, _ = ref.encode(golden_small["pcm"], n_quantizers=2)
    assert codes.shape == (2, 2, 7)
    assert np.array_equal(codes, golden_small["codes_nq2"])
    assert np.abs(zq - golden_small["zq_nq2"]).max() < LATENT_TOL


def test_c_oracle_decode_matches_golden(small, golden_small):
    cfg, ref = small
    audio = ref.decode(golden_small["zq"])
    assert audio.shape == golden_small["audio"].shape  # (2,1,2232): odd stride 5 => not frames*hop (D6: no trim)
    assert np.abs(audio - golden_small["audio"]).max() < PCM_TOL


def test_c_oracle_from_codes(small, golden_small):
    cfg, ref = small
    z = ref.from_codes(golden_small["codes"].astype(np.int64))
    assert np.abs(z - golden_small["from_codes"]).max() < LATENT_TOL


def test_c_oracle_full_size_dac44k(golden_full):
    """BASELINE config C2 at B=1: 1 s of 44.1 kHz audio through the full 76.6 M-parameter graph."""
    cfg = dac_cfg_from_meta(golden_full["meta"])
    m = golden_full["meta"]
    blob = save_blob(dac_synthetic_state_dict(cfg, seed=m["weight_seed"]))
    ref = c_oracle.RefDAC(cfg, blob)
    pcm = synthetic_pcm(m["B"], 1, m["T"], cfg.sample_rate, seed=m["pcm_seed"])
    zq, codes, lat, _ = ref.encode(pcm)
    assert codes.shape == (1, 9, 87)
    diverged = audit_code_mismatches(codes, golden_full["codes"], golden_full["gap"], GAP_TOL)
    assert diverged == 0, "the golden fixture no longer matches: codes flipped (regenerate only with an audited reason)"
    assert np.abs(zq[:, ::16, :] - golden_full["zq_slice"]).max() < LATENT_TOL
    audio = ref.decode(zq)
    assert audio.shape == (1, 1, 44544)
    assert np.abs(audio[:, :, ::29] - golden_full["audio_slice"]).max() < PCM_TOL
    assert abs(np.abs(audio.astype(np.float64)).sum() - float(golden_full["audio_abs_sum"])) < 44544 * 2e-5


# ---- unit cases of the canonical arithmetic ---------------------------------------------------

def test_snake_alpha_cases():
    x = np.array([[[-2.0, -0.5, 0.0, 0.25, 3.0]] * 3], np.float32)
    alpha = np.array([0.0, 1.0, 2.0], np.float32)
    y = c_oracle.snake(x, alpha)
    assert np.array_equal(y[0, 0], x[0, 0])  # alpha == 0 -> identity (Snake1d.cs:52 where-branch)
    for c in (1, 2):
        want = x[0, c].astype(np.float64) + np.sin(alpha[c] * x[0, c].astype(np.float64)) ** 2 / alpha[c]
        assert np.abs(y[0, c] - want).max() < 5e-7


def test_sin_tanh_accuracy():
    xs = np.linspace(-30, 30, 200001).astype(np.float32).reshape(1, 1, -1)
    y = c_oracle.snake(xs, np.array([1.0], np.float32))
    want = xs.astype(np.float64) + np.sin(xs.astype(np.float64)) ** 2
    assert np.abs(y - want).max() < 4e-6   # |x|<=30: one ulp of 30 is 1.9e-6
    t = c_oracle.tanh(np.linspace(-12, 12, 100001).astype(np.float32))
    assert np.abs(t - np.tanh(np.linspace(-12, 12, 100001).astype(np.float32).astype(np.float64))).max() < 3e-7


def test_conv_length_formulas_and_values():
    rng = np.random.default_rng(0)
    import torch
    import torch.nn.functional as F
    for (cin, cout, k, s, p, d, T) in [(3, 5, 7, 1, 9, 3, 50), (4, 6, 4, 2, 1, 1, 37), (2, 3, 16, 8, 4, 1, 100), (5, 2, 10, 5, 3, 1, 83),
                                       (1, 4, 7, 1, 3, 1, 20), (6, 1, 3, 1, 1, 1, 9)]:
        x = rng.standard_normal((2, cin, T)).astype(np.float32)
        w = rng.standard_normal((cout, cin, k)).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        y = c_oracle.conv1d(x, w, b, s, p, d)
        want = F.conv1d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), s, p, d).numpy()
        assert y.shape == want.shape
        assert np.abs(y - want).max() < 1e-4
    for (cin, cout, s, T) in [(4, 3, 2, 11), (3, 5, 8, 7), (6, 2, 5, 9), (2, 2, 4, 1)]:
        k, p = 2 * s, (s + 1) // 2
        x = rng.standard_normal((2, cin, T)).astype(np.float32)
        w = rng.standard_normal((cin, cout, k)).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        y = c_oracle.conv_transpose1d(x, w, b, s, p)
        want = F.conv_transpose1d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), s, p).numpy()
        assert y.shape == want.shape
        assert np.abs(y - want).max() < 1e-4


def test_argmin_first_index_tie_break():
    cb = np.zeros((4, 8), np.float32)
    cb[0, 0] = 1.0
    cb[1, 0] = 0.0      # codes 1 and 2 are identical -> tie, lowest index (1) must win like ATen argmin
    cb[2, 0] = 0.0
    cb[3, 0] = 2.0
    z = np.zeros((1, 8, 3), np.float32)
    idx, st, _ = c_oracle.vq_argmin(z, cb)
    assert idx.tolist() == [[1, 1, 1]]
    import torch
    assert int(torch.tensor([[1.0, 0.0, 0.0, 2.0]]).argmin(1)) == 1


def test_weight_norm_fold_formula():
    rng = np.random.default_rng(1)
    v = rng.standard_normal((6, 4, 7)).astype(np.float32) * 0.1
    g = rng.random(6).astype(np.float32) + 0.5
    w = c_oracle.fold_wn_dac(v, g)
    import torch
    tv = torch.from_numpy(v)
    want = torch.mul(tv.div(tv.pow(2).sum([1, 2], keepdim=True).sqrt().add(1e-7)), torch.from_numpy(g).reshape(6, 1, 1)).numpy()
    assert np.abs(w - want).max() < 1e-6


# ---- round 5: another preset at full width, and an adversarial quantizer ---------------------------------------

def _full_case(name):
    from conftest import load_golden
    from neuralcodecs_amd.weights import tie_codebooks
    g = load_golden(name)
    m = g["meta"]
    cfg = dac_cfg_from_meta(m)
    sd = dac_synthetic_state_dict(cfg, seed=m["weight_seed"])
    if m.get("ties"):
        tie_codebooks(sd)
    return g, m, cfg, c_oracle.RefDAC(cfg, save_blob(sd)), synthetic_pcm(m["B"], 1, m["T"], cfg.sample_rate, seed=m["pcm_seed"])


def test_c_oracle_full_size_dac24k_stride5_32_codebooks():
    """Config/DAC/DACConfig.cs:115-124 (DAC 24 kHz: 32 codebooks, rates 2-4-5-8) at FULL width: 1 s -> 75 frames x 32 stages."""
    g, m, cfg, ref, pcm = _full_case("dac24k_b1")
    zq, codes, lat, _ = ref.encode(pcm)
    assert codes.shape == (1, 32, 75)
    assert audit_code_mismatches(codes, g["codes"], g["gap"], GAP_TOL) == 0
    assert np.abs(zq[:, ::16, :] - g["zq_slice"]).max() < LATENT_TOL
    audio = ref.decode(zq)
    assert audio.shape == (1, 1, 23992)          # (L_out = 5 L - 1 through the stride-5 DecoderBlock: DecoderBlock.cs:20-44)
    assert np.abs(audio[:, :, ::29] - g["audio_slice"]).max() < PCM_TOL


def test_c_oracle_tied_codebooks_return_atens_first_index():
    """Full-size DAC 44.1 kHz whose codebooks hold every row twice (an EXACT tie of the two best distances in every frame of every stage)
    and dead codes: the ATen restatement (tests/golden/dac44k_ties_b1.npz) emits the first index, and so must the canonical argmin."""
    g, m, cfg, ref, pcm = _full_case("dac44k_ties_b1")
    assert float(g["gap"].max()) == 0.0, "the fixture's ties are not exact"
    want = g["codes"].astype(np.int64)
    assert want.max() < cfg.codebook_size // 2 and not np.any(want % 7 == 0)   # lower half only, never a dead row
    zq, codes, lat, _ = ref.encode(pcm)
    assert np.array_equal(codes, want), f"{int((codes != want).sum())} codes differ from ATen's first-index choice"
    assert np.abs(ref.decode(zq)[:, :, ::29] - g["audio_slice"]).max() < PCM_TOL
