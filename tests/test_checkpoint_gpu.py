"""GPU suite, SURVEY 8f N1 + N3 on the ENGINE (not the oracle):

N1  an HF-named safetensors file and a Descript-style .pth are written, converted (neuralcodecs_amd/checkpoint.py, the reference's
    StateDictNameConverter / DACUnpickler rules), loaded into libnc_mi355x.so through the C ABI, and the engine's codes / latents / PCM
    must equal, bit for bit, the C oracle fed a TorchSharp-named dict that THIS TEST builds from the same tensors with the literal
    rules of the reference (weight_v = weight, weight_g = ||weight|| over dims (1,2); names from the literal table in
    tests/test_checkpoint_cpu.py) -- the converter and the test share no code.
N3  the Dia <-> DAC code-matrix glue: engine entry points nc_dac_{decode,encode}_code_matrix[_dev] against oracle/dia_glue_ref.py
    (statement-by-statement restatement of Models/Dia.cs:973-1002 and Modules/Dia/AudioUtils.cs:189-199) over the C oracle.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from conftest import dac_cfg_from_meta, load_golden  # noqa: E402
from neuralcodecs_amd import DAC, checkpoint  # noqa: E402
from neuralcodecs_amd.weights import dac_synthetic_state_dict, save_blob, synthetic_pcm  # noqa: E402
from oracle import c_oracle, dia_glue_ref  # noqa: E402
from test_checkpoint_cpu import REFERENCE_KEY_MAP  # noqa: E402  (literal transcription of BuildKeyMap)


def _hf_checkpoint(cfg, seed):
    """Random HF-named DAC tensors (plain `weight`s, flat alphas) + the TorchSharp-named dict the reference would build from them."""
    native = dac_synthetic_state_dict(cfg, seed=seed)             # only used as a shape catalogue
    inv = {v: k for k, v in REFERENCE_KEY_MAP.items()}
    rng = np.random.default_rng(seed)
    hf, want = {}, {}
    for k, v in native.items():
        if k.endswith(".weight_g"):
            continue
        if k.endswith(".weight_v"):
            base = k[: -len(".weight_v")]
            w = (rng.standard_normal(v.shape) * (0.5 / np.sqrt(np.prod(v.shape[1:])))).astype(np.float32)
            hf[inv.get(base, base) + ".weight"] = w
            want[base + ".weight_v"] = w
            want[base + ".weight_g"] = np.sqrt((w * w).sum(axis=(1, 2), keepdims=True, dtype=np.float32)).astype(np.float32)
        elif k.endswith(".alpha"):
            base = k[: -len(".alpha")]
            a = rng.uniform(0.5, 2.0, v.size).astype(np.float32)
            hf[inv[base] + ".alpha"] = a                           # HF stores [C]; the reference reshapes to [1, C, 1]
            want[k] = a.reshape(1, -1, 1)
        elif k.endswith(".bias"):
            base = k[: -len(".bias")]
            bv = (rng.standard_normal(v.shape) * 0.05).astype(np.float32)
            hf[inv.get(base, base) + ".bias"] = bv
            want[k] = bv
        else:                                                      # codebooks
            cv = rng.standard_normal(v.shape).astype(np.float32)
            hf[k] = cv
            want[k] = cv
    return hf, want


def _engine_vs_oracle(cfg, blob, want_sd, pcm):
    m = DAC(cfg)
    m.load_blob(blob)
    ref = c_oracle.RefDAC(cfg, save_blob(want_sd))
    z, codes, lat, _, _ = m.encode(pcm)
    rz, rcodes, rlat, _ = ref.encode(pcm)
    assert np.array_equal(codes, rcodes) and np.array_equal(z, rz) and np.array_equal(lat, rlat)
    assert np.array_equal(m.decode(z), ref.decode(rz))
    m.dispose()


def test_hf_safetensors_checkpoint_runs_on_the_engine(tmp_path):
    from safetensors.numpy import save_file
    g = load_golden("dac_small")
    cfg = dac_cfg_from_meta(g["meta"])
    hf, want = _hf_checkpoint(cfg, seed=21)
    assert "encoder.block.1.res_unit2.conv1.weight" in hf and "decoder.block.0.conv_t1.weight" in hf and "decoder.snake1.alpha" in hf
    p = tmp_path / "model.safetensors"
    save_file({k: np.ascontiguousarray(v) for k, v in hf.items()}, str(p))
    blob, meta_cfg = checkpoint.convert_checkpoint(str(p), "dac")
    assert meta_cfg is None
    _engine_vs_oracle(cfg, blob, want, synthetic_pcm(2, 1, 2500, cfg.sample_rate, seed=3))


def test_descript_pth_checkpoint_runs_on_the_engine(tmp_path):
    import torch
    g = load_golden("dac_small")
    cfg = dac_cfg_from_meta(g["meta"])
    sd = dac_synthetic_state_dict(cfg, seed=22)
    meta = {"kwargs": dict(sample_rate=cfg.sample_rate, encoder_dim=cfg.encoder_dim, encoder_rates=list(cfg.encoder_rates),
                           decoder_dim=cfg.decoder_dim, decoder_rates=list(cfg.decoder_rates), n_codebooks=cfg.n_codebooks,
                           codebook_size=cfg.codebook_size, codebook_dim=cfg.codebook_dim)}
    p = tmp_path / "weights.pth"
    torch.save({"state_dict": {k: torch.from_numpy(v) for k, v in sd.items()}, "metadata": meta}, str(p))
    blob, got_cfg = checkpoint.convert_checkpoint(str(p), "dac")
    assert got_cfg is not None and got_cfg.encoder_rates == cfg.encoder_rates and got_cfg.codebook_size == cfg.codebook_size
    _engine_vs_oracle(got_cfg, blob, sd, synthetic_pcm(2, 1, 2500, cfg.sample_rate, seed=4))
    # a converted blob written to disk loads through nc_codec_load_weights (the LoadWeights(path) member of INeuralCodec)
    q = tmp_path / "weights.ncwb"
    q.write_bytes(blob)
    m = DAC(got_cfg)
    m.load_weights(str(q))
    assert m.encode(synthetic_pcm(1, 1, 800, cfg.sample_rate, seed=5))[1].shape[1] == cfg.n_codebooks
    with pytest.raises(FileNotFoundError):
        m.load_weights(str(tmp_path / "missing.ncwb"))
    (tmp_path / "bad.ncwb").write_bytes(blob[: len(blob) // 2])        # truncated file -> ArgumentException-class error, no crash
    with pytest.raises(ValueError):
        m.load_weights(str(tmp_path / "bad.ncwb"))
    m.dispose()


def test_dia_code_matrix_glue_vs_oracle_restatement():
    import torch
    g = load_golden("dac_small")
    cfg = dac_cfg_from_meta(g["meta"])
    blob = save_blob(dac_synthetic_state_dict(cfg, seed=g["meta"]["weight_seed"]))
    m = DAC(cfg)
    m.load_blob(blob)
    ref = c_oracle.RefDAC(cfg, blob)
    pcm = synthetic_pcm(3, 1, 2300, cfg.sample_rate, seed=9)
    # Dia.Encode: [1, T] -> [T', n_q]
    mat = m.encode_to_code_matrix(pcm[0], sample_rate=cfg.sample_rate)
    rmat = dia_glue_ref.dia_encode(ref, pcm[0])
    assert mat.dtype == np.int64 and mat.shape == rmat.shape == (ref.encode(pcm[:1])[1].shape[2], cfg.n_codebooks)
    assert np.array_equal(mat, rmat)
    # Dia.Decode: [T', n_q] -> waveform
    wav = m.decode_code_matrix(mat)
    assert np.array_equal(wav, dia_glue_ref.dia_decode(ref, rmat))
    # batched prompts / batched decode == per-clip reference calls
    bm = m.encode_to_code_matrix(pcm)
    bw = m.decode_code_matrix(bm)
    for i in range(3):
        assert np.array_equal(bm[i], dia_glue_ref.dia_encode(ref, pcm[i]))
        assert np.array_equal(bw[i], dia_glue_ref.dia_decode(ref, bm[i]))
    # fewer codebooks in the matrix (Dia can carry a prefix of the codebooks)
    assert np.array_equal(m.decode_code_matrix(mat[:, :2]), dia_glue_ref.dia_decode(ref, rmat[:, :2]))
    # device API
    dm = m.encode_to_code_matrix(torch.from_numpy(pcm).cuda())
    dw = m.decode_code_matrix(dm)
    torch.cuda.synchronize()
    assert np.array_equal(dm.cpu().numpy(), bm) and np.array_equal(dw.cpu().numpy(), bw)
    # AudioUtils.Decode: exactly one [1, n_q, T] frame
    codes = np.ascontiguousarray(np.transpose(bm[:1], (0, 2, 1)))
    assert np.array_equal(DAC.decode_one_frame(m, codes), dia_glue_ref.audio_utils_decode(ref, codes))
    with pytest.raises(ValueError, match="one frame"):
        DAC.decode_one_frame(m, np.concatenate([codes, codes]))
    with pytest.raises(ValueError, match="one frame"):
        dia_glue_ref.audio_utils_decode(ref, np.concatenate([codes, codes]))
    with pytest.raises(ValueError):
        m.decode_code_matrix(np.zeros((2, 5, cfg.n_codebooks + 1), np.int64))
    m.dispose()


# ---- N1 for SNAC and Encodec on the ENGINE (VERDICT r2 "missing" 4): file -> converter -> blob -> C ABI == oracle fed the tensors directly ----
def test_snac_checkpoint_files_run_on_the_engine(tmp_path):
    """SNAC checkpoints carry the reference's own parameter names (`...parametrizations.weight.original0/1`, Modules/SNAC/WNConv1d.cs:66-70;
    LoadWeights: Models/SNAC.cs:200-231 dispatches on the file type).  A safetensors file and a torch .pth of the same tensors must both load
    and give, on the engine, exactly what the C oracle gives on the tensors themselves."""
    import torch
    from safetensors.numpy import save_file
    from conftest import snac_cfg_from_meta
    from neuralcodecs_amd import SNAC
    from neuralcodecs_amd.weights import snac_noise, snac_synthetic_state_dict
    g = load_golden("snac_small_attn")                                  # the LocalMHA variant: every parameter family is present
    cfg = snac_cfg_from_meta(g["meta"])
    sd = snac_synthetic_state_dict(cfg, seed=31)
    assert any(k.endswith("parametrizations.weight.original1") for k in sd)
    pcm = synthetic_pcm(2, 1, 2500, cfg.sampling_rate, seed=9)
    ref = c_oracle.RefSNAC(cfg, save_blob(sd))
    rz, rzq, rcodes = ref.encode(pcm)
    p1, p2 = tmp_path / "snac.safetensors", tmp_path / "pytorch_model.bin"
    save_file({k: np.ascontiguousarray(v) for k, v in sd.items()}, str(p1))
    torch.save({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, str(p2))
    for p in (p1, p2):
        blob, meta = checkpoint.convert_checkpoint(str(p), "snac")
        assert meta is None
        m = SNAC(cfg)
        m.load_blob(blob)
        codes = m.encode(pcm)
        for a, b in zip(codes, rcodes):
            assert np.array_equal(a, b)
        noise = snac_noise(cfg, 2, rz.shape[-1], seed=4)
        assert np.array_equal(m.decode(codes, noise), ref.decode(rcodes, noise))
        q = tmp_path / (p.name + ".ncwb")                                  # LoadWeights(path) through nc_codec_load_weights
        q.write_bytes(blob)
        m2 = SNAC(cfg)
        m2.load_weights(str(q))
        for a, b in zip(m2.encode(pcm), rcodes):
            assert np.array_equal(a, b)
        m.dispose()
        m2.dispose()
    with pytest.raises(FileNotFoundError):
        checkpoint.convert_checkpoint(str(tmp_path / "nope.safetensors"), "snac")


@pytest.mark.parametrize("name", ["encodec_small48", "encodec_small24"])
def test_encodec_checkpoint_files_run_on_the_engine(tmp_path, name):
    """Encodec checkpoints are keyed like the reference's modules (SConv1d.cs:110-128; LoadWeights: Models/Encodec.cs:348-385) and carry the
    EuclideanCodebook training buffers (cluster_size / embed_avg / inited, EuclideanCodebook.cs:60-66), which the converter drops."""
    import torch
    from safetensors.numpy import save_file
    from conftest import encodec_cfg_from_meta
    from neuralcodecs_amd import Encodec
    from neuralcodecs_amd.weights import encodec_synthetic_state_dict
    g = load_golden(name)
    cfg = encodec_cfg_from_meta(g["meta"])
    sd = encodec_synthetic_state_dict(cfg, seed=33)
    full = dict(sd)
    rng = np.random.default_rng(5)
    for k in [k for k in sd if k.endswith("codebook.embed")]:            # what a real checkpoint holds beside the embedding table
        base = k[: -len("embed")]
        full[base + "embed_avg"] = rng.standard_normal(sd[k].shape).astype(np.float32)
        full[base + "cluster_size"] = rng.uniform(0, 5, sd[k].shape[0]).astype(np.float32)
        full[base + "inited"] = np.ones(1, np.float32)
    pcm = g["pcm"]
    ref = c_oracle.RefEncodec(cfg, save_blob(sd))
    rframes = ref.encode(pcm)
    raudio = ref.decode(rframes)
    p1, p2 = tmp_path / "model.safetensors", tmp_path / "encodec.th"
    save_file({k: np.ascontiguousarray(v) for k, v in full.items()}, str(p1))
    torch.save({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in full.items()}, str(p2))
    for p in (p1, p2):
        blob, _ = checkpoint.convert_checkpoint(str(p), "encodec")
        m = Encodec(cfg)
        m.load_blob(blob)
        frames = m.encode(pcm)
        assert len(frames) == len(rframes)
        for f, (rc, rs) in zip(frames, rframes):
            assert np.array_equal(f.codes, rc)
            if rs is not None:
                assert np.array_equal(f.scale, rs)
        assert np.array_equal(m.decode(frames, pcm.shape[-1]), raudio)
        m.dispose()
