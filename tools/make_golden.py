"""Generate tests/golden/*.npz with the PyTorch-CPU restatement of the reference graph (oracle/torch_ref).

Run in the build container (CPU):  python tools/make_golden.py
The fixtures are DATA (inputs + expected outputs); weights are regenerated from the seed by
neuralcodecs_amd.weights (integer counter-based generator, bit-reproducible), so only the small
reduced-width case stores nothing but I/O, and the full-size DAC-44.1kHz case stores codes, the
top-2 distance gap of every argmin (for near-tie audits) and decimated slices of z / PCM.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neuralcodecs_amd.config import DACConfig, EncodecConfig, SNACConfig  # noqa: E402
from neuralcodecs_amd.weights import (dac_synthetic_state_dict, encodec_synthetic_state_dict, snac_noise,  # noqa: E402
                                      snac_synthetic_state_dict, synthetic_pcm)
from oracle.torch_ref.dac import TorchDAC  # noqa: E402
from oracle.torch_ref.encodec import TorchEncodec  # noqa: E402
from oracle.torch_ref.snac import TorchSNAC  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)

SMALL = dict(sample_rate=16000, encoder_dim=8, encoder_rates=(2, 4, 5, 8), decoder_dim=48, decoder_rates=(8, 5, 4, 2),
             n_codebooks=4, codebook_size=64, codebook_dim=8)


def top2_gap(dists):
    g = []
    for d in dists:
        v, _ = torch.topk(d, 2, dim=1, largest=False)
        g.append((v[:, 1] - v[:, 0]).numpy())
    return np.stack(g, 0)  # [nq, B*T']


def dac_case(name, cfg_kw, B, T, wseed, pseed, full, ties=False):
    cfg = DACConfig(**cfg_kw)
    sd = dac_synthetic_state_dict(cfg, seed=wseed)
    if ties:            # duplicated codebook rows + dead codes: exact ties in every frame (neuralcodecs_amd.weights.tie_codebooks)
        from neuralcodecs_amd.weights import tie_codebooks
        tie_codebooks(sd)
    m = TorchDAC(cfg, sd)
    pcm = synthetic_pcm(B, 1, T, cfg.sample_rate, seed=pseed)
    zq, codes, lat, dists = m.encode(pcm, want_dist=True)
    audio = m.decode(zq)
    z_from = m.from_codes(codes)
    meta = dict(cfg=cfg_kw, B=B, T=T, weight_seed=wseed, pcm_seed=pseed)
    if ties:
        meta["ties"] = True
    out = dict(meta=json.dumps(meta), codes=codes.numpy().astype(np.int16), gap=top2_gap(dists).astype(np.float32))
    if full:
        out.update(zq_slice=zq.numpy()[:, ::16, :], audio_slice=audio.numpy()[:, :, ::29],
                   zq_sum=np.float64(zq.double().sum().item()), audio_abs_sum=np.float64(audio.double().abs().sum().item()),
                   from_codes_slice=z_from.numpy()[:, ::16, :])
    else:
        out.update(pcm=pcm, zq=zq.numpy(), latents=lat.numpy(), audio=audio.numpy(), from_codes=z_from.numpy())
        # n_quantizers = 2 overload (ResidualVectorQuantizer.cs:105-206 eval branch)
        zq2, codes2, lat2 = m.encode(pcm, n_quantizers=2)
        out.update(codes_nq2=codes2.numpy().astype(np.int16), zq_nq2=zq2.numpy())
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k != "meta"})


SNAC_SMALL = dict(sampling_rate=16000, encoder_dim=8, encoder_rates=(2, 3, 4, 4), decoder_dim=64, decoder_rates=(4, 4, 3, 2),
                  attn_window_size=None, codebook_size=256, vq_strides=(4, 2, 1))
SNAC_SMALL_ATTN = dict(sampling_rate=16000, encoder_dim=16, encoder_rates=(2, 3, 4, 4), decoder_dim=256, decoder_rates=(4, 4, 3, 2),
                       attn_window_size=8, codebook_size=256, vq_strides=(4, 2, 1))


def snac_case(name, cfg_kw, B, T, wseed, pseed, nseed, full, tensor_overload=False):
    cfg = SNACConfig(**cfg_kw)
    m = TorchSNAC(cfg, snac_synthetic_state_dict(cfg, seed=wseed))
    pcm = synthetic_pcm(B, 1, T, cfg.sampling_rate, seed=pseed)
    z, zq, codes, dists = (m.encode_tensor if tensor_overload else m.encode)(pcm, want_dist=True)
    noises = snac_noise(cfg, B, z.shape[-1], seed=nseed)
    audio = m.decode(codes, noises)
    meta = dict(cfg=cfg_kw, B=B, T=T, weight_seed=wseed, pcm_seed=pseed, noise_seed=nseed)
    out = dict(meta=json.dumps(meta))
    for i, (c, d) in enumerate(zip(codes, dists)):
        out[f"codes{i}"] = c.numpy().astype(np.int16)
        v, _ = torch.topk(d, 2, dim=1, largest=False)
        out[f"gap{i}"] = (v[:, 1] - v[:, 0]).numpy().astype(np.float32)
    if full:
        out.update(zq_slice=zq.numpy()[:, ::16, :], audio_slice=audio.numpy()[:, :, ::17])
    else:
        out.update(pcm=pcm, z=z.numpy(), zq=zq.numpy(), audio=audio.numpy())
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k != "meta"})


ENC_SMALL48 = dict(sampling_rate=16000, channels=2, dimension=32, norm="time_group_norm", causal=False, normalize=True,
                   segment_seconds=0.25, target_bandwidths=(3.0, 6.0, 12.0), bandwidth=6.0, codebook_size=64, n_filters=4,
                   ratios=(4, 3, 2, 2))
ENC_SMALL24 = dict(sampling_rate=16000, channels=1, dimension=32, norm="weight_norm", causal=True, normalize=False,
                   target_bandwidths=(1.5, 3.0, 6.0), bandwidth=3.0, codebook_size=64, n_filters=4, ratios=(4, 3, 2, 2))


def encodec_case(name, cfg_kw, B, T, wseed, pseed, full):
    cfg = EncodecConfig(**cfg_kw)
    m = TorchEncodec(cfg, encodec_synthetic_state_dict(cfg, seed=wseed))
    pcm = synthetic_pcm(B, cfg.channels, T, cfg.sampling_rate, seed=pseed)
    frames = m.encode(pcm, want_dist=True)
    audio = m.decode(frames)
    meta = dict(cfg=cfg_kw, B=B, T=T, weight_seed=wseed, pcm_seed=pseed, n_frames=len(frames), n_q=m.n_q())
    out = dict(meta=json.dumps(meta))
    for i, (codes, scale, emb, dists) in enumerate(frames):
        out[f"codes{i}"] = codes.numpy().astype(np.int16)
        out[f"gap{i}"] = top2_gap(dists).astype(np.float32)                      # [n_q, B*T']
        if scale is not None:
            out[f"scale{i}"] = scale.numpy()
        out[f"emb{i}"] = emb.numpy()[:, ::8, :] if full else emb.numpy()
    if full:
        out.update(audio_slice=audio.numpy()[:, :, ::23])
    else:
        out.update(pcm=pcm, audio=audio.numpy())
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k != "meta"})



def nonfinite_pcm(B, C, T, sr, seed):
    """Clips with one non-finite sample each: clip 0 a +inf, clip 1 a NaN, clip 2 a -inf next to a NaN; further clips stay
    finite.  The positions sit inside the clip so that frames before / after the poisoned receptive field stay finite."""
    pcm = synthetic_pcm(B, C, T, sr, seed=seed)
    pcm[0, 0, T // 3] = np.inf
    if B > 1:
        pcm[1, C - 1, T // 2] = np.nan
    if B > 2:
        pcm[2, 0, T // 4] = -np.inf
        pcm[2, 0, T // 4 + 1] = np.nan
    return pcm


def nonfinite_cases():
    """Goldens for non-finite inputs: what ATen does with them is the reference's behaviour (argmin returns the first NaN of a row:
    Modules/DAC/VectorQuantizer.cs:121, Modules/SNAC/VectorQuantizer.cs:137, Modules/Encodec/EuclideanCodebook.cs:181)."""
    # (1) one quantizer stage on rows that are PARTLY NaN: a latent with +-inf components makes dist = |e|^2 + |c|^2 - 2 e.c NaN exactly for the
    # codes whose cross term is +inf (inf - inf) and +inf for the others; ATen returns the first NaN index, an all-inf row gives 0.
    g = np.random.default_rng(606)
    out = {}
    for name, N, D in (("dac", 1024, 8), ("snac", 4096, 8), ("encodec", 1024, 128)):
        T = 96
        ze = g.standard_normal((2, D, T)).astype(np.float32)
        cb = g.standard_normal((N, D)).astype(np.float32)
        for t in range(0, T, 3):
            ze[0, (t // 3) % D, t] = np.inf if (t // 3) % 2 == 0 else -np.inf          # partly-NaN rows
        for t in range(1, T, 12):
            ze[1, :, t] = np.nan                                                        # all-NaN rows -> 0
        ze[1, 0, 2] = np.inf; ze[1, 1, 2] = np.inf                                      # two infinities: more NaNs (inf - inf inside e.c)
        cb[:5, :] = np.abs(cb[:5, :]) * np.sign(cb[5, :])[None, :]                      # a block of codes sharing every sign: runs of equal class
        enc = torch.from_numpy(ze).transpose(1, 2).reshape(-1, D).contiguous()
        c = torch.from_numpy(cb)
        if name == "encodec":    # EuclideanCodebook.cs:155-182
            dist = enc.pow(2).sum(1, keepdim=True).add(c.pow(2).sum(1, keepdim=True).t()).add(-2 * enc.matmul(c.t()))
            idx = dist.argmin(dim=-1)
        else:                    # DAC / SNAC VectorQuantizer.DecodeLatents
            dist = enc.pow(2).sum(1, keepdim=True) + c.pow(2).sum(1, keepdim=True).t() - torch.einsum("bd,nd->bn", enc, c).mul_(2.0)
            idx = dist.argmin(1)
        nan_rows = torch.isnan(dist).any(1)
        part = nan_rows & ~torch.isnan(dist).all(1)
        assert int(part.sum()) >= 20, name
        first_nan = torch.isnan(dist).float().argmax(1)
        assert bool((idx[nan_rows] == first_nan[nan_rows]).all())                       # ATen: the FIRST NaN of the row
        out[f"{name}_ze"] = ze
        out[f"{name}_cb"] = cb
        out[f"{name}_idx"] = idx.reshape(2, T).numpy().astype(np.int32)
        out[f"{name}_partly_nan_rows"] = np.int32(int(part.sum()))
    np.savez_compressed(os.path.join(OUT, "vq_nonfinite.npz"), meta=json.dumps(dict(seed=606)), **out)
    print("vq_nonfinite", {k: (v.shape if hasattr(v, "shape") and v.shape else v) for k, v in out.items()})

    # (2) whole codecs, reduced width, on clips with one inf / NaN sample
    cfg = DACConfig(**SMALL)
    m = TorchDAC(cfg, dac_synthetic_state_dict(cfg, seed=7))
    pcm = nonfinite_pcm(4, 1, 16000, cfg.sample_rate, seed=21)
    zq, codes, lat = m.encode(pcm)
    audio = m.decode(zq)
    meta = dict(cfg=SMALL, B=4, T=16000, weight_seed=7, pcm_seed=21)
    np.savez_compressed(os.path.join(OUT, "dac_small_nonfinite.npz"), meta=json.dumps(meta), pcm=pcm, codes=codes.numpy().astype(np.int16),
                        zq=zq.numpy(), audio=audio.numpy())
    print("dac_small_nonfinite: NaN frames per clip", torch.isnan(zq).any(1).sum(1).tolist(), "of", zq.shape[-1],
          "| NaN samples per clip", torch.isnan(audio).any(1).sum(1).tolist(), "of", audio.shape[-1])

    cfg = SNACConfig(**SNAC_SMALL)
    m = TorchSNAC(cfg, snac_synthetic_state_dict(cfg, seed=5))
    pcm = nonfinite_pcm(4, 1, 12000, cfg.sampling_rate, seed=22)
    z, zq, codes = m.encode(pcm)[:3]
    noises = snac_noise(cfg, 4, z.shape[-1], seed=98)
    audio = m.decode(codes, noises)
    meta = dict(cfg=SNAC_SMALL, B=4, T=12000, weight_seed=5, pcm_seed=22, noise_seed=98)
    out = dict(meta=json.dumps(meta), pcm=pcm, audio=audio.numpy())
    for i, c in enumerate(codes):
        out[f"codes{i}"] = c.numpy().astype(np.int16)
    np.savez_compressed(os.path.join(OUT, "snac_small_nonfinite.npz"), **out)
    print("snac_small_nonfinite: NaN frames per clip", torch.isnan(z).any(1).sum(1).tolist(), "of", z.shape[-1],
          "| NaN samples per clip", torch.isnan(audio).any(1).sum(1).tolist(), "of", audio.shape[-1])

    for name, kw, T in (("encodec_small48_nonfinite", ENC_SMALL48, 12100), ("encodec_small24_nonfinite", ENC_SMALL24, 12001)):
        cfg = EncodecConfig(**kw)
        m = TorchEncodec(cfg, encodec_synthetic_state_dict(cfg, seed=7))
        pcm = nonfinite_pcm(4, cfg.channels, T, cfg.sampling_rate, seed=23)
        frames = m.encode(pcm)
        audio = m.decode(frames)
        meta = dict(cfg=kw, B=4, T=T, weight_seed=7, pcm_seed=23, n_frames=len(frames), n_q=m.n_q())
        out = dict(meta=json.dumps(meta), pcm=pcm, audio=audio.numpy())
        for i, fr in enumerate(frames):
            out[f"codes{i}"] = fr[0].numpy().astype(np.int16)
            if fr[1] is not None:
                out[f"scale{i}"] = fr[1].numpy()
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        print(name, "frames", len(frames), "| NaN samples per clip", torch.isnan(audio).any(1).sum(1).tolist(), "of", audio.shape[-1])


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "--round2":   # fixtures added in round 2 (the round-1 files stay byte-identical)
        # SNAC.Encode(Tensor) as written (D7): 24 kHz model on an un-padded, non-multiple length (22628 samples -> 44 frames, not 48)
        snac_case("snac24k_tensor_b1", dict(), 1, 22628, 42, 1234, 77, True, tensor_overload=True)
        snac_case("snac_small_tensor", SNAC_SMALL, 2, 3100, 5, 3, 99, False, tensor_overload=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "--round6":   # non-finite inputs (the older files stay byte-identical)
        nonfinite_cases()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "--round5":   # fixtures added in round 5 (the older files stay byte-identical)
        # full-size DAC 24 kHz (Config/DAC/DACConfig.cs:115-124: 32 codebooks, rates 2-4-5-8 -- stride 5 at full width), one 1 s clip
        dac_case("dac24k_b1", dict(sample_rate=24000, n_codebooks=32, encoder_rates=(2, 4, 5, 8), decoder_rates=(8, 5, 4, 2)), 1, 24000, 42, 1234, True)
        # adversarial quantizer at full size: duplicated codebook rows (an exact tie in EVERY frame) and dead codes -> ATen's first index
        dac_case("dac44k_ties_b1", dict(), 1, 44100, 42, 1234, True, ties=True)
        sys.exit(0)
    # reduced width, ragged length (not a hop multiple), odd stride 5 (DAC-16/24 kHz presets use it)
    dac_case("dac_small", SMALL, 2, 2000, 7, 11, False)
    # full-size DAC 44.1 kHz 8 kbps, one 1 s clip (BASELINE config C2 at B=1)
    dac_case("dac44k_b1", dict(), 1, 44100, 42, 1234, True)
    # SNAC: reduced width without / with local attention (odd stride 3 exercises output_padding), ragged lengths
    snac_case("snac_small", SNAC_SMALL, 2, 3001, 5, 3, 99, False)
    snac_case("snac_small_attn", SNAC_SMALL_ATTN, 2, 2500, 6, 4, 98, False)
    # full-size SNAC 24 kHz, one 1 s clip (BASELINE config C1)
    snac_case("snac24k_b1", dict(), 1, 24000, 42, 1234, 77, True)
    snac_case("snac44k_short", dict(sampling_rate=44100, encoder_dim=64, encoder_rates=(2, 3, 8, 8), decoder_dim=1536,
                                    decoder_rates=(8, 8, 3, 2), attn_window_size=32, vq_strides=(8, 4, 2, 1)), 1, 20000, 42, 1234, 55, True)
    # Encodec: reduced width, stereo + GroupNorm + segments with a short tail (small-input reflect path, D9) / causal weight-norm mono
    encodec_case("encodec_small48", ENC_SMALL48, 2, 8100, 7, 3, False)
    encodec_case("encodec_small24", ENC_SMALL24, 2, 3001, 8, 4, False)
    # full size: Encodec 48 kHz stereo 12 kbps, one 2 s clip (BASELINE config C3 at B=1) and Encodec 24 kHz mono 6 kbps, 1 s
    encodec_case("encodec48k_b1", dict(sampling_rate=48000, channels=2, norm="time_group_norm", causal=False, normalize=True,
                                       segment_seconds=1.0, target_bandwidths=(3.0, 6.0, 12.0, 24.0), bandwidth=12.0), 1, 96000, 42, 1234, True)
    encodec_case("encodec24k_b1", dict(), 1, 24000, 42, 1234, True)
