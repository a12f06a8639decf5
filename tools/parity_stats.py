#!/usr/bin/env python3
"""Multi-clip, multi-seed parity statistics: C oracle (canonical arithmetic, what the GPU is bit-exact against) vs the PyTorch-CPU
restatement of the reference graph (oracle/torch_ref, ATen operators -- the operator family TorchSharp dispatches to), at the FULL-SIZE
configurations the bench runs.  Build container only (oracle/torch_ref never travels).

For every clip: number of RVQ codes, code flips (frames whose code differs at ANY stage / level), the smallest top-2 distance gap among
the flipped frames (a flip must be a near-tie of the ATen argmin to be explainable by summation order), max-abs difference of the
latents and of the decoded PCM, and the SHA-256 of the C oracle's code tensor (tests/test_parity_stats_cpu.py recomputes a subset of
these hashes, so the record cannot drift from the oracle).  Only counts and hashes are stored, no tensors.

    python tools/parity_stats.py [--codec dac|encodec|snac|presets|all] [--dac-clips 32] [--encodec-clips 16] [--snac-clips 8] [--out tests/golden/parity_stats.json]
"""
import argparse
import hashlib
import json
import os
import sys
import time

os.environ.setdefault("OMP_WAIT_POLICY", "passive")
os.environ.setdefault("GOMP_SPINCOUNT", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def frames_flipped(codes, ref):
    """codes, ref: [B, nq, T] -> frames with a mismatch at any stage, and (stage, flat frame) of each frame's FIRST mismatch"""
    neq = codes != ref
    bad = neq.any(axis=1)
    first = []
    for b, t in zip(*np.nonzero(bad)):
        first.append((int(np.nonzero(neq[b, :, t])[0][0]), int(b) * codes.shape[2] + int(t)))
    return int(bad.sum()), first


def dac_clip(wseed, pseed, cache, preset="dac_44khz"):
    import torch
    from neuralcodecs_amd.config import DACConfig
    from neuralcodecs_amd.weights import dac_synthetic_state_dict, save_blob, synthetic_pcm
    from oracle import c_oracle
    from oracle.torch_ref.dac import TorchDAC
    cfg = getattr(DACConfig, preset)()          # every preset the reference ships: Config/DAC/DACConfig.cs:103-135
    if (preset, wseed) not in cache:
        sd = dac_synthetic_state_dict(cfg, seed=wseed)
        cache.clear()
        cache[(preset, wseed)] = (TorchDAC(cfg, sd), c_oracle.RefDAC(cfg, save_blob(sd)))
    tm, ref = cache[(preset, wseed)]
    pcm = synthetic_pcm(1, 1, cfg.sample_rate, cfg.sample_rate, seed=pseed)
    zq, codes, lat, dists = tm.encode(pcm, want_dist=True)
    audio = tm.decode(zq)
    rz, rcodes, rlat, _ = ref.encode(pcm)
    raudio = ref.decode(rz)
    codes = codes.numpy()
    nflip, first = frames_flipped(rcodes, codes)
    gaps = []
    for st, fr in first:
        v, _ = torch.topk(dists[st][fr], 2, largest=False)
        gaps.append(float(v[1] - v[0]))
    e = dict(weight_seed=wseed, pcm_seed=pseed, n_codes=int(codes.size), n_frames=int(codes.shape[0] * codes.shape[2]), flipped_frames=nflip,
             flip_gaps=gaps, codes_sha256=sha(rcodes.astype(np.int64)))
    if nflip == 0:   # (after a flip the continuous outputs legitimately differ)
        e.update(latents_max_abs=float(np.abs(rlat - lat.numpy()).max()), z_max_abs=float(np.abs(rz - zq.numpy()).max()),
                 pcm_max_abs=float(np.abs(raudio - audio.numpy()).max()))
    return e


def encodec_clip(wseed, pseed, cache, preset="encodec_48khz"):
    import torch
    from neuralcodecs_amd.config import EncodecConfig
    from neuralcodecs_amd.weights import encodec_synthetic_state_dict, save_blob, synthetic_pcm
    from oracle import c_oracle
    from oracle.torch_ref.encodec import TorchEncodec
    cfg = getattr(EncodecConfig, preset)()      # Config/Encodec/EncodecConfig.cs:9-64
    if (preset, wseed) not in cache:
        sd = encodec_synthetic_state_dict(cfg, seed=wseed)
        cache.clear()
        cache[(preset, wseed)] = (TorchEncodec(cfg, sd), c_oracle.RefEncodec(cfg, save_blob(sd)))
    tm, ref = cache[(preset, wseed)]
    pcm = synthetic_pcm(1, cfg.channels, 2 * cfg.sampling_rate, cfg.sampling_rate, seed=pseed)
    frames = tm.encode(pcm, want_dist=True)
    audio = tm.decode(frames)
    rframes = ref.encode(pcm, want_emb=True)
    raudio = ref.decode([(c, s) for c, s, _ in rframes])
    n_codes = n_frames = nflip = 0
    gaps, emb_d, h = [], 0.0, hashlib.sha256()
    for (codes, scale, emb, dists), (rc, rs, remb) in zip(frames, rframes):
        codes = codes.numpy()
        n_codes += codes.size
        n_frames += codes.shape[0] * codes.shape[2]
        nf, first = frames_flipped(rc, codes)
        nflip += nf
        for st, fr in first:
            v, _ = torch.topk(dists[st][fr], 2, largest=False)
            gaps.append(float(v[1] - v[0]))
        emb_d = max(emb_d, float(np.abs(remb - emb.numpy()).max()))
        h.update(np.ascontiguousarray(rc.astype(np.int64)).tobytes())
    e = dict(weight_seed=wseed, pcm_seed=pseed, n_codes=int(n_codes), n_frames=int(n_frames), flipped_frames=int(nflip), flip_gaps=gaps,
             codes_sha256=h.hexdigest(), latents_max_abs=emb_d)
    if nflip == 0:
        e["pcm_max_abs"] = float(np.abs(raudio - audio.numpy()).max())
    return e


def snac_clip(wseed, pseed, cache, seconds=5.0, preset="snac_44khz"):
    import torch
    from neuralcodecs_amd.config import SNACConfig
    from neuralcodecs_amd.weights import save_blob, snac_noise, snac_synthetic_state_dict, synthetic_pcm
    from oracle import c_oracle
    from oracle.torch_ref.snac import TorchSNAC
    cfg = getattr(SNACConfig, preset)()         # Config/SNAC/SNACConfig.cs
    if (preset, wseed) not in cache:
        sd = snac_synthetic_state_dict(cfg, seed=wseed)
        cache.clear()
        cache[(preset, wseed)] = (TorchSNAC(cfg, sd), c_oracle.RefSNAC(cfg, save_blob(sd)))
    tm, ref = cache[(preset, wseed)]
    pcm = synthetic_pcm(1, 1, int(seconds * cfg.sampling_rate), cfg.sampling_rate, seed=pseed)
    z, zq, codes, dists = tm.encode(pcm, want_dist=True)
    noises = snac_noise(cfg, 1, z.shape[-1], seed=pseed + 7)
    audio = tm.decode(codes, noises)
    rz, rzq, rcodes = ref.encode(pcm)
    raudio = ref.decode(rcodes, noises)
    n_codes = nflip = 0
    gaps, h = [], hashlib.sha256()
    for c, rc, d in zip(codes, rcodes, dists):
        c = c.numpy()
        n_codes += c.size
        bad = np.nonzero((c != rc).reshape(-1))[0]
        if bad.size and nflip == 0:          # the first flipped level: later (finer) levels see a different residual
            for fr in bad:
                v, _ = torch.topk(d[int(fr)], 2, largest=False)
                gaps.append(float(v[1] - v[0]))
        nflip += int(bad.size)
        h.update(np.ascontiguousarray(rc.astype(np.int64)).tobytes())
    e = dict(weight_seed=wseed, pcm_seed=pseed, seconds=seconds, n_codes=int(n_codes), n_frames=int(n_codes), flipped_frames=int(nflip),
             flip_gaps=gaps, codes_sha256=h.hexdigest(), latents_max_abs=float(np.abs(rz - z.numpy()).max()))
    if nflip == 0:
        e["pcm_max_abs"] = float(np.abs(raudio - audio.numpy()).max())
    return e


def summarize(clips):
    ok = [c for c in clips if c["flipped_frames"] == 0]
    return dict(clips=len(clips), weight_seeds=sorted({c["weight_seed"] for c in clips}), n_codes=sum(c["n_codes"] for c in clips),
                n_frames=sum(c["n_frames"] for c in clips), flipped_frames=sum(c["flipped_frames"] for c in clips),
                max_flip_gap=max([g for c in clips for g in c["flip_gaps"]] or [0.0]),
                pcm_max_abs=max([c.get("pcm_max_abs", 0.0) for c in ok] or [0.0]),
                latents_max_abs=max([c.get("latents_max_abs", 0.0) for c in ok] or [0.0]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--codec", default="all")
    ap.add_argument("--dac-clips", type=int, default=32)
    ap.add_argument("--encodec-clips", type=int, default=16)
    ap.add_argument("--snac-clips", type=int, default=8)
    ap.add_argument("--preset-clips", type=int, default=4, help="clips per OTHER preset of the reference (--codec presets | all)")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "parity_stats.json"))
    a = ap.parse_args()
    import torch
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    rec = json.load(open(a.out)) if os.path.exists(a.out) else {}
    rec["_about"] = ("C oracle vs oracle/torch_ref (ATen CPU) at the full-size bench configurations; counts and hashes only.  "
                     "Regenerate with tools/parity_stats.py (build container).")
    plan = []
    if a.codec in ("all", "dac"):
        plan.append(("dac44k", dac_clip, a.dac_clips))
    if a.codec in ("all", "encodec"):
        plan.append(("encodec48k", encodec_clip, a.encodec_clips))
    if a.codec in ("all", "snac"):
        plan.append(("snac44k", snac_clip, a.snac_clips))
    if a.codec in ("all", "presets"):
        # round 5: every other preset the reference ships, at full width -- DAC 44 kHz-16 kbps (18 codebooks, latent 128),
        # 24 kHz (32 codebooks, stride 5) and 16 kHz (Config/DAC/DACConfig.cs:103-135), SNAC 32 kHz and 24 kHz, Encodec 24 kHz (causal, weight norm)
        from functools import partial
        for name, fn in (("dac44k_16kbps", partial(dac_clip, preset="dac_44khz_16kbps")), ("dac24k", partial(dac_clip, preset="dac_24khz")),
                         ("dac16k", partial(dac_clip, preset="dac_16khz")), ("snac32k", partial(snac_clip, seconds=2.0, preset="snac_32khz")),
                         ("snac24k", partial(snac_clip, seconds=2.0, preset="snac_24khz")), ("encodec24k", partial(encodec_clip, preset="encodec_24khz"))):
            plan.append((name, fn, a.preset_clips))
    for name, fn, n in plan:
        clips, cache = [], {}
        for i in range(n):
            wseed = 42 if i < (n + 1) // 2 else 43          # two weight sets, pcm seed per clip
            pseed = 1234 + 101 * i
            t0 = time.time()
            e = fn(wseed, pseed, cache)
            clips.append(e)
            print(name, i, {k: v for k, v in e.items() if k != "codes_sha256"}, f"{time.time() - t0:.1f}s", flush=True)
            rec[name] = dict(summary=summarize(clips), clips=clips)
            json.dump(rec, open(a.out, "w"), indent=1)
    print(json.dumps({k: v["summary"] for k, v in rec.items() if isinstance(v, dict)}, indent=1))


if __name__ == "__main__":
    main()
