#!/usr/bin/env python3
"""GPU == C oracle over several weight seeds and clip seeds at the BASELINE model sizes (runs ON THE GPU BOX; test infrastructure).

    python tools/gpu_parity_sweep.py [--out gpurun_out/gpu_parity_sweep.json]

For DAC 44.1 kHz (1 s clips), Encodec 48 kHz stereo (2 s clips) and SNAC 44.1 kHz (5 s clips): every combination of
`--weight-seeds` x `--pcm-seeds`, a batch of clips per combination through the C ABI (encode + decode) and clip by clip through the
C oracle; counts code mismatches, latent / PCM differences (max abs, must be 0: the engine's arithmetic contract is bit-exactness) and
records a SHA-256 of the codes so that two runs can be compared without the tensors.  The tests under tests/ check sampled clips of one
seed; this sweep is the many-clips-many-seeds statement behind "bit-exact" (VERDICT r2 item 1b asks for statistics, not one clip).
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import load_golden, dac_cfg_from_meta, encodec_cfg_from_meta, snac_cfg_from_meta  # noqa: E402
from functools import partial  # noqa: E402
from neuralcodecs_amd import DAC, SNAC, Encodec  # noqa: E402
from neuralcodecs_amd.config import DACConfig, EncodecConfig, SNACConfig  # noqa: E402
from neuralcodecs_amd.weights import (dac_synthetic_state_dict, encodec_synthetic_state_dict, save_blob, snac_noise,  # noqa: E402
                                      snac_synthetic_state_dict, synthetic_pcm)
from oracle import c_oracle  # noqa: E402


def _sha(arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


def _maxabs(a, b):
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))) if a.size else 0.0


def sweep_dac(wseeds, pseeds, clips, preset=None):
    cfg = getattr(DACConfig, preset)() if preset else dac_cfg_from_meta(load_golden("dac44k_b1")["meta"])
    rows = []
    for ws in wseeds:
        blob = save_blob(dac_synthetic_state_dict(cfg, seed=ws))
        m = DAC(cfg); m.load_blob(blob)
        ref = c_oracle.RefDAC(cfg, blob)
        for ps in pseeds:
            pcm = synthetic_pcm(clips, 1, cfg.sample_rate, cfg.sample_rate, seed=ps)
            z, codes, lat, _, _ = m.encode(pcm)
            audio = m.decode(z)
            rz, rcodes, rlat, _ = ref.encode(pcm)
            raudio = ref.decode(rz)
            rows.append({"weight_seed": ws, "pcm_seed": ps, "clips": clips, "codes": int(codes.size),
                         "code_mismatches": int(np.count_nonzero(codes != rcodes)), "z_max_abs": _maxabs(z, rz),
                         "latents_max_abs": _maxabs(lat, rlat), "pcm_max_abs": _maxabs(audio, raudio), "codes_sha256_16": _sha([codes])})
        m.dispose()
    return rows


def sweep_encodec(wseeds, pseeds, clips, preset=None):
    cfg = getattr(EncodecConfig, preset)() if preset else encodec_cfg_from_meta(load_golden("encodec48k_b1")["meta"])
    T = 2 * cfg.sampling_rate
    rows = []
    for ws in wseeds:
        blob = save_blob(encodec_synthetic_state_dict(cfg, seed=ws))
        m = Encodec(cfg); m.load_blob(blob)
        ref = c_oracle.RefEncodec(cfg, blob)
        for ps in pseeds:
            pcm = synthetic_pcm(clips, cfg.channels, T, cfg.sampling_rate, seed=ps)
            frames = m.encode(pcm)
            audio = m.decode(frames, T)
            rfr = ref.encode(pcm)
            raudio = ref.decode(rfr)
            rows.append({"weight_seed": ws, "pcm_seed": ps, "clips": clips, "codes": int(sum(f.codes.size for f in frames)),
                         "code_mismatches": int(sum(np.count_nonzero(f.codes != r[0]) for f, r in zip(frames, rfr))),
                         "scale_max_abs": max((_maxabs(f.scale, r[1]) if f.scale is not None else 0.0) for f, r in zip(frames, rfr)),   # (24 kHz: no scale)
                         "pcm_max_abs": _maxabs(audio, raudio), "codes_sha256_16": _sha([f.codes for f in frames])})
        m.dispose()
    return rows


def sweep_snac(wseeds, pseeds, clips, preset=None, seconds=5):
    cfg = getattr(SNACConfig, preset)() if preset else snac_cfg_from_meta(load_golden("snac44k_short")["meta"])
    T = seconds * cfg.sampling_rate
    rows = []
    for ws in wseeds:
        blob = save_blob(snac_synthetic_state_dict(cfg, seed=ws))
        m = SNAC(cfg); m.load_blob(blob)
        ref = c_oracle.RefSNAC(cfg, blob)
        for ps in pseeds:
            pcm = synthetic_pcm(clips, 1, T, cfg.sampling_rate, seed=ps)
            codes = m.encode(pcm)
            nz = snac_noise(cfg, clips, codes[-1].shape[1], seed=ps + 100)
            audio = m.decode(codes, nz)
            _, _, rcodes = ref.encode(pcm)
            raudio = ref.decode(rcodes, nz)
            rows.append({"weight_seed": ws, "pcm_seed": ps, "clips": clips, "codes": int(sum(c.size for c in codes)),
                         "code_mismatches": int(sum(np.count_nonzero(a != b) for a, b in zip(codes, rcodes))),
                         "pcm_max_abs": _maxabs(audio, raudio), "codes_sha256_16": _sha(codes)})
        m.dispose()
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "gpu_parity_sweep.json"))
    ap.add_argument("--weight-seeds", default="42,7")
    ap.add_argument("--pcm-seeds", default="1,2")
    ap.add_argument("--dac-clips", type=int, default=8)
    ap.add_argument("--encodec-clips", type=int, default=4)
    ap.add_argument("--snac-clips", type=int, default=2)
    ap.add_argument("--presets", action="store_true", help="also every other preset the reference ships, at full width (round 5)")
    ap.add_argument("--preset-clips", type=int, default=4)
    a = ap.parse_args()
    ws = [int(v) for v in a.weight_seeds.split(",")]
    ps = [int(v) for v in a.pcm_seeds.split(",")]
    out = {"what": "HIP engine (C ABI) vs C oracle, every clip of every (weight seed, clip seed) combination; mismatches and max-abs must be 0"}
    plan = [("dac44k_1s", sweep_dac, a.dac_clips), ("encodec48k_2s", sweep_encodec, a.encodec_clips), ("snac44k_5s", sweep_snac, a.snac_clips)]
    if a.presets:   # Config/DAC/DACConfig.cs:103-135, Config/SNAC/SNACConfig.cs, Config/Encodec/EncodecConfig.cs
        n = a.preset_clips
        plan += [("dac44k_16kbps_1s", partial(sweep_dac, preset="dac_44khz_16kbps"), n), ("dac24k_1s", partial(sweep_dac, preset="dac_24khz"), n),
                 ("dac16k_1s", partial(sweep_dac, preset="dac_16khz"), n), ("snac32k_2s", partial(sweep_snac, preset="snac_32khz", seconds=2), n),
                 ("snac24k_2s", partial(sweep_snac, preset="snac_24khz", seconds=2), n), ("encodec24k_2s", partial(sweep_encodec, preset="encodec_24khz"), n)]
    for name, fn, n in plan:
        t0 = time.time()
        rows = fn(ws, ps, n)
        tot = {"clips": sum(r["clips"] for r in rows), "codes": sum(r["codes"] for r in rows),
               "code_mismatches": sum(r["code_mismatches"] for r in rows), "pcm_max_abs": max(r["pcm_max_abs"] for r in rows),
               "seconds": round(time.time() - t0, 1)}
        out[name] = {"total": tot, "runs": rows}
        print(name, json.dumps(tot), flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    bad = [k for k, v in out.items() if isinstance(v, dict) and (v["total"]["code_mismatches"] or v["total"]["pcm_max_abs"] != 0.0)]
    if bad:
        print("NOT bit-exact:", bad)
        sys.exit(1)


if __name__ == "__main__":
    main()
