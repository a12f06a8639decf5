#!/usr/bin/env python3
"""Encode+decode throughput of the other BASELINE configs on one GPU (informational; the headline is bench.py):
   C3  Encodec 48 kHz stereo 12 kbps, batch 16 x 2 s        C5/8  SNAC 44.1 kHz + LocalMHA, 8 x 5 s (one GPU's share of C5)
   C1  SNAC 24 kHz mono, 1 x 1 s
Device-resident inputs, HIP-event timing around `steps` encode+decode rounds.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from neuralcodecs_amd import SNAC, Encodec  # noqa: E402
from neuralcodecs_amd.config import EncodecConfig, SNACConfig  # noqa: E402
from neuralcodecs_amd.weights import (encodec_synthetic_state_dict, save_blob, snac_noise, snac_synthetic_state_dict,  # noqa: E402
                                      synthetic_pcm)


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def class_table(m, fn, steps):
    """per-class HIP-event times of `steps` rounds (the engine's profiler: event pairs around every launch)"""
    m.profile_enable(True)
    m.profile_reset()
    timed(fn, steps, 0)
    prof = m.profile_read()
    m.profile_enable(False)
    return {n: {"ms": round(v["ms"] / steps, 4), "launches": round(v["launches"] / steps, 1)} for n, v in prof.items() if v["launches"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--classes", action="store_true", help="also print the per-class kernel times")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    out = {}
    if not a.only or "snac" in a.only:
        for name, cfg, B, secs in (("snac24k_c1", SNACConfig.snac_24khz(), 1, 1.0), ("snac44k_c5_share", SNACConfig.snac_44khz(), 8, 5.0)):
            if a.only.startswith("snac") and len(a.only) > 4 and a.only[4:] not in name:
                continue
            m = SNAC(cfg)
            m.load_blob(save_blob(snac_synthetic_state_dict(cfg, seed=42)))
            T = int(secs * cfg.sampling_rate)
            x = torch.from_numpy(synthetic_pcm(B, 1, T, cfg.sampling_rate, seed=1)).to(dev)
            frames = m.query(T)[1]
            nz = m.flat_noise(snac_noise(cfg, B, frames, seed=3), dev)   # the ABI layout: views of one device buffer, prepared once
            dt = timed(lambda: m.decode(m.encode(x), nz), a.steps, a.warmup)
            out[name] = {"ms": round(dt * 1e3, 3), "x_realtime": round(B * secs / dt, 1), "B": B, "seconds": secs}
            if a.classes:
                out[name]["classes"] = class_table(m, lambda: m.decode(m.encode(x), nz), a.steps)
            m.dispose()
    if not a.only or "encodec" in a.only:
        for name, cfg, B, secs in (("encodec48k_c3", EncodecConfig.encodec_48khz(), 16, 2.0), ("encodec24k", EncodecConfig.encodec_24khz(), 16, 2.0)):
            if a.only.startswith("encodec") and len(a.only) > 7 and a.only[7:] not in name:
                continue
            m = Encodec(cfg)
            m.load_blob(save_blob(encodec_synthetic_state_dict(cfg, seed=42)))
            T = int(secs * cfg.sampling_rate)
            x = torch.from_numpy(synthetic_pcm(B, cfg.channels, T, cfg.sampling_rate, seed=1)).to(dev)
            dt = timed(lambda: m.decode(m.encode(x), T), a.steps, a.warmup)
            out[name] = {"ms": round(dt * 1e3, 3), "x_realtime": round(B * secs / dt, 1), "B": B, "seconds": secs}
            if a.classes:
                out[name]["classes"] = class_table(m, lambda: m.decode(m.encode(x), T), a.steps)
            m.dispose()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
