#!/usr/bin/env python3
"""Convert a DAC / SNAC / Encodec checkpoint (HF safetensors, Descript .pth, torch state dict) to the engine's weight blob."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralcodecs_amd.checkpoint import convert_checkpoint  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--codec", choices=("dac", "snac", "encodec"), default="dac")
    ap.add_argument("src")
    ap.add_argument("dst")
    a = ap.parse_args()
    blob, cfg = convert_checkpoint(a.src, a.codec)
    open(a.dst, "wb").write(blob)
    print(f"wrote {a.dst}: {len(blob)} bytes" + (f"; config from metadata: {cfg}" if cfg else ""))
