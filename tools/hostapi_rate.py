#!/usr/bin/env python3
"""PCIe-inclusive rate of the headline workload through the HOST-pointer API (numpy in, numpy out): what a float[]-level caller sees.
Not the headline value (bench.py times device-resident inputs); quoted in DESIGN.md section 8."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralcodecs_amd import DAC, DACConfig
from neuralcodecs_amd.weights import dac_synthetic_state_dict, save_blob, synthetic_pcm

cfg = DACConfig.dac_44khz()
m = DAC(cfg)
m.load_blob(save_blob(dac_synthetic_state_dict(cfg, seed=42)))
B, T = 32, cfg.sample_rate
pcm = synthetic_pcm(B, 1, T, cfg.sample_rate, seed=1234)
def step():
    z, codes, lat, _, _ = m.encode(pcm)
    return m.decode(z)
for _ in range(3): step()
t0 = time.perf_counter(); n = 10
for _ in range(n): step()
dt = (time.perf_counter() - t0) / n
print(f"host-pointer API: {dt*1e3:.2f} ms per step, {B/dt:.1f} x real-time (PCIe copies and synchronisation included)")
