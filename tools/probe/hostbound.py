#!/usr/bin/env python3
"""Is the one-clip SNAC 24 kHz step (C1) bound by the GPU or by the host's launch rate?  Host enqueue time per step (no synchronisation
inside the loop, clock stopped before the final synchronise) against the synchronised step time."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuralcodecs_amd import SNAC
from neuralcodecs_amd.config import SNACConfig
from neuralcodecs_amd.weights import save_blob, snac_noise, snac_synthetic_state_dict, synthetic_pcm

dev = torch.device("cuda", 0)
cfg = SNACConfig.snac_24khz()
m = SNAC(cfg); m.load_blob(save_blob(snac_synthetic_state_dict(cfg, seed=42)))
T = cfg.sampling_rate
x = torch.from_numpy(synthetic_pcm(1, 1, T, cfg.sampling_rate, seed=1)).to(dev)
nz = [torch.from_numpy(n).to(dev) for n in snac_noise(cfg, 1, m.query(T)[1], seed=3)]
step = lambda: m.decode(m.encode(x), nz)
for _ in range(5): step()
torch.cuda.synchronize()
for n in (20, 100):
    t0 = time.perf_counter()
    for _ in range(n): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{n} steps: host enqueue {1e3*(t1-t0)/n:.3f} ms/step, with the final synchronise {1e3*(t2-t0)/n:.3f} ms/step")
t0 = time.perf_counter()
for _ in range(20):
    step(); torch.cuda.synchronize()
print(f"synchronised every step: {1e3*(time.perf_counter()-t0)/20:.3f} ms/step")
