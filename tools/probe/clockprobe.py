#!/usr/bin/env python3
"""Read the shader clock while the DAC encode+decode step runs (tools/probe/clockprobe.hip on a side stream)."""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from neuralcodecs_amd import DAC

here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libclockprobe.so"))
lib.clock_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
dev = torch.device("cuda:0")
out = torch.zeros(2, dtype=torch.int64, device=dev)
side = torch.cuda.Stream()

def probe(ms):
    lib.clock_probe_launch(out.data_ptr(), int(ms * 1e5), side.cuda_stream)

probe(20); side.synchronize()
c, w = out.tolist(); print(f"idle      : {c / w * 100:.0f} MHz (shader ticks per 100 MHz wall tick x100) over {w/1e5:.1f} ms")

from neuralcodecs_amd.config import DACConfig
from neuralcodecs_amd.weights import save_blob, dac_synthetic_state_dict, synthetic_pcm
cfg = DACConfig.dac_44khz()
m = DAC(cfg, device_index=0)
m.load_blob(save_blob(dac_synthetic_state_dict(cfg, seed=42)))
pcm = torch.from_numpy(synthetic_pcm(32, 1, cfg.sample_rate, cfg.sample_rate, seed=1234)).to(dev)
def step():
    z, codes, lat, _, _ = m.encode(pcm)
    return m.decode(z)
for _ in range(3): step()
torch.cuda.synchronize()
for ms in (40, 200, 200):
    probe(ms)
    t0 = time.time(); n = 0
    while time.time() - t0 < ms / 1000 + 0.05:
        step(); n += 1
    torch.cuda.synchronize()
    c, w = out.tolist()
    print(f"under load: {c / w * 100:.0f} MHz over {w/1e5:.1f} ms ({n} steps launched)")
