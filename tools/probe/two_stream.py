import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd())
from neuralcodecs_amd import DAC, DACConfig
from neuralcodecs_amd.weights import dac_synthetic_state_dict, save_blob, synthetic_pcm
cfg = DACConfig.dac_44khz(); T = cfg.sample_rate; B = 32
blob = save_blob(dac_synthetic_state_dict(cfg, seed=42))
dev = torch.device("cuda", 0)
pcm = torch.from_numpy(synthetic_pcm(B, 1, T, cfg.sample_rate, seed=1234)).to(dev)
def run(nstreams, steps=10):
    models = [DAC(cfg) for _ in range(nstreams)]
    for m in models: m.load_blob(blob)
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
    parts = list(torch.chunk(pcm, nstreams, dim=0))
    parts = [p.contiguous() for p in parts]
    outs = [None] * nstreams
    def step():
        for i in range(nstreams):
            with torch.cuda.stream(streams[i]):
                z, codes, lat, _, _ = models[i].encode(parts[i])
                outs[i] = (codes, models[i].decode(z))
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    res = (torch.cat([o[0] for o in outs]).cpu().numpy(), torch.cat([o[1] for o in outs]).cpu().numpy())
    for m in models: m.dispose()
    return dt, res
d1, r1 = run(1); d2, r2 = run(2); d4, r4 = run(4)
print(f"1 stream {d1*1e3:.2f} ms  2 streams {d2*1e3:.2f} ms  4 streams {d4*1e3:.2f} ms; identical outputs: {np.array_equal(r1[0], r2[0]) and np.array_equal(r1[1], r2[1]) and np.array_equal(r1[1], r4[1])}")
