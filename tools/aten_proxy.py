#!/usr/bin/env python3
"""ATen-CPU operator-sequence benchmark of the DAC encode+decode graph.

The reference's CPU path is TorchSharp 0.105, i.e. P/Invokes into libtorch's ATen CPU operators.  It cannot run here (no .NET), so this
file issues the SAME operator sequence at the SAME shapes through PyTorch's ATen CPU kernels: per-call weight-norm fold, element-wise
Snake (mul, sin, pow, addcdiv, eq, where), conv1d / conv_transpose1d, the 9-stage residual VQ (pow/sum/einsum/argmin/embedding), tanh.
It is a TIMING proxy only: nothing checks its outputs, it shares no code with oracle/ or with the reference, and it never runs inside
bench.py's timed GPU region.  Weights and PCM are the bench's synthetic tensors (any values give the same operator costs).

    python tools/aten_proxy.py [--clips 2] [--iters 3]
"""
import argparse
import json
import math
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


class DacOps:
    """The operator sequence of Models/DAC.cs Encode + Decode over a TorchSharp-keyed state dict (timing only)."""

    def __init__(self, cfg, sd):
        import torch
        self.t = torch
        self.F = torch.nn.functional
        self.cfg = cfg
        self.w = {k: torch.from_numpy(v).float() if not isinstance(v, torch.Tensor) else v.float() for k, v in sd.items()}

    def snake(self, x, key):
        t = self.t
        a = self.w[key + ".alpha"]
        return t.where(a == 0, x, t.addcdiv(x, t.sin(a * x).pow_(2), a, value=1))

    def fold(self, key):
        v, g = self.w[key + ".weight_v"], self.w[key + ".weight_g"]
        n = v.pow(2).sum([1, 2], keepdim=True).sqrt().add(1e-7)
        return v.div(n).mul(g.reshape(v.shape[0], 1, 1)).contiguous()

    def conv(self, x, key, stride=1, padding=0, dilation=1):
        return self.F.conv1d(x, self.fold(key), self.w.get(key + ".bias"), stride, padding, dilation, 1)

    def convT(self, x, key, stride, padding):
        return self.F.conv_transpose1d(x, self.fold(key), self.w.get(key + ".bias"), stride=stride, padding=padding)

    def unit(self, x, key, d):
        y = self.conv(self.snake(x, key + ".block.0"), key + ".block.1", padding=3 * d, dilation=d)
        y = self.conv(self.snake(y, key + ".block.2"), key + ".block.3")
        return y.add_(x)

    def encode(self, pcm):
        t, cfg = self.t, self.cfg
        hop = cfg.hop_length
        x = self.F.pad(pcm, [0, int(math.ceil(pcm.shape[-1] / hop) * hop) - pcm.shape[-1]])
        x = self.conv(x, "encoder.block.0", padding=3)
        for bi, s in enumerate(cfg.encoder_rates):
            p = f"encoder.block.{bi + 1}"
            for ui, d in enumerate((1, 3, 9)):
                x = self.unit(x, f"{p}.block.{ui}", d)
            x = self.conv(self.snake(x, f"{p}.block.3"), f"{p}.block.4", stride=s, padding=int(math.ceil(s / 2.0)))
        n = len(cfg.encoder_rates)
        z = self.conv(self.snake(x, f"encoder.block.{n + 1}"), f"encoder.block.{n + 2}", padding=1)
        residual, zq = z.clone(), t.zeros_like(z)
        for i in range(cfg.n_codebooks):
            p = f"quantizer.quantizers.{i}"
            ze = self.conv(residual, p + ".in_proj")
            cb = self.w[p + ".codebook.weight"]
            e = ze.transpose(1, 2).reshape(-1, cb.shape[1]).contiguous()
            dist = e.pow(2).sum(1, keepdim=True) + cb.pow(2).sum(1, keepdim=True).t() - t.einsum("bd,nd->bn", e, cb).mul_(2.0)
            idx = dist.argmin(1).reshape(ze.shape[0], ze.shape[-1])
            q = self.F.embedding(idx, cb).transpose(-2, -1).contiguous()
            q = self.conv(ze + (q - ze), p + ".out_proj")
            zq.add_(q)
            residual.sub_(q)
        return zq

    def decode(self, z):
        cfg = self.cfg
        x = self.conv(z, "decoder.model.0", padding=3)
        for bi, s in enumerate(cfg.decoder_rates):
            p = f"decoder.model.{bi + 1}"
            x = self.convT(self.snake(x, f"{p}.block.0"), f"{p}.block.1", s, int(math.ceil(s / 2.0)))
            for ui, d in enumerate((1, 3, 9)):
                x = self.unit(x, f"{p}.block.{ui + 2}", d)
        n = len(cfg.decoder_rates)
        return self.t.tanh(self.conv(self.snake(x, f"decoder.model.{n + 1}"), f"decoder.model.{n + 2}", padding=3))


def run(cfg, state_dict, pcm_h, seconds, iters=3):
    """One warm-up pass, then `iters` timed encode+decode passes; median.  Returns the cpu_baseline-style record."""
    import torch
    ops = DacOps(cfg, state_dict)
    x = torch.from_numpy(pcm_h)
    times = []
    with torch.inference_mode():
        ops.decode(ops.encode(x))
        for _ in range(max(1, iters)):
            t0 = time.perf_counter()
            ops.decode(ops.encode(x))
            times.append(time.perf_counter() - t0)
    med = statistics.median(times)
    n = pcm_h.shape[0]
    return {"value": round(n * seconds / med, 4), "unit": "audio-seconds/sec", "threads": int(torch.get_num_threads()), "cpu": cpu_model(),
            "iterations": len(times), "median_s": round(med, 3), "min_s": round(min(times), 3), "max_s": round(max(times), 3),
            "sample": f"{n} clips x {seconds:g} s, 1 warm-up + {len(times)} timed encode+decode passes through ATen CPU operators "
                      f"(tools/aten_proxy.py: the operator sequence TorchSharp-CPU dispatches to), median"}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=2)
    ap.add_argument("--iters", type=int, default=3)
    a = ap.parse_args()
    from neuralcodecs_amd import DACConfig
    from neuralcodecs_amd.weights import dac_synthetic_state_dict, synthetic_pcm
    cfg = DACConfig.dac_44khz()
    print(json.dumps(run(cfg, dac_synthetic_state_dict(cfg, seed=42), synthetic_pcm(a.clips, 1, cfg.sample_rate, cfg.sample_rate, seed=1234), 1.0, a.iters)))
