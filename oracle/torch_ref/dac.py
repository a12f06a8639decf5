"""ORACLE (test infrastructure, never shipped on the product path).

PyTorch-CPU restatement of the reference's DAC Encode / Decode / FromCodes graph,
op for op, in the order the C# issues the TorchSharp calls.  The C# cannot run
here (no .NET), and its arithmetic lives in libtorch (NuGet TorchSharp 0.105.0,
NeuralCodecs.Torch/NeuralCodecs.Torch.csproj:48-52), so this file *defines* the
golden values ("parity unpinned" by the reference itself: it has no tests).

Reference files followed (all under /root/reference/NeuralCodecs.Torch/):
  Models/DAC.cs:141-154 (Preprocess), :163-181 (Encode), :231-234 (Decode), :101-106 (FromCodes)
  Modules/DAC/Encoder.cs:21-58, EncoderBlock.cs:20-43, ResidualUnit.cs:24-59
  Modules/DAC/Snake1d.cs:49-58, WNConv1d.cs:140-156, WNConvTranspose1d.cs:142-163
  Modules/DAC/Decoder.cs:22-58, DecoderBlock.cs:20-44
  Modules/DAC/VectorQuantizer.cs:64-142, ResidualVectorQuantizer.cs:54-103,105-206,211-238
Deviations from upstream Descript DAC that are kept on purpose: SURVEY 2.3 D1, D2, D4, D5, D6, D13.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F


def _t(a) -> torch.Tensor:
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))


class TorchDAC:
    """Functional DAC over a TorchSharp-keyed state dict (numpy or torch values)."""

    def __init__(self, cfg, state_dict: Dict[str, np.ndarray]):
        self.cfg = cfg
        self.sd = {k: _t(v).to(torch.float32) for k, v in state_dict.items()}
        self.hop = cfg.hop_length
        self.taps: Dict[str, torch.Tensor] = {}
        self.record = False
        # Deviations of the C# port that can be UNDONE to reproduce upstream (Descript / HF transformers) semantics, for the structural
        # cross-check of tools/crosscheck_hf.py only: {"D1", "D2", "D4"}.  Empty = the reference's behaviour.
        self.upstream = set()

    # ---- leaf modules -------------------------------------------------------------------
    def snake(self, x: torch.Tensor, key: str) -> torch.Tensor:
        # Snake1d.cs:52  where(alpha == 0, x, addcdiv(x, sin(alpha*x).pow_(2), alpha, 1))
        alpha = self.sd[key + ".alpha"]
        if "D4" in self.upstream:                                              # upstream: x + sin^2(ax) / (a + 1e-9)
            return x + (alpha + 1e-9).reciprocal() * torch.sin(alpha * x).pow(2)
        return torch.where(alpha == 0, x, torch.addcdiv(x, torch.sin(alpha * x).pow_(2), alpha, value=1))

    def _wn_weight(self, key: str) -> torch.Tensor:
        # WNConv1d.cs:145-150 / WNConvTranspose1d.cs:146-150  (D2: eps added to the norm; g is per dim-0 slice, D13)
        v = self.sd[key + ".weight_v"]
        g = self.sd[key + ".weight_g"]
        v_norm = v.contiguous().pow(2).sum([1, 2], keepdim=True, dtype=torch.float32).sqrt()
        if "D2" in self.upstream:                                              # upstream: v * g / ||v||
            return torch.mul(v.div(v_norm), g.reshape(v.shape[0], 1, 1)).contiguous()
        normalized = v.div(v_norm.add(1e-7))
        return torch.mul(normalized, g.reshape(v.shape[0], 1, 1)).contiguous()

    def wnconv(self, x, key, stride=1, padding=0, dilation=1, groups=1):
        w = self._wn_weight(key)
        return F.conv1d(x, w, self.sd.get(key + ".bias"), stride, padding, dilation, groups)

    def wnconvT(self, x, key, stride=1, padding=0, output_padding=0):
        w = self._wn_weight(key)
        return F.conv_transpose1d(x, w, self.sd.get(key + ".bias"), stride=stride, padding=padding,
                                  output_padding=output_padding, groups=1, dilation=1)

    def res_unit(self, x, key, dilation):
        # ResidualUnit.cs:24-59
        pad = (7 - 1) * dilation // 2
        y = self.snake(x, key + ".block.0")
        y = self.wnconv(y, key + ".block.1", padding=pad, dilation=dilation)
        y = self.snake(y, key + ".block.2")
        y = self.wnconv(y, key + ".block.3")
        p = (x.shape[-1] - y.shape[-1]) // 2
        if p > 0:
            x = x[..., p:-p]
        return y.add_(x)

    def _tap(self, name, t):
        if self.record:
            self.taps[name] = t.detach().clone()

    # ---- encoder / decoder --------------------------------------------------------------
    def encoder(self, x):
        cfg = self.cfg
        x = self.wnconv(x, "encoder.block.0", padding=3)                       # Encoder.cs:31
        self._tap("enc.stem", x)
        for bi, s in enumerate(cfg.encoder_rates):                             # EncoderBlock.cs:22-34
            p = f"encoder.block.{bi + 1}"
            for ui, d in enumerate((1, 3, 9)):
                x = self.res_unit(x, f"{p}.block.{ui}", d)
                self._tap(f"enc.b{bi}.r{ui}", x)
            x = self.snake(x, f"{p}.block.3")
            x = self.wnconv(x, f"{p}.block.4", stride=s, padding=int(math.ceil(s / 2.0)))
            self._tap(f"enc.b{bi}.down", x)
        n = len(cfg.encoder_rates)
        x = self.snake(x, f"encoder.block.{n + 1}")                            # Encoder.cs:44-45
        x = self.wnconv(x, f"encoder.block.{n + 2}", padding=1)
        self._tap("enc.out", x)
        return x

    def decoder(self, x):
        cfg = self.cfg
        x = self.wnconv(x, "decoder.model.0", padding=3)                       # Decoder.cs:30
        self._tap("dec.in", x)
        for bi, s in enumerate(cfg.decoder_rates):                             # DecoderBlock.cs:23-35
            p = f"decoder.model.{bi + 1}"
            x = self.snake(x, f"{p}.block.0")
            x = self.wnconvT(x, f"{p}.block.1", stride=s, padding=int(math.ceil(s / 2.0)))
            self._tap(f"dec.b{bi}.up", x)
            for ui, d in enumerate((1, 3, 9)):
                x = self.res_unit(x, f"{p}.block.{ui + 2}", d)
                self._tap(f"dec.b{bi}.r{ui}", x)
        n = len(cfg.decoder_rates)
        x = self.snake(x, f"decoder.model.{n + 1}")                            # Decoder.cs:43-47
        x = self.wnconv(x, f"decoder.model.{n + 2}", padding=3)
        return torch.tanh(x)

    # ---- quantizer ----------------------------------------------------------------------
    def vq_decode_latents(self, i, latents):
        # VectorQuantizer.cs:99-125 (D1: plain squared-Euclidean on un-normalised vectors)
        cb = self.sd[f"quantizer.quantizers.{i}.codebook.weight"].contiguous()
        shape = latents.shape
        enc = latents.transpose(1, 2).reshape(-1, cb.shape[1]).contiguous()
        if "D1" in self.upstream:                                              # upstream: cosine-like lookup on l2-normalised vectors
            enc = F.normalize(enc)
            cb = F.normalize(cb)
        enc_sq = enc.pow(2).sum(1, keepdim=True)
        cb_sq = cb.pow(2).sum(1, keepdim=True)
        cross = torch.einsum("bd,nd->bn", enc, cb).mul_(2.0)
        dist = enc_sq + cb_sq.t() - cross
        idx = dist.argmin(1).reshape(shape[0], shape[-1]).to(torch.int64)
        return self.vq_decode_code(i, idx), idx, dist

    def vq_decode_code(self, i, idx):
        # VectorQuantizer.cs:135-142
        cb = self.sd[f"quantizer.quantizers.{i}.codebook.weight"]
        return F.embedding(idx, cb).contiguous().transpose(-2, -1).contiguous()

    def vq_forward(self, i, z):
        # VectorQuantizer.cs:64-91 (losses are computed then discarded by the caller: skipped)
        p = f"quantizer.quantizers.{i}"
        z_e = self.wnconv(z, p + ".in_proj")
        z_q, idx, dist = self.vq_decode_latents(i, z_e)
        z_q = z_e + (z_q - z_e)                                                # straight-through, restated literally
        z_q = self.wnconv(z_q, p + ".out_proj")
        return z_q, idx, z_e, dist

    def rvq_forward(self, z, n_quantizers: Optional[int] = None, want_dist: bool = False):
        # ResidualVectorQuantizer.cs:54-103 (n_quantizers None) / :105-206 eval branch (mask all-true, break at n)
        nq = self.cfg.n_codebooks if n_quantizers is None else min(n_quantizers, self.cfg.n_codebooks)
        residual = z.clone()
        z_q = torch.zeros_like(z)
        codes, latents, dists = [], [], []
        for i in range(nq):
            zqi, idx, z_e, dist = self.vq_forward(i, residual)
            z_q.add_(zqi)
            residual.sub_(zqi)
            codes.append(idx)
            latents.append(z_e)
            if want_dist:
                dists.append(dist)
        out = (z_q, torch.stack(codes, 1), torch.cat(latents, 1))
        return out + (dists,) if want_dist else out

    def from_codes(self, codes):
        # ResidualVectorQuantizer.cs:211-238; zQ starts as zeros(1) and is promoted by the first add
        z_q = torch.zeros(1, dtype=torch.float32)
        for i in range(codes.shape[1]):
            z_p = self.vq_decode_code(i, codes[:, i, :])
            z_q = z_q.add(self.wnconv(z_p, f"quantizer.quantizers.{i}.out_proj"))
        return z_q

    # ---- model API ----------------------------------------------------------------------
    def preprocess(self, audio, sample_rate=None):
        # DAC.cs:141-154
        if sample_rate is not None and sample_rate != self.cfg.sample_rate:
            raise ValueError("sample rate mismatch")
        length = audio.shape[-1]
        right = int(math.ceil(length / self.hop) * self.hop) - length
        return F.pad(audio, [0, right])

    @torch.inference_mode()
    def encode(self, audio, n_quantizers=None, sample_rate=None, want_dist=False):
        x = self.preprocess(_t(audio).to(torch.float32), sample_rate)
        z = self.encoder(x)
        self._tap("enc.z", z)
        return self.rvq_forward(z, n_quantizers, want_dist)

    @torch.inference_mode()
    def decode(self, z):
        return self.decoder(_t(z).to(torch.float32))

    @torch.inference_mode()
    def forward(self, audio, n_quantizers=None):
        z_q, codes, latents = self.encode(audio, n_quantizers)
        return {"audio": self.decode(z_q), "z": z_q, "codes": codes, "latents": latents}
