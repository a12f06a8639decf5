"""ORACLE (test infrastructure, never shipped on the product path).

PyTorch-CPU restatement of the reference's SNAC Encode / Decode graph, op for op, in the order the C# issues the
TorchSharp calls.  Defines the golden values ("parity unpinned" by the reference itself: it has no tests).

Reference files followed (all under /root/reference/NeuralCodecs.Torch/):
  Models/SNAC.cs:34-63 (ctor), :70-80 (Preprocess), :91-106 (forward), :113-122 / :129-150 (Encode), :157-192 (Decode)
  Modules/SNAC/Encoder.cs:26-69, EncoderBlock.cs:27-55, ResidualUnit.cs:25-60, Snake1d.cs:40-63
  Modules/SNAC/WNConv1d.cs:42-143 (w = v/||v|| * (g - 1e-7): D3), WNConvTranspose1d.cs:100-148
  Modules/SNAC/Decoder.cs:31-86, DecoderBlock.cs:29-70, NoiseBlock.cs:24-46 (randn at inference: D8 -> noise is injected)
  Modules/SNAC/VectorQuantizer.cs:40-141, ResidualVectorQuantizer.cs:30-135
  Modules/SNAC/LocalMHA.cs:46-135, SinusoidalEmbedding.cs:33-106, RotaryEmbedding.cs:16-68
  Config/SNAC/SNACConfig.cs:40-153, Core/Utils/MathUtils.cs:11-62 (LCM)
State-dict keys are the TorchSharp names (Sequential children by index, WNConv parameters under
"parametrizations.weight.original0/1", Modules/SNAC/WNConv1d.cs:66-70).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F


def _t(a) -> torch.Tensor:
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))


G0, G1 = ".parametrizations.weight.original0", ".parametrizations.weight.original1"


class TorchSNAC:
    def __init__(self, cfg, state_dict: Dict[str, np.ndarray]):
        self.cfg = cfg
        self.sd = {k: _t(v).to(torch.float32) for k, v in state_dict.items()}
        self.hop = cfg.hop_length
        self.latent = cfg.resolved_latent_dim
        self.attn = cfg.attn_window_size
        # Deviations of the C# port that can be UNDONE to reproduce upstream SNAC semantics, for the structural cross-check against an
        # independent composition only (tools/crosscheck_hf.py::crosscheck_snac_blocks).  Empty = the reference's behaviour.
        self.upstream = set()

    # ---- leaves ---------------------------------------------------------------------------
    def snake(self, x, key):
        alpha = self.sd[key + ".alpha"]
        if "D4" in self.upstream:                                                  # upstream: x + sin^2(ax) / (a + 1e-9)
            return x + (alpha + 1e-9).reciprocal() * torch.sin(alpha * x).pow(2)
        return torch.where(alpha == 0, x, torch.addcdiv(x, torch.sin(alpha * x).pow_(2), alpha, value=1))

    def _w(self, key):
        v, g = self.sd[key + G1], self.sd[key + G0]
        v_norm = v.contiguous().pow(2).sum([1, 2], keepdim=True, dtype=torch.float32).sqrt()
        if "D3" in self.upstream:                                                  # upstream: v * g / ||v||
            return torch.mul(v.div(v_norm), g.reshape(v.shape[0], 1, 1)).contiguous()
        return torch.mul(v.div(v_norm), g.reshape(v.shape[0], 1, 1).sub(1e-7)).contiguous()

    def conv(self, x, key, stride=1, padding=0, dilation=1, groups=1):
        return F.conv1d(x, self._w(key), self.sd.get(key + ".bias"), stride, padding, dilation, groups)

    def convT(self, x, key, stride, padding, output_padding):
        return F.conv_transpose1d(x, self._w(key), self.sd.get(key + ".bias"), stride, padding, output_padding)

    def res_unit(self, x, key, dilation, groups):
        y = self.snake(x, key + ".block.0")
        y = self.conv(y, key + ".block.1", padding=3 * dilation, dilation=dilation, groups=groups)
        y = self.snake(y, key + ".block.2")
        y = self.conv(y, key + ".block.3")
        return x.add(y)                                                           # lengths equal for k=7 (ResidualUnit.cs:52-59)

    def local_mha(self, x, key):
        # LocalMHA.cs:78-115
        B, C, T = x.shape
        w = self.attn
        heads = C // 64
        residual = x
        h = F.layer_norm(x.transpose(1, 2), (C,), self.sd[key + ".norm.weight"], self.sd[key + ".norm.bias"], 1e-5)
        windows = T // w
        qkv = F.linear(h, self.sd[key + ".to_qkv.weight"]).chunk(3, dim=-1)

        def rearr(t):
            return t.reshape(B, windows, T // windows, heads, C // heads).permute(0, 3, 1, 2, 4)
        q, k, v = (rearr(t) for t in qkv)
        # SinusoidalEmbedding.cs:67-80 (useXpos False -> scale == ones(1)); RotaryEmbedding.cs:46-68
        inv_freq = self.sd[key + ".rel_pos.inv_freq"]
        t = torch.arange(k.size(-2)).to(inv_freq.dtype)
        freqs = torch.einsum("i,j->ij", t, inv_freq)
        freqs = torch.cat([freqs, freqs], dim=-1)
        scale = torch.ones(1)

        def rot(u):
            d = u.size(-1)
            return torch.cat([u[..., d // 2:].neg(), u[..., : d // 2]], dim=-1)
        q = q.mul(freqs.cos()).mul(scale).add(rot(q).mul(freqs.sin()).mul(scale))
        k = k.mul(freqs.cos()).mul(scale.reciprocal()).add(rot(k).mul(freqs.sin()).mul(scale.reciprocal()))
        a = F.scaled_dot_product_attention(q, k, v)
        out = a.permute(0, 2, 3, 1, 4).reshape(B, windows * (T // windows), C)
        out = F.linear(out, self.sd[key + ".to_out.weight"])
        return out.transpose(1, 2).add(residual)

    # ---- encoder / decoder ----------------------------------------------------------------------
    def preprocess(self, x):
        from math import gcd
        a, b = self.cfg.vq_strides[0], (self.attn or 1)
        pad_to = self.hop * (a * b // gcd(a, b))
        L = x.shape[-1]
        right = -(-L // pad_to) * pad_to - L
        return F.pad(x, (0, right))

    def encoder(self, x):
        c = self.cfg
        d = c.encoder_dim
        x = self.conv(x, "encoder.block.0", padding=3)
        for bi, s in enumerate(c.encoder_rates):
            p = f"encoder.block.{bi + 1}"
            groups = d if c.depthwise else 1                                       # Encoder.cs:42 (dModel/2 after doubling)
            for ui, dil in enumerate((1, 3, 9)):
                x = self.res_unit(x, f"{p}.block.{ui}", dil, groups)
            x = self.snake(x, f"{p}.block.3")
            x = self.conv(x, f"{p}.block.4", stride=s, padding=-(-s // 2))
            d *= 2
        n = len(c.encoder_rates) + 1
        if self.attn:
            x = self.local_mha(x, f"encoder.block.{n}")
            n += 1
        return self.conv(x, f"encoder.block.{n}", padding=3, groups=d if c.depthwise else 1)

    def decoder(self, z, noises: Optional[Sequence[torch.Tensor]]):
        c = self.cfg
        n = 0
        if c.depthwise:
            x = self.conv(z, "decoder.model.0", padding=3, groups=self.latent)
            x = self.conv(x, "decoder.model.1")
            n = 2
        else:
            x = self.conv(z, "decoder.model.0", padding=3)
            n = 1
        if self.attn:
            x = self.local_mha(x, f"decoder.model.{n}")
            n += 1
        ch = c.decoder_dim
        out_dim = ch
        for bi, s in enumerate(c.decoder_rates):
            out_dim = ch // (1 << (bi + 1))
            p = f"decoder.model.{n}"
            x = self.snake(x, f"{p}.block.0")
            x = self.convT(x, f"{p}.block.1", s, -(-s // 2), s % 2)
            k = 2
            if c.noise:
                h = self.conv(x, f"{p}.block.2.linear")                            # NoiseBlock.cs:38-45
                nz = noises[bi] if noises is not None else torch.randn(x.shape[0], 1, x.shape[2])
                x = x + nz * h
                k = 3
            groups = out_dim if c.depthwise else 1
            for ui, dil in enumerate((1, 3, 9)):
                x = self.res_unit(x, f"{p}.block.{k + ui}", dil, groups)
            n += 1
        x = self.snake(x, f"decoder.model.{n}")
        x = self.conv(x, f"decoder.model.{n + 1}", padding=3)
        return torch.tanh(x)

    # ---- quantizer ------------------------------------------------------------------------------
    def vq(self, z, i, want_dist=False):
        p = f"quantizer.quantizers.{i}"
        s = self.cfg.vq_strides[i]
        if s > 1:
            z = F.avg_pool1d(z, kernel_size=s, stride=s)
        z_e = self.conv(z, p + ".in_proj")
        B, D, T = z_e.shape
        enc = z_e.transpose(1, 2).reshape(-1, D).contiguous()
        cb = self.sd[p + ".codebook.weight"]
        cb_full = cb
        if "D1" in self.upstream:                                  # upstream: lookup on l2-normalised vectors (VectorQuantizer.cs:125 says so, the code does not)
            enc, cb = F.normalize(enc), F.normalize(cb)
        dist = enc.pow(2).sum(1, keepdim=True) + cb.pow(2).sum(1, keepdim=True).t() - torch.einsum("bd,nd->bn", enc, cb).mul_(2.0)
        idx = dist.argmin(1).reshape(B, T)
        z_q = F.embedding(idx, cb_full).transpose(1, 2).contiguous()
        z_q = z_e + (z_q - z_e)
        z_q = self.conv(z_q, p + ".out_proj")
        if s > 1:
            z_q = z_q.repeat_interleave(s, dim=-1)
        return z_q, idx, (dist if want_dist else None)

    def quantize(self, z, want_dist=False):
        zq = torch.zeros_like(z)
        residual = z.clone()
        codes, dists = [], []
        for i in range(len(self.cfg.vq_strides)):
            zqi, idx, dist = self.vq(residual, i, want_dist)
            zq = torch.add(zq, zqi)
            residual = torch.sub(residual, zqi)
            codes.append(idx.clone())
            dists.append(dist)
        return zq, codes, dists

    def from_codes(self, codes: List[torch.Tensor]):
        zq = None
        for i, cds in enumerate(codes):
            p = f"quantizer.quantizers.{i}"
            zp = F.embedding(_t(cds).long(), self.sd[p + ".codebook.weight"]).transpose(1, 2).contiguous()
            zqi = self.conv(zp, p + ".out_proj")
            s = self.cfg.vq_strides[i]
            if s > 1:
                zqi = zqi.repeat_interleave(s, dim=-1)
            zq = zqi if zq is None else torch.add(zq, zqi)
        return zq

    # ---- API ------------------------------------------------------------------------------------
    @torch.inference_mode()
    def encode(self, pcm, want_dist=False):
        """SNAC.Encode(float[]) semantics (pads; the Tensor overload's missing pad is deviation D7)."""
        x = self.preprocess(_t(pcm).float())
        z = self.encoder(x)
        zq, codes, dists = self.quantize(z, want_dist)
        return (z, zq, codes, dists) if want_dist else (z, zq, codes)

    @torch.inference_mode()
    def encode_tensor(self, pcm, want_dist=False):
        """SNAC.Encode(Tensor) AS WRITTEN (Models/SNAC.cs:113-122, deviation D7): `preprocessed` is computed and dropped, the
        encoder runs on the un-padded tensor.  The quantizer raises (repeat_interleave + add shape mismatch) when the frame
        count is not a multiple of every vq stride -- the same libtorch exception the reference surfaces."""
        x = _t(pcm).float()
        _ = self.preprocess(x)
        z = self.encoder(x)
        zq, codes, dists = self.quantize(z, want_dist)
        return (z, zq, codes, dists) if want_dist else (z, zq, codes)

    @torch.inference_mode()
    def decode(self, codes, noises=None):
        zq = self.from_codes([_t(c) for c in codes])
        return self.decoder(zq, None if noises is None else [_t(n).float() for n in noises])

    @torch.inference_mode()
    def decode_latents(self, zq, noises=None):
        return self.decoder(_t(zq).float(), None if noises is None else [_t(n).float() for n in noises])
