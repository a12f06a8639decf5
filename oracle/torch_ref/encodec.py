"""ORACLE (test infrastructure, never shipped on the product path).

PyTorch-CPU restatement of the reference's Encodec Encode / Decode graph, op for op, in the order the C# issues the
TorchSharp calls.  Defines the golden values ("parity unpinned" by the reference itself: it has no tests).

Reference files followed (all under /root/reference/NeuralCodecs.Torch/):
  Models/Encodec.cs:46-90 (ctor: only channels/dimension/norm/causal reach SEANet, D11), :145-201 (segment props),
                    :213-235 (Decode), :259-285 (Encode), :436-455 (DecodeFrame), :457-489 (EncodeFrame: RMS normalise)
  Modules/Encodec/SEANetEncoder.cs:37-148, SEANetDecoder.cs:40-153, SEANetResnetBlock.cs:29-85
  Modules/Encodec/SConv1d.cs:144-173 (asymmetric reflect pad), :245-250 (extra padding), :258-274 (small-input path, D9)
  Modules/Encodec/SConvTranspose1d.cs:116-171 (norm BEFORE the trim), NormConv1d.cs:35-164, NormConvTranspose1d.cs:21-127
  Modules/Encodec/WNConv1d.cs:110-126 / WNConvTranspose1d.cs:120-150 (w = v/||v|| * (g - 1e-7): D3)
  Modules/Encodec/SLSTM.cs:40-57, ResidualVectorQuantizer.cs:107-157, VectorQuantizer.cs:76-115, EuclideanCodebook.cs:82-182
  AudioTools/AudioTensorDSP.cs:161-261 (LinearOverlapAdd), Utils/TorchUtils.cs:26-30 (ELU alpha 1)
State-dict keys are the TorchSharp names: encoder.layers.N.conv.{weight|weight_v|weight_g|bias}, .norm.{weight,bias},
.block.{1,3}.*, .shortcut.*, .lstm.{weight_ih_l0,...}; quantizer.layers.N.codebook.embed.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F


def _t(a) -> torch.Tensor:
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))


class TorchEncodec:
    def __init__(self, cfg, state_dict: Dict[str, np.ndarray]):
        self.cfg = cfg
        self.sd = {k: _t(v).to(torch.float32) for k, v in state_dict.items()}
        self.gn = cfg.norm == "time_group_norm"
        self.causal = cfg.causal
        self.ratios = list(cfg.ratios)
        self.hop = cfg.hop_length
        self.frame_rate = int(math.ceil(cfg.sampling_rate / float(self.hop)))          # Encodec.cs:83
        self.bits = int(math.log2(cfg.codebook_size))
        self.n_q_total = int(1000 * max(cfg.target_bandwidths) / (math.ceil(cfg.sampling_rate / self.hop) * 10))
        self.bandwidth = cfg.bandwidth
        self._lstm = {}
        # Deviations of the C# port that can be UNDONE to reproduce upstream (Meta / HF transformers) semantics, for the structural
        # cross-check of tools/crosscheck_hf.py only: {"D9"} (the small-input reflect path never trims its zero
        # extension).  Empty = the reference's behaviour.
        self.upstream = set()

    # ---- segment properties (Encodec.cs:190-196)
    @property
    def segment_length(self) -> Optional[int]:
        return None if self.cfg.segment_seconds is None else int(self.cfg.segment_seconds * self.cfg.sampling_rate)

    @property
    def segment_stride(self) -> Optional[int]:
        sl = self.segment_length
        return None if sl is None else max(1, int((1 - self.cfg.overlap) * sl))

    def n_q(self) -> int:
        bw_per_q = self.bits * self.frame_rate                                          # GetBandwidthPerQuantizer
        if self.bandwidth and self.bandwidth > 0:
            return int(max(1, math.floor(self.bandwidth * 1000 / bw_per_q)))
        return self.n_q_total

    # ---- leaves ------------------------------------------------------------------------------------
    def _weight(self, key, transposed=False):
        if key + ".conv.weight" in self.sd:
            return self.sd[key + ".conv.weight"]
        v, g = self.sd[key + ".conv.weight_v"], self.sd[key + ".conv.weight_g"]
        v_norm = v.contiguous().pow(2).sum([1, 2], keepdim=True, dtype=torch.float32).sqrt()
        return torch.mul(v.div(v_norm), g.reshape(v.shape[0], 1, 1).sub(1e-7)).contiguous()

    def _norm(self, y, key):
        if not self.gn:
            return y
        return F.group_norm(y, 1, self.sd[key + ".norm.weight"], self.sd[key + ".norm.bias"], 1e-5)

    def _pad1d(self, x, left, right):
        # SConv1d.cs:258-274: always reflect; the small-input path zero-pads first and never trims (D9)
        L = x.size(-1)
        extra = 0
        if L <= max(left, right):
            extra = max(left, right) - L + 1
            x = F.pad(x, (0, extra), mode="constant", value=0.0)
        y = F.pad(x, (left, right), mode="reflect")
        if extra and "D9" in self.upstream:                                    # upstream (ConvUtils.cs:62-100 / HF _pad1d): drop the zero extension again
            y = y[..., : y.size(-1) - extra]
        return y

    def sconv(self, x, key, k, stride=1, dilation=1):
        L = x.size(2)
        eff = (k - 1) * dilation + 1
        pad_total = eff - stride
        n_frames = np.float32(L - eff + pad_total) / np.float32(stride) + 1                 # float division as in the C#
        ideal = (int(math.ceil(float(n_frames))) - 1) * stride + (eff - pad_total)
        extra = ideal - L
        if self.causal:
            xp = self._pad1d(x, pad_total, extra)
        else:
            right = pad_total // 2
            xp = self._pad1d(x, pad_total - right, right + extra)
        y = F.conv1d(xp, self._weight(key), self.sd.get(key + ".conv.bias"), stride, 0, dilation)
        return self._norm(y, key)

    def sconvT(self, x, key, k, stride):
        y = F.conv_transpose1d(x, self._weight(key, True), self.sd.get(key + ".conv.bias"), stride)
        y = self._norm(y, key)
        pad_total = k - stride
        if self.causal:
            right = int(math.ceil(pad_total * 1.0))
            left = pad_total - right
        else:
            right = pad_total // 2
            left = pad_total - right
        return y[..., left: y.size(-1) - right]

    def resblock(self, x, key, dim):
        s = self.sconv(x, key + ".shortcut", 1)
        y = F.elu(x, 1.0)
        y = self.sconv(y, key + ".block.1", self.cfg.residual_kernel_size, 1, 1)
        y = F.elu(y, 1.0)
        y = self.sconv(y, key + ".block.3", 1)
        return torch.add(s, y)

    def slstm(self, x, key):
        # SLSTM.cs:40-57: torch LSTM(dimension, dimension, numLayers) on [T,B,C], skip add, permute back
        p = x.permute(2, 0, 1).contiguous()
        C = p.shape[-1]
        if key not in self._lstm:
            m = torch.nn.LSTM(C, C, self.cfg.lstm_layers)
            with torch.no_grad():
                for l in range(self.cfg.lstm_layers):
                    for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                        getattr(m, f"{nm}_l{l}").copy_(self.sd[f"{key}.lstm.{nm}_l{l}"])
            self._lstm[key] = m.eval()
        out, _ = self._lstm[key](p)
        return out.add(p).permute(1, 2, 0)

    # ---- SEANet ----------------------------------------------------------------------------------------
    def encoder(self, x):
        c = self.cfg
        x = self.sconv(x, "encoder.layers.0", c.kernel_size)
        n, mult = 1, 1
        for r in reversed(self.ratios):
            x = self.resblock(x, f"encoder.layers.{n}", mult * c.n_filters)
            x = F.elu(x, 1.0)
            x = self.sconv(x, f"encoder.layers.{n + 2}", 2 * r, r)
            n += 3
            mult *= 2
        x = self.slstm(x, f"encoder.layers.{n}")
        x = F.elu(x, 1.0)
        return self.sconv(x, f"encoder.layers.{n + 2}", c.last_kernel_size)

    def decoder(self, z):
        c = self.cfg
        x = self.sconv(z, "decoder.layers.0", c.kernel_size)
        x = self.slstm(x, "decoder.layers.1")
        n = 2
        for r in self.ratios:
            x = F.elu(x, 1.0)
            x = self.sconvT(x, f"decoder.layers.{n + 1}", 2 * r, r)
            x = self.resblock(x, f"decoder.layers.{n + 2}", 0)
            n += 3
        x = F.elu(x, 1.0)
        return self.sconv(x, f"decoder.layers.{n + 1}", c.last_kernel_size)

    # ---- quantizer -------------------------------------------------------------------------------------
    def rvq_encode(self, emb, want_dist=False):
        residual = emb.clone()
        codes, dists = [], []
        for i in range(self.n_q()):
            embed = self.sd[f"quantizer.layers.{i}.codebook.embed"]
            x = residual.transpose(1, 2)
            flat = x.reshape(-1, x.size(-1))
            dist = flat.pow(2).sum(1, keepdim=True).add(embed.pow(2).sum(1, keepdim=True).t()).add(-2 * flat.matmul(embed.t()))
            idx = dist.argmin(dim=-1).view(x.shape[:-1])
            q = embed[idx].transpose(1, 2)
            residual = residual - q
            codes.append(idx)
            dists.append(dist if want_dist else None)
        return torch.stack(codes, dim=1), dists

    def rvq_decode(self, codes):
        out = torch.zeros(1)
        for i in range(codes.size(1)):
            out = out + self.sd[f"quantizer.layers.{i}.codebook.embed"][codes[:, i]].transpose(1, 2)
        return out

    # ---- API -------------------------------------------------------------------------------------------
    @torch.inference_mode()
    def encode_frame(self, x, want_dist=False):
        scale = None
        if self.cfg.normalize:
            mono = x.mean([1], keepdim=True)
            volume = mono.pow(2).mean([2], keepdim=True).sqrt()
            scale = volume.add(1e-8)
            x = x.div(scale)
            scale = scale.view(-1, 1)
        emb = self.encoder(x)
        codes, dists = self.rvq_encode(emb, want_dist)
        return codes, scale, emb, dists

    @torch.inference_mode()
    def encode(self, pcm, want_dist=False):
        """Encodec.Encode(Tensor) -> list of (codes [B,nQ,T'], scale [B,1] | None) per segment (+ emb / dists for the goldens)."""
        x = _t(pcm).float()
        length = x.size(2)
        seg = self.segment_length or length
        stride = self.segment_stride or length
        frames = []
        for off in range(0, length, stride):
            frames.append(self.encode_frame(x[:, :, off: min(off + seg, length)], want_dist))
        return frames

    @torch.inference_mode()
    def decode_frame(self, codes, scale):
        out = self.decoder(self.rvq_decode(_t(codes).long()))
        if scale is not None:
            out = out * _t(scale).view(-1, 1, 1)
        return out

    @torch.inference_mode()
    def decode(self, frames):
        """Encodec.Decode(List<EncodedFrame>): frames = [(codes, scale), ...]"""
        if len(frames) == 0:
            raise ValueError("No frames provided to decode")
        if self.segment_length is None:
            if len(frames) != 1:
                raise ValueError("Expected single frame when no segmentation is used")
            return self.decode_frame(*frames[0][:2])
        outs = [self.decode_frame(f[0], f[1]) for f in frames]
        return self.linear_overlap_add(outs, self.segment_stride)

    @staticmethod
    def linear_overlap_add(frames: List[torch.Tensor], stride: int):
        # AudioTensorDSP.cs:161-261
        total = stride * (len(frames) - 1) + frames[-1].shape[-1]
        L0 = frames[0].shape[-1]
        t = torch.linspace(0, 1, L0 + 2)[1:-1]
        weight = torch.tensor(0.5) - (t - torch.tensor(0.5)).abs()
        sum_w = torch.zeros(total)
        out = torch.zeros(tuple(frames[0].shape[:-1]) + (total,))
        off = 0
        for f in frames:
            n = f.shape[-1]
            w = weight.narrow(0, 0, n)
            out.narrow(-1, off, n).add_(f.mul(w))
            sum_w.narrow(0, off, n).add_(w)
            off += stride
        if sum_w.min().item() <= 1e-10:
            sum_w = sum_w.add(1e-10)
        return out.div(sum_w)
