#!/usr/bin/env python3
"""Structural cross-check of the oracle's torch restatement against an INDEPENDENT implementation of the upstream models
(SURVEY 8c "cross-check aid"): HF `transformers` DacModel / EncodecModel, random-initialised, build container only.

The reference (C#/TorchSharp) cannot run here and holds no vectors, so nothing can pin the oracle to TorchSharp output bit for bit.
What CAN be checked is that oracle/torch_ref -- which restates the C# graph op for op -- is the same network as the upstream
model it was ported from, once the port's documented arithmetic deviations (D1 un-normalised VQ distance, D2 weight-norm
epsilon, D4 Snake epsilon) are switched back to upstream: same layer order, dilations, paddings, transposed-conv geometry, key map
(the reference's StateDictNameConverter table, neuralcodecs_amd/checkpoint.py), quantizer recursion.  A layout or structure error in
the restatement shows up here as an O(1) difference; agreement is to float32 round-off.

    python tools/crosscheck_hf.py            # prints the max-abs differences per stage
Nothing of this travels to the GPU box (transformers is only imported here and in tests/test_crosscheck_hf_cpu.py).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from neuralcodecs_amd import checkpoint  # noqa: E402
from neuralcodecs_amd.config import DACConfig, EncodecConfig  # noqa: E402
from oracle.torch_ref.dac import TorchDAC  # noqa: E402
from oracle.torch_ref.encodec import TorchEncodec  # noqa: E402


def crosscheck_dac(seed=0, T=3200):
    from transformers import DacConfig as HFDacConfig, DacModel
    torch.manual_seed(seed)
    hcfg = HFDacConfig(encoder_hidden_size=8, downsampling_ratios=[2, 4, 5, 8], decoder_hidden_size=48, upsampling_ratios=[8, 5, 4, 2],
                       n_codebooks=4, codebook_size=64, codebook_dim=8, sampling_rate=16000)
    hf = DacModel(hcfg).eval()
    with torch.no_grad():                                     # random-init gives unit Snake alphas and tiny weights: make them generic
        for n, p in hf.named_parameters():
            if n.endswith("alpha"):
                p.copy_(torch.empty_like(p).uniform_(0.5, 2.0))
            elif n.endswith("weight") and p.dim() == 3:
                p.copy_(torch.randn_like(p) * (0.6 / np.sqrt(p.shape[1] * p.shape[2])))
            elif n.endswith("codebook.weight"):
                p.copy_(torch.randn_like(p))
            elif n.endswith("bias"):
                p.copy_(torch.randn_like(p) * 0.05)
    sd_hf = {k: v.detach().numpy() for k, v in hf.state_dict().items()}
    cfg = DACConfig(sample_rate=16000, encoder_dim=8, encoder_rates=(2, 4, 5, 8), decoder_dim=48, decoder_rates=(8, 5, 4, 2), n_codebooks=4,
                    codebook_size=64, codebook_dim=8)
    native = checkpoint.convert_dac_state_dict(sd_hf)         # the reference's key map + weight split
    ours = TorchDAC(cfg, native)
    ours.upstream = {"D1", "D2", "D4"}
    x = torch.randn(2, 1, T) * 0.3
    out = {}
    with torch.inference_mode():
        z_hf = hf.encoder(x)
        z = ours.encoder(x)
        out["encoder_max_abs"] = float((z - z_hf).abs().max())
        out["encoder_scale"] = float(z_hf.abs().max())
        q_hf = hf.quantizer(z_hf)
        zq_hf, codes_hf = q_hf[0], q_hf[1]
        zq, codes, _ = ours.rvq_forward(z_hf.clone())
        out["codes_equal_frac"] = float((codes == codes_hf).float().mean())
        same = (codes == codes_hf).all(dim=1)                 # frames where no stage flipped on a near-tie
        out["zq_max_abs_same_frames"] = float(((zq - zq_hf).abs().amax(dim=1))[same].max()) if same.any() else None
        a_hf = hf.decoder(zq_hf)
        a = ours.decoder(zq_hf.clone())
        out["decoder_max_abs"] = float((a - a_hf).abs().max())
        out["decoder_scale"] = float(a_hf.abs().max())
        # and the reference's own behaviour differs where it should: D1 changes the codes
        ours.upstream = set()
        _, codes_ref, _ = ours.rvq_forward(z_hf.clone())
        out["codes_equal_frac_with_D1_as_in_reference"] = float((codes_ref == codes_hf).float().mean())
    return out


def crosscheck_encodec(seed=0, T=4000):
    """HF EncodecModel (24 kHz layout: causal, weight-norm, no segmentation) vs oracle/torch_ref/encodec.py: encoder, decoder."""
    from transformers import EncodecConfig as HFEncodecConfig, EncodecModel
    torch.manual_seed(seed)
    hcfg = HFEncodecConfig(sampling_rate=16000, audio_channels=1, num_filters=4, hidden_size=32, upsampling_ratios=[4, 3, 2, 2], codebook_size=64,
                           codebook_dim=32, target_bandwidths=[1.5, 3.0, 6.0], normalize=False, use_causal_conv=True, norm_type="weight_norm",
                           num_lstm_layers=2, compress=2, kernel_size=7, last_kernel_size=7, residual_kernel_size=3, use_conv_shortcut=True)
    hf = EncodecModel(hcfg).eval()
    sd = {}
    for k, v in hf.state_dict().items():
        sd[k] = v.detach().numpy()
    cfg = EncodecConfig(sampling_rate=16000, channels=1, dimension=32, norm="weight_norm", causal=True, normalize=False,
                        target_bandwidths=(1.5, 3.0, 6.0), bandwidth=3.0, codebook_size=64, n_filters=4, ratios=(4, 3, 2, 2))
    native = {}
    for k, v in sd.items():                                   # HF: parametrizations.weight.original0 (g) / original1 (v) -> weight_g / weight_v
        k2 = k.replace(".parametrizations.weight.original0", ".weight_g").replace(".parametrizations.weight.original1", ".weight_v")
        native[k2] = v
    ours = TorchEncodec(cfg, native)
    x = torch.randn(2, 1, T) * 0.3
    out = {}
    with torch.inference_mode():
        e_hf = hf.encoder(x)
        e = ours.encoder(x)
        out["encoder_max_abs"] = float((e - e_hf).abs().max())
        out["encoder_scale"] = float(e_hf.abs().max())
        d_hf = hf.decoder(e_hf)
        d = ours.decoder(e_hf.clone())
        out["decoder_max_abs"] = float((d - d_hf).abs().max())
    return out


def crosscheck_encodec48(seed=0, T=8064):
    """HF EncodecModel in the 48 kHz LAYOUT (stereo, GroupNorm(1,C) after every conv, non-causal asymmetric reflect padding, RMS
    normalisation, 0.25 s chunks with 1 % overlap and linear overlap-add) at reduced width vs oracle/torch_ref/encodec.py:
    Encode (per-chunk codes and scales), the Euclidean RVQ, Decode (overlap-add).  Models/Encodec.cs:213-285,457-489,
    Modules/Encodec/NormConv1d.cs:122-164, SConv1d.cs:258-274, EuclideanCodebook.cs:155-182, AudioTools/AudioTensorDSP.cs:161-261.
    T = 8064 samples at 16 kHz -> chunks of 4000 at stride 3960: 4000 + 4000 + 144; the 144-sample tail reaches the last encoder
    convolution (k = 7, pads 3 + 3) with 3 frames and takes the small-input reflect path -- like the 960-sample tail of a 2 s clip in
    the real 48 kHz model (960 -> 3 frames).  Upstream trims the zero extension again (3 code frames); the reference does not (D9: 4)."""
    from transformers import EncodecConfig as HFEncodecConfig, EncodecModel
    torch.manual_seed(seed)
    hcfg = HFEncodecConfig(sampling_rate=16000, audio_channels=2, num_filters=4, hidden_size=32, upsampling_ratios=[4, 3, 2, 2], codebook_size=64,
                           codebook_dim=32, target_bandwidths=[3.0, 6.0, 12.0], normalize=True, use_causal_conv=False, norm_type="time_group_norm",
                           chunk_length_s=0.25, overlap=0.01, num_lstm_layers=2, compress=2, kernel_size=7, last_kernel_size=7,
                           residual_kernel_size=3, use_conv_shortcut=True, pad_mode="reflect")
    hf = EncodecModel(hcfg).eval()
    with torch.no_grad():                                     # generic GroupNorm affines and codebooks (random init leaves them at 1 / 0 / zeros)
        for n, p in hf.named_parameters():
            if ".norm.weight" in n:
                p.copy_(torch.empty_like(p).uniform_(0.8, 1.2))
            elif ".norm.bias" in n:
                p.copy_(torch.randn_like(p) * 0.05)
        for n, b in hf.named_buffers():
            if n.endswith("codebook.embed"):
                b.copy_(torch.randn_like(b))
    native = {k: v.detach().numpy() for k, v in hf.state_dict().items()}
    cfg = EncodecConfig(sampling_rate=16000, channels=2, dimension=32, norm="time_group_norm", causal=False, normalize=True, segment_seconds=0.25,
                        target_bandwidths=(3.0, 6.0, 12.0), bandwidth=6.0, codebook_size=64, n_filters=4, ratios=(4, 3, 2, 2))
    ours = TorchEncodec(cfg, native)
    x = torch.randn(2, 2, T) * 0.3
    out = {"segment_length": ours.segment_length, "segment_stride": ours.segment_stride, "hf_chunk": (hcfg.chunk_length, hcfg.chunk_stride)}
    with torch.inference_mode():
        codes_hf, scales_hf, pad_hf = hf.encode(x, bandwidth=6.0, return_dict=False)
        a_hf = hf.decode(codes_hf, scales_hf, last_frame_pad_length=pad_hf, return_dict=False)[0]
        for tag, up in (("upstream", {"D9"}), ("reference", set())):
            ours.upstream = up
            frames = ours.encode(x, want_dist=False)
            out[f"{tag}_n_frames"] = len(frames)
            out[f"{tag}_frame_lens"] = [int(f[0].shape[-1]) for f in frames]
            if tag == "upstream":
                out["hf_frame_lens"] = [int(codes_hf.shape[-1])] * (codes_hf.shape[0] - 1) + [int(codes_hf.shape[-1] - pad_hf)]
                eq, tot, sc = 0, 0, 0.0
                for i, (c, s_, _e, _d) in enumerate(frames):
                    ch = codes_hf[i][..., : c.shape[-1]]
                    eq += int((c == ch).sum()); tot += c.numel()
                    sc = max(sc, float((s_ - scales_hf[i]).abs().max()))
                out["codes_equal_frac"] = eq / tot
                out["scale_max_abs"] = sc
                # decode HF's own codes with our decoder + overlap-add: isolates the decoder / overlap-add from encoder-side near-ties
                fr_hf = [(codes_hf[i][..., : frames[i][0].shape[-1]], scales_hf[i]) for i in range(len(frames))]
                a = ours.decode(fr_hf)
                n = min(a.shape[-1], a_hf.shape[-1])
                out["decode_len"] = (int(a.shape[-1]), int(a_hf.shape[-1]))
                out["decode_max_abs"] = float((a[..., :n] - a_hf[..., :n]).abs().max())
                out["decode_scale"] = float(a_hf.abs().max())
                # the encoder alone on one full chunk (normalised input), before any argmin
                xs = x[..., :4000]
                mono = xs.mean(1, keepdim=True)
                xs = xs / (mono.pow(2).mean(-1, keepdim=True).sqrt() + 1e-8)
                e_hf, e = hf.encoder(xs), ours.encoder(xs)
                out["encoder_max_abs"] = float((e - e_hf).abs().max())
                out["encoder_scale"] = float(e_hf.abs().max())
    return out


def crosscheck_snac_localmha(seed=0, C=128, T=48, window=8):
    """SNAC's LocalMHA (Modules/SNAC/LocalMHA.cs:78-115, SinusoidalEmbedding.cs:33-106, RotaryEmbedding.cs:46-68) as restated in
    oracle/torch_ref/snac.py, against an INDEPENDENT composition: torch.nn.LayerNorm / nn.Linear modules, heads split in the
    full-sequence layout, HF transformers' Llama rotary embedding (its own inv_freq table and rotate-half convention) evaluated at the
    position INSIDE the window, and plain softmax attention over the whole sequence under a block-diagonal mask -- no window reshape
    anywhere.  Agreement pins the window partition, the head / channel order of the fused qkv projection, the rotary convention and
    positions, and the 1/sqrt(64) scale."""
    import types
    from transformers import LlamaConfig
    from transformers.models.llama.modeling_llama import LlamaRotaryEmbedding, apply_rotary_pos_emb
    from oracle.torch_ref.snac import TorchSNAC
    torch.manual_seed(seed)
    heads = C // 64
    key = "mha"
    norm = torch.nn.LayerNorm(C)
    to_qkv = torch.nn.Linear(C, 3 * C, bias=False)
    to_out = torch.nn.Linear(C, C, bias=False)
    with torch.no_grad():
        norm.weight.uniform_(0.8, 1.2)
        norm.bias.normal_(0, 0.05)
    rope = LlamaRotaryEmbedding(LlamaConfig(hidden_size=C, num_attention_heads=heads, head_dim=64, rope_theta=10000.0, max_position_embeddings=window))
    sd = {key + ".norm.weight": norm.weight.detach(), key + ".norm.bias": norm.bias.detach(), key + ".to_qkv.weight": to_qkv.weight.detach(),
          key + ".to_out.weight": to_out.weight.detach(),
          key + ".rel_pos.inv_freq": (1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64)))}          # SinusoidalEmbedding.cs:52-54
    ours = TorchSNAC.__new__(TorchSNAC)                       # only local_mha is exercised: no full model behind it
    ours.sd, ours.attn = sd, window
    x = torch.randn(2, C, T)
    out = {"inv_freq_max_abs": float((sd[key + ".rel_pos.inv_freq"] - rope.inv_freq).abs().max())}
    with torch.inference_mode():
        y = ours.local_mha(x, key)
        h = norm(x.transpose(1, 2))                                           # [B, T, C]
        q, k, v = to_qkv(h).chunk(3, dim=-1)
        q, k, v = (t.reshape(2, T, heads, 64).transpose(1, 2) for t in (q, k, v))   # [B, heads, T, 64]
        pos = (torch.arange(T) % window)[None].expand(2, T)
        cos, sin = rope(x, pos)
        q, k = apply_rotary_pos_emb(q, k, cos, sin)
        scores = torch.einsum("bhtd,bhsd->bhts", q, k) / 8.0
        blk = torch.arange(T) // window
        scores = scores.masked_fill(blk[:, None] != blk[None, :], float("-inf"))
        a = torch.einsum("bhts,bhsd->bhtd", scores.softmax(-1), v)
        y_ind = to_out(a.transpose(1, 2).reshape(2, T, C)).transpose(1, 2) + x
        out["max_abs"] = float((y - y_ind).abs().max())
        out["scale"] = float(y_ind.abs().max())
        # and the check has teeth: the same composition WITHOUT the rotary embedding differs at O(0.1 .. 1)
        q0, k0, _ = (t.reshape(2, T, heads, 64).transpose(1, 2) for t in to_qkv(h).chunk(3, dim=-1))
        s2 = (torch.einsum("bhtd,bhsd->bhts", q0, k0) / 8.0).masked_fill(blk[:, None] != blk[None, :], float("-inf"))
        y2 = to_out(torch.einsum("bhts,bhsd->bhtd", s2.softmax(-1), v).transpose(1, 2).reshape(2, T, C)).transpose(1, 2) + x
        out["without_rotary_max_abs"] = float((y2 - y_ind).abs().max())
    return out


def crosscheck_snac_blocks(seed=0):
    """SNAC's convolutional graph and quantizer stage (Modules/SNAC/{Encoder, EncoderBlock, ResidualUnit, Snake1d, WNConv1d,
    WNConvTranspose1d, Decoder, DecoderBlock, NoiseBlock, VectorQuantizer}.cs) as restated in oracle/torch_ref/snac.py, against an
    INDEPENDENT composition that shares no code with it: torch.nn.Conv1d / ConvTranspose1d modules under PyTorch's own
    `parametrizations.weight_norm`, HF transformers' `Snake1d`, `nn.Sequential` containers whose state-dict names ARE the reference's
    TorchSharp names, and HF's `DacVectorQuantize` (upstream SNAC's VectorQuantize is DAC's) with avg_pool / repeat_interleave around
    it.  Agreement pins paddings, strides, output_padding, depthwise groups, the channel schedule, the noise block, the key names and
    -- with the port's deviation D1 undone -- the quantizer stage; with D1 as in the reference the codes differ, as they must."""
    from torch import nn
    import torch.nn.functional as F
    from torch.nn.utils.parametrizations import weight_norm as wn
    from transformers import DacConfig
    from transformers.models.dac.modeling_dac import Snake1d, DacVectorQuantize
    from neuralcodecs_amd.config import SNACConfig
    from oracle.torch_ref.snac import TorchSNAC
    torch.manual_seed(seed)
    cfg = SNACConfig(sampling_rate=16000, encoder_dim=16, encoder_rates=(2, 3, 4), decoder_dim=128, decoder_rates=(4, 3, 2),
                     attn_window_size=None, codebook_size=64, codebook_dim=8, vq_strides=(2, 1), noise=True, depthwise=True)

    class Unit(nn.Module):
        def __init__(self, C, dil):
            super().__init__()
            self.block = nn.Sequential(Snake1d(C), wn(nn.Conv1d(C, C, 7, dilation=dil, padding=3 * dil, groups=C)), Snake1d(C), wn(nn.Conv1d(C, C, 1)))

        def forward(self, x):
            return x + self.block(x)

    class EncBlock(nn.Module):
        def __init__(self, C, s):
            super().__init__()
            self.block = nn.Sequential(Unit(C, 1), Unit(C, 3), Unit(C, 9), Snake1d(C), wn(nn.Conv1d(C, 2 * C, 2 * s, stride=s, padding=-(-s // 2))))

        def forward(self, x):
            return self.block(x)

    class Noise(nn.Module):
        def __init__(self, C):
            super().__init__()
            self.linear = wn(nn.Conv1d(C, C, 1, bias=False))
            self.nz = None

        def forward(self, x):
            return x + self.nz * self.linear(x)

    class DecBlock(nn.Module):
        def __init__(self, Cin, Cout, s):
            super().__init__()
            self.block = nn.Sequential(Snake1d(Cin), wn(nn.ConvTranspose1d(Cin, Cout, 2 * s, stride=s, padding=-(-s // 2), output_padding=s % 2)),
                                       Noise(Cout), Unit(Cout, 1), Unit(Cout, 3), Unit(Cout, 9))

        def forward(self, x):
            return self.block(x)

    d, D = cfg.encoder_dim, cfg.resolved_latent_dim
    enc_layers = [wn(nn.Conv1d(1, d, 7, padding=3))]
    for s_ in cfg.encoder_rates:
        enc_layers.append(EncBlock(d, s_))
        d *= 2
    enc_layers.append(wn(nn.Conv1d(d, d, 7, padding=3, groups=d)))
    dec_layers = [wn(nn.Conv1d(D, D, 7, padding=3, groups=D)), wn(nn.Conv1d(D, cfg.decoder_dim, 1))]
    ch = cfg.decoder_dim
    for bi, s_ in enumerate(cfg.decoder_rates):
        dec_layers.append(DecBlock(ch // (1 << bi), ch // (1 << (bi + 1)), s_))
    out_dim = ch // (1 << len(cfg.decoder_rates))
    dec_layers += [Snake1d(out_dim), wn(nn.Conv1d(out_dim, 1, 7, padding=3)), nn.Tanh()]

    class Enc(nn.Module):
        def __init__(self):
            super().__init__()
            self.block = nn.Sequential(*enc_layers)

    class Dec(nn.Module):
        def __init__(self):
            super().__init__()
            self.model = nn.Sequential(*dec_layers)

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.encoder, self.decoder = Enc(), Dec()

    net = Net()
    with torch.no_grad():
        for n_, p_ in net.named_parameters():
            if n_.endswith("alpha"):
                p_.uniform_(0.5, 1.5)
            elif n_.endswith("original0"):
                p_.uniform_(0.6, 1.2)
            elif n_.endswith("bias"):
                p_.normal_(0, 0.05)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    # quantizer stages: weight-normed 1x1 projections + a codebook each (HF's DacVectorQuantize holds plain convolutions: it gets the
    # EFFECTIVE weights PyTorch's weight_norm produces)
    vqs = []
    for i, _ in enumerate(cfg.vq_strides):
        ip, op = wn(nn.Conv1d(D, cfg.codebook_dim, 1)), wn(nn.Conv1d(cfg.codebook_dim, D, 1))
        cb = nn.Embedding(cfg.codebook_size, cfg.codebook_dim)
        hfq = DacVectorQuantize(DacConfig(hidden_size=D, codebook_dim=cfg.codebook_dim, codebook_size=cfg.codebook_size))
        with torch.no_grad():
            hfq.in_proj.weight.copy_(ip.weight); hfq.in_proj.bias.copy_(ip.bias)
            hfq.out_proj.weight.copy_(op.weight); hfq.out_proj.bias.copy_(op.bias)
            hfq.codebook.weight.copy_(cb.weight)
        for nm, mod in (("in_proj", ip), ("out_proj", op)):
            for k, v in mod.state_dict().items():
                sd[f"quantizer.quantizers.{i}.{nm}.{k}"] = v.detach().clone()
        sd[f"quantizer.quantizers.{i}.codebook.weight"] = cb.weight.detach().clone()
        vqs.append(hfq)
    ours = TorchSNAC(cfg, {k: v.numpy() for k, v in sd.items()})
    ours.upstream = {"D3", "D4"}                 # the port's weight-norm epsilon and exact Snake reciprocal undone (arithmetic, not structure)
    x = 0.3 * torch.randn(2, 1, cfg.hop_length * 2 * 12)
    out = {}
    with torch.inference_mode():
        z = ours.encoder(x)
        z_ind = net.encoder.block(x)
        out["encoder_max_abs"], out["encoder_scale"] = float((z - z_ind).abs().max()), float(z_ind.abs().max())
        noises = []
        T = z.shape[-1]
        for bi, s_ in enumerate(cfg.decoder_rates):
            T = (T - 1) * s_ - 2 * (-(-s_ // 2)) + 2 * s_ + s_ % 2
            noises.append(torch.randn(2, 1, T))
        for blk, nz in zip([m for m in net.decoder.model if isinstance(m, DecBlock)], noises):
            blk.block[2].nz = nz
        y = ours.decoder(z, noises)
        y_ind = net.decoder.model(z)
        out["decoder_max_abs"], out["decoder_len"] = float((y - y_ind).abs().max()), [int(y.shape[-1]), int(y_ind.shape[-1])]
        out["decoder_pre_tanh_scale"] = float(net.decoder.model[:-1](z).abs().max())
        # quantizer stages on the encoder output, stride by stride
        res_o, res_i = z.clone(), z.clone()
        eq, n_codes, zq_err = 0, 0, 0.0
        ours.upstream = {"D1", "D3", "D4"}
        for i, s_ in enumerate(cfg.vq_strides):
            zq_o, idx_o, _ = ours.vq(res_o, i)
            zi = F.avg_pool1d(res_i, s_, s_) if s_ > 1 else res_i
            q, _, _, idx_i, _ = vqs[i](zi)
            zq_i = q.repeat_interleave(s_, dim=-1) if s_ > 1 else q
            same = (idx_o == idx_i)
            eq += int(same.sum()); n_codes += idx_o.numel()
            fr = same.repeat_interleave(s_, dim=-1)[:, None, :].expand_as(zq_o)
            if fr.any():
                zq_err = max(zq_err, float((zq_o - zq_i)[fr].abs().max()))
            res_o, res_i = res_o - zq_o, res_i - zq_i
        out["codes_equal_frac"], out["zq_max_abs_same_frames"] = eq / n_codes, zq_err
        ours.upstream = {"D3", "D4"}
        _, idx_ref, _ = ours.vq(z, len(cfg.vq_strides) - 1)
        _, _, _, idx_up, _ = vqs[-1](z)
        out["codes_equal_frac_with_D1_as_in_reference"] = float((idx_ref == idx_up).float().mean())
    return out


if __name__ == "__main__":
    print("dac    ", crosscheck_dac())
    print("encodec", crosscheck_encodec())
    print("encodec48", crosscheck_encodec48())
    print("snac_mha", crosscheck_snac_localmha())
    print("snac_blocks", crosscheck_snac_blocks())
