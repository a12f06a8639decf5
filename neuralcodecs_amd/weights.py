"""Weight containers for the MI355X codec engine.

Two things live here:

1. The *weight blob* ("NCWB0001"): the flat, mmap-able file/memory image that
   ``nc_codec_load_weights{,_mem}`` (include/nc_mi355x.h) consumes.  Tensor names
   follow the TorchSharp state-dict keys of the reference so that a real
   checkpoint can be converted 1:1 (SURVEY 2.4;
   NeuralCodecs.Torch/Config/DAC/StateDictNameConverter.cs:274-340 for DAC,
   Modules/SNAC/WNConv1d.cs:66-70 for SNAC's ``parametrizations.weight.original{0,1}``,
   Modules/Encodec/SConv1d.cs:110-128 for Encodec).

2. A seeded *synthetic* state-dict generator (there are no checkpoints offline).
   It is counter-based (SplitMix64 over (seed, fnv1a(name), element index)) and
   uses only integer arithmetic plus one IEEE multiply per element, so the bytes
   are identical on every machine / numpy build.  The scale of each tensor is
   chosen so activations stay O(1) through ~60 layers (variance-preserving
   fan-in scaling) and the final tanh does not saturate -- the reference's own
   initialisers (trunc_normal std 0.02, Modules/DAC/WNConv1d.cs:107-108) are for
   training and are irrelevant to inference parity.
"""
from __future__ import annotations

import io
import struct
from collections import OrderedDict
from typing import Dict, Iterable, List, Tuple

import numpy as np

from .config import DACConfig, EncodecConfig, SNACConfig

MAGIC = b"NCWB0001"
_DT = {np.dtype(np.float32): 0, np.dtype(np.int64): 1}
_DT_INV = {0: np.float32, 1: np.int64}

# --------------------------------------------------------------------------- blob


def save_blob(tensors: "OrderedDict[str, np.ndarray]") -> bytes:
    """Serialise an ordered name->array map into the NCWB0001 image."""
    index = io.BytesIO()
    offs = 0
    metas = []
    for name, arr in tensors.items():
        arr = np.ascontiguousarray(arr)
        if arr.dtype not in _DT:
            raise ValueError(f"{name}: unsupported dtype {arr.dtype}")
        nb = arr.nbytes
        metas.append((name, arr, offs, nb))
        offs += (nb + 63) & ~63
    for name, arr, off, nb in metas:
        nm = name.encode()
        index.write(struct.pack("<H", len(nm)))
        index.write(nm)
        index.write(struct.pack("<BB", _DT[arr.dtype], arr.ndim))
        for d in arr.shape:
            index.write(struct.pack("<Q", d))
        index.write(struct.pack("<QQ", off, nb))
    idx = index.getvalue()
    head = MAGIC + struct.pack("<QQ", len(metas), len(idx))
    pre = len(head) + len(idx)
    pad = (-pre) % 64
    out = io.BytesIO()
    out.write(head)
    out.write(idx)
    out.write(b"\0" * pad)
    for name, arr, off, nb in metas:
        out.write(arr.tobytes())
        out.write(b"\0" * (((nb + 63) & ~63) - nb))
    return out.getvalue()


def load_blob(buf: bytes) -> "OrderedDict[str, np.ndarray]":
    if buf[:8] != MAGIC:
        raise ValueError("not an NCWB0001 weight blob")
    n, idx_len = struct.unpack_from("<QQ", buf, 8)
    p = 24
    data0 = (24 + idx_len + 63) & ~63
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for _ in range(n):
        (ln,) = struct.unpack_from("<H", buf, p); p += 2
        name = buf[p:p + ln].decode(); p += ln
        dt, nd = struct.unpack_from("<BB", buf, p); p += 2
        dims = struct.unpack_from("<" + "Q" * nd, buf, p); p += 8 * nd
        off, nb = struct.unpack_from("<QQ", buf, p); p += 16
        out[name] = np.frombuffer(buf, dtype=_DT_INV[dt], count=nb // np.dtype(_DT_INV[dt]).itemsize,
                                  offset=data0 + off).reshape(dims).copy()
    return out


# --------------------------------------------------------------------------- counter-based RNG

_U64 = np.uint64
_GOLD = _U64(0x9E3779B97F4A7C15)


def _splitmix64(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = z + _GOLD
        z = (z ^ (z >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U64(27))) * _U64(0x94D049BB133111EB)
        return z ^ (z >> _U64(31))


def _fnv1a(name: str) -> int:
    h = 0xCBF29CE484222325
    for c in name.encode():
        h = ((h ^ c) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _bits(seed: int, name: str, n: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([(seed * 0x9E3779B97F4A7C15 + _fnv1a(name)) & 0xFFFFFFFFFFFFFFFF], dtype=_U64))[0]
        return _splitmix64(base + np.arange(n, dtype=_U64) * _U64(0xD1342543DE82EF95))


def uniform01(seed: int, name: str, n: int) -> np.ndarray:
    """float64 in [0,1): top 53 bits * 2^-53 (exact)."""
    return (_bits(seed, name, n) >> _U64(11)).astype(np.float64) * (2.0 ** -53)


_IH_STD = float(np.sqrt((65536.0 ** 2 - 1.0) / 3.0))  # std of the sum of four independent 16-bit uniforms


def approx_normal(seed: int, name: str, n: int) -> np.ndarray:
    """Irwin-Hall(4) on 16-bit fields, centred and scaled to unit variance (float64, exact integer core)."""
    b = _bits(seed, name, n)
    m = _U64(0xFFFF)
    s = ((b & m) + ((b >> _U64(16)) & m) + ((b >> _U64(32)) & m) + ((b >> _U64(48)) & m)).astype(np.int64) - 131070
    return s.astype(np.float64) * (1.0 / _IH_STD)


# --------------------------------------------------------------------------- synthetic state dicts


def _wn_pair(sd, seed, prefix, shape, fan_in, gain, norm_axes=(1, 2), bias_len=None, g_lo=0.8, g_hi=1.2):
    """weight_v / weight_g / bias of one weight-normalised conv (shapes as in the reference's WNConv*)."""
    n = int(np.prod(shape))
    v = (approx_normal(seed, prefix + ".weight_v", n) * (1.0 / np.sqrt(fan_in))).astype(np.float32).reshape(shape)
    norm = np.sqrt((v.astype(np.float64) ** 2).sum(axis=norm_axes, keepdims=True))
    gshape = norm.shape
    u = uniform01(seed, prefix + ".weight_g", int(np.prod(gshape))).reshape(gshape)
    g = (norm * (g_lo + (g_hi - g_lo) * u) * gain).astype(np.float32)
    sd[prefix + ".weight_v"] = v
    sd[prefix + ".weight_g"] = g
    if bias_len is not None:
        bound = 1.0 / np.sqrt(fan_in)
        b = ((uniform01(seed, prefix + ".bias", bias_len) * 2.0 - 1.0) * bound * 0.5).astype(np.float32)
        sd[prefix + ".bias"] = b


def _alpha(sd, seed, name, c):
    a = (0.5 + 1.5 * uniform01(seed, name, c)).astype(np.float32)
    # a few exact zeros exercise the where(alpha==0, x, ...) branch (Snake1d.cs:52)
    z = (_bits(seed, name + "#z", c) % _U64(41)) == 0
    a[z] = 0.0
    sd[name] = a.reshape(1, c, 1)


def dac_synthetic_state_dict(cfg: DACConfig, seed: int = 42) -> "OrderedDict[str, np.ndarray]":
    """Seeded synthetic DAC weights under the reference's TorchSharp key names."""
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    G_RES7, G_RES1, G_MAIN = 0.9, 0.35, 0.9
    d = cfg.encoder_dim
    # encoder (Modules/DAC/Encoder.cs:21-58)
    _wn_pair(sd, seed, "encoder.block.0", (d, 1, 7), 7, 1.6, bias_len=d)
    for bi, s in enumerate(cfg.encoder_rates):
        cin = d
        d *= 2
        p = f"encoder.block.{bi + 1}"
        for ui in range(3):
            q = f"{p}.block.{ui}"
            _alpha(sd, seed, f"{q}.block.0.alpha", cin)
            _wn_pair(sd, seed, f"{q}.block.1", (cin, cin, 7), cin * 7, G_RES7, bias_len=cin)
            _alpha(sd, seed, f"{q}.block.2.alpha", cin)
            _wn_pair(sd, seed, f"{q}.block.3", (cin, cin, 1), cin, G_RES1, bias_len=cin)
        _alpha(sd, seed, f"{p}.block.3.alpha", cin)
        _wn_pair(sd, seed, f"{p}.block.4", (d, cin, 2 * s), cin * 2 * s, G_MAIN, bias_len=d)
    latent = cfg.resolved_latent_dim
    _alpha(sd, seed, f"encoder.block.{len(cfg.encoder_rates) + 1}.alpha", d)
    _wn_pair(sd, seed, f"encoder.block.{len(cfg.encoder_rates) + 2}", (latent, d, 3), d * 3, G_MAIN, bias_len=latent)
    # quantizer (Modules/DAC/VectorQuantizer.cs:41-50)
    for i in range(cfg.n_codebooks):
        p = f"quantizer.quantizers.{i}"
        _wn_pair(sd, seed, f"{p}.in_proj", (cfg.codebook_dim, latent, 1), latent, 1.0, bias_len=cfg.codebook_dim)
        _wn_pair(sd, seed, f"{p}.out_proj", (latent, cfg.codebook_dim, 1), cfg.codebook_dim, 0.45, bias_len=latent)
        sd[f"{p}.codebook.weight"] = (approx_normal(seed, f"{p}.codebook.weight", cfg.codebook_size * cfg.codebook_dim)
                                      * 0.8).astype(np.float32).reshape(cfg.codebook_size, cfg.codebook_dim)
    # decoder (Modules/DAC/Decoder.cs:22-58, DecoderBlock.cs:20-44)
    ch = cfg.decoder_dim
    _wn_pair(sd, seed, "decoder.model.0", (ch, latent, 7), latent * 7, G_MAIN, bias_len=ch)
    out_dim = ch
    for bi, s in enumerate(cfg.decoder_rates):
        in_dim = ch // (1 << bi)
        out_dim = ch // (1 << (bi + 1))
        p = f"decoder.model.{bi + 1}"
        _alpha(sd, seed, f"{p}.block.0.alpha", in_dim)
        # conv-transpose weight [Cin, Cout, K]; weight_g is per-Cin slice; each output sees K/stride taps
        _wn_pair(sd, seed, f"{p}.block.1", (in_dim, out_dim, 2 * s), in_dim * 2, G_MAIN, bias_len=out_dim)
        for ui in range(3):
            q = f"{p}.block.{ui + 2}"
            _alpha(sd, seed, f"{q}.block.0.alpha", out_dim)
            _wn_pair(sd, seed, f"{q}.block.1", (out_dim, out_dim, 7), out_dim * 7, G_RES7, bias_len=out_dim)
            _alpha(sd, seed, f"{q}.block.2.alpha", out_dim)
            _wn_pair(sd, seed, f"{q}.block.3", (out_dim, out_dim, 1), out_dim, G_RES1, bias_len=out_dim)
    n = len(cfg.decoder_rates)
    _alpha(sd, seed, f"decoder.model.{n + 1}.alpha", out_dim)
    _wn_pair(sd, seed, f"decoder.model.{n + 2}", (1, out_dim, 7), out_dim * 7, 0.25, bias_len=1)
    return sd


def _wn_param(sd, seed, prefix, shape, fan_in, gain, bias_len=None, g_lo=0.8, g_hi=1.2):
    """SNAC / Encodec flavour: `parametrizations.weight.original0` (g, [d0,1,1]) / `original1` (v) / bias
    (Modules/SNAC/WNConv1d.cs:66-70)."""
    tmp: "OrderedDict[str, np.ndarray]" = OrderedDict()
    _wn_pair(tmp, seed, prefix, shape, fan_in, gain, bias_len=bias_len, g_lo=g_lo, g_hi=g_hi)
    sd[prefix + ".parametrizations.weight.original0"] = tmp[prefix + ".weight_g"].reshape(shape[0], 1, 1)
    sd[prefix + ".parametrizations.weight.original1"] = tmp[prefix + ".weight_v"]
    if bias_len is not None:
        sd[prefix + ".bias"] = tmp[prefix + ".bias"]


def _local_mha(sd, seed, prefix, dim):
    """LocalMHA parameters (Modules/SNAC/LocalMHA.cs:46-70): LayerNorm, bias-free qkv / out projections, inv_freq buffer."""
    sd[prefix + ".norm.weight"] = (0.8 + 0.4 * uniform01(seed, prefix + ".norm.weight", dim)).astype(np.float32)
    sd[prefix + ".norm.bias"] = (approx_normal(seed, prefix + ".norm.bias", dim) * 0.02).astype(np.float32)
    sd[prefix + ".to_qkv.weight"] = (approx_normal(seed, prefix + ".to_qkv.weight", 3 * dim * dim) / np.sqrt(dim)
                                     ).astype(np.float32).reshape(3 * dim, dim)
    sd[prefix + ".to_out.weight"] = (approx_normal(seed, prefix + ".to_out.weight", dim * dim) * (0.5 / np.sqrt(dim))
                                     ).astype(np.float32).reshape(dim, dim)
    # SinusoidalEmbedding.cs:44-47: inv_freq = 1 / 10000 ** (arange(0, 64, 2) / 64), float32 arithmetic
    power = (np.arange(0, 64, 2, dtype=np.float32) / np.float32(64)).astype(np.float32)
    sd[prefix + ".rel_pos.inv_freq"] = (np.float32(1.0) / np.power(np.float32(10000.0), power, dtype=np.float32)).astype(np.float32)


def snac_synthetic_state_dict(cfg: SNACConfig, seed: int = 42) -> "OrderedDict[str, np.ndarray]":
    """Seeded synthetic SNAC weights under the reference's TorchSharp key names."""
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    G_DW, G_RES1, G_MAIN = 1.0, 0.35, 0.9
    d = cfg.encoder_dim
    _wn_param(sd, seed, "encoder.block.0", (d, 1, 7), 7, 1.6, bias_len=d)
    for bi, s in enumerate(cfg.encoder_rates):
        cin = d
        d *= 2
        p = f"encoder.block.{bi + 1}"
        for ui in range(3):
            q = f"{p}.block.{ui}"
            _alpha(sd, seed, f"{q}.block.0.alpha", cin)
            if cfg.depthwise:
                _wn_param(sd, seed, f"{q}.block.1", (cin, 1, 7), 7, G_DW, bias_len=cin)
            else:
                _wn_param(sd, seed, f"{q}.block.1", (cin, cin, 7), cin * 7, 0.9, bias_len=cin)
            _alpha(sd, seed, f"{q}.block.2.alpha", cin)
            _wn_param(sd, seed, f"{q}.block.3", (cin, cin, 1), cin, G_RES1, bias_len=cin)
        _alpha(sd, seed, f"{p}.block.3.alpha", cin)
        _wn_param(sd, seed, f"{p}.block.4", (d, cin, 2 * s), cin * 2 * s, G_MAIN, bias_len=d)
    n = len(cfg.encoder_rates) + 1
    if cfg.attn_window_size:
        _local_mha(sd, seed, f"encoder.block.{n}", d)
        n += 1
    if cfg.depthwise:
        _wn_param(sd, seed, f"encoder.block.{n}", (d, 1, 7), 7, G_DW, bias_len=d)
    else:
        _wn_param(sd, seed, f"encoder.block.{n}", (d, d, 7), d * 7, G_MAIN, bias_len=d)
    latent = cfg.resolved_latent_dim
    for i in range(len(cfg.vq_strides)):
        p = f"quantizer.quantizers.{i}"
        _wn_param(sd, seed, f"{p}.in_proj", (cfg.codebook_dim, latent, 1), latent, 1.0, bias_len=cfg.codebook_dim)
        _wn_param(sd, seed, f"{p}.out_proj", (latent, cfg.codebook_dim, 1), cfg.codebook_dim, 0.45, bias_len=latent)
        sd[f"{p}.codebook.weight"] = (approx_normal(seed, f"{p}.codebook.weight", cfg.codebook_size * cfg.codebook_dim)
                                      * 0.8).astype(np.float32).reshape(cfg.codebook_size, cfg.codebook_dim)
    ch = cfg.decoder_dim
    if cfg.depthwise:
        _wn_param(sd, seed, "decoder.model.0", (latent, 1, 7), 7, G_DW, bias_len=latent)
        _wn_param(sd, seed, "decoder.model.1", (ch, latent, 1), latent, G_MAIN, bias_len=ch)
        n = 2
    else:
        _wn_param(sd, seed, "decoder.model.0", (ch, latent, 7), latent * 7, G_MAIN, bias_len=ch)
        n = 1
    if cfg.attn_window_size:
        _local_mha(sd, seed, f"decoder.model.{n}", ch)
        n += 1
    out_dim = ch
    for bi, s in enumerate(cfg.decoder_rates):
        in_dim = ch // (1 << bi)
        out_dim = ch // (1 << (bi + 1))
        p = f"decoder.model.{n}"
        _alpha(sd, seed, f"{p}.block.0.alpha", in_dim)
        _wn_param(sd, seed, f"{p}.block.1", (in_dim, out_dim, 2 * s), in_dim * 2, G_MAIN, bias_len=out_dim)
        k = 2
        if cfg.noise:
            _wn_param(sd, seed, f"{p}.block.2.linear", (out_dim, out_dim, 1), out_dim, 0.15)
            k = 3
        for ui in range(3):
            q = f"{p}.block.{k + ui}"
            _alpha(sd, seed, f"{q}.block.0.alpha", out_dim)
            if cfg.depthwise:
                _wn_param(sd, seed, f"{q}.block.1", (out_dim, 1, 7), 7, G_DW, bias_len=out_dim)
            else:
                _wn_param(sd, seed, f"{q}.block.1", (out_dim, out_dim, 7), out_dim * 7, 0.9, bias_len=out_dim)
            _alpha(sd, seed, f"{q}.block.2.alpha", out_dim)
            _wn_param(sd, seed, f"{q}.block.3", (out_dim, out_dim, 1), out_dim, G_RES1, bias_len=out_dim)
        n += 1
    _alpha(sd, seed, f"decoder.model.{n}.alpha", out_dim)
    _wn_param(sd, seed, f"decoder.model.{n + 1}", (1, out_dim, 7), out_dim * 7, 0.25, bias_len=1)
    return sd


def snac_noise(cfg: SNACConfig, batch: int, frames: int, seed: int = 99):
    """The NoiseBlock inputs of one decode ([B,1,T_i] per decoder block), from the counter-based generator -- the reference
    draws them with randn at inference (NoiseBlock.cs:41, deviation D8), so parity tests inject them."""
    out = []
    T = frames
    for bi, s in enumerate(cfg.decoder_rates):
        T = (T - 1) * s - 2 * (-(-s // 2)) + 2 * s + (s % 2)
        out.append(approx_normal(seed + bi, "snac.noise", batch * T).astype(np.float32).reshape(batch, 1, T))
    return out


def _enc_conv(sd, seed, cfg, prefix, shape, fan_in, gain, n_out):
    """One SConv1d / SConvTranspose1d parameter set (Modules/Encodec/SConv1d.cs:110-128): plain conv + GroupNorm affine for
    norm == time_group_norm, weight_v / weight_g for weight_norm."""
    if cfg.norm == "time_group_norm":
        n = int(np.prod(shape))
        sd[prefix + ".conv.weight"] = (approx_normal(seed, prefix + ".conv.weight", n) * (gain / np.sqrt(fan_in))
                                       ).astype(np.float32).reshape(shape)
        sd[prefix + ".conv.bias"] = ((uniform01(seed, prefix + ".conv.bias", n_out) * 2.0 - 1.0) * (0.5 / np.sqrt(fan_in))
                                     ).astype(np.float32)
        sd[prefix + ".norm.weight"] = (0.8 + 0.4 * uniform01(seed, prefix + ".norm.weight", n_out)).astype(np.float32)
        sd[prefix + ".norm.bias"] = (approx_normal(seed, prefix + ".norm.bias", n_out) * 0.05).astype(np.float32)
    else:
        tmp: "OrderedDict[str, np.ndarray]" = OrderedDict()
        _wn_pair(tmp, seed, prefix + ".conv", shape, fan_in, gain, bias_len=n_out)
        sd[prefix + ".conv.weight_v"] = tmp[prefix + ".conv.weight_v"]
        sd[prefix + ".conv.weight_g"] = tmp[prefix + ".conv.weight_g"].reshape(shape[0], 1, 1)
        sd[prefix + ".conv.bias"] = tmp[prefix + ".conv.bias"]


def encodec_synthetic_state_dict(cfg: EncodecConfig, seed: int = 42) -> "OrderedDict[str, np.ndarray]":
    """Seeded synthetic Encodec weights under the reference's TorchSharp key names (SEANetEncoder.cs:37-148,
    SEANetDecoder.cs:40-153, SLSTM.cs:31, EuclideanCodebook.cs:60-66)."""
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    nf, dim = cfg.n_filters, cfg.dimension
    rk = cfg.residual_kernel_size

    def resblock(p, d):
        h = d // cfg.compress
        _enc_conv(sd, seed, cfg, p + ".block.1", (h, d, rk), d * rk, 1.0, h)
        _enc_conv(sd, seed, cfg, p + ".block.3", (d, h, 1), h, 0.6, d)
        _enc_conv(sd, seed, cfg, p + ".shortcut", (d, d, 1), d, 0.8, d)

    def lstm(p, d):
        bound = 1.0 / np.sqrt(d)
        for l in range(cfg.lstm_layers):
            for nm, shape in (("weight_ih", (4 * d, d)), ("weight_hh", (4 * d, d)), ("bias_ih", (4 * d,)), ("bias_hh", (4 * d,))):
                k = f"{p}.lstm.{nm}_l{l}"
                sd[k] = ((uniform01(seed, k, int(np.prod(shape))) * 2.0 - 1.0) * bound).astype(np.float32).reshape(shape)

    _enc_conv(sd, seed, cfg, "encoder.layers.0", (nf, cfg.channels, cfg.kernel_size), cfg.channels * cfg.kernel_size, 1.5, nf)
    n, mult = 1, 1
    for r in reversed(cfg.ratios):
        d = mult * nf
        resblock(f"encoder.layers.{n}", d)
        _enc_conv(sd, seed, cfg, f"encoder.layers.{n + 2}", (2 * d, d, 2 * r), d * 2 * r, 1.0, 2 * d)
        n += 3
        mult *= 2
    lstm(f"encoder.layers.{n}", mult * nf)
    _enc_conv(sd, seed, cfg, f"encoder.layers.{n + 2}", (dim, mult * nf, cfg.last_kernel_size), mult * nf * cfg.last_kernel_size, 1.0, dim)
    import math
    n_q = int(1000 * max(cfg.target_bandwidths) / (math.ceil(cfg.sampling_rate / cfg.hop_length) * 10))
    for i in range(n_q):
        k = f"quantizer.layers.{i}.codebook.embed"
        sd[k] = (approx_normal(seed, k, cfg.codebook_size * dim) * (0.9 * 0.75 ** i)).astype(np.float32).reshape(cfg.codebook_size, dim)
    _enc_conv(sd, seed, cfg, "decoder.layers.0", (mult * nf, dim, cfg.kernel_size), dim * cfg.kernel_size, 1.0, mult * nf)
    lstm("decoder.layers.1", mult * nf)
    n = 2
    for r in cfg.ratios:
        d = mult * nf
        # conv-transpose weight [Cin, Cout, K]; each output sample sees K/stride = 2 taps per input channel
        _enc_conv(sd, seed, cfg, f"decoder.layers.{n + 1}", (d, d // 2, 2 * r), d * 2, 1.0, d // 2)
        resblock(f"decoder.layers.{n + 2}", d // 2)
        n += 3
        mult //= 2
    _enc_conv(sd, seed, cfg, f"decoder.layers.{n + 1}", (cfg.channels, nf, cfg.last_kernel_size), nf * cfg.last_kernel_size, 0.5,
              cfg.channels)
    return sd


def _parabolic_sine(phase_num: np.ndarray, denom: int) -> np.ndarray:
    """Parabolic sine approximation from an exact integer phase (only IEEE +,-,*,/: bit-reproducible)."""
    p = (phase_num % denom).astype(np.float64) / float(denom)      # [0,1)
    x = 2.0 * p - 1.0                                               # [-1,1)
    return -4.0 * x * (1.0 - np.abs(x))                             # ~sin(2*pi*p)


def synthetic_pcm(batch: int, channels: int, length: int, sample_rate: int, seed: int = 1234) -> np.ndarray:
    """Deterministic test audio: 0.1*N(0,1)-like noise + a fixed three-partial tone, clipped to [-1,1].

    Integer phases + basic IEEE ops only (no libm), so the float32 bytes are identical on the
    container that generated the golden fixtures and on the GPU box.  Shape [batch, channels, length].
    """
    out = np.empty((batch, channels, length), dtype=np.float32)
    n = np.arange(length, dtype=np.int64)
    for b in range(batch):
        for c in range(channels):
            nz = approx_normal(seed + b, f"pcm.{c}", length) * 0.1
            f0 = 110 * (1 + (b % 7)) + 13 * c
            tone = 0.25 * _parabolic_sine(n * f0, sample_rate) \
                + 0.15 * _parabolic_sine(n * (f0 * 3 - 7) + sample_rate // 5, sample_rate) \
                + 0.08 * _parabolic_sine(n * (f0 * 7 + 3) + sample_rate // 3, sample_rate)
            out[b, c] = np.clip(nz + tone, -1.0, 1.0).astype(np.float32)
    return out


def tie_codebooks(sd, dead_every: int = 7):
    """Adversarial quantizer weights (in place; returns sd): every codebook's upper half DUPLICATES its lower half row for row, so the two
    best distances of EVERY frame are an exact tie, and every `dead_every`-th row of the lower half is pushed far away (a dead code: never
    the nearest).  The reference's argmin (ATen: DAC/VectorQuantizer.cs:121, SNAC/VectorQuantizer.cs:137, EuclideanCodebook.cs:181) returns
    the FIRST index of a tie -- all emitted codes must lie in the lower half and avoid the dead rows."""
    for k in list(sd):
        if k.endswith(".codebook.weight") or k.endswith(".codebook.embed"):
            w = np.array(sd[k], dtype=np.float32, copy=True)
            half = w.shape[0] // 2
            if dead_every > 0:
                w[0:half:dead_every] += np.float32(64.0)
            w[half:2 * half] = w[:half]
            sd[k] = w
    return sd
