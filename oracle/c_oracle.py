"""ORACLE (test infrastructure): ctypes binding of oracle/c/libnc_ref.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "c", "libnc_ref.so")
_lib = None

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, "c", f) for f in os.listdir(os.path.join(_HERE, "c")) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class RefSnacConfig(C.Structure):
    _fields_ = [("sample_rate", C.c_int), ("encoder_dim", C.c_int), ("n_enc_rates", C.c_int), ("enc_rates", C.c_int * 8),
                ("decoder_dim", C.c_int), ("n_dec_rates", C.c_int), ("dec_rates", C.c_int * 8), ("latent_dim", C.c_int),
                ("attn_window", C.c_int), ("codebook_size", C.c_int), ("codebook_dim", C.c_int), ("n_vq", C.c_int),
                ("vq_strides", C.c_int * 8), ("noise", C.c_int), ("depthwise", C.c_int)]


class RefEncodecConfig(C.Structure):
    _fields_ = [("sample_rate", C.c_int), ("channels", C.c_int), ("dimension", C.c_int), ("n_filters", C.c_int), ("n_ratios", C.c_int),
                ("ratios", C.c_int * 8), ("lstm_layers", C.c_int), ("compress", C.c_int), ("kernel_size", C.c_int),
                ("last_kernel_size", C.c_int), ("residual_kernel_size", C.c_int), ("group_norm", C.c_int), ("causal", C.c_int),
                ("normalize", C.c_int), ("codebook_size", C.c_int), ("n_q_total", C.c_int)]


class RefDacConfig(C.Structure):
    _fields_ = [("sample_rate", C.c_int), ("encoder_dim", C.c_int), ("n_enc_rates", C.c_int), ("enc_rates", C.c_int * 8),
                ("decoder_dim", C.c_int), ("n_dec_rates", C.c_int), ("dec_rates", C.c_int * 8), ("latent_dim", C.c_int),
                ("n_codebooks", C.c_int), ("codebook_size", C.c_int), ("codebook_dim", C.c_int)]


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.ref_dac_create.restype = C.c_void_p
        L.ref_dac_create.argtypes = [C.POINTER(RefDacConfig), C.c_char_p, C.c_int64]
        L.ref_dac_destroy.argtypes = [C.c_void_p]
        L.ref_dac_frames.restype = C.c_int64
        L.ref_dac_frames.argtypes = [C.c_void_p, C.c_int64]
        L.ref_dac_encode.argtypes = [C.c_void_p, f32p, C.c_int64, C.c_int64, C.c_int, i64p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_dac_decode.argtypes = [C.c_void_p, f32p, C.c_int64, C.c_int64, f32p]
        L.ref_dac_from_codes.argtypes = [C.c_void_p, i64p, C.c_int64, C.c_int, C.c_int64, f32p]
        L.ref_conv1d.argtypes = [f32p, C.c_int64, C.c_int, C.c_int64, f32p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_void_p, f32p, C.c_int64]
        L.ref_conv_transpose1d.argtypes = [f32p, C.c_int64, C.c_int, C.c_int64, f32p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_int, f32p, C.c_int64]
        L.ref_snake.argtypes = [f32p, f32p, C.c_int64, C.c_int64, C.c_int64, f32p]
        L.ref_tanh.argtypes = [f32p, C.c_int64, f32p]
        L.ref_fold_wn_dac.argtypes = [f32p, f32p, C.c_int64, C.c_int64, f32p]
        L.ref_vq_argmin.argtypes = [f32p, C.c_int64, C.c_int, C.c_int64, f32p, C.c_int, i64p, f32p, C.c_void_p]
        L.ref_num_threads.restype = C.c_int
        L.ref_snac_create.restype = C.c_void_p
        L.ref_snac_create.argtypes = [C.POINTER(RefSnacConfig), C.c_char_p, C.c_int64]
        L.ref_snac_destroy.argtypes = [C.c_void_p]
        for f in ("ref_snac_padded_length", "ref_snac_frames", "ref_snac_decoded_length"):
            getattr(L, f).restype = C.c_int64
            getattr(L, f).argtypes = [C.c_void_p, C.c_int64]
        L.ref_snac_set_pad.argtypes = [C.c_void_p, C.c_int]
        L.ref_snac_encode.argtypes = [C.c_void_p, f32p, C.c_int64, C.c_int64, i64p, C.c_void_p, C.c_void_p]
        L.ref_snac_from_codes.argtypes = [C.c_void_p, i64p, C.c_int64, C.c_int64, f32p]
        L.ref_snac_decode.argtypes = [C.c_void_p, f32p, C.c_int64, C.c_int64, C.c_void_p, f32p]
        L.ref_fold_wn_snac.argtypes = [f32p, f32p, C.c_int64, C.c_int64, f32p]
        L.ref_encodec_create.restype = C.c_void_p
        L.ref_encodec_create.argtypes = [C.POINTER(RefEncodecConfig), C.c_char_p, C.c_int64]
        L.ref_encodec_destroy.argtypes = [C.c_void_p]
        for f in ("ref_encodec_frames", "ref_encodec_decoded_length"):
            getattr(L, f).restype = C.c_int64
            getattr(L, f).argtypes = [C.c_void_p, C.c_int64]
        L.ref_encodec_encode_frame.argtypes = [C.c_void_p, f32p, C.c_int64, C.c_int64, C.c_int, i64p, C.c_void_p, C.c_void_p]
        L.ref_encodec_decode_frame.argtypes = [C.c_void_p, i64p, C.c_int64, C.c_int, C.c_int64, C.c_void_p, f32p, C.c_void_p]
        L.ref_linear_overlap_add.argtypes = [f32p, i64p, i64p, C.c_int, C.c_int64, C.c_int64, f32p, C.c_int64]
        L.ref_group_norm1.argtypes = [f32p, C.c_int64, C.c_int, C.c_int64, C.c_int, f32p, f32p, f32p]
        _lib = L
    return _lib


def _opt(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def conv1d(x, w, bias=None, stride=1, pad=0, dil=1, groups=1, residual=None):
    x = np.ascontiguousarray(x, np.float32); w = np.ascontiguousarray(w, np.float32)
    B, Cin, Tin = x.shape
    Cout, _, K = w.shape
    Tout = (Tin + 2 * pad - dil * (K - 1) - 1) // stride + 1
    y = np.empty((B, Cout, Tout), np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    r = None if residual is None else np.ascontiguousarray(residual, np.float32)
    lib().ref_conv1d(x, B, Cin, Tin, w, _opt(b), Cout, K, stride, pad, dil, groups, _opt(r), y, Tout)
    return y


def conv_transpose1d(x, w, bias=None, stride=1, pad=0, out_pad=0):
    x = np.ascontiguousarray(x, np.float32); w = np.ascontiguousarray(w, np.float32)
    B, Cin, Tin = x.shape
    _, Cout, K = w.shape
    Tout = (Tin - 1) * stride - 2 * pad + K + out_pad
    y = np.empty((B, Cout, Tout), np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    lib().ref_conv_transpose1d(x, B, Cin, Tin, w, _opt(b), Cout, K, stride, pad, out_pad, y, Tout)
    return y


def snake(x, alpha):
    x = np.ascontiguousarray(x, np.float32)
    a = np.ascontiguousarray(alpha, np.float32).reshape(-1)
    y = np.empty_like(x)
    lib().ref_snake(x, a, x.shape[0], x.shape[1], x.shape[2], y)
    return y


def tanh(x):
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    lib().ref_tanh(x, x.size, y)
    return y


def fold_wn_dac(v, g):
    v = np.ascontiguousarray(v, np.float32); g = np.ascontiguousarray(g, np.float32).reshape(-1)
    w = np.empty_like(v)
    lib().ref_fold_wn_dac(v, g, v.shape[0], int(np.prod(v.shape[1:])), w)
    return w


def vq_argmin(z_e, codebook):
    z_e = np.ascontiguousarray(z_e, np.float32); cb = np.ascontiguousarray(codebook, np.float32)
    B, D, T = z_e.shape
    idx = np.empty((B, T), np.int64); st = np.empty_like(z_e); bd = np.empty((B, T), np.float32)
    lib().ref_vq_argmin(z_e, B, D, T, cb, cb.shape[0], idx, st, bd.ctypes.data_as(C.c_void_p))
    return idx, st, bd


class RefDAC:
    """C-oracle DAC (same call surface as the reference's DAC.Encode / Decode / FromCodes)."""

    def __init__(self, cfg, blob: bytes):
        rc = RefDacConfig()
        rc.sample_rate, rc.encoder_dim, rc.decoder_dim = cfg.sample_rate, cfg.encoder_dim, cfg.decoder_dim
        rc.n_enc_rates, rc.n_dec_rates = len(cfg.encoder_rates), len(cfg.decoder_rates)
        for i, r in enumerate(cfg.encoder_rates): rc.enc_rates[i] = r
        for i, r in enumerate(cfg.decoder_rates): rc.dec_rates[i] = r
        rc.latent_dim, rc.n_codebooks = cfg.resolved_latent_dim, cfg.n_codebooks
        rc.codebook_size, rc.codebook_dim = cfg.codebook_size, cfg.codebook_dim
        self.cfg = cfg
        self._h = lib().ref_dac_create(C.byref(rc), blob, len(blob))
        if not self._h:
            raise RuntimeError("ref_dac_create failed (missing tensors?)")

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            try:
                _lib.ref_dac_destroy(self._h)
            except Exception:
                pass
            self._h = None

    def encode(self, pcm, n_quantizers: int = 0):
        pcm = np.ascontiguousarray(pcm, np.float32)
        B, _, T = pcm.shape
        Tz = lib().ref_dac_frames(self._h, T)
        nq = self.cfg.n_codebooks if n_quantizers <= 0 else min(n_quantizers, self.cfg.n_codebooks)
        ld, D = self.cfg.resolved_latent_dim, self.cfg.codebook_dim
        codes = np.empty((B, nq, Tz), np.int64)
        zq = np.empty((B, ld, Tz), np.float32); lat = np.empty((B, nq * D, Tz), np.float32); ze = np.empty((B, ld, Tz), np.float32)
        lib().ref_dac_encode(self._h, pcm, B, T, nq, codes, _opt(zq), _opt(lat), _opt(ze))
        return zq, codes, lat, ze

    def decode(self, z):
        z = np.ascontiguousarray(z, np.float32)
        B, _, Tz = z.shape
        hop = self.cfg.hop_length
        up = 1
        L = Tz
        for s in self.cfg.decoder_rates:
            L = (L - 1) * s - 2 * ((s + 1) // 2) + 2 * s
        pcm = np.empty((B, 1, L), np.float32)
        lib().ref_dac_decode(self._h, z, B, Tz, pcm)
        return pcm

    def from_codes(self, codes):
        codes = np.ascontiguousarray(codes, np.int64)
        B, nq, Tz = codes.shape
        z = np.empty((B, self.cfg.resolved_latent_dim, Tz), np.float32)
        lib().ref_dac_from_codes(self._h, codes, B, nq, Tz, z)
        return z


class RefSNAC:
    """C-oracle SNAC (call surface of the reference's SNAC.Encode / Decode; codes are a list of per-level int64 arrays)."""

    def __init__(self, cfg, blob: bytes):
        rc = RefSnacConfig()
        rc.sample_rate, rc.encoder_dim, rc.decoder_dim = cfg.sampling_rate, cfg.encoder_dim, cfg.decoder_dim
        rc.n_enc_rates, rc.n_dec_rates, rc.n_vq = len(cfg.encoder_rates), len(cfg.decoder_rates), len(cfg.vq_strides)
        for i, r in enumerate(cfg.encoder_rates): rc.enc_rates[i] = r
        for i, r in enumerate(cfg.decoder_rates): rc.dec_rates[i] = r
        for i, r in enumerate(cfg.vq_strides): rc.vq_strides[i] = r
        rc.latent_dim, rc.attn_window = cfg.resolved_latent_dim, cfg.attn_window_size or 0
        rc.codebook_size, rc.codebook_dim = cfg.codebook_size, cfg.codebook_dim
        rc.noise, rc.depthwise = int(cfg.noise), int(cfg.depthwise)
        self.cfg = cfg
        self._h = lib().ref_snac_create(C.byref(rc), blob, len(blob))
        if not self._h:
            raise RuntimeError("ref_snac_create failed")

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            try:
                _lib.ref_snac_destroy(self._h)
            except Exception:
                pass
            self._h = None

    def level_widths(self, Tz):
        return [Tz // s for s in self.cfg.vq_strides]

    def encode_tensor(self, pcm):
        """SNAC.Encode(Tensor) as written (SNAC.cs:113-122, D7): no pad; ValueError where the reference throws."""
        lib().ref_snac_set_pad(self._h, 0)
        try:
            return self.encode(pcm)
        finally:
            lib().ref_snac_set_pad(self._h, 1)

    def encode(self, pcm):
        pcm = np.ascontiguousarray(pcm, np.float32)
        B, _, T = pcm.shape
        Tz = lib().ref_snac_frames(self._h, T)
        if Tz < 0:
            raise ValueError(f"un-padded input of {T} samples: the reference's quantizer / LocalMHA throws on this length")
        widths = self.level_widths(Tz)
        flat = np.empty((B, sum(widths)), np.int64)
        ld = self.cfg.resolved_latent_dim
        zq = np.empty((B, ld, Tz), np.float32); z = np.empty((B, ld, Tz), np.float32)
        if lib().ref_snac_encode(self._h, pcm, B, T, flat, _opt(zq), _opt(z)) != 0:
            raise RuntimeError("ref_snac_encode failed (missing tensors?)")
        codes, o = [], 0
        for w in widths:
            codes.append(np.ascontiguousarray(flat[:, o:o + w])); o += w
        return z, zq, codes

    def from_codes(self, codes):
        B = codes[0].shape[0]
        Tz = codes[-1].shape[1] * self.cfg.vq_strides[-1]
        flat = np.ascontiguousarray(np.concatenate([np.asarray(c, np.int64).reshape(B, -1) for c in codes], axis=1))
        zq = np.empty((B, self.cfg.resolved_latent_dim, Tz), np.float32)
        if lib().ref_snac_from_codes(self._h, flat, B, Tz, zq) != 0:
            raise RuntimeError("ref_snac_from_codes failed")
        return zq

    def decode_latents(self, zq, noises=None):
        zq = np.ascontiguousarray(zq, np.float32)
        B, _, Tz = zq.shape
        L = lib().ref_snac_decoded_length(self._h, Tz)
        nz = None
        if self.cfg.noise:
            if noises is None:
                raise ValueError("SNAC decode needs the NoiseBlock inputs (deviation D8)")
            nz = np.ascontiguousarray(np.concatenate([np.asarray(n, np.float32).reshape(-1) for n in noises]))
        pcm = np.empty((B, 1, L), np.float32)
        if lib().ref_snac_decode(self._h, zq, B, Tz, _opt(nz), pcm) != 0:
            raise RuntimeError("ref_snac_decode failed")
        return pcm

    def decode(self, codes, noises=None):
        return self.decode_latents(self.from_codes(codes), noises)


class RefEncodec:
    """C-oracle Encodec (call surface of the reference's Encodec.Encode / Decode: a list of (codes, scale) frames)."""

    def __init__(self, cfg, blob: bytes):
        import math
        rc = RefEncodecConfig()
        rc.sample_rate, rc.channels, rc.dimension, rc.n_filters = cfg.sampling_rate, cfg.channels, cfg.dimension, cfg.n_filters
        rc.n_ratios = len(cfg.ratios)
        for i, r in enumerate(cfg.ratios): rc.ratios[i] = r
        rc.lstm_layers, rc.compress, rc.kernel_size = cfg.lstm_layers, cfg.compress, cfg.kernel_size
        rc.last_kernel_size, rc.residual_kernel_size = cfg.last_kernel_size, cfg.residual_kernel_size
        rc.group_norm, rc.causal, rc.normalize = int(cfg.norm == "time_group_norm"), int(cfg.causal), int(cfg.normalize)
        rc.codebook_size = cfg.codebook_size
        self.frame_rate = int(math.ceil(cfg.sampling_rate / float(cfg.hop_length)))
        self.n_q_total = int(1000 * max(cfg.target_bandwidths) / (math.ceil(cfg.sampling_rate / cfg.hop_length) * 10))
        rc.n_q_total = self.n_q_total
        self.cfg = cfg
        self.bandwidth = cfg.bandwidth
        self._h = lib().ref_encodec_create(C.byref(rc), blob, len(blob))
        if not self._h:
            raise RuntimeError("ref_encodec_create failed")

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            try:
                _lib.ref_encodec_destroy(self._h)
            except Exception:
                pass
            self._h = None

    @property
    def segment_length(self):
        return None if self.cfg.segment_seconds is None else int(self.cfg.segment_seconds * self.cfg.sampling_rate)

    @property
    def segment_stride(self):
        sl = self.segment_length
        return None if sl is None else max(1, int((1 - self.cfg.overlap) * sl))

    def n_q(self):
        import math
        bw_per_q = int(math.log2(self.cfg.codebook_size)) * self.frame_rate
        if self.bandwidth and self.bandwidth > 0:
            return int(max(1, math.floor(self.bandwidth * 1000 / bw_per_q)))
        return self.n_q_total

    def encode_frame(self, x, want_emb=False):
        x = np.ascontiguousarray(x, np.float32)
        B, _, L = x.shape
        Tz = lib().ref_encodec_frames(self._h, L)
        nq = self.n_q()
        codes = np.empty((B, nq, Tz), np.int64)
        scale = np.empty((B,), np.float32) if self.cfg.normalize else None
        emb = np.empty((B, self.cfg.dimension, Tz), np.float32) if want_emb else None
        if lib().ref_encodec_encode_frame(self._h, x, B, L, nq, codes, _opt(scale), _opt(emb)) < 0:
            raise RuntimeError("ref_encodec_encode_frame failed (missing tensors?)")
        sc = None if scale is None else scale.reshape(B, 1)
        return (codes, sc, emb) if want_emb else (codes, sc)

    def encode(self, pcm, want_emb=False):
        pcm = np.ascontiguousarray(pcm, np.float32)
        length = pcm.shape[2]
        seg = self.segment_length or length
        stride = self.segment_stride or length
        return [self.encode_frame(pcm[:, :, off:min(off + seg, length)], want_emb) for off in range(0, length, stride)]

    def decode_frame(self, codes, scale):
        codes = np.ascontiguousarray(codes, np.int64)
        B, nq, Tz = codes.shape
        L = lib().ref_encodec_decoded_length(self._h, Tz)
        out = np.empty((B, self.cfg.channels, L), np.float32)
        sc = None if scale is None else np.ascontiguousarray(scale, np.float32).reshape(-1)
        if lib().ref_encodec_decode_frame(self._h, codes, B, nq, Tz, _opt(sc), out, None) < 0:
            raise RuntimeError("ref_encodec_decode_frame failed")
        return out

    def decode(self, frames):
        if len(frames) == 0:
            raise ValueError("No frames provided to decode")
        if self.segment_length is None:
            if len(frames) != 1:
                raise ValueError("Expected single frame when no segmentation is used")
            return self.decode_frame(frames[0][0], frames[0][1])
        outs = [self.decode_frame(f[0], f[1]) for f in frames]
        return linear_overlap_add(outs, self.segment_stride)


def linear_overlap_add(frames, stride):
    B, Cc = frames[0].shape[:2]
    lens = np.array([f.shape[-1] for f in frames], np.int64)
    flat = np.ascontiguousarray(np.concatenate([np.ascontiguousarray(f, np.float32).reshape(-1) for f in frames]))
    offs = np.zeros(len(frames), np.int64)
    offs[1:] = np.cumsum(lens[:-1] * B * Cc)
    total = int(stride * (len(frames) - 1) + lens[-1])
    out = np.empty((B, Cc, total), np.float32)
    lib().ref_linear_overlap_add(flat, offs, lens, len(frames), B * Cc, stride, out, total)
    return out
