"""Host-side mirror of the reference's ``Encodec`` model class over the C ABI.

Same members, argument meaning and error behaviour as NeuralCodecs.Torch/Models/Encodec.cs
(``Encodec : Module<Tensor,Tensor>, INeuralCodec``):

    Encodec(config)                         Encodec.cs:46-90   (ArgumentException on a bandwidth outside TargetBandwidths)
    load_weights(path)                      Encodec.cs:375-405 (INeuralCodec.LoadWeights)
    encode(x [B,C,T]) -> List[EncodedFrame] Encodec.cs:259-285 (one frame per 1 s segment at 48 kHz; a single frame at 24 kHz)
    encode_array(float[])                   Encodec.cs:243-252 (reshape(1, channels, -1))
    decode(frames) -> [B,C,T_dec]           Encodec.cs:213-235 (per-frame decode, x scale, triangular overlap-add)
    forward(x)                              Encodec.cs:287-291 (decode(encode(x)) trimmed to the input length)
    set_target_bandwidth(kbps)              Encodec.cs:409-419
    properties frame_rate, bits_per_codebook, num_codebooks, segment_length, segment_stride   Encodec.cs:145-201

numpy arrays use the host API; torch device tensors use the zero-copy `*_dev` API on torch's current stream.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from .config import EncodecConfig
from .weights import save_blob


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


@dataclass
class EncodedFrame:
    """Modules/Encodec/EncodedFrame.cs: codes [B, n_q, T'] int64 and the optional per-clip scale [B, 1]."""
    codes: object
    scale: Optional[object] = None


class Encodec(_lib.ProfileMixin):
    def __init__(self, config: Optional[EncodecConfig] = None, device_index: int = 0):
        if config is None:
            raise ValueError("config must not be null")
        if config.bandwidth is None or config.bandwidth not in tuple(config.target_bandwidths):
            raise ValueError(f"Invalid bandwidth {config.bandwidth}. Select one of {list(config.target_bandwidths)}")   # Encodec.cs:49-54
        self.config = config
        self.device_index = device_index
        hop = config.hop_length
        self.frame_rate = int(math.ceil(config.sampling_rate / float(hop)))                                # Encodec.cs:83
        self.bits_per_codebook = int(math.log2(config.codebook_size))
        self.num_codebooks = int(1000 * max(config.target_bandwidths) / (math.ceil(config.sampling_rate / hop) * 10))   # Encodec.cs:70-71
        self.segment_length = None if config.segment_seconds is None else int(config.segment_seconds * config.sampling_rate)
        self.segment_stride = None if self.segment_length is None else max(1, int((1 - config.overlap) * self.segment_length))
        c = _lib.NcEncodecConfig()
        c.sample_rate, c.channels, c.dimension, c.n_filters = config.sampling_rate, config.channels, config.dimension, config.n_filters
        c.n_ratios = len(config.ratios)
        for i, r in enumerate(config.ratios):
            c.ratios[i] = r
        c.lstm_layers, c.compress, c.kernel_size = config.lstm_layers, config.compress, config.kernel_size
        c.last_kernel_size, c.residual_kernel_size = config.last_kernel_size, config.residual_kernel_size
        c.time_group_norm = int(config.norm == "time_group_norm")
        c.causal, c.normalize = int(config.causal), int(config.normalize)
        c.segment_length, c.segment_stride = self.segment_length or 0, self.segment_stride or 0
        c.codebook_size, c.n_codebooks, c.frame_rate, c.bandwidth = config.codebook_size, self.num_codebooks, self.frame_rate, config.bandwidth
        self._h = C.c_void_p()
        _lib.check(_lib.lib().nc_encodec_create(C.byref(c), device_index, C.byref(self._h)))

    @property
    def Config(self) -> EncodecConfig:
        return self.config

    @property
    def current_bandwidth(self) -> float:
        return self.config.bandwidth

    # ---- INeuralCodec ----------------------------------------------------------------------
    def load_weights(self, path: str) -> None:
        _lib.check(_lib.lib().nc_codec_load_weights(self._h, str(path).encode()))

    def load_state_dict(self, state_dict) -> None:
        self.load_blob(save_blob(state_dict))

    def load_blob(self, blob: bytes) -> None:
        _lib.check(_lib.lib().nc_codec_load_weights_mem(self._h, blob, len(blob)))

    def dispose(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            try:
                _lib.lib().nc_codec_destroy(self._h)
            finally:
                self._h = C.c_void_p()

    close = dispose

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.dispose()

    def __del__(self):
        try:
            self.dispose()
        except Exception:
            pass

    def set_target_bandwidth(self, bandwidth: float) -> None:
        if bandwidth not in tuple(self.config.target_bandwidths):
            raise ValueError(f"This model doesn't support the bandwidth {bandwidth} kbps. "
                             f"Select one of {list(self.config.target_bandwidths)} kbps")              # Encodec.cs:411-416
        _lib.check(_lib.lib().nc_encodec_set_bandwidth(self._h, float(bandwidth)))
        self.config.bandwidth = bandwidth

    def query(self, T: int):
        nf, nq, dl = C.c_int32(), C.c_int32(), C.c_int64()
        _lib.check(_lib.lib().nc_encodec_query(self._h, T, C.byref(nf), C.byref(nq), None, 0, C.byref(dl)))   # count first
        lens = (C.c_int64 * max(1, nf.value))()
        _lib.check(_lib.lib().nc_encodec_query(self._h, T, C.byref(nf), C.byref(nq), lens, nf.value, C.byref(dl)))
        return nf.value, nq.value, [lens[i] for i in range(nf.value)], dl.value

    def lstm_stats(self):
        """(stepwise, timeouts): whether the handle has dropped to the step-wise LSTM kernels and how many persistent-launch timeouts it has
        seen (nc_encodec_lstm_stats)."""
        sw, tmo = C.c_int32(0), C.c_int64(0)
        _lib.check(_lib.lib().nc_encodec_lstm_stats(self._h, C.byref(sw), C.byref(tmo)))
        return bool(sw.value), int(tmo.value)

    def _bind_torch_stream(self):
        import torch
        _lib.check(_lib.lib().nc_codec_set_stream(self._h, C.c_void_p(torch.cuda.current_stream(self.device_index).cuda_stream)))   # (the handle's OWN device: one process may drive several)

    # ---- Encode ----------------------------------------------------------------------------
    def _validate(self, x):
        if x is None:
            raise ValueError("audio_data must not be null")
        if x.ndim != 3:
            raise ValueError(f"Expected 3D input tensor [B,C,T], got shape {list(x.shape)}")            # Encodec.cs:493-497
        if x.shape[1] != self.config.channels:
            raise ValueError(f"Expected {self.config.channels} channels, got {x.shape[1]}")             # Encodec.cs:499-503

    def encode(self, x, return_emb: bool = False) -> List[EncodedFrame]:
        self._validate(x)
        B, _, T = x.shape
        nf, nq, lens, _ = self.query(T)
        tot = sum(lens)
        D = self.config.dimension
        if _is_torch(x):
            import torch
            xx = x.contiguous().to(torch.float32)
            codes = torch.empty((B * nq * tot,), dtype=torch.int64, device=xx.device)
            scales = torch.empty((nf, B), dtype=torch.float32, device=xx.device)
            emb = torch.empty((B * D * tot,), dtype=torch.float32, device=xx.device) if return_emb else None
            self._bind_torch_stream()
            _lib.check(_lib.lib().nc_encodec_encode_dev(self._h, xx.data_ptr(), B, T, codes.data_ptr(), scales.data_ptr(),
                                                        emb.data_ptr() if emb is not None else None))
        else:
            xx = np.ascontiguousarray(x, dtype=np.float32)
            codes = np.empty((B * nq * tot,), np.int64)
            scales = np.empty((nf, B), np.float32)
            emb = np.empty((B * D * tot,), np.float32) if return_emb else None
            _lib.check(_lib.lib().nc_encodec_encode(self._h, xx.ctypes.data, B, T, codes.ctypes.data, scales.ctypes.data,
                                                    emb.ctypes.data if emb is not None else None))
        frames, embs, o, oe = [], [], 0, 0
        for f, Tf in enumerate(lens):
            c = codes[o:o + B * nq * Tf].reshape(B, nq, Tf)
            o += B * nq * Tf
            frames.append(EncodedFrame(c, scales[f].reshape(B, 1) if self.config.normalize else None))
            if return_emb:
                embs.append(emb[oe:oe + B * D * Tf].reshape(B, D, Tf))
                oe += B * D * Tf
        return (frames, embs) if return_emb else frames

    def encode_array(self, audio_data) -> List[EncodedFrame]:
        if audio_data is None:
            raise ValueError("audio_data must not be null")
        return self.encode(np.asarray(audio_data, dtype=np.float32).reshape(1, self.config.channels, -1))

    # ---- Decode ----------------------------------------------------------------------------
    def decode(self, frames: Sequence[EncodedFrame], length: Optional[int] = None):
        """`length` = sample count of the clip the frames were encoded from (fixes the segment layout for segmented models;
        defaults to the layout implied by the number of frames: all full segments but the last, whose length is inferred)."""
        if frames is None or len(frames) == 0:
            raise ValueError("No frames provided to decode")                                            # Encodec.cs:215-218
        if self.segment_length is None and len(frames) != 1:
            raise ValueError("Expected single frame when no segmentation is used")                      # Encodec.cs:222-225
        for f in frames:
            if f.codes is None:
                raise ValueError("Invalid frame codes in Encodec Decode")
        B, nq, _ = frames[0].codes.shape
        T = length if length is not None else self._infer_length(frames)
        nf, _, lens, Ld = self.query(T)
        if nf != len(frames) or [int(f.codes.shape[-1]) for f in frames] != lens:
            raise ValueError(f"frames do not match the segment layout of a {T}-sample clip: expected {lens}")
        if _is_torch(frames[0].codes):
            import torch
            codes = self._end_to_end([f.codes for f in frames], torch.int64)      # encode()'s own views: the ABI's flat buffers as they are
            if codes is None:
                codes = torch.cat([f.codes.reshape(-1).to(torch.int64) for f in frames]).contiguous()
            scales = None
            if self.config.normalize:
                scales = self._end_to_end([f.scale for f in frames], torch.float32)
                if scales is None:
                    scales = torch.cat([f.scale.reshape(-1).to(torch.float32) for f in frames]).contiguous()
            out = torch.empty((B, self.config.channels, Ld), dtype=torch.float32, device=codes.device)
            self._bind_torch_stream()
            _lib.check(_lib.lib().nc_encodec_decode_dev(self._h, codes.data_ptr(), scales.data_ptr() if scales is not None else None,
                                                        B, T, nq, out.data_ptr()))
            return out
        codes = np.ascontiguousarray(np.concatenate([np.asarray(f.codes, np.int64).reshape(-1) for f in frames]))
        scales = None
        if self.config.normalize:
            scales = np.ascontiguousarray(np.concatenate([np.asarray(f.scale, np.float32).reshape(-1) for f in frames]))
        out = np.empty((B, self.config.channels, Ld), np.float32)
        _lib.check(_lib.lib().nc_encodec_decode(self._h, codes.ctypes.data, scales.ctypes.data if scales is not None else None, B, T, nq,
                                                out.ctypes.data))
        return out

    @staticmethod
    def _end_to_end(parts, dtype):
        """The flat tensor the parts are consecutive contiguous views of (what encode() hands out), or None.  No copy, no kernel."""
        try:
            p0 = parts[0]
            base, off = p0.untyped_storage().data_ptr(), p0.storage_offset()
            for p in parts:
                if p.dtype != dtype or not p.is_contiguous() or p.untyped_storage().data_ptr() != base or p.storage_offset() != off:
                    return None
                off += p.numel()
            return p0.as_strided((off - p0.storage_offset(),), (1,), p0.storage_offset())
        except Exception:
            return None

    def _infer_length(self, frames) -> int:
        """Clip length implied by the frames alone, as the reference's Decode(List<EncodedFrame>) sees them (Encodec.cs:213-235):
        the smallest T whose segment layout has exactly these frame counts (nc_encodec_clip_length)."""
        T = C.c_int64()
        lens = (C.c_int64 * len(frames))(*[int(f.codes.shape[-1]) for f in frames])
        _lib.check(_lib.lib().nc_encodec_clip_length(self._h, len(frames), lens, C.byref(T)))
        return T.value

    def forward(self, x):
        frames = self.encode(x)
        return self.decode(frames, x.shape[-1])[..., : x.shape[-1]]                                    # Encodec.cs:290
