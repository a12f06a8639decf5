"""Host-side mirror of the reference's ``SNAC`` model class over the C ABI.

Same members, argument meaning and error behaviour as NeuralCodecs.Torch/Models/SNAC.cs
(``SNAC : Module<Tensor,(Tensor,List<Tensor>)>, INeuralCodec``):

    SNAC(config)                          SNAC.cs:34-63
    load_weights(path)                    SNAC.cs:196-241  (INeuralCodec.LoadWeights)
    encode(audio [B,1,T]) -> List[codes]  SNAC.cs:113-122  (int64 [B, T'/stride_i] per level; always pads, see D7)
    encode_array(float[]) -> List[float[]]SNAC.cs:129-150  (codes cast to float32 arrays, as the reference does)
    decode(List[codes], noise=None)       SNAC.cs:157-165
    decode_array(List[float[]])           SNAC.cs:173-192
    forward(audio) -> (audio_hat, codes)  SNAC.cs:91-106   (trims to the input length)
    process_audio(float[], sample_rate)   SNAC.cs:255-308  (linear resample + forward)

The NoiseBlock inputs (torch.randn at inference in the reference, NoiseBlock.cs:41) can be injected (`noise=` list of
[B,1,T_i] arrays, one per decoder block) for reproducible output; otherwise they are drawn on the device from `seed`.
numpy arrays use the host API; torch device tensors use the zero-copy `*_dev` API on torch's current stream.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from .config import SNACConfig
from .weights import save_blob


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


class SNAC(_lib.ProfileMixin):
    def __init__(self, config: Optional[SNACConfig] = None, device_index: int = 0):
        if config is None:
            raise ValueError("config must not be null")
        self.config = config
        self.device_index = device_index
        c = _lib.NcSnacConfig()
        c.sample_rate, c.encoder_dim, c.decoder_dim = config.sampling_rate, config.encoder_dim, config.decoder_dim
        c.n_encoder_rates, c.n_decoder_rates, c.n_vq_strides = len(config.encoder_rates), len(config.decoder_rates), len(config.vq_strides)
        for i, r in enumerate(config.encoder_rates):
            c.encoder_rates[i] = r
        for i, r in enumerate(config.decoder_rates):
            c.decoder_rates[i] = r
        for i, r in enumerate(config.vq_strides):
            c.vq_strides[i] = r
        c.latent_dim = config.latent_dim or 0
        c.attn_window_size = config.attn_window_size or 0
        c.codebook_size, c.codebook_dim = config.codebook_size, config.codebook_dim
        c.noise, c.depthwise = int(config.noise), int(config.depthwise)
        self._h = C.c_void_p()
        _lib.check(_lib.lib().nc_snac_create(C.byref(c), device_index, C.byref(self._h)))
        self.latent_dim = config.resolved_latent_dim
        self.hop_length = config.hop_length

    @property
    def Config(self) -> SNACConfig:
        return self.config

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

    # ---- shape helpers ---------------------------------------------------------------------
    def query(self, T: int):
        tp, fr, nl, dl = C.c_int64(), C.c_int64(), C.c_int32(), C.c_int64()
        widths = (C.c_int64 * 8)()
        _lib.check(_lib.lib().nc_snac_query(self._h, T, C.byref(tp), C.byref(fr), C.byref(nl), widths, C.byref(dl)))
        return tp.value, fr.value, [widths[i] for i in range(nl.value)], dl.value

    def query_tensor(self, T: int):
        """Frames / level widths of Encode(Tensor) as written (un-padded, SNAC.cs:113-122); ValueError where the reference throws."""
        fr, nl = C.c_int64(), C.c_int32()
        widths = (C.c_int64 * 8)()
        _lib.check(_lib.lib().nc_snac_query_tensor(self._h, T, C.byref(fr), C.byref(nl), widths))
        return fr.value, [widths[i] for i in range(nl.value)]

    def noise_shapes(self, B: int, frames: int):
        out, L = [], frames
        for s in self.config.decoder_rates:
            L = (L - 1) * s - 2 * (-(-s // 2)) + 2 * s + (s % 2)
            out.append((B, 1, L))
        return out if self.config.noise else []

    def _bind_torch_stream(self):
        import torch
        _lib.check(_lib.lib().nc_codec_set_stream(self._h, C.c_void_p(torch.cuda.current_stream(self.device_index).cuda_stream)))   # (the handle's OWN device: one process may drive several)

    def synchronize(self) -> None:
        _lib.check(_lib.lib().nc_codec_synchronize(self._h))

    # ---- Encode ----------------------------------------------------------------------------
    def encode_tensor(self, audio_data, return_latents: bool = False):
        """SNAC.Encode(Tensor) exactly as written (Models/SNAC.cs:113-122, deviation D7): Preprocess's result is dropped and the
        encoder runs on the UN-padded tensor.  `encode` below is the padding Encode(float[]) / forward behaviour."""
        return self.encode(audio_data, return_latents, _pad=False)

    def encode(self, audio_data, return_latents: bool = False, _pad: bool = True):
        if audio_data is None:
            raise ValueError("audio_data must not be null")
        if audio_data.ndim != 3 or audio_data.shape[1] != 1:
            raise ValueError("audio must be [B,1,T]")
        B, _, T = audio_data.shape
        if _pad:
            _, Tz, widths, _ = self.query(T)
        else:
            Tz, widths = self.query_tensor(T)
        total = sum(widths)
        L = _lib.lib()
        fn_dev, fn_host = (L.nc_snac_encode_dev, L.nc_snac_encode) if _pad else (L.nc_snac_encode_tensor_dev, L.nc_snac_encode_tensor)
        if _is_torch(audio_data):
            import torch
            x = audio_data.contiguous().to(torch.float32)
            flat = torch.empty((B, total), dtype=torch.int64, device=x.device)
            z = torch.empty((B, self.latent_dim, Tz), dtype=torch.float32, device=x.device)
            zq = torch.empty_like(z)
            self._bind_torch_stream()
            _lib.check(fn_dev(self._h, x.data_ptr(), B, T, flat.data_ptr(), z.data_ptr(), zq.data_ptr()))
        else:
            x = np.ascontiguousarray(audio_data, dtype=np.float32)
            flat = np.empty((B, total), np.int64)
            z = np.empty((B, self.latent_dim, Tz), np.float32)
            zq = np.empty_like(z)
            _lib.check(fn_host(self._h, x.ctypes.data, B, T, flat.ctypes.data, z.ctypes.data, zq.ctypes.data))
        codes, o = [], 0
        for w in widths:
            codes.append(flat[:, o:o + w])
            o += w
        return (codes, z, zq) if return_latents else codes

    def encode_array(self, audio_data) -> List[np.ndarray]:
        """SNAC.Encode(float[]) (SNAC.cs:129-150): B=1; codes come back as float32 arrays like the reference's ConvertAll."""
        if audio_data is None:
            raise ValueError("audio_data must not be null")
        x = np.asarray(audio_data, dtype=np.float32).reshape(1, 1, -1)
        return [np.ascontiguousarray(c[0]).astype(np.float32) for c in self.encode(x)]

    # ---- Decode ----------------------------------------------------------------------------
    def _flat_codes(self, codes: Sequence):
        if codes is None:
            raise ValueError("codes must not be null")
        if len(codes) != len(self.config.vq_strides):
            raise ValueError(f"Expected {len(self.config.vq_strides)} codebooks but got {len(codes)}")   # SNAC RVQ.cs:103
        return codes

    @staticmethod
    def _as_one_buffer(parts, B: int):
        """The [B, sum(widths)] int64 tensor the parts are column views of (what encode() hands out), or None.  No copy, no kernel."""
        try:
            p0 = parts[0]
            total = sum(int(p.shape[-1]) for p in parts)
            base, off = p0.untyped_storage().data_ptr(), p0.storage_offset()
            for p in parts:
                if (p.dtype != p0.dtype or p.dim() != 2 or p.shape[0] != B or p.untyped_storage().data_ptr() != base or p.storage_offset() != off
                        or (B > 1 and p.stride(0) != total) or (p.shape[-1] > 1 and p.stride(1) != 1)):
                    return None
                off += int(p.shape[-1])
            return p0.as_strided((B, total), (total, 1), p0.storage_offset())
        except Exception:
            return None

    @staticmethod
    def _noise_end_to_end(parts):
        try:
            import torch
            p0 = parts[0]
            base, off = p0.untyped_storage().data_ptr(), p0.storage_offset()
            for p in parts:
                if p.dtype != torch.float32 or not p.is_contiguous() or p.untyped_storage().data_ptr() != base or p.storage_offset() != off:
                    return None
                off += p.numel()
            return p0.as_strided((off - p0.storage_offset(),), (1,), p0.storage_offset())
        except Exception:
            return None

    def flat_noise(self, noise: Sequence, device=None):
        """The NoiseBlock inputs in the layout the C ABI takes them (one block per decoder stage laid end to end, nc_snac_decode): returns
        the list again, now as views of ONE device buffer, so that decode() passes the buffer without copying.  Input preparation:
        do it once per noise draw, outside a timed region."""
        import torch
        flat = torch.cat([torch.as_tensor(n, dtype=torch.float32).reshape(-1) for n in noise]).contiguous()
        if device is not None:
            flat = flat.to(device)
        out, o = [], 0
        for n in noise:
            k = int(np.prod(n.shape))
            out.append(flat[o:o + k].view(tuple(n.shape)))
            o += k
        return out

    def decode(self, codes: Sequence, noise: Optional[Sequence] = None, seed: int = 0):
        codes = self._flat_codes(codes)
        B = codes[0].shape[0]
        frames = int(codes[-1].shape[-1]) * self.config.vq_strides[-1]
        _, _, widths, _ = self.query(frames * self.hop_length)
        if [int(c.shape[-1]) for c in codes] != widths:
            raise ValueError(f"code widths {[int(c.shape[-1]) for c in codes]} do not match {widths}")
        L = frames
        for s in self.config.decoder_rates:
            L = (L - 1) * s - 2 * (-(-s // 2)) + 2 * s + (s % 2)
        if _is_torch(codes[0]):
            import torch
            flat = self._as_one_buffer(codes, B) if codes[0].dtype == torch.int64 else None   # encode()'s own views: the ABI's flat buffer as it is
            if flat is None:
                flat = torch.cat([c.reshape(B, -1).to(torch.int64) for c in codes], dim=1).contiguous()
            out = torch.empty((B, 1, L), dtype=torch.float32, device=flat.device)
            nz = None
            if noise is not None and self.config.noise:
                nz = self._noise_end_to_end(noise)                        # views of one flat buffer (flat_noise()): passed as it is
                if nz is None:
                    nz = torch.cat([n.reshape(-1).to(torch.float32) for n in noise]).contiguous()
            self._bind_torch_stream()
            _lib.check(_lib.lib().nc_snac_decode_dev(self._h, flat.data_ptr(), B, frames, nz.data_ptr() if nz is not None else None,
                                                     seed, out.data_ptr()))
            return out
        flat = np.ascontiguousarray(np.concatenate([np.asarray(c).reshape(B, -1).astype(np.int64) for c in codes], axis=1))
        out = np.empty((B, 1, L), np.float32)
        nz = None
        if noise is not None and self.config.noise:
            shapes = self.noise_shapes(B, frames)
            if [tuple(n.shape) for n in noise] != shapes:
                raise ValueError(f"noise shapes must be {shapes}")
            nz = np.ascontiguousarray(np.concatenate([np.asarray(n, np.float32).reshape(-1) for n in noise]))
        _lib.check(_lib.lib().nc_snac_decode(self._h, flat.ctypes.data, B, frames, nz.ctypes.data if nz is not None else None, seed,
                                             out.ctypes.data))
        return out

    def decode_array(self, codes: Sequence, noise=None, seed: int = 0) -> np.ndarray:
        """SNAC.Decode(List<float[]>) (SNAC.cs:173-192)."""
        if codes is None:
            raise ValueError("codes must not be null")
        if len(codes) == 0 or any(c is None for c in codes):
            raise ValueError("Codes list cannot be empty or contain null arrays")
        return self.decode([np.asarray(c).astype(np.int64).reshape(1, -1) for c in codes], noise, seed).reshape(-1)

    def from_codes(self, codes: Sequence) -> np.ndarray:
        codes = self._flat_codes(codes)
        B = codes[0].shape[0]
        frames = int(codes[-1].shape[-1]) * self.config.vq_strides[-1]
        flat = np.ascontiguousarray(np.concatenate([np.asarray(c).reshape(B, -1).astype(np.int64) for c in codes], axis=1))
        zq = np.empty((B, self.latent_dim, frames), np.float32)
        _lib.check(_lib.lib().nc_snac_from_codes(self._h, flat.ctypes.data, B, frames, zq.ctypes.data))
        return zq

    # ---- forward / ProcessAudio ----------------------------------------------------------------
    def forward(self, audio_data, noise=None, seed: int = 0):
        length = audio_data.shape[-1]
        codes = self.encode(audio_data)
        audio = self.decode(codes, noise, seed)
        return audio[..., :length], codes                                   # SNAC.cs:103 narrow(-1, 0, length)

    @staticmethod
    def resample_linear(x, src: int, dst: int):
        """SNAC.ResampleAudio (SNAC.cs:284-308): linear interpolation with the C#'s float64 position arithmetic, on the device
        (csrc/nc_audio.hip)."""
        from . import audio
        return audio.resample_linear(x, src, dst)

    def process_audio(self, audio_data, sample_rate: int, noise=None, seed: int = 0) -> np.ndarray:
        if audio_data is None or len(audio_data) == 0:
            raise ValueError("Audio data cannot be empty")
        x = np.ascontiguousarray(np.asarray(audio_data, dtype=np.float32).reshape(-1))
        L = _lib.lib()
        n_out = C.c_int64()
        _lib.check(L.nc_snac_process_audio_len(self._h, x.size, int(sample_rate), C.byref(n_out)))
        nz = None
        if noise is not None:                                               # one [1,1,T_i] block per decoder stage, laid end to end
            nz = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.float32).reshape(-1) for a in noise]))
            _, frames, _, _ = self.query(n_out.value)
            if nz.size != sum(int(np.prod(s)) for s in self.noise_shapes(1, frames)):
                raise ValueError("noise does not match the decoder stages of the resampled clip")
        out = np.empty(n_out.value, dtype=np.float32)
        # one upload -> resample -> forward -> one download, inside the engine (nc_snac_process_audio)
        _lib.check(L.nc_snac_process_audio(self._h, x.ctypes.data_as(C.c_void_p), x.size, int(sample_rate),
                                           nz.ctypes.data_as(C.c_void_p) if nz is not None else None, seed & 0xFFFFFFFFFFFFFFFF,
                                           out.ctypes.data_as(C.c_void_p)))
        return out
