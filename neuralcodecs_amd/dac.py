"""Host-side mirror of the reference's ``DAC`` model class over the C ABI.

Same members, argument meaning and error behaviour as NeuralCodecs.Torch/Models/DAC.cs
(``DAC : Module<Tensor, Dictionary<string,Tensor>>, INeuralCodec``):

    DAC(config)                      DAC.cs:51-93
    load_weights(path)               DAC.cs:345-389   (INeuralCodec.LoadWeights)
    encode(audio, n_quantizers, sample_rate) -> (z, codes, latents, commitment_loss, codebook_loss)   DAC.cs:163-181
    encode_array(float[]) -> float[] (the zQ latents, D12)                                            DAC.cs:205-224
    decode(z) / decode_array(float[])                                                                 DAC.cs:231-253
    from_codes(codes)                                                                                 DAC.cs:101-106
    forward(audio) -> dict                                                                            DAC.cs:288-303
    dispose()                                                                                         DAC.cs:329-338

Inputs may be numpy arrays (host API: synchronous, returns numpy) or torch CUDA tensors
(device API: zero-copy, enqueued on torch's current stream, returns torch tensors).  All compute
happens in libnc_mi355x.so; nothing here touches the oracle or torch operators.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib
from .config import DACConfig
from .weights import save_blob


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


class DAC(_lib.ProfileMixin):
    def __init__(self, config: Optional[DACConfig] = None, device_index: int = 0):
        if config is None:
            raise ValueError("config must not be null")  # ArgumentNullException.ThrowIfNull(config)
        self.config = config
        self.device_index = device_index
        c = _lib.NcDacConfig()
        c.sample_rate, c.encoder_dim, c.decoder_dim = config.sample_rate, config.encoder_dim, config.decoder_dim
        c.n_encoder_rates, c.n_decoder_rates = len(config.encoder_rates), len(config.decoder_rates)
        for i, r in enumerate(config.encoder_rates):
            c.encoder_rates[i] = r
        for i, r in enumerate(config.decoder_rates):
            c.decoder_rates[i] = r
        c.latent_dim = config.latent_dim or 0
        c.n_codebooks, c.codebook_size, c.codebook_dim = config.n_codebooks, config.codebook_size, config.codebook_dim
        self._h = C.c_void_p()
        _lib.check(_lib.lib().nc_dac_create(C.byref(c), device_index, C.byref(self._h)))
        self.latent_dim = config.resolved_latent_dim
        self.hop_length = config.hop_length

    # ---- INeuralCodec ----------------------------------------------------------------------
    @property
    def Config(self) -> DACConfig:
        return self.config

    def load_weights(self, path: str) -> None:
        _lib.check(_lib.lib().nc_codec_load_weights(self._h, str(path).encode()))

    def load_state_dict(self, state_dict) -> None:
        """TorchSharp-keyed tensors (weight_v/weight_g/bias/alpha/codebook.weight) -> engine."""
        blob = save_blob(state_dict)
        self.load_blob(blob)

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

    # ---- helpers ---------------------------------------------------------------------------
    def frames(self, T: int) -> int:
        return -(-T // self.hop_length)

    def decoded_length(self, frames: int) -> int:
        L = frames
        for s in self.config.decoder_rates:
            L = (L - 1) * s - 2 * ((s + 1) // 2) + 2 * s
        return L

    def _nq(self, n_quantizers) -> int:
        n = self.config.n_codebooks
        return n if (n_quantizers is None or n_quantizers <= 0 or n_quantizers > n) else int(n_quantizers)

    def _bind_torch_stream(self):
        import torch
        _lib.check(_lib.lib().nc_codec_set_stream(self._h, C.c_void_p(torch.cuda.current_stream(self.device_index).cuda_stream)))   # (the handle's OWN device: one process may drive several)

    def synchronize(self) -> None:
        _lib.check(_lib.lib().nc_codec_synchronize(self._h))

    # ---- Encode ----------------------------------------------------------------------------
    def encode(self, audio_data, n_quantizers: Optional[int] = None, sample_rate: Optional[int] = None):
        if audio_data is None:
            raise ValueError("audio_data must not be null")
        if audio_data.ndim != 3 or audio_data.shape[1] != 1:
            raise ValueError("audio must be [B,1,T]")
        B, _, T = audio_data.shape
        nq, Tz, D = self._nq(n_quantizers), self.frames(T), self.config.codebook_dim
        sr = 0 if sample_rate is None else int(sample_rate)
        if _is_torch(audio_data):
            import torch
            x = audio_data.contiguous().to(torch.float32)
            codes = torch.empty((B, nq, Tz), dtype=torch.int64, device=x.device)
            z = torch.empty((B, self.latent_dim, Tz), dtype=torch.float32, device=x.device)
            lat = torch.empty((B, nq * D, Tz), dtype=torch.float32, device=x.device)
            self._bind_torch_stream()
            _lib.check(_lib.lib().nc_dac_encode_dev(self._h, x.data_ptr(), B, T, sr, nq, codes.data_ptr(), z.data_ptr(),
                                                    lat.data_ptr()))
            # commitment / codebook loss: 0 in eval mode (ResidualVectorQuantizer.cs:143-156).  Fresh, distinct tensors on every call, as
            # in the reference: a caller may accumulate into them in place (two scalar fills; a cached pair would be shared state)
            return z, codes, lat, torch.zeros((), device=x.device), torch.zeros((), device=x.device)
        x = np.ascontiguousarray(audio_data, dtype=np.float32)
        codes = np.empty((B, nq, Tz), np.int64)
        z = np.empty((B, self.latent_dim, Tz), np.float32)
        lat = np.empty((B, nq * D, Tz), np.float32)
        _lib.check(_lib.lib().nc_dac_encode(self._h, x.ctypes.data, B, T, sr, nq, codes.ctypes.data, z.ctypes.data, lat.ctypes.data))
        return z, codes, lat, np.float32(0.0), np.float32(0.0)

    def encode_audio(self, audio_data):
        """DAC.EncodeAudio (DAC.cs:188-199): zQ only."""
        return self.encode(audio_data)[0]

    def encode_array(self, audio_data) -> np.ndarray:
        """DAC.Encode(float[]) (DAC.cs:205-224): B=1 host wrapper returning the flattened zQ latents (D12)."""
        if audio_data is None:
            raise ValueError("audio_data must not be null")
        x = np.asarray(audio_data, dtype=np.float32).reshape(1, 1, -1)
        return self.encode(x)[0].reshape(-1)

    # ---- Decode ----------------------------------------------------------------------------
    def decode(self, q_audio):
        if q_audio is None:
            raise ValueError("q_audio must not be null")
        if q_audio.ndim != 3 or q_audio.shape[1] != self.latent_dim:
            raise ValueError(f"latents must be [B,{self.latent_dim},T']")
        B, _, Tz = q_audio.shape
        L = self.decoded_length(Tz)
        if _is_torch(q_audio):
            import torch
            z = q_audio.contiguous().to(torch.float32)
            out = torch.empty((B, 1, L), dtype=torch.float32, device=z.device)
            self._bind_torch_stream()
            _lib.check(_lib.lib().nc_dac_decode_dev(self._h, z.data_ptr(), B, Tz, out.data_ptr()))
            return out
        z = np.ascontiguousarray(q_audio, dtype=np.float32)
        out = np.empty((B, 1, L), np.float32)
        _lib.check(_lib.lib().nc_dac_decode(self._h, z.ctypes.data, B, Tz, out.ctypes.data))
        return out

    def decode_array(self, q_audio) -> np.ndarray:
        """DAC.Decode(float[]) (DAC.cs:241-253): reshape(1, latent, -1) then decode."""
        if q_audio is None:
            raise ValueError("q_audio must not be null")
        z = np.asarray(q_audio, dtype=np.float32).reshape(1, self.latent_dim, -1)
        return self.decode(z).reshape(-1)

    def from_codes(self, codes):
        if codes is None:
            raise ValueError("codes must not be null")
        if codes.ndim != 3:
            raise ValueError("codes must be [B,n_q,T']")
        B, nq, Tz = codes.shape
        if _is_torch(codes):
            import torch
            c = codes.contiguous().to(torch.int64)
            z = torch.empty((B, self.latent_dim, Tz), dtype=torch.float32, device=c.device)
            self._bind_torch_stream()
            _lib.check(_lib.lib().nc_dac_from_codes_dev(self._h, c.data_ptr(), B, nq, Tz, z.data_ptr()))
            return z
        c = np.ascontiguousarray(codes, dtype=np.int64)
        z = np.empty((B, self.latent_dim, Tz), np.float32)
        _lib.check(_lib.lib().nc_dac_from_codes(self._h, c.ctypes.data, B, nq, Tz, z.ctypes.data))
        return z

    # ---- Dia <-> DAC glue -----------------------------------------------------------
    def decode_code_matrix(self, audio_codes):
        """Dia.Decode (Models/Dia.cs:973-981): codes [T, n_q] (or batched [B, T, n_q]) -> FromCodes -> Decode -> waveform
        [T*hop] (or [B, T*hop]).  One C-ABI call (nc_dac_decode_code_matrix[_dev]): the transpose to the engine's [B, n_q, T] is a
        device kernel, batched clips decode in one launch set."""
        if audio_codes is None:
            raise ValueError("audio_codes must not be null")
        single = audio_codes.ndim == 2
        c = audio_codes[None] if single else audio_codes
        if c.ndim != 3:
            raise ValueError("codes must be [T, n_q] or [B, T, n_q]")
        B, Tz, nq = (int(v) for v in c.shape)
        L = self.decoded_length(Tz)
        if _is_torch(c):
            import torch
            cc = c.contiguous().to(torch.int64)
            audio = torch.empty((B, L), dtype=torch.float32, device=cc.device)
            self._bind_torch_stream()
            _lib.check(_lib.lib().nc_dac_decode_code_matrix_dev(self._h, cc.data_ptr(), B, Tz, nq, audio.data_ptr()))
        else:
            cc = np.ascontiguousarray(c, dtype=np.int64)
            audio = np.empty((B, L), np.float32)
            _lib.check(_lib.lib().nc_dac_decode_code_matrix(self._h, cc.ctypes.data, B, Tz, nq, audio.ctypes.data))
        return audio[0] if single else audio

    def encode_to_code_matrix(self, audio, sample_rate: Optional[int] = None):
        """Dia.Encode (Models/Dia.cs:989-1002): audio [C=1, T] (or [B, 1, T]) -> Encode -> codes [T', n_q] (or [B, T', n_q]);
        one C-ABI call (nc_dac_encode_code_matrix[_dev])."""
        if audio is None:
            raise ValueError("audio must not be null")
        single = audio.ndim == 2
        a = audio[None] if single else audio
        if a.ndim != 3 or a.shape[1] != 1:
            raise ValueError("audio must be [1, T] or [B, 1, T]")
        sr = self.config.sample_rate if sample_rate is None else int(sample_rate)
        B, _, T = (int(v) for v in a.shape)
        Tz, nq = self.frames(T), self.config.n_codebooks
        if _is_torch(a):
            import torch
            x = a.contiguous().to(torch.float32)
            codes = torch.empty((B, Tz, nq), dtype=torch.int64, device=x.device)
            self._bind_torch_stream()
            _lib.check(_lib.lib().nc_dac_encode_code_matrix_dev(self._h, x.data_ptr(), B, T, sr, codes.data_ptr()))
        else:
            x = np.ascontiguousarray(a, dtype=np.float32)
            codes = np.empty((B, Tz, nq), np.int64)
            _lib.check(_lib.lib().nc_dac_encode_code_matrix(self._h, x.ctypes.data, B, T, sr, codes.ctypes.data))
        return codes[0] if single else codes

    @staticmethod
    def decode_one_frame(model: "DAC", audio_codes):
        """Modules/Dia/AudioUtils.cs:189-199: exactly one [1, n_q, T] frame -> FromCodes -> Decode."""
        if audio_codes.shape[0] != 1:
            raise ValueError(f"Expected one frame, got {audio_codes.shape[0]}")
        return model.decode(model.from_codes(audio_codes))

    # ---- forward ---------------------------------------------------------------------------
    def forward(self, audio_data, sample_rate: Optional[int] = None, n_quantizers: Optional[int] = None):
        z, codes, latents, cl, cbl = self.encode(audio_data, n_quantizers, sample_rate)
        audio = self.decode(z)
        return {"audio": audio, "z": z, "codes": codes, "latents": latents, "vq/commitment_loss": cl, "vq/codebook_loss": cbl}

    def forward_array(self, audio_data) -> np.ndarray:
        """DAC.forward(float[]) (DAC.cs:310-322)."""
        if audio_data is None:
            raise ValueError("audio_data must not be null")
        x = np.asarray(audio_data, dtype=np.float32).reshape(1, 1, -1)
        return self.forward(x)["audio"].reshape(-1)

    # ---- profiling (bench.py) ----------------------------------------------------------------
