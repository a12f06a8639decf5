"""Codec hyper-parameter presets (host side).

Mirrors the fields the reference's model constructors actually consume:
  * DACConfig      <- NeuralCodecs.Torch/Config/DAC/DACConfig.cs:8-137
  * SNACConfig     <- NeuralCodecs.Torch/Config/SNAC/SNACConfig.cs:40-153
  * EncodecConfig  <- NeuralCodecs.Torch/Config/Encodec/EncodecConfig.cs:9-64
                      (only channels/dimension/norm/causal/bandwidths reach SEANet,
                       Models/Encodec.cs:57-68; the rest are hard defaults of
                       SEANetEncoder.cs:37-56 -- SURVEY D11)
"""
from __future__ import annotations

from dataclasses import dataclass, field
from functools import reduce
from math import gcd
from typing import List, Optional, Tuple


@dataclass
class DACConfig:
    sample_rate: int = 44100
    encoder_dim: int = 64
    encoder_rates: Tuple[int, ...] = (2, 4, 8, 8)
    decoder_dim: int = 1536
    decoder_rates: Tuple[int, ...] = (8, 8, 4, 2)
    latent_dim: Optional[int] = None          # DAC.cs:64  -> encoder_dim * 2**len(rates)
    n_codebooks: int = 9
    codebook_size: int = 1024
    codebook_dim: int = 8
    architecture: str = "dac"

    @property
    def resolved_latent_dim(self) -> int:
        return self.latent_dim if self.latent_dim is not None else self.encoder_dim * (1 << len(self.encoder_rates))

    @property
    def hop_length(self) -> int:              # DAC.cs:67
        return reduce(lambda a, b: a * b, self.encoder_rates)

    # presets: DACConfig.cs:77-113
    @staticmethod
    def dac_44khz() -> "DACConfig":
        return DACConfig()

    @staticmethod
    def dac_44khz_16kbps() -> "DACConfig":
        return DACConfig(n_codebooks=18, latent_dim=128)

    @staticmethod
    def dac_24khz() -> "DACConfig":
        return DACConfig(sample_rate=24000, n_codebooks=32, encoder_rates=(2, 4, 5, 8), decoder_rates=(8, 5, 4, 2))

    @staticmethod
    def dac_16khz() -> "DACConfig":
        return DACConfig(sample_rate=16000, n_codebooks=12, encoder_rates=(2, 4, 5, 8), decoder_rates=(8, 5, 4, 2))


@dataclass
class SNACConfig:
    sampling_rate: int = 24000
    encoder_dim: int = 48
    encoder_rates: Tuple[int, ...] = (2, 4, 8, 8)
    latent_dim: Optional[int] = None
    decoder_dim: int = 1024
    decoder_rates: Tuple[int, ...] = (8, 8, 4, 2)
    attn_window_size: Optional[int] = None
    codebook_size: int = 4096
    codebook_dim: int = 8
    vq_strides: Tuple[int, ...] = (4, 2, 1)
    noise: bool = True
    depthwise: bool = True
    architecture: str = "snac"

    @property
    def resolved_latent_dim(self) -> int:
        return self.latent_dim if self.latent_dim is not None else self.encoder_dim * (1 << len(self.encoder_rates))

    @property
    def hop_length(self) -> int:
        return reduce(lambda a, b: a * b, self.encoder_rates)

    @property
    def pad_multiple(self) -> int:
        """SNAC.Preprocess: hop * lcm(vq_strides[0], attn_window or 1) (Models/SNAC.cs:70-80)."""
        a, b = self.vq_strides[0], (self.attn_window_size or 1)
        return self.hop_length * (a * b // gcd(a, b))

    @staticmethod
    def snac_24khz() -> "SNACConfig":
        return SNACConfig()

    @staticmethod
    def snac_32khz() -> "SNACConfig":
        return SNACConfig(sampling_rate=32000, encoder_dim=64, encoder_rates=(2, 3, 8, 8), decoder_dim=1536,
                          decoder_rates=(8, 8, 3, 2), attn_window_size=32, vq_strides=(8, 4, 2, 1))

    @staticmethod
    def snac_44khz() -> "SNACConfig":
        return SNACConfig(sampling_rate=44100, encoder_dim=64, encoder_rates=(2, 3, 8, 8), decoder_dim=1536,
                          decoder_rates=(8, 8, 3, 2), attn_window_size=32, vq_strides=(8, 4, 2, 1))


@dataclass
class EncodecConfig:
    sampling_rate: int = 24000
    channels: int = 1
    dimension: int = 128
    norm: str = "weight_norm"                 # 48 kHz preset: "time_group_norm"
    causal: bool = True
    normalize: bool = False
    segment_seconds: Optional[float] = None   # 48 kHz: 1.0
    overlap: float = 0.01
    target_bandwidths: Tuple[float, ...] = (1.5, 3.0, 6.0, 12.0, 24.0)
    bandwidth: float = 6.0
    codebook_size: int = 1024
    # SEANet hard defaults (SEANetEncoder.cs:37-56) -- not configurable in the reference (D11)
    n_filters: int = 32
    ratios: Tuple[int, ...] = (8, 5, 4, 2)
    lstm_layers: int = 2
    compress: int = 2
    n_residual_layers: int = 1
    kernel_size: int = 7
    last_kernel_size: int = 7
    residual_kernel_size: int = 3
    dilation_base: int = 2
    architecture: str = "encodec"

    @property
    def hop_length(self) -> int:
        return reduce(lambda a, b: a * b, self.ratios)

    @staticmethod
    def encodec_24khz() -> "EncodecConfig":
        return EncodecConfig()

    @staticmethod
    def encodec_48khz() -> "EncodecConfig":
        return EncodecConfig(sampling_rate=48000, channels=2, norm="time_group_norm", causal=False, normalize=True,
                             segment_seconds=1.0, target_bandwidths=(3.0, 6.0, 12.0, 24.0), bandwidth=12.0)
