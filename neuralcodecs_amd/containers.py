"""Code containers either side of the Encode/Decode path.

* `.ecdc` (Encodec): NeuralCodecs.Torch/Modules/Encodec/BinaryIO.cs (magic "ECDC", version byte 0, big-endian u32 JSON length,
  UTF-8 JSON metadata m/al/nc/lm/ch/sr/bw) + EncodecCompressor.cs:64-189 without the language model: per frame an optional
  big-endian (count=1, float32 scale) record, then the codes bit-packed `BitsPerCodebook` bits each, LSB first, t outer / codebook
  inner, flushed to a byte boundary (BitPacker.cs).  The LM / arithmetic-coder branch is out of scope.
  The packing itself runs on the GPU (`nc_pack_codes`); this module is the host-side framing.
* `DACFile` (AudioTools/DACFile.cs:27-105): .NET BinaryWriter framing -- int32 config-length, 7-bit-length-prefixed UTF-8 JSON
  config, int32 tensor count, then per tensor int32 rank, int64 dims, int32 element count, int32 elements.

Reader note: the reference's decompressor recomputes the frame count of a segment as ceil(len * frame_rate / sample_rate)
(EncodecCompressor.cs:300-302), which disagrees with its own encoder on tails that take SConv1d's small-input path (D9: the
960-sample tail of a 2 s 48 kHz clip has 4 frames, the formula says 3).  The reader here asks the engine for the frame layout, so
such files round-trip.
"""
from __future__ import annotations

import ctypes as C
import io
import json
import struct
from typing import BinaryIO, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib

ECDC_MAGIC = b"ECDC"


# ---------------------------------------------------------------------------------------------- bit packing (device)
def pack_codes(codes: np.ndarray, bits: int, device_index: int = 0) -> np.ndarray:
    """codes [B,K,T] integer -> uint8 [B, ceil(K*T*bits/8)] in the reference's BitPacker layout (t outer, k inner)."""
    c = np.ascontiguousarray(codes, np.int64)
    if c.ndim != 3:
        raise ValueError("codes must be [B,K,T]")
    if (c < 0).any() or (c >= (1 << bits)).any():
        raise ValueError(f"Value must be between 0 and {(1 << bits) - 1}")                  # BitPacker.cs:70-76
    B, K, T = c.shape
    out = np.empty((B, _lib.lib().nc_packed_bytes(K * T, bits)), np.uint8)
    _lib.check(_lib.lib().nc_pack_codes(device_index, c.ctypes.data, B, K, T, bits, out.ctypes.data))
    return out


def unpack_codes(packed: np.ndarray, K: int, T: int, bits: int, device_index: int = 0) -> np.ndarray:
    p = np.ascontiguousarray(packed, np.uint8)
    B = p.shape[0]
    need = _lib.lib().nc_packed_bytes(K * T, bits)
    if p.ndim != 2 or p.shape[1] < need:
        raise EOFError("Stream ended too soon")                                            # EncodecCompressor.cs:385-388
    p = np.ascontiguousarray(p[:, :need])
    out = np.empty((B, K, T), np.int64)
    _lib.check(_lib.lib().nc_unpack_codes(device_index, p.ctypes.data, B, K, T, bits, out.ctypes.data))
    return out


# ---------------------------------------------------------------------------------------------- .ecdc
def ecdc_write_header(stream: BinaryIO, metadata: Dict[str, object]) -> None:
    meta = json.dumps(metadata, separators=(",", ":")).encode("utf-8")
    stream.write(ECDC_MAGIC + bytes([0]) + struct.pack(">I", len(meta)) + meta)


def ecdc_read_header(stream: BinaryIO) -> Dict[str, object]:
    magic = stream.read(4)
    if len(magic) != 4:
        raise IOError("Failed to read Encodec header: incomplete header magic number")
    if magic != ECDC_MAGIC:
        raise ValueError("Invalid Encodec header magic number")                             # InvalidDataException
    ver = stream.read(1)
    if len(ver) != 1 or ver[0] != 0:
        raise ValueError(f"Unsupported header version: {ver[0] if ver else -1}")
    lb = stream.read(4)
    if len(lb) != 4:
        raise IOError("Failed to read Encodec header: incomplete metadata length")
    (n,) = struct.unpack(">I", lb)
    body = stream.read(n)
    if len(body) != n:
        raise IOError("Failed to read Encodec header: incomplete metadata")
    meta = json.loads(body.decode("utf-8"))
    for k in ("m", "al", "nc", "lm"):
        if k not in meta:
            raise ValueError(f"Missing required metadata key: {k}")                         # BinaryIO.ValidateMetadata
    return meta


def ecdc_compress(model, wav: np.ndarray, stream: Optional[BinaryIO] = None) -> bytes:
    """EncodecCompressor.CompressToStreamAsync(model, wav [C,L], stream, useLm: false)."""
    wav = np.asarray(wav, np.float32)
    if wav.ndim != 2:
        raise ValueError("Only single waveform can be encoded (shape should be [C, L])")
    if wav.shape[0] != model.config.channels:
        raise ValueError(f"Expected {model.config.channels} channels, got {wav.shape[0]}")
    own = stream is None
    stream = stream or io.BytesIO()
    frames = model.encode(wav[None])
    meta = {"m": f"encodec_{model.config.sampling_rate // 1000}khz", "al": int(wav.shape[-1]), "nc": int(frames[0].codes.shape[1]),
            "lm": False, "ch": int(wav.shape[0]), "sr": int(model.config.sampling_rate), "bw": float(model.config.bandwidth)}
    ecdc_write_header(stream, meta)
    for f in frames:
        if f.scale is not None:
            stream.write(struct.pack(">i", 1) + struct.pack(">f", float(np.asarray(f.scale).reshape(-1)[0])))
        stream.write(pack_codes(np.asarray(f.codes), model.bits_per_codebook)[0].tobytes())
    return stream.getvalue() if own else b""


def ecdc_read_frames(model, stream: BinaryIO):
    """-> (metadata, frames) of a no-LM `.ecdc` stream; the model must match the metadata (channels, bandwidth)."""
    from .encodec import EncodedFrame
    meta = ecdc_read_header(stream)
    if bool(meta["lm"]):
        raise NotImplementedError("language-model coded .ecdc streams are out of scope (SURVEY 2.1 row 3b)")
    if int(meta.get("ch", 1)) != model.config.channels:
        raise ValueError(f"Model has {model.config.channels} channels but compressed data has {meta.get('ch', 1)} channels")
    if "bw" in meta and float(meta["bw"]) != float(model.config.bandwidth):
        model.set_target_bandwidth(float(meta["bw"]))
    T = int(meta["al"])
    K = int(meta["nc"])
    _, _, lens, _ = model.query(T)
    bits = model.bits_per_codebook
    frames = []
    for Tf in lens:
        scale = None
        if model.config.normalize:
            rec = stream.read(4)
            if len(rec) != 4:
                raise EOFError("Stream ended too soon")
            (n,) = struct.unpack(">i", rec)
            if n <= 0 or n > 1000:
                raise ValueError(f"Invalid scale count: {n}")
            vals = stream.read(4 * n)
            scale = np.array(struct.unpack(">" + "f" * n, vals), np.float32).reshape(1, -1)[:, :1]
        nb = _lib.lib().nc_packed_bytes(K * Tf, bits)
        buf = stream.read(nb)
        if len(buf) != nb:
            raise EOFError("Stream ended too soon")
        frames.append(EncodedFrame(unpack_codes(np.frombuffer(buf, np.uint8)[None], K, Tf, bits), scale))
    return meta, frames


def ecdc_decompress(model, data: bytes) -> Tuple[np.ndarray, int]:
    """EncodecCompressor.DecompressAsync -> (waveform [C, al], sample_rate)."""
    meta, frames = ecdc_read_frames(model, io.BytesIO(data))
    wav = model.decode(frames, int(meta["al"]))
    return wav[0, :, : int(meta["al"])], int(meta.get("sr", model.config.sampling_rate))


# ---------------------------------------------------------------------------------------------- DACFile
def _write_7bit_string(stream: BinaryIO, s: str) -> None:
    b = s.encode("utf-8")
    n = len(b)
    while n >= 0x80:
        stream.write(bytes([(n & 0x7F) | 0x80]))
        n >>= 7
    stream.write(bytes([n]))
    stream.write(b)


def _read_7bit_string(stream: BinaryIO) -> str:
    n, shift = 0, 0
    while True:
        c = stream.read(1)
        if not c:
            raise EOFError("truncated DACFile")
        n |= (c[0] & 0x7F) << shift
        if not c[0] & 0x80:
            break
        shift += 7
    return stream.read(n).decode("utf-8")


def dac_config_json(cfg) -> str:
    """The reference serialises DACConfig with System.Text.Json; property names per Config/DAC/DACConfig.cs:22-76."""
    return json.dumps({"model_type": cfg.architecture, "codebook_dim": cfg.codebook_dim, "codebook_size": cfg.codebook_size,
                       "decoder_hidden_size": cfg.decoder_dim, "upsampling_ratios": list(cfg.decoder_rates),
                       "encoder_hidden_size": cfg.encoder_dim, "downsampling_ratios": list(cfg.encoder_rates), "hop_length": cfg.hop_length,
                       "n_codebooks": cfg.n_codebooks, "sampling_rate": cfg.sample_rate}, separators=(",", ":"))


def dacfile_save(path_or_stream, codes: Sequence[np.ndarray], cfg) -> None:
    """DACFile.SaveAsync (AudioTools/DACFile.cs:62-88)."""
    own = isinstance(path_or_stream, (str, bytes))
    f = open(path_or_stream, "wb") if own else path_or_stream
    try:
        js = dac_config_json(cfg)
        f.write(struct.pack("<i", len(js)))
        _write_7bit_string(f, js)
        f.write(struct.pack("<i", len(codes)))
        for c in codes:
            c = np.asarray(c)
            f.write(struct.pack("<i", c.ndim))
            for d in c.shape:
                f.write(struct.pack("<q", int(d)))
            flat = np.ascontiguousarray(c, np.int32).reshape(-1)
            f.write(struct.pack("<i", flat.size))
            f.write(flat.astype("<i4").tobytes())
    finally:
        if own:
            f.close()


def dacfile_load(path_or_stream):
    """DACFile.LoadAsync (AudioTools/DACFile.cs:27-56) -> (codes list of int64 arrays, config dict)."""
    own = isinstance(path_or_stream, (str, bytes))
    f = open(path_or_stream, "rb") if own else path_or_stream
    try:
        struct.unpack("<i", f.read(4))
        cfg = json.loads(_read_7bit_string(f))
        (n,) = struct.unpack("<i", f.read(4))
        codes = []
        for _ in range(n):
            (rank,) = struct.unpack("<i", f.read(4))
            shape = [struct.unpack("<q", f.read(8))[0] for _ in range(rank)]
            (cnt,) = struct.unpack("<i", f.read(4))
            data = np.frombuffer(f.read(4 * cnt), "<i4")
            if data.size != cnt:
                raise EOFError("truncated DACFile")
            codes.append(data.astype(np.int64).reshape(shape))
        return codes, cfg
    finally:
        if own:
            f.close()
