"""Checkpoint import: real DAC / SNAC / Encodec checkpoints -> the engine's NCWB0001 weight blob.

Restates the reference's key handling (host-side, no arithmetic on the hot path):
  * NeuralCodecs.Torch/Config/DAC/StateDictNameConverter.cs:274-340 (`BuildKeyMap`), :342-376 (`TranslateKey`), :36-60
    (`ConvertFromSafetensor`: a plain HF `weight` becomes `weight_v = weight`, `weight_g = ||weight||` over dims (1,2), float32)
  * NeuralCodecs.Torch/Config/DAC/DACUnpickler.cs:353-433: a Descript `.pth` is a zip-pickle holding {state_dict, metadata}; the
    metadata kwargs become the DACConfig
  * SNAC checkpoints already use `...parametrizations.weight.original0/1` (Modules/SNAC/WNConv1d.cs:66-70; Models/SNAC.cs:216-231):
    pass-through.  Encodec checkpoints keyed like the reference's modules (SConv1d.cs:110-128) pass through as well.

    python tools/convert_checkpoint.py --codec dac  weights.safetensors  out.ncwb
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import numpy as np

from .config import DACConfig
from .weights import save_blob


def dac_key_map(resunit_count: int = 3, decoder_blocks: int = 4, encoder_blocks: int = 4) -> Dict[str, str]:
    """StateDictNameConverter.BuildKeyMap: HF-safetensors module prefix -> TorchSharp module prefix."""
    m: Dict[str, str] = {}
    m["decoder.conv1"] = "decoder.model.0"
    m["decoder.snake1"] = f"decoder.model.{decoder_blocks + 1}"
    m["decoder.conv2"] = f"decoder.model.{decoder_blocks + 2}"
    for b in range(decoder_blocks):
        m[f"decoder.block.{b}.snake1"] = f"decoder.model.{b + 1}.block.0"
        m[f"decoder.block.{b}.conv_t1"] = f"decoder.model.{b + 1}.block.1"
        for u in range(1, resunit_count + 1):
            s, t = f"decoder.block.{b}.res_unit{u}", f"decoder.model.{b + 1}.block.{u + 1}"
            m[f"{s}.snake1"], m[f"{s}.conv1"], m[f"{s}.snake2"], m[f"{s}.conv2"] = (f"{t}.block.0", f"{t}.block.1", f"{t}.block.2",
                                                                                 f"{t}.block.3")
    m["encoder.conv1"] = "encoder.block.0"
    m["encoder.snake1"] = f"encoder.block.{encoder_blocks + 1}"
    m["encoder.conv2"] = f"encoder.block.{encoder_blocks + 2}"
    for b in range(encoder_blocks):
        m[f"encoder.block.{b}.snake1"] = f"encoder.block.{b + 1}.block.3"
        m[f"encoder.block.{b}.conv1"] = f"encoder.block.{b + 1}.block.4"
        for u in range(1, resunit_count + 1):
            s, t = f"encoder.block.{b}.res_unit{u}", f"encoder.block.{b + 1}.block.{u - 1}"
            m[f"{s}.snake1"], m[f"{s}.conv1"], m[f"{s}.snake2"], m[f"{s}.conv2"] = (f"{t}.block.0", f"{t}.block.1", f"{t}.block.2",
                                                                                 f"{t}.block.3")
    return m


def _weight_norm_split(w: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    w = np.ascontiguousarray(w, np.float32)
    g = np.sqrt((w * w).sum(axis=(1, 2), keepdims=True, dtype=np.float32)).astype(np.float32)
    return w, g


def is_hf_dac(sd: Dict[str, np.ndarray]) -> bool:
    return any(k.startswith("encoder.conv1.") or ".res_unit" in k or ".conv_t1." in k for k in sd)


def convert_dac_state_dict(sd: Dict[str, np.ndarray]) -> "OrderedDict[str, np.ndarray]":
    """Any supported DAC naming -> TorchSharp names (weight_v / weight_g / bias / alpha / codebook.weight)."""
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    if not is_hf_dac(sd):                                        # native / Descript names: already what the engine loads
        for k, v in sd.items():
            out[k] = np.ascontiguousarray(np.asarray(v), np.float32)
        return out
    n_dec = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("decoder.block."))
    n_enc = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("encoder.block."))
    km = dac_key_map(3, n_dec, n_enc)
    for k, v in sd.items():
        v = np.ascontiguousarray(np.asarray(v), np.float32)
        base = k
        for suf in (".weight", ".bias", ".alpha"):
            if suf in k:
                base = k.split(suf)[0]
                break
        if base in km:
            t = km[base]
            if k.endswith(".weight"):
                out[t + ".weight_v"], out[t + ".weight_g"] = _weight_norm_split(v)
            elif k.endswith(".bias"):
                out[t + ".bias"] = v
            elif k.endswith(".alpha"):
                out[t + ".alpha"] = v.reshape(1, -1, 1)
            else:
                out[t + k[len(base):]] = v
        elif k.endswith(".weight") and (".in_proj" in base or ".out_proj" in base):
            out[base + ".weight_v"], out[base + ".weight_g"] = _weight_norm_split(v)
        else:
            out[k] = v                                            # quantizer biases, codebook.weight, anything already native
    return out


def dac_config_from_metadata(meta: Optional[dict]) -> Optional[DACConfig]:
    """DACUnpickler.cs:412-433: metadata["kwargs"] of a Descript checkpoint -> config."""
    if not meta:
        return None
    kw = meta.get("kwargs", meta)
    f = {}
    for src, dst in (("sample_rate", "sample_rate"), ("encoder_dim", "encoder_dim"), ("decoder_dim", "decoder_dim"),
                     ("n_codebooks", "n_codebooks"), ("codebook_size", "codebook_size"), ("codebook_dim", "codebook_dim"),
                     ("latent_dim", "latent_dim")):
        if src in kw and kw[src] is not None:
            f[dst] = int(kw[src])
    if "encoder_rates" in kw:
        f["encoder_rates"] = tuple(int(x) for x in kw["encoder_rates"])
    if "decoder_rates" in kw:
        f["decoder_rates"] = tuple(int(x) for x in kw["decoder_rates"])
    return DACConfig(**f)


def read_checkpoint(path: str) -> Tuple[Dict[str, np.ndarray], Optional[dict]]:
    """safetensors, or a torch zip-pickle holding a state dict / {state_dict, metadata}.  FileNotFoundError like LoadWeights."""
    if not os.path.exists(path):
        raise FileNotFoundError(f"Weights not found at {path}")
    if path.endswith(".safetensors"):
        from safetensors.numpy import load_file
        return dict(load_file(path)), None
    import torch
    obj = torch.load(path, map_location="cpu", weights_only=True)
    meta = None
    if isinstance(obj, dict) and "state_dict" in obj:
        meta = obj.get("metadata")
        obj = obj["state_dict"]
    return {k: v.detach().cpu().float().numpy() for k, v in obj.items() if hasattr(v, "detach")}, meta


def convert_checkpoint(path: str, codec: str = "dac") -> Tuple[bytes, Optional[object]]:
    """-> (NCWB0001 blob, config inferred from the checkpoint metadata or None)."""
    sd, meta = read_checkpoint(path)
    if codec == "dac":
        return save_blob(convert_dac_state_dict(sd)), dac_config_from_metadata(meta)
    if codec in ("snac", "encodec"):
        keep = OrderedDict((k, np.ascontiguousarray(v, np.float32)) for k, v in sd.items()
                           if not k.endswith(("cluster_size", "embed_avg", "inited")))     # EuclideanCodebook.cs:60-66 training buffers
        return save_blob(keep), None
    raise ValueError(f"unknown codec {codec}")
