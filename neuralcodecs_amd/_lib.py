"""ctypes binding of libnc_mi355x.so (the C ABI declared in include/nc_mi355x.h).

The library is the product: if it is missing or cannot be loaded this module raises -- there is
no Python/torch fallback for any operator.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnc_mi355x.so")

NC_OK, NC_EINVAL, NC_ENOTFOUND, NC_ESTATE, NC_EDEVICE, NC_ENOMEM, NC_EUNSUPPORTED = range(7)
NC_KC_NAMES = ("conv_k7", "conv_k1", "conv_down", "conv_up", "conv_misc", "rvq", "elem", "dwconv", "norm", "attn", "lstm", "stem", "head")


class NcError(RuntimeError):
    """Base of engine errors (the reference's NeuralCodecException, Core/Exceptions/NeuralCodecException.cs:10-73)."""


class NcDeviceError(NcError):
    pass


class NcDacConfig(C.Structure):
    _fields_ = [("sample_rate", C.c_int32), ("encoder_dim", C.c_int32), ("n_encoder_rates", C.c_int32),
                ("encoder_rates", C.c_int32 * 8), ("decoder_dim", C.c_int32), ("n_decoder_rates", C.c_int32),
                ("decoder_rates", C.c_int32 * 8), ("latent_dim", C.c_int32), ("n_codebooks", C.c_int32),
                ("codebook_size", C.c_int32), ("codebook_dim", C.c_int32)]


class NcSnacConfig(C.Structure):
    _fields_ = [("sample_rate", C.c_int32), ("encoder_dim", C.c_int32), ("n_encoder_rates", C.c_int32),
                ("encoder_rates", C.c_int32 * 8), ("decoder_dim", C.c_int32), ("n_decoder_rates", C.c_int32),
                ("decoder_rates", C.c_int32 * 8), ("latent_dim", C.c_int32), ("attn_window_size", C.c_int32),
                ("codebook_size", C.c_int32), ("codebook_dim", C.c_int32), ("n_vq_strides", C.c_int32),
                ("vq_strides", C.c_int32 * 8), ("noise", C.c_int32), ("depthwise", C.c_int32)]


class NcEncodecConfig(C.Structure):
    _fields_ = [("sample_rate", C.c_int32), ("channels", C.c_int32), ("dimension", C.c_int32), ("n_filters", C.c_int32),
                ("n_ratios", C.c_int32), ("ratios", C.c_int32 * 8), ("lstm_layers", C.c_int32), ("compress", C.c_int32),
                ("kernel_size", C.c_int32), ("last_kernel_size", C.c_int32), ("residual_kernel_size", C.c_int32),
                ("time_group_norm", C.c_int32), ("causal", C.c_int32), ("normalize", C.c_int32), ("segment_length", C.c_int32),
                ("segment_stride", C.c_int32), ("codebook_size", C.c_int32), ("n_codebooks", C.c_int32), ("frame_rate", C.c_int32),
                ("bandwidth", C.c_float)]


class NcProfileEntry(C.Structure):
    _fields_ = [("launches", C.c_int64), ("ms", C.c_double), ("flops", C.c_double), ("bytes", C.c_double)]


class NcConvDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32), ("K", C.c_int32), ("stride", C.c_int32),
                ("pad", C.c_int32), ("dil", C.c_int32), ("out_pad", C.c_int32), ("Tin", C.c_int64),
                ("transposed", C.c_int32), ("tanh_out", C.c_int32)]


_lib = None

# every symbol include/nc_mi355x.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("nc_last_error", C.c_char_p, []),
    ("nc_version", C.c_char_p, []),
    ("nc_device_count", C.c_int, []),
    ("nc_debug_switches", C.c_char_p, []),
    ("nc_dac_create", C.c_int, [C.POINTER(NcDacConfig), C.c_int, C.POINTER(_P)]),
    ("nc_codec_destroy", C.c_int, [_P]),
    ("nc_codec_load_weights", C.c_int, [_P, C.c_char_p]),
    ("nc_codec_load_weights_mem", C.c_int, [_P, _P, C.c_size_t]),
    ("nc_blob_check", C.c_int, [_P, C.c_size_t, C.POINTER(C.c_int32)]),
    ("nc_codec_set_stream", C.c_int, [_P, _P]),
    ("nc_codec_reset_stream", C.c_int, [_P]),
    ("nc_codec_synchronize", C.c_int, [_P]),
    ("nc_codec_check_errors", C.c_int, [_P]),
    ("nc_encodec_lstm_stats", C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    ("nc_dac_query", C.c_int, [_P, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("nc_dac_encode", C.c_int, [_P, _P, C.c_int32, C.c_int64, C.c_int32, C.c_int32, _P, _P, _P]),
    ("nc_dac_encode_dev", C.c_int, [_P, _P, C.c_int32, C.c_int64, C.c_int32, C.c_int32, _P, _P, _P]),
    ("nc_dac_decode", C.c_int, [_P, _P, C.c_int32, C.c_int64, _P]),
    ("nc_dac_decode_dev", C.c_int, [_P, _P, C.c_int32, C.c_int64, _P]),
    ("nc_dac_from_codes", C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int64, _P]),
    ("nc_dac_from_codes_dev", C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int64, _P]),
    ("nc_dac_decode_code_matrix", C.c_int, [_P, _P, C.c_int32, C.c_int64, C.c_int32, _P]),
    ("nc_dac_decode_code_matrix_dev", C.c_int, [_P, _P, C.c_int32, C.c_int64, C.c_int32, _P]),
    ("nc_dac_encode_code_matrix", C.c_int, [_P, _P, C.c_int32, C.c_int64, C.c_int32, _P]),
    ("nc_dac_encode_code_matrix_dev", C.c_int, [_P, _P, C.c_int32, C.c_int64, C.c_int32, _P]),
    ("nc_snac_create", C.c_int, [C.POINTER(NcSnacConfig), C.c_int, C.POINTER(_P)]),
    ("nc_snac_query", C.c_int, [_P, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                                C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("nc_snac_noise_len", C.c_int, [_P, C.c_int32, C.c_int64, C.POINTER(C.c_int64)]),
    ("nc_snac_process_audio_len", C.c_int, [_P, C.c_int64, C.c_int32, C.POINTER(C.c_int64)]),
    ("nc_snac_process_audio", C.c_int, [_P, _P, C.c_int64, C.c_int32, _P, C.c_uint64, _P]),
    ("nc_snac_encode", C.c_int, [_P, _P, C.c_int32, C.c_int64, _P, _P, _P]),
    ("nc_snac_encode_dev", C.c_int, [_P, _P, C.c_int32, C.c_int64, _P, _P, _P]),
    ("nc_snac_query_tensor", C.c_int, [_P, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    ("nc_snac_encode_tensor", C.c_int, [_P, _P, C.c_int32, C.c_int64, _P, _P, _P]),
    ("nc_snac_encode_tensor_dev", C.c_int, [_P, _P, C.c_int32, C.c_int64, _P, _P, _P]),
    ("nc_snac_from_codes", C.c_int, [_P, _P, C.c_int32, C.c_int64, _P]),
    ("nc_snac_from_codes_dev", C.c_int, [_P, _P, C.c_int32, C.c_int64, _P]),
    ("nc_snac_decode", C.c_int, [_P, _P, C.c_int32, C.c_int64, _P, C.c_uint64, _P]),
    ("nc_snac_decode_dev", C.c_int, [_P, _P, C.c_int32, C.c_int64, _P, C.c_uint64, _P]),
    ("nc_encodec_create", C.c_int, [C.POINTER(NcEncodecConfig), C.c_int, C.POINTER(_P)]),
    ("nc_encodec_set_bandwidth", C.c_int, [_P, C.c_float]),
    ("nc_encodec_query", C.c_int, [_P, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.c_int32,
                                   C.POINTER(C.c_int64)]),
    ("nc_encodec_clip_length", C.c_int, [_P, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("nc_encodec_encode", C.c_int, [_P, _P, C.c_int32, C.c_int64, _P, _P, _P]),
    ("nc_encodec_encode_dev", C.c_int, [_P, _P, C.c_int32, C.c_int64, _P, _P, _P]),
    ("nc_encodec_decode", C.c_int, [_P, _P, _P, C.c_int32, C.c_int64, C.c_int32, _P]),
    ("nc_encodec_decode_dev", C.c_int, [_P, _P, _P, C.c_int32, C.c_int64, C.c_int32, _P]),
    ("nc_packed_bytes", C.c_int64, [C.c_int64, C.c_int32]),
    ("nc_pack_codes_dev", C.c_int, [C.c_int, _P, C.c_int32, C.c_int32, C.c_int64, C.c_int32, _P, _P]),
    ("nc_unpack_codes_dev", C.c_int, [C.c_int, _P, C.c_int32, C.c_int32, C.c_int64, C.c_int32, _P, _P]),
    ("nc_pack_codes", C.c_int, [C.c_int, _P, C.c_int32, C.c_int32, C.c_int64, C.c_int32, _P]),
    ("nc_unpack_codes", C.c_int, [C.c_int, _P, C.c_int32, C.c_int32, C.c_int64, C.c_int32, _P]),
    ("nc_audio_resample_len", C.c_int64, [C.c_int64, C.c_int32, C.c_int32]),
    ("nc_audio_pcm16_to_float_dev", C.c_int, [C.c_int, _P, C.c_int64, C.c_int32, C.c_int32, _P, _P]),
    ("nc_audio_float_to_pcm16_dev", C.c_int, [C.c_int, _P, C.c_int64, _P, _P]),
    ("nc_audio_mix_to_mono_dev", C.c_int, [C.c_int, _P, C.c_int64, C.c_int32, _P, _P]),
    ("nc_audio_interleave_dev", C.c_int, [C.c_int, _P, C.c_int64, C.c_int32, _P, _P]),
    ("nc_audio_deinterleave_dev", C.c_int, [C.c_int, _P, C.c_int64, C.c_int32, _P, _P]),
    ("nc_audio_resample_linear_dev", C.c_int, [C.c_int, _P, C.c_int32, C.c_int64, C.c_int32, C.c_int32, _P, _P]),
    ("nc_group_unique_id", C.c_int, [_P]),
    ("nc_group_create_rank", C.c_int, [C.c_int32, C.c_int32, _P, _P, C.POINTER(_P)]),
    ("nc_group_create_local", C.c_int, [C.c_int32, C.POINTER(_P), C.POINTER(_P)]),
    ("nc_group_create_local_ex", C.c_int, [C.c_int32, C.POINTER(_P), C.c_uint32, C.POINTER(_P)]),
    ("nc_group_destroy", C.c_int, [_P]),
    ("nc_group_info", C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("nc_group_set_code_bits", C.c_int, [_P, C.c_int32]),
    ("nc_group_dac_encode_allgather_dev", C.c_int, [_P, _P, C.c_int32, C.c_int64, C.c_int32, C.c_int32, _P, _P, _P]),
    ("nc_group_snac_encode_allgather_dev", C.c_int, [_P, _P, C.c_int32, C.c_int64, _P]),
    ("nc_group_wait", C.c_int, [_P]),
    ("nc_group_dac_encode_allgather", C.c_int, [_P, _P, C.c_int32, C.c_int64, C.c_int32, C.c_int32, _P, _P]),
    ("nc_group_snac_encode_allgather", C.c_int, [_P, _P, C.c_int32, C.c_int64, _P]),
    ("nc_group_dac_encode_allgather_local_dev", C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_int32), C.c_int64, C.c_int32, C.c_int32, C.POINTER(_P),
                                                          C.POINTER(_P), C.POINTER(_P)]),
    ("nc_group_snac_encode_allgather_local_dev", C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_int32), C.c_int64, C.POINTER(_P)]),
    ("nc_group_encodec_encode_allgather_dev", C.c_int, [_P, _P, C.c_int32, C.c_int64, _P, _P]),
    ("nc_group_encodec_encode_allgather_local_dev", C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_int32), C.c_int64, C.POINTER(_P), C.POINTER(_P)]),
    ("nc_codec_profile_enable", C.c_int, [_P, C.c_int32]),
    ("nc_codec_profile_reset", C.c_int, [_P]),
    ("nc_codec_profile_read", C.c_int, [_P, C.POINTER(NcProfileEntry)]),
    ("nc_op_conv1d", C.c_int, [C.c_int, C.POINTER(NcConvDesc), _P, _P, _P, _P, _P, _P, _P, C.POINTER(C.c_int64)]),
    ("nc_op_conv1d_bench", C.c_int, [C.c_int, C.POINTER(NcConvDesc), C.c_int32, C.c_int32, C.POINTER(C.c_double)]),
    ("nc_op_res_unit", C.c_int, [C.c_int, C.c_int32, C.c_int32, C.c_int64, C.c_int32, _P, _P, _P, _P, _P, _P, _P, C.c_int32, _P,
                                 C.c_int32, C.POINTER(C.c_double)]),
    ("nc_op_vq_argmin", C.c_int, [C.c_int, _P, C.c_int32, C.c_int32, C.c_int64, _P, C.c_int32, _P, _P]),
    ("nc_op_euclid_rvq", C.c_int, [C.c_int, _P, C.c_int32, C.c_int32, C.c_int64, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    ("nc_op_fold_weight_norm", C.c_int, [_P, _P, C.c_int64, C.c_int64, _P]),
]


def lib():
    """Load the engine.  Importing torch first (when present) makes the loader bind the already
    loaded libamdhip64.so.7 of the torch wheel, so both share one HIP runtime in this process."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("NC_MI355X_LIB") or LIB_PATH          # (diagnostic: A/B of library builds, tools/probe/ab_libs*.sh)
    if not os.path.exists(path):
        raise NcError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(make -C neuralcodecs_amd/csrc). There is no fallback implementation.")
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401  (shares its HIP runtime with the engine)
        except Exception:
            pass
    L = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, res, args in SYMBOLS:
        try:
            fn = getattr(L, name)
        except AttributeError:
            # A/B against an OLDER build of the library (NC_MI355X_LIB=<path>): only the exports named in NC_ALLOW_MISSING_EXPORTS (comma
            # separated) may be absent -- a symbol dropped by accident from any build must fail here, not as an AttributeError later
            if os.environ.get("NC_MI355X_LIB") and name in [x.strip() for x in os.environ.get("NC_ALLOW_MISSING_EXPORTS", "").split(",")]:
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def lib_path() -> str:
    return os.environ.get("NC_MI355X_LIB") or LIB_PATH


def lib_sha256() -> str:
    """SHA-256 of the engine library file this process loads: profiles/traffic.json records the one its counters were measured on."""
    import hashlib
    h = hashlib.sha256()
    with open(lib_path(), "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def check(status: int) -> None:
    if status == NC_OK:
        return
    msg = lib().nc_last_error().decode(errors="replace")
    if status == NC_EINVAL:
        raise ValueError(msg)              # ArgumentException / ArgumentNullException
    if status == NC_ENOTFOUND:
        raise FileNotFoundError(msg)       # FileNotFoundException
    if status == NC_ESTATE:
        raise RuntimeError(msg)            # InvalidOperationException
    if status == NC_ENOMEM:
        raise MemoryError(msg)
    if status == NC_EDEVICE:
        raise NcDeviceError(msg)
    raise NcError(f"status {status}: {msg}")


class ProfileMixin:
    """HIP-event kernel-class profile of a codec handle (nc_codec_profile_*): the numbers bench.py's `roofline` objects come from."""

    def check_errors(self) -> None:
        """nc_codec_check_errors: device-side failures of an earlier device-pointer call (raises NcDeviceError); call it once the
        caller's own synchronisation (torch.cuda.synchronize) has made the stream idle."""
        check(lib().nc_codec_check_errors(self._h))

    def profile_enable(self, on: bool = True):
        check(lib().nc_codec_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self):
        check(lib().nc_codec_profile_reset(self._h))

    def profile_read(self):
        arr = (NcProfileEntry * len(NC_KC_NAMES))()
        check(lib().nc_codec_profile_read(self._h, arr))
        return {n: {"launches": arr[i].launches, "ms": arr[i].ms, "flops": arr[i].flops, "bytes": arr[i].bytes}
                for i, n in enumerate(NC_KC_NAMES)}
