#!/usr/bin/env python3
"""Generate the C# P/Invoke shim of libnc_mi355x.so from include/nc_mi355x.h.

    python tools/gen_csharp_shim.py            # writes bindings/csharp/*.cs
    python tools/gen_csharp_shim.py --check    # exit 1 when the committed files differ from what the header generates

Outputs (a maintainer of the reference drops them into NeuralCodecs.Torch/Native/ and Models/):
  bindings/csharp/NcMi355x.cs       every enum, struct layout and NC_API export of the header as [DllImport] stubs + the status ->
                                    exception map of SURVEY 8b.  Mechanical: one stub per export, parameter for parameter.
  bindings/csharp/DAC.Native.cs     partial-class bodies of the managed members the reference exposes (INeuralCodec, DAC.Encode /
  bindings/csharp/SNAC.Native.cs    Decode / FromCodes, SNAC.Encode / Decode, Encodec.Encode / Decode / SetTargetBandwidth) written
  bindings/csharp/Encodec.Native.cs against those stubs.  Templates (below), checked mechanically: every NcMi355x.nc_* call they
                                    make must exist in the header with the same number of arguments (tests/test_csharp_shim_cpu.py).
dotnet is not available in the build image, so the files are generated and cross-checked, not compiled.
"""
import argparse
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "nc_mi355x.h")
OUT = os.path.join(ROOT, "bindings", "csharp")

SCALARS = {"int": "int", "int32_t": "int", "int64_t": "long", "uint64_t": "ulong", "size_t": "nuint", "float": "float", "double": "double",
           "int16_t": "short", "uint8_t": "byte", "nc_status": "NcStatus"}
OPAQUE = {"nc_codec", "nc_group"}


def strip_comments(src):
    return re.sub(r"/\*.*?\*/", " ", src, flags=re.S)


def pascal(name):
    return "".join(p[:1].upper() + p[1:] for p in name.split("_") if p)


def parse_header(path=HDR):
    raw = open(path).read()
    nocom = strip_comments(raw)
    src = "\n".join(l for l in nocom.splitlines() if not l.lstrip().startswith("#"))   # (the NC_API define itself is not an export)
    enums, structs, funcs, defines = [], [], [], []
    for m in re.finditer(r"typedef\s+enum\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        items = []
        for it in m.group(1).split(","):
            it = it.strip()
            if not it:
                continue
            k, _, v = it.partition("=")
            items.append((k.strip(), v.strip()))
        enums.append((m.group(2), items))
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = []
        for decl in m.group(1).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            ty, rest = decl.split(" ", 1)
            for nm in rest.split(","):
                nm = nm.strip()
                am = re.match(r"(\w+)\[(\d+)\]$", nm)
                fields.append((ty, am.group(1), int(am.group(2))) if am else (ty, nm, 0))
        structs.append((m.group(2), fields))
    for m in re.finditer(r"NC_API\s+(.*?)\b(nc_\w+)\s*\((.*?)\)\s*;", src, flags=re.S):
        ret = " ".join(m.group(1).split())
        params = []
        body = " ".join(m.group(3).split())
        if body and body != "void":
            for p in body.split(","):
                p = p.strip()
                pm = re.match(r"(.*?)(\w+)$", p)
                params.append((" ".join(pm.group(1).split()), pm.group(2)))
        funcs.append((ret, m.group(2), params))
    for m in re.finditer(r"#define\s+(NC_[A-Z_]+)\s+(\d+)", nocom):
        defines.append((m.group(1), int(m.group(2))))
    return enums, structs, funcs, defines


def cs_struct_name(c):
    return pascal(c)            # nc_dac_config -> NcDacConfig


def cs_param(cty, name):
    """C parameter type -> (C# type text, note).  Pointers to data stay raw pointers (the callers pin arrays with `fixed`)."""
    t = cty.replace("const ", "").strip()
    const = cty.strip().startswith("const")
    if t == "char*":
        return "[MarshalAs(UnmanagedType.LPUTF8Str)] string"
    m = re.match(r"(\w+)\s*\*\s*const\s*\*$", cty.replace("const nc_", "nc_", 1)) or re.match(r"(\w+)\s*\*\s*const\s*\*$", t)
    if m and m.group(1) in OPAQUE:
        return "IntPtr[]"
    if t.endswith("**"):
        base = t[:-2].strip()
        if base in OPAQUE:
            return "out IntPtr"
    if t.endswith("*"):
        base = t[:-1].strip()
        if base in OPAQUE:
            return "IntPtr"
        if base == "void":
            return "void*"
        if base in SCALARS:
            return SCALARS[base] + "*"
        if base.startswith("nc_"):
            return ("in " if const else "") + cs_struct_name(base) + ("" if const else "*")
    if t in SCALARS:
        return SCALARS[t]
    raise SystemExit(f"gen_csharp_shim: no C# mapping for parameter type '{cty}' ({name})")


def cs_ret(cty):
    t = cty.replace("const ", "").strip()
    if t == "char*":
        return "IntPtr"
    if t in SCALARS:
        return SCALARS[t]
    raise SystemExit(f"gen_csharp_shim: no C# mapping for return type '{cty}'")


KEYWORDS = {"out", "in", "ref", "params", "string", "object", "base", "event", "fixed", "lock", "checked"}


def cs_ident(n):
    return "@" + n if n in KEYWORDS else n


def gen_native(enums, structs, funcs, defines):
    o = []
    o.append("// <auto-generated> by tools/gen_csharp_shim.py from include/nc_mi355x.h -- do not edit; regenerate instead. </auto-generated>")
    o.append("// P/Invoke surface of libnc_mi355x.so (the MI355X-native Encode / RVQ / Decode engine): one stub per NC_API export,")
    o.append("// parameter for parameter; struct layouts are LayoutKind.Sequential mirrors of the C structs (all fields 4- or 8-byte scalars).")
    o.append("using System;")
    o.append("using System.Runtime.InteropServices;")
    o.append("")
    o.append("namespace NeuralCodecs.Torch.Native;")
    o.append("")
    for name, items in enums:
        o.append(f"internal enum {pascal(name)}")
        o.append("{")
        for k, v in items:
            o.append(f"    {k}{' = ' + v if v else ''},")
        o.append("}")
        o.append("")
    for name, fields in structs:
        o.append("[StructLayout(LayoutKind.Sequential)]")
        o.append(f"internal unsafe struct {cs_struct_name(name)}   // {name}")
        o.append("{")
        for ty, fn, n in fields:
            cs = SCALARS[ty]
            o.append(f"    public fixed {cs} {fn}[{n}];" if n else f"    public {cs} {fn};")
        o.append("}")
        o.append("")
    o.append("internal static unsafe class NcMi355x")
    o.append("{")
    o.append('    private const string Lib = "nc_mi355x";   // libnc_mi355x.so on the loader path')
    for k, v in defines:
        o.append(f"    public const int {k} = {v};")
    o.append("")
    for ret, name, params in funcs:
        ps = ", ".join(f"{cs_param(t, n)} {cs_ident(n)}" for t, n in params)
        o.append(f"    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern {cs_ret(ret)} {name}({ps});")
    o.append("")
    o.append("    public static string LastError() => Marshal.PtrToStringUTF8(nc_last_error()) ?? string.Empty;")
    o.append("")
    o.append("    // status -> the exception types the reference throws today (SURVEY 8b)")
    o.append("    public static void Check(NcStatus s)")
    o.append("    {")
    o.append("        if (s == NcStatus.NC_OK) return;")
    o.append("        string msg = LastError();")
    o.append("        if (msg.Length == 0) msg = s.ToString();")
    o.append("        throw s switch")
    o.append("        {")
    o.append("            NcStatus.NC_EINVAL => new ArgumentException(msg),                      // Models/DAC.cs:146, Models/Encodec.cs:493-503")
    o.append("            NcStatus.NC_ENOTFOUND => new System.IO.FileNotFoundException(msg),     // Models/DAC.cs:347-350")
    o.append("            NcStatus.NC_ESTATE => new InvalidOperationException(msg),              // Models/DAC.cs:385-388")
    o.append("            NcStatus.NC_ENOMEM => new OutOfMemoryException(msg),")
    o.append("            NcStatus.NC_EUNSUPPORTED => new NotSupportedException(msg),")
    o.append("            _ => new NeuralCodecs.Core.Exceptions.CodecException(msg),             // NC_EDEVICE")
    o.append("        };")
    o.append("    }")
    o.append("}")
    return "\n".join(o) + "\n"


# ---- partial-class bodies (templates; every NcMi355x.nc_* call is checked against the header) -----------------------------------------
DAC_CS = r'''// <auto-generated> by tools/gen_csharp_shim.py -- template section; regenerate, do not edit. </auto-generated>
// Bodies of the managed DAC members (NeuralCodecs.Torch/Models/DAC.cs) over libnc_mi355x.so: signatures, exceptions and INeuralCodec
// stay as in the reference; no TorchSharp operator runs between a member's entry and its return.
using System;
using System.Collections.Generic;
using NeuralCodecs.Core;
using NeuralCodecs.Core.Configuration;
using NeuralCodecs.Torch.Config.DAC;
using NeuralCodecs.Torch.Native;

namespace NeuralCodecs.Torch.Models;

public sealed unsafe partial class DACNative : INeuralCodec
{
    private IntPtr _h;
    private readonly DACConfig _config;
    public IModelConfig Config => _config;                                  // INeuralCodec.cs:13

    public DACNative(DACConfig config, int deviceIndex = 0)                 // Models/DAC.cs:51-93
    {
        _config = config ?? throw new ArgumentNullException(nameof(config));
        var c = new NcDacConfig
        {
            sample_rate = config.SamplingRate, encoder_dim = config.EncoderDim, n_encoder_rates = config.EncoderRates.Length,
            decoder_dim = config.DecoderDim, n_decoder_rates = config.DecoderRates.Length, latent_dim = config.LatentDim ?? 0,
            n_codebooks = config.NumCodebooks, codebook_size = config.CodebookSize, codebook_dim = config.CodebookDim,
        };
        for (int i = 0; i < config.EncoderRates.Length; ++i) c.encoder_rates[i] = config.EncoderRates[i];
        for (int i = 0; i < config.DecoderRates.Length; ++i) c.decoder_rates[i] = config.DecoderRates[i];
        NcMi355x.Check(NcMi355x.nc_dac_create(in c, deviceIndex, out _h));
    }

    public void LoadWeights(string path)                                    // Models/DAC.cs:345-389 (NCWB blob: tools/convert_checkpoint.py)
    {
        if (string.IsNullOrEmpty(path)) throw new ArgumentException("path");
        NcMi355x.Check(NcMi355x.nc_codec_load_weights(_h, path));
    }

    /// <summary>Encode(Tensor audio [B,1,T], nQuantizers, sampleRate) -> (z, codes, latents): Models/DAC.cs:163-181 on host arrays.</summary>
    public (float[] z, long[] codes, float[] latents, long frames, int nQ) Encode(float[] audio, int B, long T, int? nQuantizers = null, int? sampleRate = null)
    {
        ArgumentNullException.ThrowIfNull(audio);
        long padded, frames;
        NcMi355x.Check(NcMi355x.nc_dac_query(_h, T, &padded, &frames));
        int nq = (nQuantizers is int n && n > 0 && n <= _config.NumCodebooks) ? n : _config.NumCodebooks;
        int latent = _config.LatentDim ?? _config.EncoderDim << _config.EncoderRates.Length;
        var z = new float[(long)B * latent * frames];
        var codes = new long[(long)B * nq * frames];
        var lat = new float[(long)B * nq * _config.CodebookDim * frames];
        fixed (float* p = audio, pz = z, pl = lat) fixed (long* pc = codes)
            NcMi355x.Check(NcMi355x.nc_dac_encode(_h, p, B, T, sampleRate ?? 0, nQuantizers ?? 0, pc, pz, pl));
        return (z, codes, lat, frames, nq);
    }

    public float[] Encode(float[] audioData)                               // Models/DAC.cs:205-224: returns the zQ latents (D12)
    {
        ArgumentNullException.ThrowIfNull(audioData);
        return Encode(audioData, 1, audioData.Length).z;
    }

    public float[] Decode(float[] qAudio, int B, long frames)              // Models/DAC.cs:231-234 on host arrays
    {
        ArgumentNullException.ThrowIfNull(qAudio);
        long padded, fr;
        NcMi355x.Check(NcMi355x.nc_dac_query(_h, 1, &padded, &fr));
        long hop = padded;                                                  // T = 1 pads to one hop
        var pcm = new float[(long)B * frames * hop];
        fixed (float* pz = qAudio, pp = pcm)
            NcMi355x.Check(NcMi355x.nc_dac_decode(_h, pz, B, frames, pp));
        return pcm;
    }

    public float[] Decode(float[] qAudio)                                  // Models/DAC.cs:241-253: reshape(1, latent, -1)
    {
        ArgumentNullException.ThrowIfNull(qAudio);
        int latent = _config.LatentDim ?? _config.EncoderDim << _config.EncoderRates.Length;
        return Decode(qAudio, 1, qAudio.Length / latent);
    }

    public float[] FromCodes(long[] codes, int B, int nQ, long frames)     // Models/DAC.cs:101-106
    {
        ArgumentNullException.ThrowIfNull(codes);
        int latent = _config.LatentDim ?? _config.EncoderDim << _config.EncoderRates.Length;
        var z = new float[(long)B * latent * frames];
        fixed (long* pc = codes) fixed (float* pz = z)
            NcMi355x.Check(NcMi355x.nc_dac_from_codes(_h, pc, B, nQ, frames, pz));
        return z;
    }

    public float[] forward(float[] audioData) => Decode(Encode(audioData)); // Models/DAC.cs:310-322

    /// <summary>Dia glue (Models/Dia.cs:973-981, Modules/Dia/AudioUtils.cs:189-199): codes [B,T',n_q] -> PCM.</summary>
    public float[] DecodeCodeMatrix(long[] codesTq, int B, long frames, int nQ)
    {
        long padded, fr;
        NcMi355x.Check(NcMi355x.nc_dac_query(_h, 1, &padded, &fr));
        var pcm = new float[(long)B * frames * padded];
        fixed (long* pc = codesTq) fixed (float* pp = pcm)
            NcMi355x.Check(NcMi355x.nc_dac_decode_code_matrix(_h, pc, B, frames, nQ, pp));
        return pcm;
    }

    public void Dispose()                                                   // Models/DAC.cs:329-338
    {
        if (_h != IntPtr.Zero) { NcMi355x.nc_codec_destroy(_h); _h = IntPtr.Zero; }
        GC.SuppressFinalize(this);
    }
}
'''

SNAC_CS = r'''// <auto-generated> by tools/gen_csharp_shim.py -- template section; regenerate, do not edit. </auto-generated>
// Bodies of the managed SNAC members (NeuralCodecs.Torch/Models/SNAC.cs) over libnc_mi355x.so.
using System;
using System.Collections.Generic;
using NeuralCodecs.Core;
using NeuralCodecs.Core.Configuration;
using NeuralCodecs.Torch.Config.SNAC;
using NeuralCodecs.Torch.Native;

namespace NeuralCodecs.Torch.Models;

public sealed unsafe partial class SNACNative : INeuralCodec
{
    private IntPtr _h;
    private readonly SNACConfig _config;
    public IModelConfig Config => _config;

    public SNACNative(SNACConfig config, int deviceIndex = 0)               // Models/SNAC.cs:34-63
    {
        _config = config ?? throw new ArgumentNullException(nameof(config));
        var c = new NcSnacConfig
        {
            sample_rate = config.SamplingRate, encoder_dim = config.EncoderDim, n_encoder_rates = config.EncoderRates.Length,
            decoder_dim = config.DecoderDim, n_decoder_rates = config.DecoderRates.Length, latent_dim = config.LatentDim ?? 0,
            attn_window_size = config.AttnWindowSize ?? 0, codebook_size = config.CodebookSize, codebook_dim = config.CodebookDim,
            n_vq_strides = config.VQStrides.Length, noise = config.Noise ? 1 : 0, depthwise = config.Depthwise ? 1 : 0,
        };
        for (int i = 0; i < config.EncoderRates.Length; ++i) c.encoder_rates[i] = config.EncoderRates[i];
        for (int i = 0; i < config.DecoderRates.Length; ++i) c.decoder_rates[i] = config.DecoderRates[i];
        for (int i = 0; i < config.VQStrides.Length; ++i) c.vq_strides[i] = config.VQStrides[i];
        NcMi355x.Check(NcMi355x.nc_snac_create(in c, deviceIndex, out _h));
    }

    public void LoadWeights(string path)                                    // Models/SNAC.cs:200-231
    {
        if (string.IsNullOrEmpty(path)) throw new ArgumentException("path");
        NcMi355x.Check(NcMi355x.nc_codec_load_weights(_h, path));
    }

    /// <summary>SNAC.Encode(float[]) (Models/SNAC.cs:129-150): Preprocess pads; one long[] per level, coarse first.</summary>
    public List<long[]> Encode(float[] audioData, int B = 1)
    {
        ArgumentNullException.ThrowIfNull(audioData);
        long T = audioData.Length / B, padded, frames, decoded;
        int nLevels;
        long* widths = stackalloc long[8];
        NcMi355x.Check(NcMi355x.nc_snac_query(_h, T, &padded, &frames, &nLevels, widths, &decoded));
        long per = 0;
        for (int i = 0; i < nLevels; ++i) per += widths[i];
        var flat = new long[B * per];
        fixed (float* p = audioData) fixed (long* pc = flat)
            NcMi355x.Check(NcMi355x.nc_snac_encode(_h, p, B, T, pc, null, null));
        var levels = new List<long[]>(nLevels);
        long off = 0;
        for (int i = 0; i < nLevels; ++i)                                   // the levels of a clip sit side by side: split per level
        {
            var lv = new long[B * widths[i]];
            for (int b = 0; b < B; ++b) Array.Copy(flat, b * per + off, lv, b * widths[i], widths[i]);
            levels.Add(lv);
            off += widths[i];
        }
        return levels;
    }

    /// <summary>SNAC.Encode(Tensor) exactly as written (Models/SNAC.cs:113-122, deviation D7: no padding).</summary>
    public List<long[]> EncodeTensor(float[] audioData, int B = 1)
    {
        ArgumentNullException.ThrowIfNull(audioData);
        long T = audioData.Length / B, frames;
        int nLevels;
        long* widths = stackalloc long[8];
        NcMi355x.Check(NcMi355x.nc_snac_query_tensor(_h, T, &frames, &nLevels, widths));
        long per = 0;
        for (int i = 0; i < nLevels; ++i) per += widths[i];
        var flat = new long[B * per];
        fixed (float* p = audioData) fixed (long* pc = flat)
            NcMi355x.Check(NcMi355x.nc_snac_encode_tensor(_h, p, B, T, pc, null, null));
        var levels = new List<long[]>(nLevels);
        long off = 0;
        for (int i = 0; i < nLevels; ++i)
        {
            var lv = new long[B * widths[i]];
            for (int b = 0; b < B; ++b) Array.Copy(flat, b * per + off, lv, b * widths[i], widths[i]);
            levels.Add(lv);
            off += widths[i];
        }
        return levels;
    }

    /// <summary>SNAC.Decode(List codes) (Models/SNAC.cs:157-192); noise = null draws N(0,1) on the device (the reference's randn, D8).</summary>
    public float[] Decode(List<long[]> codes, int B = 1, float[]? noise = null, ulong? seed = null)
    {
        if (codes is null || codes.Count == 0) throw new ArgumentException("codes");
        long frames = codes[^1].Length / B;                                 // the finest level has one code per frame
        long padded, fr, decoded;
        int nLevels;
        long* widths = stackalloc long[8];
        NcMi355x.Check(NcMi355x.nc_snac_query(_h, frames * (long)HopLength, &padded, &fr, &nLevels, widths, &decoded));
        if (codes.Count != nLevels) throw new ArgumentException($"Expected {nLevels} code levels, got {codes.Count}");   // SNAC/ResidualVectorQuantizer.cs:103
        long per = 0;
        for (int i = 0; i < nLevels; ++i) per += widths[i];
        var flat = new long[B * per];
        long off = 0;
        for (int i = 0; i < nLevels; ++i)
        {
            for (int b = 0; b < B; ++b) Array.Copy(codes[i], b * widths[i], flat, b * per + off, widths[i]);
            off += widths[i];
        }
        var pcm = new float[B * decoded];
        fixed (long* pc = flat) fixed (float* pn = noise, pp = pcm)
            NcMi355x.Check(NcMi355x.nc_snac_decode(_h, pc, B, frames, pn, seed ?? (ulong)Random.Shared.NextInt64(), pp));
        return pcm;
    }

    private int HopLength { get { int h = 1; foreach (int r in _config.EncoderRates) h *= r; return h; } }

    public void Dispose()
    {
        if (_h != IntPtr.Zero) { NcMi355x.nc_codec_destroy(_h); _h = IntPtr.Zero; }
        GC.SuppressFinalize(this);
    }
}
'''

ENCODEC_CS = r'''// <auto-generated> by tools/gen_csharp_shim.py -- template section; regenerate, do not edit. </auto-generated>
// Bodies of the managed Encodec members (NeuralCodecs.Torch/Models/Encodec.cs) over libnc_mi355x.so.
using System;
using System.Collections.Generic;
using System.Linq;
using NeuralCodecs.Core;
using NeuralCodecs.Core.Configuration;
using NeuralCodecs.Torch.Config.Encodec;
using NeuralCodecs.Torch.Native;

namespace NeuralCodecs.Torch.Models;

/// <summary>EncodedFrame (Modules/Encodec/EncodedFrame.cs) on host arrays: Codes [B,n_q,T'_f] int64, Scale [B] or null.</summary>
public sealed record EncodedFrameNative(long[] Codes, float[]? Scale, int NQ, long Frames);

public sealed unsafe partial class EncodecNative : INeuralCodec
{
    private IntPtr _h;
    private readonly EncodecConfig _config;
    private float _bandwidth;
    public IModelConfig Config => _config;

    public EncodecNative(EncodecConfig config, int deviceIndex = 0)         // Models/Encodec.cs:46-90 (D11: SEANet hard defaults)
    {
        _config = config ?? throw new ArgumentNullException(nameof(config));
        int[] ratios = { 8, 5, 4, 2 };
        int hop = ratios.Aggregate(1, (a, b) => a * b);
        int frameRate = (int)Math.Ceiling(config.SamplingRate / (double)hop);                        // Encodec.cs:83
        float? seg = config.ChunkLengthSeconds;
        int segLen = seg.HasValue ? (int)(seg.Value * config.SamplingRate) : 0;                       // Encodec.cs:190
        int segStride = seg.HasValue ? Math.Max(1, (int)((1 - config.Overlap) * segLen)) : 0;         // Encodec.cs:196
        _bandwidth = config.TargetBandwidths.Max();
        var c = new NcEncodecConfig
        {
            sample_rate = config.SamplingRate, channels = config.AudioChannels, dimension = config.HiddenSize, n_filters = 32, n_ratios = 4,
            lstm_layers = 2, compress = 2, kernel_size = 7, last_kernel_size = 7, residual_kernel_size = 3,
            time_group_norm = config.NormType == "time_group_norm" ? 1 : 0, causal = config.UseCausalConv ? 1 : 0,
            normalize = config.Normalize ? 1 : 0, segment_length = segLen, segment_stride = segStride, codebook_size = config.CodebookSize,
            n_codebooks = (int)(1000 * config.TargetBandwidths.Max() / (frameRate * 10)),              // Encodec.cs:70-71
            frame_rate = frameRate, bandwidth = _bandwidth,
        };
        for (int i = 0; i < 4; ++i) c.ratios[i] = ratios[i];
        NcMi355x.Check(NcMi355x.nc_encodec_create(in c, deviceIndex, out _h));
    }

    public void LoadWeights(string path)                                    // Models/Encodec.cs:348-385
    {
        if (string.IsNullOrEmpty(path)) throw new ArgumentException("path");
        NcMi355x.Check(NcMi355x.nc_codec_load_weights(_h, path));
    }

    public void SetTargetBandwidth(float bandwidth)                         // Models/Encodec.cs:409-419
    {
        if (!_config.TargetBandwidths.Contains(bandwidth))
            throw new ArgumentException($"This model doesn't support the bandwidth {bandwidth}.");
        NcMi355x.Check(NcMi355x.nc_encodec_set_bandwidth(_h, bandwidth));
        _bandwidth = bandwidth;
    }

    /// <summary>Encodec.Encode(Tensor x [B,C,T]) (Models/Encodec.cs:259-285): one EncodedFrame per segment.</summary>
    public List<EncodedFrameNative> Encode(float[] audio, int B, long T)
    {
        ArgumentNullException.ThrowIfNull(audio);                                                     // Encodec.cs:245
        int nFrames, nQ;
        long decoded;
        long* lens = stackalloc long[4096];
        NcMi355x.Check(NcMi355x.nc_encodec_query(_h, T, &nFrames, &nQ, lens, 4096, &decoded));
        long total = 0;
        for (int f = 0; f < nFrames; ++f) total += lens[f];
        var codes = new long[B * nQ * total];
        var scales = new float[nFrames * B];
        fixed (float* p = audio, ps = scales) fixed (long* pc = codes)
            NcMi355x.Check(NcMi355x.nc_encodec_encode(_h, p, B, T, pc, ps, null));
        var frames = new List<EncodedFrameNative>(nFrames);
        long off = 0;
        for (int f = 0; f < nFrames; ++f)
        {
            long n = (long)B * nQ * lens[f];
            var c = new long[n];
            Array.Copy(codes, off, c, 0, n);
            float[]? sc = _config.Normalize ? scales.AsSpan(f * B, B).ToArray() : null;
            frames.Add(new EncodedFrameNative(c, sc, nQ, lens[f]));
            off += n;
        }
        return frames;
    }

    public List<EncodedFrameNative> Encode(float[] audioData) => Encode(audioData, 1, audioData.Length / _config.AudioChannels);   // Encodec.cs:243-257

    /// <summary>Encodec.Decode(List of EncodedFrame) (Models/Encodec.cs:213-235): decode + linear overlap-add; T = the encoded clip length.</summary>
    public float[] Decode(List<EncodedFrameNative> frames, int B, long T)
    {
        if (frames is null || frames.Count == 0) throw new ArgumentException("No frames provided to decode");        // Encodec.cs:215-218
        int nFrames, nQq;
        long decoded;
        long* lens = stackalloc long[4096];
        NcMi355x.Check(NcMi355x.nc_encodec_query(_h, T, &nFrames, &nQq, lens, 4096, &decoded));
        if (frames.Count != nFrames) throw new ArgumentException($"Expected {nFrames} frames for clips of {T} samples, got {frames.Count}");
        int nQ = frames[0].NQ;
        var codes = new long[frames.Sum(f => (long)f.Codes.Length)];
        var scales = new float[nFrames * B];
        long off = 0;
        for (int f = 0; f < nFrames; ++f)
        {
            Array.Copy(frames[f].Codes, 0, codes, off, frames[f].Codes.Length);
            off += frames[f].Codes.Length;
            if (frames[f].Scale is float[] s) Array.Copy(s, 0, scales, f * B, B);
        }
        var pcm = new float[(long)B * _config.AudioChannels * decoded];
        fixed (long* pc = codes) fixed (float* ps = scales, pp = pcm)
            NcMi355x.Check(NcMi355x.nc_encodec_decode(_h, pc, _config.Normalize ? ps : null, B, T, nQ, pp));
        return pcm;
    }

    public void Dispose()
    {
        if (_h != IntPtr.Zero) { NcMi355x.nc_codec_destroy(_h); _h = IntPtr.Zero; }
        GC.SuppressFinalize(this);
    }
}
'''

TEMPLATES = {"DAC.Native.cs": DAC_CS, "SNAC.Native.cs": SNAC_CS, "Encodec.Native.cs": ENCODEC_CS}


def split_args(s):
    """top-level comma split of a call's argument text"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{<":
            depth += 1
        elif ch in ")]}>":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def check_templates(funcs):
    """every NcMi355x.nc_* call of the partial-class templates names a header export with the same number of arguments"""
    sig = {name: params for _, name, params in funcs}
    errors = []
    used = set()
    for fname, text in TEMPLATES.items():
        for m in re.finditer(r"NcMi355x\.(nc_\w+)\s*\(", text):
            name = m.group(1)
            i, depth = m.end(), 1
            while depth and i < len(text):
                depth += text[i] in "([{"
                depth -= text[i] in ")]}"
                i += 1
            args = split_args(text[m.end(): i - 1])
            used.add(name)
            if name not in sig:
                errors.append(f"{fname}: {name} is not declared in include/nc_mi355x.h")
            elif len(args) != len(sig[name]):
                errors.append(f"{fname}: {name} called with {len(args)} arguments, the header declares {len(sig[name])}")
    return errors, used


def generate():
    enums, structs, funcs, defines = parse_header()
    errs, _ = check_templates(funcs)
    if errs:
        raise SystemExit("gen_csharp_shim: " + "; ".join(errs))
    files = {"NcMi355x.cs": gen_native(enums, structs, funcs, defines)}
    files.update(TEMPLATES)
    return files


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    files = generate()
    bad = []
    for name, text in files.items():
        p = os.path.join(OUT, name)
        if a.check:
            if not os.path.exists(p) or open(p).read() != text:
                bad.append(name)
        else:
            os.makedirs(OUT, exist_ok=True)
            open(p, "w").write(text)
    if a.check and bad:
        print("out of date:", ", ".join(bad))
        sys.exit(1)
    print("ok:", ", ".join(files))
