// SNACNative: the managed SNAC model (NeuralCodecs.Torch/Models/SNAC.cs) over libnc_mi355x.so -- the reference's public members
// with the reference's signatures and exceptions.  Checked against the reference's .cs files by tests/test_csharp_shim_cpu.py.
using System;
using System.Collections.Generic;
using System.Linq;
using NeuralCodecs.Core;
using NeuralCodecs.Core.Configuration;
using NeuralCodecs.Torch.Config.SNAC;
using NeuralCodecs.Torch.Native;
using TorchSharp;
using static TorchSharp.torch;

namespace NeuralCodecs.Torch.Models;

public sealed unsafe class SNACNative : INeuralCodec
{
    private IntPtr _h;
    private readonly SNACConfig _config;
    private readonly int _hopLength;

    public IModelConfig Config => _config;                                          // Models/SNAC.cs:28

    public SNACNative(SNACConfig config)                                            // Models/SNAC.cs:34-63
    {
        _config = config ?? throw new ArgumentNullException(nameof(config));
        _config.LatentDim = config.LatentDim ?? config.EncoderDim * (1 << config.EncoderRates.Length);   // SNAC.cs:37
        _hopLength = 1;
        foreach (int r in config.EncoderRates) _hopLength *= r;                     // SNAC.cs:38
        var c = new NcSnacConfig
        {
            sample_rate = config.SampleRate, encoder_dim = config.EncoderDim, n_encoder_rates = config.EncoderRates.Length,
            decoder_dim = config.DecoderDim, n_decoder_rates = config.DecoderRates.Length, latent_dim = config.LatentDim ?? 0,
            attn_window_size = config.AttnWindowSize ?? 0, codebook_size = config.CodebookSize, codebook_dim = config.CodebookDim,
            n_vq_strides = config.VQStrides.Length, noise = config.Noise ? 1 : 0, depthwise = config.Depthwise ? 1 : 0,
        };
        for (int i = 0; i < config.EncoderRates.Length; ++i) c.encoder_rates[i] = config.EncoderRates[i];
        for (int i = 0; i < config.DecoderRates.Length; ++i) c.decoder_rates[i] = config.DecoderRates[i];
        for (int i = 0; i < config.VQStrides.Length; ++i) c.vq_strides[i] = config.VQStrides[i];
        NcMi355x.Check(NcMi355x.nc_snac_create(in c, NcMi355x.DeviceIndex(config.Device), out _h));
    }

    public void LoadWeights(string path)                                            // Models/SNAC.cs:200-246
    {
        if (string.IsNullOrEmpty(path)) throw new ArgumentException("Weights path cannot be empty", nameof(path));
        NcMi355x.Check(NcMi355x.nc_codec_load_weights(_h, path));
    }

    // ---- host-array cores ------------------------------------------------------------------------------------------------------------
    /// <summary>audio [B,1,T] -> one long[B * T'/stride_i] per level, coarse first.  pad = true is Preprocess + encode
    /// (Encode(float[]) / forward, SNAC.cs:70-80,129-150); pad = false is Encode(Tensor) exactly as written (SNAC.cs:113-122, D7).</summary>
    public List<long[]> EncodeHost(float[] audio, int B, long T, bool pad, out long frames)
    {
        ArgumentNullException.ThrowIfNull(audio);
        long padded, decoded, fr;
        int nLevels;
        long* widths = stackalloc long[8];
        if (pad) NcMi355x.Check(NcMi355x.nc_snac_query(_h, T, &padded, &fr, &nLevels, widths, &decoded));
        else NcMi355x.Check(NcMi355x.nc_snac_query_tensor(_h, T, &fr, &nLevels, widths));   // NC_EINVAL where the reference's quantizer / LocalMHA throw
        frames = fr;
        long per = 0;
        for (int i = 0; i < nLevels; ++i) per += widths[i];
        var flat = new long[B * per];
        fixed (float* p = audio) fixed (long* pc = flat)
        {
            if (pad) NcMi355x.Check(NcMi355x.nc_snac_encode(_h, p, B, T, pc, null, null));
            else NcMi355x.Check(NcMi355x.nc_snac_encode_tensor(_h, p, B, T, pc, null, null));
        }
        var levels = new List<long[]>(nLevels);
        long off = 0;
        for (int i = 0; i < nLevels; ++i)                                           // the levels of a clip sit side by side: split per level
        {
            var lv = new long[B * widths[i]];
            for (int b = 0; b < B; ++b) Array.Copy(flat, b * per + off, lv, b * widths[i], widths[i]);
            levels.Add(lv);
            off += widths[i];
        }
        return levels;
    }

    /// <summary>codes (one long[B * width_i] per level) -> audio [B,1,decoded]; noise = null draws N(0,1) on the device from seed
    /// (the reference's randn at inference, NoiseBlock.cs:41, D8).</summary>
    public float[] DecodeHost(List<long[]> codes, int B, float[]? noise = null, ulong? seed = null)
    {
        if (codes is null || codes.Count == 0) throw new ArgumentException("Codes list cannot be empty or contain null arrays", nameof(codes));
        long frames = codes[^1].Length / B * _config.VQStrides[^1];                 // level i holds frames / stride_i codes per clip
        long padded, fr, decoded;
        int nLevels;
        long* widths = stackalloc long[8];
        NcMi355x.Check(NcMi355x.nc_snac_query(_h, frames * _hopLength, &padded, &fr, &nLevels, widths, &decoded));
        if (codes.Count != nLevels)
            throw new ArgumentException($"Expected {nLevels} codes, got {codes.Count}");   // Modules/SNAC/ResidualVectorQuantizer.cs:103
        long per = 0;
        for (int i = 0; i < nLevels; ++i) per += widths[i];
        var flat = new long[B * per];
        long off = 0;
        for (int i = 0; i < nLevels; ++i)
        {
            if (codes[i].Length != B * widths[i]) throw new ArgumentException($"Level {i}: expected {B * widths[i]} codes, got {codes[i].Length}");
            for (int b = 0; b < B; ++b) Array.Copy(codes[i], b * widths[i], flat, b * per + off, widths[i]);
            off += widths[i];
        }
        var pcm = new float[B * decoded];
        fixed (long* pc = flat) fixed (float* pn = noise, pp = pcm)
            NcMi355x.Check(NcMi355x.nc_snac_decode(_h, pc, B, frames, pn, seed ?? (ulong)Random.Shared.NextInt64(), pp));
        return pcm;
    }

    // ---- the reference's members, signature for signature ---------------------------------------------------------------------------
    public (Tensor audio, List<Tensor> codes) forward(Tensor audioData)             // Models/SNAC.cs:91-106
    {
        int B = (int)audioData.shape[0];
        long length = audioData.shape[^1];
        var levels = EncodeHost(NcTensor.Floats(audioData), B, length, true, out long frames);
        var pcm = DecodeHost(levels, B);
        long decoded = pcm.Length / B;
        var audioHat = NcTensor.From(pcm, B, 1, decoded).narrow(-1, 0, length);     // SNAC.cs:103
        return (audioHat, levels.ConvertAll(lv => NcTensor.From(lv, B, lv.Length / B)));
    }

    public List<Tensor> Encode(Tensor audioData)                                    // Models/SNAC.cs:113-122 as written: the encoder sees the un-padded tensor (D7)
    {
        int B = (int)audioData.shape[0];
        var levels = EncodeHost(NcTensor.Floats(audioData), B, audioData.shape[^1], false, out _);
        return levels.ConvertAll(lv => NcTensor.From(lv, B, lv.Length / B));
    }

    public List<float[]> Encode(float[] audioData)                                  // Models/SNAC.cs:129-150: codes as float arrays
    {
        ArgumentNullException.ThrowIfNull(audioData);
        var levels = EncodeHost(audioData, 1, audioData.Length, true, out _);
        return levels.ConvertAll(lv => Array.ConvertAll(lv, v => (float)v));       // SNAC.cs:147 .to(float32)
    }

    public Tensor Decode(List<Tensor> codes)                                        // Models/SNAC.cs:157-165
    {
        if (codes is null || codes.Count == 0) throw new ArgumentException("Codes list cannot be empty", nameof(codes));
        int B = (int)codes[0].shape[0];
        var pcm = DecodeHost(codes.ConvertAll(NcTensor.Longs), B);
        return NcTensor.From(pcm, B, 1, pcm.Length / B);
    }

    public float[] Decode(List<float[]> codes)                                      // Models/SNAC.cs:173-192
    {
        ArgumentNullException.ThrowIfNull(codes);
        if (codes.Count == 0 || codes.Any(code => code == null))
            throw new ArgumentException("Codes list cannot be empty or contain null arrays", nameof(codes));
        return DecodeHost(codes.ConvertAll(code => Array.ConvertAll(code, v => (long)v)), 1);   // SNAC.cs:185 tensor(code, int64)
    }

    public float[] ProcessAudio(float[] audioData, int sampleRate)                  // Models/SNAC.cs:255-282 (+ ResampleAudio :284-308)
    {
        if (audioData == null || audioData.Length == 0)
            throw new ArgumentException("Audio data cannot be empty", nameof(audioData));
        long nOut;
        NcMi355x.Check(NcMi355x.nc_snac_process_audio_len(_h, audioData.Length, sampleRate, &nOut));
        var result = new float[nOut];
        fixed (float* p = audioData, po = result)                                   // one upload -> resample -> forward -> one download, inside the engine
            NcMi355x.Check(NcMi355x.nc_snac_process_audio(_h, p, audioData.Length, sampleRate, null, (ulong)Random.Shared.NextInt64(), po));
        return result;
    }

    public void Dispose()
    {
        if (_h != IntPtr.Zero) { NcMi355x.nc_codec_destroy(_h); _h = IntPtr.Zero; }
        GC.SuppressFinalize(this);
    }

    ~SNACNative() { if (_h != IntPtr.Zero) NcMi355x.nc_codec_destroy(_h); }
}
