// EncodecNative: the managed Encodec model (NeuralCodecs.Torch/Models/Encodec.cs) over libnc_mi355x.so -- the reference's public
// members and properties with the reference's signatures and exceptions; frames are the reference's own EncodedFrame record
// (Modules/Encodec/EncodedFrame.cs).  Checked against the reference's .cs files by tests/test_csharp_shim_cpu.py.
using System;
using System.Collections.Generic;
using System.Linq;
using NeuralCodecs.Core;
using NeuralCodecs.Core.Configuration;
using NeuralCodecs.Torch.Config.Encodec;
using NeuralCodecs.Torch.Modules.Encodec;
using NeuralCodecs.Torch.Native;
using TorchSharp;
using static TorchSharp.torch;

namespace NeuralCodecs.Torch.Models;

public sealed unsafe class EncodecNative : INeuralCodec
{
    private static readonly int[] SeanetRatios = { 8, 5, 4, 2 };                    // SEANetEncoder.cs:55: the config's ratios never reach SEANet (D11)

    private IntPtr _h;
    private readonly EncodecConfig _config;
    private readonly float _overlap;
    private readonly float? _segment;
    private readonly List<float> _targetBandwidths;
    private readonly int _numCodebooks;
    private float? _bandwidth;

    public EncodecNative(EncodecConfig config)                                      // Models/Encodec.cs:46-90
    {
        _config = config ?? throw new ArgumentNullException(nameof(config));
        if (config.Bandwidth is null || !((IList<float>)config.TargetBandwidths).Contains(config.Bandwidth.Value))
        {
            throw new ArgumentException(                                            // Encodec.cs:49-54
                $"Invalid bandwidth {config.Bandwidth}. " +
                $"Select one of [{string.Join(", ", config.TargetBandwidths)}]");
        }
        _targetBandwidths = config.TargetBandwidths.ToList();
        _bandwidth = config.Bandwidth;
        int totalRatio = SeanetRatios.Aggregate((a, b) => a * b);                  // encoder.TotalRatio, SEANetEncoder.cs:140
        _numCodebooks = (int)(1000 * config.TargetBandwidths.Max() /
            (Math.Ceiling(config.SampleRate / (double)totalRatio) * 10));           // Encodec.cs:70-71
        SampleRate = config.SampleRate;
        _segment = config.ChunkLengthSeconds;
        Channels = config.Channels;
        Normalize = config.Normalize;
        _overlap = config.Overlap ?? 0;                                             // Encodec.cs:84
        FrameRate = (int)Math.Ceiling(SampleRate / (float)totalRatio);              // Encodec.cs:86
        BitsPerCodebook = (int)Math.Log2(config.CodebookSize);                      // Encodec.cs:87

        var c = new NcEncodecConfig
        {
            sample_rate = config.SampleRate, channels = config.Channels, dimension = config.HiddenSize, n_filters = 32, n_ratios = 4,
            lstm_layers = 2, compress = 2, kernel_size = 7, last_kernel_size = 7, residual_kernel_size = 3,   // SEANetEncoder.cs:37-56 defaults
            time_group_norm = config.NormType == "time_group_norm" ? 1 : 0, causal = config.UseCausalConv ? 1 : 0,
            normalize = config.Normalize ? 1 : 0, segment_length = SegmentLength ?? 0, segment_stride = SegmentStride ?? 0,
            codebook_size = config.CodebookSize, n_codebooks = _numCodebooks, frame_rate = FrameRate, bandwidth = config.Bandwidth.Value,
        };
        for (int i = 0; i < 4; ++i) c.ratios[i] = SeanetRatios[i];
        NcMi355x.Check(NcMi355x.nc_encodec_create(in c, NcMi355x.DeviceIndex(config.Device), out _h));
    }

    // ---- properties (Models/Encodec.cs:145-201) ---------------------------------------------------------------------------------------
    public int BitsPerCodebook { get; }
    public int Channels { get; }
    public IModelConfig Config => _config;
    public float? CurrentBandwidth => _bandwidth;
    public int FrameRate { get; }
    public bool Normalize { get; }
    public int NumCodebooks => _numCodebooks;
    public int SampleRate { get; }
    public int? SegmentLength => _segment.HasValue ? (int)(_segment.Value * SampleRate) : null;
    public int? SegmentStride => SegmentLength.HasValue ?
        Math.Max(1, (int)((1 - _overlap) * SegmentLength.Value)) : null;
    public IReadOnlyList<float> TargetBandwidths => _targetBandwidths;

    public void LoadWeights(string path)                                            // Models/Encodec.cs:348-385
    {
        if (string.IsNullOrEmpty(path)) throw new ArgumentException("Weights path cannot be empty", nameof(path));
        NcMi355x.Check(NcMi355x.nc_codec_load_weights(_h, path));
    }

    public void SetTargetBandwidth(float bandwidth)                                 // Models/Encodec.cs:409-419
    {
        if (!_targetBandwidths.Contains(bandwidth))
        {
            throw new ArgumentException(
                $"This model doesn't support the bandwidth {bandwidth} kbps. " +
                $"Select one of [{string.Join(", ", _targetBandwidths)} kbps]");
        }
        NcMi355x.Check(NcMi355x.nc_encodec_set_bandwidth(_h, bandwidth));
        _bandwidth = bandwidth;
        _config.Bandwidth = bandwidth;
    }

    // ---- host-array cores ------------------------------------------------------------------------------------------------------------
    /// <summary>x [B,C,T] -> per segment (codes [B,n_q,T'_f], scale [B] or null).  Models/Encodec.cs:259-285, EncodeFrame :457-489.</summary>
    public List<(long[] codes, float[]? scale, int nQ, long frames)> EncodeHost(float[] audio, int B, long T)
    {
        ArgumentNullException.ThrowIfNull(audio);
        int nFrames, nQ;
        long decoded;
        NcMi355x.Check(NcMi355x.nc_encodec_query(_h, T, &nFrames, &nQ, null, 0, &decoded));   // count first
        var lens = new long[nFrames];
        fixed (long* pl = lens) NcMi355x.Check(NcMi355x.nc_encodec_query(_h, T, &nFrames, &nQ, pl, nFrames, &decoded));
        long total = lens.Sum();
        var codes = new long[B * nQ * total];
        var scales = new float[nFrames * B];
        fixed (float* p = audio, ps = scales) fixed (long* pc = codes)
            NcMi355x.Check(NcMi355x.nc_encodec_encode(_h, p, B, T, pc, ps, null));
        var frames = new List<(long[], float[]?, int, long)>(nFrames);
        long off = 0;
        for (int f = 0; f < nFrames; ++f)
        {
            long n = (long)B * nQ * lens[f];
            var c = new long[n];
            Array.Copy(codes, off, c, 0, n);
            float[]? sc = Normalize ? scales.AsSpan(f * B, B).ToArray() : null;
            frames.Add((c, sc, nQ, lens[f]));
            off += n;
        }
        return frames;
    }

    /// <summary>frames -> audio [B,C,decoded]: decode + scale + linear overlap-add (Models/Encodec.cs:213-235,436-455).  The frames
    /// alone fix the output, as in the reference: the clip length of their layout comes from nc_encodec_clip_length.</summary>
    public float[] DecodeHost(List<(long[] codes, float[]? scale, int nQ, long frames)> frames, int B, out long decoded)
    {
        if (frames.Count == 0) throw new ArgumentException("No frames provided to decode");              // Encodec.cs:215-218
        if (SegmentLength == null && frames.Count != 1)
            throw new ArgumentException("Expected single frame when no segmentation is used");              // Encodec.cs:222-225
        long T, dec;
        int nFrames, nQq;
        var lens = frames.ConvertAll(f => f.frames).ToArray();
        fixed (long* pl = lens)
        {
            NcMi355x.Check(NcMi355x.nc_encodec_clip_length(_h, frames.Count, pl, &T));   // NC_EINVAL: no clip is cut into frames of these lengths
            NcMi355x.Check(NcMi355x.nc_encodec_query(_h, T, &nFrames, &nQq, null, 0, &dec));
        }
        int nQ = frames[0].nQ;
        for (int f = 0; f < frames.Count; ++f)
            if (frames[f].nQ != nQ || frames[f].codes.Length != (long)B * nQ * lens[f])
                throw new ArgumentException($"Frame {f}: expected [{B}, {nQ}, {lens[f]}] codes");
        var codes = new long[frames.Sum(f => (long)f.codes.Length)];
        var scales = new float[frames.Count * B];
        long off = 0;
        for (int f = 0; f < frames.Count; ++f)
        {
            Array.Copy(frames[f].codes, 0, codes, off, frames[f].codes.Length);
            off += frames[f].codes.Length;
            if (frames[f].scale is float[] s) Array.Copy(s, 0, scales, f * B, B);
        }
        var pcm = new float[(long)B * Channels * dec];
        fixed (long* pc = codes) fixed (float* ps = scales, pp = pcm)
            NcMi355x.Check(NcMi355x.nc_encodec_decode(_h, pc, Normalize ? ps : null, B, T, nQ, pp));
        decoded = dec;
        return pcm;
    }

    // ---- the reference's members, signature for signature ---------------------------------------------------------------------------
    public Tensor Decode(List<EncodedFrame> encodedFrames)                          // Models/Encodec.cs:213-235
    {
        if (encodedFrames.Count == 0)
        {
            throw new ArgumentException("No frames provided to decode");
        }
        var host = new List<(long[], float[]?, int, long)>(encodedFrames.Count);
        int B = 0;
        foreach (var frame in encodedFrames)
        {
            if (frame.Codes?.IsInvalid != false)
                throw new ArgumentException("Invalid frame codes in Encodec Decode");                      // Encodec.cs:438-442
            B = (int)frame.Codes.shape[0];
            host.Add((NcTensor.Longs(frame.Codes), frame.Scale is null ? null : NcTensor.Floats(frame.Scale),
                      (int)frame.Codes.shape[1], frame.Codes.shape[2]));
        }
        var pcm = DecodeHost(host, B, out long decoded);
        return NcTensor.From(pcm, B, Channels, decoded);
    }

    public List<EncodedFrame> Encode(float[] audioData)                             // Models/Encodec.cs:243-257
    {
        ArgumentNullException.ThrowIfNull(audioData);
        return ToFrames(EncodeHost(audioData, 1, audioData.Length / _config.Channels), 1);   // reshape(1, Channels, -1)
    }

    public List<EncodedFrame> Encode(Tensor x)                                      // Models/Encodec.cs:259-285
    {
        ValidateInputTensor(x);
        long channels = x.size(1);
        if (channels is <= 0 or > 2)
        {
            throw new ArgumentException($"Invalid number of channels: {channels}");
        }
        int B = (int)x.size(0);
        return ToFrames(EncodeHost(NcTensor.Floats(x), B, x.size(2)), B);
    }

    public Tensor forward(Tensor x)                                                 // Models/Encodec.cs:292-296
    {
        var frames = Encode(x);
        return Decode(frames).slice(2, 0, x.size(-1), 1);
    }

    private List<EncodedFrame> ToFrames(List<(long[] codes, float[]? scale, int nQ, long frames)> host, int B)
    {
        return host.ConvertAll(f => new EncodedFrame(NcTensor.From(f.codes, B, f.nQ, f.frames),
                                                     f.scale is null ? null : NcTensor.From(f.scale, B, 1)));   // scale.view(-1, 1), Encodec.cs:479
    }

    private void ValidateInputTensor(Tensor x)                                      // Models/Encodec.cs:491-504
    {
        if (x.dim() != 3)
        {
            throw new ArgumentException(
                $"Expected 3D input tensor [B,C,T], got shape [{string.Join(", ", x.shape)}]");
        }
        if (x.shape[1] != _config.Channels)
        {
            throw new ArgumentException(
                $"Expected {_config.Channels} channels, got {x.shape[1]}");
        }
    }

    public void Dispose()
    {
        if (_h != IntPtr.Zero) { NcMi355x.nc_codec_destroy(_h); _h = IntPtr.Zero; }
        GC.SuppressFinalize(this);
    }

    ~EncodecNative() { if (_h != IntPtr.Zero) NcMi355x.nc_codec_destroy(_h); }
}
