// DACNative: the managed DAC model (NeuralCodecs.Torch/Models/DAC.cs) over libnc_mi355x.so.
// Every public member of the reference class is here with the reference's signature, exceptions and INeuralCodec contract; TorchSharp
// tensors exist at the API surface only (marshalled through host arrays) -- no TorchSharp operator runs between a member's entry
// and its return.  Checked mechanically against the reference's own .cs files by tests/test_csharp_shim_cpu.py (dotnet is not
// in the build image): config properties by name and type, member signatures, exception constructors, NcMi355x call arity.
using System;
using System.Collections.Generic;
using NeuralCodecs.Core;
using NeuralCodecs.Core.Configuration;
using NeuralCodecs.Torch.Config.DAC;
using NeuralCodecs.Torch.Native;
using TorchSharp;
using static TorchSharp.torch;

namespace NeuralCodecs.Torch.Models;

public sealed unsafe class DACNative : INeuralCodec
{
    private IntPtr _h;
    private readonly DACConfig _config;
    private readonly int _latentDim;
    private readonly int _hopLength;

    public IModelConfig Config => _config;                                          // Models/DAC.cs:38

    public DACNative(DACConfig config)                                              // Models/DAC.cs:51-93
    {
        _config = config ?? throw new ArgumentNullException(nameof(config));
        config.EncoderRates ??= [2, 4, 8, 8];                                       // DAC.cs:57
        config.DecoderRates ??= [8, 8, 4, 2];                                       // DAC.cs:59
        _latentDim = config.LatentDim ?? config.EncoderDim * (1 << config.EncoderRates.Length);   // DAC.cs:63
        _hopLength = 1;
        foreach (int r in config.EncoderRates) _hopLength *= r;                     // DAC.cs:66
        var c = new NcDacConfig
        {
            sample_rate = config.SampleRate, encoder_dim = config.EncoderDim, n_encoder_rates = config.EncoderRates.Length,
            decoder_dim = config.DecoderDim, n_decoder_rates = config.DecoderRates.Length, latent_dim = config.LatentDim ?? 0,
            n_codebooks = config.NumCodebooks, codebook_size = config.CodebookSize, codebook_dim = config.CodebookDim,
        };
        for (int i = 0; i < config.EncoderRates.Length; ++i) c.encoder_rates[i] = config.EncoderRates[i];
        for (int i = 0; i < config.DecoderRates.Length; ++i) c.decoder_rates[i] = config.DecoderRates[i];
        NcMi355x.Check(NcMi355x.nc_dac_create(in c, NcMi355x.DeviceIndex(config.Device), out _h));
    }

    public void LoadWeights(string path)                                            // Models/DAC.cs:345-389 (NCWB blob: tools/convert_checkpoint.py)
    {
        if (string.IsNullOrEmpty(path)) throw new ArgumentException("Weights path cannot be empty", nameof(path));
        NcMi355x.Check(NcMi355x.nc_codec_load_weights(_h, path));                   // NC_ENOTFOUND -> FileNotFoundException (DAC.cs:347-350)
    }

    // ---- host-array cores (what every overload below marshals into) ---------------------------------------------------------------
    /// <summary>audio [B,1,T] -> (zQ [B,latent,T'], codes [B,nQ,T'], latents [B,nQ*codebookDim,T']); Models/DAC.cs:163-181.</summary>
    public (float[] z, long[] codes, float[] latents, long frames, int nQ) EncodeHost(float[] audio, int B, long T, int? nQuantizers = null,
                                                                                         int? sampleRate = null)
    {
        ArgumentNullException.ThrowIfNull(audio);
        long padded, frames;
        NcMi355x.Check(NcMi355x.nc_dac_query(_h, T, &padded, &frames));
        int nq = (nQuantizers is int n && n > 0 && n <= _config.NumCodebooks) ? n : _config.NumCodebooks;
        var z = new float[(long)B * _latentDim * frames];
        var codes = new long[(long)B * nq * frames];
        var lat = new float[(long)B * nq * _config.CodebookDim * frames];
        fixed (float* p = audio, pz = z, pl = lat) fixed (long* pc = codes)         // sample-rate mismatch -> NC_EINVAL -> ArgumentException (DAC.cs:144-149)
            NcMi355x.Check(NcMi355x.nc_dac_encode(_h, p, B, T, sampleRate ?? 0, nQuantizers ?? 0, pc, pz, pl));
        return (z, codes, lat, frames, nq);
    }

    public float[] DecodeHost(float[] qAudio, int B, long frames)                   // Models/DAC.cs:231-234 on host arrays
    {
        ArgumentNullException.ThrowIfNull(qAudio);
        var pcm = new float[(long)B * frames * _hopLength];
        fixed (float* pz = qAudio, pp = pcm)
            NcMi355x.Check(NcMi355x.nc_dac_decode(_h, pz, B, frames, pp));
        return pcm;
    }

    public float[] FromCodesHost(long[] codes, int B, int nQ, long frames)          // Models/DAC.cs:101-106 on host arrays
    {
        ArgumentNullException.ThrowIfNull(codes);
        var z = new float[(long)B * _latentDim * frames];
        fixed (long* pc = codes) fixed (float* pz = z)
            NcMi355x.Check(NcMi355x.nc_dac_from_codes(_h, pc, B, nQ, frames, pz));
        return z;
    }

    // ---- the reference's members, signature for signature ---------------------------------------------------------------------------
    public Tensor FromCodes(Tensor codes)                                           // Models/DAC.cs:101-106: codes [B,nQ,T'] int64
    {
        int B = (int)codes.shape[0], nQ = (int)codes.shape[1];
        long frames = codes.shape[2];
        return NcTensor.From(FromCodesHost(NcTensor.Longs(codes), B, nQ, frames), B, _latentDim, frames);
    }

    public (Tensor z, Tensor codes, Tensor latents, Tensor commitmentLoss, Tensor codebookLoss)
        Encode(Tensor audioData, int? nQuantizers = null, int? sampleRate = null)  // Models/DAC.cs:163-181
    {
        int B = (int)audioData.shape[0];
        long T = audioData.shape[^1];
        var (z, codes, lat, frames, nq) = EncodeHost(NcTensor.Floats(audioData), B, T, nQuantizers, sampleRate);
        return (NcTensor.From(z, B, _latentDim, frames), NcTensor.From(codes, B, nq, frames),
                NcTensor.From(lat, B, (long)nq * _config.CodebookDim, frames),
                torch.zeros(1), torch.zeros(1));                                    // eval mode: both losses are 0 (ResidualVectorQuantizer.cs:143-156)
    }

    public Tensor EncodeAudio(Tensor audioData) => Encode(audioData).z;             // Models/DAC.cs:188-198

    public float[] Encode(float[] audioData)                                        // Models/DAC.cs:205-224: returns the zQ latents (D12)
    {
        ArgumentNullException.ThrowIfNull(audioData);
        return EncodeHost(audioData, 1, audioData.Length).z;
    }

    public Tensor Decode(Tensor qAudio)                                             // Models/DAC.cs:231-234: z [B,latent,T'] -> [B,1,T'*hop]
    {
        int B = (int)qAudio.shape[0];
        long frames = qAudio.shape[2];
        return NcTensor.From(DecodeHost(NcTensor.Floats(qAudio), B, frames), B, 1, frames * _hopLength);
    }

    public float[] Decode(float[] qAudio)                                           // Models/DAC.cs:241-253: reshape(1, latent, -1)
    {
        ArgumentNullException.ThrowIfNull(qAudio);
        return DecodeHost(qAudio, 1, qAudio.Length / _latentDim);
    }

    public Dictionary<string, Tensor> forward(Tensor audioData, int? sampleRate, int? nQuantizers)   // Models/DAC.cs:262-281
    {
        var (z, codes, latents, commitmentLoss, codebookLoss) = Encode(audioData, nQuantizers, sampleRate);
        var audio = Decode(z);
        return new Dictionary<string, Tensor>
        {
            ["audio"] = audio, ["z"] = z, ["codes"] = codes, ["latents"] = latents,
            ["vq/commitment_loss"] = commitmentLoss, ["vq/codebook_loss"] = codebookLoss,
        };
    }

    public Dictionary<string, Tensor> forward(Tensor audioData) => forward(audioData, null, null);   // Models/DAC.cs:288-303

    public float[] forward(float[] audioData)                                       // Models/DAC.cs:310-322
    {
        ArgumentNullException.ThrowIfNull(audioData);
        return Decode(Encode(audioData));
    }

    /// <summary>Dia glue (Models/Dia.cs:973-981, Modules/Dia/AudioUtils.cs:189-199): codes [B,T',n_q] -> PCM [B,1,T'*hop].</summary>
    public float[] DecodeCodeMatrix(long[] codesTq, int B, long frames, int nQ)   // Models/Dia.cs:973-981
    {
        ArgumentNullException.ThrowIfNull(codesTq);
        var pcm = new float[(long)B * frames * _hopLength];
        fixed (long* pc = codesTq) fixed (float* pp = pcm)
            NcMi355x.Check(NcMi355x.nc_dac_decode_code_matrix(_h, pc, B, frames, nQ, pp));
        return pcm;
    }

    /// <summary>Dia glue (Models/Dia.cs:989-1002): audio [B,1,T] -> codes [B,T',n_q].</summary>
    public long[] EncodeCodeMatrix(float[] audio, int B, long T)                   // Models/Dia.cs:989-1002
    {
        ArgumentNullException.ThrowIfNull(audio);
        long padded, frames;
        NcMi355x.Check(NcMi355x.nc_dac_query(_h, T, &padded, &frames));
        var codes = new long[(long)B * frames * _config.NumCodebooks];
        fixed (float* p = audio) fixed (long* pc = codes)
            NcMi355x.Check(NcMi355x.nc_dac_encode_code_matrix(_h, p, B, T, 0, pc));
        return codes;
    }

    public void Dispose()                                                           // Models/DAC.cs:329-338
    {
        if (_h != IntPtr.Zero) { NcMi355x.nc_codec_destroy(_h); _h = IntPtr.Zero; }
        GC.SuppressFinalize(this);
    }

    ~DACNative() { if (_h != IntPtr.Zero) NcMi355x.nc_codec_destroy(_h); }
}
