// Factories for the native models, shaped like the reference's NeuralCodecs.Create{SNAC,DAC,Encodec}Async (NeuralCodecs.Torch/
// NeuralCodecs.cs:38-80: same parameter lists; the reference class is `public static partial class NeuralCodecs`, so these members
// join it), and the tensor <-> host-array marshalling the *Native models use at their TorchSharp-typed API surface.
// `path` names an NCWB weight blob (tools/convert_checkpoint.py converts HF safetensors / Descript .pth / SNAC torch files).
using System;
using System.IO;
using System.Threading.Tasks;
using NeuralCodecs.Core;
using NeuralCodecs.Core.Configuration;
using NeuralCodecs.Core.Exceptions;
using NeuralCodecs.Core.Loading;
using NeuralCodecs.Torch.Config.DAC;
using NeuralCodecs.Torch.Config.Encodec;
using NeuralCodecs.Torch.Config.SNAC;
using NeuralCodecs.Torch.Models;
using TorchSharp;
using static TorchSharp.torch;

namespace NeuralCodecs.Torch;

public static partial class NeuralCodecs
{
    public static Task<SNACNative> CreateSNACNativeAsync(string path, SNACConfig? config = null, ModelLoadOptions? options = null)   // NeuralCodecs.cs:38-44
    {
        return LoadNative(path, options, device =>
        {
            config ??= new SNACConfig();
            if (device is not null) config.Device = device;
            return new SNACNative(config);
        });
    }

    public static Task<DACNative> CreateDACNativeAsync(string path, DACConfig? config = null, ModelLoadOptions? options = null)      // NeuralCodecs.cs:56-63
    {
        return LoadNative(path, options, device =>
        {
            config ??= new DACConfig();
            if (device is not null) config.Device = device;
            return new DACNative(config);
        });
    }

    public static Task<EncodecNative> CreateEncodecNativeAsync(string path, EncodecConfig? config = null, ModelLoadOptions? options = null)   // NeuralCodecs.cs:74-80
    {
        return LoadNative(path, options, device =>
        {
            config ??= new EncodecConfig();
            if (device is not null) config.Device = device;
            return new EncodecNative(config);
        });
    }

    // TorchModelLoader.LoadLocalModel (TorchModelLoader.cs:360-384): missing file -> LoadException; any other failure is wrapped
    // in a LoadException; the weights are read on a pool thread (TorchModelLoader.cs:488-492).
    private static Task<TModel> LoadNative<TModel>(string path, ModelLoadOptions? options, Func<DeviceConfiguration?, TModel> create)
        where TModel : class, INeuralCodec
    {
        if (string.IsNullOrEmpty(path)) throw new ArgumentException("Model path cannot be empty", nameof(path));
        return Task.Run(() =>
        {
            if (!File.Exists(path)) throw new LoadException($"Model file not found at {path}");
            TModel? model = null;
            try
            {
                model = create(options?.Device);
                model.LoadWeights(path);
                return model;
            }
            catch (Exception ex) when (ex is not LoadException)
            {
                model?.Dispose();
                throw new LoadException($"Failed to load local model: {path}. {ex.Message}", ex);
            }
        });
    }
}

/// <summary>Tensor <-> host array marshalling; only calls the reference itself makes on TorchSharp (DAC.cs:213-223, SNAC.cs:185).</summary>
internal static class NcTensor
{
    public static float[] Floats(Tensor t) => t.cpu().detach().to(torch.float32).contiguous().data<float>().ToArray();
    public static long[] Longs(Tensor t) => t.cpu().detach().to(torch.int64).contiguous().data<long>().ToArray();
    public static Tensor From(float[] a, params long[] shape) => torch.tensor(a, dtype: torch.float32).reshape(shape);
    public static Tensor From(long[] a, params long[] shape) => torch.tensor(a, dtype: torch.int64).reshape(shape);
}
