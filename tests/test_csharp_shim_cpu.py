"""CPU suite: the C# binding (bindings/csharp/) held to BOTH of its sides without a C# compiler (dotnet is not in the image).

Engine side: NcMi355x.cs is GENERATED from include/nc_mi355x.h (one [DllImport] stub per NC_API export with the header's parameter
count, struct layouts equal to the ctypes mirrors the parity tests drive the library with), and every NcMi355x.nc_* call of the
hand-written model classes names a header export with that many arguments.

Reference side (build container only: these tests SKIP where /root/reference is absent, e.g. on the GPU box): the model classes
DACNative / SNACNative / EncodecNative and the Create*NativeAsync factories are compared with the reference's own sources --
  * every `config.X` / `_config.X` they read is a public property of the reference's Config/{DAC,SNAC,Encodec}/*Config.cs with the
    type the code assumes, and nullable ones are only used in nullable-safe forms;
  * every public method and property of Models/{DAC,SNAC,Encodec}.cs (SURVEY 8b's preserved surface) exists with the same return
    type, name and parameter types, and has a body; the constructor takes the reference's config type; INeuralCodec is implemented;
  * the factories have the parameter lists of NeuralCodecs.cs:38-80;
  * every reference exception type they construct has a constructor of that arity; every `using NeuralCodecs.*` names a namespace
    the reference declares; the EncodedFrame record is used with its real members;
  * the Encodec constructor keeps the reference's bandwidth rule (Models/Encodec.cs:48-56) and `Overlap ?? 0` (:84)."""
import ctypes as C
import glob
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import csharp_parse as P  # noqa: E402
import gen_csharp_shim as gen  # noqa: E402

REF = "/root/reference"
REF_T = os.path.join(REF, "NeuralCodecs.Torch")
REF_C = os.path.join(REF, "NeuralCodecs.Core")
needs_reference = pytest.mark.skipif(not os.path.isdir(REF_T), reason="the reference's sources exist in the build container only")

MODELS = {  # native class -> (its file, reference class, reference file, reference config class, config file)
    "DACNative": ("DAC.Native.cs", "DAC", "Models/DAC.cs", "DACConfig", "Config/DAC/DACConfig.cs"),
    "SNACNative": ("SNAC.Native.cs", "SNAC", "Models/SNAC.cs", "SNACConfig", "Config/SNAC/SNACConfig.cs"),
    "EncodecNative": ("Encodec.Native.cs", "Encodec", "Models/Encodec.cs", "EncodecConfig", "Config/Encodec/EncodecConfig.cs"),
}
# Reference members that are NOT on the Encode/Decode path and are deliberately not mirrored (SURVEY 8b lists the preserved surface):
NOT_MIRRORED = {
    "DAC": {"props": {"Quantizer"}, "methods": set()},                      # exposes the TorchSharp module object (DAC.cs:39)
    "SNAC": {"props": set(), "methods": set()},
    "Encodec": {"props": {"Device", "Encoder"}, "methods": {"GetLanguageModel"}},   # torch.Device / SEANetEncoder module objects; the LM is SURVEY 2.1 out of scope
}


def _read(path):
    return open(path, encoding="utf-8-sig", errors="replace").read()            # one reference file holds a Latin-1 byte


def _native(fname):
    return _read(os.path.join(gen.OUT, fname))


# ---------------------------------------------------------------------------------------------------------------- engine side
def test_committed_stub_file_is_what_the_header_generates():
    files = gen.generate()
    assert set(files) == {"NcMi355x.cs"}
    for name, text in files.items():
        p = os.path.join(gen.OUT, name)
        assert os.path.exists(p), f"bindings/csharp/{name} missing: run tools/gen_csharp_shim.py"
        assert open(p).read() == text, f"bindings/csharp/{name} is out of date: run tools/gen_csharp_shim.py"
    for f in gen.MODEL_FILES:
        assert os.path.exists(os.path.join(gen.OUT, f))


def test_one_stub_per_export_with_the_headers_parameter_count():
    _, _, funcs, _ = gen.parse_header()
    text = open(os.path.join(gen.OUT, "NcMi355x.cs")).read()
    stubs = {m.group(1): m.group(2) for m in re.finditer(r"public static extern \S+ (nc_\w+)\((.*?)\);", text)}
    assert len(funcs) >= 70 and set(stubs) == {name for _, name, _ in funcs}
    for _, name, params in funcs:
        n = len(gen.split_args(stubs[name]))
        assert n == len(params), f"{name}: stub has {n} parameters, header {len(params)}"
    from neuralcodecs_amd import _lib
    assert set(stubs) == {s[0] for s in _lib.SYMBOLS}          # the same surface the Python mirror binds


def test_struct_layouts_match_the_ctypes_mirrors():
    from neuralcodecs_amd import _lib
    _, structs, _, _ = gen.parse_header()
    sizes = {"int32_t": 4, "int64_t": 8, "float": 4, "double": 8}
    mirrors = {"nc_dac_config": _lib.NcDacConfig, "nc_snac_config": _lib.NcSnacConfig, "nc_encodec_config": _lib.NcEncodecConfig,
               "nc_conv_desc": _lib.NcConvDesc, "nc_profile_entry": _lib.NcProfileEntry}
    assert {n for n, _ in structs} == set(mirrors)
    for name, fields in structs:
        off, align = 0, 1
        for ty, _, n in fields:                                # LayoutKind.Sequential == natural C layout for these scalar fields
            sz = sizes[ty]
            align = max(align, sz)
            off = (off + sz - 1) // sz * sz + sz * max(n, 1)
        size = (off + align - 1) // align * align
        assert size == C.sizeof(mirrors[name]), f"{name}: generated layout {size} B, ctypes mirror {C.sizeof(mirrors[name])} B"
        flat = [f for f, *_ in mirrors[name]._fields_]
        assert [f for _, f, _ in fields] == flat, f"{name}: field order differs from the ctypes mirror"


def test_model_classes_only_call_declared_exports():
    _, structs, funcs, _ = gen.parse_header()
    errors, used = gen.check_templates(funcs)
    assert not errors, errors
    for need in ("nc_dac_create", "nc_dac_encode", "nc_dac_decode", "nc_dac_from_codes", "nc_dac_decode_code_matrix", "nc_dac_encode_code_matrix",
                 "nc_snac_create", "nc_snac_encode", "nc_snac_encode_tensor", "nc_snac_decode", "nc_snac_process_audio",
                 "nc_encodec_create", "nc_encodec_encode", "nc_encodec_decode", "nc_encodec_set_bandwidth", "nc_encodec_clip_length",
                 "nc_codec_load_weights", "nc_codec_destroy"):
        assert need in used, f"the model classes never call {need}"
    # struct initialisers name real fields of the generated structs
    fields = {gen.cs_struct_name(n): {f for _, f, _ in fl} for n, fl in structs}
    for fname in gen.MODEL_FILES[:3]:
        text = P.strip_comments(_native(fname))
        for m in re.finditer(r"new (Nc\w+Config)\s*\{(.*?)\};", text, flags=re.S):
            for fm in re.finditer(r"(\w+)\s*=(?!=)", m.group(2)):
                assert fm.group(1) in fields[m.group(1)], f"{fname}: {m.group(1)} has no field {fm.group(1)}"
        for m in re.finditer(r"\bc\.(\w+)\[", text):
            assert any(m.group(1) in f for f in fields.values()), f"{fname}: no struct has an array field {m.group(1)}"


def test_integration_tables_are_current():
    """INTEGRATION.md's member -> C ABI call tables are derived from the classes (tools/gen_csharp_shim.py rewrites them)."""
    assert gen.integration_md_current(), "INTEGRATION.md member tables are stale: run python tools/gen_csharp_shim.py"
    tables = gen.integration_tables()
    for must in ("ProcessAudio(float[], int)", "nc_snac_process_audio", "nc_encodec_clip_length", "Encode(Tensor, int?, int?)", "Models/Encodec.cs:213-235"):
        assert must in tables


# ------------------------------------------------------------------------------------------------------------- reference side
# (property, type) pairs the model classes assume when they read the reference's config objects
CONFIG_TYPES = {
    "DACConfig": {"SampleRate": "int", "EncoderDim": "int", "EncoderRates": "int[]", "DecoderDim": "int", "DecoderRates": "int[]",
                  "LatentDim": "int?", "NumCodebooks": "int", "CodebookSize": "int", "CodebookDim": "int", "Device": "DeviceConfiguration"},
    "SNACConfig": {"SampleRate": "int", "EncoderDim": "int", "EncoderRates": "int[]", "DecoderDim": "int", "DecoderRates": "int[]",
                   "LatentDim": "int?", "AttnWindowSize": "int?", "CodebookSize": "int", "CodebookDim": "int", "VQStrides": "int[]",
                   "Noise": "bool", "Depthwise": "bool", "Device": "DeviceConfiguration"},
    "EncodecConfig": {"SampleRate": "int", "Channels": "int", "Bandwidth": "float?", "TargetBandwidths": "float[]", "ChunkLengthSeconds": "float?",
                      "Overlap": "float?", "Normalize": "bool", "HiddenSize": "int", "NormType": "string", "UseCausalConv": "bool",
                      "CodebookSize": "int", "Device": "DeviceConfiguration"},
}
NULLABLE_SAFE = re.compile(r"\s*(\?\?|\.HasValue|\.Value\b|\s+is\b|;|=(?!=)|\}|,\s*$)")   # what may follow a nullable property read


@needs_reference
@pytest.mark.parametrize("native", sorted(MODELS))
def test_config_properties_exist_in_the_reference_with_the_assumed_types(native):
    fname, _, _, cfg_cls, cfg_file = MODELS[native]
    ref_props = P.members(_read(os.path.join(REF_T, cfg_file)), cfg_cls)["props"]
    text = P.strip_comments(_native(fname))
    used = set(re.findall(r"\b_?config\.(\w+)", text))
    assert used, "the class never reads its config?"
    for prop in sorted(used):
        assert prop in ref_props, f"{fname}: config.{prop} -- {cfg_cls} has no such public property ({cfg_file})"
        assert prop in CONFIG_TYPES[cfg_cls], f"{fname}: config.{prop} is read but its assumed type is not declared in CONFIG_TYPES"
        assert ref_props[prop] == CONFIG_TYPES[cfg_cls][prop], f"{cfg_cls}.{prop} is {ref_props[prop]} in the reference, the binding assumes {CONFIG_TYPES[cfg_cls][prop]}"
    assert set(CONFIG_TYPES[cfg_cls]) == used, f"CONFIG_TYPES[{cfg_cls}] lists properties the class no longer reads: {set(CONFIG_TYPES[cfg_cls]) - used}"
    for prop, ty in CONFIG_TYPES[cfg_cls].items():
        if not ty.endswith("?"):
            continue
        for m in re.finditer(r"\b_?config\." + prop + r"\b", text):
            tail = text[m.end():m.end() + 12]
            assert NULLABLE_SAFE.match(tail), f"{fname}: nullable {cfg_cls}.{prop} ({ty}) used as a plain value: ...{text[m.start():m.end() + 24]!r}"


@needs_reference
@pytest.mark.parametrize("native", sorted(MODELS))
def test_every_public_member_of_the_reference_model_is_mirrored(native):
    fname, ref_cls, ref_file, cfg_cls, _ = MODELS[native]
    ref = P.members(_read(os.path.join(REF_T, ref_file)), ref_cls)
    mine = P.members(_native(fname), native)
    have = {P.signature(r, n, p): body for r, n, p, body in mine["methods"]}
    assert len(ref["methods"]) >= 6
    for r, n, p, _ in ref["methods"]:
        if n in NOT_MIRRORED[ref_cls]["methods"]:
            continue
        sig = P.signature(r, n, p)
        assert sig in have, f"{native} lacks the reference member `{sig}` ({ref_file})"
        assert have[sig], f"{native}.{n} has no body"
    for name, ty in ref["props"].items():
        if name in NOT_MIRRORED[ref_cls]["props"]:
            continue
        assert mine["props"].get(name) == ty, f"{native} lacks the reference property `{ty} {name}` ({ref_file})"
    assert [[(t, n) for t, n, _ in c] for c in mine["ctors"]] == [[(cfg_cls, "config")]], f"{native}: constructor must take ({cfg_cls} config) like {ref_file}"
    # INeuralCodec (Core/INeuralCodec.cs:8-20): Config, LoadWeights + IDisposable
    iprops, imethods = P.interface_members(_read(os.path.join(REF_C, "INeuralCodec.cs")), "INeuralCodec")
    assert iprops == {"Config": "IModelConfig"} and [P.signature(*m) for m in imethods] == ["void LoadWeights(string)"]
    assert re.search(r"class " + native + r"\s*:\s*INeuralCodec\b", _native(fname))
    assert mine["props"].get("Config") == "IModelConfig" and "void LoadWeights(string)" in have and "void Dispose()" in have


@needs_reference
def test_surveyed_surface_is_present():
    """SURVEY 8b's list, spelled out (so that a change of NOT_MIRRORED cannot silently drop one of them)."""
    want = {
        "DACNative": ["(Tensor z, Tensor codes, Tensor latents, Tensor commitmentLoss, Tensor codebookLoss) Encode(Tensor, int?, int?)", "float[] Encode(float[])",
                      "Tensor Decode(Tensor)", "float[] Decode(float[])", "Tensor FromCodes(Tensor)", "Dictionary<string, Tensor> forward(Tensor, int?, int?)",
                      "Dictionary<string, Tensor> forward(Tensor)", "float[] forward(float[])"],
        "SNACNative": ["List<Tensor> Encode(Tensor)", "List<float[]> Encode(float[])", "Tensor Decode(List<Tensor>)", "float[] Decode(List<float[]>)",
                       "float[] ProcessAudio(float[], int)", "(Tensor audio, List<Tensor> codes) forward(Tensor)"],
        "EncodecNative": ["List<EncodedFrame> Encode(float[])", "List<EncodedFrame> Encode(Tensor)", "Tensor Decode(List<EncodedFrame>)",
                          "void SetTargetBandwidth(float)", "Tensor forward(Tensor)"],
    }
    for native, sigs in want.items():
        mine = P.members(_native(MODELS[native][0]), native)
        have = {P.signature(r, n, p) for r, n, p, _ in mine["methods"]}
        for s in sigs:
            assert s in have, f"{native}: {s}"
    props = P.members(_native("Encodec.Native.cs"), "EncodecNative")["props"]
    for name, ty in (("FrameRate", "int"), ("BitsPerCodebook", "int"), ("NumCodebooks", "int"), ("SegmentLength", "int?"), ("SegmentStride", "int?")):
        assert props.get(name) == ty                                          # Models/Encodec.cs:145-201


@needs_reference
def test_factories_have_the_reference_parameter_lists():
    ref = {n: (r, p) for r, n, p, _ in P.members(_read(os.path.join(REF_T, "NeuralCodecs.cs")), "NeuralCodecs")["methods"]}
    mine = {n: (r, p, b) for r, n, p, b in P.members(_native("NeuralCodecs.Native.cs"), "NeuralCodecs")["methods"]}
    assert re.search(r"public static partial class NeuralCodecs\b", _read(os.path.join(REF_T, "NeuralCodecs.cs")))      # so that ours can join it
    assert re.search(r"public static partial class NeuralCodecs\b", _native("NeuralCodecs.Native.cs"))
    assert re.search(r"^namespace NeuralCodecs\.Torch;", _native("NeuralCodecs.Native.cs"), flags=re.M)
    for kind in ("SNAC", "DAC", "Encodec"):
        r, p = ref[f"Create{kind}Async"]
        nr, np_, body = mine[f"Create{kind}NativeAsync"]
        assert r == f"Task<{kind}>" and nr == f"Task<{kind}Native>" and body
        assert p == np_, f"Create{kind}NativeAsync{np_} differs from the reference's Create{kind}Async{p}"          # types, names and defaults


def _reference_namespaces():
    ns = set()
    for path in glob.glob(os.path.join(REF, "NeuralCodecs.*", "**", "*.cs"), recursive=True):
        for m in re.finditer(r"^\s*namespace\s+([\w.]+)", _read(path), flags=re.M):
            ns.add(m.group(1))
    return ns


@needs_reference
def test_usings_exceptions_and_records_resolve_against_the_reference():
    namespaces = _reference_namespaces()
    exc = {}
    for path in glob.glob(os.path.join(REF_C, "Exceptions", "*.cs")):
        cls = os.path.splitext(os.path.basename(path))[0]
        exc[cls] = P.members(_read(path), cls)["ctors"]
    assert {"LoadException", "CodecException", "NeuralCodecException"} <= set(exc)
    frame = P.strip_comments(_read(os.path.join(REF_T, "Modules/Encodec/EncodedFrame.cs")))
    fm = re.search(r"public record EncodedFrame\((.*?)\);", frame)
    frame_params = P.parse_params(fm.group(1))
    assert [(t, n) for t, n, _ in frame_params] == [("Tensor", "Codes"), ("Tensor?", "Scale")]
    for fname in list(gen.MODEL_FILES) + ["NcMi355x.cs"]:
        text = P.strip_comments(_native(fname))
        for m in re.finditer(r"^using (?:static )?(NeuralCodecs[\w.]*);", text, flags=re.M):
            if m.group(1) == "NeuralCodecs.Torch.Native":
                continue                                                   # ours: declared by NcMi355x.cs
            assert m.group(1) in namespaces, f"{fname}: `using {m.group(1)}` -- the reference declares no such namespace"
        for m in re.finditer(r"\bnew (?:[\w.]+\.)?(\w+Exception)\s*\(", text):
            cls = m.group(1)
            if cls not in exc:
                continue                                                   # a BCL type
            e = P._balanced(text, m.end() - 1, "(", ")")
            n_args = len(P.split_top(text[m.end():e - 1]))
            ok = any(sum(1 for _, _, d in c if not d) <= n_args <= len(c) for c in exc[cls])
            assert ok, f"{fname}: new {cls}(...) with {n_args} argument(s): the reference has no such constructor"
        for m in re.finditer(r"\bnew EncodedFrame\s*\(", text):
            e = P._balanced(text, m.end() - 1, "(", ")")
            assert len(P.split_top(text[m.end():e - 1])) == 2
        for m in re.finditer(r"\bframe\.(\w+)", text):
            assert m.group(1) in ("Codes", "Scale")
    assert "namespace NeuralCodecs.Torch.Native;" in _native("NcMi355x.cs")
    # ModelLoadOptions.Device / DeviceConfiguration.{Type,Index} / DeviceType.CUDA as the binding uses them
    opts = P.members(_read(os.path.join(REF_C, "Loading/ModelLoadOptions.cs")), "ModelLoadOptions")["props"]
    dev = P.members(_read(os.path.join(REF_C, "Configuration/DeviceConfiguration.cs")), "DeviceConfiguration")["props"]
    assert opts.get("Device") == "DeviceConfiguration?" and dev.get("Type") == "DeviceType" and dev.get("Index") == "int"
    assert re.search(r"\bCUDA\b", _read(os.path.join(REF_C, "Configuration/DeviceType.cs")))


@needs_reference
def test_encodec_constructor_keeps_the_reference_bandwidth_rule():
    """Models/Encodec.cs:48-56,84: the bandwidth comes from config.Bandwidth, validated against TargetBandwidths with an
    ArgumentException; Overlap defaults to 0 (the generated class started from TargetBandwidths.Max())."""
    squash = lambda s: re.sub(r"\s+", "", s)
    ref = squash(P.strip_comments(_read(os.path.join(REF_T, "Models/Encodec.cs"))))
    mine_src = P.strip_comments(_native("Encodec.Native.cs"))
    ctor = mine_src[mine_src.index("public EncodecNative(EncodecConfig config)"):]
    ctor = squash(ctor[:P._balanced(ctor, ctor.index("{"), "{", "}")])
    rule = "if(config.Bandwidthisnull||!((IList<float>)config.TargetBandwidths).Contains(config.Bandwidth.Value)){thrownewArgumentException("
    assert rule in ref and rule in ctor
    assert "_bandwidth=config.Bandwidth;" in ref and "_bandwidth=config.Bandwidth;" in ctor
    assert "_overlap=config.Overlap??0;" in ref and "_overlap=config.Overlap??0;" in ctor
    assert "bandwidth=config.Bandwidth.Value" in ctor and "bandwidth=config.TargetBandwidths.Max()" not in ctor
    assert ctor.index("thrownewArgumentException(") < ctor.index("nc_encodec_create")                      # validated before anything is created
    # SetTargetBandwidth (Encodec.cs:409-419): membership check, then the switch; the config follows
    stb = squash(mine_src[mine_src.index("public void SetTargetBandwidth(float bandwidth)"):mine_src.index("EncodeHost(")])
    assert "if(!_targetBandwidths.Contains(bandwidth)){thrownewArgumentException(" in stb and "_config.Bandwidth=bandwidth;" in stb
    assert "if(!_targetBandwidths.Contains(bandwidth)){thrownewArgumentException(" in ref and "_config.Bandwidth=bandwidth;" in ref
