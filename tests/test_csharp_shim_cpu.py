"""CPU suite: the C# P/Invoke shim (bindings/csharp/) is GENERATED from include/nc_mi355x.h and checked mechanically (dotnet is not in
the image, so it cannot be compiled here): the committed files equal what the header generates today; there is one [DllImport] stub
per NC_API export with the header's parameter count; every struct mirrors its C struct field for field at the same size as the ctypes
mirrors the parity tests drive the library with; and every NcMi355x.nc_* call made by the DAC / SNAC / Encodec partial-class bodies
exists in the header with that many arguments.  Reference surface kept: Core/INeuralCodec.cs:8-20, Models/DAC.cs:163-253,
Models/SNAC.cs:113-192, Models/Encodec.cs:213-285,409-419."""
import ctypes as C
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import gen_csharp_shim as gen  # noqa: E402


def test_committed_shim_is_what_the_header_generates():
    files = gen.generate()
    assert set(files) == {"NcMi355x.cs", "DAC.Native.cs", "SNAC.Native.cs", "Encodec.Native.cs"}
    for name, text in files.items():
        p = os.path.join(gen.OUT, name)
        assert os.path.exists(p), f"bindings/csharp/{name} missing: run tools/gen_csharp_shim.py"
        assert open(p).read() == text, f"bindings/csharp/{name} is out of date: run tools/gen_csharp_shim.py"


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


def test_partial_class_bodies_only_call_declared_exports():
    _, _, funcs, _ = gen.parse_header()
    errors, used = gen.check_templates(funcs)
    assert not errors, errors
    for need in ("nc_dac_create", "nc_dac_encode", "nc_dac_decode", "nc_dac_from_codes", "nc_snac_encode", "nc_snac_encode_tensor", "nc_snac_decode",
                 "nc_encodec_encode", "nc_encodec_decode", "nc_encodec_set_bandwidth", "nc_codec_load_weights", "nc_codec_destroy"):
        assert need in used, f"the partial classes never call {need}"
