#!/usr/bin/env python3
"""Generate the C# P/Invoke shim of libnc_mi355x.so from include/nc_mi355x.h.

    python tools/gen_csharp_shim.py            # writes bindings/csharp/*.cs
    python tools/gen_csharp_shim.py --check    # exit 1 when the committed files differ from what the header generates

Output (a maintainer of the reference drops bindings/csharp/ into NeuralCodecs.Torch/Native/):
  bindings/csharp/NcMi355x.cs       GENERATED: every enum, struct layout and NC_API export of the header as [DllImport] stubs + the
                                    status -> exception map of SURVEY 8b.  Mechanical: one stub per export, parameter for parameter.
Checked, not generated (hand-written against those stubs and against the reference's own types):
  bindings/csharp/DAC.Native.cs     DACNative / SNACNative / EncodecNative : INeuralCodec with every public member of the reference's
  bindings/csharp/SNAC.Native.cs    DAC / SNAC / Encodec classes, signature for signature; NeuralCodecs.Create*NativeAsync factories.
  bindings/csharp/Encodec.Native.cs Every NcMi355x.nc_* call they make must exist in the header with the same number of arguments
  bindings/csharp/NeuralCodecs.Native.cs   (check_templates); tests/test_csharp_shim_cpu.py additionally parses the REFERENCE's .cs files
                                    (Config/*Config.cs, Models/*.cs, NeuralCodecs.cs, Core/Exceptions) and holds the classes to them.
dotnet is not available in the build image, so the files are cross-checked, not compiled.
"""
import argparse
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "nc_mi355x.h")
OUT = os.path.join(ROOT, "bindings", "csharp")

SCALARS = {"int": "int", "int32_t": "int", "int64_t": "long", "uint64_t": "ulong", "uint32_t": "uint", "size_t": "nuint", "float": "float", "double": "double",
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
    if re.match(r"(\w+)\s*\*\s*const\s*\*$", t) and re.match(r"(\w+)", t).group(1) in SCALARS:
        return "IntPtr[]"           # an array of per-device DEVICE pointers (local-mode groups): opaque addresses on the managed side
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
    o.append("    // IModelConfig.Device (Core/Configuration/DeviceConfiguration.cs) -> device ordinal of the engine: the GPU index when the")
    o.append("    // config names an accelerator, device 0 otherwise (the engine has no CPU path; the reference's default config says CPU)")
    o.append("    public static int DeviceIndex(NeuralCodecs.Core.Configuration.DeviceConfiguration? d) =>")
    o.append("        d is not null && d.Type == NeuralCodecs.Core.Configuration.DeviceType.CUDA ? d.Index : 0;")
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
    o.append("            _ => new NeuralCodecs.Core.Exceptions.NeuralCodecException(msg),       // NC_EDEVICE")
    o.append("        };")
    o.append("    }")
    o.append("}")
    return "\n".join(o) + "\n"


# ---- hand-written model classes (bindings/csharp/*.Native.cs): checked here and in tests/test_csharp_shim_cpu.py -------------------------
MODEL_FILES = ("DAC.Native.cs", "SNAC.Native.cs", "Encodec.Native.cs", "NeuralCodecs.Native.cs")


def model_sources():
    return {f: open(os.path.join(OUT, f)).read() for f in MODEL_FILES}


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
    for fname, text in model_sources().items():
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


def integration_tables():
    """Member -> C ABI call tables of INTEGRATION.md, derived from the model classes themselves: for every public member, the
    nc_* exports its body reaches (through the class's own host-array cores), and the reference line it cites in its trailing comment."""
    import csharp_parse as P
    out = []
    for fname, cls in (("DAC.Native.cs", "DACNative"), ("SNAC.Native.cs", "SNACNative"), ("Encodec.Native.cs", "EncodecNative")):
        raw = open(os.path.join(OUT, fname)).read()
        m = P.members(raw, cls)
        names = [n for _, n, _, _ in m["methods"]]
        direct = [set(re.findall(r"NcMi355x\.(nc_\w+)", b)) for b in m["bodies"]]
        callees = [{n for n in set(names) if re.search(r"(?<![\w.])" + n + r"\s*\(", b)} for b in m["bodies"]]

        def reach(i, seen):
            calls = set(direct[i])
            for j, n in enumerate(names):
                if n in callees[i] and j not in seen and j != i:
                    calls |= reach(j, seen | {i})
            return calls

        cites = {}                                              # "Name(params" -> the Models/...cs:line citation on the declaration line
        for line in raw.splitlines():
            decl = re.sub(r"^(\s*public\s+)\([^)]*\)\s+", r"\1T ", line)                 # a tuple return type holds parentheses
            cm = re.match(r"\s*(?:public\s+[^(]*?)?(\w+)\(([^)]*)\).*//\s*(Models/[\w./]+:[\d\-,]+)", decl)
            if cm and cm.group(1) in names:
                cites.setdefault((cm.group(1), re.sub(r"\s+", "", cm.group(2))), cm.group(3))
        ctor = re.search(r"public " + cls + r"\((.*?)\).*//\s*(Models/[\w./]+:[\d\-,]+)", raw)
        cbody = raw[ctor.end():]
        out.append(f"**`{cls}`** (`bindings/csharp/{fname}`)")
        out.append("")
        out.append("| Managed member (reference signature) | reference | C ABI calls |")
        out.append("|---|---|---|")
        out.append(f"| `new {cls}({ctor.group(1)})` | `{ctor.group(2)}` | " + ", ".join(f"`{c}`" for c in sorted(set(re.findall(r"NcMi355x\.(nc_\w+_create)", cbody)))) + " |")
        for i, (ret, name, params, _) in enumerate(m["methods"]):
            if name.endswith("Host"):
                continue
            key = re.sub(r"[\s?]+", "", ",".join(f"{t}{n}" for t, n, _ in params))
            cite = next((v for (n2, p2), v in cites.items() if n2 == name and re.sub(r"=\w+|\?", "", p2) == key), "")
            calls = sorted(reach(i, set()))
            out.append(f"| `{ret} {name}({', '.join(t for t, _, _ in params)})` | {('`' + cite + '`') if cite else ''} | " + (", ".join(f"`{c}`" for c in calls) or "managed only") + " |")
        out.append("")
    return "\n".join(out)


BEGIN_MARK, END_MARK = "<!-- BEGIN generated member tables (tools/gen_csharp_shim.py) -->", "<!-- END generated member tables -->"


def integration_md_current():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    if BEGIN_MARK not in md or END_MARK not in md:
        return False
    cur = md[md.index(BEGIN_MARK) + len(BEGIN_MARK):md.index(END_MARK)].strip()
    return cur == integration_tables().strip()


def write_integration_md():
    path = os.path.join(ROOT, "INTEGRATION.md")
    md = open(path).read()
    block = BEGIN_MARK + "\n" + integration_tables().strip() + "\n" + END_MARK
    if BEGIN_MARK in md and END_MARK in md:
        md = md[:md.index(BEGIN_MARK)] + block + md[md.index(END_MARK) + len(END_MARK):]
        open(path, "w").write(md)
        return True
    return False


def generate():
    enums, structs, funcs, defines = parse_header()
    errs, _ = check_templates(funcs)
    if errs:
        raise SystemExit("gen_csharp_shim: " + "; ".join(errs))
    return {"NcMi355x.cs": gen_native(enums, structs, funcs, defines)}


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
    if a.check and not integration_md_current():
        bad.append("INTEGRATION.md member tables")
    if not a.check:
        write_integration_md()
    if a.check and bad:
        print("out of date:", ", ".join(bad))
        sys.exit(1)
    print("ok:", ", ".join(files))
