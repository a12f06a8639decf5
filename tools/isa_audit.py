#!/usr/bin/env python3
"""ISA audit of the matrix-core loops (round 4): compile a kernel source to gfx950 assembly and report, per kernel and per loop that holds
>= MIN_MFMA matrix-core instructions, what else sits in the loop -- vector-ALU instructions (they cost matrix-pipe issue time, DESIGN 8
round 4), `v_readlane` reloads of spilled scalar registers (vector instructions the source never asked for), LDS reads, waits -- plus the
kernel's register and spill counts.  Static counts over ALL paths of a loop (run-time branches included), so compare like with like.

    python tools/isa_audit.py neuralcodecs_amd/csrc/nc_conv_k7.hip [--grep 'ILi3ELi2ELi7'] [--min-mfma 64] [-D...]
    python tools/isa_audit.py neuralcodecs_amd/csrc/build/nc_conv_k7.o [--grep ...] [--json out.json]     # a BUILT object: seconds
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels(asm):
    name, lines = None, []
    for l in open(asm):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            name, lines = m.group(1), []
            continue
        if name is None:
            continue
        t = l.split(";")[0].strip()
        if t:
            lines.append(t)
        if "s_endpgm" in l:
            yield name, lines
            name = None


def meta(asm):
    out, cur = {}, {}
    for l in open(asm):
        m = re.match(r"\s+\.(name|sgpr_count|sgpr_spill_count|vgpr_count|vgpr_spill_count):\s+(\S+)", l)
        if m:
            cur[m.group(1)] = m.group(2)
            if m.group(1) == "vgpr_spill_count" or (m.group(1) == "name" and len(cur) > 1 and "vgpr_count" in cur):
                pass
        if l.strip().startswith(".wavefront_size") and "name" in cur:
            out[cur["name"]] = dict(cur)
            cur = {}
    return out


def audit_object(obj, grep="", min_mfma=64):
    """Round 5: the same audit over a BUILT object (neuralcodecs_amd/csrc/build/*.o) -- the gfx950 code object is unbundled and
    disassembled (about two seconds, no compilation), so a test can hold the shipped kernels to a committed table.  Returns
    {kernel: {"loops": [{"mfma", "valu", "v_readlane", "ds_read", "waits"}, ...]}} for the loops with >= min_mfma matrix-core instructions
    (static counts over all paths of a loop, like the source form)."""
    llvm = "/opt/rocm/lib/llvm/bin"
    out = {}
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "k.co")
        subprocess.check_call([llvm + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
        subprocess.check_call([llvm + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               "--input=" + fat, "--output=" + co])
        dis = subprocess.run([llvm + "/llvm-objdump", "-d", "--symbolize-operands", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout
        notes = subprocess.run([llvm + "/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    # the code object's kernel metadata (round 6): registers, spills and scratch -- a spill the loop audit cannot see (scratch traffic sits
    # outside the counted mnemonics) shows up here as private_segment_fixed_size / vgpr_spill_count
    res, cur = {}, {}
    for l in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(name|private_segment_fixed_size|sgpr_count|sgpr_spill_count|vgpr_count|vgpr_spill_count|agpr_count|group_segment_fixed_size):\s+(\S+)", l)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "name":
            if v.startswith("_Z") and not v.endswith(".kd"):
                cur = res.setdefault(v, {})
            continue
        if cur is not None:
            cur[k] = int(v)
    name, lines = None, []

    def flush():
        if name is None or (grep and not re.search(grep, name)):
            return
        labels = {t[1:-2]: i for i, t in enumerate(lines) if re.match(r"^<L\d+>:$", t)}
        loops, seen = [], set()
        for i, t in enumerate(lines):
            m = re.match(r"s_c?branch\w*\s+(L\d+)", t)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                lo = labels[m.group(1)]
                c = Counter(x.split()[0] for x in lines[lo:i] if not x.startswith("<"))
                n = sum(v for k, v in c.items() if k.startswith("v_mfma"))
                if n >= min_mfma and (lo, n) not in seen:
                    seen.add((lo, n))
                    loops.append({"mfma": n, "valu": sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma")),
                                  "v_readlane": c["v_readlane_b32"], "ds_read": sum(v for k, v in c.items() if k.startswith("ds_read")),
                                  "waits": c["s_waitcnt"]})
        if loops:
            r = res.get(name, {})
            out[name] = {"loops": loops, "scratch_bytes": r.get("private_segment_fixed_size"), "vgpr": r.get("vgpr_count"), "agpr": r.get("agpr_count"),
                         "vgpr_spill": r.get("vgpr_spill_count"), "sgpr": r.get("sgpr_count"), "sgpr_spill": r.get("sgpr_spill_count")}
    for l in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(_Z\w+)>:$", l)
        if m:
            flush()
            name, lines = m.group(1), []
            continue
        if name is None:
            continue
        m = re.match(r"^[0-9a-f]+ (<L\d+>:)$", l)
        if m:
            lines.append(m.group(1))
            continue
        t = l.split("//")[0].strip()
        if t:
            lines.append(t)
    flush()
    return out


# The hot instances of the BASELINE configurations (profiles/r0*_*.kernel_stats.txt name them): (object, regular expression over the mangled
# kernel name, what it runs).  `--table` audits them from the built objects into one JSON record; tests/test_isa_audit_cpu.py holds the build
# to the committed record (profiles/r05_isa_audit.json): an in-loop `v_readlane` count may fall, never rise.
HOT = [
    ("nc_conv_k7.o", r"conv_mfma_kernelILi3ELi2ELi7ELi8ELi10ELb0ELi2ELi4ELi0ELb0ELi0ELb0E", "k = 7 residual-unit convolution, 96 x 256 tiles (DAC C = 192 / 384 / 768)"),
    ("nc_conv_k7.o", r"conv_mfma_kernelILi2ELi2ELi7ELi8ELi10ELb0ELi2ELi4ELi0ELb0ELi0ELb0E", "k = 7, 64 x 256 tiles (DAC C = 512)"),
    ("nc_conv_k7f.o", r"conv_mfma_kernelILi3ELi2ELi7ELi8ELi10ELb1E", "fused residual unit C = 96"),
    ("nc_conv_k7f.o", r"conv_mfma_kernelILi4ELi2ELi7ELi8ELi10ELb1E", "fused residual unit C = 128"),
    ("nc_conv_k7f.o", r"conv_mfma_kernelILi2ELi2ELi7ELi8ELi10ELb1E", "fused residual unit C = 64"),
    ("nc_conv_k7g.o", r"conv_mfma_kernelILi8ELi1ELi7ELi4ELi5ELb1E", "wide fused residual unit C = 256"),
    ("nc_conv_k2.o", r"conv_mfma_kernelILi3ELi2ELi2ELi16ELi20ELb0ELi2ELi4ELi0ELb0ELi1ELb0E", "two-tap sub-pixel up-convolution, 96 x 256 tiles (DAC / SNAC conv_up)"),
    ("nc_conv_k2.o", r"conv_mfma_kernelILi4ELi2ELi2ELi16ELi20ELb0ELi2ELi4ELi0ELb0ELi1ELb0E", "two-tap sub-pixel up-convolution, 128 x 256 tiles"),
    ("nc_conv_xv2.o", r"conv_mfma_kernelILi2ELi2ELi2ELi16ELi20ELb0ELi2ELi4ELi0ELb0ELi1ELb0ELb1ELb0E", "XV-only two-tap up-convolution, 64 x 256 tiles (DAC 192 -> 96, 384 -> 192; SNAC 192 -> 96)"),
    ("nc_conv_xv2.o", r"conv_mfma_kernelILi3ELi2ELi2ELi16ELi20ELb0ELi2ELi4ELi0ELb0ELi1ELb0ELb1ELb0E", "XV-only two-tap up-convolution, 96 x 256 tiles (DAC 768 -> 384)"),
    ("nc_conv_xv2.o", r"conv_mfma_kernelILi4ELi2ELi2ELi16ELi20ELb0ELi2ELi4ELi0ELb0ELi1ELb0ELb1ELb0E", "XV-only two-tap up-convolution, 128 x 256 tiles (SNAC 1536 -> 768, 768 -> 384)"),
    ("nc_conv_xv2g.o", r"conv_mfma_kernelILi3ELi2ELi2ELi16ELi20ELb0ELi2ELi4ELi0ELb0ELi2ELb0ELb1ELb0E", "XV-only two-tap up-convolution, any stride (SNAC stride 3)"),
    ("nc_conv_xv7.o", r"conv_mfma_kernelILi3ELi2ELi7ELi8ELi10ELb0ELi2ELi4ELi0ELb0ELi0ELb0ELb1ELb0E", "XV-only k = 7, 96 x 256 tiles (DAC C = 192 / 384 long rows)"),
    ("nc_conv_xv7f.o", r"conv_mfma_kernelILi3ELi2ELi7ELi8ELi10ELb1ELi2ELi4ELi0ELb0ELi0ELb0ELb1ELb0E", "XV-only fused residual unit C = 96"),
    ("nc_conv_xv7f.o", r"conv_mfma_kernelILi2ELi2ELi7ELi8ELi10ELb1ELi2ELi4ELi0ELb0ELi0ELb0ELb1ELb0E", "XV-only fused residual unit C = 64"),
    ("nc_conv_k4.o", r"conv_mfma_kernelILi4ELi2ELi4ELi8ELi18ELb0ELi2ELi4ELi0ELb0ELi0ELb0E", "k = 4 stride-2 down-convolution (DAC 64 -> 128)"),
    ("nc_conv_k8.o", r"conv_mfma_kernelILi4ELi2ELi8ELi4ELi18ELb0ELi2ELi4ELi0ELb0ELi0ELb0E", "k = 8 stride-4 down-convolution (DAC 128 -> 256)"),
    ("nc_conv_k16.o", r"conv_mfma_kernelILi4ELi2ELi16ELi2ELi18ELb0ELi2ELi4ELi0ELb0ELi0ELb0E", "k = 16 stride-8 down-convolution (SNAC 44 kHz, long rows)"),
    ("nc_conv_k16.o", r"conv_mfma_kernelILi2ELi2ELi16ELi2ELi18ELb0ELi2ELi4ELi0ELb0ELi0ELb0E", "k = 16 stride-8, 64 x 256 tiles"),
    ("nc_resa.o", r"res_a_kernelILi1ELb1E", "Encodec residual block, fused first pass, C = 32"),
    ("nc_resa.o", r"res_a_kernelILi2ELb1E", "Encodec residual block, fused first pass, C = 64"),
]


def compiler_version():
    """First line of `hipcc --version`'s clang line: the record is only comparable under the compiler that produced it."""
    try:
        out = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True).stdout
        for l in out.splitlines():
            if "clang version" in l:
                return l.strip()
        return out.splitlines()[0].strip() if out else "unknown"
    except OSError:
        return "unknown"


def hot_table(build_dir):
    tab = {}
    for obj, rx, what in HOT:
        path = os.path.join(build_dir, obj)
        if not os.path.exists(path):
            continue
        for k, v in audit_object(path, rx, 48).items():
            tab[k] = {"object": obj, "what": what, "loops": v["loops"], "max_in_loop_v_readlane": max(lp["v_readlane"] for lp in v["loops"]),
                      "scratch_bytes": v["scratch_bytes"], "vgpr": v["vgpr"], "vgpr_spill": v["vgpr_spill"], "sgpr": v["sgpr"], "sgpr_spill": v["sgpr_spill"]}
    return tab


def main():
    if len(sys.argv) >= 2 and sys.argv[1] == "--table":
        import json
        tab = hot_table(os.path.join(ROOT, "neuralcodecs_amd", "csrc", "build"))
        for k, v in tab.items():
            print("readlane %-3d scratch %-4s vgpr %-3s (+%s spilled) sgpr %-3s (+%s spilled)  %s  (%s)" % (
                v["max_in_loop_v_readlane"], v["scratch_bytes"], v["vgpr"], v["vgpr_spill"], v["sgpr"], v["sgpr_spill"], k, v["what"]))
        if len(sys.argv) >= 3:
            json.dump({"_about": "tools/isa_audit.py --table: in-loop instruction counts of the hot matrix-core instances and their register / spill / "
                                 "scratch figures, from the built objects (static over all paths of a loop); tests/test_isa_audit_cpu.py holds the "
                                 "build to max_in_loop_v_readlane, the vector-instruction counts, scratch_bytes and vgpr_spill",
                       "compiler": compiler_version(), "kernels": tab}, open(sys.argv[2], "w"), indent=1)
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("source", help="a .hip source (compiled to assembly here) or a built .o (disassembled: fast)")
    ap.add_argument("--grep", default="")
    ap.add_argument("--min-mfma", type=int, default=64)
    ap.add_argument("-D", action="append", default=[])
    ap.add_argument("--json", default="", help="built-object form: write the table to this file")
    a = ap.parse_args()
    src = os.path.abspath(a.source)
    if src.endswith(".o"):
        import json
        tab = audit_object(src, a.grep, a.min_mfma)
        for k, v in tab.items():
            print(k)
            for lp in v["loops"]:
                print("    loop  mfma %4d  valu %5d  of which v_readlane %4d  ds_read %4d  s_waitcnt %4d" % (lp["mfma"], lp["valu"], lp["v_readlane"], lp["ds_read"], lp["waits"]))
        if a.json:
            json.dump(tab, open(a.json, "w"), indent=1)
        return
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden",
               "-I" + os.path.dirname(src), "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", asm, src] + ["-D" + d for d in a.D]
        r = subprocess.run(cmd, cwd=os.path.dirname(src), capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr[-2000:])
        md = meta(asm)
        for name, lines in kernels(asm):
            if a.grep and not re.search(a.grep, name):
                continue
            labels = {t[:-1]: i for i, t in enumerate(lines) if t.endswith(":")}
            rows = []
            for i, t in enumerate(lines):
                m = re.match(r"s_c?branch\w*\s+(\S+)", t)
                if m and m.group(1) in labels and labels[m.group(1)] < i:
                    lo = labels[m.group(1)]
                    c = Counter(x.split()[0] for x in lines[lo:i])
                    n = sum(v for k, v in c.items() if k.startswith("v_mfma"))
                    if n >= a.min_mfma:
                        valu = sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma"))
                        rows.append((lines[lo], n, valu, c["v_readlane_b32"], sum(v for k, v in c.items() if k.startswith("ds_read")), c["s_waitcnt"]))
            if not rows:
                continue
            g = md.get(name, {})
            print(f"{name}\n    sgpr {g.get('sgpr_count')} (+{g.get('sgpr_spill_count')} spilled)  vgpr {g.get('vgpr_count')} (+{g.get('vgpr_spill_count')} spilled)")
            seen = set()
            for lab, n, valu, rl, dr, wc in rows:
                if (lab, n) in seen:
                    continue
                seen.add((lab, n))
                print(f"    loop {lab:14s} mfma {n:4d}  valu {valu:5d}  of which v_readlane {rl:4d}  ds_read {dr:4d}  s_waitcnt {wc:4d}")


if __name__ == "__main__":
    main()
