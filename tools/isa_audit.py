#!/usr/bin/env python3
"""ISA audit of the matrix-core loops (round 4): compile a kernel source to gfx950 assembly and report, per kernel and per loop that holds
>= MIN_MFMA matrix-core instructions, what else sits in the loop -- vector-ALU instructions (they cost matrix-pipe issue time, DESIGN 8
round 4), `v_readlane` reloads of spilled scalar registers (vector instructions the source never asked for), LDS reads, waits -- plus the
kernel's register and spill counts.  Static counts over ALL paths of a loop (run-time branches included), so compare like with like.

    python tools/isa_audit.py neuralcodecs_amd/csrc/nc_conv_k7.hip [--grep 'ILi3ELi2ELi7'] [--min-mfma 64] [-D...]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source")
    ap.add_argument("--grep", default="")
    ap.add_argument("--min-mfma", type=int, default=64)
    ap.add_argument("-D", action="append", default=[])
    a = ap.parse_args()
    src = os.path.abspath(a.source)
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
