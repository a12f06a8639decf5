"""CPU suite: the C-ABI library loads, exports every symbol the header declares, and fails loudly without a GPU."""
import os
import re
import sys

import numpy as np
import pytest

from neuralcodecs_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "nc_mi355x.h")).read()
    return sorted(set(re.findall(r"NC_API\s+[\w\s\*]+?\b(nc_\w+)\s*\(", src)))


def test_header_symbols_are_exported():
    L = _lib.lib()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/nc_mi355x.h but not exported by libnc_mi355x.so"
    bound = {s[0] for s in _lib.SYMBOLS}
    assert set(declared) == bound, f"ctypes table out of sync: {set(declared) ^ bound}"


def test_version_and_device_count():
    L = _lib.lib()
    assert b"gfx950" in L.nc_version()
    assert L.nc_device_count() >= 0


def test_host_side_weight_norm_fold_matches_oracle():
    from neuralcodecs_amd import ops
    from oracle import c_oracle
    rng = np.random.default_rng(3)
    v = rng.standard_normal((9, 5, 7)).astype(np.float32)
    g = rng.random(9).astype(np.float32)
    assert np.array_equal(ops.fold_weight_norm(v, g), c_oracle.fold_wn_dac(v, g))


def test_null_config_is_rejected():
    from neuralcodecs_amd import DAC
    with pytest.raises(ValueError):
        DAC(None)


@pytest.mark.skipif(_lib.lib().nc_device_count() > 0, reason="GPU present: covered by the gpu suite")
def test_no_cpu_fallback():
    """Without a device every entry point must fail with NC_EDEVICE -- never silently compute on the host."""
    from neuralcodecs_amd import DAC, DACConfig, ops
    with pytest.raises(_lib.NcDeviceError):
        DAC(DACConfig())
    with pytest.raises(_lib.NcDeviceError):
        ops.conv1d(np.zeros((1, 2, 8), np.float32), np.zeros((3, 2, 3), np.float32))


def test_struct_layouts_agree_with_a_compiled_c_consumer(tmp_path):
    """A C99 program built against include/nc_mi355x.h reports sizeof / offsetof of every struct that crosses the boundary; the
    ctypes mirrors in neuralcodecs_amd/_lib.py must agree field by field (and the program links against the library and calls it)."""
    import ctypes as C
    import subprocess
    structs = {"nc_dac_config": _lib.NcDacConfig, "nc_snac_config": _lib.NcSnacConfig, "nc_encodec_config": _lib.NcEncodecConfig,
               "nc_profile_entry": _lib.NcProfileEntry, "nc_conv_desc": _lib.NcConvDesc}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "nc_mi355x.h"', 'int main(void) {',
             '  printf("version %s\\n", nc_version());', '  printf("devices %d\\n", nc_device_count() >= 0);',
             '  printf("kc %d\\n", (int)NC_KC_COUNT);']
    for cname, ct in structs.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in ct._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "consumer.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "consumer"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-l:libnc_mi355x.so", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.check_output([str(exe)], text=True).splitlines()
    got = {}
    for ln in out:
        parts = ln.split()
        if parts[0] == "version":
            assert "gfx950" in ln
        elif parts[0] == "kc":
            assert int(parts[1]) == len(_lib.NC_KC_NAMES)
        elif parts[0] != "devices":
            got[(parts[0], parts[1])] = int(parts[2])
    for cname, ct in structs.items():
        assert got[(cname, "size")] == C.sizeof(ct), cname
        for fname, _ in ct._fields_:
            assert got[(cname, fname)] == getattr(ct, fname).offset, f"{cname}.{fname}"


def test_blob_parser_rejects_truncated_and_crafted_images():
    """nc_blob_check runs the loaders' parser on the host: every prefix of a valid image, and images with corrupted index fields,
    must come back NC_EINVAL (never a crash / out-of-bounds read); the intact image passes."""
    import ctypes as C
    import struct
    from neuralcodecs_amd.weights import save_blob
    L = _lib.lib()
    sd = {"a.weight_v": np.arange(24, dtype=np.float32).reshape(2, 3, 4), "b.bias": np.ones(5, np.float32), "c.alpha": np.zeros((1, 7, 1), np.float32)}
    blob = save_blob(sd)
    n = C.c_int32()
    assert L.nc_blob_check(blob, len(blob), C.byref(n)) == _lib.NC_OK and n.value == 3
    ok_from = min(c for c in range(len(blob) + 1) if L.nc_blob_check(blob[:c], c, None) == _lib.NC_OK)
    assert len(blob) - 64 < ok_from <= len(blob)                   # only the alignment padding behind the last tensor may be cut
    for cut in range(0, ok_from):
        assert L.nc_blob_check(blob[:cut], cut, None) == _lib.NC_EINVAL, f"prefix of {cut} bytes accepted"
    idx_len = struct.unpack_from("<Q", blob, 16)[0]
    bad = bytearray(blob)
    struct.pack_into("<Q", bad, 16, 2 ** 63)                       # index length that wraps the data origin
    assert L.nc_blob_check(bytes(bad), len(bad), None) == _lib.NC_EINVAL
    bad = bytearray(blob)
    struct.pack_into("<Q", bad, 8, 1000)                           # more tensors than the index holds
    assert L.nc_blob_check(bytes(bad), len(bad), None) == _lib.NC_EINVAL
    # corrupt every byte of the index in turn: accepted or rejected, never a crash; a changed byte count must be rejected
    for pos in range(24, 24 + int(idx_len)):
        bad = bytearray(blob)
        bad[pos] ^= 0xFF
        assert L.nc_blob_check(bytes(bad), len(bad), None) in (_lib.NC_OK, _lib.NC_EINVAL)
    assert L.nc_blob_check(b"NCWB0001" + b"\\0" * 8, 16, None) == _lib.NC_EINVAL
    assert L.nc_blob_check(None, 0, None) == _lib.NC_EINVAL


def test_group_entry_points_fail_cleanly_without_rccl():
    """on a host whose librccl cannot be opened every nc_group_* entry point must return NC_EDEVICE (with a message), not
    crash.  NC_RCCL_LIB names the one library the loader tries; a child process (the loader caches its result) points it at nothing."""
    import subprocess
    code = (
        "import ctypes as C, os, sys\n"
        "sys.path.insert(0, %r)\n"
        "from neuralcodecs_amd import _lib\n"
        "L = _lib.lib()\n"
        "buf = (C.c_char * 128)()\n"
        "st = L.nc_group_unique_id(buf)\n"
        "msg = L.nc_last_error().decode()\n"
        "assert st == _lib.NC_EDEVICE, st\n"
        "assert 'RCCL is not available' in msg, msg\n"
        "st2 = L.nc_group_unique_id(buf)\n"          # the cached failure path
        "assert st2 == _lib.NC_EDEVICE\n"
        "print('ok')\n" % ROOT)
    env = dict(os.environ, NC_RCCL_LIB="/nonexistent/librccl-not-here.so")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout, r.stderr)


def test_every_diagnostic_switch_is_in_the_one_table_and_documented():
    """the switches were 51 scattered getenv calls.  They now live in ONE table (csrc/nc_util.hip) that the accessors
    enforce; nc_debug_switches() lists it.  Hold the sources and DESIGN.md 10 to it."""
    import glob
    import re
    rows = [ln.split("\t") for ln in _lib.lib().nc_debug_switches().decode().splitlines()]
    names = {r[0] for r in rows}
    assert len(rows) == len(names) >= 50 and all(len(r) == 3 and r[1] in "bpis" and r[2] for r in rows)
    src_dir = os.path.join(ROOT, "neuralcodecs_amd", "csrc")
    for path in glob.glob(os.path.join(src_dir, "*.hip")) + glob.glob(os.path.join(src_dir, "*.h")):
        text = open(path).read()
        if not path.endswith("nc_util.hip"):
            assert "getenv" not in text, f"{os.path.basename(path)} reads the environment directly: use env_flag / env_int / env_str (nc_common.h)"
        for name in re.findall(r'env_(?:flag|present|int|str)\("(\w+)"', text) + re.findall(r'experiment_mode\("(\w+)"\)', text):
            assert name in names, f"{os.path.basename(path)}: {name} is not in the table"
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    missing = sorted(n for n in names if n not in design)
    assert not missing, f"DESIGN.md 10 does not mention {missing}"


def test_gpu_side_c_consumer_compiles_and_fails_loudly_without_a_device(tmp_path):
    """tests/abi_consumer.c (the C99 program the GPU suite runs through encode + decode) builds against the header with -Werror; on a host
    without a device it stops at nc_device_count with exit code 3 -- no silent host computation."""
    import subprocess
    import struct
    exe = tmp_path / "abi_consumer"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "abi_consumer.c"),
                           "-o", str(exe), "-L", libdir, "-l:libnc_mi355x.so", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-lm"])
    if _lib.lib().nc_device_count() > 0:
        pytest.skip("GPU present: tests/test_dac_gpu.py runs the program")
    import ctypes as C
    (tmp_path / "config.bin").write_bytes(b"\0" * C.sizeof(_lib.NcDacConfig))
    (tmp_path / "pcm.f32").write_bytes(struct.pack("<4f", 0, 0, 0, 0))
    r = subprocess.run([str(exe), str(tmp_path / "config.bin"), "none", str(tmp_path / "pcm.f32"), "1", "4", "none", "none"], capture_output=True, text=True)
    assert r.returncode == 3 and "no gfx950 device" in r.stderr, (r.returncode, r.stderr)
