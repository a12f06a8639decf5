"""CPU suite: the C-ABI library loads, exports every symbol the header declares, and fails loudly without a GPU."""
import os
import re

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
