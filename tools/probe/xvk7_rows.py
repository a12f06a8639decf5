#!/usr/bin/env python3
"""Runs ON THE GPU BOX: the XV-only k = 7 instance against the legacy one as a function of the row length (C = 768, dilation 9, B = 32):
where does the short-row loss of profiles/r05_xvk7_per_layer.txt come from -- row alignment (696 * 4 B is no multiple of 128) or the share of
edge tiles?   python tools/probe/xvk7_rows.py   (spawns itself per setting: the switches are read once per process)"""
import ctypes as C
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ROWS = (696, 704, 1024, 1392, 2048, 2784, 5568)
if len(sys.argv) > 1:
    from neuralcodecs_amd import _lib
    lib = _lib.lib()
    for T in ROWS:
        for d in (1, 9):
            desc = _lib.NcConvDesc(32, 768, 768, 7, 1, 3 * d, d, 0, T, 0, 0)
            ms = C.c_double()
            _lib.check(lib.nc_op_conv1d_bench(0, C.byref(desc), 3, 40, C.byref(ms)))
            print(f"{sys.argv[1]:8s} T={T:5d} d={d} {ms.value * 1e3:9.1f} us  {2.0 * 768 * 768 * 7 * T * 32 / ms.value / 1e9:7.1f} TF/s", flush=True)
else:
    for name, env in (("legacy", {"NC_NO_XV_K7": "1"}), ("xv", {"NC_XV_K7_MIN_COLS": "0", "NC_NO_FLAT": "1"}), ("legacy-noflat", {"NC_NO_XV_K7": "1", "NC_NO_FLAT": "1"})):
        subprocess.run([sys.executable, os.path.abspath(__file__), name], env=dict(os.environ, **env))
