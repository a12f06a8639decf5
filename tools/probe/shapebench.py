#!/usr/bin/env python3
"""Time arbitrary conv shapes through nc_op_conv1d_bench:  python tools/probe/shapebench.py B,Cin,Cout,K,stride,pad,T[,transposed[,fuse]] ..."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuralcodecs_amd import _lib  # noqa: E402

lib = _lib.lib()
for spec in sys.argv[1:]:
    v = [int(x) for x in spec.split(",")]
    B, cin, cout, k, s, p, T = v[:7]
    tr = v[7] if len(v) > 7 else 0
    fuse = v[8] if len(v) > 8 else 0      # 1 snake-in, 2 snake-out, 4 residual, 8 GroupNorm sums out, 16 Encodec input mode
    desc = _lib.NcConvDesc(B, cin, cout, k, s, p, 1, 0, T, tr, 0)
    ms = C.c_double()
    _lib.check(lib.nc_op_conv1d_bench(0, C.byref(desc), fuse, 10, C.byref(ms)))
    Tout = (T - 1) * s - 2 * p + k if tr else (T + 2 * p - (k - 1) - 1) // s + 1
    by = 4.0 * B * (cin * T + cout * Tout)
    fl = 2.0 * cin * cout * k * (T if tr else Tout) * B
    print(f"{spec:40s} {ms.value*1e3:9.1f} us  {fl/ms.value/1e9:7.1f} TF/s  {by/ms.value/1e6:7.0f} GB/s", flush=True)
