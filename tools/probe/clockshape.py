#!/usr/bin/env python3
"""Shader clock while ONE conv shape loops (nc_op_conv1d_bench) -- is a layer's rate a clock (power) effect?
   python tools/probe/clockshape.py B,Cin,Cout,K,stride,pad,T[,transposed[,fuse]] ...     (tools/probe/libclockprobe.so on a side stream)"""
import ctypes as C
import os
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, "..", ".."))
from neuralcodecs_amd import _lib  # noqa: E402

probe_lib = C.CDLL(os.path.join(here, "libclockprobe.so"))
probe_lib.clock_probe_launch.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
dev = torch.device("cuda:0")
out = torch.zeros(2, dtype=torch.int64, device=dev)
side = torch.cuda.Stream()
lib = _lib.lib()
for spec in sys.argv[1:]:
    v = [int(x) for x in spec.split(",")]
    B, cin, cout, k, s, p, T = v[:7]
    tr = v[7] if len(v) > 7 else 0
    fuse = v[8] if len(v) > 8 else 0
    desc = _lib.NcConvDesc(B, cin, cout, k, s, p, 1, 0, T, tr, 0)
    ms = C.c_double()
    _lib.check(lib.nc_op_conv1d_bench(0, C.byref(desc), fuse, 5, C.byref(ms)))   # warm-up, gives the per-launch time
    iters = max(20, int(250.0 / ms.value))                                        # ~250 ms of back-to-back launches
    probe_lib.clock_probe_launch(out.data_ptr(), int(150 * 1e5), side.cuda_stream)  # 150 ms window starting now
    _lib.check(lib.nc_op_conv1d_bench(0, C.byref(desc), fuse, iters, C.byref(ms)))
    side.synchronize()
    c, w = out.tolist()
    Tout = (T - 1) * s - 2 * p + k if tr else (T + 2 * p - (k - 1) - 1) // s + 1
    fl = 2.0 * cin * cout * k * (T if tr else Tout) * B
    print(f"{spec:40s} {ms.value*1e3:9.1f} us  {fl/ms.value/1e9:7.1f} TF/s   shader clock {c / w * 100:.0f} MHz over {w/1e5:.0f} ms", flush=True)
