#!/usr/bin/env python3
"""Diagnostic: per-wave phase timeline of the k=7 conv kernel on one CU (needs the NC_DBG_TRACE build of libnc_mi355x.so)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from neuralcodecs_amd import _lib
lib = _lib.lib()
raw = C.CDLL(_lib.LIB_PATH)
W, S = 96, 640
cin = int(sys.argv[1]) if len(sys.argv) > 1 else 384
T = int(sys.argv[2]) if len(sys.argv) > 2 else 5568
FUSE = int(sys.argv[3]) if len(sys.argv) > 3 else 3
desc = _lib.NcConvDesc(32, cin, cin, 7, 1, 3, 1, 0, T, 0, 0)
ms = C.c_double()
buf = np.zeros(W * S, np.uint64); cnt = C.c_uint(0)
_lib.check(lib.nc_op_conv1d_bench(0, C.byref(desc), FUSE, 1, C.byref(ms)))   # includes a warm-up launch
raw.nc_dbg_trace_read(buf.ctypes.data_as(C.c_void_p), C.byref(cnt), 1)
_lib.check(lib.nc_op_conv1d_bench(0, C.byref(desc), FUSE, 1, C.byref(ms)))
raw.nc_dbg_trace_read(buf.ctypes.data_as(C.c_void_p), C.byref(cnt), 1)
print("ms", ms.value, "waves traced", cnt.value)
buf = buf.reshape(W, S)
np.save(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "convtrace.npy"), buf[: min(cnt.value, W)])
