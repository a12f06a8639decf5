#!/usr/bin/env python3
"""Reads the in-kernel phase trace of the convolution template (-DNC_CONV_TRACE builds; NC_CONV_TRACE_FILE=<path>, NC_CONV_TRACE_SEL="K,Cin,dil")
and prints, per reduction block and averaged over the traced wavefronts: the matrix-core segments, the staging runs at the segment heads,
the wait at the closing barrier -- in shader cycles and as shares of a block.  python tools/probe/conv_trace.py <file>"""
import struct
import sys

import numpy as np

raw = open(sys.argv[1], "rb").read()
nwg, nw, ncb, nst, TM, TN, K, xv = struct.unpack("8i", raw[:32])
t = np.frombuffer(raw[32:], dtype=np.uint64).reshape(nwg, 8, ncb, nst)[:, :nw].astype(np.int64)
ok = (t > 0).all(axis=(2, 3))
print(f"tile {TM}x{TN} K={K} xv={xv}: {int(ok.sum())} of {nwg * nw} wavefronts traced, {ncb} blocks each")
t = t[ok]                                   # [waves, ncb, 8]
# stamps: 0 top | 1,2 staging at seg 1 head | 3,4 seg 2 | 5,6 seg 3 | 7 before barrier ; next block's 0 = behind the barrier
seg0 = t[:, :, 1] - t[:, :, 0]
st1 = t[:, :, 2] - t[:, :, 1]
seg1 = t[:, :, 3] - t[:, :, 2]
st2 = t[:, :, 4] - t[:, :, 3]
seg2 = t[:, :, 5] - t[:, :, 4]
st3 = t[:, :, 6] - t[:, :, 5]
seg3 = t[:, :, 7] - t[:, :, 6]
bar = t[:, 1:, 0] - t[:, :-1, 7]
blk = t[:, 1:, 0] - t[:, :-1, 0]
mf = TM * TN * (K * (16 if K == 2 else 8 if K == 7 else 2) // 2) * 64 // 4   # matrix-pipe cycles of ONE segment of one wave (4 segments)
def row(name, a):
    print(f"  {name:22s} mean {a.mean():9.0f}  p10 {np.percentile(a, 10):9.0f}  p90 {np.percentile(a, 90):9.0f}  ({100.0 * a.mean() / blk.mean():5.1f} % of a block)")
row("block (top to top)", blk)
for n, a in (("segment 0", seg0), ("staging @ seg 1", st1), ("segment 1", seg1), ("staging @ seg 2", st2), ("segment 2", seg2), ("staging @ seg 3", st3), ("segment 3", seg3), ("barrier wait", bar)):
    row(n, a)
print(f"  matrix-pipe cycles of one wave's segment: {mf}; of its block: {4 * mf} = {100.0 * 4 * mf / blk.mean():.1f} % of the block time (two waves share a pipe: 50 % = saturated)")
# skew between the wavefronts of a workgroup at the barrier
