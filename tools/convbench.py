#!/usr/bin/env python3
"""Per-layer timing of the DAC-44.1 kHz convolution shapes (B=32, 1 s clips) through nc_op_conv1d_bench.

    python tools/convbench.py [--iters 5] [--filter k7]
Prints one line per distinct layer shape: launches per encode+decode step, avg ms, TFLOP/s, algorithmic GB/s.
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralcodecs_amd import _lib  # noqa: E402

# (name, count per step, Cin, Cout, K, stride, pad, dil, Tin, transposed, fuse)   fuse: 1 snake-in, 2 snake-out, 4 residual
def dac44k_layers(B=32):
    L = []
    T = 44544
    L.append(("enc.stem", 1, 1, 64, 7, 1, 3, 1, T, 0, 0))
    c = 64
    for s in (2, 4, 8, 8):
        for d in (1, 3, 9):
            L.append((f"enc.k7 C{c} d{d}", 1, c, c, 7, 1, 3 * d, d, T, 0, 3))
        L.append((f"enc.k1 C{c}", 3, c, c, 1, 1, 0, 1, T, 0, 4))
        L.append((f"enc.down C{c} s{s}", 1, c, 2 * c, 2 * s, s, (s + 1) // 2, 1, T, 0, 1))
        c *= 2
        T //= s
    L.append(("enc.out k3", 1, 1024, 1024, 3, 1, 1, 1, T, 0, 1))
    L.append(("rvq.in_proj", 9, 1024, 8, 1, 1, 0, 1, T, 0, 0))
    L.append(("rvq.out_proj", 9, 8, 1024, 1, 1, 0, 1, T, 0, 0))
    L.append(("dec.in k7", 1, 1024, 1536, 7, 1, 3, 1, T, 0, 0))
    c = 1536
    for s in (8, 8, 4, 2):
        L.append((f"dec.up C{c} s{s}", 1, c, c // 2, 2 * s, s, (s + 1) // 2, 1, T, 1, 1))
        c //= 2
        T *= s
        for d in (1, 3, 9):
            L.append((f"dec.k7 C{c} d{d}", 1, c, c, 7, 1, 3 * d, d, T, 0, 3))
        L.append((f"dec.k1 C{c}", 3, c, c, 1, 1, 0, 1, T, 0, 4))
    L.append(("dec.head", 1, 96, 1, 7, 1, 3, 1, T, 0, 0))   # Snake arrives fused by the producer
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--filter", default="")
    a = ap.parse_args()
    lib = _lib.lib()
    tot = 0.0
    totf = 0.0
    print(f"{'layer':22s} {'n':>2s} {'ms':>8s} {'TF/s':>7s} {'GB/s':>7s}  {'ms*n':>7s}")
    for name, n, cin, cout, k, s, p, d, T, tr, fuse in dac44k_layers(a.batch):
        if a.filter and a.filter not in name:
            continue
        desc = _lib.NcConvDesc(a.batch, cin, cout, k, s, p, d, 0, T, tr, 0)
        ms = C.c_double()
        _lib.check(lib.nc_op_conv1d_bench(0, C.byref(desc), fuse, a.iters, C.byref(ms)))
        Tout = (T - 1) * s - 2 * p + k if tr else (T + 2 * p - d * (k - 1) - 1) // s + 1
        fl = 2.0 * cin * cout * k * (T if tr else Tout) * a.batch
        by = 4.0 * a.batch * (cin * T + cout * Tout * (2 if fuse & 4 else 1)) + 4.0 * cin * cout * k
        print(f"{name:22s} {n:2d} {ms.value:8.3f} {fl/ms.value/1e9:7.1f} {by/ms.value/1e6:7.0f}  {ms.value*n:7.2f}", flush=True)
        tot += ms.value * n
        totf += fl * n
    if not a.filter or "ru" in a.filter:
        import numpy as np
        from neuralcodecs_amd import ops
        rng = np.random.default_rng(0)
        for Cc, T, B in ((64, 44544, 8), (128, 22272, 8), (96, 44544, 8), (192, 22272, 16), (256, 5568, 32)):
            x = rng.standard_normal((B, Cc, T)).astype(np.float32)
            w7 = (rng.standard_normal((Cc, Cc, 7)) / np.sqrt(Cc * 7)).astype(np.float32); b7 = np.zeros(Cc, np.float32)
            w1 = (rng.standard_normal((Cc, Cc, 1)) / np.sqrt(Cc)).astype(np.float32); b1 = np.zeros(Cc, np.float32)
            al = np.full(Cc, 1.3, np.float32)
            fl = 2.0 * Cc * Cc * 8 * T * B
            for fused in (False, True):
                _, ms = ops.res_unit(x, w7, b7, al, al, w1, b1, dil=3, fused=fused, iters=a.iters)
                print(f"res_unit C{Cc} B{B} {'fused  ' if fused else '2-launch'} {ms:8.3f} ms {fl/ms/1e9:7.1f} TF/s", flush=True)
    if tot > 0:
        print(f"sum over one encode+decode step: {tot:.2f} ms, {totf/tot/1e9:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
