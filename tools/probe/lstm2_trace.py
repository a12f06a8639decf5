#!/usr/bin/env python3
"""Reads an NC_LSTM2_TRACE dump (nc_lstm.hip stamps, s_memrealtime at 100 MHz) and prints where a step of the fused LSTM goes.
roles: 0,1 L0 chains | 2,3 L1 chains | 4,5 ih chain A,B | 6 L0 gate | 7 L1 gate
chain / ih stamps: 0 before poll, 1 flags seen, 2 operands valid, 3 posted      gate stamps: 0 waiting, 1 partials in, 2 gates done, 3 stored"""
import sys
import numpy as np

raw = open(sys.argv[1], "rb").read()
nwg, nt, nr, ns = np.frombuffer(raw[:16], np.int32)
body = np.frombuffer(raw[16:], np.uint64)
ck = body[-4:].astype(np.float64)
if ck[3] > ck[1]:
    print(f"shader clock between the probes: {(ck[2] - ck[0]) / ((ck[3] - ck[1]) * 0.01) / 1e3:.3f} GHz")
tr = body[:-4].reshape(nwg, nt, nr, ns, 4).astype(np.float64) * 0.01   # us
t0 = tr[tr > 0].min()
tr = np.where(tr > 0, tr - t0, np.nan)
for g in range(nt):
    print(f"tile {g}")
    for s in range(ns - 1):
        pub0 = tr[:, g, 6, s, 3]                      # L0 gate stored step s
        c = tr[:, g, 0:2, s + 1, :]                   # L0 chains, step s+1
        gt = tr[:, g, 6, s + 1, :]
        print(f" step+{s}: L0 publish [{np.nanmin(pub0):7.2f} .. {np.nanmax(pub0):7.2f}]  chain: poll-start {np.nanmean(c[..., 0]):7.2f} flags {np.nanmean(c[..., 1]):7.2f}"
              f" (max {np.nanmax(c[..., 1]):7.2f}) valid {np.nanmean(c[..., 2]):7.2f} posted {np.nanmean(c[..., 3]):7.2f} | gate: in {np.nanmean(gt[:, 1]):7.2f}"
              f" done {np.nanmean(gt[:, 2]):7.2f} stored {np.nanmean(gt[:, 3]):7.2f}")
    for s in range(ns - 1):
        pub1 = tr[:, g, 7, s, 3]
        c = tr[:, g, 2:4, s + 1, :]
        ih = tr[:, g, 4:6, s + 1, :]
        gt = tr[:, g, 7, s + 1, :]
        print(f" step+{s}: L1 publish [{np.nanmin(pub1):7.2f} .. {np.nanmax(pub1):7.2f}]  chain: flags {np.nanmean(c[..., 1]):7.2f} valid {np.nanmean(c[..., 2]):7.2f} posted {np.nanmean(c[..., 3]):7.2f}"
              f" | ihA: start {np.nanmean(ih[:, 0, 0]):7.2f} flags {np.nanmean(ih[:, 0, 1]):7.2f} valid {np.nanmean(ih[:, 0, 2]):7.2f} done {np.nanmean(ih[:, 0, 3]):7.2f}"
              f" ihB: valid {np.nanmean(ih[:, 1, 2]):7.2f} done {np.nanmean(ih[:, 1, 3]):7.2f} | gate: in {np.nanmean(gt[:, 1]):7.2f} stored {np.nanmean(gt[:, 3]):7.2f}")
