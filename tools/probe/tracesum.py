import numpy as np, sys
b = np.load(sys.argv[1])
rows = []
for w in range(len(b)):
    r = b[w]; bid, hw, wave = int(r[0]), int(r[1]), int(r[2]); st = r[3:].astype(np.int64); n = int((st > 0).sum())
    rows.append((hw & 15, (hw >> 4) & 3, int(st[0]), bid, wave, n, st[:n]))
rows.sort()
for slot in (0, 1):
    seq = [x for x in rows if x[0] == slot and x[1] == 0]
    prev_end = None
    for (_, simd, t0, bid, wave, n, st) in seq[1:5]:
        entry, first, last, end = st[0], st[1], st[n-2], st[n-1]
        gap = entry - prev_end if prev_end else 0
        print(f"slot{slot} bid{bid:5d} n{n} dispatch_gap {gap:7d} prologue {first-entry:7d} loop {last-first:8d} epilogue {end-last:7d}")
        prev_end = end
