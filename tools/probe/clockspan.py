import numpy as np, sys
b = np.load(sys.argv[1]); ms = float(sys.argv[2])
st = b[:, 3:].astype(np.int64)
lo = min(int(r[r > 0].min()) for r in st if (r > 0).any()); hi = max(int(r.max()) for r in st)
# two launches were traced (warm-up + timed): the span covers both, separated by a gap; use per-launch halves
first = sorted(int(r[r > 0].min()) for r in st if (r > 0).any())
print(f"stamp span {hi-lo} ticks over two launches; kernel {ms:.3f} ms each")
