import numpy as np, sys
b = np.load(sys.argv[1])
rows = []
for w in range(len(b)):
    r = b[w]; bid, hw, wave = int(r[0]), int(r[1]), int(r[2]); st = r[3:].astype(np.int64); n = int((st > 0).sum())
    rows.append((hw & 15, (hw >> 4) & 3, int(st[0]), bid, wave, n, st[:n]))
rows.sort()
for slot in (0, 1):
    seq = [x for x in rows if x[0] == slot and x[1] == 0]
    for x in seq[2:4]:
        st = x[6][1:-2]
        steps = len(st) // 10
        a = st[:steps * 10].reshape(steps, 10)
        stage = [(a[:, 2*s+1] - a[:, 2*s]).mean() for s in range(4)]
        mfma = [(a[:, 2*s+2] - a[:, 2*s+1]).mean() for s in range(4)]
        bar = (a[:, 9] - a[:, 8]).mean()
        tot = (a[1:, 0] - a[:-1, 0]).mean()
        print(f"slot{slot} bid{x[3]} steps{steps} stage {[int(v) for v in stage]} mfma {[int(v) for v in mfma]} barrier {int(bar)} step {int(tot)} stage_frac {sum(stage)/tot:.2f}")
