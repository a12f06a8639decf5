#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (--kernel-trace [--stats]) into the text tables kept under profiles/.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db [--skip-first N] > profiles/rNN_x.kernel_stats.txt

Prints (a) the per-kernel-name table (calls, total, average, min, max, share) exactly as `--stats` defines it, and
(b) the same grouped by (kernel name, grid size, LDS bytes) so every layer shape of one template shows on its own line
with its VGPR/AGPR/SGPR/LDS allocation.  --skip-first drops the first N dispatches of every group (warm-up).
"""
import argparse
import sqlite3
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--skip-first", type=int, default=0)
    ap.add_argument("--top", type=int, default=60)
    a = ap.parse_args()
    db = sqlite3.connect(a.db)
    rows = db.execute("select name, start, end, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count "
                      "from kernels order by start").fetchall()
    if not rows:
        print("no kernel dispatches in", a.db)
        return 1

    def short(n):
        n = n.replace("void ", "").replace("nc::", "")
        return n if len(n) <= 96 else n[:93] + "..."

    def table(keyfn, title, fmtkey):
        groups = {}
        for r in rows:
            groups.setdefault(keyfn(r), []).append(r)
        out = []
        for k, rs in groups.items():
            rs = rs[a.skip_first:] if len(rs) > a.skip_first else rs
            d = [(r[2] - r[1]) for r in rs]
            out.append((sum(d), k, len(d), sum(d) / len(d), min(d), max(d), rs[0]))
        out.sort(reverse=True)
        tot = sum(o[0] for o in out)
        print(f"## {title}   (total kernel time {tot/1e6:.3f} ms over {sum(o[2] for o in out)} dispatches)")
        print(f"{'calls':>6} {'total_ms':>10} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'share%':>7}  key")
        for t, k, n, avg, mn, mx, r0 in out[:a.top]:
            print(f"{n:6d} {t/1e6:10.3f} {avg/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*t/tot:7.2f}  {fmtkey(k, r0)}")
        print()

    print(f"# rocprofv3 kernel-trace summary of {a.db}  (skip-first={a.skip_first})\n")
    table(lambda r: r[0], "per kernel name (== rocprofv3 --stats kernel table)", lambda k, r0: short(k))
    table(lambda r: (r[0], r[3], r[5]), "per (kernel, grid, LDS) = per layer shape",
          lambda k, r0: f"grid={k[1]//max(r0[4],1)}x{r0[4]} lds={k[2]} vgpr={r0[6]} agpr={r0[7]} sgpr={r0[8]}  {short(k[0])}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
