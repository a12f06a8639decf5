#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs (one counter per pass) per kernel shape.

    python tools/pmc_summary.py FETCH_SIZE=gpurun_out/pmc_fetch/p_counter_collection.csv WRITE_SIZE=... > profiles/rNN_hbm_traffic.txt

Units (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of the rocprofv3 derived metric
(TCC_EA0_RDREQ x 64 B / 1024); on gfx950 FETCH_SIZE counts 128-B read requests as 64 B for wide coalesced streams, so the read
bytes are DOUBLED before use ("corr" column).  WRITE_SIZE is uncalibrated and given as reported.
"""
import csv
import sys
from collections import OrderedDict


def load(path):
    rows = list(csv.DictReader(open(path)))
    out = OrderedDict()
    for r in rows:
        k = (r["Kernel_Name"], r["Grid_Size"], r["LDS_Block_Size"])
        out.setdefault(k, []).append(float(r["Counter_Value"]))
    return out


def short(n):
    n = n.replace("void ", "").replace("nc::", "")
    return n if len(n) <= 70 else n[:67] + "..."


def main():
    data = {}
    for a in sys.argv[1:]:
        name, path = a.split("=", 1)
        data[name] = load(path)
    keys = []
    for d in data.values():
        for k in d:
            if k not in keys:
                keys.append(k)
    print("# per (kernel, grid, LDS): mean counter value per launch; FETCH corrected x2 (gfx950), values in KiB as reported by rocprofv3")
    print(f"{'launches':>8} {'FETCH_KiB':>12} {'FETCHx2_MB':>11} {'WRITE_KiB':>12} {'WRITE_MB':>9}  key")
    rows = []
    for k in keys:
        f = data.get("FETCH_SIZE", {}).get(k, [])
        w = data.get("WRITE_SIZE", {}).get(k, [])
        fm = sum(f) / len(f) if f else float("nan")
        wm = sum(w) / len(w) if w else float("nan")
        rows.append((fm if fm == fm else 0, k, len(f) or len(w), fm, wm))
    rows.sort(reverse=True)
    for _, k, n, fm, wm in rows[:60]:
        print(f"{n:8d} {fm:12.0f} {2*fm*1024/1e6:11.1f} {wm:12.0f} {wm*1024/1e6:9.1f}  grid={int(k[1])//256} lds={k[2]} {short(k[0])}")


if __name__ == "__main__":
    main()
