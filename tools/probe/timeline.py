#!/usr/bin/env python3
"""Kernel timeline of the last step in a rocprofv3 rocpd database: start (us), duration, queue, gap to the previous kernel on that queue,
grid (workgroups), short kernel name.  python tools/probe/timeline.py <p_results.db> [steps]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = db.execute("select name,start,end,queue_id,grid_x,workgroup_x from kernels order by start").fetchall()
rs = rows[-len(rows) // steps:]
t0 = rs[0][1]


def short(s):
    s = s.replace("void ", "").replace("nc::", "")
    m = re.match(r"([\w:]+)(<[^>]*>)?", s)
    return (m.group(1) + (m.group(2) or ""))[:70] if m else s[:70]


prev = {}
busy = {}
for n, st, en, q, gx, wx in rs:
    gap = (st - prev.get(q, st)) / 1e3
    prev[q] = en
    busy[q] = busy.get(q, 0) + (en - st)
    print(f"{(st-t0)/1e3:9.1f} +{(en-st)/1e3:8.1f}us q{q} gap{gap:7.1f} g{gx//max(wx,1):6d} {short(n)}")
print("span ms", (rs[-1][2] - t0) / 1e6, {q: round(v / 1e6, 3) for q, v in busy.items()})
