#!/usr/bin/env python
"""From a rocprofv3 kernel trace (rocpd .db): the step cycle — per kernel its mean duration and the mean idle gap
before it (end of the previous kernel -> start of this one), over the steady-state part of the run."""
import sqlite3
import sys
from collections import defaultdict

con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end from kernels order by start").fetchall()
rows = rows[len(rows) // 4:]                     # skip warm-up / reset-heavy start
dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
prev_end = None
for name, s, e in rows:
    n = name.split("(")[0][:60]
    dur[n] += e - s
    if prev_end is not None:
        gap[n] += max(0, s - prev_end)
    cnt[n] += 1
    prev_end = e
print(f"{'kernel':60s} {'calls':>6s} {'avg_us':>8s} {'gap_before_us':>14s}")
for n in sorted(dur, key=lambda k: -dur[k]):
    print(f"{n:60s} {cnt[n]:6d} {dur[n]/cnt[n]/1e3:8.2f} {gap[n]/cnt[n]/1e3:14.2f}")
span = rows[-1][2] - rows[0][1]
busy = sum(dur.values())
print(f"span {span/1e6:.3f} ms, busy {busy/1e6:.3f} ms ({100*busy/span:.1f} %), {len(rows)} launches")
