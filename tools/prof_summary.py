#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats result (rocpd sqlite .db, or *_kernel_stats.csv)
into the small text table committed under profiles/."""
import glob
import sqlite3
import sys


def main(path, out=None):
    con = sqlite3.connect(path)
    rows = con.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = [f"# rocprofv3 --kernel-trace --stats summary of {path.split('/')[-1]}",
             f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}"]
    for name, n, s, a, mn, mx in rows:
        lines.append(f"{name[:70]:70s} {n:7d} {s/1e6:10.3f} {a/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*s/tot:6.2f}")
    regs = con.execute("select distinct name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, "
                       "grid_x, grid_y, workgroup_x from kernels group by name").fetchall()
    lines.append("")
    lines.append(f"{'kernel':70s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds':>6s} {'scratch':>7s} {'grid':>14s} {'wg':>4s}")
    for name, v, a, s, l, sc, gx, gy, wx in regs:
        lines.append(f"{name[:70]:70s} {v:5d} {a:5d} {s:5d} {l:6d} {sc:7d} {str(gx)+'x'+str(gy):>14s} {wx:4d}")
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
