#!/usr/bin/env python
"""print the DESIGN.md results table from a directory of bench lines (tools/r03_profiles.sh output)"""
import glob, json, os, sys
d = sys.argv[1]
rows = []
for f in sorted(glob.glob(os.path.join(d, "bench_*.json"))):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        continue
    r = j["roofline"]; t = r.get("traffic_detail") or {}
    name = os.path.basename(f)[6:-5]
    rows.append((name, j["value"] / 1e6, j["ms_per_step"] * 1e3, r["kernel_avg_ms"] * 1e3, r["frac"], (r["traffic"] or 0) / 1e6,
                 r["algorithmic_bytes_per_launch"] / 1e6, j["nr_iterations"]["mean"], j["nr_iterations"]["max"], j.get("repeats"),
                 j["ms_per_step_repeats"]["min"] * 1e3, j["ms_per_step_repeats"]["max"] * 1e3))
print("| config | env-steps/s | per batch step (median; min–max of the repeats) | NR kernel | roofline.frac (HBM) | traffic / algorithmic per launch | mean / max NR its |")
print("|---|---|---|---|---|---|---|")
for n, v, ms, nr, fr, tr, al, im, ix, rep, mn, mx in rows:
    print(f"| {n} | {v:.2f} M | {ms:.1f} µs ({mn:.1f}–{mx:.1f}, {rep} blocks) | {nr:.1f} µs | {fr:.4f} | {tr:.0f} / {al:.1f} MB | {im:.2f} / {ix} |")
