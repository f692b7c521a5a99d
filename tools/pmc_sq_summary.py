#!/usr/bin/env python
"""Summarise the SQ counter passes of `rocprofv3 -i tools/pmc_sq.txt --output-format csv -- python bench.py`
(one *_counter_collection.csv per pass) for the k_nr_tree dispatches into the table kept under profiles/.

usage: pmc_sq_summary.py <out.txt> <pass1.csv> [<pass2.csv> ...]"""
import collections
import csv
import sys


def main(out, files, sub="k_nr_tree"):
    vals = collections.defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            if sub in row["Kernel_Name"]:
                vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
    mean = {}
    for k, v in vals.items():
        v = [x for x in v if x > 0.25 * max(v)] if max(v) > 0 else v      # drop the early-exit reset retries
        mean[k] = sum(v) / len(v)
    waves, wc = mean.get("SQ_WAVES", 1.0), mean.get("SQ_WAVE_CYCLES", 1.0)
    lines = [f"# rocprofv3 -i tools/pmc_sq.txt (SQ counters, {len(files)} passes) of bench.py, kernel {sub}",
             "# per-launch means; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are in quad-cycles",
             f"{'counter':28s} {'per launch':>14s} {'per wave':>14s} {'frac of wave cycles':>20s}"]
    for k in sorted(mean):
        lines.append(f"{k:28s} {mean[k]:14.1f} {mean[k] / waves:14.1f} {mean[k] / wc:20.3f}")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    if sys.argv[1] == "--kernel":                  # pmc_sq_summary.py --kernel <substring> <out.txt> <pass.csv> ...
        main(sys.argv[3], sys.argv[4:], sys.argv[2])
    else:
        main(sys.argv[1], sys.argv[2:])
