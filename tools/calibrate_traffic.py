#!/usr/bin/env python
"""Calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on the library's own global access pattern (the x2 question).
Runs itself twice under `rocprofv3 --pmc` (separate passes) around mapdn_debug_stream copies of a known size and prints
counter bytes / known bytes for both patterns and three sizes (the largest beyond the 256 MB Infinity Cache).
    python tools/calibrate_traffic.py            -> table on stdout (commit under profiles/)"""
import csv
import glob
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CASES = [(pat, rows) for pat in (0, 1) for rows in (512, 2048, 8192)]      # x Bp 4096 x 16 B = 32 MB, 128 MB, 512 MB each way
BP = 4096


def inner():
    import torch
    from mapdn_amd import _lib
    lib = _lib.load()
    for pat, rows in CASES:
        n = rows * BP * 2
        src = torch.rand(n, dtype=torch.float64, device="cuda:0")
        dst = torch.zeros_like(src)
        torch.cuda.synchronize()
        for _ in range(3):
            assert lib.mapdn_debug_stream(src.data_ptr(), dst.data_ptr(), rows, BP, pat, torch.cuda.current_stream().cuda_stream) == 0
        torch.cuda.synchronize()
        assert torch.equal(src, dst)
        del src, dst


def one_pass(counter):
    d = tempfile.mkdtemp(prefix="mapdn_cal_")
    try:
        subprocess.run(["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "c", "--", sys.executable, os.path.abspath(__file__), "--inner"],
                       cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True, timeout=600)
        vals = []
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            rows_ = sorted(csv.DictReader(open(f)), key=lambda r: int(r.get("Dispatch_Id", 0)))
            vals += [float(r["Counter_Value"]) for r in rows_ if r["Counter_Name"] == counter and "k_calib_stream" in r["Kernel_Name"]]
        return vals
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    if "--inner" in sys.argv:
        inner()
        sys.exit(0)
    f, w = one_pass("FETCH_SIZE"), one_pass("WRITE_SIZE")
    print("# rocprofv3 FETCH_SIZE / WRITE_SIZE (KiB x 1024) against the known bytes of mapdn_debug_stream copies; 3 launches per case")
    print("# pattern 0 = k_nr_tree's: 16-byte raw-buffer loads / stores, 256 contiguous bytes per 16-lane worker; 1 = whole wave per row (1 KB)")
    print(f"{'pattern':>7s} {'MB each way':>12s} {'FETCH/known':>24s} {'WRITE/known':>24s}")
    for i, (pat, rows) in enumerate(CASES):
        known = rows * BP * 16
        fr = [f[3 * i + j] * 1024 / known for j in range(3)] if len(f) >= 3 * (i + 1) else []
        wr = [w[3 * i + j] * 1024 / known for j in range(3)] if len(w) >= 3 * (i + 1) else []
        print(f"{pat:7d} {known / 1e6:12.1f} {' '.join(f'{x:7.3f}' for x in fr):>24s} {' '.join(f'{x:7.3f}' for x in wr):>24s}")
