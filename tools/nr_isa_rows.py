#!/usr/bin/env python
"""Instruction counts per ROW of k_nr_tree from the gfx950 ISA: compiles one part of the instantiation list with
-save-temps, cuts the chosen instantiation's code at its s_barrier instructions (every row of a sweep ends in one when W > 1) and
prints, per segment, how many instructions of each class it holds — f64 arithmetic split into mul / add / fma(c) / rcp, 32-bit VALU
(flag and slot decoding, LDS address arithmetic), moves / selects (0/1 masks, AGPR parking), LDS, VMEM, SALU, s_waitcnt.

    python tools/nr_isa_rows.py [--part 0] [--inst 4,16,1,0,1] > profiles/r06_nr_row_instruction_counts.txt
"""
import argparse
import collections
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def classify(op):
    if op.startswith("v_rcp_f64"):
        return "f64 rcp"
    if op.startswith("v_mul_f64"):
        return "f64 mul"
    if op.startswith("v_add_f64"):
        return "f64 add"
    if op.startswith("v_fma_f64") or op.startswith("v_fmac_f64"):
        return "f64 fma"
    if "f64" in op and op.startswith("v_"):
        return "f64 other (max / cmp)"
    if op.startswith(("v_mov", "v_accvgpr", "v_cndmask", "v_pk_mov")):
        return "move / select"
    if op.startswith(("v_readlane", "v_readfirstlane")):
        return "lane -> scalar"
    if op.startswith("v_"):
        return "VALU 32-bit (decode / address)"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_barrier"):
        return "s_barrier"
    return "SALU / branch"


COLS = ["f64 mul", "f64 add", "f64 fma", "f64 rcp", "f64 other (max / cmp)", "VALU 32-bit (decode / address)", "move / select", "lane -> scalar",
        "LDS", "VMEM", "SALU / branch", "s_waitcnt"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--part", type=int, default=0)
    ap.add_argument("--inst", default="4,16,1,0,1", help="W,L,HL,GL,RES of the instantiation")
    a = ap.parse_args()
    w, l, hl, gl, res = (int(x) for x in a.inst.split(","))
    sym = f"_ZN5mapdn9k_nr_treeILi{w}ELi{l}ELb{hl}ELb{gl}ELi{res}EE"
    d = tempfile.mkdtemp(prefix="nr_isa_")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-DNR_INST_PART={a.part}", "-c",
                    os.path.join(ROOT, "mapdn_amd", "csrc", "nr_inst.hip"), "-o", os.path.join(d, "nr.o"), "-save-temps=obj"], check=True, cwd=d)
    asm = open(os.path.join(d, "nr_inst-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
    start = next(i for i, ln in enumerate(asm) if ln.startswith(sym) and ln.rstrip().endswith(("PhS2_:", "S2_: ; @" + ln.split(":")[0])) or (ln.startswith(sym) and ":" in ln))
    end = next(i for i in range(start, len(asm)) if ".end_amdhsa_kernel" in asm[i])
    segs, cur = [], []
    for ln in asm[start + 1:end]:
        t = ln.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        if op.startswith("s_endpgm"):
            break
        cur.append(op)
        if op == "s_barrier":
            segs.append(cur); cur = []
    segs.append(cur)
    print(f"# k_nr_tree<{w}, {l}, {bool(hl)}, {bool(gl)}, {res}> (gfx950, hipcc -O3): instructions between consecutive s_barrier, in code order")
    print("# (a row of a sweep ends in one barrier; the long first / last segments are the set-up, the peeled flat-start sweep, the update and the epilogue)")
    print(f"{'seg':>4s} {'total':>6s} " + " ".join(f"{c[:14]:>14s}" for c in COLS))
    for i, s in enumerate(segs):
        c = collections.Counter(classify(o) for o in s)
        print(f"{i:4d} {len(s):6d} " + " ".join(f"{c.get(k, 0):14d}" for k in COLS))


if __name__ == "__main__":
    main()
