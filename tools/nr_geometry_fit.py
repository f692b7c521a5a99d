#!/usr/bin/env python
"""Least-squares fit of the k_nr_tree launch-time model used by the geometry chooser (capi.hip::nr_model_ns) to the launch
times measured in rounds 2 and 3 (profiles/r02_nr_geometry_case141.txt, profiles/r03_geometry_case322.txt,
profiles/r03_final_*_kernel_stats.txt).  Prints the parameters and measured vs modelled time per point.  CPU only."""
import numpy as np
from scipy.optimize import least_squares

NCU = 256
# (n, W, L, rows, h_lds, workgroups, resident per CU, measured us, era)   era: 3 = round-3 final kernel, 2.5 = mid round 3, 2 = round 2
D = [
    (32, 1, 16, 11, 1, 256, 2, 54.3, 3), (32, 1, 16, 11, 1, 256, 2, 58.8, 2), (32, 2, 16, 11, 1, 256, 2, 67.6, 2),
    (140, 4, 16, 17, 1, 256, 1, 78.0, 3), (140, 4, 16, 17, 1, 256, 1, 92.9, 2), (140, 2, 16, 20, 1, 256, 1, 100.9, 2),
    (140, 1, 16, 36, 1, 256, 1, 149.3, 2), (140, 1, 8, 20, 0, 512, 5, 98.8, 2), (140, 2, 16, 20, 1, 512, 1, 190.5, 2),
    (140, 2, 16, 20, 0, 512, 2, 125.8, 2), (140, 2, 16, 20, 0, 512, 2, 116.0, 3), (140, 1, 8, 20, 0, 1024, 5, 120.2, 2),
    (140, 2, 16, 20, 0, 1024, 2, 245.6, 2), (140, 1, 8, 20, 0, 2048, 5, 245.7, 2),
    (321, 4, 8, 17, 1, 128, 1, 78.0, 3), (321, 4, 8, 17, 1, 128, 1, 84.1, 2.5), (321, 4, 16, 23, 0, 64, 1, 115.6, 2.5),
    (321, 2, 16, 42, 0, 64, 1, 175.0, 2.5), (321, 4, 8, 17, 1, 512, 1, 161.6, 2.5), (321, 4, 16, 23, 0, 256, 1, 138.8, 2.5),
    (321, 4, 16, 23, 0, 256, 1, 121.0, 3), (321, 2, 16, 42, 0, 256, 1, 191.3, 2.5), (321, 4, 16, 23, 0, 512, 1, 259.0, 3),
    (140, 4, 16, 25, 1, 256, 1, 99.4, 3), (140, 2, 8, 17, 1, 512, 2, 88.9, 3),
]


def model(x, row):
    c0, b, r1, r2, r4, p, q, s, e2, e25 = x
    n, W, L, R, h, wgs, res, _, era = row
    Wt = W * 64 / L
    r = {1: r1, 2: r2, 4: r4}[W]
    rounds = np.ceil(wgs / (NCU * res))
    fill = wgs / NCU
    conc = min(res, max(1.0, fill))
    base = c0 + b * n / Wt + r * R * (1 + p * (1 - h))
    load = 1 + q * (1 - h) * min(1.0, fill)
    share = 1 + s * max(0.0, conc * W / 4 - 1)
    return rounds * base * load * share * {3: 1.0, 2: e2, 2.5: e25}[era]


o = least_squares(lambda x: [(model(x, r) - r[7]) / r[7] for r in D], [12, 2, 2, 2.5, 2.7, 0.1, 0.1, 0.5, 1.15, 1.08],
                  bounds=([0, 0, 0.5, 0.5, 0.5, 0, 0, 0.5, 1.0, 1.0], [40, 6, 6, 6, 6, 1, 1, 0.5001, 1.4, 1.3]))
print("c0 b r1 r2 r4 p q s e2 e2.5 =", np.round(o.x, 3))
for r in D:
    print(r[:7], "measured", r[7], "model", round(model(o.x, r), 1), "era", r[8])
