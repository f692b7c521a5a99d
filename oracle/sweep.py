"""TEST INFRASTRUCTURE — independent second solver for radial feeders (backward/forward sweep).

Shares no code with oracle/pp_restated.py's Newton-Raphson and no code with the HIP kernels: it is
the classical current-summation ladder iteration (Shirmohammadi et al. 1988) on the branch pi-model.
Used to guard against a bug common to the restated NR and the GPU NR (SURVEY.md 8(c) pin (ii)).
"""
from __future__ import annotations

import numpy as np


def sweep_solve(net, p_load, q_load, p_sgen, q_sgen, tol=1e-13, max_it=500):
    """Return complex bus voltages (p.u.).  Lines only (tap = 1), net must be a tree."""
    nb = net.n_bus
    on = net.line_in_service.astype(bool)
    f = net.line_from_bus[on].astype(int)
    t = net.line_to_bus[on].astype(int)
    assert f.shape[0] == nb - 1 and net.n_branch_pu == 0, "sweep solver needs a radial line-only net"
    zb = net.bus_vn_kv[f] ** 2 / net.sn_mva
    length, par = net.line_length_km[on], net.line_parallel[on]
    z = (net.line_r_ohm_per_km[on] + 1j * net.line_x_ohm_per_km[on]) * length / par / zb
    ysh_half = 0.5 * (net.line_g_us_per_km[on] * 1e-6 + 1j * 2 * np.pi * net.f_hz * net.line_c_nf_per_km[on] * 1e-9) \
        * zb * length * par
    # orient the tree from the slack
    adj = [[] for _ in range(nb)]
    for k in range(nb - 1):
        adj[f[k]].append((t[k], k))
        adj[t[k]].append((f[k], k))
    root = net.ext_grid_bus
    parent = np.full(nb, -1)
    pedge = np.full(nb, -1)
    order = [root]
    seen = np.zeros(nb, bool)
    seen[root] = True
    for u in order:
        for w, k in adj[u]:
            if not seen[w]:
                seen[w] = True
                parent[w], pedge[w] = u, k
                order.append(w)
    assert len(order) == nb, "net is not connected"
    # bus shunt admittance from line charging (+ explicit shunts)
    ysh = np.zeros(nb, complex)
    np.add.at(ysh, f, ysh_half)
    np.add.at(ysh, t, ysh_half)
    if net.shunt_bus.shape[0]:
        np.add.at(ysh, net.shunt_bus, (net.shunt_p_mw - 1j * net.shunt_q_mvar) / net.sn_mva)
    s = np.zeros(nb, complex)                  # consumer-sign demand, p.u.
    np.add.at(s, net.load_bus, (np.asarray(p_load) + 1j * np.asarray(q_load)) / net.sn_mva)
    np.add.at(s, net.sgen_bus, -(np.asarray(p_sgen) + 1j * np.asarray(q_sgen)) / net.sn_mva)
    v = np.full(nb, net.ext_grid_vm_pu, complex)
    for _ in range(max_it):
        inode = np.conj(s / v) + ysh * v       # current drawn at each bus
        ibr = inode.copy()                     # current in the branch feeding each bus
        for u in reversed(order[1:]):
            ibr[parent[u]] += ibr[u]
        vnew = v.copy()
        for u in order[1:]:
            vnew[u] = vnew[parent[u]] - z[pedge[u]] * ibr[u]
        err = np.abs(vnew - v).max()
        v = vnew
        if err < tol:
            break
    return v
