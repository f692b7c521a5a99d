"""TEST INFRASTRUCTURE — the restated `pp.runpp` (oracle/pp_restated.py) in BATCHED numpy form: one Newton-Raphson power
flow for many envs of the same radial feeder at once, every array carrying an env axis.

Used by bench.py's `cpu_baseline` leg only ("its batched-numpy form across all cores") and
cross-checked against `runpp_restated` in tests/test_oracle.py; the product never imports it.

Same algorithm as pypower's newtonpf (reference call site voltage_control_env.py:557; flat start, polar Newton steps with
the exact Jacobian, ||F||inf < tolerance_mva / sn_mva, <= 10 iterations, per-env stop), except that the linear solve
`dx = -spsolve(J, F)` — SuperLU on one env's sparse Jacobian — is replaced by the block-2x2 elimination of the radial
feeder's tree (leaf -> root, then root -> leaf), which has no fill and vectorises over the env axis.  The Newton iterates are
the same up to rounding (checked: |dV| <= 1e-12, identical iteration counts).  Unknowns are [d theta, d|V| / |V|] per bus.
"""
from __future__ import annotations

import numpy as np

from .pp_restated import MAX_ITER, TOLERANCE_MVA, PPResult, build_branches, make_ybus


class BatchedRunpp:
    def __init__(self, net):
        self.net = net
        nb = net.n_bus
        ybus, yf, yt = make_ybus(net)
        self.ybus, self.yf, self.yt = ybus.tocsr(), yf.tocsr(), yt.tocsr()
        f, t, _, _, _, _, is_line = build_branches(net)
        self.f, self.t, self.is_line = f, t, is_line
        Y = self.ybus.toarray()
        adj = [[] for _ in range(nb)]
        for a, b in zip(f, t):
            if b not in adj[a]:
                adj[a].append(int(b)); adj[b].append(int(a))
        if sum(len(x) for x in adj) // 2 != nb - 1:
            raise ValueError("BatchedRunpp: radial feeders only")
        slack = int(net.ext_grid_bus)
        parent = np.full(nb, -1)
        order = [slack]
        for u in order:                                    # BFS from the slack
            for w in adj[u]:
                if w != slack and parent[w] < 0:
                    parent[w] = u; order.append(w)
        if len(order) != nb:
            raise ValueError("BatchedRunpp: network not connected")
        self.slack, self.parent = slack, parent
        self.elim = order[:0:-1]                           # children before parents, slack excluded
        self.ykk = np.array([Y[k, k] for k in range(nb)])
        self.ykp = np.array([Y[k, parent[k]] if parent[k] >= 0 else 0.0 for k in range(nb)])
        self.ypk = np.array([Y[parent[k], k] if parent[k] >= 0 else 0.0 for k in range(nb)])
        self.tol = TOLERANCE_MVA / net.sn_mva

    # bus_demand / make_sbus of pp_restated, with an env axis
    def _demand(self, pl, ql, pv, qs):
        net = self.net
        B = pl.shape[0]
        pd_ = np.zeros((B, net.n_bus)); qd = np.zeros((B, net.n_bus))
        for arr, src, idx, sc, sgn in ((pd_, pl, net.load_bus, net.load_scaling, 1.0), (qd, ql, net.load_bus, net.load_scaling, 1.0),
                                       (pd_, pv, net.sgen_bus, net.sgen_scaling, -1.0), (qd, qs, net.sgen_bus, net.sgen_scaling, -1.0)):
            for j, b in enumerate(idx):
                arr[:, b] += sgn * src[:, j] * sc[j]
        return pd_, qd

    def __call__(self, p_load, q_load, p_sgen, q_sgen):
        net = self.net
        pl, ql, pv, qs = (np.atleast_2d(np.asarray(x, dtype=np.float64)) for x in (p_load, q_load, p_sgen, q_sgen))
        B, nb = pl.shape[0], net.n_bus
        pd_, qd = self._demand(pl, ql, pv, qs)
        sbus = -(pd_ + 1j * qd) / net.sn_mva                                   # [B, nb]
        V, conv, it = self.newton(sbus)
        return self._results(V, conv, it, pd_, qd, sbus)

    def newton(self, sbus, V0=None, max_iter=MAX_ITER):
        """The Newton iteration alone: sbus [B, nb] complex -> (V [B, nb], converged [B], iterations [B]).  V0 = None is runpp's
        init="auto" (flat start at the slack set-point); a [B, nb] array is init="results" (tools/warm_start_study.py)."""
        net = self.net
        B, nb = sbus.shape
        V = (np.full((B, nb), net.ext_grid_vm_pu, dtype=np.complex128) if V0 is None     # init="auto": flat start at the slack set-point
             else np.array(V0, dtype=np.complex128))
        nonslack = np.array([k for k in range(nb) if k != self.slack])
        it = np.zeros(B, np.int64)
        conv = np.zeros(B, bool)
        active = np.ones(B, bool)
        par = self.parent
        MAX_ITER = max_iter                                                    # (shadows the module constant inside the loop below)
        for sweep in range(MAX_ITER + 1):
            S = V * np.conj((self.ybus @ V.T).T)                               # [B, nb]
            F = S - sbus
            Fn = np.maximum(np.abs(F.real[:, nonslack]).max(1), np.abs(F.imag[:, nonslack]).max(1))
            conv = np.where(active, Fn < self.tol, conv)
            active = active & ~conv & (it < MAX_ITER)
            if not active.any():
                break
            # ---- J z = F by block elimination along the tree (all envs at once; converged envs are masked at the update)
            D = np.empty((nb, 4, B)); r = np.empty((nb, 2, B)); G = np.empty((nb, 4, B)); h = np.empty((nb, 2, B))
            akk = (np.abs(V) ** 2 * np.conj(self.ykk)[None, :]).T               # [nb, B]  A_kk = |V_k|^2 conj(Y_kk)
            St = S.T
            D[:, 0] = -(St.imag - akk.imag); D[:, 1] = St.real + akk.real
            D[:, 2] = St.real - akk.real; D[:, 3] = St.imag + akk.imag
            r[:, 0] = F.real.T; r[:, 1] = F.imag.T
            Vp = V[:, np.where(par >= 0, par, 0)]
            akp = (V * np.conj(self.ykp[None, :] * Vp)).T                      # A_kp = V_k conj(Y_kp V_p)
            apk = (Vp * np.conj(self.ypk[None, :] * V)).T                      # A_pk = V_p conj(Y_pk V_k)
            for k in self.elim:
                d0, d1, d2, d3 = D[k]
                idet = 1.0 / (d0 * d3 - d1 * d2)
                i0, i1, i2, i3 = d3 * idet, -d1 * idet, -d2 * idet, d0 * idet
                h0 = i0 * r[k, 0] + i1 * r[k, 1]; h1 = i2 * r[k, 0] + i3 * r[k, 1]
                h[k, 0], h[k, 1] = h0, h1
                p = par[k]
                if p == self.slack:
                    G[k] = 0.0
                    continue
                ar, ai = akp[k].real, akp[k].imag                              # U = [[Im A_kp, Re A_kp], [-Re A_kp, Im A_kp]]
                g0 = i0 * ai - i1 * ar; g1 = i0 * ar + i1 * ai; g2 = i2 * ai - i3 * ar; g3 = i2 * ar + i3 * ai
                G[k, 0], G[k, 1], G[k, 2], G[k, 3] = g0, g1, g2, g3
                br, bi = apk[k].real, apk[k].imag                              # L likewise from A_pk
                D[p, 0] -= bi * g0 + br * g2; D[p, 1] -= bi * g1 + br * g3
                D[p, 2] -= bi * g2 - br * g0; D[p, 3] -= bi * g3 - br * g1
                r[p, 0] -= bi * h0 + br * h1; r[p, 1] -= bi * h1 - br * h0
            x = np.zeros((nb, 2, B))
            for k in reversed(self.elim):
                p = par[k]
                x[k, 0] = h[k, 0] - (G[k, 0] * x[p, 0] + G[k, 1] * x[p, 1])
                x[k, 1] = h[k, 1] - (G[k, 2] * x[p, 0] + G[k, 3] * x[p, 1])
            # newtonpf: Va += dx_a, Vm += dx_m, V = Vm e^{jVa} with dx_a = -x0, dx_m = -|V| x1
            va = np.angle(V) - x[:, 0].T
            vm = np.abs(V) * (1.0 - x[:, 1].T)
            Vn = vm * np.exp(1j * va)
            V = np.where(active[:, None], Vn, V)
            it = it + active
        return V, conv, it

    def _results(self, V, conv, it, pd_, qd, sbus):
        # ---- pfsoln + result tables (pp_restated.runpp_restated, batched)
        net = self.net
        B = V.shape[0]
        vm = np.abs(V); va_deg = np.angle(V) * 180.0 / np.pi
        s_inj = V * np.conj((self.ybus @ V.T).T) * net.sn_mva
        p_bus, q_bus = pd_.copy(), qd.copy()
        p_bus[:, self.slack] = -s_inj[:, self.slack].real
        q_bus[:, self.slack] = -s_inj[:, self.slack].imag
        for j, b in enumerate(net.shunt_bus):
            p_bus[:, b] += net.shunt_p_mw[j] * vm[:, b] ** 2
            q_bus[:, b] += net.shunt_q_mvar[j] * vm[:, b] ** 2
        sf = V[:, self.f] * np.conj((self.yf @ V.T).T) * net.sn_mva
        st = V[:, self.t] * np.conj((self.yt @ V.T).T) * net.sn_mva
        pl_full = np.zeros((B, net.n_line))
        pl_full[:, net.line_in_service.astype(bool)] = (sf.real + st.real)[:, self.is_line]
        return [PPResult(vm_pu=vm[e], va_degree=va_deg[e], p_mw=p_bus[e], q_mvar=q_bus[e], pl_mw=pl_full[e],
                         converged=bool(conv[e]), iterations=int(it[e]), V=V[e], Sbus=sbus[e]) for e in range(B)]
