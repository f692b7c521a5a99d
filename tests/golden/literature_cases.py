"""Known answers that do NOT pass through oracle/pp_restated.py — public literature data and circuit theory in SI units.

None of this comes from the reference repository (which ships neither networks nor tests).  Three kinds of pins for the
`pp.runpp` half of the hot path (reference call site voltage_control_env.py:557):

1. Baran & Wu's 33-bus feeder (IEEE Trans. Power Delivery 4(2), 1989) beyond its base case: the textbook optimal
   reconfiguration (sectionalising switches 7, 9, 14, 32 and tie 37 open, ties 33-36 closed: 139.55 kW, V_min 0.9378 p.u.
   at bus 32) — a second RADIAL topology on the same data — and the weakly MESHED operation with all five ties closed
   (123.29 kW), which exercises the general-topology solvers.
2. Baran & Wu's 69-bus feeder (IEEE Trans. Power Delivery 4(1), 1989; the PG&E 12.66 kV system).  The branch / load
   table below is restated from the literature; two load totals circulate (3 802.19 kW / 2 694.60 kVAr and
   3 801.89 kW / 2 694.10 kVAr) — this table sums to the second one.  Published results for it: losses 224.9-225.0 kW and
   102.1-102.2 kVAr, V_min 0.9092 p.u. at bus 65.  A wrong datum in the restated table would show up as a miss of those
   figures; the tolerances below are the spread of the published values, not solver accuracy.
3. Two- and three-bus nets with line charging, conductance, parallel circuits, a shunt element and a non-unit slack
   set-point, solved in CLOSED FORM from the pi-circuit in ohms / siemens / kV / MVA (Thevenin reduction + the quadratic of
   the two-bus load-flow problem).  They pin pandapower's per-unit rules (build_branch._calc_line_parameter:
   c_nf_per_km, g_us_per_km, parallel, length; shunt sign) without using them.
"""
import numpy as np

from mapdn_amd.netspec import NetSpec, Profiles, _BW33_TIES, add_lines, case33bw_base

# ---------------------------------------------------------------------------------------------- 33-bus variants
BW33_PUBLISHED = {
    # name: (loss kW, V_min p.u., bus of V_min (1-based))
    "base": (202.68, 0.9131, 18),
    "reconfigured": (139.55, 0.9378, 32),
    "meshed": (123.29, None, None),
}


def bw33_variant(kind):
    """(NetSpec, p_load_mw, q_load_mvar) of the Baran-Wu feeder: 'base' | 'reconfigured' | 'meshed'"""
    net, p, q = case33bw_base()
    if kind == "base":
        return net, p, q
    t = _BW33_TIES
    m = add_lines(net, t[:, 0].astype(np.int32) - 1, t[:, 1].astype(np.int32) - 1, t[:, 2], t[:, 3])   # switches 33..37
    if kind == "reconfigured":
        ins = np.ones(37, np.uint8)
        for sw in (7, 9, 14, 32, 37):
            ins[sw - 1] = 0
        m.line_in_service = ins
    m.name = f"bw33_{kind}"
    return m, p, q


# ---------------------------------------------------------------------------------------------- 69-bus feeder
# from, to, r [ohm], x [ohm]
_BW69_BR = np.array([
    [1, 2, 0.0005, 0.0012], [2, 3, 0.0005, 0.0012], [3, 4, 0.0015, 0.0036], [4, 5, 0.0251, 0.0294], [5, 6, 0.3660, 0.1864],
    [6, 7, 0.3811, 0.1941], [7, 8, 0.0922, 0.0470], [8, 9, 0.0493, 0.0251], [9, 10, 0.8190, 0.2707], [10, 11, 0.1872, 0.0619],
    [11, 12, 0.7114, 0.2351], [12, 13, 1.0300, 0.3400], [13, 14, 1.0440, 0.3450], [14, 15, 1.0580, 0.3496], [15, 16, 0.1966, 0.0650],
    [16, 17, 0.3744, 0.1238], [17, 18, 0.0047, 0.0016], [18, 19, 0.3276, 0.1083], [19, 20, 0.2106, 0.0690], [20, 21, 0.3416, 0.1129],
    [21, 22, 0.0140, 0.0046], [22, 23, 0.1591, 0.0526], [23, 24, 0.3463, 0.1145], [24, 25, 0.7488, 0.2475], [25, 26, 0.3089, 0.1021],
    [26, 27, 0.1732, 0.0572], [3, 28, 0.0044, 0.0108], [28, 29, 0.0640, 0.1565], [29, 30, 0.3978, 0.1315], [30, 31, 0.0702, 0.0232],
    [31, 32, 0.3510, 0.1160], [32, 33, 0.8390, 0.2816], [33, 34, 1.7080, 0.5646], [34, 35, 1.4740, 0.4873], [3, 36, 0.0044, 0.0108],
    [36, 37, 0.0640, 0.1565], [37, 38, 0.1053, 0.1230], [38, 39, 0.0304, 0.0355], [39, 40, 0.0018, 0.0021], [40, 41, 0.7283, 0.8509],
    [41, 42, 0.3100, 0.3623], [42, 43, 0.0410, 0.0478], [43, 44, 0.0092, 0.0116], [44, 45, 0.1089, 0.1373], [45, 46, 0.0009, 0.0012],
    [4, 47, 0.0034, 0.0084], [47, 48, 0.0851, 0.2083], [48, 49, 0.2898, 0.7091], [49, 50, 0.0822, 0.2011], [8, 51, 0.0928, 0.0473],
    [51, 52, 0.3319, 0.1114], [9, 53, 0.1740, 0.0886], [53, 54, 0.2030, 0.1034], [54, 55, 0.2842, 0.1447], [55, 56, 0.2813, 0.1433],
    [56, 57, 1.5900, 0.5337], [57, 58, 0.7837, 0.2630], [58, 59, 0.3042, 0.1006], [59, 60, 0.3861, 0.1172], [60, 61, 0.5075, 0.2585],
    [61, 62, 0.0974, 0.0496], [62, 63, 0.1450, 0.0738], [63, 64, 0.7105, 0.3619], [64, 65, 1.0410, 0.5302], [11, 66, 0.2012, 0.0611],
    [66, 67, 0.0047, 0.0014], [12, 68, 0.7394, 0.2444], [68, 69, 0.0047, 0.0016],
])
# bus, P [kW], Q [kVAr]
_BW69_LD = np.array([
    [6, 2.6, 2.2], [7, 40.4, 30], [8, 75, 54], [9, 30, 22], [10, 28, 19], [11, 145, 104], [12, 145, 104], [13, 8, 5], [14, 8, 5.5],
    [16, 45.5, 30], [17, 60, 35], [18, 60, 35], [20, 1, 0.6], [21, 114, 81], [22, 5, 3.5], [24, 28, 20], [26, 14, 10], [27, 14, 10],
    [28, 26, 18.6], [29, 26, 18.6], [33, 14, 10], [34, 19.5, 14], [35, 6, 4], [36, 26, 18.55], [37, 26, 18.55], [39, 24, 17],
    [40, 24, 17], [41, 1.2, 1], [43, 6, 4.3], [45, 39.22, 26.3], [46, 39.22, 26.3], [48, 79, 56.4], [49, 384.7, 274.5],
    [50, 384.7, 274.5], [51, 40.5, 28.3], [52, 3.6, 2.7], [53, 4.35, 3.5], [54, 26.4, 19], [55, 24, 17.2], [59, 100, 72],
    [61, 1244, 888], [62, 32, 23], [64, 227, 162], [65, 59, 42], [66, 18, 13], [67, 18, 13], [68, 28, 20], [69, 28, 20],
])
BW69_PUBLISHED = dict(p_total_kw=3801.89, q_total_kvar=2694.10, loss_kw=(224.9, 225.05), loss_kvar=(102.0, 102.3),
                      v_min=0.9092, v_min_bus=65)


def bw69():
    """(NetSpec, p_load_mw[48], q_load_mvar[48]); one idle sgen on bus 61 (the product needs an agent)."""
    br, ld = _BW69_BR, _BW69_LD
    n = br.shape[0]
    zone = np.ones(69, np.int32); zone[0] = 0
    net = NetSpec(name="bw69", bus_vn_kv=np.full(69, 12.66), bus_zone=zone,
                  line_from_bus=br[:, 0].astype(np.int32) - 1, line_to_bus=br[:, 1].astype(np.int32) - 1,
                  line_r_ohm_per_km=br[:, 2], line_x_ohm_per_km=br[:, 3], line_c_nf_per_km=np.zeros(n), line_g_us_per_km=np.zeros(n),
                  line_length_km=np.ones(n), line_parallel=np.ones(n, np.int32), line_in_service=np.ones(n, np.uint8),
                  load_bus=ld[:, 0].astype(np.int32) - 1, sgen_bus=np.array([60], np.int32), sgen_zone=np.array([1], np.int32),
                  ext_grid_bus=0, ext_grid_vm_pu=1.0, sn_mva=1.0, f_hz=50.0)
    return net, ld[:, 1] * 1e-3, ld[:, 2] * 1e-3


# ---------------------------------------------------------------------------------------------- closed forms (SI units)
def _line_si(r_km, x_km, c_nf_km, g_us_km, length, parallel, f_hz):
    """series impedance [ohm] and total shunt admittance [S] of a line of `parallel` identical circuits"""
    z = (r_km + 1j * x_km) * length / parallel
    y = (g_us_km * 1e-6 + 1j * 2 * np.pi * f_hz * c_nf_km * 1e-9) * length * parallel
    return z, y


def _two_bus_quadratic(e_th, z_th, s_load):
    """V at a constant-power load S [MVA] behind a Thevenin source (E_th [kV], Z_th [ohm]): rotate so that E is real,
    v = E - Z conj(S)/conj(v)  =>  |v|^2 + W = E conj(v), W = Z conj(S) = a + jb  =>  v_i = -b/E, v_r^2 - E v_r + v_i^2 + a = 0
    (upper root = the operating solution)."""
    e, delta = abs(e_th), np.angle(e_th)
    w = z_th * np.conj(s_load)
    vi = -w.imag / e
    vr = 0.5 * (e + np.sqrt(e * e - 4.0 * (vi * vi + w.real)))
    return (vr + 1j * vi) * np.exp(1j * delta)


TWO_BUS = dict(vn_kv=20.0, vm_slack=1.02, sn_mva=5.0, f_hz=50.0,
               line=dict(r=0.31, x=0.42, c=310.0, g=4.0, length=7.5, parallel=2), p_mw=3.1, q_mvar=1.3)
THREE_BUS = dict(vn_kv=11.0, vm_slack=0.99, sn_mva=2.0, f_hz=60.0,
                 line_a=dict(r=0.21, x=0.37, c=260.0, g=0.0, length=4.0, parallel=2),
                 line_b=dict(r=0.64, x=0.31, c=180.0, g=2.5, length=6.5, parallel=1),
                 shunt2_p_mw=0.04, shunt2_q_mvar=-0.35,      # capacitor bank + small loss at bus 2 (pandapower: consumer sign at 1 p.u.)
                 p_mw=0.9, q_mvar=0.35)


def _mk_net(name, nb, vn, lines, load_bus, c, shunt=None):
    k = len(lines)
    zone = np.ones(nb, np.int32); zone[0] = 0
    kw = {}
    if shunt:
        kw = dict(shunt_bus=np.array([shunt[0]], np.int32), shunt_p_mw=np.array([shunt[1]]), shunt_q_mvar=np.array([shunt[2]]))
    return NetSpec(name=name, bus_vn_kv=np.full(nb, vn), bus_zone=zone,
                   line_from_bus=np.arange(k, dtype=np.int32), line_to_bus=np.arange(1, k + 1, dtype=np.int32),
                   line_r_ohm_per_km=[l["r"] for l in lines], line_x_ohm_per_km=[l["x"] for l in lines],
                   line_c_nf_per_km=[l["c"] for l in lines], line_g_us_per_km=[l["g"] for l in lines],
                   line_length_km=[l["length"] for l in lines], line_parallel=np.array([l["parallel"] for l in lines], np.int32),
                   line_in_service=np.ones(k, np.uint8), load_bus=np.array([load_bus], np.int32),
                   sgen_bus=np.array([nb - 1], np.int32), sgen_zone=np.array([1], np.int32),
                   ext_grid_bus=0, ext_grid_vm_pu=c["vm_slack"], sn_mva=c["sn_mva"], f_hz=c["f_hz"], **kw)


def two_bus():
    """(NetSpec, p, q, expected): expected = dict(V [p.u., complex, per bus], pl_mw [per line], p_slack_mw, q_slack_mvar)"""
    c = TWO_BUS
    L = c["line"]
    z, y = _line_si(L["r"], L["x"], L["c"], L["g"], L["length"], L["parallel"], c["f_hz"])
    e1 = c["vm_slack"] * c["vn_kv"]
    s = c["p_mw"] + 1j * c["q_mvar"]
    k = 1 + z * y / 2
    v2 = _two_bus_quadratic(e1 / k, z / k, s)
    i_f = (e1 - v2) / z + e1 * y / 2
    i_t = (v2 - e1) / z + v2 * y / 2
    s_f, s_t = e1 * np.conj(i_f), v2 * np.conj(i_t)
    exp = dict(V=np.array([e1, v2]) / c["vn_kv"], pl_mw=np.array([(s_f + s_t).real]), p_slack_mw=-s_f.real, q_slack_mvar=-s_f.imag)
    net = _mk_net("two_bus", 2, c["vn_kv"], [L], 1, c)
    return net, np.array([c["p_mw"]]), np.array([c["q_mvar"]]), exp


def three_bus():
    """chain slack - 2 - 3: constant-power load at bus 3, a shunt element at the (otherwise unloaded) bus 2"""
    c = THREE_BUS
    a, b = c["line_a"], c["line_b"]
    za, ya = _line_si(a["r"], a["x"], a["c"], a["g"], a["length"], a["parallel"], c["f_hz"])
    zb, yb = _line_si(b["r"], b["x"], b["c"], b["g"], b["length"], b["parallel"], c["f_hz"])
    e1 = c["vm_slack"] * c["vn_kv"]
    ysh2 = (c["shunt2_p_mw"] - 1j * c["shunt2_q_mvar"]) / c["vn_kv"] ** 2        # S = V^2 conj(Y): p + jq consumed at V = vn
    y2 = ya / 2 + yb / 2 + ysh2
    k2 = 1 + za * y2
    e2, z2 = e1 / k2, za / k2                                                     # Thevenin at bus 2
    k3 = 1 + (z2 + zb) * yb / 2
    e3, z3 = e2 / k3, (z2 + zb) / k3                                              # ... at bus 3
    s = c["p_mw"] + 1j * c["q_mvar"]
    v3 = _two_bus_quadratic(e3, z3, s)
    ib = np.conj(s / v3) + v3 * yb / 2                                            # series current of line b (2 -> 3)
    v2 = v3 + zb * ib
    ia_f = (e1 - v2) / za + e1 * ya / 2
    ia_t = (v2 - e1) / za + v2 * ya / 2
    ib_f = (v2 - v3) / zb + v2 * yb / 2
    ib_t = (v3 - v2) / zb + v3 * yb / 2
    pl = np.array([(e1 * np.conj(ia_f) + v2 * np.conj(ia_t)).real, (v2 * np.conj(ib_f) + v3 * np.conj(ib_t)).real])
    s_f = e1 * np.conj(ia_f)
    exp = dict(V=np.array([e1, v2, v3]) / c["vn_kv"], pl_mw=pl, p_slack_mw=-s_f.real, q_slack_mvar=-s_f.imag,
               p_bus2_mw=c["shunt2_p_mw"] * abs(v2 / c["vn_kv"]) ** 2, q_bus2_mvar=c["shunt2_q_mvar"] * abs(v2 / c["vn_kv"]) ** 2)
    net = _mk_net("three_bus", 3, c["vn_kv"], [a, b], 2, c, shunt=(1, c["shunt2_p_mw"], c["shunt2_q_mvar"]))
    return net, np.array([c["p_mw"]]), np.array([c["q_mvar"]]), exp


def flat_profiles(net, p, q, T=1200):
    """profile tables that hold the given loads in every row (the product needs tables to construct an env)"""
    return Profiles(pv=np.zeros((T, net.n_sgen)), load_p=np.tile(p, (T, 1)), load_q=np.tile(q, (T, 1)), time_delta_min=3)
