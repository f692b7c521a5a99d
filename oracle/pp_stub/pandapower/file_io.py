"""``pp.from_pickle(path)`` stand-in (call site voltage_control_env.py:403-404): the real ``model.p`` is an
external download; the tables are rebuilt from ``netspec.npz`` in the same directory, with pandapower's
table / column names, dtypes and index conventions for every column the reference (or runpp) reads."""
import os

import numpy as np
import pandas as pd

from .auxiliary import pandapowerNet


def _zone_name(z):
    return "main" if int(z) == 0 else f"zone{int(z)}"


def net_from_netspec(ns):
    net = pandapowerNet()
    net["name"] = ns.name
    net["sn_mva"] = float(ns.sn_mva)
    net["f_hz"] = float(ns.f_hz)
    nb = ns.n_bus
    net["bus"] = pd.DataFrame({"name": [f"bus{i}" for i in range(nb)], "vn_kv": ns.bus_vn_kv.astype(np.float64),
                               "type": "b", "zone": [_zone_name(z) for z in ns.bus_zone], "in_service": True})
    net["line"] = pd.DataFrame({
        "from_bus": ns.line_from_bus.astype(np.int64), "to_bus": ns.line_to_bus.astype(np.int64),
        "length_km": ns.line_length_km, "r_ohm_per_km": ns.line_r_ohm_per_km, "x_ohm_per_km": ns.line_x_ohm_per_km,
        "c_nf_per_km": ns.line_c_nf_per_km, "g_us_per_km": ns.line_g_us_per_km, "max_i_ka": 1.0, "df": 1.0,
        "parallel": ns.line_parallel.astype(np.int64), "type": "ol", "in_service": ns.line_in_service.astype(bool)})
    net["load"] = pd.DataFrame({"name": None, "bus": ns.load_bus.astype(np.int64), "p_mw": 0.0, "q_mvar": 0.0,
                                "const_z_percent": 0.0, "const_i_percent": 0.0, "sn_mva": np.nan, "scaling": 1.0,
                                "in_service": True, "type": "wye"})
    net["sgen"] = pd.DataFrame({"name": [_zone_name(z) for z in ns.sgen_zone], "bus": ns.sgen_bus.astype(np.int64),
                                "p_mw": 0.0, "q_mvar": 0.0, "sn_mva": np.nan, "scaling": 1.0, "in_service": True,
                                "type": "PV", "current_source": True})
    net["ext_grid"] = pd.DataFrame({"name": [None], "bus": [int(ns.ext_grid_bus)], "vm_pu": [float(ns.ext_grid_vm_pu)],
                                    "va_degree": [0.0], "in_service": [True]})
    net["shunt"] = pd.DataFrame({"bus": ns.shunt_bus.astype(np.int64), "p_mw": ns.shunt_p_mw, "q_mvar": ns.shunt_q_mvar,
                                 "vn_kv": ns.bus_vn_kv[ns.shunt_bus] if ns.shunt_bus.shape[0] else np.zeros(0),
                                 "step": 1, "max_step": 1, "in_service": True})
    # generic per-unit pi branches ride along untouched (pandapower would hold them as trafo / impedance rows)
    net["_branch_pu"] = {k: getattr(ns, k).copy() for k in
                         ("br_from_bus", "br_to_bus", "br_r_pu", "br_x_pu", "br_b_pu", "br_ratio", "br_shift_deg")}
    net["trafo"] = pd.DataFrame()
    for t in ("res_bus", "res_line", "res_sgen", "res_load", "res_ext_grid"):
        net[t] = pd.DataFrame()
    net["converged"] = False
    return net


def from_pickle(filename, convert=True):
    here = os.path.dirname(os.path.abspath(filename))
    npz = os.path.join(here, "netspec.npz")
    if not os.path.exists(npz):
        raise FileNotFoundError(f"pandapower stub: {npz} not found (model.p itself is never read)")
    from mapdn_amd.data import load_netspec
    return net_from_netspec(load_netspec(npz))
