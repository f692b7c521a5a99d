"""``pp.runpp(net)`` stand-in: tables -> oracle/pp_restated.runpp_restated -> result tables.

Result tables follow pandapower/results_bus.py / results_branch.py / results_gen.py: res_bus has exactly the
float64 columns vm_pu, va_degree, p_mw, q_mvar indexed like net.bus (all-float => `.loc[label]` rows are
views, which is what makes the reference's chained `+=` at voltage_control_env.py:239-244 effective);
res_line carries pl_mw (+ the other flow columns as NaN placeholders are NOT invented: only pl_mw, ql_mvar-free);
res_sgen = (p_mw, q_mvar) * scaling of in-service sgens.  Not converged -> LoadflowNotConverged."""
import numpy as np
import pandas as pd

from .powerflow import LoadflowNotConverged


def _netspec_of(net):
    from mapdn_amd.netspec import NetSpec

    def zid(z):
        return 0 if z == "main" else int(str(z).replace("zone", ""))
    bus = net.bus.sort_index()
    line = net.line.sort_index()
    br = net["_branch_pu"]
    return NetSpec(
        name=str(net.get("name", "net")), bus_vn_kv=bus["vn_kv"].to_numpy(), bus_zone=np.array([zid(z) for z in bus["zone"]]),
        line_from_bus=line["from_bus"].to_numpy(), line_to_bus=line["to_bus"].to_numpy(),
        line_r_ohm_per_km=line["r_ohm_per_km"].to_numpy(), line_x_ohm_per_km=line["x_ohm_per_km"].to_numpy(),
        line_c_nf_per_km=line["c_nf_per_km"].to_numpy(), line_g_us_per_km=line["g_us_per_km"].to_numpy(),
        line_length_km=line["length_km"].to_numpy(), line_parallel=line["parallel"].to_numpy(),
        line_in_service=line["in_service"].to_numpy().astype(np.uint8),
        load_bus=net.load["bus"].to_numpy(), sgen_bus=net.sgen["bus"].to_numpy(),
        sgen_zone=np.array([zid(z) for z in net.sgen["name"]]),
        ext_grid_bus=int(net.ext_grid["bus"].iloc[0]), ext_grid_vm_pu=float(net.ext_grid["vm_pu"].iloc[0]),
        sn_mva=float(net.sn_mva), f_hz=float(net.f_hz),
        shunt_bus=net.shunt["bus"].to_numpy(), shunt_p_mw=net.shunt["p_mw"].to_numpy(), shunt_q_mvar=net.shunt["q_mvar"].to_numpy(),
        **br)


def runpp(net, **kwargs):
    if kwargs:
        raise NotImplementedError("pandapower stub: runpp is only called with defaults by the reference (:124,165,557)")
    from oracle.pp_restated import runpp_restated
    ns = net.get("_netspec_cache")
    if ns is None:                      # topology columns never change inside MAPDN: convert once per net object family
        ns = _netspec_of(net)
        net["_netspec_cache"] = ns
    ld, sg = net.load, net.sgen
    on_l = ld["in_service"].to_numpy(bool) * ld["scaling"].to_numpy(np.float64)
    on_s = sg["in_service"].to_numpy(bool) * sg["scaling"].to_numpy(np.float64)
    p_load = ld["p_mw"].to_numpy(np.float64) * on_l
    q_load = ld["q_mvar"].to_numpy(np.float64) * on_l
    p_sgen = sg["p_mw"].to_numpy(np.float64) * on_s
    q_sgen = sg["q_mvar"].to_numpy(np.float64) * on_s
    res = runpp_restated(ns, p_load, q_load, p_sgen, q_sgen)
    net["converged"] = bool(res.converged)
    net["_iterations"] = int(res.iterations)
    if not res.converged:
        raise LoadflowNotConverged("Power Flow nr did not converge after 10 iterations!")
    net["res_bus"] = pd.DataFrame({"vm_pu": res.vm_pu, "va_degree": res.va_degree, "p_mw": res.p_mw, "q_mvar": res.q_mvar},
                                  index=net.bus.sort_index().index)
    net["res_line"] = pd.DataFrame({"pl_mw": res.pl_mw}, index=net.line.sort_index().index)
    net["res_sgen"] = pd.DataFrame({"p_mw": p_sgen, "q_mvar": q_sgen}, index=sg.index)
    net["res_load"] = pd.DataFrame({"p_mw": p_load, "q_mvar": q_load}, index=ld.index)
    ref = int(net.ext_grid["bus"].iloc[0])
    net["res_ext_grid"] = pd.DataFrame({"p_mw": [-res.p_mw[ref]], "q_mvar": [-res.q_mvar[ref]]}, index=net.ext_grid.index)
