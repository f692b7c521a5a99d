"""On-disk formats on either side of the hot path.

* the three CSV profile tables of a MAPDN scenario directory — same parsing rules as the reference
  loaders (voltage_control_env.py:407-438): first column = timestamp, remaining columns = one
  element each, multiplied by pv_scale / demand_scale; the sampling interval and the day count
  come from the index exactly like `_select_start_day` (:394-398);
* `NetSpec` <-> `netspec.npz` (the pandapower columns the path reads, see netspec.py);
* `from_pandapower(net)` for boxes that do have pandapower (column copying, guarded import).
"""
from __future__ import annotations

import dataclasses
import os

import numpy as np

from .netspec import NetSpec, Profiles

CSV_NAMES = ("pv_active.csv", "load_active.csv", "load_reactive.csv")


def load_profiles_csv(data_path: str, pv_scale: float = 1.0, demand_scale: float = 1.0) -> Profiles:
    import pandas as pd
    tabs = []
    for name, scale in zip(CSV_NAMES, (pv_scale, demand_scale, demand_scale)):
        df = pd.read_csv(os.path.join(data_path, name), index_col=None, float_precision="round_trip")   # :412,423,434
        idx = pd.to_datetime(df.iloc[:, 0])                                        # :413
        tabs.append((idx, df.iloc[::1, 1:].to_numpy(dtype=np.float64) * scale))    # :415
    idx = tabs[0][0]
    pv_days = (idx.iloc[-1] - idx.iloc[0]).days                                    # :395
    time_delta = (idx.iloc[1] - idx.iloc[0]).seconds // 60                         # :396
    if not all(t[1].shape[0] == tabs[0][1].shape[0] for t in tabs):
        raise ValueError("the three profile tables must have the same number of rows")
    return Profiles(pv=tabs[0][1], load_p=tabs[1][1], load_q=tabs[2][1], time_delta_min=int(time_delta), days=int(pv_days))


def save_profiles_csv(prof: Profiles, data_path: str, start="2012-01-01 00:00:00", float_format="%.17g") -> None:
    """Write tables in the reference's CSV layout (used by tests and to export synthetic scenarios).
    %.17g text round-trips through `float_precision="round_trip"` (load_profiles_csv); pandas' default parser —
    the one the reference uses (:412) — is exact only for <= 15 significant digits."""
    import pandas as pd
    os.makedirs(data_path, exist_ok=True)
    idx = pd.date_range(start=start, periods=prof.n_rows, freq=f"{prof.time_delta_min}min")
    for name, tab in zip(CSV_NAMES, (prof.pv, prof.load_p, prof.load_q)):
        df = pd.DataFrame(tab, columns=[str(i) for i in range(tab.shape[1])])
        df.insert(0, "time", idx.strftime("%Y-%m-%d %H:%M:%S"))
        df.to_csv(os.path.join(data_path, name), index=False, float_format=float_format)


def save_netspec(net: NetSpec, path: str) -> None:
    d = {}
    for f in dataclasses.fields(net):
        v = getattr(net, f.name)
        d[f.name] = np.asarray(v) if not isinstance(v, str) else np.array(v)
    np.savez_compressed(path, **d)


def load_netspec(path: str) -> NetSpec:
    z = np.load(path, allow_pickle=False)
    kw = {}
    for f in dataclasses.fields(NetSpec):
        if f.name not in z.files:
            continue
        v = z[f.name]
        kw[f.name] = str(v) if f.name == "name" else (v.item() if v.shape == () else v)
    return NetSpec(**kw)


def from_pandapower(net) -> NetSpec:
    """pandapowerNet -> NetSpec by column copying (lines, loads, sgens, one ext_grid, shunts).
    Transformers and bus-bus switches are not converted here."""
    if len(net.ext_grid) != 1:
        raise NotImplementedError("exactly one ext_grid expected")
    if len(getattr(net, "trafo", [])):
        raise NotImplementedError("transformers: convert to per-unit pi branches (NetSpec.br_*) first")
    bus_index = np.sort(net.bus.index.to_numpy())
    if not np.array_equal(bus_index, np.arange(len(bus_index))):
        raise NotImplementedError("bus indices must be 0..n-1")
    zones = net.bus["zone"].sort_index().to_numpy()

    def zid(z):
        return 0 if z == "main" else int(str(z).replace("zone", ""))
    line = net.line.sort_index()
    sh = getattr(net, "shunt", None)
    kw = dict(
        name=str(getattr(net, "name", "net")), bus_vn_kv=net.bus["vn_kv"].sort_index().to_numpy(),
        bus_zone=np.array([zid(z) for z in zones]),
        line_from_bus=line["from_bus"].to_numpy(), line_to_bus=line["to_bus"].to_numpy(),
        line_r_ohm_per_km=line["r_ohm_per_km"].to_numpy(), line_x_ohm_per_km=line["x_ohm_per_km"].to_numpy(),
        line_c_nf_per_km=line["c_nf_per_km"].to_numpy(),
        line_g_us_per_km=line["g_us_per_km"].to_numpy() if "g_us_per_km" in line else np.zeros(len(line)),
        line_length_km=line["length_km"].to_numpy(), line_parallel=line["parallel"].to_numpy(),
        line_in_service=line["in_service"].to_numpy().astype(np.uint8),
        load_bus=net.load["bus"].to_numpy(), sgen_bus=net.sgen["bus"].to_numpy(),
        sgen_zone=np.array([zid(z) for z in net.sgen["name"].to_numpy()]),
        ext_grid_bus=int(net.ext_grid["bus"].iloc[0]), ext_grid_vm_pu=float(net.ext_grid["vm_pu"].iloc[0]),
        sn_mva=float(net.sn_mva), f_hz=float(net.f_hz))
    if sh is not None and len(sh):
        kw.update(shunt_bus=sh["bus"].to_numpy(), shunt_p_mw=sh["p_mw"].to_numpy(), shunt_q_mvar=sh["q_mvar"].to_numpy())
    return NetSpec(**kw)


def load_scenario(data_path: str, pv_scale: float = 1.0, demand_scale: float = 1.0):
    """(NetSpec, Profiles) of a scenario directory: netspec.npz (or model.p with pandapower) + the CSVs."""
    npz = os.path.join(data_path, "netspec.npz")
    if os.path.exists(npz):
        net = load_netspec(npz)
    elif os.path.exists(os.path.join(data_path, "model.p")):
        try:
            import pandapower as pp
        except ImportError as e:
            raise NotImplementedError("model.p is a pandapower pickle and pandapower is not installed; "
                                      "export a netspec.npz with mapdn_amd.data.save_netspec") from e
        net = from_pandapower(pp.from_pickle(os.path.join(data_path, "model.p")))
    else:
        raise FileNotFoundError(f"no netspec.npz / model.p in {data_path}")
    return net, load_profiles_csv(data_path, pv_scale, demand_scale)
