"""CPU tests of the on-disk formats either side of the path (CSV profile tables, netspec.npz)."""
import numpy as np

from mapdn_amd.data import load_netspec, load_profiles_csv, load_scenario, save_netspec, save_profiles_csv
from mapdn_amd.netspec import Profiles, make_case


def test_csv_roundtrip_and_time_axis(tmp_path):
    net, prof = make_case("case33")
    small = Profiles(pv=prof.pv[:1500], load_p=prof.load_p[:1500], load_q=prof.load_q[:1500], time_delta_min=3)
    d = str(tmp_path / "case33_3min_final")
    save_profiles_csv(small, d)
    back = load_profiles_csv(d)
    assert back.time_delta_min == 3 and back.n_rows == 1500
    assert back.days == 3                      # (index[-1] - index[0]).days for 1500 three-minute rows
    assert np.array_equal(back.pv, small.pv) and np.array_equal(back.load_p, small.load_p) and np.array_equal(back.load_q, small.load_q)
    scaled = load_profiles_csv(d, pv_scale=0.5, demand_scale=2.0)                  # voltage_control_env.py:415,426,437
    assert np.array_equal(scaled.pv, 0.5 * small.pv) and np.array_equal(scaled.load_q, 2.0 * small.load_q)
    # derived quantities the env computes from the tables (:70-72, :515-520, :445)
    assert np.allclose(back.stds()[0], small.pv.std(axis=0) / 100) and np.allclose(back.s_max(), 1.2 * small.pv.max(axis=0))
    assert back.start_row(2, 5, 7) == 7 + 5 * 20 + 2 * 480


def test_netspec_npz_roundtrip_and_scenario_dir(tmp_path):
    for case in ("case33", "case322"):
        net, prof = make_case(case)
        p = str(tmp_path / f"{case}.npz")
        save_netspec(net, p)
        back = load_netspec(p)
        assert back.name == net.name and back.sn_mva == net.sn_mva and back.ext_grid_bus == net.ext_grid_bus
        for k in ("bus_vn_kv", "bus_zone", "line_from_bus", "line_r_ohm_per_km", "load_bus", "sgen_bus", "sgen_zone", "line_in_service"):
            assert np.array_equal(getattr(back, k), getattr(net, k)) and getattr(back, k).dtype == getattr(net, k).dtype
    d = str(tmp_path / "scen")
    net, prof = make_case("case33")
    save_profiles_csv(Profiles(pv=prof.pv[:1000], load_p=prof.load_p[:1000], load_q=prof.load_q[:1000]), d)
    save_netspec(net, d + "/netspec.npz")
    n2, p2 = load_scenario(d)
    assert n2.n_bus == 33 and p2.n_rows == 1000 and np.array_equal(p2.pv, prof.pv[:1000])
