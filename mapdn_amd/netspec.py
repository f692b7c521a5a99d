"""Plain-array description of a distribution network + the synthetic MAPDN-shaped cases.

The reference keeps its network as a pandapower pickle (``model.p``, loaded at
/root/reference/environments/var_voltage_control/voltage_control_env.py:400-405) and three CSV
profile tables (``:407-438``).  Neither pandapower nor the data ship with the reference, so this
module defines the *minimum* set of columns the hot path reads, as numpy arrays with pandapower's
column names and units, plus deterministic generators for the three scenario shapes
(reference README.md:299-303: 33/141/322 buses, 32/84/337 loads, 4/9/22 regions, 6/22/38 PVs).

Nothing here computes a power flow: per-unit conversion / Ybus live in the C++ host side of
``libmapdn_hip.so`` (product) and, independently, in ``oracle/pp_restated.py`` (checker).
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass, field

import numpy as np

MAIN_ZONE = 0  # zone id of the "main" zone, which has no agent (voltage_control_env.py:84)


@dataclass
class NetSpec:
    """Columns of a pandapower net that ``runpp`` + ``VoltageControl`` actually read.

    Index conventions follow pandapower: element tables are positional (row i == pandapower index i)
    and bus references are pandapower bus indices, which here are required to be ``0..n_bus-1``.
    """

    name: str
    # bus table
    bus_vn_kv: np.ndarray            # [nb] f64
    bus_zone: np.ndarray             # [nb] i32, 0 == "main", k == "zone{k}"   (bus.zone, env.py:536)
    # line table (pandapower `net.line`)
    line_from_bus: np.ndarray        # [n_line] i32
    line_to_bus: np.ndarray          # [n_line] i32
    line_r_ohm_per_km: np.ndarray    # [n_line] f64
    line_x_ohm_per_km: np.ndarray
    line_c_nf_per_km: np.ndarray
    line_g_us_per_km: np.ndarray
    line_length_km: np.ndarray
    line_parallel: np.ndarray        # [n_line] i32
    line_in_service: np.ndarray      # [n_line] u8
    # element tables
    load_bus: np.ndarray             # [nl] i32
    sgen_bus: np.ndarray             # [ns] i32
    sgen_zone: np.ndarray            # [ns] i32   (sgen.name == "zone{k}", env.py:532)
    ext_grid_bus: int = 0
    ext_grid_vm_pu: float = 1.0
    sn_mva: float = 1.0
    f_hz: float = 50.0
    # generic per-unit pi-branches (transformers after T->pi conversion, or anything else that is
    # not a `line`): MATPOWER/ppc branch columns  f, t, r, x, b, ratio(0 => 1), shift_degree
    br_from_bus: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    br_to_bus: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    br_r_pu: np.ndarray = field(default_factory=lambda: np.zeros(0))
    br_x_pu: np.ndarray = field(default_factory=lambda: np.zeros(0))
    br_b_pu: np.ndarray = field(default_factory=lambda: np.zeros(0))
    br_ratio: np.ndarray = field(default_factory=lambda: np.zeros(0))
    br_shift_deg: np.ndarray = field(default_factory=lambda: np.zeros(0))
    br_g_pu: np.ndarray = field(default_factory=lambda: np.zeros(0))      # ppc BR_B = br_b_pu - 1j*br_g_pu (trafo iron losses)
    # constant-impedance shunts (pandapower `net.shunt`, MW/MVAr at 1 p.u.; consumer sign)
    shunt_bus: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    shunt_p_mw: np.ndarray = field(default_factory=lambda: np.zeros(0))
    shunt_q_mvar: np.ndarray = field(default_factory=lambda: np.zeros(0))
    # net.load.scaling / net.sgen.scaling times in_service (pd2ppc: PD = sum p_mw * scaling over in-service elements);
    # empty = all ones
    load_scaling: np.ndarray = field(default_factory=lambda: np.zeros(0))
    sgen_scaling: np.ndarray = field(default_factory=lambda: np.zeros(0))
    # bus fusion (closed bus-bus switches, pd2ppc's bus lookup): bus_alias[b] = the REPRESENTATIVE bus of b's group, b itself for a bus
    # that is not fused.  Fused buses are one electrical node — same vm_pu / va in res_bus — but stay rows of every table the env
    # reads: their own p_mw / q_mvar, their own place in the zone frames, the reward's averages.  empty = no fusion
    bus_alias: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))

    def __post_init__(self):
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        for k in ("bus_vn_kv", "line_r_ohm_per_km", "line_x_ohm_per_km", "line_c_nf_per_km",
                  "line_g_us_per_km", "line_length_km", "br_r_pu", "br_x_pu", "br_b_pu",
                  "br_ratio", "br_shift_deg", "shunt_p_mw", "shunt_q_mvar", "br_g_pu", "load_scaling", "sgen_scaling"):
            setattr(self, k, f64(getattr(self, k)))
        if self.br_g_pu.shape[0] == 0 and np.shape(self.br_r_pu)[0]:
            self.br_g_pu = np.zeros(np.shape(self.br_r_pu)[0])
        if self.load_scaling.shape[0] == 0:
            self.load_scaling = np.ones(np.shape(self.load_bus)[0])
        if self.sgen_scaling.shape[0] == 0:
            self.sgen_scaling = np.ones(np.shape(self.sgen_bus)[0])
        if np.shape(self.bus_alias)[0] == 0:
            self.bus_alias = np.arange(np.shape(self.bus_vn_kv)[0])
        for k in ("bus_zone", "line_from_bus", "line_to_bus", "line_parallel", "load_bus",
                  "sgen_bus", "sgen_zone", "br_from_bus", "br_to_bus", "shunt_bus", "bus_alias"):
            setattr(self, k, i32(getattr(self, k)))
        if not np.array_equal(self.bus_alias[self.bus_alias], self.bus_alias):
            raise ValueError("bus_alias must map every bus to a representative that represents itself")
        self.line_in_service = np.ascontiguousarray(self.line_in_service, dtype=np.uint8)

    # ---- sizes -------------------------------------------------------------------------------
    @property
    def n_bus(self) -> int:
        return int(self.bus_vn_kv.shape[0])

    @property
    def n_line(self) -> int:
        return int(self.line_from_bus.shape[0])

    @property
    def n_branch_pu(self) -> int:
        return int(self.br_from_bus.shape[0])

    @property
    def n_load(self) -> int:
        return int(self.load_bus.shape[0])

    @property
    def n_sgen(self) -> int:
        return int(self.sgen_bus.shape[0])

    @property
    def has_fused_buses(self) -> bool:
        return bool((self.bus_alias != np.arange(self.n_bus)).any())

    @property
    def n_zones(self) -> int:
        """number of non-main zones"""
        return int(self.bus_zone.max())

    def copy(self) -> "NetSpec":
        return dataclasses.replace(
            self, **{f.name: np.array(getattr(self, f.name), copy=True)
                     for f in dataclasses.fields(self) if isinstance(getattr(self, f.name), np.ndarray)})

    # ---- integer artefacts of get_obs (bit-exact between product and oracle) ----------------
    def zone_buses(self, zone: int) -> np.ndarray:
        """ascending bus indices of a zone: `res_bus.sort_index().loc[bus.zone == name]` (env.py:536)"""
        return np.nonzero(self.bus_zone == zone)[0].astype(np.int32)

    def agent_zone_sizes(self) -> np.ndarray:
        return np.array([self.zone_buses(int(z)).shape[0] for z in self.sgen_zone], dtype=np.int32)

    def obs_size(self) -> int:
        """4*Z_max + 2 for the default 5-key state_space (env.py:254-274)"""
        return int(4 * self.agent_zone_sizes().max() + 2)

    def state_size(self) -> int:
        return 4 * self.n_bus + 2 * self.n_sgen  # env.py:217-229


@dataclass
class Profiles:
    """The three CSV tables after scaling (env.py:407-438): rows = 3-min intervals."""

    pv: np.ndarray               # [T, ns] MW
    load_p: np.ndarray           # [T, nl] MW
    load_q: np.ndarray           # [T, nl] MVAr
    time_delta_min: int = 3
    days: int = 0                # (index[-1]-index[0]).days, env.py:395

    def __post_init__(self):
        self.pv = np.ascontiguousarray(self.pv, dtype=np.float64)
        self.load_p = np.ascontiguousarray(self.load_p, dtype=np.float64)
        self.load_q = np.ascontiguousarray(self.load_q, dtype=np.float64)
        if not self.days:
            per_day = 24 * 60 // self.time_delta_min
            # a table of exactly D days spans D days minus one interval => .days == D-1
            self.days = (self.pv.shape[0] - 1) // per_day

    @property
    def n_rows(self) -> int:
        return int(self.pv.shape[0])

    @property
    def intervals_per_hour(self) -> int:
        return 60 // self.time_delta_min

    @property
    def intervals_per_day(self) -> int:
        return 24 * self.intervals_per_hour

    # env.py:70-72 (population std over the whole table, /100).  The reference takes it of `DataFrame.values`, an F-ordered block: numpy
    # then sums every column pairwise along its contiguous axis.  On a C-ordered table the same call adds row after row, which at the
    # real data's length (526 080 rows) lands 2e-12 away — so the order is reproduced here (and in csrc/colstats.hpp).
    def stds(self):
        f = np.asfortranarray
        return (f(self.pv).std(axis=0) / 100.0, f(self.load_p).std(axis=0) / 100.0, f(self.load_q).std(axis=0) / 100.0)

    # env.py:515-520
    def s_max(self, factor: float = 1.2) -> np.ndarray:
        return factor * self.pv.max(axis=0)

    def n_start_days(self, episode_limit: int) -> int:
        """`np.random.choice(pv_days - episode_days)` upper bound (env.py:395-398)"""
        episode_days = episode_limit // self.intervals_per_day + 1
        return self.days - episode_days

    def start_row(self, day: int, hour: int, interval: int) -> int:
        """env.py:445"""
        return interval + hour * self.intervals_per_hour + day * self.intervals_per_day


# ------------------------------------------------------------------------------------------------
# IEEE 33-bus feeder of Baran & Wu (1989), 12.66 kV — public literature data (NOT from the
# reference repo).  Rows: from, to (1-based), r [ohm], x [ohm], then P [kW], Q [kVAr] at `to`.
# ------------------------------------------------------------------------------------------------
_BW33 = np.array([
    [1, 2, 0.0922, 0.0470, 100, 60], [2, 3, 0.4930, 0.2511, 90, 40], [3, 4, 0.3660, 0.1864, 120, 80],
    [4, 5, 0.3811, 0.1941, 60, 30], [5, 6, 0.8190, 0.7070, 60, 20], [6, 7, 0.1872, 0.6188, 200, 100],
    [7, 8, 0.7114, 0.2351, 200, 100], [8, 9, 1.0300, 0.7400, 60, 20], [9, 10, 1.0440, 0.7400, 60, 20],
    [10, 11, 0.1966, 0.0650, 45, 30], [11, 12, 0.3744, 0.1238, 60, 35], [12, 13, 1.4680, 1.1550, 60, 35],
    [13, 14, 0.5416, 0.7129, 120, 80], [14, 15, 0.5910, 0.5260, 60, 10], [15, 16, 0.7463, 0.5450, 60, 20],
    [16, 17, 1.2890, 1.7210, 60, 20], [17, 18, 0.7320, 0.5740, 90, 40], [2, 19, 0.1640, 0.1565, 90, 40],
    [19, 20, 1.5042, 1.3554, 90, 40], [20, 21, 0.4095, 0.4784, 90, 40], [21, 22, 0.7089, 0.9373, 90, 40],
    [3, 23, 0.4512, 0.3083, 90, 50], [23, 24, 0.8980, 0.7091, 420, 200], [24, 25, 0.8960, 0.7011, 420, 200],
    [6, 26, 0.2030, 0.1034, 60, 25], [26, 27, 0.2842, 0.1447, 60, 25], [27, 28, 1.0590, 0.9337, 60, 20],
    [28, 29, 0.8042, 0.7006, 120, 70], [29, 30, 0.5075, 0.2585, 200, 600], [30, 31, 0.9744, 0.9630, 150, 70],
    [31, 32, 0.3105, 0.3619, 210, 100], [32, 33, 0.3410, 0.5302, 60, 40],
])


def case33bw_base():
    """(NetSpec without PVs, p_load_mw[32], q_load_mvar[32]) of the Baran-Wu base case."""
    f = _BW33[:, 0].astype(np.int32) - 1
    t = _BW33[:, 1].astype(np.int32) - 1
    n_line = f.shape[0]
    # zones (our choice, the reference's zoning lives in the absent model.p):
    #   main: trunk buses 1-6; zone1: lateral 19-22; zone2: lateral 23-25; zone3: lateral 26-33;
    #   zone4: trunk 7-18   (1-based bus numbers)
    zone = np.zeros(33, np.int32)
    zone[18:22] = 1
    zone[22:25] = 2
    zone[25:33] = 3
    zone[6:18] = 4
    net = NetSpec(
        name="case33",
        bus_vn_kv=np.full(33, 12.66), bus_zone=zone,
        line_from_bus=f, line_to_bus=t,
        line_r_ohm_per_km=_BW33[:, 2], line_x_ohm_per_km=_BW33[:, 3],
        line_c_nf_per_km=np.zeros(n_line), line_g_us_per_km=np.zeros(n_line),
        line_length_km=np.ones(n_line), line_parallel=np.ones(n_line, np.int32),
        line_in_service=np.ones(n_line, np.uint8),
        load_bus=t.copy(),                       # one load per non-slack bus, in branch-row order
        sgen_bus=np.array([20, 23, 28, 31, 11, 16], np.int32),   # 0-based bus indices
        sgen_zone=np.array([1, 2, 3, 3, 4, 4], np.int32),
        ext_grid_bus=0, ext_grid_vm_pu=1.0, sn_mva=1.0, f_hz=50.0,
    )
    return net, _BW33[:, 4] * 1e-3, _BW33[:, 5] * 1e-3


# The five normally-open tie lines of the Baran-Wu feeder (public literature data): from, to (1-based), r, x [ohm].
_BW33_TIES = np.array([[8, 21, 2.0, 2.0], [9, 15, 2.0, 2.0], [12, 22, 2.0, 2.0], [18, 33, 0.5, 0.5], [25, 29, 0.5, 0.5]])


def add_lines(net: "NetSpec", from_bus, to_bus, r_ohm, x_ohm, c_nf=0.0, length_km=1.0) -> "NetSpec":
    """A copy of `net` with extra in-service rows appended to the line table (closing tie switches makes the net
    meshed; pandapower's runpp solves it all the same — voltage_control_env.py:557)."""
    out = net.copy()
    k = len(np.atleast_1d(from_bus))
    cat = lambda a, b, dt: np.concatenate([a, np.broadcast_to(np.asarray(b, dt), (k,))]).astype(dt)
    out.line_from_bus = cat(net.line_from_bus, from_bus, np.int32)
    out.line_to_bus = cat(net.line_to_bus, to_bus, np.int32)
    out.line_r_ohm_per_km = cat(net.line_r_ohm_per_km, r_ohm, np.float64)
    out.line_x_ohm_per_km = cat(net.line_x_ohm_per_km, x_ohm, np.float64)
    out.line_c_nf_per_km = cat(net.line_c_nf_per_km, c_nf, np.float64)
    out.line_g_us_per_km = cat(net.line_g_us_per_km, 0.0, np.float64)
    out.line_length_km = cat(net.line_length_km, length_km, np.float64)
    out.line_parallel = cat(net.line_parallel, 1, np.int32)
    out.line_in_service = cat(net.line_in_service, 1, np.uint8)
    out.__post_init__()
    return out


def case33_meshed(net: "NetSpec", n_ties: int = 5) -> "NetSpec":
    """case33 with the first `n_ties` Baran-Wu tie lines closed (a weakly meshed 33-bus net)."""
    t = _BW33_TIES[:n_ties]
    out = add_lines(net, t[:, 0].astype(np.int32) - 1, t[:, 1].astype(np.int32) - 1, t[:, 2], t[:, 3])
    out.name = f"{net.name}_meshed{n_ties}"
    return out


# ------------------------------------------------------------------------------------------------
# seeded random radial feeders with the case141 / case322 shapes
# ------------------------------------------------------------------------------------------------
def _radial_case(name, nb, n_load, n_sgen, n_zones, n_trunk, vn_kv, sn_mva, p_load_total_mw,
                 seed, drop_target):
    rng = np.random.default_rng(seed)
    assert (nb - n_trunk) % n_zones == 0
    zsize = (nb - n_trunk) // n_zones
    # build in "natural" labels first: 0..n_trunk-1 = trunk chain (main zone), then zones
    parent = np.full(nb, -1, np.int64)
    zone_nat = np.zeros(nb, np.int32)
    for i in range(1, n_trunk):
        parent[i] = i - 1
    for z in range(n_zones):
        base = n_trunk + z * zsize
        attach = int(rng.integers(1, n_trunk))          # zone hangs off a (non-slack) trunk bus
        for j in range(zsize):
            i = base + j
            zone_nat[i] = z + 1
            if j == 0:
                parent[i] = attach
            elif rng.random() < 0.7:
                parent[i] = i - 1                        # feeder-like: mostly chains
            else:
                parent[i] = base + int(rng.integers(0, j))
    # scramble non-slack bus labels so index handling (sort_index / zone masks) is non-trivial
    perm = np.concatenate([[0], 1 + rng.permutation(nb - 1)])      # natural -> bus index
    zone = np.zeros(nb, np.int32)
    zone[perm] = zone_nat
    child_nat = np.arange(1, nb)
    order = rng.permutation(nb - 1)                                # line-table row order
    f = perm[parent[child_nat]][order]
    t = perm[child_nat][order]
    swap = rng.random(nb - 1) < 0.3                                # some lines entered "backwards"
    f2 = np.where(swap, t, f)
    t2 = np.where(swap, f, t)
    n_line = nb - 1
    r_km = rng.uniform(0.1, 0.6, n_line)
    x_km = r_km * rng.uniform(0.3, 1.0, n_line)
    length = rng.uniform(0.1, 1.0, n_line)
    c_nf = rng.uniform(5.0, 15.0, n_line)
    # loads
    nonslack = perm[1:]
    if n_load <= nb - 1:
        load_bus = rng.choice(nonslack, size=n_load, replace=False)
    else:
        n_with = 300 if nb - 1 >= 300 else nb - 1
        first = rng.choice(nonslack, size=n_with, replace=False)
        second = rng.choice(first, size=n_load - n_with, replace=False)
        load_bus = np.concatenate([first, second])
        load_bus = load_bus[rng.permutation(n_load)]
    w = rng.uniform(0.5, 1.5, n_load)
    p_nom = p_load_total_mw * w / w.sum()
    # sgens: round-robin over zones so every zone has at least one when n_sgen >= n_zones
    sgen_zone = (np.arange(n_sgen) % n_zones + 1).astype(np.int32)
    sgen_zone = sgen_zone[rng.permutation(n_sgen)]
    sgen_bus = np.array([rng.choice(np.nonzero(zone == z)[0]) for z in sgen_zone], np.int32)

    # scale impedances so that the linearised (DistFlow) voltage drop at nominal load is `drop_target`
    zbase = vn_kv ** 2 / sn_mva
    r_pu = r_km * length / zbase
    x_pu = x_km * length / zbase
    pq = np.zeros((nb, 2))
    np.add.at(pq[:, 0], load_bus, p_nom / sn_mva)
    np.add.at(pq[:, 1], load_bus, p_nom * np.tan(np.arccos(0.95)) / sn_mva)
    # accumulate downstream power in natural labels (children have larger natural index than parents
    # except zone attach points, so process in reverse natural order)
    inv = np.empty(nb, np.int64)
    inv[perm] = np.arange(nb)
    down = pq[perm].copy()                               # natural order
    for i in range(nb - 1, 0, -1):
        down[parent[i]] += down[i]
    edge_r = np.zeros(nb)
    edge_x = np.zeros(nb)
    edge_r[inv[perm[child_nat][order]]] = r_pu
    edge_x[inv[perm[child_nat][order]]] = x_pu
    drop = np.zeros(nb)
    for i in range(1, nb):
        drop[i] = drop[parent[i]] + edge_r[i] * down[i, 0] + edge_x[i] * down[i, 1]
    scale = drop_target / drop.max()
    net = NetSpec(
        name=name, bus_vn_kv=np.full(nb, vn_kv), bus_zone=zone,
        line_from_bus=f2, line_to_bus=t2,
        line_r_ohm_per_km=r_km * scale, line_x_ohm_per_km=x_km * scale,
        line_c_nf_per_km=c_nf, line_g_us_per_km=np.zeros(n_line),
        line_length_km=length, line_parallel=np.ones(n_line, np.int32),
        line_in_service=np.ones(n_line, np.uint8),
        load_bus=load_bus, sgen_bus=sgen_bus, sgen_zone=sgen_zone,
        ext_grid_bus=0, ext_grid_vm_pu=1.0, sn_mva=sn_mva, f_hz=50.0,
    )
    return net, p_nom


def _tod_shapes(T, per_day):
    tod = (np.arange(T) % per_day) / per_day * 24.0                 # hour of day
    sun = np.maximum(0.0, np.sin(np.pi * (tod - 6.0) / 12.0))       # 06:00-18:00 half sine
    hump = 0.5 * np.exp(-0.5 * ((tod - 8.0) / 1.5) ** 2) + np.exp(-0.5 * ((tod - 19.0) / 2.0) ** 2)
    hump = hump / hump.max()
    return sun, hump


def synth_profiles(p_nom_load, q_nom_load, p_pv_max, days=10, seed=0, time_delta_min=3):
    """Deterministic PV / load tables of the reference CSV shape."""
    rng = np.random.default_rng(seed + 1000)
    per_day = 24 * 60 // time_delta_min
    T = days * per_day
    sun, hump = _tod_shapes(T, per_day)
    ns, nl = p_pv_max.shape[0], p_nom_load.shape[0]
    pv = p_pv_max[None, :] * sun[:, None] * (0.7 + 0.3 * rng.random((T, ns)))
    lf = (0.6 + 0.4 * hump)[:, None] * (0.9 + 0.2 * rng.random((T, nl)))
    load_p = p_nom_load[None, :] * lf / 1.1          # peak <= nominal
    load_q = q_nom_load[None, :] * lf / 1.1
    return Profiles(pv=pv, load_p=load_p, load_q=load_q, time_delta_min=time_delta_min)


_CASES = {}


def add_fused_buses(net: NetSpec, reps, move_loads=(), move_sgens=(), move_shunts=()) -> NetSpec:
    """`net` plus one new bus per entry of `reps`, each FUSED with that existing bus (a closed bus-bus switch: bus_alias), and
    some elements moved onto the new buses — (element index, which new bus) pairs.  Electrically nothing changes (the fused buses
    are one node); the env's tables get more rows.  Test nets for the bus-fusion path."""
    nb, k = net.n_bus, len(reps)
    reps = np.asarray(reps, dtype=np.int32)
    lb, sb, hb = net.load_bus.copy(), net.sgen_bus.copy(), net.shunt_bus.copy()
    for arr, moves in ((lb, move_loads), (sb, move_sgens), (hb, move_shunts)):
        for el, which in moves:
            if net.bus_alias[arr[el]] != net.bus_alias[reps[which]]:
                raise ValueError("an element can only move within its fused group")
            arr[el] = nb + which
    return dataclasses.replace(
        net, name=net.name + f"_fused{k}", bus_vn_kv=np.concatenate([net.bus_vn_kv, net.bus_vn_kv[reps]]),
        bus_zone=np.concatenate([net.bus_zone, net.bus_zone[reps]]), bus_alias=np.concatenate([net.bus_alias, net.bus_alias[reps]]),
        load_bus=lb, sgen_bus=sb, shunt_bus=hb)


def make_case(name: str, days: int = 10, seed: int = 0):
    """Return (NetSpec, Profiles) for 'case33' | 'case141' | 'case322' (synthetic, deterministic; + 'case141_deep', a depth stress).

    Load / PV totals follow reference README.md:299-303 (p_max^L 3.5/20/1.5 MW, p_max^PV 8.75/80/3.75 MW).
    """
    key = (name, days, seed)
    if key in _CASES:
        net, prof = _CASES[key]
        return net.copy(), prof
    if name == "case33":
        net, p_nom, q_nom = case33bw_base()
        scale = 3.5 / p_nom.sum()
        p_nom, q_nom = p_nom * scale, q_nom * scale
        pv_total = 8.75
    elif name == "case141":
        net, p_nom = _radial_case("case141", 141, 84, 22, 9, 15, 12.47, 10.0, 20.0, seed + 141, 0.05)
        q_nom = p_nom * np.tan(np.arccos(0.95))
        pv_total = 80.0
    elif name == "case141_deep":
        # the 141-bus shape on a chain-heavy topology (45-bus trunk, 8 lateral zones of 12 buses): a depth stress for the tree
        # solver, whose sweep length is the radius of the feeder — not one of the reference's scenarios
        net, p_nom = _radial_case("case141_deep", 141, 84, 22, 8, 45, 12.47, 10.0, 20.0, seed + 1141, 0.05)
        q_nom = p_nom * np.tan(np.arccos(0.95))
        pv_total = 80.0
    elif name == "case322":
        net, p_nom = _radial_case("case322", 322, 337, 38, 22, 14, 20.0, 1.0, 1.5, seed + 322, 0.05)
        q_nom = p_nom * np.tan(np.arccos(0.95))
        pv_total = 3.75
    else:
        raise ValueError(f"unknown case {name!r}")
    rng = np.random.default_rng(seed + 7)
    w = rng.uniform(0.7, 1.3, net.n_sgen)
    p_pv_max = pv_total * w / w.sum()
    prof = synth_profiles(p_nom, q_nom, p_pv_max, days=days, seed=seed)
    _CASES[key] = (net, prof)
    return net.copy(), prof
