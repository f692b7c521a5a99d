"""TEST INFRASTRUCTURE: a byte-level pickle ASSEMBLER that writes a pandapower net the way the reference's pinned stack would
(environment.yml:66,125,133,134: Python 3.7.9, numpy 1.19.5, pandapower 2.7.0, pandas 1.1.3) — without that stack, and without letting
today's pandas / numpy choose the opcodes, module paths or state layouts.  The real `model.p` (voltage_control_env.py:403-404) is an
external download; what it contains is fixed by those versions:

  form A  `pandapower.to_pickle(net, path)` (io_utils.to_dict_with_coord_transform + pickle.dump(..., protocol=2)): a plain dict, every
          table {"DF": DataFrame.to_dict("split"), "dtypes": {column: numpy.dtype}}.  Protocol 2 under Python 3: str = BINUNICODE,
          bool = NEWTRUE / NEWFALSE, dtype = `numpy dtype` REDUCE ('f8', False, True) + BUILD (3, '<', None, None, None, -1, -1, 0).
  form B  `pickle.dump(net, f)` (Python 3.7 default protocol 3): `pandapower.auxiliary.pandapowerNet` NEWOBJ + SETITEMS + BUILD
          ((mapping, False): ADict.__getstate__), the tables real pandas-1.1.3 DataFrames:
            DataFrame    NEWOBJ, BUILD {"_mgr": BlockManager, "_typ": "dataframe", "_metadata": [], "attrs": {}}   (NDFrame.__getstate__)
            BlockManager NEWOBJ, BUILD (axes, block_values, block_items, {"0.14.1": {"axes": ..., "blocks": [{"values", "mgr_locs"}]}})
            Index        `pandas.core.indexes.base _new_Index` REDUCE (cls, {"data": ndarray, "name": None}) with cls
                         `pandas.core.indexes.numeric Int64Index` for integer labels (gone from pandas 2), `...base Index` for str
            ndarray      `numpy.core.multiarray _reconstruct` REDUCE (ndarray, (0,), b"b") + BUILD (1, shape, dtype, False, raw bytes /
                         list of objects); consolidated blocks [columns of one dtype, rows], mgr_locs a `builtins slice` or int64 array.
Every GLOBAL the two files reference is listed in `GLOBALS_A` / `GLOBALS_B` so that the test can hold the unpickler's allow-list to it."""
import struct

import numpy as np
import pandas as pd

GLOBALS_A = {("numpy", "dtype")}
GLOBALS_B2 = {("_codecs", "encode"), ("__builtin__", "bytes"), ("__builtin__", "slice")}       # protocol 2 instead of ("builtins", "slice")
GLOBALS_B = {("numpy", "dtype"), ("numpy", "ndarray"), ("numpy.core.multiarray", "_reconstruct"), ("builtins", "slice"),
             ("pandas.core.frame", "DataFrame"), ("pandas.core.internals.managers", "BlockManager"),
             ("pandas.core.indexes.base", "_new_Index"), ("pandas.core.indexes.base", "Index"),
             ("pandas.core.indexes.numeric", "Int64Index"), ("pandas.core.indexes.range", "RangeIndex"),
             ("pandapower.auxiliary", "pandapowerNet")}


class Asm:
    def __init__(self, proto):
        self.proto, self.out, self.n_memo = proto, [bytes([0x80, proto])], 0

    def raw(self, b):
        self.out.append(b)

    def glob(self, module, name):
        self.raw(b"c" + module.encode() + b"\n" + name.encode() + b"\n")

    def put(self):
        i = self.n_memo
        self.n_memo += 1
        self.raw(b"q" + bytes([i]) if i < 256 else b"r" + struct.pack("<I", i))
        return i

    def get(self, i):
        self.raw(b"h" + bytes([i]) if i < 256 else b"j" + struct.pack("<I", i))

    # ---- scalars and builtin containers, as CPython 3.7's pickler writes them ----
    def obj(self, v):
        if v is None:
            self.raw(b"N")
        elif isinstance(v, (bool, np.bool_)):
            self.raw(b"\x88" if v else b"\x89")
        elif isinstance(v, (int, np.integer)):
            v = int(v)
            if 0 <= v < 256:
                self.raw(b"K" + bytes([v]))
            elif 0 <= v < 65536:
                self.raw(b"M" + struct.pack("<H", v))
            elif -2 ** 31 <= v < 2 ** 31:
                self.raw(b"J" + struct.pack("<i", v))
            else:
                b = v.to_bytes((v.bit_length() + 8) // 8, "little", signed=True)
                self.raw(b"\x8a" + bytes([len(b)]) + b)
        elif isinstance(v, (float, np.floating)):
            self.raw(b"G" + struct.pack(">d", float(v)))
        elif isinstance(v, str):
            u = v.encode("utf-8")
            self.raw(b"X" + struct.pack("<I", len(u)) + u)
        elif isinstance(v, bytes):
            self.bytes_(v)
        elif isinstance(v, np.dtype):
            self.dtype(v)
        elif isinstance(v, tuple):
            if len(v) == 0:
                self.raw(b")")
            elif len(v) <= 3:
                for x in v:
                    self.obj(x)
                self.raw({1: b"\x85", 2: b"\x86", 3: b"\x87"}[len(v)])
            else:
                self.raw(b"(")
                for x in v:
                    self.obj(x)
                self.raw(b"t")
        elif isinstance(v, list):
            self.raw(b"]")
            if v:
                self.raw(b"(")
                for x in v:
                    self.obj(x)
                self.raw(b"e")
        elif isinstance(v, dict):
            self.raw(b"}")
            if v:
                self.raw(b"(")
                for k, x in v.items():
                    self.obj(k); self.obj(x)
                self.raw(b"u")
        elif isinstance(v, slice):
            self.glob("builtins" if self.proto >= 3 else "__builtin__", "slice")
            self.obj((v.start, v.stop, v.step))
            self.raw(b"R")
        elif callable(getattr(v, "_emit", None)):
            v._emit(self)
        else:
            raise TypeError(type(v))

    def bytes_(self, b):
        if self.proto >= 3:
            self.raw((b"C" + bytes([len(b)]) if len(b) < 256 else b"B" + struct.pack("<I", len(b))) + b)
        elif not b:                                   # Python 3, protocol <= 2: bytes() call, Python-2 module name (fix_imports)
            self.glob("__builtin__", "bytes"); self.raw(b")R")
        else:                                         # ... and _codecs.encode(latin-1 text, "latin1") for the rest
            self.glob("_codecs", "encode")
            self.obj((b.decode("latin1"), "latin1"))
            self.raw(b"R")

    def dtype(self, dt):
        """numpy 1.19 arraydescr_reduce: numpy.dtype(str, False, True) + version-3 state"""
        dt = np.dtype(dt)
        self.glob("numpy", "dtype")
        if dt.kind == "O":
            self.obj(("O8", False, True)); self.raw(b"R")
            self.obj((3, "|", None, None, None, -1, -1, 63))
        else:
            self.obj((dt.str[1:], False, True)); self.raw(b"R")
            self.obj((3, "|" if dt.itemsize == 1 else "<", None, None, None, -1, -1, 0))
        self.raw(b"b")

    # ---- numpy 1.19 ndarray.__reduce__ ----
    def ndarray(self, a):
        a = np.asarray(a)
        self.glob("numpy.core.multiarray", "_reconstruct")
        self.glob("numpy", "ndarray")
        self.obj((0,)); self.bytes_(b"b")
        self.raw(b"\x87R")
        self.raw(b"(")
        self.obj(1); self.obj(tuple(int(s) for s in a.shape)); self.dtype(a.dtype); self.obj(False)
        if a.dtype.kind == "O":
            self.obj([x for x in a.ravel(order="C").tolist()])
        else:
            self.bytes_(np.ascontiguousarray(a).tobytes())
        self.raw(b"tb")

    # ---- pandas 1.1.3 ----
    def index(self, labels, kind):
        """kind 'int64' -> pandas.core.indexes.numeric.Int64Index, 'object' -> pandas.core.indexes.base.Index, 'range' -> RangeIndex"""
        self.glob("pandas.core.indexes.base", "_new_Index")
        if kind == "range":
            self.glob("pandas.core.indexes.range", "RangeIndex")
            self.raw(b"}(")
            self.obj("name"); self.obj(None); self.obj("start"); self.obj(0); self.obj("stop"); self.obj(len(labels)); self.obj("step"); self.obj(1)
            self.raw(b"u")
        else:
            self.glob(*(("pandas.core.indexes.numeric", "Int64Index") if kind == "int64" else ("pandas.core.indexes.base", "Index")))
            self.raw(b"}(")
            self.obj("data"); self.ndarray(np.asarray(labels, dtype=np.int64 if kind == "int64" else object))
            self.obj("name"); self.obj(None)
            self.raw(b"u")
        self.raw(b"\x86R")

    def dataframe(self, df, row_index="int64"):
        """consolidated blocks in pandas 1.1.3's form_blocks order (float, int, bool, object)"""
        cols = list(df.columns)
        kinds = [("f", np.float64), ("iu", None), ("b", np.bool_), ("O", object)]
        blocks = []
        for kind, _ in kinds:
            pos = [i for i, c in enumerate(cols) if df[c].dtype.kind in kind]
            by_dtype = {}
            for i in pos:
                by_dtype.setdefault(df[cols[i]].dtype, []).append(i)
            for dt, ps in by_dtype.items():
                vals = np.empty((len(ps), len(df)), dtype=dt)
                for r, i in enumerate(ps):
                    vals[r] = df[cols[i]].to_numpy()
                step = ps[1] - ps[0] if len(ps) > 1 else 1
                contiguous = all(ps[k + 1] - ps[k] == step for k in range(len(ps) - 1))
                locs = slice(ps[0], ps[-1] + 1 if step == 1 else ps[-1] + step, step) if contiguous else np.asarray(ps, dtype=np.int64)
                blocks.append((vals, ps, locs))

        def axes():
            self.raw(b"](")
            self.index(cols, "object")
            self.index(list(df.index), row_index)
            self.raw(b"e")

        def locs_(l):
            self.obj(l) if isinstance(l, slice) else self.ndarray(l)

        self.glob("pandas.core.frame", "DataFrame"); self.raw(b")\x81")
        self.raw(b"}(")
        self.obj("_mgr")
        self.glob("pandas.core.internals.managers", "BlockManager"); self.raw(b")\x81")
        self.raw(b"(")                                              # the 4-tuple of BlockManager.__getstate__
        axes()
        self.raw(b"](" if blocks else b"]")
        for vals, _, _ in blocks:
            self.ndarray(vals)
        if blocks:
            self.raw(b"e")
        self.raw(b"](" if blocks else b"]")
        for _, ps, _ in blocks:
            self.index([cols[i] for i in ps], "object")
        if blocks:
            self.raw(b"e")
        self.raw(b"}("); self.obj("0.14.1")
        self.raw(b"}("); self.obj("axes"); axes(); self.obj("blocks")
        self.raw(b"](" if blocks else b"]")
        for vals, _, l in blocks:
            self.raw(b"}("); self.obj("values"); self.ndarray(vals); self.obj("mgr_locs"); locs_(l); self.raw(b"u")
        if blocks:
            self.raw(b"e")
        self.raw(b"u"); self.raw(b"u")
        self.raw(b"tb")                                             # close the 4-tuple, BUILD the BlockManager
        self.obj("_typ"); self.obj("dataframe"); self.obj("_metadata"); self.obj([]); self.obj("attrs"); self.obj({})
        self.raw(b"ub")

    def done(self):
        return b"".join(self.out) + b"."


def to_pickle_form(net_items: dict) -> bytes:
    """form A: what pandapower 2.7.0's to_pickle writes for {key: DataFrame | scalar | dict}"""
    a = Asm(2)
    a.raw(b"}(")
    for k, v in net_items.items():
        a.obj(k)
        if isinstance(v, pd.DataFrame):
            split = {"index": [int(i) for i in v.index], "columns": [str(c) for c in v.columns],
                     "data": [[(x.item() if hasattr(x, "item") else x) for x in row] for row in v.itertuples(index=False, name=None)]}
            a.raw(b"}("); a.obj("DF"); a.obj(split); a.obj("dtypes")
            a.raw(b"}(")
            for c, dt in zip(v.columns, v.dtypes):
                a.obj(str(c)); a.dtype(dt)
            a.raw(b"u"); a.raw(b"u")
        else:
            a.obj(v)
    a.raw(b"u")
    return a.done()


def object_form(net_items: dict, range_index_tables=(), proto=3) -> bytes:
    """form B: pickle.dump(pandapowerNet) under Python 3.7 / pandas 1.1.3 (protocol 3, the 3.7 default; or protocol=2: bytes then
    travel as _codecs.encode(text, 'latin1') / __builtin__.bytes() and builtins carry their Python-2 module name)"""
    a = Asm(proto)
    a.glob("pandapower.auxiliary", "pandapowerNet"); a.raw(b")\x81")
    memo = {}
    a.raw(b"(")
    for k, v in net_items.items():
        a.obj(k)
        if isinstance(v, pd.DataFrame):
            a.dataframe(v, "range" if k in range_index_tables else "int64")
        else:
            a.obj(v)
        memo[k] = a.put()
    a.raw(b"u")
    a.raw(b"}(")                                                     # ADict.__getstate__: (self.copy(), _allow_invalid_attributes)
    for k in net_items:
        a.obj(k); a.get(memo[k])
    a.raw(b"u\x89\x86b")
    return a.done()


def globals_in(data: bytes):
    import pickletools
    out = set()
    for op, arg, _ in pickletools.genops(data):
        if op.name == "GLOBAL":
            out.add(tuple(arg.split(" ", 1)))
    return out
