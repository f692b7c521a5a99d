"""Type / dtype / shape summary of a value crossing the env boundary — shared by the fixture generator (run on the reference)
and the replay test (run on the drop-in).  Python / numpy scalar flavours are folded (np.float64 IS a float, np.bool_ and bool
both count as 'bool'): callers of the reference cannot tell them apart either; arrays keep dtype and shape."""
import numpy as np


def summarize(x):
    if x is None:
        return "None"
    if isinstance(x, (bool, np.bool_)):
        return "bool"
    if isinstance(x, (int, np.integer)):
        return "int"
    if isinstance(x, (float, np.floating)):
        return "float"
    if isinstance(x, str):
        return "str"
    if isinstance(x, np.ndarray):
        kind = {"f": "float", "i": "int", "u": "int", "b": "bool"}.get(x.dtype.kind, x.dtype.kind)
        return ["ndarray", kind, list(x.shape)]                     # float32 / float64 both 'float': the reference's loops feed
    if isinstance(x, dict):                                         # float32 actions and read float64 observations
        return ["dict", sorted(x.keys()), sorted({summarize(v) if isinstance(summarize(v), str) else "obj" for v in x.values()})]
    if isinstance(x, (list, tuple)):
        inner = [summarize(v) for v in x]
        same = all(i == inner[0] for i in inner) if inner else True
        return ["tuple" if isinstance(x, tuple) else "list", len(x), inner[0] if (inner and same) else inner]
    return type(x).__name__
