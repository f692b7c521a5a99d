def create_generic_coordinates(*a, **k):
    raise NotImplementedError("pandapower stub: plotting is out of scope")
