"""pandapower/auxiliary.py stand-ins: the exception base class and the attribute-dict net container."""


class ppException(Exception):
    """pandapower.auxiliary.ppException (caught at voltage_control_env.py:126,167,559)"""


class pandapowerNet(dict):
    """attribute-style dict of tables, like pandapower.auxiliary.pandapowerNet (ADict)"""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e

    def __setattr__(self, name, value):
        self[name] = value

    def __deepcopy__(self, memo):
        import copy
        out = pandapowerNet()
        memo[id(self)] = out
        for k, v in self.items():
            out[k] = copy.deepcopy(v, memo)
        return out
