"""TEST INFRASTRUCTURE — a stand-in for the third-party package ``pandapower`` (reference pin
pandapower==2.7.0, /root/reference/environment.yml:133) that exists ONLY so that the reference's own
``environments/var_voltage_control/voltage_control_env.py`` can be imported and executed UNMODIFIED in
this container (tests/golden/make_env_golden.py), which pins oracle/env_restated.py — and through it
the HIP path — on outputs of the reference's real env code.

What is real and what is not:
  * real       : every line of the reference class (reset / manual_reset / step / _calc_reward / get_obs /
                 get_state / tester getters, the pandas chained-assignment add-back at :238-244, the
                 voltage barriers) runs as written, on real pandas DataFrames with pandapower's table
                 and column names and dtypes;
  * stand-in   : ``runpp`` solves the power flow with oracle/pp_restated.py (the restatement of
                 pandapower's published algorithm; PARITY UNPINNED against pandapower itself, see that
                 file's header) and writes res_bus / res_line / res_sgen / res_ext_grid the way pandapower's
                 result extraction does; ``from_pickle`` builds the net tables from the ``netspec.npz``
                 that sits next to the (absent) ``model.p``.
Nothing under mapdn_amd/ imports this package; it is never on sys.path in the product or in bench.py.
"""
from .auxiliary import ppException, pandapowerNet          # noqa: F401
from .powerflow import LoadflowNotConverged                # noqa: F401
from .run import runpp                                      # noqa: F401
from .file_io import from_pickle                            # noqa: F401

__version__ = "2.7.0+stub"
