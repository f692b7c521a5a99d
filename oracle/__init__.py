"""oracle — CPU restatement of the reference path.  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by mapdn_amd/.  Parity against pandapower itself
is unpinned here (pandapower is not installable offline); see DESIGN.md section 2 for the pins in place."""
