"""GPU-vs-oracle agreement of (Newton iterations, convergence flag): exact, except at the tolerance edge — see
oracle.pp_restated.iterations_agree and tests/test_nr_tolerance_edge.py (INTEGRATION.md "Iteration counts at the tolerance edge")."""
from oracle.pp_restated import TOLERANCE_MVA, iterate_norms, iterations_agree


def same_newton_count(net, inputs, gpu_iterations, gpu_converged, r, tolerance_mva=TOLERANCE_MVA, tolerance_is_pu=False):
    """r = runpp_restated(net, *inputs): True when the GPU's count / flag equal the oracle's, or differ only because an iterate's
    ||F||inf sits within the evaluation noise of the tolerance (the norms are recomputed only in that rare case)"""
    if int(gpu_iterations) == r.iterations and bool(gpu_converged) == bool(r.converged):
        return True
    if getattr(net, "has_fused_buses", False):
        return False
    tol = tolerance_mva / (1.0 if tolerance_is_pu else net.sn_mva)
    return iterations_agree(int(gpu_iterations), bool(gpu_converged), r.iterations, r.converged, iterate_norms(net, *inputs), tol)
