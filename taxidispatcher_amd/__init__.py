"""taxidispatcher_amd — MI355X (gfx950) cab<->request assignment path.

cost-matrix build -> LCM greedy pre-reduce -> optimal N x N assignment as hand-written HIP
kernels behind a C ABI (include/taxidispatcher_amd.h).  See DESIGN.md / INTEGRATION.md.
"""
from . import _ffi
from ._ffi import TdError, init, shutdown
from .dispatch import (BIG_COST, LCM, Solver, LCM_heuristic, LCM_simulator, assign, build_assign, calculate_cost, calculate_cost_by_id,
                       combined, cost_build, count_sum, expand_x, filter_out, find_pool, find_pool_n, last_stats, merge_pools,
                       procedure_solve, set_line_metric, solve, solve_cost, tick)

__all__ = ["TdError", "init", "shutdown", "BIG_COST", "LCM", "Solver", "LCM_heuristic", "LCM_simulator", "assign", "build_assign",
           "calculate_cost", "calculate_cost_by_id", "combined", "cost_build", "count_sum", "expand_x", "filter_out", "find_pool",
           "find_pool_n", "merge_pools", "last_stats", "procedure_solve", "set_line_metric", "solve", "solve_cost", "tick"]
