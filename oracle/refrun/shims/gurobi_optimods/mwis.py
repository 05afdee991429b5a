"""Exact maximum-weight independent set with the call signature of
`gurobi_optimods.mwis.maximum_weighted_independent_set(adjacency_matrix, weights, verbose=False)`
(sole call site in the reference: traceweaver_v3.py:1411).

Gurobi is closed source and unobtainable here, so the binary programme
    max  sum_i w_i x_i   s.t.  x_i + x_j <= 1  for every edge (i, j),  x binary
is handed to HiGHS through scipy.optimize.milp.  Any exact solver returns the same objective value
and, whenever the optimum is unique, the same vertex set; ties are solver-specific ("parity
unpinned" at this boundary).

TEST INFRASTRUCTURE ONLY.
"""
import numpy as np
import scipy.sparse as sp
from scipy.optimize import Bounds, LinearConstraint, milp

CALL_LOG = []  # (n_nodes, n_edges) per call, read by gen_golden.py


def maximum_weighted_independent_set(adjacency_matrix, weights, verbose=False):
    weights = np.asarray(weights, dtype=np.float64)
    n = weights.shape[0]
    adj = sp.coo_matrix(adjacency_matrix)
    rows, cols = adj.row, adj.col
    keep = rows != cols
    rows, cols = rows[keep], cols[keep]
    m = rows.shape[0]
    CALL_LOG.append((n, m))
    if n == 0:
        return np.array([], dtype=np.int64)
    if m == 0:
        return np.nonzero(weights > 0)[0]
    a = sp.coo_matrix(
        (np.ones(2 * m), (np.repeat(np.arange(m), 2), np.stack([rows, cols], 1).ravel())),
        shape=(m, n),
    ).tocsr()
    res = milp(
        c=-weights,
        constraints=LinearConstraint(a, -np.inf, 1.0),
        integrality=np.ones(n),
        bounds=Bounds(0, 1),
        options={"mip_rel_gap": 0.0, "presolve": True},
    )
    if res.x is None:
        raise RuntimeError("HiGHS failed on MWIS instance: %s" % res.message)
    return np.nonzero(res.x > 0.5)[0]
