"""Per-edge delay-mixture refit between the two passes (ComputeEpPairDistParams5, traceweaver_v3.py:706-818).

`fit_edge_sklearn` is the reference's procedure, line for line, with the library the reference uses
(scikit-learn): fit 1..min(5, #unique) diagonal mixtures, keep the component count with the smallest BIC,
refit a full-covariance mixture with random_state=100.  The model-selection fits draw from numpy's
global RNG exactly like the reference (traceweaver_v3.py:774, SURVEY.md hazard H9), so results are
reproducible only under `np.random.seed`.
"""
import numpy as np

MAX_COMP = 5


def fit_edge_sklearn(samples, n_selected=None):
    """Returns (n_components, params[5, 3] = weight, mean, precision_cholesky); n = 0 for no samples
    (the reference stores (0, 0) there and scores with a sigma = 0.001 Gaussian, traceweaver_v3.py:765-766)."""
    from sklearn import mixture

    d = np.asarray(samples, dtype=np.float64).reshape(-1, 1)
    out = np.zeros((MAX_COMP, 3))
    if len(d) == 0:
        return 0, out
    if n_selected is None:
        max_n = min(len(np.unique(d)), MAX_COMP)
        models, n_comps = [], []
        for n in range(1, max_n + 1):
            try:
                models.append(mixture.GaussianMixture(n_components=n, covariance_type="diag").fit(d))
                n_comps.append(n)
            except ValueError:
                continue
        n_selected = n_comps[int(np.argmin([m.bic(d) for m in models]))]
    g = mixture.GaussianMixture(n_components=n_selected, random_state=100).fit(d)
    out[:n_selected, 0] = g.weights_
    out[:n_selected, 1] = g.means_.ravel()
    out[:n_selected, 2] = g.precisions_cholesky_.ravel()
    return int(n_selected), out


def fit_unit(gaps, fit=fit_edge_sklearn):
    """gaps: [nslot, n_in] with NaN for dropped samples / unscored slots (Engine.gaps()).  Returns
    (mix_n [nslot] int32, mix_p [nslot, 5, 3])."""
    nslot = gaps.shape[0]
    mix_n = np.zeros(nslot, dtype=np.int32)
    mix_p = np.zeros((nslot, MAX_COMP, 3))
    for q in range(nslot):
        row = gaps[q]
        keep = ~np.isnan(row)
        if not keep.any():
            continue
        mix_n[q], mix_p[q] = fit(row[keep])
    return mix_n, mix_p
