"""TEST INFRASTRUCTURE (checker only; nothing under traceweaver_amd/ may import this).

CPU restatement of the reference's per-edge mixture refit, ComputeEpPairDistParams5
(algorithms/traceweaver_v3.py:764-786), with scikit-learn's procedure written out for one-dimensional samples and the
random draws taken from an explicit *tape* of uniforms instead of a RandomState:

    for n in 1..min(5, #unique):  GaussianMixture(n, covariance_type="diag").fit(x)     V3:772-779   (global numpy RNG)
    n_selected = argmin BIC                                                               V3:780
    GaussianMixture(n_selected, random_state=100).fit(x)   (full covariance)             V3:784-785

GaussianMixture.fit = KMeans(n, n_init=1) labels (k-means++ seeding + Lloyd) -> one-hot responsibilities -> M step -> EM
until |change of mean log-likelihood| < 1e-3 (<= 100 iterations), reg_covar 1e-6.  scikit-learn (third-party, not in
/root/reference; pinned 1.5.1 by the reference's requirements.txt, 1.7.2 installed here) is restated from its published
algorithm; tests/test_refit.py pins this restatement against scikit-learn itself run on the same uniforms
(`RandomState(seed)` hands out exactly the doubles `RandomState(seed).random_sample()` would): labels of the k-means
start identical, selected component counts identical, parameters to ~1e-9.

Draws per fit (independent of the data): 1 for the first centre (`choice(n, p)` = one double), then 2 + int(ln n) per
further centre (`uniform(size=trials)`) -> 1, 3, 7, 10, 13 doubles for n = 1..5.
"""
import numpy as np

MAX_COMP = 5
REG_COVAR = 1.0e-6
TOL = 1.0e-3
MAX_ITER = 100
KM_MAX_ITER = 300
KM_TOL = 1.0e-4
EPS10 = 10 * np.finfo(np.float64).eps
LOG2PI = float(np.log(2 * np.pi))


def draws_per_fit(k):
    return 1 + (k - 1) * (2 + int(np.log(k)))


def draws_per_row(max_n):
    """Doubles one row's model-selection fits consume (n = 1..max_n)."""
    return sum(draws_per_fit(k) for k in range(1, max_n + 1))


def refit_tape(k):
    """The draws of `GaussianMixture(k, random_state=100)`: a fresh RandomState(100) per fit (V3:784)."""
    return np.random.RandomState(100).random_sample(draws_per_fit(k))


def _dist(c, x, xx):
    """sklearn.metrics.pairwise._euclidean_distances(c, X, Y_norm_squared=xx, squared=True) for one feature."""
    d = -2.0 * (c * x)
    d += c * c
    d += xx
    return np.maximum(d, 0.0)


def kmeans_plusplus(xc, k, tape):
    """sklearn.cluster._kmeans._kmeans_plusplus on centred 1-D samples in their original order.  Returns centre values."""
    n = len(xc)
    xx = xc * xc
    trials = 2 + int(np.log(k))
    t = 0
    cdf = np.cumsum(np.full(n, 1.0 / n))
    cdf /= cdf[-1]
    cid = int(np.searchsorted(cdf, tape[t], side="right"))
    t += 1
    centers = [xc[cid]]
    closest = _dist(xc[cid], xc, xx)
    pot = float(closest.sum())
    for _ in range(1, k):
        rand_vals = tape[t:t + trials] * pot
        t += trials
        cand = np.searchsorted(np.cumsum(closest), rand_vals)
        np.clip(cand, None, n - 1, out=cand)
        dc = np.stack([np.minimum(closest, _dist(xc[c], xc, xx)) for c in cand])
        pots = dc.sum(axis=1)
        best = int(np.argmin(pots))
        pot = float(pots[best])
        closest = dc[best]
        centers.append(xc[cand[best]])
    return np.array(centers)


def lloyd(xc, centers, tol):
    """sklearn.cluster._kmeans._kmeans_single_lloyd (dense, unit weights), one feature.  Returns labels."""
    n, k = len(xc), len(centers)
    centers = centers.copy()
    labels_old = np.full(n, -1)
    strict = False

    def assign(c):
        pw = c * c + (-2.0) * (xc[:, None] * c[None, :])
        return np.argmin(pw, axis=1)   # first minimum, like the strict `<` scan

    for _ in range(KM_MAX_ITER):
        labels = assign(centers)
        w = np.bincount(labels, minlength=k).astype(np.float64)
        s = np.bincount(labels, weights=xc, minlength=k)
        empty = np.flatnonzero(w == 0)
        if len(empty):   # _relocate_empty_clusters_dense: the points farthest from their centres found new clusters
            dist = (xc - centers[labels]) ** 2
            far = np.argsort(-dist, kind="stable")[:len(empty)]
            for e, f in zip(empty, far):
                s[labels[f]] -= xc[f]
                w[labels[f]] -= 1.0
                s[e] = xc[f]
                w[e] = 1.0
        new = s * (1.0 / w)
        shift = np.sqrt((new - centers) ** 2)
        centers = new
        if np.array_equal(labels, labels_old):
            strict = True
            break
        if (shift ** 2).sum() <= tol:
            break
        labels_old = labels
    if not strict:
        labels = assign(centers)
    return labels


def kmeans_labels(x, k, tape):
    """KMeans(n_clusters=k, n_init=1).fit(x).labels_ with the draws of `tape`."""
    mean = x.mean()
    xc = x - mean
    tol = float(np.var(x)) * KM_TOL
    return lloyd(xc, kmeans_plusplus(xc, k, tape), tol)


# How the M step's column sums are taken: "pairwise" (numpy's reduction, the default: what the restatement is pinned to scikit-learn
# with), "sequential" (left to right) or "exact" (math.fsum: correctly rounded).  All three are evaluations of the same procedure in
# binary64; where they select different component counts for a row, the row's fit depends on the summation order -- in scikit-learn
# too, whose sums are BLAS reductions (tests/golden/make_refit_tie_rows.py lists those rows).
SUM_MODE = "pairwise"


def _colsum(a):
    if SUM_MODE == "pairwise":
        return a.sum(axis=0)
    if SUM_MODE == "sequential":
        return np.cumsum(a, axis=0)[-1]
    import math
    return np.array([math.fsum(a[:, j]) for j in range(a.shape[1])])


def _params(x, resp, full):
    nk = _colsum(resp) + EPS10
    means = _colsum(resp * x[:, None]) / nk
    if full:
        d = x[:, None] - means[None, :]
        cov = _colsum(resp * d * d) / nk + REG_COVAR
    else:
        cov = _colsum(resp * (x * x)[:, None]) / nk - means ** 2 + REG_COVAR
    return nk, means, cov


def _log_prob(x, w, means, pc, full):
    if full:
        y = x[:, None] * pc[None, :] - (means * pc)[None, :]
        lp = y * y
    else:
        prec = pc ** 2
        lp = (means ** 2 * prec)[None, :] - 2.0 * (x[:, None] * (means * prec)[None, :]) + (x * x)[:, None] * prec[None, :]
    return -0.5 * (LOG2PI + lp) + np.log(pc)[None, :] + np.log(w)[None, :]


def _logsumexp(a):
    m = a.max(axis=1)
    return m + np.log(np.exp(a - m[:, None]).sum(axis=1))


def gmm_fit(x, k, tape, full):
    """GaussianMixture(k, covariance_type = "full" if full else "diag").fit(x) with the draws of `tape`.
    Returns (weights, means, covariances, precisions_cholesky, mean log-likelihood under the final parameters) or None when
    scikit-learn raises ValueError (a covariance <= 0: the reference skips that component count, V3:777-779)."""
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    labels = kmeans_labels(x, k, tape)
    resp = np.zeros((n, k))
    resp[np.arange(n), labels] = 1.0
    nk, means, cov = _params(x, resp, full)
    w = nk / n
    if np.any(cov <= 0.0):
        return None
    pc = 1.0 / np.sqrt(cov)
    lower = -np.inf
    for _ in range(MAX_ITER):
        prev = lower
        wl = _log_prob(x, w, means, pc, full)
        lpn = _logsumexp(wl)
        resp = np.exp(wl - lpn[:, None])
        nk, means, cov = _params(x, resp, full)
        w = nk / nk.sum()
        if np.any(cov <= 0.0):
            return None
        pc = 1.0 / np.sqrt(cov)
        lower = float(lpn.mean())
        if abs(lower - prev) < TOL:
            break
    score = float(_logsumexp(_log_prob(x, w, means, pc, full)).mean())
    return w, means, cov, pc, score


def bic(score, n, k):
    return -2.0 * score * n + (3 * k - 1) * np.log(n)


def fit_edge(x, tape):
    """V3:764-786 for one edge.  `tape`: the uniforms the model-selection fits draw, n = 1, 2, ... in turn
    (draws_per_row(max_n) doubles).  Returns (n_selected, params[5, 3] = weight, mean, precision_cholesky)."""
    x = np.asarray(x, dtype=np.float64)
    out = np.zeros((MAX_COMP, 3))
    if len(x) == 0:
        return 0, out
    max_n = min(len(np.unique(x)), MAX_COMP)
    t, bics, ns = 0, [], []
    for k in range(1, max_n + 1):
        m = gmm_fit(x, k, tape[t:t + draws_per_fit(k)], full=False)
        t += draws_per_fit(k)
        if m is not None:
            bics.append(bic(m[4], len(x), k))
            ns.append(k)
    n_sel = ns[int(np.argmin(bics))]
    w, means, _, pc, _ = gmm_fit(x, n_sel, refit_tape(n_sel), full=True)
    out[:n_sel, 0], out[:n_sel, 1], out[:n_sel, 2] = w, means, pc
    return n_sel, out
