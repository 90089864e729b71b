"""Seeded synthetic designs for the five BASELINE.json configs.

There is no R in the image, so these stand in for what
``Covariance$get_D_data()`` / ``MeanFunction$X`` / ``Model$sim_data()`` hand to
the Rcpp exports (R6ModelExtMCML.R:399-419): the (cov, data, eff_range) triple,
dense column-major Z and X, y, and the start vector c(beta, theta, sigma|1).

cov is int32 rows x 5 = (block id, block dim, function id, n variables,
parameter index) (src/mcml_optim.cpp:20-21); data is each block's data matrix
flattened column-major and concatenated (:22).  Function ids follow the build's
table: 1 gr, 3 ar1, 7 fexp.
"""
import numpy as np

FN_GR, FN_FEXP0, FN_AR1, FN_SQEXP, FN_FEXP, FN_SQEXP0 = 1, 2, 3, 4, 7, 14


def _fexp_D(xy, theta):
    d = np.sqrt(((xy[:, None, :] - xy[None, :, :]) ** 2).sum(-1))
    return theta[0] * np.exp(-d / theta[1])


def geospatial(n, seed=20240601, theta=(0.25, 0.1), sigma=1.0, beta0=1.0):
    """configs 2/3: gaussian-identity, ~(1|fexp(x,y)), Z = I, X = 1."""
    rng = np.random.default_rng(seed)
    xy = rng.random((n, 2))
    cov = np.array([[0, n, FN_FEXP, 2, 0]], dtype=np.int32, order="F")
    data = np.concatenate([xy[:, 0], xy[:, 1]])
    D = _fexp_D(xy, theta)
    L = np.linalg.cholesky(D)
    y = beta0 + L @ rng.standard_normal(n) + sigma * rng.standard_normal(n)
    return dict(cov=cov, data=data, eff_range=np.zeros(1), Z=np.eye(n, order="F"),
                X=np.ones((n, 1), order="F"), y=y, family="gaussian", link="identity",
                start=np.array([beta0, theta[0], theta[1], sigma]), theta=np.array(theta),
                beta=np.array([beta0]), sigma=sigma, n=n, Q=n, P=1)


def _indicator(n, Q, cols_per_row):
    Z = np.zeros((n, Q), order="F")
    for cols in cols_per_row:
        Z[np.arange(n), cols] = 1.0
    return Z


def cluster_rct(ncl=10, nt=5, nind=10, seed=20240602, theta=(0.25, 0.10), family="binomial"):
    """config 1: ~(1|gr(cl))+(1|gr(cl,t)), X = int + factor(t) - 1."""
    rng = np.random.default_rng(seed)
    cl = np.repeat(np.arange(ncl), nt * nind)
    t = np.tile(np.repeat(np.arange(nt), nind), ncl)
    n = cl.size
    rows, data = [], []
    for c in range(ncl):
        rows.append([c, 1, FN_GR, 1, 0]); data.append([float(c + 1)])
    for c in range(ncl):
        for tt in range(nt):
            rows.append([ncl + c * nt + tt, 1, FN_GR, 2, 1]); data.append([float(c + 1), float(tt + 1)])
    cov = np.array(rows, dtype=np.int32, order="F")
    data = np.concatenate([np.asarray(d) for d in data])
    Q = ncl + ncl * nt
    Z = _indicator(n, Q, [cl, ncl + cl * nt + t])
    X = np.zeros((n, 1 + nt), order="F")
    X[:, 0] = (cl >= ncl // 2).astype(float)
    X[np.arange(n), 1 + t] = 1.0
    beta = np.concatenate([[0.5], rng.standard_normal(nt)])
    u = np.concatenate([theta[0] * rng.standard_normal(ncl), theta[1] * rng.standard_normal(ncl * nt)])
    eta = X @ beta + Z @ u
    if family == "binomial":
        y = (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(float); link = "logit"
    else:
        y = rng.poisson(np.exp(eta)).astype(float); link = "log"
    return dict(cov=cov, data=data, eff_range=np.zeros(len(rows)), Z=Z, X=X, y=y, family=family,
                link=link, start=np.concatenate([beta, theta, [1.0]]), theta=np.array(theta),
                beta=beta, sigma=1.0, n=n, Q=Q, P=1 + nt)


def stepped_wedge(ncl=40, nt=8, nind=50, seed=20240603, theta=(0.25, 0.8)):
    """config 4: binomial-logit, ~(1|gr(cl)*ar1(t)): ncl blocks of dim nt."""
    rng = np.random.default_rng(seed)
    cl = np.repeat(np.arange(ncl), nt * nind)
    t = np.tile(np.repeat(np.arange(nt), nind), ncl)
    n = cl.size
    rows, data = [], []
    for c in range(ncl):
        rows.append([c, nt, FN_GR, 1, 0]); rows.append([c, nt, FN_AR1, 1, 1])
        data.append(np.concatenate([np.full(nt, c + 1.0), np.arange(1.0, nt + 1.0)]))
    cov = np.array(rows, dtype=np.int32, order="F")
    data = np.concatenate(data)
    Q = ncl * nt
    Z = _indicator(n, Q, [cl * nt + t])
    X = np.zeros((n, 1 + nt), order="F")
    X[:, 0] = (t >= (cl % (nt - 1)) + 1).astype(float)      # staggered roll-out
    X[np.arange(n), 1 + t] = 1.0
    beta = np.concatenate([[0.5], 0.3 * rng.standard_normal(nt)])
    dt = np.abs(np.arange(nt)[:, None] - np.arange(nt)[None, :])
    Lb = np.linalg.cholesky(theta[0] ** 2 * theta[1] ** dt)
    u = np.concatenate([Lb @ rng.standard_normal(nt) for _ in range(ncl)])
    eta = X @ beta + Z @ u
    y = (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(float)
    return dict(cov=cov, data=data, eff_range=np.zeros(len(rows)), Z=Z, X=X, y=y, family="binomial",
                link="logit", start=np.concatenate([beta, theta, [1.0]]), theta=np.array(theta),
                beta=beta, sigma=1.0, n=n, Q=Q, P=1 + nt)


def longitudinal(nsubj=2000, nvisit=10, seed=20240604, theta=(0.5, 0.2)):
    """config 5: poisson-log, (1|gr(subj))+(1|gr(subj,visit)); all blocks dim 1."""
    rng = np.random.default_rng(seed)
    subj = np.repeat(np.arange(nsubj), nvisit)
    vis = np.tile(np.arange(nvisit), nsubj)
    n = subj.size
    rows = [[s, 1, FN_GR, 1, 0] for s in range(nsubj)]
    data = [np.array([s + 1.0]) for s in range(nsubj)]
    for s in range(nsubj):
        for v in range(nvisit):
            rows.append([nsubj + s * nvisit + v, 1, FN_GR, 2, 1]); data.append(np.array([s + 1.0, v + 1.0]))
    cov = np.array(rows, dtype=np.int32, order="F")
    data = np.concatenate(data)
    Q = nsubj + nsubj * nvisit
    Z = _indicator(n, Q, [subj, nsubj + subj * nvisit + vis])
    X = np.zeros((n, 2), order="F")
    X[:, 0] = 1.0
    X[:, 1] = vis / float(nvisit)
    beta = np.array([0.0, 0.3])
    u = np.concatenate([theta[0] * rng.standard_normal(nsubj), theta[1] * rng.standard_normal(nsubj * nvisit)])
    y = rng.poisson(np.exp(X @ beta + Z @ u)).astype(float)
    return dict(cov=cov, data=data, eff_range=np.zeros(len(rows)), Z=Z, X=X, y=y, family="poisson",
                link="log", start=np.concatenate([beta, theta, [1.0]]), theta=np.array(theta),
                beta=beta, sigma=1.0, n=n, Q=Q, P=2)
