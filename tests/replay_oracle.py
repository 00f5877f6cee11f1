"""numpy restatement of the evidence recursion over a death sequence -- TEST INFRASTRUCTURE ONLY (the checker of the
device merge pchip_merge_records and of the engine's own evidence; never imported by the product).

update_evidence / calculate_logZ_estimate (src/polychord/run_time_info.f90:211-296, 652-678) for one volume with any
number of live points n(L) = #{entry contour < L} - #{deaths before}; pinned against the reference by
tests/golden/ref_replay.json (a dead-birth file the reference wrote and the evidence its .stats reports)."""
import numpy as np


def replay(logL, entry, rows=None, p0=0, nP=0):
    """-> dict(logZ, varlogZ, logweights (death order), nlive[, post_mean, post_var]); logL need not be sorted"""
    logL = np.asarray(logL, dtype=np.float64); entry = np.asarray(entry, dtype=np.float64)
    order = np.argsort(logL, kind="stable")
    d = logL[order]
    b = np.sort(entry)
    n = (np.searchsorted(b, d, side="left") - np.arange(d.size)).astype(np.float64)
    n = np.maximum(n, 1.0)
    l0, l1, l2 = np.log(n), np.log(n + 1.0), np.log(n + 2.0)
    logX = np.concatenate(([0.0], np.cumsum(l0 - l1)))
    logXX = np.concatenate(([0.0], np.cumsum(l0 - l2)))
    Xm, XXm, Xi = logX[:-1], logXX[:-1], logX[1:]
    logZ = np.logaddexp.reduce(Xm + d - l1)
    t = XXm + d + l0 - l1 - l2 - Xi
    ZX = np.logaddexp.accumulate(t) + Xi
    ZXm = np.concatenate(([-np.inf], ZX[:-1]))
    log2 = np.log(2.0)
    logZ2 = np.logaddexp.reduce(np.logaddexp(log2 + ZXm + d - l1, log2 + XXm + 2 * d - l1 - l2))
    out = dict(logZ=float(2 * logZ - 0.5 * logZ2), varlogZ=float(logZ2 - 2 * logZ), logweights=Xm - l1, nlive=n.astype(np.int64), order=order)
    if rows is not None:
        w = np.exp(Xm - l1 + d - (Xm - l1 + d).max())
        x = np.asarray(rows)[order][:, p0:p0 + nP]
        mean = (w[:, None] * x).sum(0) / w.sum()
        out["post_mean"] = mean
        out["post_var"] = (w[:, None] * x * x).sum(0) / w.sum() - mean ** 2
    return out


def evidence_replay(logL, entry):
    r = replay(logL, entry)
    return r["logZ"], r["varlogZ"]


def lived_records(run):
    """(logL, entry contour) of the points of a run that entered the live set"""
    dead, lw = run["dead"], run["logweights"]
    keep = lw > -1e29
    entry = run["entry"] if "entry" in run else dead[:, -2]
    return dead[keep, -1], entry[keep]


def combined_evidence(run_logZ, run_varlogZ):
    """evidence of R independent runs from the runs' own log-normal evidences (pchip_merged.evidence_rule 1): mean of the Z_r in linear
    space; variance of its log = the larger of the propagated one and the scatter between the runs.  -> (logZ, varlogZ)"""
    lz = np.asarray(run_logZ, dtype=np.float64); v = np.asarray(run_varlogZ, dtype=np.float64)
    R = lz.size
    m = np.exp(lz + 0.5 * v)                 # <Z_r>
    q = np.exp(2.0 * lz + 2.0 * v)           # <Z_r^2>
    mean = m.mean()
    second = (q.sum() + m.sum() ** 2 - (m ** 2).sum()) / R ** 2
    var = np.log(second) - 2.0 * np.log(mean)
    if R > 1:
        var = max(var, np.log1p(m.var(ddof=1) / R / mean ** 2))
    var = max(var, 0.0)
    return float(np.log(mean) - 0.5 * var), float(var)
