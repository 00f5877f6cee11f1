"""Repeat-sharded multi-GPU merge.

Every rank runs an independent nested-sampling run (own seed) on its own MI355X.  The runs are
statistically combinable: all-gather the (logL, birth) records of every point that ever lived,
count the live points n(L) = #born below - #died before at each death, and run the reference's
evidence recursion (src/polychord/run_time_info.f90:211-296, 652-678) over the merged sequence.
The collective is one RCCL all-gather over xGMI (torch.distributed backend "nccl"); the recursion
is a handful of vectorised log-space scans on the host.
"""
import numpy as np


def evidence_replay(logL, birth):
    """(logZ, var(logZ)) of the merged death sequence; single-cluster recursion, any n(L)."""
    d = np.sort(np.asarray(logL, dtype=np.float64))
    b = np.sort(np.asarray(birth, dtype=np.float64))
    n = (np.searchsorted(b, d, side="left") - np.arange(d.size)).astype(np.float64)
    n = np.maximum(n, 1.0)
    l0, l1, l2 = np.log(n), np.log(n + 1.0), np.log(n + 2.0)
    # log X_{i-1}, log XX_{i-1} (volumes BEFORE death i)
    logX = np.concatenate(([0.0], np.cumsum(l0 - l1)))
    logXX = np.concatenate(([0.0], np.cumsum(l0 - l2)))
    Xm, XXm, Xi = logX[:-1], logXX[:-1], logX[1:]
    logZ = np.logaddexp.reduce(Xm + d - l1)
    # ZX_i / X_i = ZX_{i-1} / X_{i-1} + XX_{i-1} L_i n/((n+1)(n+2)) / X_i
    t = XXm + d + l0 - l1 - l2 - Xi
    zx_over_x = np.logaddexp.accumulate(t)
    ZX = zx_over_x + Xi                                   # ZX after death i
    ZXm = np.concatenate(([-np.inf], ZX[:-1]))            # before death i
    log2 = np.log(2.0)
    logZ2 = np.logaddexp.reduce(np.logaddexp(log2 + ZXm + d - l1, log2 + XXm + 2 * d - l1 - l2))
    return float(2 * logZ - 0.5 * logZ2), float(logZ2 - 2 * logZ)


def lived_records(run):
    """(logL, birth) of the points that entered the live set (failed spawns carry weight logzero)."""
    dead, lw = run["dead"], run["logweights"]
    keep = lw > -1e29
    # the ENTRY contour (contour when the point joined the live set), not the birth column: with B > 1
    # chains per nursery a baby is born under an older contour than the one it replaces a point at.
    entry = run["entry"] if "entry" in run else dead[:, -2]
    return dead[keep, -1], entry[keep]


def merge_runs(run, dist, torch, local_rank):
    logL, birth = lived_records(run)
    if dist is None:
        lz, var = evidence_replay(logL, birth)
        return {"n_runs": 1, "logZ": lz, "logZerr": float(np.sqrt(abs(var))), "records": int(logL.size)}
    world = dist.get_world_size()
    dev = torch.device("cuda", local_rank) if torch.cuda.is_available() and dist.get_backend() == "nccl" else torch.device("cpu")
    cnt = torch.tensor([logL.size], dtype=torch.int64, device=dev)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt)
    nmax = int(max(int(c.item()) for c in cnts))
    rec = torch.full((nmax, 2), float("nan"), dtype=torch.float64, device=dev)
    rec[:logL.size, 0] = torch.from_numpy(np.ascontiguousarray(logL)).to(dev)
    rec[:logL.size, 1] = torch.from_numpy(np.ascontiguousarray(birth)).to(dev)
    recs = [torch.empty_like(rec) for _ in range(world)]
    dist.all_gather(recs, rec)                            # RCCL all-gather over xGMI
    allL, allB = [], []
    for c, r in zip(cnts, recs):
        k = int(c.item())
        a = r[:k].cpu().numpy()
        allL.append(a[:, 0]); allB.append(a[:, 1])
    lz, var = evidence_replay(np.concatenate(allL), np.concatenate(allB))
    return {"n_runs": world, "logZ": lz, "logZerr": float(np.sqrt(abs(var))), "records": int(sum(int(c.item()) for c in cnts))}
