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


def evidence_replay_torch(torch, logL, birth):
    """evidence_replay with torch ops on whatever device the records live on (the RCCL all-gather leaves
    them in HBM; the sort, the counting and the log-space scans run there)."""
    d, _ = torch.sort(logL.to(torch.float64))
    b, _ = torch.sort(birth.to(torch.float64))
    n = (torch.searchsorted(b, d, right=False) - torch.arange(d.numel(), device=d.device)).to(torch.float64)
    n = torch.clamp(n, min=1.0)
    l0, l1, l2 = torch.log(n), torch.log(n + 1.0), torch.log(n + 2.0)
    zero = torch.zeros(1, dtype=torch.float64, device=d.device)
    logX = torch.cat((zero, torch.cumsum(l0 - l1, 0)))
    logXX = torch.cat((zero, torch.cumsum(l0 - l2, 0)))
    Xm, XXm, Xi = logX[:-1], logXX[:-1], logX[1:]
    logZ = torch.logsumexp(Xm + d - l1, 0)
    ZX = torch.logcumsumexp(XXm + d + l0 - l1 - l2 - Xi, 0) + Xi
    ZXm = torch.cat((torch.full((1,), -float("inf"), dtype=torch.float64, device=d.device), ZX[:-1]))
    log2 = 0.6931471805599453
    logZ2 = torch.logsumexp(torch.logaddexp(log2 + ZXm + d - l1, log2 + XXm + 2 * d - l1 - l2), 0)
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
    on_gpu = torch is not None and torch.cuda.is_available()
    if dist is None:
        if on_gpu:
            dev = torch.device("cuda", local_rank)
            lz, var = evidence_replay_torch(torch, torch.from_numpy(np.ascontiguousarray(logL)).to(dev),
                                            torch.from_numpy(np.ascontiguousarray(birth)).to(dev))
        else:
            lz, var = evidence_replay(logL, birth)
        return {"n_runs": 1, "logZ": lz, "logZerr": float(np.sqrt(abs(var))), "records": int(logL.size)}
    world = dist.get_world_size()
    dev = torch.device("cuda", local_rank) if on_gpu and dist.get_backend() == "nccl" else torch.device("cpu")
    cnt = torch.tensor([logL.size], dtype=torch.int64, device=dev)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt)
    ks = [int(c.item()) for c in cnts]
    nmax = max(ks)
    rec = torch.zeros((nmax, 2), dtype=torch.float64, device=dev)
    rec[:logL.size, 0] = torch.from_numpy(np.ascontiguousarray(logL)).to(dev)
    rec[:logL.size, 1] = torch.from_numpy(np.ascontiguousarray(birth)).to(dev)
    recs = [torch.empty_like(rec) for _ in range(world)]
    dist.all_gather(recs, rec)                            # RCCL all-gather over xGMI
    allL = torch.cat([r[:k, 0] for k, r in zip(ks, recs)])
    allB = torch.cat([r[:k, 1] for k, r in zip(ks, recs)])
    lz, var = evidence_replay_torch(torch, allL, allB)
    return {"n_runs": world, "logZ": lz, "logZerr": float(np.sqrt(abs(var))), "records": int(sum(ks))}
