"""Repeat-sharded runs: the exchange step between GPUs.

Every rank runs an independent nested-sampling run (own seed) on its own MI355X.  The runs are statistically
combinable: gather the dead points of every run -- the full rows (cube, theta, phi, birth, logL) and the contour at
which each point entered its live set -- and the union is a nested-sampling run with n(L) = sum of the runs' live
points.  Between processes the gather is ONE padded RCCL all-gather over xGMI (torch.distributed backend "nccl": counts
first, then rows; `gloo` on CPU in the tests); the merge itself -- evidence recursion of run_time_info.f90:211-296 and
:652-678 over the merged death sequence, posterior weights and moments -- runs on the device in the library
(pchip_merge_records, csrc/pc_merge.hip).  There is no CPU merge in the product: without a HIP device it fails loudly.
"""
import ctypes as C

import numpy as np

from . import _ctypes_api as api


class Merged(C.Structure):
    _fields_ = [("logZ", C.c_double), ("varlogZ", C.c_double), ("n", C.c_long), ("nTotal", C.c_int), ("nruns", C.c_int),
                ("rows", C.POINTER(C.c_double)), ("logweights", C.POINTER(C.c_double)), ("nlive", C.POINTER(C.c_int)),
                ("post_mean", C.POINTER(C.c_double)), ("post_var", C.POINTER(C.c_double)),
                ("t_merge_s", C.c_double), ("t_runs_s", C.c_double), ("nlike", C.c_long), ("ndead_all", C.c_long)]


def _lib():
    lib = api.load()
    if not getattr(lib, "_merge_bound", False):
        lib.pchip_merge_records.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_long), C.c_void_p, C.c_void_p, C.c_int,
                                            C.c_int, C.POINTER(Merged)]
        lib.pchip_merge_records.restype = C.c_int
        lib.pchip_merged_free.argtypes = [C.POINTER(Merged)]
        lib.pchip_merged_free.restype = None
        lib.pchip_merged_write.argtypes = [C.POINTER(Merged), C.c_int, C.c_int, C.c_char_p, C.c_char_p]
        lib.pchip_merged_write.restype = C.c_int
        lib._merge_bound = True
    return lib


def lived_records(run):
    """(rows, entry) of the points that entered the live set, in the order they died (failed spawns carry logweight =
    logzero and are no part of the run's death sequence).  entry = the contour when the point joined the live set, not
    the birth column: with B > 1 chains per nursery a baby is born under an older contour than the one it replaces a
    point at."""
    dead, lw = run["dead"], run["logweights"]
    keep = lw > -1e29
    entry = run["entry"] if "entry" in run else dead[:, -2]
    return np.ascontiguousarray(dead[keep]), np.ascontiguousarray(entry[keep])


def merged_dict(m, nDims, nDerived, want_rows):
    n, nT, nP = m.n, m.nTotal, nDims + nDerived
    out = {"n_runs": m.nruns, "logZ": m.logZ, "varlogZ": m.varlogZ, "logZerr": float(np.sqrt(abs(m.varlogZ))), "records": int(n),
           "post_mean": np.ctypeslib.as_array(m.post_mean, shape=(nP,)).copy(),
           "post_var": np.ctypeslib.as_array(m.post_var, shape=(nP,)).copy(),
           "logweights": np.ctypeslib.as_array(m.logweights, shape=(max(n, 1),))[:n].copy(),
           "nlive": np.ctypeslib.as_array(m.nlive, shape=(max(n, 1),))[:n].copy(),
           "t_merge_s": m.t_merge_s, "t_runs_s": m.t_runs_s, "nlike": int(m.nlike), "ndead_all": int(m.ndead_all)}
    if want_rows and n > 0:
        out["rows"] = np.ctypeslib.as_array(m.rows, shape=(n, nT)).copy()
    return out


def merge_records(nDims, nDerived, counts, rows, entry, on_device=False, want_rows=False, write=None):
    """pchip_merge_records -> dict.  rows / entry: host numpy arrays (uploaded by the library), or -- on_device -- integer
    device addresses of the gathered buffers.  write = (base_dir, file_root): also <root>.stats / _dead-birth.txt / .txt."""
    lib = _lib()
    cnt = (C.c_long * len(counts))(*[int(c) for c in counts])
    m = Merged()
    if on_device:
        rp, ep = C.c_void_p(int(rows)), C.c_void_p(int(entry))
    else:
        rows = np.ascontiguousarray(rows, dtype=np.float64); entry = np.ascontiguousarray(entry, dtype=np.float64)
        rp, ep = rows.ctypes.data_as(C.c_void_p), entry.ctypes.data_as(C.c_void_p)
    want = 1 if (want_rows or write) else 0
    rc = lib.pchip_merge_records(nDims, nDerived, len(counts), cnt, rp, ep, 1 if on_device else 0, want, C.byref(m))
    if rc != 0:
        raise RuntimeError(f"pchip_merge_records failed with code {rc}")
    try:
        if write:
            if lib.pchip_merged_write(C.byref(m), nDims, nDerived, str(write[0]).encode(), str(write[1]).encode()) != 0:
                raise RuntimeError("pchip_merged_write failed")
        return merged_dict(m, nDims, nDerived, want_rows)
    finally:
        lib.pchip_merged_free(C.byref(m))


def gather_records(run, dist, torch, device):
    """all-gather of the lived records of every rank: counts first, then ONE padded [nmax][nTotal + 1] buffer per rank
    (rows | entry contour).  Returns (gathered [sum counts][nTotal + 1] tensor on `device`, counts).  dist = None: this
    rank's records alone."""
    # the run's arrays go to the device whole (views of the engine's pinned result buffers: one DMA each) and the points that
    # lived are picked there -- a boolean-mask copy of 30 MB on the host cost more than the merge itself
    dead = torch.from_numpy(run["dead"]).to(device, non_blocking=True)
    lw = torch.from_numpy(run["logweights"]).to(device, non_blocking=True)
    entry = (torch.from_numpy(run["entry"]) if "entry" in run else torch.from_numpy(run["dead"][:, -2].copy())).to(device, non_blocking=True)
    keep = lw > -1e29
    rec = torch.cat([dead[keep], entry[keep][:, None]], dim=1)
    if dist is None:
        return rec.contiguous(), [int(rec.shape[0])]
    world = dist.get_world_size()
    cnt = torch.tensor([rec.shape[0]], dtype=torch.int64, device=device)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt)
    ks = [int(c.item()) for c in cnts]
    nmax = max(max(ks), 1)
    pad = torch.zeros((nmax, rec.shape[1]), dtype=torch.float64, device=device)
    pad[:rec.shape[0]] = rec
    bufs = torch.empty((world * nmax, rec.shape[1]), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(bufs, pad)               # RCCL all-gather over xGMI (gloo: same call on CPU tensors)
    return torch.cat([bufs[r * nmax:r * nmax + k] for r, k in enumerate(ks)]).contiguous(), ks


def merge_runs(run, dist, torch, local_rank, nDims, nDerived, want_rows=False, write=None):
    """this rank's run + everybody else's -> the merged result (every rank computes it, like an all-reduce)."""
    on_gpu = torch is not None and torch.cuda.is_available()
    if not on_gpu:
        if dist is not None:
            raise RuntimeError("merge_runs between processes needs the GPUs (backend nccl); gather_records is the part that also runs on gloo")
        rows, entry = lived_records(run)
        return merge_records(nDims, nDerived, [rows.shape[0]], rows, entry, want_rows=want_rows, write=write)
    dev = torch.device("cuda", local_rank)
    g, ks = gather_records(run, dist, torch, dev)
    nT = g.shape[1] - 1
    rows = g[:, :nT].contiguous(); entry = g[:, nT].contiguous()
    torch.cuda.synchronize(dev)
    return merge_records(nDims, nDerived, ks, rows.data_ptr(), entry.data_ptr(), on_device=True, want_rows=want_rows, write=write)
