"""Repeat-sharded runs: the exchange step between GPUs.

Every rank runs an independent nested-sampling run (own seed) on its own MI355X.  The runs are statistically
combinable: gather the dead points of every run -- the full rows (cube, theta, phi, birth, logL) and the contour at
which each point entered its live set -- and the union is a nested-sampling run with n(L) = sum of the runs' live
points.  Between processes the gather is ONE padded RCCL all-gather over xGMI (counts first, then rows), made INSIDE the
library (pchip_comm_merge, csrc/pc_merge.hip: librccl.so resolved with dlopen, the lived records picked on the device,
ncclAllGather, device merge); `Comm` below is its handle, and the only thing it needs from the launcher is a way to hand
128 bytes from rank 0 to the others (here: the torch.distributed store the ranks were started with).  The merge itself
-- evidence recursion of run_time_info.f90:211-296 and :652-678 over the merged death sequence, posterior weights and
moments -- runs on the device (pchip_merge_records).  `gather_records` is the same exchange on torch tensors: the CPU /
gloo transport of the tests (two ranks that share one GPU cannot form an RCCL communicator).  There is no CPU merge in
the product: without a HIP device it fails loudly.
"""
import ctypes as C

import numpy as np

from . import _ctypes_api as api


class Merged(C.Structure):
    _fields_ = [("logZ", C.c_double), ("varlogZ", C.c_double), ("n", C.c_long), ("nTotal", C.c_int), ("nruns", C.c_int),
                ("rows", C.POINTER(C.c_double)), ("logweights", C.POINTER(C.c_double)), ("nlive", C.POINTER(C.c_int)),
                ("post_mean", C.POINTER(C.c_double)), ("post_var", C.POINTER(C.c_double)),
                ("t_merge_s", C.c_double), ("t_runs_s", C.c_double), ("nlike", C.c_long), ("ndead_all", C.c_long),
                ("runs_logZ_mean", C.c_double), ("runs_logZ_sem", C.c_double),
                ("logZ_replay", C.c_double), ("varlogZ_replay", C.c_double), ("evidence_rule", C.c_int), ("nclustered", C.c_int)]


def _lib():
    lib = api.load()
    if not getattr(lib, "_merge_bound", False):
        lib.pchip_merge_records.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_long), C.c_void_p, C.c_void_p, C.c_int,
                                            C.c_int, C.POINTER(Merged)]
        lib.pchip_merge_records.restype = C.c_int
        lib.pchip_merge_records_ex.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_long), C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(Merged)]
        lib.pchip_merge_records_ex.restype = C.c_int
        lib.pchip_merged_free.argtypes = [C.POINTER(Merged)]
        lib.pchip_merged_free.restype = None
        lib.pchip_merged_write.argtypes = [C.POINTER(Merged), C.c_int, C.c_int, C.c_char_p, C.c_char_p]
        lib.pchip_merged_write.restype = C.c_int
        lib.pchip_comm_get_id.argtypes = [C.c_char_p]
        lib.pchip_comm_get_id.restype = C.c_int
        lib.pchip_comm_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        lib.pchip_comm_create.restype = C.c_int
        lib.pchip_comm_destroy.argtypes = [C.c_void_p]
        lib.pchip_comm_destroy.restype = None
        lib.pchip_comm_merge.argtypes = [C.c_void_p, C.POINTER(api.Result), C.c_double, C.c_int, C.c_int, C.c_int, C.POINTER(Merged)]
        lib.pchip_comm_merge.restype = C.c_int
        lib.pchip_comm_merge_many.argtypes = [C.c_void_p, C.POINTER(api.Result), C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.POINTER(Merged)]
        lib.pchip_comm_merge_many.restype = C.c_int
        lib.pchip_comm_library.argtypes = []
        lib.pchip_comm_library.restype = C.c_char_p
        if lib.pchip_sizeof(b"merged") != C.sizeof(Merged):
            raise ImportError("pchip_merged is %d bytes in the library, %d in this binding" % (lib.pchip_sizeof(b"merged"), C.sizeof(Merged)))
        lib._merge_bound = True
    return lib


def _logzero(run, logzero):
    """the run's own logzero (settings.logzero travels with the result dict); -1e30 is the reference's default"""
    if logzero is not None:
        return float(logzero)
    return float(run.get("logzero", -1e30))


def lived_records(run, logzero=None):
    """(rows, entry) of the points that entered the live set, in the order they died (failed spawns carry logweight =
    logzero and are no part of the run's death sequence: the same test as the library's, logweight > logzero).  entry =
    the contour when the point joined the live set, not the birth column: with B > 1 chains per nursery a baby is born
    under an older contour than the one it replaces a point at."""
    dead, lw = run["dead"], run["logweights"]
    keep = lw > _logzero(run, logzero)
    entry = run["entry"] if "entry" in run else dead[:, -2]
    return np.ascontiguousarray(dead[keep]), np.ascontiguousarray(entry[keep])


def merged_dict(m, nDims, nDerived, want_rows):
    n, nT, nP = m.n, m.nTotal, nDims + nDerived
    arr = lambda p, k, dt: np.ctypeslib.as_array(p, shape=(k,)).copy() if (k > 0 and bool(p)) else np.zeros(0, dtype=dt)   # (a union without records has no arrays)
    out = {"n_runs": m.nruns, "logZ": m.logZ, "varlogZ": m.varlogZ, "logZerr": float(np.sqrt(abs(m.varlogZ))), "records": int(n),
           "post_mean": np.ctypeslib.as_array(m.post_mean, shape=(nP,)).copy(),
           "post_var": np.ctypeslib.as_array(m.post_var, shape=(nP,)).copy(),
           "logweights": arr(m.logweights, n, np.float64),
           "nlive": arr(m.nlive, n, np.int32),
           "t_merge_s": m.t_merge_s, "t_runs_s": m.t_runs_s, "nlike": int(m.nlike), "ndead_all": int(m.ndead_all),
           "runs_logZ_mean": m.runs_logZ_mean, "runs_logZ_sem": m.runs_logZ_sem,
           # which evidence logZ is (include/polychord_hip.h pchip_merged): 0 = the replay of the union (every run ended with one cluster),
           # 1 = the runs' own evidences combined in linear space + the runs' own weights (a run had clusters); the replay beside it
           "evidence_rule": int(m.evidence_rule), "nclustered": int(m.nclustered), "logZ_replay": m.logZ_replay,
           "varlogZ_replay": m.varlogZ_replay, "logZerr_replay": float(np.sqrt(abs(m.varlogZ_replay)))}
    if want_rows and n > 0:
        out["rows"] = np.ctypeslib.as_array(m.rows, shape=(n, nT)).copy()
    return out


def clustered(run):
    """did this run ever hold more than one cluster?  (then its dead points carry cluster-volume weights the union's replay does not know:
    pchip_merged.evidence_rule; pchip_result.ncluster_peak -- a record without it, e.g. the test oracle's: clusters alive at the end or dead
    before it)"""
    if "ncluster_peak" in run:
        return int(run["ncluster_peak"] > 1)
    return int(run["ncluster"] > 1 or run["ncluster_dead"] > 1)


def merge_records(nDims, nDerived, counts, rows, entry, on_device=False, want_rows=False, write=None, ownw=None, run_logZ=None,
                  run_varlogZ=None, run_clustered=None):
    """pchip_merge_records[_ex] -> dict.  rows / entry (/ ownw): host numpy arrays (uploaded by the library), or -- on_device -- integer
    device addresses of the gathered buffers.  ownw + run_logZ + run_varlogZ + run_clustered: what each run knows about itself (the
    records' own log weights, the runs' evidences, did it end with clusters) -- with them a union that holds a clustered run quotes the
    runs' own evidences and weights (evidence_rule 1).  write = (base_dir, file_root): also <root>.stats / _dead-birth.txt / .txt."""
    lib = _lib()
    own = (ownw, run_logZ, run_varlogZ, run_clustered)
    if any(x is not None for x in own) and not all(x is not None for x in own):
        # (some of the four: the _ex path would index None, and ownw alone would silently quote evidence_rule 0 for a clustered run)
        raise ValueError("merge_records: ownw, run_logZ, run_varlogZ and run_clustered go together (all four, or none)")
    if run_logZ is not None and not (len(run_logZ) == len(run_varlogZ) == len(run_clustered) == len(counts)):
        raise ValueError("merge_records: run_logZ / run_varlogZ / run_clustered need one entry per run")
    cnt = (C.c_long * len(counts))(*[int(c) for c in counts])
    m = Merged()
    if on_device:
        rp, ep, wp = C.c_void_p(int(rows)), C.c_void_p(int(entry)), (C.c_void_p(int(ownw)) if ownw is not None else None)
    else:
        rows = np.ascontiguousarray(rows, dtype=np.float64); entry = np.ascontiguousarray(entry, dtype=np.float64)
        rp, ep, wp = rows.ctypes.data_as(C.c_void_p), entry.ctypes.data_as(C.c_void_p), None
        if ownw is not None:
            ownw = np.ascontiguousarray(ownw, dtype=np.float64); wp = ownw.ctypes.data_as(C.c_void_p)
    want = 1 if (want_rows or write) else 0
    if ownw is not None and run_logZ is not None:
        R = len(counts)
        lz = (C.c_double * R)(*[float(x) for x in run_logZ]); vz = (C.c_double * R)(*[float(x) for x in run_varlogZ])
        cl = (C.c_int * R)(*[int(x) for x in run_clustered])
        rc = lib.pchip_merge_records_ex(nDims, nDerived, R, cnt, rp, ep, wp, lz, vz, cl, 1 if on_device else 0, want, C.byref(m))
    else:
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


class Comm:
    """RCCL communicator of the library (pchip_comm_*): one rank per GPU.  `exchange(id_bytes_or_None) -> id_bytes` hands
    rank 0's 128 bytes to everybody (default: torch.distributed's object broadcast over whatever backend the ranks were
    started with -- control plane only; the records travel over RCCL inside the library)."""

    def __init__(self, rank, world, device, exchange=None):
        lib = _lib()
        self.lib, self.rank, self.world, self.device, self.h = lib, rank, world, device, C.c_void_p()
        ident = C.create_string_buffer(128)
        # (rank 0 tells the others when it has no id to give: they must not wait in the broadcast / ncclCommInitRank for ever)
        failed = rank == 0 and lib.pchip_comm_get_id(ident) != 0
        if exchange is None:
            import torch.distributed as dist

            def exchange(b):
                box = [b]
                dist.broadcast_object_list(box, src=0)
                return box[0]
        raw = exchange((b"" if failed else ident.raw) if rank == 0 else None) if world > 1 else (b"" if failed else ident.raw)
        if not raw:
            raise RuntimeError("pchip_comm_get_id failed on rank 0 (librccl.so not loadable?)")
        if lib.pchip_comm_create(raw, world, rank, device, C.byref(self.h)) != 0:
            raise RuntimeError("pchip_comm_create failed")

    @property
    def library(self):
        p = self.lib.pchip_comm_library()
        return p.decode() if p else None

    def close(self):
        if self.h:
            self.lib.pchip_comm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_ulong)
_HIP_H2D, _HIP_D2H, _HIP_D2D = 1, 2, 3


def hip_runtime():
    """the HIP runtime the library itself runs on (ctypes): hipMemcpy / hipSetDevice for the all-gathers written in Python below"""
    hip = C.CDLL("libamdhip64.so", mode=C.RTLD_GLOBAL)
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpy.restype = C.c_int
    hip.hipSetDevice.argtypes = [C.c_int]
    return hip


class CallbackComm:
    """pchip_comm_create_with: the library's exchange (header, counts, status, ONE padded block per rank, unpad, merge -- pchip_comm_merge_many,
    every statement of it) over the CALLER's all-gather instead of RCCL.  all_gather(send_ptr, recv_ptr, nbytes) -> 0: every rank's nbytes of
    device memory at send_ptr into recv_ptr (device, world * nbytes, rank after rank) on all ranks.  Used like `Comm` (comm_merge,
    comm_merge_many, run_repeats(comm=...))."""

    def __init__(self, rank, world, device, all_gather):
        lib = _lib()
        self.lib, self.rank, self.world, self.device, self.h = lib, rank, world, device, C.c_void_p()

        def cb(user, send, recv, nbytes):
            try:
                return int(all_gather(send, recv, int(nbytes)) or 0)
            except Exception as e:      # noqa: BLE001 -- an exception must not cross the C frames: the library turns the code into its error
                import sys
                sys.stderr.write("CallbackComm: all_gather raised %s: %s\n" % (type(e).__name__, e))
                return 1
        self._cb = ALLGATHER_FN(cb)                                   # (kept alive as long as the communicator)
        lib.pchip_comm_create_with.argtypes = [ALLGATHER_FN, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        lib.pchip_comm_create_with.restype = C.c_int
        if lib.pchip_comm_create_with(self._cb, None, world, rank, device, C.byref(self.h)) != 0:
            raise RuntimeError("pchip_comm_create_with failed")

    library = "caller's all-gather"

    def close(self):
        if self.h:
            self.lib.pchip_comm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def host_all_gather(dist, torch, device_ordinal):
    """an all-gather for CallbackComm over torch.distributed on HOST tensors (gloo): device -> host, all_gather_into_tensor, host -> device.
    What lets `bench.py --gpus 2 --backend gloo` run two ranks on ONE GPU through the library's whole exchange path (RCCL forms no
    communicator with two ranks on a device); never the product path between GPUs."""
    hip = hip_runtime()

    def all_gather(send, recv, nbytes):
        world = dist.get_world_size()
        hip.hipSetDevice(device_ordinal)
        mine = torch.empty(nbytes, dtype=torch.uint8)
        if hip.hipMemcpy(mine.data_ptr(), send, nbytes, _HIP_D2H) != 0:
            return 2
        out = torch.empty(nbytes * world, dtype=torch.uint8)
        dist.all_gather_into_tensor(out, mine)
        return 0 if hip.hipMemcpy(recv, out.data_ptr(), nbytes * world, _HIP_H2D) == 0 else 2
    return all_gather


def comm_merge(run, comm, nDims, nDerived, want_rows=False, write=None, logzero=None):
    """this rank's run + everybody else's -> the merged result, all inside the library: records picked on the device,
    all-gathered over RCCL (comm = None: this process alone), merged on every rank (like an all-reduce)."""
    lib = _lib()
    own = run.get("_owner")
    if own is None:
        raise ValueError("comm_merge needs a result of _ctypes_api.run (the pchip_result travels with it)")
    m = Merged()
    want = 1 if (want_rows or write) else 0
    rc = lib.pchip_comm_merge(comm.h if comm is not None else None, C.byref(own.res), _logzero(run, logzero), nDims, nDerived, want, C.byref(m))
    if rc != 0:
        raise RuntimeError(f"pchip_comm_merge failed with code {rc}")
    try:
        if write:
            if lib.pchip_merged_write(C.byref(m), nDims, nDerived, str(write[0]).encode(), str(write[1]).encode()) != 0:
                raise RuntimeError("pchip_merged_write failed")
        return merged_dict(m, nDims, nDerived, want_rows)
    finally:
        lib.pchip_merged_free(C.byref(m))


def comm_merge_many(runs, comm, nDims, nDerived, want_rows=False, write=None, logzero=None):
    """this rank's runs (e.g. the R runs it made in step) + everybody else's -> the merged result of all of them (pchip_comm_merge_many)"""
    lib = _lib()
    n = len(runs)
    res = (api.Result * n)()
    for k, run in enumerate(runs):
        own = run.get("_owner")
        if own is None:
            raise ValueError("comm_merge_many needs results of _ctypes_api.run / run_repeats (the pchip_result travels with them)")
        C.memmove(C.byref(res[k]), C.byref(own.res), C.sizeof(api.Result))       # (a view: the blocks stay the owners')
    m = Merged()
    want = 1 if (want_rows or write) else 0
    rc = lib.pchip_comm_merge_many(comm.h if comm is not None else None, res, n, _logzero(runs[0], logzero), nDims, nDerived, want, C.byref(m))
    if rc != 0:
        raise RuntimeError(f"pchip_comm_merge_many failed with code {rc}")
    try:
        if write:
            if lib.pchip_merged_write(C.byref(m), nDims, nDerived, str(write[0]).encode(), str(write[1]).encode()) != 0:
                raise RuntimeError("pchip_merged_write failed")
        return merged_dict(m, nDims, nDerived, want_rows)
    finally:
        lib.pchip_merged_free(C.byref(m))


def gather_records(run, dist, torch, device, logzero=None, with_own=False):
    """The exchange on torch tensors (tests: gloo between ranks that share a GPU, or CPU only): counts first, then ONE padded
    [nmax][nTotal + 1] buffer per rank (rows | entry contour).  Returns (gathered [sum counts][nTotal + 1] tensor on
    `device`, counts).  dist = None: this rank's records alone.  with_own: one more column, the records' own log weights, and a third
    return value [(logZ, varlogZ, clustered)] per rank -- what pchip_merge_records_ex wants.  The product path between GPUs is comm_merge."""
    rows, entry = lived_records(run, logzero)
    cols = [torch.from_numpy(rows), torch.from_numpy(entry)[:, None]]
    if with_own:
        cols.append(torch.from_numpy(np.ascontiguousarray(run["logweights"][run["logweights"] > _logzero(run, logzero)]))[:, None])
    rec = torch.cat(cols, dim=1).to(device)
    mine = (float(run["logZ"]), float(run["varlogZ"]), clustered(run)) if with_own else None
    if dist is None:
        return (rec.contiguous(), [int(rec.shape[0])], [mine]) if with_own else (rec.contiguous(), [int(rec.shape[0])])
    world = dist.get_world_size()
    cnt = torch.tensor([rec.shape[0]], dtype=torch.int64, device=device)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt)
    ks = [int(c.item()) for c in cnts]
    nmax = max(max(ks), 1)
    pad = torch.zeros((nmax, rec.shape[1]), dtype=torch.float64, device=device)
    pad[:rec.shape[0]] = rec
    bufs = torch.empty((world * nmax, rec.shape[1]), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(bufs, pad)
    g = torch.cat([bufs[r * nmax:r * nmax + k] for r, k in enumerate(ks)]).contiguous()
    if not with_own:
        return g, ks
    ev = torch.tensor(list(mine), dtype=torch.float64, device=device)
    evs = [torch.zeros_like(ev) for _ in range(world)]
    dist.all_gather(evs, ev)
    return g, ks, [(float(e[0]), float(e[1]), int(e[2])) for e in evs]


def merge_runs(run, comm, nDims, nDerived, want_rows=False, write=None):
    """this rank's run (+ the other ranks' through `comm`) -> the merged result; what bench.py calls once per job"""
    return comm_merge(run, comm, nDims, nDerived, want_rows=want_rows, write=write)
