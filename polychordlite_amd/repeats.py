"""Independent repeats of one problem -- on one GPU, or spread over several (BASELINE's repeat-sharded mode).

A single run keeps about one wavefront per SIMD busy (B <= 1024 chains per nursery, the contraction on one CU), so
independent repeats overlap on one device; and they shard across devices with nothing to exchange until the end.
`run_repeats` is the front door: the library runs the repeats on host threads of its own (pchip_run_repeats: one engine
per run in flight, each on its own stream of its device) and merges their dead points on the first device
(pchip_merge_records): evidence with an error ~ 1/sqrt(repeats), posterior moments, the merged dead points and,
on request, <root>.stats / <root>_dead-birth.txt / <root>.txt of the union in the reference's formats.
"""
import ctypes as C

import numpy as np

from . import _ctypes_api as api
from . import merge as mg


def run_repeats(settings, like, prior, seeds, max_in_flight=4, devices=None, want_rows=False, write=None, comm=None, agree=None):
    """One nested-sampling run per seed, at most `max_in_flight` at a time per device, on `devices` (HIP ordinals of this
    process; None = the device of `settings`), merged.  comm (merge.Comm, one rank per GPU): every rank makes its own seeds' runs and
    the union of ALL ranks' runs is merged on every rank (pchip_comm_merge_many: one RCCL all-gather of the lived records).

    Returns (merged, runs): merged = {"n_runs", "logZ", "logZerr", "evidence_rule", "logZ_replay", "post_mean", "post_var", "logweights",
    "nlive", "records", "nlike", "nlike_local", "t_runs_s", "t_merge_s"[, "rows"]}; runs = the per-seed result dicts of `_ctypes_api.run`
    (this rank's).

    agree (with comm): `agree(ok) -> bool`, a collective of the caller's (e.g. an all-reduce of a failure flag over torch.distributed) that
    EVERY rank calls exactly once between its local runs and the exchange.  A rank whose runs failed must not leave the others waiting in
    the all-gather: when any rank reports a failure every rank raises here, before the exchange, and the collectives stay matched."""
    seeds = [int(s) for s in seeds]
    if not seeds:
        raise ValueError("run_repeats needs at least one seed")
    lib = mg._lib()
    f = lib.pchip_run_repeats_ex
    f.restype = C.c_int
    f.argtypes = [C.POINTER(api.Settings), C.POINTER(api.Like), C.POINTER(api.Prior), C.c_int, C.POINTER(C.c_int), C.c_int,
                  C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(api.Result), C.POINTER(mg.Merged)]
    n = len(seeds)
    sd = (C.c_int * n)(*seeds)
    devs = list(devices) if devices else []
    dv = (C.c_int * max(len(devs), 1))(*devs) if devs else None
    res = (api.Result * n)()
    m = mg.Merged()
    import time
    t0 = time.perf_counter()
    if comm is not None and not settings.device_records:
        # (the runs leave their lived records on the device for the exchange: no second trip over the host link)
        s2 = api.Settings(); C.memmove(C.byref(s2), C.byref(settings), C.sizeof(settings)); s2.device_records = 1
        settings = s2
    rc = f(C.byref(settings), C.byref(like), C.byref(prior), n, sd, len(devs), dv, int(max_in_flight), 1 if (want_rows or write) else 0, res,
           C.byref(m) if comm is None else None)
    if comm is not None and agree is not None:
        if not agree(rc == 0):
            if rc == 0:                                               # (a failed call has freed its results itself)
                for k in range(n):
                    lib.pchip_result_free(C.byref(res[k]))
            raise RuntimeError(f"run_repeats: the runs of at least one rank failed (this rank: code {rc}); the exchange was skipped on every rank")
    if rc != 0:
        raise RuntimeError(f"pchip_run_repeats failed with code {rc}")
    t_runs = time.perf_counter() - t0
    runs = []
    for k in range(n):
        r = api.Result()
        C.memmove(C.byref(r), C.byref(res[k]), C.sizeof(r))          # each dict owns (and frees) its own result block
        runs.append(api.result_dict(r, settings))
    if comm is None:
        try:
            if write:
                if lib.pchip_merged_write(C.byref(m), settings.nDims, settings.nDerived, str(write[0]).encode(), str(write[1]).encode()) != 0:
                    raise RuntimeError("pchip_merged_write failed")
            merged = mg.merged_dict(m, settings.nDims, settings.nDerived, want_rows)
        finally:
            lib.pchip_merged_free(C.byref(m))
    else:
        merged = mg.comm_merge_many(runs, comm, settings.nDims, settings.nDerived, want_rows=want_rows, write=write)
        merged["t_runs_s"] = t_runs
    merged["nlike_local"] = int(sum(r["nlike"] for r in runs))
    merged["runs_logZ"] = [r["logZ"] for r in runs]      # (their mean and its error: merged["runs_logZ_mean"], ["runs_logZ_sem"], from the library)
    return merged, runs
